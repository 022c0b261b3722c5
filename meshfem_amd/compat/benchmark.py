"""`benchmark` (src/python_bindings/benchmark.cc: the `_benchmark` module over GlobalBenchmark.hh / Timer.hh): named
hierarchical wall-clock timers -- reset, start/stop_timer_section, start/stop_timer, report, to_dict. Host-side timers
only; the device phases of a context are in `Context.timing()` / `Simulator.benchmarkReport()` and, with MFH_ROCTX=1, in
rocprofv3's marker trace under the same section names."""
import sys
import time

_sections = {}          # name -> dict(elapsed, started, timers: name -> dict(time, started))
_stack = []


def reset():
    _sections.clear()
    del _stack[:]


def start_timer_section(name):
    s = _sections.setdefault(name, dict(elapsed=0.0, started=None, timers={}))
    if s["started"] is not None:
        raise RuntimeError("Timer section '%s' already running" % name)
    s["started"] = time.perf_counter()
    _stack.append(name)


def stop_timer_section(name):
    s = _sections.get(name)
    if s is None or s["started"] is None:
        raise RuntimeError("Timer section '%s' is not running" % name)
    s["elapsed"] += time.perf_counter() - s["started"]
    s["started"] = None
    if name in _stack:
        _stack.remove(name)


def _current():
    if not _stack:
        start_timer_section("")            # timers outside every section live in the unnamed root, like g_timer
    return _sections[_stack[-1]]


def start_timer(name):
    t = _current()["timers"].setdefault(name, dict(time=0.0, started=None))
    if t["started"] is not None:
        raise RuntimeError("Timer '%s' already running" % name)
    t["started"] = time.perf_counter()


def stop_timer(name):
    for sec in reversed([_sections[n] for n in _stack] or list(_sections.values())):
        t = sec["timers"].get(name)
        if t is not None and t["started"] is not None:
            t["time"] += time.perf_counter() - t["started"]
            t["started"] = None
            return
    raise RuntimeError("Timer '%s' is not running" % name)


def to_dict():
    return {n: (s["elapsed"], {k: t["time"] for k, t in s["timers"].items()}) for n, s in _sections.items()}


def report(include_messages=False, out=None):
    out = out or sys.stdout
    for n, (el, timers) in to_dict().items():
        out.write("%s\t%.6f\n" % (n or "(root)", el))
        for k, t in timers.items():
            out.write("    %s\t%.6f\n" % (k, t))
