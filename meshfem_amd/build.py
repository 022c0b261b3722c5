"""Build libmeshfem_hip.so (HIP/gfx950) in-tree with hipcc. No GPU is needed to build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmeshfem_hip.so")
SOURCES = ["mfh_api.cpp", "mfh_pool.cpp", "mfh_solver.cpp", "mfh_simulator.cpp", "mfh_mesh.cpp", "mfh_symbolic.cpp", "mfh_twolevel.cpp", "mfh_multigrid.cpp", "mfh_symbolic_gpu.hip", "mfh_kernels.hip", "mfh_kernels_solver.hip", "mfh_peer.hip"]
HOST_ONLY = ["mfh_twolevel.cpp"]
HEADERS = ["mfh_internal.hh", "mfh_ctx.hh", "mfh_device.hh", "mfh_comm.hh", os.path.join("..", "..", "include", "meshfem_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    """Compile every HIP/C++ source of the library for gfx950 and link the shared object."""
    if not force and not needs_build():
        return LIB
    objs = []
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
    common += os.environ.get("MFH_CXXFLAGS", "").split()      # experiments only (e.g. -DMFH_EXPERIMENTS); the default build has none
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        cmd = [_hipcc()] + common + ["-c", os.path.join(CSRC, s), "-o", obj]
        if s in HOST_ONLY:   # host code with dense inner loops: AVX2/FMA for the host pass only
            cmd = [_hipcc()] + common + ["-Xarch_host", "-mavx2", "-Xarch_host", "-mfma", "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
