"""Build the compiled pybind11 modules `mesh`, `tensors`, `sparse_matrices`, `periodic_homogenization` (the names of the
reference's extension modules, src/python_bindings/CMakeLists.txt:10-33) in-tree, linked against ../libmeshfem_hip.so.
    import sys, meshfem_amd.pybind; sys.path.insert(0, meshfem_amd.pybind.PATH); import mesh, tensors, ..."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "src")
MODULES = ["tensors", "mesh", "sparse_matrices", "periodic_homogenization"]
SUFFIX = sysconfig.get_config_var("EXT_SUFFIX") or ".so"


def target(name):
    return os.path.join(HERE, name + SUFFIX)


def needs_build():
    deps = [os.path.join(SRC, "common.hh"), os.path.join(HERE, "..", "..", "include", "MeshFEMHip", "LinearElasticity.hh"),
            os.path.join(HERE, "..", "..", "include", "MeshFEMHip", "PeriodicHomogenization.hh"), os.path.join(HERE, "..", "..", "include", "meshfem_hip.h")]
    for n in MODULES:
        t = target(n)
        if not os.path.exists(t):
            return True
        mt = os.path.getmtime(t)
        if any(os.path.getmtime(d) > mt for d in deps + [os.path.join(SRC, n + ".cc")]):
            return True
    return False


def build(force=False, verbose=True):
    if not force and not needs_build():
        return [target(n) for n in MODULES]
    import pybind11
    inc = ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]]
    lib_dir = os.path.dirname(HERE)
    procs = []
    for n in MODULES:
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"] + inc + \
              [os.path.join(SRC, n + ".cc"), "-o", target(n), "-L" + lib_dir, "-lmeshfem_hip", "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    return [target(n) for n in MODULES]


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv)))
