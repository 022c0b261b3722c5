"""Compiled pybind11 modules named like the reference's extension modules (src/python_bindings/CMakeLists.txt:10-33):
`mesh`, `tensors`, `sparse_matrices`, `periodic_homogenization`, over the C ABI / the C++ facade of libmeshfem_hip.

    import sys, meshfem_amd.pybind; sys.path.insert(0, meshfem_amd.pybind.PATH)
    import mesh, tensors, sparse_matrices, periodic_homogenization

`meshfem_amd.pybind.build.build()` compiles them in-tree (g++, pybind11 headers); `__graft_entry__.build()` does so too.
They are the ONE implementation of the Python binding surface (the pure-Python shims of rounds 1-3 were retired in round 4)."""
import os

PATH = os.path.dirname(os.path.abspath(__file__))
