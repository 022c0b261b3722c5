"""Modules named like the reference's pybind11 extension modules (src/python_bindings/CMakeLists.txt:10-33)
so that scripts written against them run on the MI355X path:

    import sys, meshfem_amd.compat; sys.path.insert(0, meshfem_amd.compat.PATH)
    import mesh, tensors, sparse_matrices, periodic_homogenization, differential_operators

Only the surface that touches the hot path is provided: `mesh.{Mesh, PeriodicCondition,
MSHFieldWriter, MSHFieldParser}`,
`tensors.ElasticityTensor{2,3}D`, `sparse_matrices.{Triplet, TripletMatrix, SuiteSparseMatrix, SPSDSystem}`,
`periodic_homogenization.{homogenize, probe}`, `differential_operators.{laplacian, mass, mass_elasticity, bilaplacian,
gradient}`. They are pure-Python shims over the C ABI (ctypes), not
compiled pybind11 modules; viewers, filters, optimisers and the other bound modules are out of scope."""
import os

PATH = os.path.dirname(os.path.abspath(__file__))
