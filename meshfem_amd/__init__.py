"""meshfem_amd: MI355X-native per-element stiffness assembly + PCG solve for MeshFEM.

Python mirror of the reference's interface for this path (LinearElasticity::Simulator,
SPSDSystem, ElasticityTensor), on top of the C ABI in include/meshfem_hip.h
(libmeshfem_hip.so, hand-written HIP for gfx950). There is no CPU fallback."""
from ._lib import (MeshFEMHipError, ASSEMBLE_GATHER, ASSEMBLE_ATOMIC, NEUMANN_TRACTION, NEUMANN_PRESSURE,
                   NEUMANN_FORCE, PRECOND_BLOCK_JACOBI, PRECOND_JACOBI, PRECOND_NONE, PRECOND_TWO_LEVEL, PRECOND_MULTIGRID, PRECOND_AUTO, LIB_PATH,
                   OP_ELASTICITY, OP_LAPLACIAN, OP_MASS, SOLVE_PIN, SOLVE_NO_RIGID_MOTION, SOLVE_ALLOW_ILL_POSED)
from .core import Context, device_cache_trim, device_cache_stats, device_arena_stats, device_reserve, device_reserve_for, context_bytes_estimate
from .linear_elasticity import Simulator
from .tensors import ElasticityTensor, ElasticityTensor2D, ElasticityTensor3D
from . import homogenization

__all__ = ["Context", "Simulator", "ElasticityTensor", "ElasticityTensor2D", "ElasticityTensor3D", "homogenization",
           "MeshFEMHipError", "LIB_PATH"]
