"""ctypes binding of libmeshfem_hip.so (include/meshfem_hip.h). There is no fallback: if the
shared object is missing this module raises, and if there is no HIP device `Context()` raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MESHFEM_HIP_LIB") or os.path.join(_HERE, "libmeshfem_hip.so")   # override: A/B runs of two builds of the library

OK, ERR_INVALID, ERR_STATE, ERR_HIP, ERR_NOT_CONVERGED, ERR_UNSUPPORTED = range(6)
ASSEMBLE_GATHER, ASSEMBLE_ATOMIC = 0, 1
NEUMANN_TRACTION, NEUMANN_PRESSURE, NEUMANN_FORCE = 0, 1, 2
PRECOND_BLOCK_JACOBI, PRECOND_JACOBI, PRECOND_NONE, PRECOND_TWO_LEVEL, PRECOND_MULTIGRID, PRECOND_AUTO = 0, 1, 2, 3, 4, 5
OP_ELASTICITY, OP_LAPLACIAN, OP_MASS = 0, 1, 2
SOLVE_PIN, SOLVE_NO_RIGID_MOTION, SOLVE_ALLOW_ILL_POSED = 1, 2, 4


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("rel_residual", C.c_double),
                ("true_rel_residual", C.c_double), ("solve_ms", C.c_double), ("setup_ms", C.c_double),
                ("used_graph", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DistStats(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("transport", C.c_int32), ("peer_enabled", C.c_int32),
                ("halo_nodes_sent", C.c_int64), ("halo_nodes_received", C.c_int64), ("halo_bytes_per_exchange", C.c_int64),
                ("interior_items", C.c_int64), ("boundary_items", C.c_int64),
                ("exchange_ms", C.c_double), ("interior_ms", C.c_double), ("boundary_ms", C.c_double), ("exposed_wait_ms", C.c_double),
                ("operator_ms", C.c_double), ("profiled_applications", C.c_int32), ("reserved", C.c_int32), ("exchanges", C.c_int64),
                ("peer_halo_messages", C.c_int64), ("peer_halo_bytes", C.c_int64), ("allreduces_small", C.c_int64),
                ("allreduces_large", C.c_int64), ("fallback_exchanges", C.c_int64), ("fallback_allreduces", C.c_int64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["transport_name"] = {0: "none", 1: "RCCL send/recv", 2: "peer copies (HIP IPC)", 3: "caller callbacks"}.get(d["transport"], "?")
        return d


class Timing(C.Structure):
    _fields_ = [("symbolic_ms", C.c_double), ("geometry_ms", C.c_double), ("assemble_ms", C.c_double),
                ("upload_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_P = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double
_pi64, _pi32 = C.POINTER(C.c_int64), C.POINTER(C.c_int32)

# name -> (restype, argtypes). Mirrors include/meshfem_hip.h one to one (tests check the symbol list).
PROTOTYPES = {
    "mfh_create": (_i32, [_i32, C.POINTER(_P)]),
    "mfh_destroy": (None, [_P]),
    "mfh_last_error": (C.c_char_p, [_P]),
    "mfh_version": (C.c_char_p, []),
    "mfh_stream": (_P, [_P]),
    "mfh_set_stream": (_i32, [_P, _P]),
    "mfh_mesh_build": (_i32, [_P, _i32, _i32, _i64, _i64, _P, _P]),
    "mfh_mesh_set": (_i32, [_P, _i32, _i32, _i64, _i64, _i64, _P, _P]),
    "mfh_mesh_sizes": (_i32, [_P, _pi64, _pi64, _pi64, _pi64, _pi64, _pi32, _pi32]),
    "mfh_mesh_get_elem_nodes": (_i32, [_P, _P]),
    "mfh_mesh_get_node_positions": (_i32, [_P, _P]),
    "mfh_mesh_get_boundary_elem_nodes": (_i32, [_P, _P]),
    "mfh_mesh_get_boundary_nodes": (_i32, [_P, _P]),
    "mfh_mesh_get_boundary_elem_geometry": (_i32, [_P, _P, _P]),
    "mfh_mesh_get_elem_volumes": (_i32, [_P, _P]),
    "mfh_material_const": (_i32, [_P, _P]),
    "mfh_material_isotropic": (_i32, [_P, _f64, _f64]),
    "mfh_material_iso_field": (_i32, [_P, _P, _P]),
    "mfh_material_ortho_field": (_i32, [_P, _P]),
    "mfh_material_tensor_field": (_i32, [_P, _P]),
    "mfh_material_get": (_i32, [_P, _i64, _P]),
    "mfh_dof_map": (_i32, [_P, _P, _i64]),
    "mfh_apply_periodic_conditions": (_i32, [_P, _f64, _pi64]),
    "mfh_get_dof_map": (_i32, [_P, _P, _pi64]),
    "mfh_dof_map_partitioned": (_i32, [_P, _P, _i64, _i64]),
    "mfh_assemble": (_i32, [_P, _i32]),
    "mfh_symbolic": (_i32, [_P, _i32]),
    "mfh_symbolic_sizes": (_i32, [_P, _pi64, _pi64, _pi32, _pi32]),
    "mfh_symbolic_get": (_i32, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "mfh_matrix_info": (_i32, [_P, _pi64, _pi64, _pi64]),
    "mfh_matrix_storage": (_i32, [_P, _pi32, _pi64]),
    "mfh_export_bsr": (_i32, [_P, _P, _P, _P]),
    "mfh_export_upper_triplets": (_i32, [_P, _P, _P, _P, C.POINTER(C.c_uint64)]),
    "mfh_element_stiffness": (_i32, [_P, _i64, _i64, _P]),
    "mfh_clear_fixed": (_i32, [_P]),
    "mfh_fix_variables": (_i32, [_P, _i64, _P, _P]),
    "mfh_set_preconditioner": (_i32, [_P, _i32]),
    "mfh_precond_info": (_i32, [_P, _pi32, _pi64, C.POINTER(_f64), C.POINTER(C.c_char_p)]),
    "mfh_precond_choice": (_i32, [_P, _P, _P, _P]),
    "mfh_multigrid_info": (_i32, [_P, _pi64, _pi64, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
    "mfh_multigrid_level_info": (_i32, [_P, _i32, _pi64, _pi32]),
    "mfh_solve": (_i32, [_P, _i32, _P, _P, _f64, _i32, C.POINTER(SolveInfo)]),
    "mfh_solve_batch": (_i32, [_P, _i32, _P, _P, _f64, _i32, C.POINTER(SolveInfo)]),
    "mfh_apply_K": (_i32, [_P, _P, _P]),
    "mfh_bc_clear": (_i32, [_P]),
    "mfh_bc_dirichlet_box": (_i32, [_P, _P, _P, _i32, _P, _i32]),
    "mfh_bc_neumann_box": (_i32, [_P, _P, _P, _i32, _P, _i32]),
    "mfh_bc_delta_force": (_i32, [_P, _i64, _P]),
    "mfh_bc_dirichlet_nodes": (_i32, [_P, _i64, _P, _P, _i32]),
    "mfh_bc_neumann_elements": (_i32, [_P, _i64, _P, _P]),
    "mfh_bc_dirichlet_vars": (_i32, [_P, _P, _P, _pi64]),
    "mfh_pin_node": (_i32, [_P, _pi64]),
    "mfh_neumann_load": (_i32, [_P, _P]),
    "mfh_constant_strain_load": (_i32, [_P, _P, _P]),
    "mfh_sim_solve": (_i32, [_P, _P, _i32, _P, _f64, _i32, C.POINTER(SolveInfo)]),
    "mfh_average_strain": (_i32, [_P, _P, _P]),
    "mfh_average_stress": (_i32, [_P, _P, _P]),
    "mfh_integrated_stress": (_i32, [_P, _P, _P, _P]),
    "mfh_strain_field": (_i32, [_P, _P, _i32, _P]),
    "mfh_boundary_strain_field": (_i32, [_P, _P, _i32, _P]),
    "mfh_mesh_get_boundary_elem_parents": (_i32, [_P, _P]),
    "mfh_mesh_get_boundary_elem_internal": (_i32, [_P, _P]),
    "mfh_mesh_update_vertices": (_i32, [_P, _P]),
    "mfh_sim_solve_constrained": (_i32, [_P, _P, _i32, _P, _i32, _P, _f64, _i32, C.POINTER(SolveInfo)]),
    "mfh_sim_solve_batch": (_i32, [_P, _i32, _P, _i32, _P, _f64, _i32, C.POINTER(SolveInfo)]),
    "mfh_solve_cell_problems": (_i32, [_P, _i32, _P, _i32, _P, _f64, _i32, C.POINTER(SolveInfo)]),
    "mfh_matrix_free_info": (_i32, [_P, _P, _P, _P, _P, _P, _P]),
    "mfh_set_operator": (_i32, [_P, _i32]),
    "mfh_matrix_set_upper_triplets": (_i32, [_P, _i64, _i64, _P, _P, _P]),
    "mfh_average_gradient": (_i32, [_P, _P, _P]),
    "mfh_apply_delta_K": (_i32, [_P, _P, _P, _P]),
    "mfh_delta_constant_strain_load": (_i32, [_P, _P, _P, _P]),
    "mfh_delta_average_strain": (_i32, [_P, _P, _P, _P, _i32, _P]),
    "mfh_mutual_energies": (_i32, [_P, _P, _P, _P]),
    "mfh_mutual_energy_differential": (_i32, [_P, _P, _P]),
    "mfh_dev_spmv": (_i32, [_P, _P, _P]),
    "mfh_dev_precond": (_i32, [_P, _P, _P]),
    "mfh_tl_partitioned_begin": (_i32, [_P, _i32, _P, _P, _P]),
    "mfh_tl_partitioned_finish": (_i32, [_P, _P]),
    "mfh_dev_tl_restrict": (_i32, [_P, _P, _P]),
    "mfh_dev_tl_apply": (_i32, [_P, _P, _P, _P]),
    "mfh_dev_pcg_update_xr": (_i32, [_P, _P, _P, _P, _P, _P, _P]),
    "mfh_dev_pcg_direction": (_i32, [_P, _P, _P, _P, _P]),
    "mfh_dev_dots": (_i32, [_P, _P, _P, _P]),
    "mfh_dev_mask_fixed": (_i32, [_P, _P]),
    "mfh_dev_set_fixed_values": (_i32, [_P, _P]),
    "mfh_dev_sync": (_i32, [_P]),
    "mfh_rccl_get_unique_id": (_i32, [_P]),
    "mfh_comm_create_rccl": (_i32, [_P, _P, _i32, _i32, C.POINTER(_P)]),
    "mfh_comm_create_callbacks": (_i32, [_i32, _i32, _P, _P, _P, C.POINTER(_P)]),
    "mfh_comm_destroy": (None, [_P]),
    "mfh_comm_describe": (C.c_char_p, [_P]),
    "mfh_comm_allreduce": (_i32, [_P, _P, _P, _i64]),
    "mfh_comm_selftest": (_i32, [_P, _P]),
    "mfh_comm_preflight": (_i32, [_P, _P, _i64, C.POINTER(_f64), _i64]),
    "mfh_device_cache_trim": (_i32, []),
    "mfh_device_cache_stats": (_i32, [_i32, _pi64, _pi64, _pi64, _pi64, _pi64]),
    "mfh_device_arena_stats": (_i32, [_i32, _pi64]),
    "mfh_device_reserve": (_i32, [_i32, _i64, _i32]),
    "mfh_device_reserve_for": (_i32, [_i32, _i32, _i32, _i64, _i32]),
    "mfh_context_bytes_estimate": (_i32, [_i32, _i32, _i64, _P, _P]),
    "mfh_comm_enable_peer": (_i32, [_P, _P]),
    "mfh_comm_disable_peer": (_i32, [_P, _P]),
    "mfh_dist_get_stats": (_i32, [_P, _P]),
    "mfh_dist_setup": (_i32, [_P, _P, _i32, _P, _P, _P, _P]),
    "mfh_dist_two_level": (_i32, [_P, _i32, _P, _P]),
    "mfh_dist_solve": (_i32, [_P, _i32, _P, _P, _f64, _i32, C.POINTER(SolveInfo)]),
    "mfh_dist_apply_K": (_i32, [_P, _P, _P]),
    "mfh_dev_memcpy": (_i32, [_P, _P, _P, _i64, _i32, _P]),
    "mfh_get_timing": (_i32, [_P, C.POINTER(Timing)]),
    "mfh_time_assembly_kernel": (_i32, [_P, _i32, _i32, C.POINTER(_f64)]),
    "mfh_time_spmv_kernel": (_i32, [_P, _i32, C.POINTER(_f64)]),
    "mfh_set_option": (_i32, [_P, C.c_char_p, _f64]),
    "mfh_debug_spd_inverse": (_i32, [_i64, _P]),
    "mfh_debug_spd_inverse_device": (_i32, [_P, _i64, _P]),
    "mfh_debug_device_node_tables": (_i32, [_P, _P, _P]),
    "mfh_placement_info": (_i32, [_P, _i32, C.POINTER(C.c_double), C.POINTER(_i32)]),
    "mfh_debug_arena_alloc": (_i32, [_P, _i64, C.POINTER(_P)]),
    "mfh_debug_arena_free": (_i32, [_P, _P]),
    "mfh_debug_row_chunks": (_i32, [_i64, _P, _i32, _i64, _P, _i64, _i32, _P, _i64, _P]),
}

# callback types of mfh_comm_create_callbacks
ALLREDUCE_FN = C.CFUNCTYPE(_i32, _P, _P, _i64, _P)
EXCHANGE_FN = C.CFUNCTYPE(_i32, _P, _i32, C.POINTER(_i32), C.POINTER(_P), C.POINTER(_i64), C.POINTER(_P), C.POINTER(_i64), _P)

_lib = None


def load():
    """Load libmeshfem_hip.so and declare every prototype. Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libmeshfem_hip.so not found at %s: build it with `python -m meshfem_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own libamdhip64; if torch is imported AFTER this library has
    # initialised /opt/rocm's runtime the process ends up with two HIP runtimes and torch reports
    # "No HIP GPUs are available". Loading torch first makes both share one runtime. (C/C++ users of
    # the .so are unaffected; opt out with MESHFEM_NO_TORCH_PRELOAD=1.)
    if os.environ.get("MESHFEM_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class MeshFEMHipError(RuntimeError):
    """The C ABI's status + message, raised like the reference's std::runtime_error."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def ptr(a):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def as_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)
