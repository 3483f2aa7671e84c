"""ctypes bindings for the two product libraries.  Loading fails loudly when a library is missing;
there is no fallback path of any kind."""
import ctypes as C
import os

from ._build import kernels_lib_path, host_lib_path, harness_lib_path

vp = C.c_void_p
sz = C.c_size_t
dbl = C.c_double
i32 = C.c_int
pi32 = C.POINTER(C.c_int)
pdbl = C.POINTER(C.c_double)

# name -> argtypes (every function returns int unless listed in _RESTYPES)
KERNEL_API = {
    "mi355x_error_string": [i32],
    "mi355x_device_count": [pi32],
    "mi355x_set_device": [i32],
    "mi355x_get_device": [pi32],
    "mi355x_device_name": [C.c_char_p, sz],
    "mi355x_device_synchronize": [],
    "mi355x_mem_info": [C.POINTER(sz), C.POINTER(sz)],
    "mi355x_handle_create": [C.POINTER(vp)],
    "mi355x_handle_destroy": [vp],
    "mi355x_handle_synchronize": [vp],
    "mi355x_handle_wait_result": [vp],
    "mi355x_handle_stream": [vp],
    "mi355x_handle_host_scratch": [vp],
    "mi355x_handle_device_scratch": [vp],
    "mi355x_graph_capture_begin": [vp],
    "mi355x_graph_capture_end": [vp, C.POINTER(vp)],
    "mi355x_graph_launch": [vp, vp],
    "mi355x_graph_destroy": [vp],
    "mi355x_malloc": [C.POINTER(vp), sz],
    "mi355x_free": [vp],
    "mi355x_host_malloc": [C.POINTER(vp), sz],
    "mi355x_host_free": [vp],
    "mi355x_memcpy_h2d": [vp, vp, vp, sz],
    "mi355x_memcpy_d2h": [vp, vp, vp, sz],
    "mi355x_memcpy_d2d": [vp, vp, vp, sz],
    "mi355x_memset": [vp, vp, i32, sz],
    "mi355x_event_create": [C.POINTER(vp)],
    "mi355x_event_destroy": [vp],
    "mi355x_event_record": [vp, vp],
    "mi355x_event_synchronize": [vp],
    "mi355x_event_elapsed_ms": [vp, vp, C.POINTER(C.c_float)],
    "mi355x_handle_wait_event": [vp, vp],
    "mi355x_vec_set": [vp, sz, dbl, vp],
    "mi355x_vec_copy": [vp, sz, vp, vp],
    "mi355x_vec_scale": [vp, sz, dbl, vp],
    "mi355x_vec_swap": [vp, sz, vp, vp],
    "mi355x_vec_axpy": [vp, sz, dbl, vp, vp],
    "mi355x_vec_aypx": [vp, sz, dbl, vp, vp],
    "mi355x_vec_axpby": [vp, sz, dbl, dbl, vp, vp],
    "mi355x_vec_waxpy": [vp, sz, dbl, vp, vp, vp],
    "mi355x_vec_axpbypcz": [vp, sz, dbl, dbl, dbl, vp, vp, vp],
    "mi355x_vec_pointwise_mult": [vp, sz, vp, vp, vp],
    "mi355x_vec_pointwise_divide": [vp, sz, vp, vp, vp],
    "mi355x_vec_reciprocal": [vp, sz, vp],
    "mi355x_vec_jacobi_invert": [vp, sz, vp, vp],
    "mi355x_vec_maxpy": [vp, sz, i32, pdbl, C.POINTER(vp), vp],
    "mi355x_vec_dot": [vp, sz, vp, vp, vp],
    "mi355x_vec_norm": [vp, sz, i32, vp, vp],
    "mi355x_vec_dotnorm2": [vp, sz, vp, vp, vp],
    "mi355x_vec_mdot": [vp, sz, i32, vp, C.POINTER(vp), vp],
    "mi355x_vec_maxpy_dev_norm2": [vp, sz, i32, vp, C.c_double, C.POINTER(vp), vp, vp],
    "mi355x_vec_scale_rnorm_dev": [vp, sz, vp, vp],
    "mi355x_vec_cg_update": [vp, sz, dbl, vp, vp, vp, vp, vp, vp, vp],
    "mi355x_vec_aypx_dev": [vp, sz, vp, dbl, vp, vp],
    "mi355x_vec_pmult_dot": [vp, sz, vp, vp, vp, vp, vp],
    "mi355x_vec_pmult_dotnorm2": [vp, sz, vp, vp, vp, vp, vp],
    "mi355x_vec_bcgs_update": [vp, sz, dbl, dbl, vp, vp, vp, vp, vp, vp, vp],
    "mi355x_handle_publish": [vp, vp, i32],
    "mi355x_handle_publish_at": [vp, vp, i32, i32],
    "mi355x_vec_cg_update_dev": [vp, sz, dbl, vp, dbl, i32, vp, vp, vp, vp, vp, vp, vp, i32],
    "mi355x_pbjacobi_apply": [vp, i32, i32, vp, vp, vp],
    "mi355x_spmv_bsr4_mfma": [vp, i32, i32, vp, vp, vp, vp, vp],
    "mi355x_trisolve_plan_create": [vp, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(vp)],
    "mi355x_trisolve_plan_create_ordered": [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, C.POINTER(vp)],
    "mi355x_trisolve_plan_create_scaled": [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(vp)],
    "mi355x_trisolve_plan_destroy": [vp],
    "mi355x_trisolve_apply": [vp, vp, vp, vp, vp],
    "mi355x_trisolve_aborted": [vp, pi32],
    "mi355x_trisolve_debug_get": [vp, i32, vp, sz, C.POINTER(sz)],
    "mi355x_spmv_plan_create": [vp, i32, vp, vp, C.POINTER(vp)],
    "mi355x_spmv_plan_destroy": [vp],
    "mi355x_spmv_plan_compress_indices": [vp, vp, vp, vp],
    "mi355x_spmv_plan_use_patterns": [vp, i32, vp],
    "mi355x_spmv_plan_value_patterns": [vp, vp, vp, vp, vp, vp],
    "mi355x_spmv_plan_drop_value_patterns": [vp],
    "mi355x_spmv_plan_use_value_patterns": [vp, i32, vp],
    "mi355x_spmv_plan_is_compressed": [vp, pi32],
    "mi355x_spmv_plan_dot_available": [vp, vp, pi32],
    "mi355x_spmv_plan_group_rows": [vp, vp, vp, vp, i32, vp],
    "mi355x_spmv_plan_set_pairsum": [vp, i32],
    "mi355x_spmv_plan_group_info": [vp, pi32, C.POINTER(C.c_long), pi32],
    "mi355x_spmv_plan_info": [vp, pi32, pi32, C.POINTER(sz)],
    "mi355x_spmv_csr": [vp, vp, vp, vp, vp, vp, vp],
    "mi355x_spmv_csr_add": [vp, vp, vp, vp, vp, vp, vp, vp],
    "mi355x_spmv_csr_scaled": [vp, vp, vp, vp, vp, vp, vp, vp],
    "mi355x_spmv_csr_dot": [vp, vp, vp, vp, vp, vp, vp],
    "mi355x_spmv_dot_finish": [vp, vp, vp],
    "mi355x_spmv_tiled_probe": [i32, vp, vp, C.POINTER(dbl)],
    "mi355x_spmv_tiled_build": [i32, i32, vp, vp, i32, C.POINTER(vp)],
    "mi355x_spmv_tiled_info": [vp, C.POINTER(C.c_long), C.POINTER(C.c_long), pi32, pi32, C.POINTER(C.c_long)],
    "mi355x_spmv_tiled_geometry": [pi32, pi32, pi32, pi32],
    "mi355x_spmv_tiled_upload": [vp, vp, vp],
    "mi355x_spmv_tiled_refresh_values": [vp, vp, vp],
    "mi355x_spmv_tiled": [vp, vp, vp, vp, vp],
    "mi355x_spmv_tiled_parts": [vp, vp, vp, vp, vp, i32],
    "mi355x_spmv_tiled_debug_get": [vp, i32, vp, sz, C.POINTER(sz)],
    "mi355x_spmv_tiled_drop_host": [vp],
    "mi355x_spmv_tiled_destroy": [vp],
    "mi355x_host_threads": [i32],
    "mi355x_csr_get_diagonal": [vp, i32, vp, vp, vp, vp],
    "mi355x_csr_diagonal_scale": [vp, i32, vp, vp, vp, vp, vp],
    "mi355x_csr_assemble": [vp, i32, vp, vp, vp, vp, vp],
    "mi355x_spmv_bsr": [vp, i32, i32, vp, vp, vp, vp, vp],
    "mi355x_spmv_bsr_planned": [vp, vp, i32, vp, vp, vp, vp, vp],
    "mi355x_spmv_bsr_planned_add": [vp, vp, i32, vp, vp, vp, vp, vp, vp],
    "mi355x_spmv_bsr_planned_form": [vp, vp, i32, i32, vp, vp, vp, vp, vp],
    "mi355x_ilu0_lower_level": [vp, i32, vp, vp, vp, vp, vp, vp],
    "mi355x_ilu0_upper_level": [vp, i32, vp, vp, vp, vp, vp],
    "mi355x_pack": [vp, sz, vp, vp, vp],
    "mi355x_unpack_insert": [vp, sz, vp, vp, vp],
    "mi355x_unpack_add": [vp, sz, vp, vp, vp],
    "mi355x_unpack_max": [vp, sz, vp, vp, vp],
    "mi355x_stream_triad": [vp, sz, dbl, vp, vp, vp],
    # mi355x_comm.h
    "mi355x_comm_get_unique_id": [C.c_char_p],
    "mi355x_comm_init_rank": [C.POINTER(vp), i32, i32, C.c_char_p],
    "mi355x_comm_destroy": [vp],
    "mi355x_comm_rank": [vp, pi32, pi32],
    "mi355x_comm_error_string": [i32],
    "mi355x_comm_allreduce_sum": [vp, vp, vp, sz],
    "mi355x_comm_allreduce_max": [vp, vp, vp, sz],
    "mi355x_comm_group_start": [],
    "mi355x_comm_group_end": [],
    "mi355x_comm_send": [vp, vp, vp, sz, i32],
    "mi355x_comm_recv": [vp, vp, vp, sz, i32],
}
_RESTYPES = {
    "mi355x_error_string": C.c_char_p,
    "mi355x_comm_error_string": C.c_char_p,
    "mi355x_handle_stream": vp,
    "mi355x_handle_host_scratch": vp,
    "mi355x_handle_device_scratch": vp,
}

_kernels = None
_host = None


def load_kernels():
    """dlopen csrc/libmi355x_kernels.so (RTLD_GLOBAL so the host library resolves against it)."""
    global _kernels
    if _kernels is None:
        path = kernels_lib_path()
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, args in KERNEL_API.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _kernels = lib
    return _kernels


_harness = None


def load_harness():
    """dlopen harness/libpetscharness.so (the stand-in PETSc; RTLD_GLOBAL so the plugin resolves against it)."""
    global _harness
    if _harness is None:
        path = harness_lib_path()
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        _harness = C.CDLL(path, mode=C.RTLD_GLOBAL)
    return _harness


def load_host():
    """dlopen host/libpetschipmi355x.so (the plugin); signatures are declared by petsc-dev_amd/petsc.py."""
    global _host
    if _host is None:
        load_kernels()
        load_harness()
        path = host_lib_path()
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        _host = C.CDLL(path, mode=C.RTLD_GLOBAL)
    return _host
