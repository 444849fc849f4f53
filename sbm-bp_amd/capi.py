"""ctypes binding of the C ABI declared in include/sbmbp.h. Fails loudly when the HIP library is
missing — there is no fallback implementation behind these symbols."""
import ctypes as C
import os

from sbm_bp_amd.build import lib_path

c_dp = C.POINTER(C.c_double)
c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_uint64)
c_i32p = C.POINTER(C.c_int32)


class SbmbpError(RuntimeError):
    def __init__(self, code, what, detail):
        super().__init__("sbmbp: %s (code %d)%s" % (what, code, (": " + detail) if detail else ""))
        self.code = code


class InferResult(C.Structure):
    _fields_ = [("entropy", C.c_double), ("free_energy", C.c_double), ("overlap", C.c_double),
                ("niter", C.c_int), ("last_maxdiff", C.c_double)]


class LearnResult(C.Structure):
    _fields_ = [("em_steps", C.c_int), ("status", C.c_int), ("free_energy", C.c_double), ("overlap", C.c_double),
                ("total_sweeps", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("sweeps", C.c_uint64), ("edge_msg_updates", C.c_uint64), ("sweep_kernel_ms", C.c_double),
                ("sweep_launches", C.c_uint64), ("bytes_per_sweep", C.c_double), ("device_bytes", C.c_uint64),
                ("n_blocks", C.c_uint32), ("n_hub_rows", C.c_uint32), ("psi_form_sweeps", C.c_uint64),
                ("hub_edges", C.c_uint64)]


class DistInfo(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("n_global", C.c_uint32), ("row0", C.c_uint32), ("n_own", C.c_uint32),
                ("n_halo", C.c_uint32), ("n_chunks", C.c_uint32), ("halo_components", C.c_uint32), ("n_edges", C.c_uint64),
                ("e2_global", C.c_uint64), ("n_halo_msgs", C.c_uint64), ("sent_rows_per_sweep", C.c_uint64),
                ("busiest_peer_rows", C.c_uint64)]


class ConvState(C.Structure):
    _fields_ = [("maxdiff", C.c_double), ("conv_iter", C.c_int), ("sweep_idx", C.c_int), ("stop", C.c_int),
                ("last_exact", C.c_int), ("pause", C.c_int), ("ar_field_level", C.c_int), ("ar_generic_level", C.c_int)]


class ShardDesc(C.Structure):
    _fields_ = [("n_global", C.c_uint32), ("n_own", C.c_uint32), ("n_halo", C.c_uint32), ("row0", C.c_uint32),
                ("n_edges", C.c_uint64), ("edge0", C.c_uint64), ("row_ptr", c_u64p), ("nbr_local", c_u32p),
                ("psi_buf0", C.c_void_p), ("psi_buf1", C.c_void_p), ("red_buf", C.c_void_p), ("n_chunks", C.c_uint32),
                ("chunk_row", c_u32p), ("rev_local", c_u32p), ("n_halo_msgs", C.c_uint64), ("table_deg", c_u32p)]


COMM_ID_BYTES = 256


class CommCallbacks(C.Structure):
    EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, c_dp, c_u64p, c_dp, c_u64p, C.c_int)
    ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, c_dp, C.c_uint64, c_dp)
    ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, c_dp, C.c_uint64, C.c_int)
    _fields_ = [("user", C.c_void_p), ("exchange", EXCHANGE), ("allgather", ALLGATHER), ("allreduce", ALLREDUCE)]


_D = C.c_void_p  # sbmbp_dist_t*
# every symbol include/sbmbp.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sbmbp_strerror": (C.c_char_p, [C.c_int]),
    "sbmbp_last_error": (C.c_char_p, []),
    "sbmbp_version": (C.c_char_p, []),
    "sbmbp_device_count": (C.c_int, []),
    "sbmbp_graph_load_edgelist": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_uint32]),
    "sbmbp_graph_from_edges": (C.c_int, [C.POINTER(C.c_void_p), c_u32p, C.c_uint64, C.c_uint32]),
    "sbmbp_graph_from_csr": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint64, c_u64p, c_u32p, c_u32p]),
    "sbmbp_graph_num_vertices": (C.c_uint32, [C.c_void_p]),
    "sbmbp_graph_num_directed_edges": (C.c_uint64, [C.c_void_p]),
    "sbmbp_graph_max_degree": (C.c_uint32, [C.c_void_p]),
    "sbmbp_graph_copy_csr": (C.c_int, [C.c_void_p, c_u64p, c_u32p, c_u32p]),
    "sbmbp_graph_destroy": (None, [C.c_void_p]),
    "sbmbp_param_from_epsilon_c": (C.c_int, [C.c_uint32, C.c_uint32, C.c_double, C.c_double, c_dp, c_u32p]),
    "sbmbp_param_from_direct": (C.c_int, [C.c_uint32, C.c_uint32, c_dp, c_dp, c_dp, c_u32p]),
    "sbmbp_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    "sbmbp_destroy": (None, [C.c_void_p]),
    "sbmbp_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sbmbp_init_messages": (C.c_int, [C.c_void_p, C.c_uint32, c_i32p, c_u32p, C.c_uint32, C.c_int]),
    "sbmbp_host_init_state": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, c_i32p, C.c_uint32, c_dp, c_dp]),
    "sbmbp_init_messages_device": (C.c_int, [C.c_void_p, C.c_uint64, c_u32p]),
    "sbmbp_set_params": (C.c_int, [C.c_void_p, c_dp, c_u32p, C.c_double]),
    "sbmbp_get_params": (C.c_int, [C.c_void_p, c_dp, c_u32p]),
    "sbmbp_set_state": (C.c_int, [C.c_void_p, c_dp, c_dp]),
    "sbmbp_get_state": (C.c_int, [C.c_void_p, c_dp, c_dp]),
    "sbmbp_get_field": (C.c_int, [C.c_void_p, c_dp]),
    "sbmbp_set_schedule": (C.c_int, [C.c_void_p, C.c_double, C.c_uint32]),
    "sbmbp_set_gather_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "sbmbp_set_auto_relax": (C.c_int, [C.c_void_p, C.c_int]),
    "sbmbp_get_relaxation": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), c_dp, c_dp]),
    "sbmbp_set_learning_schedule": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "sbmbp_converge": (C.c_int, [C.c_void_p, C.c_double, C.c_uint32, C.c_double, C.POINTER(C.c_int), c_dp]),
    "sbmbp_sweep": (C.c_int, [C.c_void_p, C.c_double, C.c_uint32, c_dp]),
    "sbmbp_free_energy": (C.c_int, [C.c_void_p, c_dp, c_dp]),
    "sbmbp_entropy": (C.c_int, [C.c_void_p, c_dp, c_dp]),
    "sbmbp_set_nonedge_mode": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "sbmbp_em_expectations": (C.c_int, [C.c_void_p, c_dp, c_dp, c_dp]),
    "sbmbp_confusion": (C.c_int, [C.c_void_p, c_dp]),
    "sbmbp_overlap": (C.c_int, [C.c_void_p, c_dp]),
    "sbmbp_inference": (C.c_int, [C.c_void_p, C.c_float, C.c_uint32, C.c_float, C.POINTER(InferResult)]),
    "sbmbp_learning": (C.c_int, [C.c_void_p, C.c_float, C.c_uint32, C.c_float, C.c_float, C.POINTER(LearnResult)]),
    "sbmbp_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "sbmbp_reset_stats": (C.c_int, [C.c_void_p]),
    "sbmbp_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    # shard steps (sbm-bp_amd/distributed.py); desc/state structs are declared there
    "sbmbp_shard_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(ShardDesc), C.c_uint32, C.c_uint32, C.c_int]),
    "sbmbp_shard_begin": (C.c_int, [C.c_void_p, C.c_double, C.c_int]),
    "sbmbp_shard_set_labels": (C.c_int, [C.c_void_p, c_i32p, c_u32p, C.c_uint32, C.c_int, C.c_int]),
    "sbmbp_shard_query": (C.c_int, [C.c_void_p, C.c_int]),
    "sbmbp_shard_sweep_explicit": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double]),
    "sbmbp_shard_msg_halo": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "sbmbp_shard_pack_msgs": (C.c_int, [C.c_void_p, C.c_uint32, c_u32p, C.c_uint32, c_dp]),
    "sbmbp_shard_set_incoming": (C.c_int, [C.c_void_p, C.c_int]),
    "sbmbp_shard_state_record": (C.c_int, [C.c_void_p, C.c_int]),
    "sbmbp_shard_state_wait": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(ConvState)]),
    "sbmbp_shard_resume": (C.c_int, [C.c_void_p]),
    "sbmbp_shard_pack": (C.c_int, [C.c_void_p, C.c_uint32, c_u32p, C.c_uint32, c_dp, C.c_uint32]),
    "sbmbp_shard_nonedge_exact_partial": (C.c_int, [C.c_void_p, c_dp, C.c_int]),
    "sbmbp_shard_set_io": (C.c_int, [C.c_void_p, c_u32p, c_u32p, c_dp, c_dp, c_dp, C.c_uint32]),
    "sbmbp_shard_unpack": (C.c_int, [C.c_void_p, C.c_uint32, c_dp, c_u32p, C.c_uint32, C.c_uint32]),
    "sbmbp_shard_read_buffer": (C.c_int, [C.c_void_p, C.c_uint32]),
    "sbmbp_shard_field_partial": (C.c_int, [C.c_void_p, C.c_uint32]),
    "sbmbp_shard_sweep_partial": (C.c_int, [C.c_void_p, C.c_uint32]),
    "sbmbp_shard_sweep_chunk": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "sbmbp_shard_sweep_chunk_on": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "sbmbp_shard_sweep_fold": (C.c_int, [C.c_void_p]),
    "sbmbp_shard_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int]),
    "sbmbp_shard_msgdiff_partial": (C.c_int, [C.c_void_p]),
    "sbmbp_shard_rowsums_partial": (C.c_int, [C.c_void_p]),
    "sbmbp_shard_fe_partial": (C.c_int, [C.c_void_p, C.c_int]),
    "sbmbp_shard_fe_finish": (C.c_int, [C.c_void_p, c_dp]),
    "sbmbp_shard_nonedge_partial": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "sbmbp_shard_nonedge_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dp]),
    "sbmbp_shard_em_partial": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "sbmbp_shard_em_finish": (C.c_int, [C.c_void_p, c_dp, c_dp, c_dp]),
    "sbmbp_shard_poll": (C.c_int, [C.c_void_p, C.POINTER(ConvState)]),
    "sbmbp_shard_commit": (C.c_int, [C.c_void_p, C.c_uint32]),
    # multi-GPU: communicators and the per-rank driver
    "sbmbp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "sbmbp_comm_init_rank": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "sbmbp_comm_init_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "sbmbp_comm_init_callbacks": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(CommCallbacks)]),
    "sbmbp_comm_init_null": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "sbmbp_comm_destroy": (None, [C.c_void_p]),
    "sbmbp_comm_abort": (None, [C.c_void_p]),
    "sbmbp_comm_rank": (C.c_int, [C.c_void_p]),
    "sbmbp_comm_size": (C.c_int, [C.c_void_p]),
    "sbmbp_comm_transport": (C.c_char_p, [C.c_void_p]),
    "sbmbp_dist_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]),
    "sbmbp_dist_destroy": (None, [_D]),
    "sbmbp_dist_info": (C.c_int, [_D, C.POINTER(DistInfo)]),
    "sbmbp_dist_peer_rows": (C.c_int, [_D, c_u64p, c_u64p]),
    "sbmbp_dist_init_messages": (C.c_int, [_D, C.c_uint32, c_i32p, c_u32p, C.c_uint32, C.c_int]),
    "sbmbp_dist_init_messages_device": (C.c_int, [_D, C.c_uint64, c_u32p]),
    "sbmbp_dist_set_state": (C.c_int, [_D, c_dp, c_dp]),
    "sbmbp_dist_get_state": (C.c_int, [_D, c_dp, c_dp]),
    "sbmbp_dist_gather_marginals": (C.c_int, [_D, c_dp]),
    "sbmbp_dist_set_params": (C.c_int, [_D, c_dp, c_u32p, C.c_double]),
    "sbmbp_dist_get_params": (C.c_int, [_D, c_dp, c_u32p]),
    "sbmbp_dist_set_schedule": (C.c_int, [_D, C.c_double, C.c_uint32]),
    "sbmbp_dist_set_learning_schedule": (C.c_int, [_D, C.c_double, C.c_double]),
    "sbmbp_dist_set_gather_mode": (C.c_int, [_D, C.c_int]),
    "sbmbp_dist_set_auto_relax": (C.c_int, [_D, C.c_int]),
    "sbmbp_dist_get_relaxation": (C.c_int, [_D, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sbmbp_dist_converge": (C.c_int, [_D, C.c_double, C.c_uint32, C.c_double, C.POINTER(C.c_int), c_dp]),
    "sbmbp_dist_sweep": (C.c_int, [_D, C.c_double, C.c_uint32, c_dp]),
    "sbmbp_dist_free_energy": (C.c_int, [_D, c_dp, c_dp]),
    "sbmbp_dist_entropy": (C.c_int, [_D, c_dp, c_dp]),
    "sbmbp_dist_em_expectations": (C.c_int, [_D, c_dp, c_dp, c_dp]),
    "sbmbp_dist_confusion": (C.c_int, [_D, c_dp]),
    "sbmbp_dist_overlap": (C.c_int, [_D, c_dp]),
    "sbmbp_dist_inference": (C.c_int, [_D, C.c_float, C.c_uint32, C.c_float, C.POINTER(InferResult)]),
    "sbmbp_dist_learning": (C.c_int, [_D, C.c_float, C.c_uint32, C.c_float, C.c_float, C.POINTER(LearnResult)]),
    "sbmbp_dist_get_stats": (C.c_int, [_D, C.POINTER(Stats)]),
    "sbmbp_dist_reset_stats": (C.c_int, [_D]),
    "sbmbp_dist_set_timing": (C.c_int, [_D, C.c_int]),
    "sbmbp_dist_phase_times": (C.c_int, [_D, c_dp, c_u64p]),
    "sbmbp_plan_summary": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.POINTER(DistInfo), c_u64p, c_u64p, c_u64p]),
    "sbmbp_plan_arrays": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, c_u32p, c_u32p, c_u32p, c_u64p, c_u64p, c_u32p, c_u32p,
                                    c_u32p, c_u32p, c_u32p]),
}

_LIB = None


def load_library():
    """dlopen csrc/libsbmbp_hip.so (built in-tree by sbm_bp_amd.build) and type every entry point."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # In a process that also uses PyTorch, torch's bundled HIP runtime must be loaded FIRST: our library
    # then binds to that same runtime by soname (libamdhip64.so.7) and device pointers are shared.
    # Loaded in the other order the process would hold two HIP runtimes and torch would see no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise SbmbpError(-3, "native library missing", path + " (run __graft_entry__.build(); there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(code):
    if code != 0:
        lib = load_library()
        raise SbmbpError(code, lib.sbmbp_strerror(code).decode(), lib.sbmbp_last_error().decode())
