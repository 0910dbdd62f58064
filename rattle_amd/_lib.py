"""ctypes binding of librattle_hip.so (include/rattle_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C rattle_amd/csrc`.
There is no fallback: if the shared object is missing, or no HIP device is present when a
context is created, the call fails loudly.
"""
from __future__ import annotations

import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")      # before the HIP runtime starts: one hardware queue per POA column class (csrc/poa.hip)
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RATTLE_HIP_LIB", os.path.join(_HERE, "csrc", "librattle_hip.so"))


class ClusterParams(C.Structure):
    _fields_ = [("t_s", C.c_double), ("t_v", C.c_double), ("bv_threshold", C.c_double),
                ("min_bv_threshold", C.c_double), ("bv_falloff", C.c_double), ("min_reads_cluster", C.c_int),
                ("use_hc", C.c_int), ("repr_percentile", C.c_double), ("is_rna", C.c_int)]


class ClusterSet(C.Structure):
    _fields_ = [("n_clusters", C.c_uint32), ("main_id", C.POINTER(C.c_int32)), ("main_rev", C.POINTER(C.c_uint8)),
                ("offsets", C.POINTER(C.c_uint32)), ("member_id", C.POINTER(C.c_int32)),
                ("member_rev", C.POINTER(C.c_uint8)), ("counters", C.c_uint64 * 8), ("gene_id", C.POINTER(C.c_int32))]


class CorrectParams(C.Structure):
    _fields_ = [("min_occ", C.c_double), ("gap_occ", C.c_double), ("err_ratio", C.c_double), ("split", C.c_int),
                ("min_reads", C.c_int), ("n_threads", C.c_int), ("vote_order", C.c_char * 8),
                ("n_pack_orders", C.c_uint32), ("pack_order_cluster", C.POINTER(C.c_uint32)),
                ("pack_order_offsets", C.POINTER(C.c_uint32)), ("pack_order_perm", C.POINTER(C.c_uint32)),
                ("max_pack_cells", C.c_uint64), ("corrected_ready", C.c_void_p), ("corrected_ready_user", C.c_void_p)]


class ReadSet(C.Structure):
    _fields_ = [("n", C.c_uint32), ("read_id", C.POINTER(C.c_int32)), ("cluster_id", C.POINTER(C.c_int32)),
                ("n_reads", C.POINTER(C.c_int32)), ("off", C.POINTER(C.c_uint64)), ("seq", C.POINTER(C.c_char)),
                ("qual", C.POINTER(C.c_char))]


class SkipList(C.Structure):
    _fields_ = [("n", C.c_uint32), ("cluster_id", C.POINTER(C.c_int32)), ("pack", C.POINTER(C.c_uint32)),
                ("stage", C.POINTER(C.c_uint32)), ("read_off", C.POINTER(C.c_uint64)), ("read_id", C.POINTER(C.c_int32))]


class Correction(C.Structure):
    _fields_ = [("corrected", ReadSet), ("uncorrected", ReadSet), ("consensi", ReadSet), ("counters", C.c_uint64 * 8),
                ("skipped", SkipList), ("corrected_pack", C.POINTER(C.c_uint32)), ("uncorrected_pack", C.POINTER(C.c_uint32))]


class PackPlan(C.Structure):
    _fields_ = [("n_packs", C.c_uint32), ("pack_first", C.POINTER(C.c_uint32)), ("member_id", C.POINTER(C.c_int32)),
                ("member_rev", C.POINTER(C.c_uint8)), ("pack_cluster", C.POINTER(C.c_int32)),
                ("pack_local", C.POINTER(C.c_uint32)), ("pack_cost", C.POINTER(C.c_uint64)),
                ("pack_owner", C.POINTER(C.c_uint32)), ("n_unqueued", C.c_uint32), ("unqueued_id", C.POINTER(C.c_int32)),
                ("unqueued_cluster", C.POINTER(C.c_int32))]


# int fn(void *user, const void *send, uint64 send_bytes, void *recv, const uint64 *recv_bytes)
ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64))


class MsaSet(C.Structure):
    _fields_ = [("n_packs", C.c_uint32), ("width", C.POINTER(C.c_uint32)), ("row_offset", C.POINTER(C.c_uint64)),
                ("rows", C.POINTER(C.c_char)), ("counters", C.c_uint64 * 8)]


_P = C.POINTER
_u8p, _u16p, _u32p, _u64p = _P(C.c_uint8), _P(C.c_uint16), _P(C.c_uint32), _P(C.c_uint64)
_i32p, _f64p = _P(C.c_int32), _P(C.c_double)

# name -> (restype, argtypes); kept in step with include/rattle_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "rattle_hip_last_error": (C.c_char_p, []),
    "rattle_hip_abi_version": (C.c_int, []),
    "rattle_hip_ctx_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "rattle_hip_ctx_create_host": (C.c_int, [_P(C.c_void_p)]),
    "rattle_hip_ctx_destroy": (None, [C.c_void_p]),
    "rattle_hip_load_reads": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_uint32, C.c_int, C.c_int]),
    "rattle_hip_get_read_index": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, _u32p, _i32p, _u64p, _u32p]),
    "rattle_hip_bv_filter": (C.c_int, [C.c_void_p, _u32p, C.c_uint32, _u32p, C.c_uint32, _u32p, _u16p, C.c_int, _u8p]),
    "rattle_hip_pair_score": (C.c_int, [C.c_void_p, _u32p, _u32p, _u8p, C.c_uint32, _i32p, _i32p, _i32p, _f64p, _i32p]),
    "rattle_hip_cluster_reads": (C.c_int, [C.c_void_p, _P(ClusterParams), _P(_P(ClusterSet))]),
    "rattle_hip_cluster_subset": (C.c_int, [C.c_void_p, _P(ClusterParams), _u32p, C.c_uint32, _P(_P(ClusterSet))]),
    "rattle_hip_cluster_subsets": (C.c_int, [C.c_void_p, _P(ClusterParams), _u32p, _u64p, C.c_uint32, _P(_P(ClusterSet)), C.c_int]),
    "rattle_hip_debug_phred_symbol": (C.c_int, [C.c_double, _P(C.c_int), _P(C.c_int)]),
    "rattle_hip_stage_reads": (C.c_int, [C.c_void_p, _u8p, _u8p, _u64p, C.c_uint32]),
    "rattle_hip_unstage_reads": (C.c_int, [C.c_void_p]),
    "rattle_hip_cluster_unsorted": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_uint32, C.c_int, _P(ClusterParams),
                                              _P(_P(ClusterSet))]),
    "rattle_hip_cluster_iso_unsorted": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_uint32, C.c_int, C.c_int, _P(ClusterParams),
                                                  _P(ClusterParams), _P(_P(ClusterSet)), _u32p]),
    "rattle_hip_cluster_set_free": (None, [_P(ClusterSet)]),
    "rattle_hip_poa_msa": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_uint32, _u32p, C.c_uint32, _P(_P(MsaSet))]),
    "rattle_hip_msa_set_free": (None, [_P(MsaSet)]),
    "rattle_hip_correct_reads": (C.c_int, [C.c_void_p, _u8p, _u8p, _u64p, C.c_uint32, C.c_uint32, _u32p, _i32p, _u8p,
                                           _P(CorrectParams), _P(_P(Correction))]),
    "rattle_hip_correction_free": (None, [_P(Correction)]),
    "rattle_hip_reserve_arena": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rattle_hip_set_exchange": (C.c_int, [C.c_void_p, C.c_int, C.c_int, ALLGATHERV_FN, C.c_void_p]),
    "rattle_hip_comm_unique_id": (C.c_int, [_u8p]),
    "rattle_hip_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _u8p]),
    "rattle_hip_comm_destroy": (C.c_int, [C.c_void_p]),
    "rattle_hip_comm_stats": (C.c_int, [C.c_void_p, _u64p, _u64p]),
    "rattle_hip_comm_probe": (C.c_int, [C.c_void_p]),
    "rattle_hip_correction_gather": (C.c_int, [C.c_void_p, _P(Correction), C.c_int, _P(_P(Correction))]),
    "rattle_hip_plan_packs": (C.c_int, [_u64p, C.c_uint32, C.c_uint32, _u32p, _i32p, _u8p, _P(CorrectParams), C.c_int,
                                        _P(_P(PackPlan))]),
    "rattle_hip_pack_plan_free": (None, [_P(PackPlan)]),
    "rattle_hip_lpt_assign": (C.c_int, [_u64p, C.c_uint32, C.c_int, _u32p]),
    "rattle_hip_kernel_stats": (C.c_int, [C.c_void_p, C.c_int, _f64p, _u64p, _u64p]),
    "rattle_hip_kernel_stats_reset": (C.c_int, [C.c_void_p]),
    "rattle_hip_stage_ms": (C.c_int, [C.c_void_p, _f64p, C.c_int]),
}

_lib = None


def load():
    """Load librattle_hip.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class RattleError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        raise RattleError(f"librattle_hip error {rc}: {load().rattle_hip_last_error().decode()}")
