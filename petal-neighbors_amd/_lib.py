"""ctypes binding of libpetal_mi355x.so (include/petal_mi355x.h).

The library is the product: if it is missing or does not export a declared
symbol this module raises -- there is no Python/NumPy fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PN_LIBRARY_PATH: a diagnostic build (build.py with PN_DIAG_FLAGS writes libpetal_mi355x_diag.so) for tools/ only
LIB_PATH = os.environ.get("PN_LIBRARY_PATH") or os.path.join(_HERE, "libpetal_mi355x.so")

PN_OK, PN_ERR_EMPTY, PN_ERR_NOT_CONTIGUOUS, PN_ERR_INVALID, PN_ERR_DEVICE, PN_ERR_NOMEM, \
    PN_ERR_UNSUPPORTED, PN_ERR_EMPTY_MATRIX, PN_ERR_COMM = range(9)
PN_COMM_ID_BYTES = 128
PN_ENGINE_AUTO, PN_ENGINE_EXACT, PN_ENGINE_MFMA, PN_ENGINE_BF16 = 0, 1, 2, 3
PN_OPT_ENGINE, PN_OPT_SEGMENTS, PN_OPT_INDEX_BASE, PN_OPT_PROFILE, PN_OPT_FILTER_SLOTS = 1, 2, 3, 4, 5
PN_OPT_MFMA_STRUCTURE = 6
PN_OPT_EXCHANGE_ALWAYS = 7
PN_OPT_SHARED_THRESHOLDS = 8
PN_OPT_BF16_WAVES = 9
PN_OPT_SEED_MODEL = 10


class PnInfo(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("dim", C.c_uint64), ("row_stride_device", C.c_uint64),
                ("elem_bytes", C.c_int32), ("device", C.c_int32), ("mfma_eligible", C.c_int32),
                ("bf16_eligible", C.c_int32), ("bf16_layout", C.c_int32), ("seed_model", C.c_int32)]


class PnStats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("fallback_queries", C.c_uint64), ("candidates", C.c_uint64),
                ("hot_launches", C.c_uint64), ("hot_ms", C.c_double), ("last_call_ms", C.c_double),
                ("radius_results", C.c_uint64), ("evaluations", C.c_uint64),
                ("shard_ms", C.c_double), ("exchange_ms", C.c_double), ("reserved", C.c_uint64 * 1)]


class PnShardedInfo(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("dim", C.c_uint64), ("local_first_row", C.c_uint64),
                ("local_rows", C.c_uint64), ("n_shards", C.c_int32), ("world", C.c_int32),
                ("local_shards", C.c_int32), ("rank", C.c_int32), ("mfma_eligible", C.c_int32),
                ("bf16_eligible", C.c_int32)]


_sz, _ssz, _i, _vp, _u64 = C.c_size_t, C.c_ssize_t, C.c_int, C.c_void_p, C.c_uint64
_u64p, _f32p, _f64p = C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_double)

# every symbol include/petal_mi355x.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "pn_last_error": (C.c_char_p, []),
    "pn_strerror": (C.c_char_p, [_i]),
    "pn_abi_version": (_i, []),
    "pn_device_count": (_i, [C.POINTER(_i)]),
    "pn_index_create_f32": (_i, [_vp, _sz, _sz, _ssz, _ssz, _i, C.POINTER(_vp)]),
    "pn_index_create_f64": (_i, [_vp, _sz, _sz, _ssz, _ssz, _i, C.POINTER(_vp)]),
    "pn_index_create_cosine_f32": (_i, [_vp, _sz, _sz, _ssz, _ssz, _i, C.POINTER(_vp)]),
    "pn_index_create_cosine_f64": (_i, [_vp, _sz, _sz, _ssz, _ssz, _i, C.POINTER(_vp)]),
    "pn_pairwise_device_f32": (_i, [_vp, _sz, _sz, _sz, _i, _vp, _vp]),
    "pn_pairwise_device_f64": (_i, [_vp, _sz, _sz, _sz, _i, _vp, _vp]),
    "pn_index_create_device_f32": (_i, [_vp, _sz, _sz, _sz, _i, _vp, C.POINTER(_vp)]),
    "pn_index_create_device_f64": (_i, [_vp, _sz, _sz, _sz, _i, _vp, C.POINTER(_vp)]),
    "pn_index_destroy": (None, [_vp]),
    "pn_index_info": (_i, [_vp, C.POINTER(PnInfo)]),
    "pn_index_set_option": (_i, [_vp, _i, C.c_int64]),
    "pn_index_get_stats": (_i, [_vp, C.POINTER(PnStats), _i]),
    "pn_query_f32": (_i, [_vp, _vp, _sz, _sz, _ssz, _sz, _vp, _vp]),
    "pn_query_f64": (_i, [_vp, _vp, _sz, _sz, _ssz, _sz, _vp, _vp]),
    "pn_query_device_f32": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp]),
    "pn_query_device_f64": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp]),
    "pn_query_nearest_f32": (_i, [_vp, _vp, _sz, _sz, _ssz, _vp, _vp]),
    "pn_query_nearest_f64": (_i, [_vp, _vp, _sz, _sz, _ssz, _vp, _vp]),
    "pn_query_radius_f32": (_i, [_vp, _vp, _sz, _sz, _ssz, C.c_float, _vp, C.POINTER(_vp)]),
    "pn_query_radius_device_f32": (_i, [_vp, _vp, _sz, _sz, _sz, C.c_float, _vp, _vp, _sz, _vp, _vp]),
    "pn_query_radius_device_f64": (_i, [_vp, _vp, _sz, _sz, _sz, C.c_double, _vp, _vp, _sz, _vp, _vp]),
    "pn_query_radius_f64": (_i, [_vp, _vp, _sz, _sz, _ssz, C.c_double, _vp, C.POINTER(_vp)]),
    "pn_free": (None, [_vp]),
    "pn_pairwise_f32": (_i, [_vp, _sz, _sz, _ssz, _i, _vp]),
    "pn_pairwise_f64": (_i, [_vp, _sz, _sz, _ssz, _i, _vp]),
    "pn_pairwise_cosine_f32": (_i, [_vp, _sz, _sz, _ssz, _i, _vp]),
    "pn_pairwise_cosine_f64": (_i, [_vp, _sz, _sz, _ssz, _i, _vp]),
    "pn_cosine_f32": (C.c_float, [_vp, _sz, _vp, _sz]),
    "pn_cosine_f64": (C.c_double, [_vp, _sz, _vp, _sz]),
    "pn_euclidean_f32": (C.c_float, [_vp, _vp, _sz]),
    "pn_euclidean_f64": (C.c_double, [_vp, _vp, _sz]),
    "pn_reuclidean_f32": (C.c_float, [_vp, _vp, _sz]),
    "pn_reuclidean_f64": (C.c_double, [_vp, _vp, _sz]),
    "pn_rdistance_to_distance_f32": (C.c_float, [C.c_float]),
    "pn_rdistance_to_distance_f64": (C.c_double, [C.c_double]),
    "pn_distance_to_rdistance_f32": (C.c_float, [C.c_float]),
    "pn_distance_to_rdistance_f64": (C.c_double, [C.c_double]),
    "pn_merge_topk_device_f32": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _vp, _vp, _i, _vp]),
    "pn_merge_topk_device_f64": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _vp, _vp, _i, _vp]),
    "pn_fill_uniform_device_f32": (_i, [_vp, _u64, _u64, _u64, _i, _vp]),
    "pn_bf16_bounds_f32": (_i, [_vp, _vp, _sz, _sz, _ssz, _sz, _vp, _vp, _vp]),
    "pn_bf16_selftest": (_i, [_i, _vp]),
    "pn_debug_seed_model_feedback": (_i, [_vp, _i, _u64, _u64, _vp]),
    "pn_sharded_query_radius_device_f32": (_i, [_vp, _vp, _sz, _sz, _sz, C.c_float, _vp, _vp, _sz, _vp, _vp]),
    "pn_sharded_query_radius_device_f64": (_i, [_vp, _vp, _sz, _sz, _sz, C.c_double, _vp, _vp, _sz, _vp, _vp]),
    "pn_tree_num_nodes": (_i, [_vp, _u64p]),
    "pn_tree_children_of": (_i, [_vp, _u64, C.POINTER(_i), _u64p, _u64p]),
    "pn_tree_points_of": (_i, [_vp, _u64, C.POINTER(_u64p), _u64p]),
    "pn_tree_radius_of_f32": (_i, [_vp, _u64, _f32p]),
    "pn_tree_radius_of_f64": (_i, [_vp, _u64, _f64p]),
    "pn_tree_compare_nodes": (_i, [_vp, _u64, _u64, C.POINTER(_i)]),
    "pn_tree_node_distance_lower_bound_f32": (_i, [_vp, _u64, _u64, _f32p]),
    "pn_tree_node_distance_lower_bound_f64": (_i, [_vp, _u64, _u64, _f64p]),
    "pn_tree_centroid_of": (_i, [_vp, _u64, _vp]),
    "pn_comm_unique_id": (_i, [_vp]),
    "pn_sharded_create_f32": (_i, [_vp, _sz, _sz, _ssz, _ssz, C.POINTER(_i), _i, C.POINTER(_vp)]),
    "pn_sharded_create_rank_device_f32": (_i, [_vp, _sz, _sz, _sz, _u64, _i, _i, _vp, _i, _vp, C.POINTER(_vp)]),
    "pn_sharded_destroy": (None, [_vp]),
    "pn_sharded_info": (_i, [_vp, C.POINTER(PnShardedInfo)]),
    "pn_sharded_set_option": (_i, [_vp, _i, C.c_int64]),
    "pn_sharded_get_stats": (_i, [_vp, C.POINTER(PnStats), _i]),
    "pn_sharded_query_f32": (_i, [_vp, _vp, _sz, _sz, _ssz, _sz, _vp, _vp]),
    "pn_sharded_query_device_f32": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp]),
    "pn_sharded_query_radius_f32": (_i, [_vp, _vp, _sz, _sz, _ssz, C.c_float, _vp, C.POINTER(_vp)]),
    "pn_sharded_create_f64": (_i, [_vp, _sz, _sz, _ssz, _ssz, C.POINTER(_i), _i, C.POINTER(_vp)]),
    "pn_sharded_create_cosine_f32": (_i, [_vp, _sz, _sz, _ssz, _ssz, C.POINTER(_i), _i, C.POINTER(_vp)]),
    "pn_sharded_create_cosine_f64": (_i, [_vp, _sz, _sz, _ssz, _ssz, C.POINTER(_i), _i, C.POINTER(_vp)]),
    "pn_sharded_create_rank_device_f64": (_i, [_vp, _sz, _sz, _sz, _u64, _i, _i, _vp, _i, _vp, C.POINTER(_vp)]),
    "pn_sharded_query_f64": (_i, [_vp, _vp, _sz, _sz, _ssz, _sz, _vp, _vp]),
    "pn_sharded_query_device_f64": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp]),
    "pn_sharded_query_radius_f64": (_i, [_vp, _vp, _sz, _sz, _ssz, C.c_double, _vp, C.POINTER(_vp)]),
}

_lib = None


class LibraryMissing(RuntimeError):
    pass


def lib():
    """Load the HIP library. Raises LibraryMissing loudly; never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} is missing: build it with `python petal-neighbors_amd/build.py` "
            "(or __graft_entry__.build()). petal_neighbors_amd has no CPU fallback.")
    try:
        # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the HIP runtime already in
        # the process when ours is bound, or two runtimes would fight over the device.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(L, name)
        except AttributeError as e:
            raise LibraryMissing(f"{LIB_PATH} does not export {name}") from e
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def last_error() -> str:
    return lib().pn_last_error().decode("utf-8", "replace")
