"""ctypes loader for libwepp_place.so (the C-ABI of include/wepp_place.h).

There is no Python or CPU fallback: if the shared library has not been built
(`make -C wepp_amd/csrc`, or `__graft_entry__.build()`), importing this module
raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WEPP_PLACE_LIB lets an experiment load another build of the same library
LIB_PATH = os.environ.get("WEPP_PLACE_LIB") or os.path.join(_HERE, "libwepp_place.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first "
        "(make -C wepp_amd/csrc, or python -c 'import __graft_entry__ as g; g.build()'). "
        "wepp_amd has no CPU fallback."
    )

# libwepp_place.so and PyTorch both need libamdhip64.so.7.  PyTorch bundles its
# own copy; if ours (from /opt/rocm) were loaded first, torch would later be
# bound to a runtime it was not built with and report "No HIP GPUs".  Loading
# torch first makes the whole process share torch's HIP runtime.  A C/C++ host
# that links the library directly is not concerned.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

lib = ctypes.CDLL(LIB_PATH)

c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u8p = ctypes.POINTER(ctypes.c_uint8)


class TreeDescC(ctypes.Structure):
    _fields_ = [
        ("n_nodes", ctypes.c_uint32),
        ("parent", c_i32p),
        ("mut_off", c_u32p),
        ("mut_pos", c_i32p),
        ("mut_ref", c_u8p),
        ("mut_par", c_u8p),
        ("mut_mut", c_u8p),
    ]


class MatStats(ctypes.Structure):
    _fields_ = [
        ("n_nodes", ctypes.c_uint64),
        ("n_mutations", ctypes.c_uint64),
        ("n_masked", ctypes.c_uint64),
        ("n_events", ctypes.c_uint64),
        ("n_blocks", ctypes.c_uint64),
        ("n_leaves", ctypes.c_uint64),
        ("max_depth", ctypes.c_uint32),
        ("max_position", ctypes.c_uint32),
        ("stream_bytes", ctypes.c_uint64),
        ("device_bytes", ctypes.c_uint64),
        ("n_streams", ctypes.c_uint32),
        ("stream_tau", ctypes.c_int32 * 16),
        ("stream_nodes", ctypes.c_uint64 * 16),
        ("stream_bytes_of", ctypes.c_uint64 * 16),
        ("n_window_crowns", ctypes.c_uint32),
        ("window_crown_nodes", ctypes.c_uint64),
        ("n_window_streams", ctypes.c_uint32),
        ("n_window_streams_crown", ctypes.c_uint32),
        ("window_stream_nodes", ctypes.c_uint64),
        ("window_size", ctypes.c_uint32),
        ("window_stride", ctypes.c_uint32),
        ("window_uncovered_positions", ctypes.c_uint32),
        ("seed_chunks", ctypes.c_uint32),
        ("seed_chunk_blocks", ctypes.c_uint32),
        ("seed_sig_bytes", ctypes.c_uint64),
    ]


class GenTreeParams(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("n_nodes", ctypes.c_uint32),
        ("genome_len", ctypes.c_uint32),
        ("p_recent_parent", ctypes.c_double),
        ("zipf_s", ctypes.c_double),
        ("p_back_mutation", ctypes.c_double),
        ("p_ambiguous", ctypes.c_double),
        ("p_masked_node", ctypes.c_double),
        ("root_mutations", ctypes.c_uint32),
        ("depth_choices", ctypes.c_uint32),
        ("p_hub", ctypes.c_double),
        ("n_hubs", ctypes.c_uint32),
    ]


class GenTreeShape(ctypes.Structure):
    _fields_ = [
        ("n_nodes", ctypes.c_uint32), ("n_leaves", ctypes.c_uint32), ("max_depth", ctypes.c_uint32),
        ("max_children", ctypes.c_uint32), ("path_mutations_median", ctypes.c_uint32),
        ("path_mutations_p95", ctypes.c_uint32), ("path_mutations_max", ctypes.c_uint32),
        ("path_mutations_mean", ctypes.c_double), ("mutations_per_node", ctypes.c_double),
    ]


class GenReadsParams(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("n_reads", ctypes.c_uint32),
        ("read_len", ctypes.c_uint32),
        ("amplicon_len", ctypes.c_uint32),
        ("amplicon_step", ctypes.c_uint32),
        ("p_substitution", ctypes.c_double),
        ("p_n", ctypes.c_double),
        ("p_iupac", ctypes.c_double),
    ]


class EppReadsC(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_uint32), ("read_off", ctypes.c_void_p), ("read_word", ctypes.c_void_p),
                ("start", ctypes.c_void_p), ("end", ctypes.c_void_p), ("degree", ctypes.c_void_p)]


class EppOutC(ctypes.Structure):
    _fields_ = [("max_parsimony", ctypes.c_void_p), ("multiplicity", ctypes.c_void_p), ("epp_off", ctypes.c_void_p),
                ("epp_nodes", ctypes.c_void_p), ("epp_capacity", ctypes.c_uint64), ("hap_score", ctypes.c_void_p),
                ("hap_read_counts", ctypes.c_void_p), ("hap_divergence", ctypes.c_void_p)]


# every symbol include/wepp_place.h declares (tests/test_abi.py checks the list
# against the header)
_V = ctypes.c_void_p
_SIGS = {
    "wepp_mat_create": (ctypes.c_int, [ctypes.POINTER(TreeDescC), ctypes.c_int, ctypes.POINTER(_V)]),
    "wepp_mat_destroy": (ctypes.c_int, [_V]),
    "wepp_mat_get_stats": (ctypes.c_int, [_V, ctypes.POINTER(MatStats)]),
    "wepp_mat_bfs_order": (ctypes.c_int, [_V, _V]),
    "wepp_place_batch": (ctypes.c_int, [_V, _V, _V, ctypes.c_uint32, _V, _V, _V, _V, _V]),
    "wepp_imputed_mutations": (ctypes.c_int, [_V, _V, _V, ctypes.c_uint32, _V, _V, _V, _V, ctypes.c_uint64]),
    "wepp_best_nodes": (ctypes.c_int, [_V, _V, _V, ctypes.c_uint32, _V, _V, _V, _V, ctypes.c_uint64]),
    "wepp_excess_mutations": (ctypes.c_int, [_V, _V, _V, ctypes.c_uint32, ctypes.c_uint32, _V, _V, _V, _V, _V, _V, _V,
                                             ctypes.c_uint64]),
    "wepp_place_batch_device": (
        ctypes.c_int,
        [_V, _V, _V, ctypes.c_uint32, ctypes.c_uint64, _V, _V, _V, _V, _V],
    ),
    "wepp_mat_set_tile_reads": (ctypes.c_int, [_V, ctypes.c_uint32]),
    "wepp_mat_set_use_crowns": (ctypes.c_int, [_V, ctypes.c_int]),
    "wepp_mat_set_use_walk": (ctypes.c_int, [_V, ctypes.c_int]),
    "wepp_mat_set_pipeline": (ctypes.c_int, [_V, ctypes.c_uint32]),
    "wepp_mat_set_use_seeds": (ctypes.c_int, [_V, ctypes.c_int]),
    "wepp_mat_last_seeds": (ctypes.c_int, [_V] + [ctypes.POINTER(ctypes.c_uint64)] * 4 + [_V]),
    "wepp_mat_last_tiers": (ctypes.c_int, [_V, _V, ctypes.c_uint32]),
    "wepp_mat_last_plans": (ctypes.c_int, [_V, _V, _V, ctypes.c_uint32]),
    "wepp_mat_last_crowns": (ctypes.c_int, [_V, _V, _V, ctypes.c_uint32]),
    "wepp_mat_last_walk": (ctypes.c_int, [_V, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "wepp_mat_timing_reset": (ctypes.c_int, [_V]),
    "wepp_mat_last_timing": (
        ctypes.c_int,
        [_V, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64),
         ctypes.POINTER(ctypes.c_uint64)],
    ),
    "wepp_fitch_sites": (ctypes.c_int, [ctypes.POINTER(TreeDescC), ctypes.c_int, ctypes.c_uint32, _V, _V, _V, _V,
                                        ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), _V, _V, _V, _V]),
    "wepp_fitch_plan_create": (ctypes.c_int, [ctypes.POINTER(TreeDescC), ctypes.c_int, ctypes.POINTER(_V)]),
    "wepp_fitch_plan_run": (ctypes.c_int, [_V, ctypes.c_uint32, _V, _V, _V, _V, ctypes.c_uint64,
                                           ctypes.POINTER(ctypes.c_uint64), _V, _V, _V, _V]),
    "wepp_fitch_plan_destroy": (ctypes.c_int, [_V]),
    "wepp_fitch_last_timing": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double)] * 4),
    "wepp_epp_map": (ctypes.c_int, [_V, ctypes.POINTER(EppReadsC), ctypes.c_uint32, ctypes.c_uint32,
                                    ctypes.POINTER(EppOutC)]),
    "wepp_epp_fetch_lists": (ctypes.c_int, [_V, _V, ctypes.c_uint64]),
    "wepp_mat_dfs_order": (ctypes.c_int, [_V, _V]),
    "wepp_epp_last_timing": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double)] * 4 + [
        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32),
        ctypes.POINTER(ctypes.c_uint32)]),
    "wepp_last_error": (ctypes.c_char_p, []),
    "wepp_gen_tree_create": (ctypes.c_int, [ctypes.POINTER(GenTreeParams), ctypes.POINTER(_V)]),
    "wepp_gen_tree_desc": (ctypes.c_int, [_V, ctypes.POINTER(TreeDescC)]),
    "wepp_gen_tree_get_shape": (ctypes.c_int, [_V, ctypes.POINTER(GenTreeShape)]),
    "wepp_gen_tree_destroy": (ctypes.c_int, [_V]),
    "wepp_gen_reads_create": (ctypes.c_int, [_V, ctypes.POINTER(GenReadsParams), ctypes.POINTER(_V)]),
    "wepp_gen_reads_get": (
        ctypes.c_int,
        [_V, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(c_u32p), ctypes.POINTER(c_u32p)],
    ),
    "wepp_gen_reads_windows": (ctypes.c_int, [_V, ctypes.POINTER(c_i32p), ctypes.POINTER(c_i32p)]),
    "wepp_gen_reads_destroy": (ctypes.c_int, [_V]),
    "wepp_flat_create": (ctypes.c_int, [ctypes.POINTER(TreeDescC), ctypes.POINTER(_V)]),
    "wepp_flat_get": (
        ctypes.c_int,
        [_V, ctypes.c_char_p, ctypes.POINTER(_V), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)],
    ),
    "wepp_flat_scalars": (ctypes.c_int, [_V, ctypes.POINTER(MatStats), ctypes.POINTER(ctypes.c_uint32)]),
    "wepp_flat_destroy": (ctypes.c_int, [_V]),
    "wepp_flat_save": (ctypes.c_int, [_V, ctypes.c_char_p]),
    "wepp_flat_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_V)]),
    "wepp_mat_upload": (ctypes.c_int, [_V, ctypes.c_int, ctypes.POINTER(_V)]),
    "wepp_debug_flatten_count": (ctypes.c_uint64, []),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the .so is stale
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = sorted(_SIGS)


class WeppError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"wepp error {code}: {msg}")
        self.code = code


def check(rc):
    if rc != 0:
        raise WeppError(rc, lib.wepp_last_error().decode("utf-8", "replace"))
