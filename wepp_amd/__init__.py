"""wepp_amd -- MI355X-native parsimonious read placement (the usher_mapper /
usher_common hot path of TurakhiaLab/WEPP) behind a C-ABI shared library.

The compute path is the HIP library `libwepp_place.so` (include/wepp_place.h);
this package is only the ctypes binding used by the tests and bench.py.
"""
from .api import (A, C, G, T, N, PLAN_WALK8, PLAN_WALK16, PLAN_SWEEP, PLAN_WALKC8, PLAN_WALKC16, PLAN_WIN, PLAN_SEED, PLAN_NAMES, WINDOW_CROWN_LEVELS, WINDOW_CROWN_SLOT, EppReads, FitchPlan, FlatView, GenTree, Mat, fitch_last_timing, PlacementResult, Reads, Tree, epp_last_timing, fitch_sites, flatten_count, generate_tree,
                  pack_read_word, unpack_read_word)
from ._lib import WeppError, LIB_PATH

__all__ = ["A", "C", "G", "T", "N", "PLAN_WALK8", "PLAN_WALK16", "PLAN_SWEEP", "PLAN_WALKC8", "PLAN_WALKC16", "PLAN_WIN", "PLAN_SEED", "PLAN_NAMES", "WINDOW_CROWN_LEVELS", "WINDOW_CROWN_SLOT", "EppReads", "FitchPlan", "fitch_last_timing", "FlatView", "GenTree", "Mat", "PlacementResult", "Reads", "Tree", "epp_last_timing",
           "fitch_sites", "flatten_count", "generate_tree", "pack_read_word", "unpack_read_word", "WeppError", "LIB_PATH"]
