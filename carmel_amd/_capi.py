"""ctypes binding of libcarmel_hip.so (include/carmel_hip.h).  The library is the product; this module is glue.

There is deliberately no fallback: if the shared library is missing or a symbol is absent the import fails, and
every compute entry point needs a GPU (carmel_hip_create refuses to run without one).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CARMEL_HIP_LIB") or os.path.join(_HERE, "libcarmel_hip.so")  # override: A/B builds

# every symbol include/carmel_hip.h declares (tests/test_capi_symbols.py checks the header against this list)
SYMBOLS = [
    "carmel_hip_last_error", "carmel_hip_device_count", "carmel_hip_create", "carmel_hip_destroy",
    "carmel_hip_set_corpus", "carmel_hip_build_lattices", "carmel_hip_set_norm", "carmel_hip_set_prior",
    "carmel_hip_set_cascade", "carmel_hip_normalize", "carmel_hip_set_weights", "carmel_hip_get_weights",
    "carmel_hip_get_arc_weights", "carmel_hip_estimate", "carmel_hip_estimate_async",
    "carmel_hip_estimate_finish", "carmel_hip_counts_dev", "carmel_hip_counts_len", "carmel_hip_stream",
    "carmel_hip_use_external_counts", "carmel_hip_synchronize", "carmel_hip_last_sweep_ms", "carmel_hip_read_scalars",
    "carmel_hip_get_counts", "carmel_hip_set_counts", "carmel_hip_maximize", "carmel_hip_keep_em_weights", "carmel_hip_random_restart", "carmel_hip_save_counts",
    "carmel_hip_fractional_counts", "carmel_hip_set_digamma",
    "carmel_hip_save_best", "carmel_hip_load_best", "carmel_hip_host_build", "carmel_hip_host_dims",
    "carmel_hip_host_export", "carmel_hip_host_export_lanes", "carmel_hip_host_export_waves", "carmel_hip_host_transpose", "carmel_hip_host_tile_sweep", "carmel_hip_host_free",
    "carmel_hip_gibbs_create", "carmel_hip_gibbs_destroy", "carmel_hip_gibbs_n_blocks", "carmel_hip_gibbs_lattice_stats", "carmel_hip_gibbs_max_sample",
    "carmel_hip_gibbs_run", "carmel_hip_gibbs_run_ex", "carmel_hip_gibbs_set_prior_inference", "carmel_hip_gibbs_prior_trace",
    "carmel_hip_gibbs_n_prior_scales", "carmel_hip_gibbs_set_run_share", "carmel_hip_gibbs_best_stats", "carmel_hip_forests_set_prior_inference", "carmel_hip_forests_prior_trace", "carmel_hip_gibbs_get_sample", "carmel_hip_gibbs_set_observer", "carmel_hip_gibbs_current_probs", "carmel_hip_gibbs_get_state", "carmel_hip_gibbs_final_counts", "carmel_hip_gibbs_uniform", "carmel_hip_gibbs_power", "carmel_hip_gibbs_best_run", "carmel_hip_gibbs_set_init_weights",
    "carmel_hip_forests_create", "carmel_hip_forests_destroy", "carmel_hip_forests_estimate",
    "carmel_hip_forests_get_counts", "carmel_hip_forests_maximize", "carmel_hip_forests_get_weights",
    "carmel_hip_forests_set_weights", "carmel_hip_forests_set_alphas", "carmel_hip_forests_gibbs", "carmel_hip_forests_best_run", "carmel_hip_forests_final_counts", "carmel_hip_forests_get_sample",
    "carmel_hip_forests_max_sample", "carmel_hip_forests_viterbi", "carmel_hip_forests_get_viterbi",
    "carmel_hip_compose", "carmel_hip_composition_states", "carmel_hip_composition_arcs", "carmel_hip_composition_seconds",
    "carmel_hip_composition_export", "carmel_hip_composition_free",
    "carmel_hip_debug_lattice_fingerprint", "carmel_hip_lattice_layout", "carmel_hip_lattice_tile_sweep", "carmel_hip_lattice_fused_lanes", "carmel_hip_lattice_weight_source", "carmel_hip_accumulate_counts",
    "carmel_hip_comm_unique_id", "carmel_hip_comm_create", "carmel_hip_comm_destroy", "carmel_hip_comm_rank",
    "carmel_hip_comm_world", "carmel_hip_allreduce_counts", "carmel_hip_comm_allreduce_host",
    "carmel_hip_comm_abort", "carmel_hip_comm_transport_name", "carmel_hip_comm_create_custom", "carmel_hip_comm_set_sendrecv", "carmel_hip_comm_selftest", "carmel_hip_exchange_plan",
    "carmel_hip_exchange_info", "carmel_hip_exchange_measure", "carmel_hip_exchange_clear", "carmel_hip_set_layout_policy", "carmel_hip_set_matrix_fb",
    "carmel_hip_set_option", "carmel_hip_get_option", "carmel_hip_option_count", "carmel_hip_option_name",
]


class LatticeStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_pairs", "n_pairs_kept", "explored_states", "explored_arcs", "kept_states", "kept_arcs",
        "n_cyclic_pairs", "n_bundles", "max_levels", "device_bytes")] + [("build_seconds", C.c_double)] + [
            (n, C.c_uint64) for n in ("last_pair_explored_states", "last_pair_kept_states", "last_pair_kept_arcs",
                                      "n_windowed_pairs")]


class EstimateResult(C.Structure):
    _fields_ = [("sum_logprob", C.c_double), ("sum_weighted_logprob", C.c_double), ("n_pairs", C.c_uint64),
                ("kernel_ms", C.c_double)]


GIBBS_OBSERVER_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double)  # carmel_hip_gibbs_observer_fn


class GibbsOpts(C.Structure):
    _fields_ = [("iter", C.c_uint32), ("burnin", C.c_uint32), ("seed", C.c_uint64), ("mode", C.c_int),
                ("uniform_p0", C.c_int), ("dirichlet_p0", C.c_int), ("final_counts", C.c_int),
                ("exclude_prior", C.c_int), ("min_prior", C.c_double), ("high_temp", C.c_double),
                ("low_temp", C.c_double), ("expectation", C.c_int), ("restarts", C.c_uint32), ("argmax_final", C.c_int),
                ("argmax_sum", C.c_int), ("include_self", C.c_int), ("random_start", C.c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "carmel_amd: %s not found — build it with `make -C carmel_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback for the EM hot path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise ImportError("carmel_amd: %s lacks symbol %s" % (LIB_PATH, s))
    lib.carmel_hip_last_error.restype = C.c_char_p
    lib.carmel_hip_counts_dev.restype = C.c_void_p
    lib.carmel_hip_stream.restype = C.c_void_p
    lib.carmel_hip_counts_len.restype = C.c_uint64
    vp = C.c_void_p
    lib.carmel_hip_last_error.argtypes = []
    lib.carmel_hip_device_count.argtypes = []
    lib.carmel_hip_set_option.argtypes = [C.c_char_p, C.c_char_p]
    lib.carmel_hip_get_option.argtypes = [C.c_char_p]
    lib.carmel_hip_get_option.restype = C.c_char_p
    lib.carmel_hip_option_count.argtypes = []
    lib.carmel_hip_option_name.argtypes = [C.c_int]
    lib.carmel_hip_option_name.restype = C.c_char_p
    lib.carmel_hip_random_restart.argtypes = [vp, C.c_uint64, C.c_uint32]
    lib.carmel_hip_keep_em_weights.argtypes = [vp]
    lib.carmel_hip_comm_unique_id.argtypes = [vp]
    lib.carmel_hip_debug_lattice_fingerprint.argtypes = [vp, vp]
    lib.carmel_hip_compose.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, vp,
                                       C.c_uint32, vp, C.c_uint32, C.c_uint32]
    lib.carmel_hip_composition_states.argtypes = [vp]
    lib.carmel_hip_composition_states.restype = C.c_uint64
    lib.carmel_hip_composition_arcs.argtypes = [vp]
    lib.carmel_hip_composition_arcs.restype = C.c_uint64
    lib.carmel_hip_composition_seconds.argtypes = [vp]
    lib.carmel_hip_composition_seconds.restype = C.c_double
    lib.carmel_hip_composition_export.argtypes = [vp] * 11
    lib.carmel_hip_composition_free.argtypes = [vp]
    lib.carmel_hip_comm_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp]
    lib.carmel_hip_comm_destroy.argtypes = [vp]
    lib.carmel_hip_comm_abort.argtypes = [vp]
    lib.carmel_hip_comm_transport_name.argtypes = [vp]
    lib.carmel_hip_comm_transport_name.restype = C.c_char_p
    lib.carmel_hip_comm_create_custom.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp]
    lib.carmel_hip_comm_set_sendrecv.argtypes = [vp, vp]
    lib.carmel_hip_comm_selftest.argtypes = [vp, C.c_uint32]
    lib.carmel_hip_exchange_plan.argtypes = [vp, vp, C.c_uint32, C.c_int]
    lib.carmel_hip_exchange_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_uint64)]
    lib.carmel_hip_exchange_measure.argtypes = [vp, C.c_uint32, C.POINTER(C.c_double)]
    lib.carmel_hip_exchange_clear.argtypes = [vp]
    lib.carmel_hip_set_layout_policy.argtypes = [vp, C.c_int]
    lib.carmel_hip_set_matrix_fb.argtypes = [vp, C.c_int]
    lib.carmel_hip_comm_rank.argtypes = [vp]
    lib.carmel_hip_comm_world.argtypes = [vp]
    lib.carmel_hip_allreduce_counts.argtypes = [vp, vp]
    lib.carmel_hip_comm_allreduce_host.argtypes = [vp, vp, C.c_uint32, C.c_int]
    lib.carmel_hip_fractional_counts.argtypes = [vp]
    lib.carmel_hip_set_digamma.argtypes = [vp, C.c_uint32, vp, vp]
    lib.carmel_hip_forests_set_alphas.argtypes = [vp, vp, C.c_uint32]
    lib.carmel_hip_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp]
    lib.carmel_hip_destroy.argtypes = [vp]
    lib.carmel_hip_set_corpus.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp]
    lib.carmel_hip_build_lattices.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(LatticeStats)]
    lib.carmel_hip_set_norm.argtypes = [vp, C.c_int, C.c_double]
    lib.carmel_hip_set_prior.argtypes = [vp, C.c_double, C.c_int]
    lib.carmel_hip_set_cascade.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint64, vp, vp]
    lib.carmel_hip_normalize.argtypes = [vp]
    lib.carmel_hip_set_weights.argtypes = [vp, vp]
    lib.carmel_hip_get_weights.argtypes = [vp, vp]
    lib.carmel_hip_get_arc_weights.argtypes = [vp, vp]
    lib.carmel_hip_estimate.argtypes = [vp, C.POINTER(EstimateResult), vp]
    lib.carmel_hip_estimate_async.argtypes = [vp]
    lib.carmel_hip_estimate_finish.argtypes = [vp, C.POINTER(EstimateResult), vp]
    lib.carmel_hip_counts_dev.argtypes = [vp]
    lib.carmel_hip_counts_len.argtypes = [vp]
    lib.carmel_hip_stream.argtypes = [vp]
    lib.carmel_hip_use_external_counts.argtypes = [vp, vp]
    lib.carmel_hip_synchronize.argtypes = [vp]
    lib.carmel_hip_last_sweep_ms.argtypes = [vp, C.POINTER(C.c_double)]
    lib.carmel_hip_read_scalars.argtypes = [vp, C.POINTER(EstimateResult)]
    lib.carmel_hip_get_counts.argtypes = [vp, vp]
    lib.carmel_hip_set_counts.argtypes = [vp, vp]
    lib.carmel_hip_maximize.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
    lib.carmel_hip_save_counts.argtypes = [vp]
    lib.carmel_hip_save_best.argtypes = [vp]
    lib.carmel_hip_load_best.argtypes = [vp]
    lib.carmel_hip_host_build.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp, vp,
                                          C.c_uint64, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                          C.c_int]
    lib.carmel_hip_host_dims.argtypes = [vp, vp]
    lib.carmel_hip_host_dims.restype = None
    lib.carmel_hip_host_export.argtypes = [vp] + [vp] * 12
    lib.carmel_hip_host_export.restype = None
    lib.carmel_hip_host_export_lanes.argtypes = [vp] * 8
    lib.carmel_hip_host_export_waves.argtypes = [vp] * 10
    lib.carmel_hip_host_transpose.argtypes = [vp] * 12
    lib.carmel_hip_host_transpose.restype = None
    lib.carmel_hip_host_tile_sweep.argtypes = [vp] * 3
    lib.carmel_hip_host_tile_sweep.restype = None
    lib.carmel_hip_host_export_lanes.restype = None
    lib.carmel_hip_host_free.argtypes = [vp]
    lib.carmel_hip_host_free.restype = None
    lib.carmel_hip_gibbs_create.argtypes = [C.POINTER(vp), vp, C.POINTER(GibbsOpts)]
    lib.carmel_hip_gibbs_destroy.argtypes = [vp]
    lib.carmel_hip_gibbs_n_blocks.argtypes = [vp]
    lib.carmel_hip_gibbs_lattice_stats.argtypes = [vp, C.POINTER(LatticeStats)]
    lib.carmel_hip_gibbs_n_blocks.restype = C.c_uint32
    lib.carmel_hip_gibbs_max_sample.argtypes = [vp]
    lib.carmel_hip_gibbs_max_sample.restype = C.c_uint32
    lib.carmel_hip_gibbs_set_init_weights.argtypes = [vp, vp]
    lib.carmel_hip_gibbs_best_run.argtypes = [vp]
    lib.carmel_hip_gibbs_best_run.restype = C.c_uint32
    lib.carmel_hip_gibbs_run.argtypes = [vp, vp, vp]
    lib.carmel_hip_gibbs_run_ex.argtypes = [vp, vp, vp, vp]
    lib.carmel_hip_gibbs_set_prior_inference.argtypes = [vp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32]
    lib.carmel_hip_gibbs_prior_trace.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    lib.carmel_hip_gibbs_n_prior_scales.argtypes = [vp]
    lib.carmel_hip_gibbs_set_run_share.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.carmel_hip_gibbs_best_stats.argtypes = [vp, vp, vp]
    lib.carmel_hip_lattice_layout.argtypes = [vp]
    lib.carmel_hip_lattice_tile_sweep.argtypes = [vp]
    lib.carmel_hip_lattice_fused_lanes.argtypes = [vp]
    lib.carmel_hip_lattice_weight_source.argtypes = [vp]
    lib.carmel_hip_accumulate_counts.argtypes = [vp, C.c_int]
    lib.carmel_hip_forests_set_prior_inference.argtypes = [vp, C.c_double, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
    lib.carmel_hip_forests_prior_trace.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp]
    lib.carmel_hip_gibbs_n_prior_scales.restype = C.c_uint32
    lib.carmel_hip_gibbs_get_sample.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]
    lib.carmel_hip_gibbs_set_observer.argtypes = [vp, C.c_uint32, GIBBS_OBSERVER_FN, vp]
    lib.carmel_hip_gibbs_current_probs.argtypes = [vp, vp]
    lib.carmel_hip_gibbs_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.carmel_hip_gibbs_final_counts.argtypes = [vp, vp]
    lib.carmel_hip_gibbs_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.carmel_hip_gibbs_uniform.restype = C.c_double
    lib.carmel_hip_gibbs_power.argtypes = [C.c_double, C.c_double, C.c_uint32, C.c_uint32]
    lib.carmel_hip_gibbs_power.restype = C.c_double
    lib.carmel_hip_forests_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint64, vp, vp, vp, vp, C.c_uint32, vp,
                                              C.c_uint64, vp, vp]
    lib.carmel_hip_forests_destroy.argtypes = [vp]
    lib.carmel_hip_forests_estimate.argtypes = [vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint64), vp]
    lib.carmel_hip_forests_get_counts.argtypes = [vp, C.c_double, vp]
    lib.carmel_hip_forests_maximize.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
    lib.carmel_hip_forests_get_weights.argtypes = [vp, vp]
    lib.carmel_hip_forests_set_weights.argtypes = [vp, vp]
    lib.carmel_hip_forests_gibbs.argtypes = [vp, C.POINTER(GibbsOpts), C.c_double, vp, vp]
    lib.carmel_hip_forests_best_run.argtypes = [vp]
    lib.carmel_hip_forests_best_run.restype = C.c_uint32
    lib.carmel_hip_forests_final_counts.argtypes = [vp, vp]
    lib.carmel_hip_forests_get_sample.argtypes = [vp, C.c_uint64, vp, C.POINTER(C.c_uint32)]
    lib.carmel_hip_forests_max_sample.argtypes = [vp]
    lib.carmel_hip_forests_max_sample.restype = C.c_uint32
    lib.carmel_hip_forests_viterbi.argtypes = [vp, vp]
    lib.carmel_hip_forests_get_viterbi.argtypes = [vp, C.c_uint64, vp, vp, C.POINTER(C.c_uint32)]
    for s in SYMBOLS:  # a prototype for every entry point: without one ctypes passes Python ints as C int (64-bit seeds
        if getattr(lib, s).argtypes is None:  # and bare handles would be truncated)
            raise ImportError("carmel_amd: no ctypes prototype for %s" % s)
    return lib


lib = _load()


class CarmelHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        RuntimeError.__init__(self, "%s failed (%d): %s" % (where, code, lib.carmel_hip_last_error().decode()))


def check(rc, where):
    if rc != 0:
        raise CarmelHipError(rc, where)


def set_option(key, value):
    """carmel_hip_set_option: one of the library's switches (include/carmel_hip.h); value None unsets it"""
    rc = lib.carmel_hip_set_option(key.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise ValueError("libcarmel_hip has no option %r (it has: %s)" % (key, ", ".join(option_names())))


def get_option(key):
    v = lib.carmel_hip_get_option(key.encode())
    return None if v is None else v.decode()


def option_names():
    return [lib.carmel_hip_option_name(i).decode() for i in range(lib.carmel_hip_option_count())]


def options_from_env(environ=None):
    """what the front ends do at start-up (csrc/host/env_options.hpp): CARMEL_HIP_<KEY>=v -> set_option("<key>", v), CARMEL_TIMING ->
    "timing".  The library itself never reads the environment; bench.py and the tools call this, the package does not."""
    names = set(option_names())
    for k, v in (os.environ if environ is None else environ).items():
        key = "timing" if k == "CARMEL_TIMING" else k[len("CARMEL_HIP_"):].lower() if k.startswith("CARMEL_HIP_") and k != "CARMEL_HIP_LIB" else None
        if key in names:
            set_option(key, v)


def ptr(a):
    """data pointer of a contiguous numpy array (or None)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
