"""Python mirror of the reference's training interface for this path, over the C-ABI.

  HipForwardBackward  ~ forward_backward (carmel/src/train.cc:224-460): estimate() / maximize()
  train()             ~ WFST::train      (carmel/src/train.cc:503-678): iteration control + convergence

It exists so the parity tests read like the reference's own usage and so bench.py / torch.distributed can drive
one trainer per GPU; the arithmetic all happens in libcarmel_hip.so on the GPU.
"""
import ctypes as C
import math

import numpy as np

from . import _capi
from ._capi import EstimateResult, LatticeStats, check, lib, ptr
from .model import NORM_CONDITIONAL


class TrainOpts(object):
    """WFST::train_opts (carmel/src/fst.h:1080-1095) + the -e / -X defaults of carmel.cc:896-897"""

    def __init__(self, max_iter=500, converge_arc_delta=1e-4, converge_ppx_ratio=.999,
                 learning_rate_growth_factor=1.0):
        self.max_iter = max_iter
        self.converge_arc_delta = converge_arc_delta
        self.converge_ppx_ratio = converge_ppx_ratio
        self.learning_rate_growth_factor = learning_rate_growth_factor


class HipComm(object):
    """carmel_hip_comm: the RCCL communicator of corpus-sharded EM (one process per GPU).  Rank 0 makes the id with
    HipComm.unique_id() and hands the 128 bytes to the other ranks by any means."""

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        check(lib.carmel_hip_comm_unique_id(buf), "carmel_hip_comm_unique_id")
        return bytes(buf.raw)

    def __init__(self, device, rank, world, id_bytes):
        assert len(id_bytes) == 128
        h = C.c_void_p()
        check(lib.carmel_hip_comm_create(C.byref(h), device, rank, world, C.c_char_p(id_bytes)), "carmel_hip_comm_create")
        self.h, self.rank, self.world = h, rank, world

    @classmethod
    def custom(cls, plugin_path, session, device, rank, world):
        """a communicator over a transport of the caller's own (carmel_hip_comm_create_custom): `plugin_path` is a shared
        library exporting carmel_hip_transport_open(session, rank, world, device, carmel_hip_transport*) -- e.g. the test
        transport tests/native/libhosttransport.so, which lets several ranks share one GPU"""
        plug = C.CDLL(plugin_path)
        tr = (C.c_void_p * 6)()
        plug.carmel_hip_transport_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        rc = plug.carmel_hip_transport_open(session.encode(), rank, world, device, C.cast(tr, C.c_void_p))
        if rc != 0:
            raise RuntimeError("carmel_hip_transport_open(%s) failed: %d" % (plugin_path, rc))
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(lib.carmel_hip_comm_create_custom(C.byref(h), device, rank, world, C.cast(tr, C.c_void_p)), "carmel_hip_comm_create_custom")
        self.h, self.rank, self.world, self._plug = h, rank, world, plug
        if hasattr(plug, "carmel_hip_transport_sendrecv"):  # point-to-point groups: the exchange's direct form
            check(lib.carmel_hip_comm_set_sendrecv(h, C.cast(plug.carmel_hip_transport_sendrecv, C.c_void_p)), "carmel_hip_comm_set_sendrecv")
        return self

    @property
    def transport(self):
        return lib.carmel_hip_comm_transport_name(self.h).decode()

    def allreduce_host(self, values, op_max=False):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(lib.carmel_hip_comm_allreduce_host(self.h, ptr(v), len(v), 1 if op_max else 0), "carmel_hip_comm_allreduce_host")
        return v

    def selftest(self, n=0):
        """collective: the point-to-point groups of the exchange's direct form on this transport (carmel_hip_comm_selftest)"""
        check(lib.carmel_hip_comm_selftest(self.h, n), "carmel_hip_comm_selftest")

    def abort(self):
        """after a collective failed on some rank: drop what is enqueued instead of waiting for it"""
        if self.h:
            lib.carmel_hip_comm_abort(self.h)
            self.h = None

    def close(self):
        if self.h:
            lib.carmel_hip_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipForwardBackward(object):
    def __init__(self, wfst, corpus, norm_group=NORM_CONDITIONAL, add_count=0.0, smooth_floor=0.0,
                 weight_is_prior_count=False, device=0, prune=True, host_threads=0, cascade=None,
                 normalize_first=True):
        self.wfst, self.corpus = wfst, corpus
        h = C.c_void_p()
        check(lib.carmel_hip_create(C.byref(h), device, wfst.n_states, wfst.final, wfst.n_arcs, ptr(wfst.src),
                                    ptr(wfst.dst), ptr(wfst.isym), ptr(wfst.osym), ptr(wfst.logw), ptr(wfst.group)),
              "carmel_hip_create")
        self.h = h
        self.cascade = cascade
        if cascade is not None:
            c = cascade
            check(lib.carmel_hip_set_cascade(
                h, len(c["param_logw"]), ptr(c["param_logw"]), ptr(c["param_group"]), ptr(c["param_member"]),
                ptr(c["param_src"]), ptr(c["param_in"]), len(c["member_norm"]), ptr(c["member_norm"]),
                ptr(c["member_add_count"]), len(c["chain_off"]) - 1, ptr(c["chain_off"]), ptr(c["chain_param"])),
                "carmel_hip_set_cascade")
            self.n_params = len(c["param_logw"])
        else:
            check(lib.carmel_hip_set_norm(h, norm_group, add_count), "carmel_hip_set_norm")
            self.n_params = wfst.n_arcs
        if normalize_first:  # WFST::train: cascade.normalize(methods) before anything else (train.cc:509)
            check(lib.carmel_hip_normalize(h), "carmel_hip_normalize")
        if cascade is None or smooth_floor:
            check(lib.carmel_hip_set_prior(h, smooth_floor, 1 if weight_is_prior_count else 0), "carmel_hip_set_prior")
        check(lib.carmel_hip_set_corpus(h, corpus.n_pairs, ptr(corpus.in_off), ptr(corpus.in_sym),
                                        ptr(corpus.out_off), ptr(corpus.out_sym), ptr(corpus.weight)),
              "carmel_hip_set_corpus")
        self.has_deriv = np.zeros(corpus.n_pairs, dtype=np.uint8)
        self.lattice_stats = LatticeStats()
        check(lib.carmel_hip_build_lattices(h, 1 if prune else 0, host_threads, ptr(self.has_deriv),
                                            C.byref(self.lattice_stats)), "carmel_hip_build_lattices")
        self.stats = corpus.stats(self.has_deriv.astype(bool))  # cached_derivs.h:87-98 recount
        self.last = None

    def rebuild_lattices(self, prune=True, host_threads=0):
        """carmel_hip_build_lattices again (after carmel_hip_set_layout_policy)"""
        check(lib.carmel_hip_build_lattices(self.h, 1 if prune else 0, host_threads, ptr(self.has_deriv),
                                            C.byref(self.lattice_stats)), "carmel_hip_build_lattices")

    def close(self):
        if self.h:
            lib.carmel_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- E-step --------------------------------------------------------------------------------------
    def estimate(self, per_pair=False):
        """returns (ln unweighted corpus prob, ln weighted corpus prob); counts stay on the GPU"""
        res = EstimateResult()
        pp = np.empty(self.corpus.n_pairs) if per_pair else None
        check(lib.carmel_hip_estimate(self.h, C.byref(res), ptr(pp)), "carmel_hip_estimate")
        self.last = res
        self.pair_logprob = pp
        return res.sum_logprob, res.sum_weighted_logprob

    def estimate_async(self):
        check(lib.carmel_hip_estimate_async(self.h), "carmel_hip_estimate_async")

    def synchronize(self):
        check(lib.carmel_hip_synchronize(self.h), "carmel_hip_synchronize")

    def allreduce_counts(self, comm):
        """enqueue the RCCL all-reduce (sum) of counts[n_arcs + 4] on the trainer's stream (no host sync)"""
        check(lib.carmel_hip_allreduce_counts(self.h, comm.h), "carmel_hip_allreduce_counts")

    def layout_description(self):
        """how the derivation lattices of this shard are held (carmel_hip_lattice_layout + carmel_hip_lattice_stats)"""
        ls, lay = self.lattice_stats, lib.carmel_hip_lattice_layout(self.h)
        if lay == 1:
            return "unrolled over string positions (never stored)"
        if lay == 2:
            return "unrolled, rank-1 dense form (never stored)"
        tiles, fused = lib.carmel_hip_lattice_tile_sweep(self.h), lib.carmel_hip_lattice_fused_lanes(self.h)
        src = self.weight_source
        return "explicit: %d lattices, %d of them windowed; %d lane groups + bundles; %.2f GB in HBM%s%s%s%s" % (
            ls.n_pairs_kept, ls.n_windowed_pairs, ls.n_bundles, ls.device_bytes / 1e9,
            "; laid out for the tile sweep (%d tiles)" % tiles if tiles else "",
            "; fused lanes (%d tiles)" % fused if fused else "",
            "; weights from the WFST's table (%s)" % " and ".join(
                n for b, n in ((1, "tile passes"), (2, "wave sweeps")) if src & b) if src & 3 else "",
            "; wave posteriors straight to the count pass" if src & 4 else "")

    @property
    def weight_source(self):
        """bit 0: the tile passes fetch their weights from the WFST's table (no bucket pass), bit 1: the one-per-wavefront sweeps
        do, bit 2: those sweeps write the count pass's input themselves (carmel_hip_lattice_weight_source)"""
        return lib.carmel_hip_lattice_weight_source(self.h)

    @property
    def tile_sweep_tiles(self):
        """> 0: the E-step is bucket pass, tile_sweep_kernel, bucket pass (carmel_hip_lattice_tile_sweep)"""
        return lib.carmel_hip_lattice_tile_sweep(self.h)

    @property
    def fused_lane_tiles(self):
        """> 0: the lane sweep's backward pass writes XC itself (carmel_hip_lattice_fused_lanes)"""
        return lib.carmel_hip_lattice_fused_lanes(self.h)

    EXCHANGE_FORMS = {"auto": 0, "allreduce": 1, "collectives": 2, "direct": 3}

    def exchange_plan(self, comm, n_chunks=0, force_allreduce=False, form="auto"):
        """plan the per-iteration exchange (collective); returns exchange_info().  form: "auto" (direct point-to-point groups
        where the transport has them, else the collectives), "allreduce", "collectives", "direct" """
        f = 1 if force_allreduce else self.EXCHANGE_FORMS[form]
        check(lib.carmel_hip_exchange_plan(self.h, comm.h, n_chunks, f), "carmel_hip_exchange_plan")
        self._comm = comm  # the plan points at the communicator: keep it from being collected before the trainer
        return self.exchange_info()

    def exchange_info(self):
        sh, k = C.c_int(0), C.c_uint32(0)
        rs, ag, ar = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib.carmel_hip_exchange_info(self.h, C.byref(sh), C.byref(k), C.byref(rs), C.byref(ag), C.byref(ar)),
              "carmel_hip_exchange_info")
        return dict(sharded=bool(sh.value), form=("allreduce", "collectives", "direct")[sh.value], n_chunks=k.value, bytes_reduce_scatter=rs.value, bytes_all_gather=ag.value,
                    bytes_all_reduce=ar.value)

    def exchange_measure(self, reps=5):
        """ms of one iteration's exchange on its own (all its collectives back to back; collective)"""
        ms = C.c_double(0)
        check(lib.carmel_hip_exchange_measure(self.h, reps, C.byref(ms)), "carmel_hip_exchange_measure")
        return ms.value

    def exchange_clear(self):
        check(lib.carmel_hip_exchange_clear(self.h), "carmel_hip_exchange_clear")
        self._comm = None

    def set_matrix_fb(self, on=True):
        """carmel --matrix-fb: the E-step over the dense (input, output, state) matrix instead of derivation lattices"""
        check(lib.carmel_hip_set_matrix_fb(self.h, int(on)), "carmel_hip_set_matrix_fb")

    def set_layout_policy(self, allow_unrolled):
        check(lib.carmel_hip_set_layout_policy(self.h, int(allow_unrolled)), "carmel_hip_set_layout_policy")

    def last_kernel_ms(self):
        ms = C.c_double(0)
        check(lib.carmel_hip_last_sweep_ms(self.h, C.byref(ms)), "carmel_hip_last_sweep_ms")
        return ms.value

    def read_scalars(self):
        res = EstimateResult()
        check(lib.carmel_hip_read_scalars(self.h, C.byref(res)), "carmel_hip_read_scalars")
        return res.sum_logprob, res.sum_weighted_logprob, res.n_pairs

    def use_external_counts(self, dev_ptr):
        check(lib.carmel_hip_use_external_counts(self.h, C.c_void_p(dev_ptr)), "carmel_hip_use_external_counts")

    def counts(self):
        out = np.empty(self.wfst.n_arcs)
        check(lib.carmel_hip_get_counts(self.h, ptr(out)), "carmel_hip_get_counts")
        return out

    # -- M-step --------------------------------------------------------------------------------------
    def maximize(self, delta_scale=1.0):
        mc = C.c_double(0)
        check(lib.carmel_hip_maximize(self.h, delta_scale, C.byref(mc)), "carmel_hip_maximize")
        return mc.value

    def keep_em_weights(self):
        check(lib.carmel_hip_keep_em_weights(self.h), "carmel_hip_keep_em_weights")

    def weights(self):
        out = np.empty(self.n_params)
        check(lib.carmel_hip_get_weights(self.h, ptr(out)), "carmel_hip_get_weights")
        return out

    def arc_weights(self):
        out = np.empty(self.wfst.n_arcs)
        check(lib.carmel_hip_get_arc_weights(self.h, ptr(out)), "carmel_hip_get_arc_weights")
        return out

    def set_weights(self, logw):
        logw = np.ascontiguousarray(logw, dtype=np.float64)
        assert len(logw) == self.n_params
        check(lib.carmel_hip_set_weights(self.h, ptr(logw)), "carmel_hip_set_weights")

    def save_counts(self):
        check(lib.carmel_hip_save_counts(self.h), "carmel_hip_save_counts")

    def save_best(self):
        check(lib.carmel_hip_save_best(self.h), "carmel_hip_save_best")

    def load_best(self):
        check(lib.carmel_hip_load_best(self.h), "carmel_hip_load_best")


def _rel_ppx_ratio_ln(new_ppx_ln, old_ppx_ln):
    # logweight::relative_perplexity_ratio (graehl/shared/weight.h:247-249): (new/old).root(|ln new|)
    if new_ppx_ln == 0:
        return float("nan")
    return (new_ppx_ln - old_ppx_ln) / abs(new_ppx_ln)


def train(fb, opts=None, log=None):
    """WFST::train's loop (carmel/src/train.cc:552-667) for one start (no random restarts).

    Returns (best per-example perplexity ln, trace); trace rows are dicts with the fields of one reference log
    line (train.cc:587-613).  On return the trainer holds the weights that produced the best estimate
    (train.cc:592-600, 673-674)."""
    opts = opts or TrainOpts()
    using_cascade = fb.cascade is not None
    st = fb.stats
    W = st["total_weight"]
    n_sym = max(st["n_input"], st["n_output"])
    best = float("inf")
    have_good = False
    last_change = 10.0
    last_ppx = float("inf")
    learning_rate = 1.0
    growth = 1.0 if using_cascade else opts.learning_rate_growth_factor
    trace = []
    it = 0
    last_was_reset = False
    while True:
        first_time = it == 0
        it += 1
        cascade_counts = using_cascade and not first_time
        if cascade_counts:
            fb.save_counts()
        if opts.max_iter is not None and it > opts.max_iter and have_good:
            break
        lp, wlp = fb.estimate()
        # a corpus probability within two ulps per pair of 1 counts as 1: the reference's log-domain arithmetic lands on ln P = 0
        # exactly there and its convergence test divides zero by zero (host/carmel_main.cpp snap_certain)
        tol = 4.45e-16 * max(st["n_pairs"], 1)
        lp, wlp = (0.0 if abs(lp) <= tol else lp), (0.0 if abs(wlp) <= tol else wlp)
        new_ppx = -wlp / W  # ln of p.ppxper(totalEmpiricalWeight)  (weight.h:311)
        rec = dict(iter=it, log2_prob=lp / math.log(2), log2_ppx_symbol=(-lp / n_sym) / math.log(2) if n_sym else 0.0,
                   log2_ppx_example=(-lp / st["n_pairs"]) / math.log(2), n_symbol=n_sym, n_example=st["n_pairs"],
                   new_best=False, rel_ppx_ratio_ln=float("nan"), last_change=last_change, rate=learning_rate)
        if new_ppx < best and (not using_cascade or cascade_counts):
            rec["new_best"] = True
            best = new_ppx
            have_good = True
            fb.save_best()
        if first_time:
            ratio_ln = -float("inf")
        else:
            ratio_ln = _rel_ppx_ratio_ln(new_ppx, last_ppx)
            rec["rel_ppx_ratio_ln"] = ratio_ln
        trace.append(rec)
        if log:
            log(rec)
        if not last_was_reset:
            if ratio_ln >= math.log(opts.converge_ppx_ratio):
                if learning_rate > 1:
                    # "Failed to improve (relaxation rate too high); starting again at learning rate 1" (train.cc:639-643)
                    learning_rate = 1.0
                    fb.keep_em_weights()
                    last_was_reset = True
                    continue
                if have_good:
                    break
            elif learning_rate < 20:
                learning_rate *= growth
        else:
            last_was_reset = False
        last_change = fb.maximize(learning_rate)
        if last_change <= opts.converge_arc_delta and have_good:
            break
        last_ppx = new_ppx
    fb.load_best()
    return best, trace


class HipGibbs(object):
    """carmel_gibbs (carmel/src/gibbs.cc:15-41) over a HipForwardBackward trainer: `carmel --crp`.

    mode 0 resamples the blocks strictly in order (the reference's chain); mode 1 is the parallel stale-count
    sweep.  After run() the trainer's parameters are the time-averaged probabilities (probs_to_cascade)."""

    def __init__(self, fb, iters, burnin=0, seed=1, mode=0, uniform_p0=False, dirichlet_p0=False, final_counts=False,
                 exclude_prior=False, min_prior=0.01, high_temp=1.0, low_temp=1.0, expectation=False, restarts=0,
                 argmax_final=False, argmax_sum=False, include_self=False, random_start=False):
        from ._capi import GibbsOpts
        self.fb = fb
        self.opts = GibbsOpts(iters, burnin, seed, mode, int(uniform_p0), int(dirichlet_p0), int(final_counts),
                              int(exclude_prior), min_prior, high_temp, low_temp, int(expectation), restarts,
                              int(argmax_final), int(argmax_sum), int(include_self), int(random_start))
        h = C.c_void_p()
        check(lib.carmel_hip_gibbs_create(C.byref(h), fb.h, C.byref(self.opts)), "carmel_hip_gibbs_create")
        self.h = h
        self.n_blocks = lib.carmel_hip_gibbs_n_blocks(h)

    def run(self, after=False):
        """with restarts > 0 the traces of all runs follow each other: run r is [r * (iter + 1), (r + 1) * (iter + 1)).
        after=True (exact mode) also fills iter_after_logprob (carmel_hip_gibbs_run_ex)"""
        n = (self.opts.iter + 1) * (self.opts.restarts + 1)
        self.iter_logprob, self.iter_cheap_logprob = np.zeros(n), np.zeros(n)
        self.iter_after_logprob = np.zeros(n) if after else None
        check(lib.carmel_hip_gibbs_run_ex(self.h, ptr(self.iter_logprob), ptr(self.iter_cheap_logprob),
                                          ptr(self.iter_after_logprob)), "carmel_hip_gibbs_run_ex")
        return self.iter_logprob

    def set_init_weights(self, arc_logw):
        """--init-em / --init-from-p0: ln weight per composed arc the first sweep of the first run samples from"""
        a = None if arc_logw is None else np.ascontiguousarray(arc_logw, dtype=np.float64)
        check(lib.carmel_hip_gibbs_set_init_weights(self.h, ptr(a)), "carmel_hip_gibbs_set_init_weights")

    def set_prior_inference(self, stddev, global_=False, local=False, restart_fresh=False, start=0, end=0, groupby=None,
                            n_states=None):
        """--prior-inference-stddev / -global / -local / -restart-fresh / -start / -end, --prior-groupby (one 0/1/2 per
        member transducer): prior-scale inference after every inferring sweep (gibbs.hpp:525-563).  n_states: the members'
        state counts (a JOINT member has a norm group, and so a scale, per state -- with arcs or not)"""
        ns = None if n_states is None else np.ascontiguousarray(n_states, dtype=np.uint32)
        n = len(ns) if ns is not None else (len(groupby) if groupby is not None else 0)
        gb = None if groupby is None else np.ascontiguousarray(list(groupby) + [1] * (n - len(groupby)), dtype=np.int32)
        check(lib.carmel_hip_gibbs_set_prior_inference(self.h, float(stddev), int(global_), int(local), int(restart_fresh),
                                                       int(start), int(end), ptr(gb), ptr(ns), n),
              "carmel_hip_gibbs_set_prior_inference")

    def prior_trace(self):
        """(per sweep {proposed, accepted, ln p1, ln p2, a2, p_accept}, cumulative scale per scale group)"""
        n = (self.opts.iter + 1) * (self.opts.restarts + 1)
        tr, cum = np.zeros((n, 6)), np.zeros(max(1, lib.carmel_hip_gibbs_n_prior_scales(self.h)))
        check(lib.carmel_hip_gibbs_prior_trace(self.h, ptr(tr), n, ptr(cum), len(cum)), "carmel_hip_gibbs_prior_trace")
        return tr, cum[:lib.carmel_hip_gibbs_n_prior_scales(self.h)]

    @property
    def best_run(self):
        return lib.carmel_hip_gibbs_best_run(self.h)

    def sample(self, block):
        buf = np.zeros(max(1, lib.carmel_hip_gibbs_max_sample(self.h)), np.uint32)
        n = C.c_uint32(0)
        check(lib.carmel_hip_gibbs_get_sample(self.h, block, ptr(buf), C.byref(n)), "carmel_hip_gibbs_get_sample")
        return buf[:n.value].tolist()

    def observe(self, every, fn):
        """fn(run, iter, time) after every sweep whose number divides by `every` (carmel --print-every); inside it
        sample(b) and current_probs() show the chain as it stands"""
        from ._capi import GIBBS_OBSERVER_FN
        self._obs = GIBBS_OBSERVER_FN(lambda ctx, run, it, time: fn(run, it, time)) if fn else GIBBS_OBSERVER_FN()
        check(lib.carmel_hip_gibbs_set_observer(self.h, every if fn else 0, self._obs, None), "carmel_hip_gibbs_set_observer")

    def current_probs(self):
        """gibbs_base::proposal_prob of every parameter from the counts as they stand"""
        out = np.zeros(self.fb.n_params, np.float64)
        check(lib.carmel_hip_gibbs_current_probs(self.h, out.ctypes.data_as(C.c_void_p)), "carmel_hip_gibbs_current_probs")
        return out

    def uniform(self, it, block, step):
        return lib.carmel_hip_gibbs_uniform(self.opts.seed, it, block, step)

    def close(self):
        if self.h:
            lib.carmel_hip_gibbs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
