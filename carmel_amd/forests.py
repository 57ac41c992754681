"""Python mirror of forest-em's FForests (forest-em/forest-em.hpp) over the C-ABI: EM_executor concept
(estimate / maximize, graehl/shared/em.hpp:74-89) and run_gibbs."""
import ctypes as C

import numpy as np

from ._capi import GibbsOpts, check, lib, ptr


class HipForests(object):
    def __init__(self, node_off, label, ref, nxt, n_rules, rule_logw, group_off, group_rule, device=0):
        self.node_off = np.ascontiguousarray(node_off, dtype=np.uint64)
        self.label = np.ascontiguousarray(label, dtype=np.uint32)
        self.ref = np.ascontiguousarray(ref, dtype=np.int32)
        self.next = np.ascontiguousarray(nxt, dtype=np.uint32)
        self.group_off = np.ascontiguousarray(group_off, dtype=np.uint64)
        self.group_rule = np.ascontiguousarray(group_rule, dtype=np.uint32)
        self.n_rules = int(n_rules)
        self.n_forests = len(self.node_off) - 1
        lw = np.ascontiguousarray(rule_logw, dtype=np.float64)
        h = C.c_void_p()
        check(lib.carmel_hip_forests_create(C.byref(h), device, self.n_forests, ptr(self.node_off), ptr(self.label),
                                            ptr(self.ref), ptr(self.next), self.n_rules, ptr(lw),
                                            len(self.group_off) - 1, ptr(self.group_off), ptr(self.group_rule)),
              "carmel_hip_forests_create")
        self.h = h

    def close(self):
        if self.h:
            lib.carmel_hip_forests_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def estimate(self, prior_count=0.0, per_forest=False):
        avg, nz = C.c_double(0), C.c_uint64(0)
        pf = np.zeros(self.n_forests) if per_forest else None
        check(lib.carmel_hip_forests_estimate(self.h, prior_count, C.byref(avg), C.byref(nz), ptr(pf)),
              "carmel_hip_forests_estimate")
        self.per_forest_logprob, self.n_zero = pf, nz.value
        return avg.value

    def counts(self, prior_count=0.0):
        c = np.zeros(self.n_rules)
        check(lib.carmel_hip_forests_get_counts(self.h, prior_count, ptr(c)), "carmel_hip_forests_get_counts")
        return c

    def maximize(self, prior_count=0.0, add_k=0.0, zero_zerocounts=False):
        d = C.c_double(0)
        check(lib.carmel_hip_forests_maximize(self.h, prior_count, add_k, int(zero_zerocounts), C.byref(d)),
              "carmel_hip_forests_maximize")
        return d.value

    def weights(self):
        w = np.zeros(self.n_rules)
        check(lib.carmel_hip_forests_get_weights(self.h, ptr(w)), "carmel_hip_forests_get_weights")
        return w

    def set_weights(self, lw):
        lw = np.ascontiguousarray(lw, dtype=np.float64)
        check(lib.carmel_hip_forests_set_weights(self.h, ptr(lw)), "carmel_hip_forests_set_weights")

    def gibbs(self, iters, burnin=0, alpha=0.1, seed=1, mode=0, uniform_p0=False, final_counts=False, alphas=None,
              high_temp=1.0, low_temp=1.0, prior_inference=None, exclude_prior=False, restarts=0, argmax_final=False,
              argmax_sum=False):
        """high_temp/low_temp: annealing, choices at probabilities^(1/temperature) (--high-temp/--low-temp).
        prior_inference: dict(stddev, global_, local, start, end) -- --prior-inference-* (gibbs.hpp:525-563), exact mode;
        afterwards self.prior_trace (per sweep {proposed, accepted, ln p1, ln p2, a2, p_accept}) and self.prior_cumulative.
        alphas: forest-em --alpha=FILE, one prior strength per rule id (negative = locked); None: the scalar alpha"""
        al = None if alphas is None else np.ascontiguousarray(alphas, dtype=np.float64)
        check(lib.carmel_hip_forests_set_alphas(self.h, ptr(al), 0 if al is None else len(al)), "carmel_hip_forests_set_alphas")
        # restarts: --crp-restarts, the runs side by side on the device (carmel_hip_forests_gibbs): iter_logprob is then
        # (restarts + 1) x (iters + 1), self.best_run the run that was kept
        o = GibbsOpts(iters, burnin, seed, mode, int(uniform_p0), 0, int(final_counts), int(exclude_prior), 0.01, high_temp, low_temp,
                      0, restarts, int(argmax_final), int(argmax_sum))
        self.iter_logprob, self.iter_cheap_logprob = np.zeros((restarts + 1) * (iters + 1)), np.zeros((restarts + 1) * (iters + 1))
        pi = dict(prior_inference or {})
        check(lib.carmel_hip_forests_set_prior_inference(self.h, float(pi.get("stddev", 0.0)), int(pi.get("global_", False)),
                                                         int(pi.get("local", False)), int(pi.get("start", 0)),
                                                         int(pi.get("end", 0))), "carmel_hip_forests_set_prior_inference")
        check(lib.carmel_hip_forests_gibbs(self.h, C.byref(o), alpha, ptr(self.iter_logprob),
                                           ptr(self.iter_cheap_logprob)), "carmel_hip_forests_gibbs")
        ns = C.c_uint32(0)
        self.prior_trace, cum = np.zeros((iters + 1, 6)), np.zeros(1 << 16)
        check(lib.carmel_hip_forests_prior_trace(self.h, ptr(self.prior_trace), iters + 1, ptr(cum), len(cum), C.byref(ns)),
              "carmel_hip_forests_prior_trace")
        self.prior_cumulative = cum[:ns.value]
        self.best_run = int(lib.carmel_hip_forests_best_run(self.h))
        if restarts:
            self.iter_logprob = self.iter_logprob.reshape(restarts + 1, iters + 1)
            self.iter_cheap_logprob = self.iter_cheap_logprob.reshape(restarts + 1, iters + 1)
        return self.iter_logprob

    def max_sample(self):
        """rules in the largest derivation of any forest"""
        return int(lib.carmel_hip_forests_max_sample(self.h))

    def viterbi(self):
        """forest.hpp:507-632: ln of the best derivation of every forest under the current weights; `best_derivation(f)` then
        gives forest f's derivation in pre-order as (rule, number of children) pairs"""
        best = np.zeros(self.n_forests, np.float64)
        check(lib.carmel_hip_forests_viterbi(self.h, ptr(best)), "carmel_hip_forests_viterbi")
        return best

    def best_derivation(self, forest):
        cap = max(1, lib.carmel_hip_forests_max_sample(self.h))
        rules, arity = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        n = C.c_uint32(0)
        check(lib.carmel_hip_forests_get_viterbi(self.h, forest, ptr(rules), ptr(arity), C.byref(n)),
              "carmel_hip_forests_get_viterbi")
        return list(zip(rules[:n.value].tolist(), arity[:n.value].tolist()))

    def sample(self, forest):
        buf = np.zeros(max(1, lib.carmel_hip_forests_max_sample(self.h)), np.uint32)
        n = C.c_uint32(0)
        check(lib.carmel_hip_forests_get_sample(self.h, forest, ptr(buf), C.byref(n)), "carmel_hip_forests_get_sample")
        return buf[:n.value].tolist()
