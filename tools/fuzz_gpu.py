#!/usr/bin/env python3
"""Runs the randomised GPU parity tests on seeds beyond the ones the test suite fixes (a fuzzing campaign, through gpurun):
    python tools/fuzz_gpu.py [first_seed] [n_seeds]
Every failure is printed with its test and seed; exit status 1 if any."""
import os
import sys
import tempfile
import traceback
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
from oracle import binding as oracle  # noqa: E402  (the checker)
import test_cli_gpu as C  # noqa: E402
import test_gibbs_gpu as G  # noqa: E402
import test_gpu_parity as P  # noqa: E402
import test_compose_gpu as K  # noqa: E402
import test_forest_gpu as F  # noqa: E402
import test_matrix_fb_gpu as M  # noqa: E402
import numpy as np  # noqa: E402


class MP(object):
    """a stand-in for the tests' `hipopt` fixture (and for pytest's monkeypatch: setenv): the library's switches are options now
    (carmel_hip_set_option); the variable is exported too, for the child processes that translate their environment"""
    def __init__(self):
        self.saved, self.opts = {}, {}

    @staticmethod
    def _key(k):
        return "timing" if k == "CARMEL_TIMING" else k[len("CARMEL_HIP_"):].lower() if k.startswith("CARMEL_HIP_") else None

    def setenv(self, k, v):
        import carmel_amd
        self.saved.setdefault(k, os.environ.get(k))
        os.environ[k] = v
        if self._key(k) in carmel_amd.option_names():
            self.opts.setdefault(self._key(k), carmel_amd.get_option(self._key(k)))
            carmel_amd.set_option(self._key(k), v)

    def delenv(self, k, raising=False):
        import carmel_amd
        self.saved.setdefault(k, os.environ.get(k))
        os.environ.pop(k, None)
        if self._key(k) in carmel_amd.option_names():
            self.opts.setdefault(self._key(k), carmel_amd.get_option(self._key(k)))
            carmel_amd.set_option(self._key(k), None)

    def set(self, key, value):
        env = "CARMEL_TIMING" if key == "timing" else "CARMEL_HIP_" + key.upper()
        if value is None:
            self.delenv(env)
        else:
            self.setenv(env, str(value))

    def unset(self, key):
        self.set(key, None)

    def undo(self):
        import carmel_amd
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        for k, v in self.opts.items():
            carmel_amd.set_option(k, v)
        self.saved, self.opts = {}, {}


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    cases = [
        ("cli random cascades", lambda d, s: C.test_random_cascades_train_like_the_oracle(d, s)),
        ("cli one-tape cascades", lambda d, s: C.test_random_one_tape_cascades(d, s)),
        ("gibbs exact chain", lambda d, s: G.test_gibbs_exact_chain_on_random_cascades(oracle, s)),
        ("gibbs prior inference", lambda d, s: G.test_gibbs_prior_scale_inference_follows_the_oracle(oracle, s)),
        ("gibbs lane sampler", lambda d, s, mp=None: G.test_parallel_sweep_lane_layout_on_random_taggers(oracle, mp, s)),
        ("estep random shapes", lambda d, s: P.test_estep_random_shapes(oracle, 2 + s % 100000)),
        ("windowed lane groups", lambda d, s, mp=None: P.test_windowed_lane_groups(oracle, mp, s, [8, 16, 32, 64][s % 4])),
        ("compose on the device", lambda d, s: K.test_random_transducers(oracle, d, s, ["32", "2"][s % 2])),
        ("forest em", lambda d, s: F.test_forest_em_matches_oracle(oracle, 20 + (s * 37) % 400, s)),
        ("forest sweep formulations", lambda d, s: F.test_parallel_sweep_formulations_on_wide_and_deep_forests(
            oracle, dict(or_max=2 + s % 11, and_max=1 + (s // 11) % 7, depth=2 + s % 3, spine=(s % 5 == 0) * (20 + s % 30), seed=s,
                         **({"temps": (2.5, 0.5)} if s % 4 == 1 else {})))),
    ]
    def scatter_case(d, s, mp=None):
        """random ambiguous models: the four forms of the transposition (first pass scatters / second pass gathers, per
        direction), per-item and run-length indices, must give the same counts bit for bit"""
        w, c = P.ambiguous(s, n_states=20 + s % 200, deg=4 + s % 9, n_sym=2 + s % 7, n_pairs=200 + (s * 13) % 3000, p_eps=0.05 * (s % 5),
                           lo=2, hi=6 + s % 30)
        ref = None
        for runs in ("0", "1"):
            mp.setenv("CARMEL_HIP_TRANS_RUNS", runs)
            for mode in ("0", "1", "2", "3"):
                mp.setenv("CARMEL_HIP_TRANS_SCATTER", mode)
                fb = P._fb(w, c)
                lp, _ = fb.estimate(per_pair=True)
                got = (lp, fb.pair_logprob.copy(), fb.counts().copy())
                fb.close()
                if ref is None:
                    ref = got
                assert got[0] == ref[0] and np.array_equal(got[1], ref[1]), (runs, mode)
                # (these models have a few hundred arcs with 10^4 .. 10^5 uses each: nearly every arc is a split hub whose pieces
                # meet in one atomic add each, so the last bit moves from run to run -- of the SAME form too)
                assert np.allclose(got[2], ref[2], rtol=1e-13, atol=0), (runs, mode)

    def forest_exact_case(d, s):
        """forest_exact.hip against the oracle's chain on random forests: sizes, depths, priors and burn-in vary with the seed"""
        ftext, ntext = F.synth_forests(20 + s % 200, 10 + s % 60, s)
        of, hf = F.make(oracle, ftext, ntext, s)
        iters, burnin, alpha = 3 + s % 5, s % 3, 0.05 + 0.1 * (s % 7)
        hf.gibbs(iters, burnin=burnin, alpha=alpha, seed=s, mode=0)
        ref = of.gibbs(s, iters, burnin=burnin, alpha=alpha)
        for b in range(hf.n_forests):
            assert hf.sample(b) == ref["samples"][b], b
        np.testing.assert_allclose(hf.iter_logprob, ref["iter_logprob"], rtol=1e-10)
        np.testing.assert_allclose(hf.iter_cheap_logprob, ref["iter_cheap_logprob"], rtol=1e-10)
        np.testing.assert_allclose(np.exp(hf.weights()), np.exp(of.weights()), rtol=1e-9, atol=1e-15)
        hf.close()

    def tile_sweep_case(d, s, mp=None):
        """random corpora of small lattices (paths, diamonds, epsilons, pairs without a derivation, per-pair weights, partial last
        groups): tile_sweep_kernel = the three kernels it replaces, bit for bit; = the five-kernel layout up to the order of
        the counts' sums; = the oracle"""
        rng = np.random.default_rng(s)
        w = P.synth.random_wfst(5 + s % 120, 2 + s % 5, n_sym=2 + (s // 3) % 6, p_eps=0.04 * (s % 6), seed=s)
        c = P.synth.random_walk_corpus(w, 30 + (s * 131) % 6000, min_arcs=1 + s % 3, max_arcs=3 + s % 12, seed=s, out_degree=2 + s % 5)
        c.weight[:] = rng.uniform(0.25, 4.0, c.n_pairs)
        if s % 4 == 0:  # some dead arcs
            w.logw[rng.random(w.n_arcs) < 0.02] = -np.inf
        out = {}
        for mode in ("fused", "kernels", "layout"):
            for k in ("CARMEL_HIP_TILE_SWEEP_KERNEL", "CARMEL_HIP_TILE_SWEEP"):
                mp.delenv(k)
            if mode == "kernels":
                mp.setenv("CARMEL_HIP_TILE_SWEEP_KERNEL", "0")
            if mode == "layout":
                mp.setenv("CARMEL_HIP_TILE_SWEEP", "0")
            fb = P._fb(w, c)
            if mode == "fused" and not fb.tile_sweep_tiles:
                fb.close()
                raise pytest.skip.Exception("a lattice beyond the tile sweep")
            lp, _ = fb.estimate(per_pair=True)
            out[mode] = (lp, fb.pair_logprob.copy(), fb.counts().copy())
            fb.close()
        a, b, l = out["fused"], out["kernels"], out["layout"]
        assert (a[0] == b[0] or (np.isnan(a[0]) and np.isnan(b[0]))) and np.array_equal(a[1], b[1])
        assert np.allclose(a[2], b[2], rtol=1e-13, atol=0)  # (split hub arcs: an atomic add per piece)
        assert np.array_equal(a[1], l[1]) and np.allclose(a[2], l[2], rtol=1e-12, atol=0)
        _, _, r = P.oracle_estep(oracle, w, c, normalize=False) if s % 4 == 0 else P.oracle_estep(oracle, w, c)
        if s % 4 != 0:
            ok = r["has_deriv"]
            np.testing.assert_allclose(a[1][ok], r["pair_logprob"][ok], rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(a[2], np.exp(r["counts_ln"]), rtol=P.RTOL, atol=1e-14)

    def fused_lane_case(d, s, mp=None):
        """random corpora of one-per-lane lattices beyond the tile sweep (longer walks, clusters that make real sums, windows forced
        onto small lattices for some seeds, epsilons, dead arcs, per-pair weights, partial groups): sweep_lane_kernel<.., XC> = sweep ->
        post -> trans_c_tile_small on the same layout, bit for bit; = the 16384-position layout up to the order of the counts' sums;
        = the oracle"""
        rng = np.random.default_rng(s)
        if s % 3 == 0:
            members = 2 + s % 3
            ncl = 5 + s % 60
            w = P.synth.clustered_wfst(members * ncl + 1, members * (2 + s % 3), members=members, seed=s)
            c = P.synth.clustered_walk_corpus(w, 40 + (s * 17) % 3000, members * (2 + s % 3), members=members, min_arcs=2 + s % 5,
                                              max_arcs=6 + s % 40, seed=s)
        else:
            w = P.synth.random_wfst(20 + s % 400, 2 + s % 5, n_sym=3 + (s // 3) % 8, p_eps=0.03 * (s % 5), seed=s)
            c = P.synth.random_walk_corpus(w, 40 + (s * 131) % 4000, min_arcs=2 + s % 9, max_arcs=20 + s % 80, seed=s, out_degree=2 + s % 5)
        c.weight[:] = rng.uniform(0.25, 4.0, c.n_pairs)
        if s % 4 == 0:
            w.logw[rng.random(w.n_arcs) < 0.01] = -np.inf
        mp.setenv("CARMEL_HIP_WAVE_MIN_WIDTH", "1e9")
        if s % 2:
            mp.setenv("CARMEL_HIP_LANE_WINDOW_MIN", str(8 + s % 30))
        out = {}
        for mode in ("fused", "kernels", "layout"):
            for k in ("CARMEL_HIP_LANE_FUSED_KERNEL", "CARMEL_HIP_LANE_FUSED"):
                mp.delenv(k)
            if mode == "kernels":
                mp.setenv("CARMEL_HIP_LANE_FUSED_KERNEL", "0")
            if mode == "layout":
                mp.setenv("CARMEL_HIP_LANE_FUSED", "0")
            fb = P._fb(w, c)
            if mode == "fused" and not fb.fused_lane_tiles:
                fb.close()
                raise pytest.skip.Exception("not a fused-lane corpus (the tile sweep takes it, or a lattice no lane takes)")
            lp, _ = fb.estimate(per_pair=True)
            out[mode] = (lp, fb.pair_logprob.copy(), fb.counts().copy())
            fb.close()
        a, b, l = out["fused"], out["kernels"], out["layout"]
        assert (a[0] == b[0] or (np.isnan(a[0]) and np.isnan(b[0]))) and np.array_equal(a[1], b[1])
        assert np.allclose(a[2], b[2], rtol=1e-13, atol=0)
        assert np.array_equal(a[1], l[1]) and np.allclose(a[2], l[2], rtol=1e-12, atol=0)
        _, _, r = P.oracle_estep(oracle, w, c, normalize=False) if s % 4 == 0 else P.oracle_estep(oracle, w, c)
        if s % 4 != 0:
            ok = r["has_deriv"]
            np.testing.assert_allclose(a[1][ok], r["pair_logprob"][ok], rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(a[2], np.exp(r["counts_ln"]), rtol=P.RTOL, atol=1e-14)

    def chains_case(d, s, mp=None):
        """--crp-restarts as concurrent chains on random cascades: the oracle's chain run by run, the same kept run"""
        a, b, corpus_text, normby, priors = G._random_cascade_case(oracle, s)
        norms = [G.NORM_JOINT if ch == "J" else G.NORM_CONDITIONAL for ch in normby]
        oc, ocorp, fb = G._setup(oracle, [a, b], corpus_text, norms, priors)
        from carmel_amd.trainer import HipGibbs
        iters, burnin, restarts = 3 + s % 4, s % 2, 1 + s % 5
        mp.setenv("CARMEL_HIP_GIBBS_CHAINS", str([64, 2, 3][s % 3]))
        gs = HipGibbs(fb, iters, burnin=burnin, seed=s, mode=0, restarts=restarts)
        got = gs.run()
        ref = oracle.gibbs_run(oc, ocorp, gs.uniform, normby=normby, priors=priors, iters=iters, burnin=burnin, restarts=restarts)
        np.testing.assert_allclose(got, ref["iter_logprob"], rtol=1e-10)
        assert G.same_kept_run(gs.best_run, ref, iters, burnin)
        if gs.best_run == ref["best_run"]:  # (a tie in the last bit keeps another run: its sample and weights are that run's)
            for blk in range(gs.n_blocks):
                assert gs.sample(blk) == ref["samples"][blk]
            np.testing.assert_allclose(np.exp(fb.weights()), np.exp(ref["param_logw"]), rtol=1e-8, atol=1e-14)
        gs.close()
        fb.close()

    def forest_chains_case(d, s, mp=None):
        """forest-em --crp-restarts as concurrent chains on random forests: every run the oracle's, the same kept run"""
        from carmel_amd._capi import lib
        ftext, ntext = F.synth_forests(20 + s % 150, 10 + s % 50, s)
        of, hf = F.make(oracle, ftext, ntext, s)
        w0 = of.weights().copy()
        iters, burnin, alpha, R = 3 + s % 4, s % 3, 0.05 + 0.1 * (s % 7), 1 + s % 5
        mp.setenv("CARMEL_HIP_GIBBS_CHAINS", str([64, 2, 3][s % 3]))
        lp = hf.gibbs(iters, burnin=burnin, alpha=alpha, seed=s, mode=0, restarts=R)
        stats, runs = [], []
        for r in range(R + 1):
            of.set_weights(w0)
            ref = of.gibbs(lambda i, b, k, r=r: lib.carmel_hip_gibbs_uniform(s, r * (iters + 1) + i, b, k), iters, burnin=burnin, alpha=alpha)
            np.testing.assert_allclose(lp[r], ref["iter_logprob"], rtol=1e-10)
            stats.append(ref["iter_logprob"][min(burnin, iters):].sum())
            runs.append((ref["samples"], of.weights().copy()))
        best = int(np.argmax(stats))  # (the earliest of equal maxima, as the sequential loop keeps it)
        if hf.best_run != best:  # a tie in the last bits may keep another run
            assert abs(stats[hf.best_run] - stats[best]) <= 1e-9 * abs(stats[best])
        for b in range(hf.n_forests):
            assert hf.sample(b) == runs[hf.best_run][0][b]
        np.testing.assert_allclose(np.exp(hf.weights()), np.exp(runs[hf.best_run][1]), rtol=1e-9, atol=1e-15)
        hf.close()

    def host_layout_case(d, s, mp=None):
        """corpora the device builder leaves to the host (wide lattices, cycles, mixtures): the device-built slot order and
        transposition tables against the host's own counting sorts -- the same image, the same counts"""
        import test_lattice_gpu as L
        if s % 3 == 0:
            w = P.synth.random_wfst(10 + s % 60, 3 + s % 4, n_sym=2 + s % 4, p_eps=0.2 + 0.02 * (s % 5), seed=s)  # *e*:*e* cycles
            c = P.synth.random_walk_corpus(w, 50 + (s * 7) % 400, min_arcs=2, max_arcs=6 + s % 8, seed=s, out_degree=3 + s % 4)
        else:
            w = P.synth.random_wfst(2000 + (s * 97) % 20000, 6 + s % 8, n_sym=2 + s % 4, p_eps=0.05 * (s % 3), seed=s)
            c = P.synth.random_walk_corpus(w, 20 + s % 120, min_arcs=4, max_arcs=10 + s % 10, seed=s, out_degree=6 + s % 8)
        host = L._build(w, c, False, device_tables=False)
        hyb = L._build(w, c, False, device_tables=True)
        L._same(host, hyb, atomics=True)

    def tables_case(d, s):
        """carmel --print-counts-* / --print-norms-* on random cascades: the front end's tables (device state, define_param's ids
        through the hash-table walk) against the Python restatement over the oracle's per-sweep state, character for character"""
        a, b, corpus_text, _, priors = G._random_cascade_case(oracle, s)
        pa, pb, pc = (str(d / n) for n in ("a.fst", "b.fst", "corpus"))
        open(pa, "w").write(a)
        open(pb, "w").write(b)
        open(pc, "w").write(corpus_text)
        joint = s % 2 == 1
        kw = [dict(), dict(width=9, norm_order=True), dict(rich=True, width=6), dict(sparse=0.1 * (1 + s % 4)), dict(width=4 + s % 12)][s % 5]
        extra = (["-j"] if joint else []) + (["--width=%d" % kw["width"]] if "width" in kw else []) + (["--norm-order"] if kw.get("norm_order") else []) + \
            (["--print-counts-rich"] if kw.get("rich") else []) + (["--print-counts-sparse=%.17g" % kw["sparse"]] if "sparse" in kw else [])
        C._check_tables(oracle, d, [pc, pa, pb], corpus_text, extra, kw, "JJ" if joint else "CC", priors, 1 + s, N=3 + s % 4, B=s % 3, E=1 + s % 3)

    def table_weights_case(d, s, mp=None):
        """random ambiguous models, lattices one per lane and one per wavefront: the sweeps' weights through the transposition's
        passes or straight from the WFST's table (CARMEL_HIP_TILE_GATHER, CARMEL_HIP_WAVE_GATHER), the wave sweeps' posteriors
        through `post` or straight to the count pass's input (CARMEL_HIP_WAVE_XC) -- the same ln p bit for bit, the same counts
        (up to the split arcs' atomics), and the oracle's"""
        w, c = P.ambiguous(s, n_states=12 + s % 120, deg=3 + s % 7, n_sym=2 + s % 5, n_pairs=100 + (s * 13) % 1500, p_eps=0.04 * (s % 4),
                           lo=2, hi=6 + s % 25)
        mp.setenv("CARMEL_HIP_TRANS_RUNS", "0")
        waves = s % 3 != 0
        if waves:
            mp.setenv("CARMEL_HIP_WAVE_MIN_WIDTH", "0")
            mp.setenv("CARMEL_HIP_WAVE_RING", str(s % 2))
            mp.setenv("CARMEL_HIP_LANE_STATES", "0" if s % 3 == 1 else str(6 + s % 20))
        ref = None
        combos = [("1", "1", "1"), ("0", "0", "0"), ("1", "0", "1"), ("0", "1", "0")] if waves else [("1", "1", "1"), ("0", "1", "1")]
        for tg, wg, xc in combos:
            mp.setenv("CARMEL_HIP_TILE_GATHER", tg)
            mp.setenv("CARMEL_HIP_WAVE_GATHER", wg)
            mp.setenv("CARMEL_HIP_WAVE_XC", xc)
            fb = P._fb(w, c)
            src = fb.weight_source
            lp, _ = fb.estimate(per_pair=True)
            got = (lp, fb.pair_logprob.copy(), fb.counts().copy(), fb.has_deriv.copy())
            fb.close()
            if ref is None:
                ref = got
                ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
                ow.normalize(0, 0.0)
                r = oracle.estimate(ow, oc)
                ok = r["has_deriv"]
                assert np.array_equal(ok, got[3].astype(bool))
                np.testing.assert_allclose(got[1][ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9)
                np.testing.assert_allclose(got[2], np.exp(r["counts_ln"]), rtol=1e-7, atol=1e-12)
            assert got[0] == ref[0] and np.array_equal(got[1], ref[1]), (tg, wg, xc, src)
            np.testing.assert_allclose(got[2], ref[2], rtol=1e-12, atol=0, err_msg=str((tg, wg, xc, src)))

    cases += [
        ("weights from the table", table_weights_case),
        ("crp tables", tables_case),
        ("forest crp chains", forest_chains_case),
        ("host layout tables", host_layout_case),
        ("fused lanes", fused_lane_case),
        ("crp chains", chains_case),
        ("tile sweep", tile_sweep_case),
        ("forest exact chain", forest_exact_case),
        ("matrix fb", lambda d, s: M.test_matrix_estep_against_the_oracle_and_the_lattices(
            oracle, s, dict(n_states=5 + s % 80, deg=2 + s % 9, n_sym=2 + s % 6, n_pairs=20 + (s * 7) % 300, p_eps=0.05 * (s % 9), lo=1 + s % 3,
                            hi=4 + s % 12))),
        ("transposition forms", scatter_case),
    ]
    only = os.environ.get("FUZZ_ONLY")
    if only:
        cases = [c for c in cases if only in c[0]]
    fails = 0
    for name, fn in cases:
        ok = skipped = 0
        for seed in range(first, first + n):
            if os.environ.get("FUZZ_VERBOSE"):
                print("  %s seed %d" % (name, seed), flush=True)
            with tempfile.TemporaryDirectory() as d:
                mp = MP()
                try:
                    if "mp" in fn.__code__.co_varnames[:fn.__code__.co_argcount]:
                        fn(pathlib.Path(d), seed, mp)
                    else:
                        fn(pathlib.Path(d), seed)
                    ok += 1
                except pytest.skip.Exception:
                    ok += 1
                    skipped += 1
                except BaseException as e:  # noqa: BLE001
                    fails += 1
                    where = traceback.extract_tb(e.__traceback__)[-1]
                    print("FAIL %s seed %d at %s:%d: %s" % (name, seed, os.path.basename(where.filename), where.lineno,
                                                           "".join(traceback.format_exception_only(type(e), e)).strip()[:600]))
                finally:
                    mp.undo()
        print("%-28s %d / %d ok%s" % (name, ok, n, " (%d of them skipped)" % skipped if skipped else ""), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
