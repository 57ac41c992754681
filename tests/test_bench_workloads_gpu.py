"""GPU: the workloads bench.py actually times -- config 5's synth.random_forests (forest EM, the exact chain, the parallel
stale-count sweep), config 3 at its full 200 000 lines, config 4a / long (clustered transducers) -- against the oracle on
slices the oracle finishes in seconds and through size-independent properties at full size.

The distributional tests pin the samplers on something the oracle does not supply: the EXACT stationary distribution of the
chain on a corpus small enough to enumerate (transition matrix of one sweep over all joint derivation assignments, built
here in numpy from forest-em's definition of a block's proposal, forest-em.hpp:750-766 / forest.hpp:725-758 /
gibbs.hpp:589-592, 769-792), with a z-test per rule at a stated level."""
import math
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from forest_enum import TOY_FORESTS, TOY_NORM, TOY2_FORESTS, TOY2_NORM, derivations as _derivations, match_bfs, group_priors, toy_setup

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "carmel_amd", "bin", "carmel")


# ---------------------------------------------------------------- config 5: synth.random_forests -----------------------------
def _c5(n_forests):
    from carmel_amd import synth
    node_off, label, ref, nxt, n_rules, goff, grule = synth.random_forests(n_forests)
    lw = np.log(np.random.default_rng(4).uniform(0.05, 1.0, n_rules))
    return node_off, label, ref, nxt, n_rules, goff, grule, lw


def _norm_text(goff, grule):
    return "(" + " ".join("(" + " ".join(str(int(r)) for r in grule[int(goff[g]):int(goff[g + 1])]) + ")"
                          for g in range(len(goff) - 1)) + ")"


def _slice(arrs, n):
    node_off, label, ref, nxt = arrs[:4]
    e = int(node_off[n])
    return node_off[:n + 1], label[:e], ref[:e], nxt[:e]


def test_c5_slice_forest_em_and_exact_chain_match_the_oracle(oracle):
    """the first 2 000 forests of the benchmarked generator (same seed, same parameters as bench.py --config c5): four
    EM iterations and the exact chain, draw for draw, as test_forest_em_matches_oracle / test_forest_gibbs_exact_chain do on
    their own tiny generator"""
    from carmel_amd import synth
    from carmel_amd._capi import lib
    from carmel_amd.forests import HipForests
    n = 2000
    node_off, label, ref, nxt, n_rules, goff, grule, lw = _c5(n)
    of = oracle.OracleForests(synth.forests_to_text(node_off, label, ref, nxt, 0, n), _norm_text(goff, grule))
    assert of.n_rules <= n_rules and of.n_nodes == len(label)
    olw = lw[:of.n_rules]
    of.set_weights(olw)
    hf = HipForests(node_off, label, ref, nxt, n_rules, lw, goff, grule)
    for it in range(4):
        avg = hf.estimate(prior_count=0.01, per_forest=True)
        oavg, ocounts_ln, opf = of.estimate(prior_count=0.01)
        assert avg == pytest.approx(oavg, rel=1e-10)
        np.testing.assert_allclose(hf.per_forest_logprob, opf, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(hf.counts(prior_count=0.01)[1:of.n_rules], np.exp(ocounts_ln)[1:], rtol=1e-8, atol=1e-12)
        d, od = hf.maximize(prior_count=0.01), of.maximize()
        gw, ow = hf.weights()[:of.n_rules], of.weights()
        fin = np.isfinite(ow)
        assert np.array_equal(fin, np.isfinite(gw))
        np.testing.assert_allclose(gw[fin], ow[fin], rtol=1e-8, atol=1e-12)
    hf.close()
    # the exact chain on the same slice, from the same starting weights
    of.set_weights(olw)
    hf = HipForests(node_off, label, ref, nxt, n_rules, lw, goff, grule)
    iters = 3
    hf.gibbs(iters, burnin=1, alpha=0.1, seed=4, mode=0)
    r = of.gibbs(lambda i, b, s: lib.carmel_hip_gibbs_uniform(4, i, b, s), iters, burnin=1, alpha=0.1)
    for b in range(n):
        assert hf.sample(b) == r["samples"][b]
    np.testing.assert_allclose(hf.iter_logprob, r["iter_logprob"], rtol=1e-10)
    np.testing.assert_allclose(hf.iter_cheap_logprob, r["iter_cheap_logprob"], rtol=1e-10)
    np.testing.assert_allclose(np.exp(hf.weights()[:of.n_rules]), np.exp(of.weights()), rtol=1e-9, atol=1e-15)
    hf.close()


def test_c5_full_size_parallel_sweep_properties():
    """bench.py --config c5's workload at full size (100 000 forests, 4.5 M nodes, 500 000 parameters), parallel stale-count
    sweeps (the several-lanes-per-forest sampler): (a) every forest's sample is a derivation of that forest -- its rules, in the
    breadth-first order of the walk, are matched against the forest's own AND/OR structure, so the number of sampled rules
    equals the AND nodes visited; (b) the counts behind the
    final weights are exactly the sum over the samples: with --final-counts every rule's weight must equal
    (uses in the samples + prior) / (that sum over its norm group), and every group sums to one; (c) a run is reproducible
    for a seed; (d) the per-sweep probability is finite at every sweep"""
    from carmel_amd.forests import HipForests
    nf = 100000
    node_off, label, ref, nxt, n_rules, goff, grule, lw = _c5(nf)
    res = []
    for rep in range(2):
        hf = HipForests(node_off, label, ref, nxt, n_rules, lw, goff, grule)
        hf.gibbs(12, burnin=3, alpha=0.1, seed=4, mode=1, final_counts=True)
        assert np.all(np.isfinite(hf.iter_cheap_logprob)) and np.all(hf.iter_cheap_logprob < 0)
        samples = [hf.sample(f) for f in range(nf)] if rep == 0 else [hf.sample(f) for f in range(0, nf, 97)]
        res.append((samples, hf.weights().copy(), hf.iter_cheap_logprob.copy()))
        hf.close()
    samples, wts, probs = res[0]
    # (c)
    assert samples[::97] == res[1][0]
    assert np.array_equal(wts, res[1][1]) or np.allclose(wts, res[1][1], rtol=1e-12, atol=0)
    np.testing.assert_allclose(probs, res[1][2], rtol=1e-12)
    # (a): every 20th forest walked in Python (the matcher is pure Python), all of them for length sanity
    for f in range(0, nf, 20):
        b = int(node_off[f])
        e = int(node_off[f + 1])
        s = samples[f]
        assert len(s) > 0
        assert match_bfs(label[b:e], ref[b:e], nxt[b:e], s), "forest %d: the sample is not a derivation of the forest" % f
    assert all(len(s) > 0 for s in samples)  # (a shared sub-forest can be expanded more than once: no upper bound by nodes)
    # (b)
    uses = np.bincount(np.concatenate([np.asarray(s, np.int64) for s in samples]), minlength=n_rules).astype(np.float64)
    gid, gsize, p0, prior = group_priors(n_rules, goff, grule, lw, 0.1)  # alpha * p0 * |group| (gibbs.hpp:589-592)
    in_g = gid >= 0
    tot = np.bincount(gid[in_g], weights=(uses + prior)[in_g], minlength=len(gsize))
    expect = (uses + prior)[in_g] / tot[gid[in_g]]
    np.testing.assert_allclose(np.exp(wts[in_g]), expect, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(np.bincount(gid[in_g], weights=np.exp(wts[in_g]), minlength=len(gsize)), 1.0, rtol=1e-9)


@pytest.mark.parametrize("corpus", [1, 2])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_sampler_marginals_against_the_enumerated_stationary_distribution(oracle, mode, corpus, hipopt):
    """5 forests x 4-5 derivations each = 1 600 joint states: the sweep's transition matrix is built exactly
    (tests/forest_enum.py) and its stationary rule-usage expectations are compared with what the GPU chain time-averages (the
    rule weights after the run are (average use + prior) / (its norm group's), from_gibbs forest-em.hpp:736-742).  16
    independent chains of 6 000 sweeps (burn-in 500) per mode; per rule z = (mean over chains - exact) / (standard error over
    chains); two-sided test at overall level alpha = 1e-3, Bonferroni over the 8 rules: Student t with 15 degrees of freedom
    at 1e-3 / 16 per tail is 4.9 -- |z| < 4.9.  mode 0 = the reference's chain, mode 1 = --crp-parallel (the stale-count
    sweep has its OWN stationary distribution; the last assertion checks that on this corpus the other sweep's distribution
    is outside what the test resolves, i.e. that the test discriminates between the two chains)."""
    from carmel_amd.forests import HipForests
    if mode == 2:  # the parallel sweep's one-forest-per-lane kernel (mode 1 is the several-lanes-per-forest default)
        hipopt.set("forest_multi", "0")
        mode = 1
    # corpus 2: a five-way OR, a nine-child AND (a frontier wider than a forest's eight lanes), a shared sub-forest expanded twice
    of = oracle.OracleForests(*((TOY_FORESTS, TOY_NORM) if corpus == 1 else (TOY2_FORESTS, TOY2_NORM)))  # (the oracle only parses)
    n_rules = of.n_rules
    lw = np.log(np.random.default_rng(3).uniform(0.2, 1.0, n_rules))
    alpha = 0.5
    derivs, exact = toy_setup(of.node_off, of.label, of.ref, of.next, n_rules, of.group_off, of.group_rule, lw, alpha)
    assert [len(d) for d in derivs] == ([5, 4, 4, 5, 4] if corpus == 1 else [9, 2, 6])
    hf = HipForests(of.node_off, of.label, of.ref, of.next, n_rules, lw, of.group_off, of.group_rule)
    chains = []
    n_chains, n_sweeps = (12, 4000) if mode == 0 else (16, 6000)  # (the exact chain is one launch per forest: 60 us each)
    for seed in range(n_chains):
        hf.set_weights(lw)
        hf.gibbs(n_sweeps, burnin=500, alpha=alpha, seed=5 + 7919 * seed, mode=mode)
        chains.append(np.exp(hf.weights()[1:]))
    hf.close()
    chains = np.asarray(chains)
    mean, se = chains.mean(0), chains.std(0, ddof=1) / math.sqrt(len(chains))
    z = (mean - exact[mode == 1][1:]) / se
    assert np.all(np.abs(z) < (4.9 if n_chains == 16 else 5.5)), (z, mean, exact[mode == 1][1:])
    if corpus == 1:  # (on corpus 2 the two sweeps' stationary expectations differ by 3e-7: nothing to tell apart)
        z_other = (mean - exact[mode != 1][1:]) / se
        assert np.max(np.abs(z_other)) > 6.0, z_other


def test_parallel_sweep_approaches_the_exact_chain_as_the_corpus_grows():
    """the stale-count sweep's bias is O(1 / blocks): on 300 and on 1 500 forests of the benchmarked generator (restricted to
    few rules so that every rule is used often) the time-averaged rule probabilities of the parallel chain and of the exact
    chain agree within 3 standard errors + a bias allowance that shrinks with the corpus (0.02 -> 0.006 absolute)"""
    from carmel_amd import synth
    from carmel_amd.forests import HipForests
    for nf, allow in ((300, 0.02), (1500, 0.006)):
        node_off, label, ref, nxt, n_rules, goff, grule = synth.random_forests(nf, n_rules=60, mean_nodes=20, seed=12)
        lw = np.log(np.random.default_rng(5).uniform(0.2, 1.0, n_rules))
        res = {}
        for mode in (0, 1):
            runs = []
            for seed in range(4):
                hf = HipForests(node_off, label, ref, nxt, n_rules, lw, goff, grule)
                hf.gibbs(120 if mode == 0 else 1500, burnin=100, alpha=0.3, seed=7 + seed, mode=mode)
                runs.append(np.exp(hf.weights()[1:]))
                hf.close()
            runs = np.asarray(runs)
            res[mode] = (runs.mean(0), runs.std(0, ddof=1) / 2.0)
        diff = np.abs(res[0][0] - res[1][0])
        tol = 3.0 * np.sqrt(res[0][1] ** 2 + res[1][1] ** 2) + allow
        assert np.all(diff < tol), (nf, float(diff.max()), float(tol.min()))


# ---------------------------------------------------------------- config 3 at full size ---------------------------------------
ITER = re.compile(r"i=(\d+) \(rate=1\): probability=2\^(\S+) ")
NUM = re.compile(r"(?<![\w\"])(\d+\.\d+(?:e[-+]\d+)?|\d+e[-+]\d+)(?![\w\"])")


def _run_cli(args, env):
    p = subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=dict(os.environ, **env))
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stderr


def test_c3_full_size_run_and_slice_parity(tmp_path):
    """bench.py --config c3's workload: 200 000 lines through the front end in the dense (matrix-core) form -- EM is
    monotone (the corpus probability never falls), the trained channel's rows (conditional on the plain symbol) sum to one,
    the language model stays as it was; and on the first 2 000 lines of the SAME corpus the dense sweep, the table walk
    (CARMEL_HIP_DENSE=0) and -- on 300 lines -- explicit lattices (CARMEL_HIP_UNROLLED=0) log the same probabilities
    (rel 1e-9 / 1e-7) and write the same channel"""
    from carmel_amd import synth
    lm, ch, co = synth.cipher_files(200000)
    lines = co.split("\n")  # pairs are (blank line, data line)
    pa, pb = str(tmp_path / "lm.wfsa"), str(tmp_path / "ch.fst")
    open(pa, "w").write(lm)
    open(pb, "w").write(ch)

    def corpus(n):
        p = str(tmp_path / ("corpus%d" % n))
        open(p, "w").write("\n".join(lines[:2 * n]) + "\n")
        return p

    def train(n, iters, env, name):
        d = tmp_path / name
        d.mkdir()
        err = _run_cli(["--train-cascade", "--normby=NC", "-HJ", "-M", str(iters), "-X", "1.1", "-e", "0", corpus(n), pa, pb],
                       dict(CARMEL_TRAINED_DIR=str(d), CARMEL_TIMING="1", **env))
        probs = [float(m.group(2)) for m in ITER.finditer(err)]
        return err, probs, open(str(d / "ch.fst.trained")).read(), open(str(d / "lm.wfsa.trained")).read()

    err, probs, chan, lmt = train(200000, 5, {}, "full")
    assert "layout=unrolled_dense" in err
    assert len(probs) == 5 and all(b >= a - 1e-9 * abs(a) for a, b in zip(probs, probs[1:])), probs
    def weight(tok):  # the writer's spellings: a real number, or e^x when |ln w| >= 82 (wfstio.cc:47-50)
        return math.exp(float(tok[2:])) if tok.startswith("e^") else float(tok)
    rows = {}
    for m in re.finditer(r'\(0 \(0 "(.)" "(.)" (\S+?)\)\)', chan):
        rows.setdefault(m.group(1), []).append(weight(m.group(3)))
    assert len(rows) == 27
    for k, v in rows.items():
        assert sum(v) == pytest.approx(1.0, rel=1e-9), k
    # locked language model: the same weights as given (the writer prints 15 digits; states may be listed in another order)
    a, b = sorted(float(x) for x in NUM.findall(lmt)), sorted(float(x) for x in NUM.findall(lm))
    assert len(a) == len(b)
    np.testing.assert_allclose(a, b, rtol=1e-12)
    # the slice: dense = table walk (= explicit lattices on a shorter slice)
    e1, p1, c1, _ = train(2000, 4, {}, "dense2k")
    e2, p2, c2, _ = train(2000, 4, {"CARMEL_HIP_DENSE": "0"}, "tables2k")
    assert "layout=unrolled_dense" in e1 and "layout=unrolled " in e2
    np.testing.assert_allclose(p1, p2, rtol=1e-9)
    for u, v in zip(NUM.findall(c1), NUM.findall(c2)):
        assert float(u) == pytest.approx(float(v), rel=1e-8, abs=1e-14)
    e3, p3, c3, _ = train(300, 4, {}, "dense300")
    e4, p4, c4, _ = train(300, 4, {"CARMEL_HIP_UNROLLED": "0"}, "explicit300")
    assert "layout=explicit" in e4
    np.testing.assert_allclose(p3, p4, rtol=1e-7)
    for u, v in zip(NUM.findall(c3), NUM.findall(c4)):
        assert float(u) == pytest.approx(float(v), rel=1e-6, abs=1e-12)


# ---------------------------------------------------------------- config 4a / long: clustered transducers ---------------------
@pytest.mark.parametrize("name,pairs", [("c4a", 3000), ("long", 24), ("toya", None)])
def test_clustered_workloads_match_the_oracle(oracle, name, pairs):
    """slices of bench.py's ambiguous workloads (same transducer, the first pairs of the same corpus): per-pair ln p, expected
    counts and three EM iterations against the oracle"""
    from carmel_amd import synth
    from carmel_amd.trainer import HipForwardBackward
    w, c = synth.make_config(name, n_pairs=pairs)
    ls_ratio = None
    fb = HipForwardBackward(w, c)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    ow.normalize(0, 0.0)
    for it in range(3):
        lp, wlp = fb.estimate(per_pair=True)
        r = oracle.estimate(ow, oc)
        ok = r["has_deriv"]
        assert ok.all() and np.array_equal(ok, fb.has_deriv.astype(bool))
        np.testing.assert_allclose(fb.pair_logprob, r["pair_logprob"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=1e-7, atol=1e-12)
        assert lp == pytest.approx(r["sum_logprob"], rel=1e-10, abs=1e-9)
        fb.maximize(1.0)
        ow.set_logw(fb.weights())
    ls = fb.lattice_stats
    ls_ratio = ls.kept_arcs / ls.kept_states
    assert ls_ratio > (2.0 if name != "long" else 6.0)  # the point of these workloads: every state is a real sum
    fb.close()


def test_c4a_full_size_properties():
    """bench.py's c4a at full size (10^6 pairs, ~2*10^8 lattice arcs): flow conservation (the expected counts of the arcs into
    the final state sum to the corpus weight; per state, counts in = counts out), EM monotone over three iterations, the
    weights normalised per (state, input symbol), two E-steps identical to 1e-13"""
    from carmel_amd import synth
    from carmel_amd.trainer import HipForwardBackward
    w, c = synth.make_config("c4a")
    fb = HipForwardBackward(w, c)
    assert fb.has_deriv.all()
    lps = []
    for it in range(3):
        lp, _ = fb.estimate()
        lps.append(lp)
        cnt = fb.counts()
        if it == 0:
            fb.estimate()
            # (bit-identical but for the hub arcs: an arc used more than 16 384 times is summed in pieces that meet in one
            # atomic add each -- the clusters next to the start are used by a third of the corpus)
            np.testing.assert_allclose(cnt, fb.counts(), rtol=1e-13, atol=0)
            assert cnt[w.dst == w.final].sum() == pytest.approx(c.n_pairs, rel=1e-9)
            inflow = np.bincount(w.dst, weights=cnt, minlength=w.n_states)
            outflow = np.bincount(w.src, weights=cnt, minlength=w.n_states)
            mid = np.arange(w.n_states)
            mid = mid[(mid != 0) & (mid != w.final)]
            np.testing.assert_allclose(inflow[mid], outflow[mid], rtol=1e-8, atol=1e-9)
            assert outflow[0] == pytest.approx(c.n_pairs + inflow[0], rel=1e-9)
        fb.maximize(1.0)
    assert lps[1] >= lps[0] and lps[2] >= lps[1]
    lw = fb.weights()
    key = w.src.astype(np.int64) * (1 << 20) + w.isym.astype(np.int64)
    _, inv = np.unique(key, return_inverse=True)
    sums = np.bincount(inv, weights=np.exp(lw))
    used = np.bincount(inv, weights=(lw > -np.inf).astype(np.float64)) > 0
    np.testing.assert_allclose(sums[used], 1.0, rtol=1e-9)
    fb.close()
