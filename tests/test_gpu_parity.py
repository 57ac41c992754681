"""GPU parity tests proper (-m gpu): the HIP path through the C-ABI against the CPU oracle.

Tolerances: posteriors / expected counts / per-pair probabilities within 1e-7 relative (north_star asks 1e-5; the
only differences are summation order and one streaming logsumexp per state instead of pairwise log-adds);
arc indexing exact (the count vector is compared arc id by arc id)."""
import json
import math
import os

import numpy as np
import pytest

from helpers import hip_env, set_hip_option  # noqa: E402,F401


def sig6(x):
    """the reference prints these with 6 significant digits (default ostream precision)"""
    return float("%.6g" % x)

from carmel_amd import synth
from carmel_amd.model import NORM_CONDITIONAL, NORM_JOINT, NORM_NONE, Corpus, Wfst

pytestmark = pytest.mark.gpu
RTOL = 1e-7
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fb(*a, **k):
    from carmel_amd.trainer import HipForwardBackward
    return HipForwardBackward(*a, **k)


def ambiguous(seed, n_states=40, deg=8, n_sym=4, n_pairs=200, p_eps=0.15, lo=3, hi=12):
    w = synth.random_wfst(n_states, deg, n_sym=n_sym, p_eps=p_eps, seed=seed)
    c = synth.random_walk_corpus(w, n_pairs, min_arcs=lo, max_arcs=hi, seed=seed, out_degree=deg)
    return w, c


def oracle_estep(oracle, w, c, group=NORM_CONDITIONAL, normalize=True):
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    if normalize:
        ow.normalize(group, 0.0)
    return ow, oc, oracle.estimate(ow, oc)


@pytest.mark.parametrize("seed,kw", [
    (1, {}), (2, dict(n_sym=3, deg=10, n_states=30)), (3, dict(n_sym=64, deg=20, n_states=500, lo=5, hi=40)),
    (4, dict(n_sym=2, deg=6, n_states=12, hi=20, n_pairs=80)),
])
def test_estep_counts_and_probs(oracle, seed, kw):
    w, c = ambiguous(seed, **kw)
    rng = np.random.default_rng(seed)
    c.weight[:] = rng.uniform(0.5, 3.0, c.n_pairs)  # per-example weights (train.cc:330-331)
    fb = _fb(w, c)
    lp, wlp = fb.estimate(per_pair=True)
    _, _, r = oracle_estep(oracle, w, c)
    ok = r["has_deriv"]
    assert np.array_equal(ok, fb.has_deriv.astype(bool))
    np.testing.assert_allclose(fb.pair_logprob[ok], r["pair_logprob"][ok], rtol=1e-10, atol=1e-10)
    assert np.all(np.isneginf(fb.pair_logprob[~ok]))
    np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=RTOL, atol=1e-14)
    assert lp == pytest.approx(r["sum_logprob"], rel=1e-12)
    assert wlp == pytest.approx(r["sum_weighted_logprob"], rel=1e-12)
    fb.close()


@pytest.mark.parametrize("seed", range(10))
def test_estep_random_shapes(oracle, seed):
    """randomised shapes: tiny to mid-sized transducers, few to many epsilons (cyclic lattices included), one pair to
    hundreds, and hub arcs used thousands of times (single-arc and split buckets of the transposition)"""
    rng = np.random.default_rng(100 + seed)
    if seed == 0:  # an arc with more uses than one bucket holds (pieces meet in one atomic add each)
        kw = dict(n_states=2, deg=2, n_sym=2, p_eps=0.0, n_pairs=1500, lo=40, hi=60)
    elif seed == 1:  # hub arcs: a 3-state machine, every pair crosses the same few arcs dozens of times
        kw = dict(n_states=3, deg=2, n_sym=2, p_eps=0.0, n_pairs=int(rng.integers(900, 1500)), lo=20, hi=40)
    else:
        kw = dict(n_states=int(rng.integers(3, 200)), deg=int(rng.integers(2, 12)), n_sym=int(rng.integers(2, 8)),
                  p_eps=float(rng.uniform(0, 0.4)), n_pairs=int(rng.integers(1, 400)), lo=int(rng.integers(1, 4)),
                  hi=int(rng.integers(4, 30)))
    w, c = ambiguous(200 + seed, **kw)
    c.weight[:] = rng.uniform(0.25, 2.0, c.n_pairs)
    fb = _fb(w, c)
    lp, wlp = fb.estimate(per_pair=True)
    _, _, r = oracle_estep(oracle, w, c)
    ok = r["has_deriv"]
    assert np.array_equal(ok, fb.has_deriv.astype(bool)), kw
    np.testing.assert_allclose(fb.pair_logprob[ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9, err_msg=str(kw))
    np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=RTOL, atol=1e-13, err_msg=str(kw))
    assert wlp == pytest.approx(r["sum_weighted_logprob"], rel=1e-11)
    fb.close()


@pytest.mark.parametrize("scale", [1, 12])
def test_mixed_class_corpus(oracle, scale):
    """one corpus of all three lattice classes over one transducer (synth.mixed_wfst, the shape of `bench.py --config mix`:
    single paths, clusters of 3, clusters of 8; short and long pairs interleaved in corpus order): per-pair ln p, the expected
    counts and the next weights against the oracle.  scale 12: enough long wide lattices for the one-per-wavefront layout to
    sit beside the lane groups in one lattice set."""
    regions = ((300, 1, 6, 16, (3, 12), 0.7), (90, 3, 4, 4, (3, 12), 0.2), (24, 8, 4, 4, (8, 30 * (2 if scale > 1 else 1)), 0.1))
    w, layout = synth.mixed_wfst(regions, seed=9)
    c, reg = synth.mixed_walk_corpus(w, layout, 400 * scale, seed=9)
    assert set(np.unique(reg)) == {0, 1, 2}
    fb = _fb(w, c)
    lp, wlp = fb.estimate(per_pair=True)
    ow, oc, r = oracle_estep(oracle, w, c)
    ok = r["has_deriv"]
    assert ok.all() and np.array_equal(ok, fb.has_deriv.astype(bool))
    np.testing.assert_allclose(fb.pair_logprob[ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=RTOL, atol=1e-13)
    assert lp == pytest.approx(r["sum_logprob"], rel=1e-11)
    fb.close()
    # ... and three iterations of training: the oracle's log lines and weights
    from carmel_amd.trainer import TrainOpts, train
    fb = _fb(w, c)
    best, trace = train(fb, TrainOpts(max_iter=3))
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    obest, otrace = oracle.train(ow, oc, norm_group=NORM_CONDITIONAL, max_iter=3)
    assert len(trace) == len(otrace) >= 3
    for a, b in zip(trace, otrace):
        assert a["log2_prob"] == pytest.approx(b["log2_prob"], rel=1e-9)
    np.testing.assert_allclose(np.exp(fb.weights()), np.exp(ow.arrays()["logw"]), rtol=1e-6, atol=1e-12)
    fb.close()


def test_big_single_lattice_classes(oracle):
    # long pairs over a tiny alphabet: lattices of thousands of states -> the 256- and 1024-thread classes
    w = synth.random_wfst(6, 5, n_sym=3, p_eps=0.2, seed=21)
    c = synth.random_walk_corpus(w, 6, min_arcs=60, max_arcs=90, seed=21, out_degree=5)
    fb = _fb(w, c)
    fb.estimate(per_pair=True)
    _, _, r = oracle_estep(oracle, w, c)
    ok = r["has_deriv"]
    assert fb.lattice_stats.kept_states > 2000
    np.testing.assert_allclose(fb.pair_logprob[ok], r["pair_logprob"][ok], rtol=1e-10)
    np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=RTOL, atol=1e-14)
    fb.close()


def test_cyclic_lattice_matches_reference_order(oracle):
    src, dst = [0, 1, 1, 2, 2], [1, 2, 3, 1, 3]
    isym = osym = [2, 0, 3, 0, 3]
    w = Wfst(4, 3, src, dst, isym, osym, np.log([1.0, 0.3, 0.7, 0.4, 0.6]))
    c = Corpus.from_lists([([2, 3], [2, 3]), ([2, 3], [2, 3])], [1.0, 2.5])
    fb = _fb(w, c, norm_group=NORM_NONE, normalize_first=False)
    fb.estimate(per_pair=True)
    assert fb.lattice_stats.n_cyclic_pairs == 2
    _, _, r = oracle_estep(oracle, w, c, normalize=False)
    np.testing.assert_allclose(fb.pair_logprob, r["pair_logprob"], rtol=1e-12)
    np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=1e-10, atol=1e-300)
    fb.close()


def test_empty_ragged_and_no_derivation(oracle):
    w = Wfst(3, 2, [0, 1], [1, 2], [2, 3], [2, 3], np.log([1.0, 1.0]))
    c = Corpus.from_lists([([2, 3], [2, 3]), ([2, 2], [2, 3]), ([], []), ([2, 3], [2, 3])])
    fb = _fb(w, c)
    lp, _ = fb.estimate(per_pair=True)
    assert fb.has_deriv.tolist() == [1, 0, 0, 1]
    assert fb.stats["n_pairs"] == 2
    np.testing.assert_allclose(fb.counts(), [2.0, 2.0])
    assert lp == pytest.approx(0.0, abs=1e-12)
    fb.close()
    # every pair without a derivation: the reference aborts training (train.cc:241-252)
    from carmel_amd import CarmelHipError
    c2 = Corpus.from_lists([([3], [3])])
    fb2 = _fb(w, c2)
    with pytest.raises(CarmelHipError) as e:
        fb2.estimate()
    assert e.value.code == -4
    fb2.close()


@pytest.mark.parametrize("group", [NORM_CONDITIONAL, NORM_JOINT])
@pytest.mark.parametrize("add_count", [0.0, 0.25])
def test_mstep_normalize(oracle, group, add_count):
    w, c = ambiguous(7)
    grp = w.group.copy()
    grp[::7] = 0  # lock every 7th arc: locked arcs keep their weight and reserve mass (fst.cc:196-230)
    w.logw[::7] = np.log(0.05)
    w = Wfst(w.n_states, w.final, w.src, w.dst, w.isym, w.osym, w.logw, grp)
    fb = _fb(w, c, norm_group=group, add_count=add_count)
    ow = oracle.OracleWfst.from_arrays(w)
    ow.normalize(group, add_count)
    np.testing.assert_allclose(np.exp(fb.weights()), np.exp(ow.arrays()["logw"]), rtol=1e-12, atol=1e-300)
    fb.close()


def _with_ties(w, seed=0, frac=0.25, lock=True, n_ties=5):
    """lock some arcs and tie others into a few !N groups spread over many states (fst.cc:107-152, 169-195)"""
    rng = np.random.default_rng(seed)
    grp = w.group.copy()
    if lock:
        grp[::7] = 0
        w.logw[::7] = np.log(0.05)
    free = np.flatnonzero(grp != 0)
    tied = rng.choice(free, size=max(2, int(len(free) * frac)), replace=False)
    grp[tied] = rng.integers(1, n_ties + 1, size=len(tied)).astype(np.uint32)  # tie ids 1..n_ties
    return Wfst(w.n_states, w.final, w.src, w.dst, w.isym, w.osym, w.logw, grp)


@pytest.mark.parametrize("group", [NORM_CONDITIONAL, NORM_JOINT])
@pytest.mark.parametrize("add_count", [0.0, 0.2])
def test_mstep_normalize_tied_groups(oracle, group, add_count):
    w, c = ambiguous(9)
    w = _with_ties(w)
    fb = _fb(w, c, norm_group=group, add_count=add_count)
    ow = oracle.OracleWfst.from_arrays(w)
    ow.normalize(group, add_count)
    np.testing.assert_allclose(np.exp(fb.weights()), np.exp(ow.arrays()["logw"]), rtol=1e-11, atol=1e-300)
    fb.close()


def test_training_with_tied_groups(oracle):
    from carmel_amd.trainer import TrainOpts, train
    w, c = ambiguous(11, n_pairs=120)
    w.logw[:] = 0.0
    w = _with_ties(w, seed=3, frac=0.06, lock=False, n_ties=2)  # mild tying: the corpus keeps a non-zero probability
    fb = _fb(w, c, norm_group=NORM_CONDITIONAL)
    best, trace = train(fb, TrainOpts(max_iter=8))
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    obest, otrace = oracle.train(ow, oc, norm_group=NORM_CONDITIONAL, max_iter=8)
    assert len(trace) == len(otrace) and len(trace) >= 4
    for a, b in zip(trace, otrace):
        assert np.isfinite(a["log2_prob"])
        assert a["log2_prob"] == pytest.approx(b["log2_prob"], rel=1e-9)
        if a["iter"] > 1:
            assert a["last_change"] == pytest.approx(b["last_change"], rel=1e-6, abs=1e-12)
    np.testing.assert_allclose(np.exp(fb.weights()), np.exp(ow.arrays()["logw"]), rtol=1e-6, atol=1e-12)
    fb.close()


@pytest.mark.parametrize("group", [NORM_CONDITIONAL, NORM_JOINT])
def test_full_training_trace(oracle, group):
    from carmel_amd.trainer import TrainOpts, train
    w, c = ambiguous(5, n_pairs=150)
    w.logw[:] = 0.0  # flat start
    fb = _fb(w, c, norm_group=group)
    best, trace = train(fb, TrainOpts(max_iter=12))
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    obest, otrace = oracle.train(ow, oc, norm_group=group, max_iter=12)
    assert len(trace) == len(otrace)
    for a, b in zip(trace, otrace):
        assert a["iter"] == int(b["iter"])
        assert a["log2_prob"] == pytest.approx(b["log2_prob"], rel=1e-9)
        assert a["log2_ppx_example"] == pytest.approx(b["log2_ppx_example"], rel=1e-9)
        assert a["new_best"] == bool(b["new_best"])
        if a["iter"] > 1:
            assert a["rel_ppx_ratio_ln"] == pytest.approx(b["rel_ppx_ratio_ln"], rel=1e-6, abs=1e-12)
            assert a["last_change"] == pytest.approx(b["last_change"], rel=1e-6, abs=1e-12)
    assert best == pytest.approx(obest, rel=1e-9)
    np.testing.assert_allclose(np.exp(fb.weights()), np.exp(ow.arrays()["logw"]), rtol=1e-6, atol=1e-12)
    fb.close()


@pytest.mark.parametrize("growth", [1.3, 2.5])
def test_overrelaxed_em(oracle, growth):
    """carmel -o: w <- old * (em/old)^rate, renormalised (train.cc:157-171), rate grows by the factor while EM improves
    and falls back to 1 with the plain EM weights when it does not (train.cc:629-648)"""
    from carmel_amd.trainer import TrainOpts, train
    w, c = ambiguous(21, n_pairs=150)
    w.logw[:] = 0.0
    fb = _fb(w, c, norm_group=NORM_CONDITIONAL)
    best, trace = train(fb, TrainOpts(max_iter=25, learning_rate_growth_factor=growth))
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    obest, otrace = oracle.train(ow, oc, norm_group=NORM_CONDITIONAL, max_iter=25, rate_growth=growth)
    assert len(trace) == len(otrace)
    assert max(r["rate"] for r in trace) > 1.0
    for a, b in zip(trace, otrace):
        assert a["log2_prob"] == pytest.approx(b["log2_prob"], rel=1e-8)
        assert a["new_best"] == bool(b["new_best"])
    assert best == pytest.approx(obest, rel=1e-8)
    np.testing.assert_allclose(np.exp(fb.weights()), np.exp(ow.arrays()["logw"]), rtol=1e-5, atol=1e-10)
    fb.close()


def test_golden_epron_jpron(oracle, golden_dir):
    """carmel -t epron-jpron.data epron-jpron.fst — the reference's recorded run (commands.trace:7-77)"""
    from carmel_amd.trainer import TrainOpts, train
    gold = json.load(open(os.path.join(golden_dir, "trace_expected.json")))["epron-jpron"]
    ow = oracle.OracleWfst.parse(open(os.path.join(golden_dir, "epron-jpron.fst")).read())
    ow.reduce()
    oc = oracle.OracleCorpus.parse(ow, open(os.path.join(golden_dir, "epron-jpron.data")).read())
    a = ow.arrays()
    w = Wfst(a["n_states"], a["final"], a["src"], a["dst"], a["isym"], a["osym"], a["logw"], a["group"])
    ca = oc.arrays()
    c = Corpus(ca["in_off"], ca["in_sym"], ca["out_off"], ca["out_sym"], ca["weight"])
    fb = _fb(w, c)
    best, trace = train(fb, TrainOpts())
    assert len(trace) == len(gold["iters"]) == 5
    for t, g in zip(trace, gold["iters"]):
        assert sig6(t["log2_prob"]) == g["log2_prob"]
        assert sig6(t["log2_ppx_example"]) == g["log2_ppx_example"]
        assert t["new_best"] == g["new_best"]
    # max{d(weight)} printed at iteration i is the change made by M-step i-1
    for t, g in zip(trace[2:], gold["iters"][2:]):
        assert t["last_change"] == pytest.approx(g["max_dweight"], rel=1e-9)
    # final weights: the 15-digit transducer the reference printed
    ow.set_logw(fb.weights())
    got = ow.write(full=False, onearc=False)
    import re
    num = re.compile(r"(?<![\w\"])(\d+\.\d+(?:e[-+]\d+)?|\d+e[-+]\d+)(?![\w\"])")
    gl, el = got.strip().split("\n"), gold["final_wfst"].strip().split("\n")
    assert len(gl) == len(el)
    for x, y in zip(gl, el):
        assert num.sub("#", x) == num.sub("#", y)
        for u, v in zip(num.findall(x), num.findall(y)):
            assert float(u) == pytest.approx(float(v), rel=1e-9)
    fb.close()


def test_full_size_properties_c2():
    """BASELINE.json configs[1] at full size: properties that do not need the oracle"""
    w, c = synth.make_config("c2")
    fb = _fb(w, c)
    lp, _ = fb.estimate(per_pair=True)
    counts = fb.counts()
    assert fb.has_deriv.all()
    # every derivation enters the (out-arc-free) final state exactly once
    assert counts[w.dst == w.final].sum() == pytest.approx(c.n_pairs, rel=1e-9)
    # sum over pairs of ln p == the scalar the kernel accumulated
    assert fb.pair_logprob.sum() == pytest.approx(lp, rel=1e-9)
    # flow conservation: expected count into a non-final, non-start state == expected count out of it
    inflow = np.bincount(w.dst, weights=counts, minlength=w.n_states)
    outflow = np.bincount(w.src, weights=counts, minlength=w.n_states)
    mid = np.ones(w.n_states, bool)
    mid[[0, w.final]] = False
    np.testing.assert_allclose(inflow[mid], outflow[mid], rtol=1e-9, atol=1e-9)
    # EM monotonicity: likelihood does not decrease across iterations
    prev = lp
    for _ in range(3):
        fb.maximize(1.0)
        cur, _ = fb.estimate()
        assert cur >= prev - 1e-6 * abs(prev)
        prev = cur
    # idempotence of normalisation: after an M-step every (state, input) group sums to 1
    wts = np.exp(fb.weights())
    key = w.src.astype(np.int64) * (1 << 20) + w.isym
    _, inv = np.unique(key, return_inverse=True)
    sums = np.bincount(inv, weights=wts)
    touched = np.bincount(inv, weights=(wts > 0)) > 0
    np.testing.assert_allclose(sums[touched], 1.0, rtol=1e-9)
    fb.close()


def test_full_size_properties_c4():
    """BASELINE.json configs[3] at full size (10^6 states / 10^7 arcs / 10^6 pairs): size-independent properties, and
    the two formulations of the E-step (blocked transposition vs gathers + per-arc slot sums) against each other"""
    w, c = synth.make_config("c4")
    fb = _fb(w, c)
    assert fb.has_deriv.all()
    lp, _ = fb.estimate(per_pair=True)
    counts = fb.counts()
    assert counts[w.dst == w.final].sum() == pytest.approx(c.n_pairs, rel=1e-9)
    assert fb.pair_logprob.sum() == pytest.approx(lp, rel=1e-9)
    inflow = np.bincount(w.dst, weights=counts, minlength=w.n_states)
    outflow = np.bincount(w.src, weights=counts, minlength=w.n_states)
    mid = np.ones(w.n_states, bool)
    mid[[0, w.final]] = False
    np.testing.assert_allclose(inflow[mid], outflow[mid], rtol=1e-9, atol=1e-9)
    # bit-reproducibility: a second E-step on the same weights gives the same counts (fixed summation order; only
    # the <= 1 atomic per piece of a split hub arc may differ in the last bits)
    plp = fb.pair_logprob.copy()
    lp2, _ = fb.estimate()
    c2 = fb.counts()
    assert lp2 == lp
    assert (c2 != counts).sum() <= 16 and np.allclose(c2, counts, rtol=1e-13, atol=0)
    set_hip_option("transpose", "0")
    try:
        fg = _fb(w, c)
    finally:
        set_hip_option("transpose", None)
    lpg, _ = fg.estimate(per_pair=True)
    assert lpg == pytest.approx(lp, rel=1e-12)
    np.testing.assert_allclose(fg.counts(), counts, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(fg.pair_logprob, plp, rtol=1e-12)
    fg.close()
    prev = lp
    for _ in range(2):
        fb.maximize(1.0)
        cur, _ = fb.estimate()
        assert cur >= prev - 1e-6 * abs(prev)
        prev = cur
    wts = np.exp(fb.weights())
    key = w.src.astype(np.int64) * (1 << 20) + w.isym
    _, inv = np.unique(key, return_inverse=True)
    sums = np.bincount(inv, weights=wts)
    touched = np.bincount(inv, weights=(wts > 0)) > 0
    np.testing.assert_allclose(sums[touched], 1.0, rtol=1e-9)
    fb.close()


@pytest.mark.parametrize("name,n_pairs", [("c2", None), ("c4", 300000)])
def test_run_length_transposition_indices_are_the_per_item_ones(name, n_pairs, hipopt, capfd):
    """TransArgs::tr_* / br_*: the run-length form of the transposition's source indices (run starts as a bit mask in LDS)
    must move the very same items -- counts and ln p bit for bit those of the per-item index arrays"""
    w, c = synth.make_config(name, n_pairs=n_pairs)
    out = {}
    hipopt.set("timing", "1")
    for mode in ("0", "1"):
        hipopt.set("trans_runs", mode)
        capfd.readouterr()
        fb = _fb(w, c)
        lp, _ = fb.estimate(per_pair=True)
        out[mode] = (lp, fb.pair_logprob.copy(), fb.counts().copy(), capfd.readouterr().err)
        fb.maximize(1.0)
        lp2, _ = fb.estimate()
        out[mode] += (lp2, fb.counts().copy())
        fb.close()
    a, b = out["0"], out["1"]
    assert "-> run-length indices" in b[3] and "-> run-length indices" not in a[3]
    assert a[0] == b[0] and a[4] == b[4]
    assert np.array_equal(a[1], b[1])
    # (the one atomic add per piece of a split hub arc aside)
    for x, y in ((a[2], b[2]), (a[5], b[5])):
        assert (x != y).sum() <= 16 and np.allclose(x, y, rtol=1e-13, atol=0)


@pytest.mark.parametrize("name,n_pairs", [("c2", None), ("c4", 300000), ("c4a", 40000)])
def test_scattering_first_pass_of_the_transposition_moves_the_same_items(name, n_pairs, hipopt):
    """TransArgs::scatter: either direction of the blocked transposition may do its random access in the FIRST pass (a
    scattered write of one run per tile / bucket) and read sequentially in the second, instead of writing sequentially
    and gathering.  Same items, same per-arc summation order: counts and ln p bit for bit, with per-item and with
    run-length indices"""
    w, c = synth.make_config(name, n_pairs=n_pairs)
    out = {}
    for runs in ("0", "1"):
        hipopt.set("trans_runs", runs)
        for mode in ("0", "1", "2", "3"):
            hipopt.set("trans_scatter", mode)
            fb = _fb(w, c)
            lp, _ = fb.estimate(per_pair=True)
            res = (lp, fb.pair_logprob.copy(), fb.counts().copy())
            fb.maximize(1.0)
            lp2, _ = fb.estimate()
            out[runs, mode] = res + (lp2, fb.counts().copy())
            fb.close()
    a = out["0", "0"]
    for key, b in out.items():
        assert a[0] == b[0] and a[3] == b[3], key
        assert np.array_equal(a[1], b[1]), key
        # (the one atomic add per piece of a split hub arc aside)
        for x, y in ((a[2], b[2]), (a[4], b[4])):
            assert (x != y).sum() <= 16 and np.allclose(x, y, rtol=1e-13, atol=0), key


@pytest.mark.parametrize("kind", ["paths", "ambiguous"])
def test_tile_sweep_is_the_three_kernels_it_replaces(oracle, kind, hipopt):
    """tile_sweep_kernel (tile_sweep.hip): on a corpus of small plain lane lattices the weights' way into lattice order, the lane
    sweeps and the posteriors' way out are one persistent kernel working out of LDS.  Same layout, the three kernels instead
    (CARMEL_HIP_TILE_SWEEP_KERNEL=0): the same bits (ln p per pair, every count, the weights after an M-step; the sign of a
    zero aside).  The five-kernel layout (CARMEL_HIP_TILE_SWEEP=0): the same ln p, counts equal up to the order of their sums.
    And both are the oracle's."""
    if kind == "paths":  # config 4's shape: nearly every lattice a single path (the record-free sweep of a group)
        w, c = synth.make_config("c4", n_pairs=120000)
    else:  # three symbols, no epsilons: lattices of up to a few dozen arcs with real sums
        w = synth.random_wfst(60, 4, n_sym=3, p_eps=0.0, seed=5)
        c = synth.random_walk_corpus(w, 40000, min_arcs=3, max_arcs=9, seed=5, out_degree=4)
    out = {}
    for mode in ("fused", "kernels", "layout"):
        hipopt.unset("tile_sweep_kernel")
        hipopt.unset("tile_sweep")
        if mode == "kernels":
            hipopt.set("tile_sweep_kernel", "0")
        if mode == "layout":
            hipopt.set("tile_sweep", "0")
        fb = _fb(w, c)
        assert (fb.tile_sweep_tiles > 0) == (mode != "layout")
        lp, _ = fb.estimate(per_pair=True)
        res = [lp, fb.pair_logprob.copy(), fb.counts().copy()]
        fb.maximize(1.0)
        lp2, _ = fb.estimate()
        out[mode] = res + [lp2, fb.counts().copy()]
        fb.close()
    a, b, l = out["fused"], out["kernels"], out["layout"]
    assert a[0] == b[0] and a[3] == b[3] and np.array_equal(a[1], b[1])
    for x, y in ((a[2], b[2]), (a[4], b[4])):  # (the one atomic add per piece of a split hub arc aside)
        assert (x != y).sum() <= 16 and np.allclose(x, y, rtol=1e-13, atol=0)
    assert np.array_equal(a[1], l[1])
    np.testing.assert_allclose(a[2], l[2], rtol=1e-12, atol=0)
    np.testing.assert_allclose(a[4], l[4], rtol=1e-11, atol=0)
    if kind == "ambiguous":
        _, _, r = oracle_estep(oracle, w, c)
        ok = r["has_deriv"]
        np.testing.assert_allclose(a[1][ok], r["pair_logprob"][ok], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(a[2], np.exp(r["counts_ln"]), rtol=RTOL, atol=1e-14)


@pytest.mark.parametrize("kind", ["clustered", "long_plain", "ragged"])
def test_fused_lane_sweep_is_the_kernels_it_replaces(oracle, kind, hipopt):
    """sweep_lane_kernel<.., XC> (kernels.hip; LatticeSet::lane_fused): on a corpus of one-per-lane lattices the tile sweep does
    not take, the lane sweep's backward pass stages a tile's posteriors in LDS and writes them to XC in tile-major item order
    itself.  Same layout with CARMEL_HIP_LANE_FUSED_KERNEL=0 (sweep -> post -> trans_c_tile): the same bits -- ln p per pair,
    every count, the weights after an M-step.  The 16384-position layout (CARMEL_HIP_LANE_FUSED=0): the same ln p, counts
    equal up to the order of their sums.  And all are the oracle's."""
    if kind == "clustered":  # c4a's shape: windowed groups (three in-arcs per state) next to plain ones
        hipopt.set("lane_window_min", "20")
        w = synth.clustered_wfst(3 * 400 + 1, 12, members=3, seed=5)
        c = synth.clustered_walk_corpus(w, 9000, 12, members=3, min_arcs=3, max_arcs=40, seed=5)
    elif kind == "long_plain":  # single paths and small ambiguities beyond the tile sweep's 48 arcs: plain groups of several tiles
        w = synth.random_wfst(3000, 6, seed=12)
        c = synth.random_walk_corpus(w, 6000, min_arcs=3, max_arcs=90, seed=12, out_degree=6)
    else:  # few ragged lattices: partial groups, tiles with a handful of items, epsilons (kept off the one-per-wavefront layout)
        hipopt.set("wave_min_width", "1e9")
        w = synth.random_wfst(300, 4, n_sym=6, p_eps=0.1, seed=3)
        c = synth.random_walk_corpus(w, 150, min_arcs=20, max_arcs=60, seed=3, out_degree=4)
    out = {}
    for mode in ("fused", "kernels", "layout"):
        for k in ("lane_fused_kernel", "lane_fused"):
            hipopt.unset(k)
        if mode == "kernels":
            hipopt.set("lane_fused_kernel", "0")
        if mode == "layout":
            hipopt.set("lane_fused", "0")
        fb = _fb(w, c)
        assert fb.tile_sweep_tiles == 0 and (fb.fused_lane_tiles > 0) == (mode != "layout")
        lp, _ = fb.estimate(per_pair=True)
        res = [lp, fb.pair_logprob.copy(), fb.counts().copy()]
        fb.maximize(1.0)
        lp2, _ = fb.estimate()
        out[mode] = res + [lp2, fb.counts().copy()]
        fb.close()
    a, b = out["fused"], out["kernels"]
    assert a[0] == b[0] and a[3] == b[3] and np.array_equal(a[1], b[1])
    for x, y in ((a[2], b[2]), (a[4], b[4])):  # (the one atomic add per piece of a split hub arc aside)
        assert (x != y).sum() <= 16 and np.allclose(x, y, rtol=1e-13, atol=0)
    _, _, r = oracle_estep(oracle, w, c)
    ok = r["has_deriv"]
    assert ok.sum() > 0 and np.isfinite(a[1][ok]).all()
    l = out["layout"]
    assert np.array_equal(a[1], l[1])
    np.testing.assert_allclose(a[2], l[2], rtol=1e-12, atol=0)
    np.testing.assert_allclose(a[4], l[4], rtol=1e-11, atol=0)
    np.testing.assert_allclose(a[1][ok], r["pair_logprob"][ok], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(a[2], np.exp(r["counts_ln"]), rtol=RTOL, atol=1e-14)


def test_tile_sweep_with_lattices_without_an_arc(oracle):
    """pairs of two empty strings under a model whose start state is final: lattices of one state and no arc (ln p = 0, no
    counts).  Enough of them fill tiles of the tile-sweep layout that hold no item at all; beside them short paths and
    two-way ambiguities through the same state."""
    rng = np.random.default_rng(7)
    # one state, start = final; symbols: 0 = epsilon, inputs 1..3, outputs 1..2 (input 3 may write either output)
    isym = np.array([1, 2, 3, 3], np.uint32)
    osym = np.array([1, 2, 1, 2], np.uint32)
    w = Wfst(1, 0, np.zeros(4, np.uint32), np.zeros(4, np.uint32), isym, osym, np.log([0.4, 0.3, 0.2, 0.1]))
    pairs = [([], [])] * 2500
    for _ in range(3000):
        n = int(rng.integers(1, 9))
        arcs = rng.integers(0, 4, n)
        pairs.append((isym[arcs].tolist(), osym[arcs].tolist()))
    order = rng.permutation(len(pairs))
    c = Corpus.from_lists([pairs[i] for i in order], weights=rng.uniform(0.5, 2.0, len(pairs)))
    fb = _fb(w, c)
    assert fb.tile_sweep_tiles > 0
    lp, _ = fb.estimate(per_pair=True)
    _, _, r = oracle_estep(oracle, w, c)
    assert r["has_deriv"].all() and np.array_equal(r["has_deriv"], fb.has_deriv.astype(bool))
    empty = np.array([len(pairs[i][0]) == 0 for i in order])
    assert (fb.pair_logprob[empty] == 0.0).all()
    np.testing.assert_allclose(fb.pair_logprob, r["pair_logprob"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=RTOL, atol=1e-14)
    fb.close()


def _cascade_from_golden(oracle, golden_dir, names, corpus_name):
    texts = [open(os.path.join(golden_dir, n)).read() for n in names]
    oc = oracle.OracleCascade(texts)
    comp = oc.composed()
    a = comp.arrays()
    w = Wfst(a["n_states"], a["final"], a["src"], a["dst"], a["isym"], a["osym"], a["logw"], a["group"])
    ca = oc.corpus(open(os.path.join(golden_dir, corpus_name)).read()).arrays()
    c = Corpus(ca["in_off"], ca["in_sym"], ca["out_off"], ca["out_sym"], ca["weight"])
    return oc, w, c


def test_golden_cipher_cascade(oracle, golden_dir):
    """carmel --train-cascade -HJ cipher.data cipher.wfsa cipher.fst (commands.trace:6903-6952): composed-arc
    weights are gathered from the chains, counts scattered back, each member normalised by its own method"""
    import re
    from carmel_amd.trainer import TrainOpts, train
    gold = json.load(open(os.path.join(golden_dir, "trace_expected.json")))["cipher"]
    oc, w, c = _cascade_from_golden(oracle, golden_dir, ["cipher.wfsa", "cipher.fst"], "cipher.data")
    assert (w.n_states, w.n_arcs) == (gold["composed"]["states"], gold["composed"]["arcs"])
    fb = _fb(w, c, cascade=oc.as_dict([NORM_CONDITIONAL, NORM_CONDITIONAL]))
    best, trace = train(fb, TrainOpts())
    assert len(trace) == len(gold["iters"]) == 22
    for t, g in zip(trace, gold["iters"]):
        assert sig6(t["log2_prob"]) == g["log2_prob"]
        assert sig6(t["log2_ppx_example"]) == g["log2_ppx_example"]
        assert t["new_best"] == g["new_best"]
    for t, g in zip(trace[1:], gold["iters"][1:]):
        assert t["rel_ppx_ratio_ln"] == pytest.approx(math.log(float(g["rel_ppx_ratio"])), rel=1e-7)

    def weights(txt):
        d = {}
        for s, t, i, o, x in re.findall(r'\((\S+) \((\S+) (\S+) (\S+) ([^()! ]+)!?\)\)', txt):
            d[(s, t, i, o)] = math.exp(float(x[2:])) if x.startswith("e^") else float(x)
        return d
    pw = fb.weights()
    for m, name in enumerate(["cipher.wfsa.trained", "cipher.fst.trained"]):
        got = weights(oc.write_member(m, pw))
        exp = weights(open(os.path.join(golden_dir, name)).read())
        assert set(got) == set(exp) and len(exp) > 500
        for k in exp:
            assert got[k] == pytest.approx(exp[k], rel=1e-7, abs=1e-300)
    fb.close()


def test_golden_tagging_cascade(oracle, golden_dir):
    """carmel --train-cascade -HJ tagging.data tagging.fsa tagging.fst (commands.trace:5866-5890): 46 states /
    400994 composed arcs, 1005 pairs, 9 iterations"""
    from carmel_amd.trainer import TrainOpts, train
    gold = json.load(open(os.path.join(golden_dir, "trace_expected.json")))["tagging"]
    oc, w, c = _cascade_from_golden(oracle, golden_dir, ["tagging.fsa", "tagging.fst"], "tagging.data")
    assert (w.n_states, w.n_arcs) == (gold["composed"]["states"], gold["composed"]["arcs"])
    fb = _fb(w, c, cascade=oc.as_dict([NORM_CONDITIONAL, NORM_CONDITIONAL]))
    best, trace = train(fb, TrainOpts())
    assert len(trace) == len(gold["iters"]) == 9
    for t, g in zip(trace, gold["iters"]):
        assert sig6(t["log2_prob"]) == g["log2_prob"]
        assert sig6(t["log2_ppx_example"]) == g["log2_ppx_example"]
        assert t["new_best"] == g["new_best"]
        assert t["n_example"] == g["n_example"]
    fb.close()


@pytest.mark.parametrize("group", [NORM_CONDITIONAL, NORM_JOINT])
@pytest.mark.parametrize("lock,add_count", [(False, 0.0), (True, 0.0), (True, 0.3)])
def test_mstep_normalize_large_unnormalised_model(oracle, group, lock, add_count):
    """WFST::normalize (fst.cc:86-244) of an UNNORMALISED 1.2 M-arc model through the one-pass window kernel: with 12 arcs
    per state every 256-parameter workgroup boundary cuts a norm group, so each workgroup reads weights that its
    neighbours are rewriting in the same launch -- they must come from the snapshot, not from the live array."""
    w = synth.random_wfst(100001, 12, n_sym=4, p_eps=0.1, seed=21)
    rng = np.random.default_rng(5)
    logw = np.log(rng.uniform(0.05, 4.0, w.n_arcs))
    grp = w.group.copy()
    if lock:
        grp[::5] = 0
        logw[::5] = np.log(0.02)
    w = Wfst(w.n_states, w.final, w.src, w.dst, w.isym, w.osym, logw, grp)
    c = synth.random_walk_corpus(w, 20, min_arcs=3, max_arcs=8, seed=21, out_degree=12)
    ow = oracle.OracleWfst.from_arrays(w)
    ow.normalize(group, add_count)
    want = np.exp(ow.arrays()["logw"])
    got = []
    for _ in range(2):
        fb = _fb(w, c, norm_group=group, add_count=add_count)  # normalises once at construction (train.cc:509)
        got.append(fb.weights())
        fb.close()
    np.testing.assert_allclose(np.exp(got[0]), want, rtol=1e-12, atol=1e-300)
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("seed,window", [(0, 64), (1, 64), (2, 32), (3, 16), (4, 64), (5, 8)])
def test_windowed_lane_groups(oracle, hipopt, seed, window):
    """lattices whose arcs span few states of the topological numbering are swept one per lane through a RING of LDS rows
    (lattice.hpp, LaneGroup::window): the forward values are parked in a global column and gathered back per record.  The
    window is forced onto small lattices here (it normally starts above 40 states); results are the oracle's."""
    hipopt.set("lane_window_min", "4")
    hipopt.set("lane_window", str(window))
    rng = np.random.default_rng(300 + seed)
    if seed % 2 == 0:  # narrow, long lattices: few states per level
        kw = dict(n_states=int(rng.integers(4, 40)), deg=int(rng.integers(2, 5)), n_sym=int(rng.integers(3, 9)), p_eps=0.05,
                  n_pairs=int(rng.integers(100, 500)), lo=10, hi=int(rng.integers(20, 70)))
    else:
        kw = dict(n_states=int(rng.integers(3, 100)), deg=int(rng.integers(2, 10)), n_sym=int(rng.integers(2, 6)),
                  p_eps=float(rng.uniform(0, 0.3)), n_pairs=int(rng.integers(50, 400)), lo=int(rng.integers(2, 6)),
                  hi=int(rng.integers(8, 40)))
    w, c = ambiguous(400 + seed, **kw)
    c.weight[:] = rng.uniform(0.25, 2.0, c.n_pairs)
    fb = _fb(w, c)
    assert fb.lattice_stats.n_windowed_pairs > 0, kw
    lp, wlp = fb.estimate(per_pair=True)
    _, _, r = oracle_estep(oracle, w, c)
    ok = r["has_deriv"]
    assert np.array_equal(ok, fb.has_deriv.astype(bool)), kw
    np.testing.assert_allclose(fb.pair_logprob[ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9, err_msg=str(kw))
    np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=RTOL, atol=1e-13, err_msg=str(kw))
    assert wlp == pytest.approx(r["sum_weighted_logprob"], rel=1e-11)
    # the same corpus without windows: the same numbers
    hipopt.set("lane_window", "0")
    fb2 = _fb(w, c)
    assert fb2.lattice_stats.n_windowed_pairs == 0
    fb2.estimate(per_pair=True)
    np.testing.assert_allclose(fb.counts(), fb2.counts(), rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(fb.pair_logprob[ok], fb2.pair_logprob[ok], rtol=1e-13)
    fb.close()
    fb2.close()


def test_windowed_tagging_cascade(oracle, golden_dir, hipopt):
    """the tutorial's tagging cascade (sentence lattices of positions x candidate tags: up to hundreds of states, arcs
    between neighbouring positions only) with every lattice above 8 states windowed: the recorded trace still holds"""
    from carmel_amd.trainer import TrainOpts, train
    hipopt.set("lane_window_min", "8")
    gold = json.load(open(os.path.join(golden_dir, "trace_expected.json")))["tagging"]
    oc, w, c = _cascade_from_golden(oracle, golden_dir, ["tagging.fsa", "tagging.fst"], "tagging.data")
    fb = _fb(w, c, cascade=oc.as_dict([NORM_CONDITIONAL, NORM_CONDITIONAL]))
    assert fb.lattice_stats.n_windowed_pairs > 500
    best, trace = train(fb, TrainOpts())
    assert len(trace) == len(gold["iters"]) == 9
    for t, g in zip(trace, gold["iters"]):
        assert sig6(t["log2_prob"]) == g["log2_prob"]
        assert sig6(t["log2_ppx_example"]) == g["log2_ppx_example"]
    fb.close()


def test_linear_count_floor_against_the_log_counts(oracle):
    """Expected counts cross the boundary as LINEAR f64 (include/carmel_hip.h; the one buffer summed across ranks); the
    reference keeps them as logs and adds with log-add (derivations.h:439-447, weight.h:765-801).  The two agree wherever a
    posterior is representable: this test walks a pair's two derivations further and further apart -- the second 1e-100,
    1e-250, 1e-300 and e^-800 times as probable as the first -- and pins the floor: down to 1e-300 the GPU's count is the
    oracle's exp(log count) to 1e-7; below the smallest normal double (e^-708.4) it is EXACTLY 0 where the reference holds
    e^-800.  What that changes: a normalisation group whose every member's count is below the floor is a zero group here
    (weights 0, fst.cc:217-221) where the reference would still normalise its tiny counts; nothing else -- a count that
    small moves no weight by more than 1e-308.  ("Numerics".)"""
    # states: 0 -a-> 1 -b-> 3 (final) and 0 -a-> 2 -b-> 3: two derivations of the pair (a b, a b); state 2's arcs are the unlikely ones
    for ln_ratio in (-230.0, -575.0, -690.0, -800.0):
        src = np.array([0, 0, 1, 2], np.uint32)
        dst = np.array([1, 2, 3, 3], np.uint32)
        sym = np.array([2, 2, 3, 3], np.uint32)
        logw = np.array([0.0, ln_ratio, 0.0, 0.0])
        w = Wfst(4, 3, src, dst, sym, sym, logw)
        c = Corpus.from_lists([([2, 3], [2, 3])] * 3)
        from carmel_amd.trainer import HipForwardBackward
        fb = HipForwardBackward(w, c, norm_group=NORM_NONE, normalize_first=False)
        fb.estimate(per_pair=True)
        got = fb.counts()
        ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
        r = oracle.estimate(ow, oc)
        want_ln = r["counts_ln"]
        np.testing.assert_allclose(fb.pair_logprob, r["pair_logprob"], rtol=1e-12, atol=1e-12)
        assert want_ln[1] == pytest.approx(math.log(3.0) + ln_ratio, rel=1e-12)  # the reference keeps the log count
        if ln_ratio > -700:
            np.testing.assert_allclose(got, np.exp(want_ln), rtol=1e-7, atol=0)
            assert got[1] > 0
        else:
            assert got[1] == 0.0 and got[3] == 0.0  # below the floor: exactly zero, never a denormal or a NaN
            np.testing.assert_allclose(got[[0, 2]], np.exp(want_ln[[0, 2]]), rtol=1e-12)
        fb.close()


@pytest.mark.parametrize("shape", ["tile-sweep", "lanes", "cascade"])
def test_the_iteration_without_a_synchronisation_in_the_middle(oracle, shape):
    """estimate_async -> maximize with nothing in between (the corpus scalars run on the side stream behind the count pass and are
    joined lazily; the M-step's largest change comes back through the pinned mailbox): three iterations that way end in the
    weights, largest changes and corpus probabilities of three iterations that read the scalars after every E-step -- and of
    the same with the mailbox off (CARMEL_HIP_MAILBOX=0 in a child process: the switch is read once)"""
    import subprocess
    import sys
    code = """
import os, sys, json
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_parity as P
from carmel_amd import synth
shape, mode = sys.argv[1], sys.argv[2]
if shape == "tile-sweep":
    w = synth.random_wfst(300, 4, n_sym=6, p_eps=0.0, seed=3)
    c = synth.random_walk_corpus(w, 4000, min_arcs=3, max_arcs=20, seed=3, out_degree=4)
    fb = P._fb(w, c)
elif shape == "lanes":
    w, c = P.ambiguous(7, n_states=60, deg=8, n_sym=3, n_pairs=1500, lo=4, hi=30)
    fb = P._fb(w, c)
else:
    from oracle import binding as ob
    from conftest import GOLDEN
    oc, w, c = P._cascade_from_golden(ob, GOLDEN, ["cipher.wfsa", "cipher.fst"], "cipher.data")
    fb = P._fb(w, c, cascade=oc.as_dict([P.NORM_CONDITIONAL, P.NORM_CONDITIONAL]))
out = []
for it in range(3):
    fb.estimate_async()
    if mode == "read":
        out += list(fb.read_scalars())
    out.append(fb.maximize(1.0))
fb.estimate_async()
out += list(fb.read_scalars())
out += list(fb.weights())
print(json.dumps([float(v) if np.isfinite(v) else str(v) for v in out]))
""" % (ROOT_DIR, os.path.join(ROOT_DIR, "tests"))
    runs = {}
    for mode, env in (("lazy", {}), ("read", {}), ("lazy-nobox", {"CARMEL_HIP_MAILBOX": "0"})):
        p = subprocess.run([sys.executable, "-c", code, shape, mode.split("-")[0]], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           universal_newlines=True, env=dict(os.environ, **env), timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        runs[mode] = json.loads(p.stdout.strip().split("\n")[-1])
    def same(a, b):  # (arcs whose items fill several buckets are summed with atomics: the last bit moves from run to run)
        assert len(a) == len(b)
        for u, v in zip(a, b):
            assert u == v if isinstance(u, str) or isinstance(v, str) else u == pytest.approx(v, rel=1e-11, abs=1e-300)
    n = len(runs["lazy"])
    same(runs["lazy"], runs["lazy-nobox"])
    # the reading run carries three more scalars per iteration: compare what both have
    lazy, read = runs["lazy"], runs["read"]
    same(lazy[:3], [read[3], read[7], read[11]])   # the largest changes
    same(lazy[3:], read[12:])                      # the last scalars and the weights
    assert len(read) == n + 9
