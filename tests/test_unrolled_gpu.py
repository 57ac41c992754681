"""GPU: the unrolled (position x state) sweep for one-tape transducers (carmel_amd/csrc/unrolled.hpp) against the
explicit-lattice path and the oracle."""
import os

import numpy as np
import pytest

from helpers import hip_env, set_hip_option  # noqa: E402,F401

from carmel_amd import HipForwardBackward, Wfst, synth
from carmel_amd.model import Corpus

pytestmark = pytest.mark.gpu


def one_tape(seed, n_states=12, deg=9, n_sym=5, n_pairs=300, lo=2, hi=30, eps_arcs=True, tape="out"):
    """random acceptor-like transducer (*e*:x arcs, plus a few *e*:*e* arcs that form no cycle) and strings drawn
    from it"""
    rng = np.random.default_rng(seed)
    F = n_states - 1
    src, dst, sym = [], [], []
    for s in range(F):
        for _ in range(deg):
            src.append(s)
            dst.append(int(rng.integers(0, n_states)))
            sym.append(int(rng.integers(2, 2 + n_sym)))
        if eps_arcs and s + 1 < n_states and rng.random() < 0.4:  # forward *e*:*e* arc: acyclic by construction
            src.append(s)
            dst.append(int(rng.integers(s + 1, n_states)))
            sym.append(0)
    src, dst, sym = (np.asarray(v, np.uint32) for v in (src, dst, sym))
    zeros = np.zeros_like(sym)
    logw = np.log(rng.uniform(0.05, 1.0, len(src)))
    w = Wfst(n_states, F, src, dst, zeros if tape == "out" else sym, sym if tape == "out" else zeros, logw)
    # strings: random walks to the final state
    order = np.argsort(src, kind="stable")
    outs = {s: [k for k in order if src[k] == s] for s in range(n_states)}
    seqs = []
    while len(seqs) < n_pairs:
        cur, out = 0, []
        for _ in range(int(rng.integers(lo, hi + 1)) * 3):
            if cur == F or not outs[cur]:
                break
            k = outs[cur][int(rng.integers(0, len(outs[cur])))]
            if sym[k]:
                out.append(int(sym[k]))
            cur = int(dst[k])
        if cur == F and lo <= len(out) <= hi:
            seqs.append(out)
    # a few strings without a derivation and one with a symbol the transducer never writes
    seqs[3] = [2 + n_sym] * 4
    off = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.uint64)
    flat = np.asarray([x for q in seqs for x in q], np.uint32)
    empty_off, empty = np.zeros(len(seqs) + 1, np.uint64), np.zeros(0, np.uint32)
    c = Corpus(empty_off, empty, off, flat) if tape == "out" else Corpus(off, flat, empty_off, empty)
    c.weight[:] = rng.uniform(0.5, 2.0, c.n_pairs)
    return w, c


def both_paths(w, c, **kw):
    set_hip_option("unrolled", "1")  # whenever eligible (by default sparse lattices stay explicit)
    a = HipForwardBackward(w, c, **kw)
    set_hip_option("unrolled", "0")
    try:
        b = HipForwardBackward(w, c, **kw)
    finally:
        set_hip_option("unrolled", None)
    return a, b


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(tape="in")), (3, dict(eps_arcs=False, n_states=30, deg=14, n_sym=9)),
                                     (4, dict(n_states=64, deg=6, hi=60)), (5, dict(n_states=9, deg=5, n_pairs=1001, hi=45)),
                                     (6, dict(n_states=33, deg=20, n_sym=3, n_pairs=77)),
                                     (7, dict(n_states=100, deg=5, hi=40)), (8, dict(n_states=300, deg=4, n_pairs=120)),
                                     (9, dict(n_states=65, deg=8, eps_arcs=False, n_pairs=90))])
def test_unrolled_estep_equals_explicit_and_oracle(oracle, seed, kw, ragged, hipopt):
    """state counts on both sides of the 16 / 32 / 64-lane group sizes (4, 2 or 1 pairs per wavefront), pair counts
    that do not fill the last wavefront; `ragged` forces per-symbol table slabs where slabs of one size would be used"""
    if ragged:
        hipopt.set("unrolled_ragged", "1")
    w, c = one_tape(seed, **kw)
    u, e = both_paths(w, c)
    assert u.lattice_stats.n_bundles == 0 and e.lattice_stats.n_bundles > 0  # the first really ran unrolled
    assert (u.has_deriv == e.has_deriv).all() and not u.has_deriv[3]
    assert u.lattice_stats.kept_arcs == e.lattice_stats.kept_arcs
    assert u.lattice_stats.kept_states == e.lattice_stats.kept_states
    lu, le = u.estimate(per_pair=True), e.estimate(per_pair=True)
    assert lu[0] == pytest.approx(le[0], rel=1e-10) and lu[1] == pytest.approx(le[1], rel=1e-10)
    np.testing.assert_allclose(u.pair_logprob[u.has_deriv > 0], e.pair_logprob[e.has_deriv > 0], rtol=1e-10)
    np.testing.assert_allclose(u.counts(), e.counts(), rtol=1e-8, atol=1e-12)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    ow.normalize(0, 0.0)
    r = oracle.estimate(ow, oc)
    np.testing.assert_allclose(u.counts(), np.exp(r["counts_ln"]), rtol=1e-7, atol=1e-12)
    u.close()
    e.close()


def test_unrolled_training_equals_explicit(oracle):
    from carmel_amd.trainer import TrainOpts, train
    w, c = one_tape(7, n_states=20, deg=10, n_sym=6, n_pairs=200)
    w.logw[:] = 0.0
    u, e = both_paths(w, c)
    bu, tu = train(u, TrainOpts(max_iter=10))
    be, te = train(e, TrainOpts(max_iter=10))
    assert len(tu) == len(te)
    for a, b in zip(tu, te):
        assert a["log2_prob"] == pytest.approx(b["log2_prob"], rel=1e-9)
        assert a["new_best"] == b["new_best"]
    np.testing.assert_allclose(np.exp(u.weights()), np.exp(e.weights()), rtol=1e-6, atol=1e-12)
    u.close()
    e.close()


@pytest.mark.parametrize("mode,layout", [("cipher", 1), ("dense", 2)])
def test_composed_arc_counts_under_the_unrolled_cascade_sweep(mode, layout, hipopt):
    """the reference always has arc_counts::counts per COMPOSED arc (train.h:28-40, derivations.h:432-449); the unrolled /
    dense sweep of a cascade accumulates per parameter, so carmel_hip_get_counts runs one on-demand pass over explicit
    lattices with the composed weights as they stand: the same counts as a trainer that keeps explicit lattices"""
    import multirank_worker as mw
    from carmel_amd._capi import lib
    w, u = mw.build(mode, 0, 1)
    assert lib.carmel_hip_lattice_layout(u.h) == layout
    hipopt.set("unrolled", "0")
    w2, e = mw.build(mode, 0, 1)
    assert lib.carmel_hip_lattice_layout(e.h) == 0
    for it in range(2):
        lu, le = u.estimate(), e.estimate()
        assert lu[0] == pytest.approx(le[0], rel=1e-10)
        cu, ce = u.counts(), e.counts()
        assert cu.shape == ce.shape == (w.n_arcs,) and ce.sum() > 0
        np.testing.assert_allclose(cu, ce, rtol=1e-8, atol=1e-12)
        u.maximize(1.0)
        e.maximize(1.0)
    u.close()
    e.close()
