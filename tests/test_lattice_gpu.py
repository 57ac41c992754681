"""GPU: derivation-lattice construction on the device (csrc/lattice_gpu.hip) against the host builder: the two must
leave the SAME image in device memory -- lane groups, record streams, posterior slots, transposition tables, checked
through checksums of every array -- and therefore the same has_derivation flags, statistics and expected counts, bit for
bit.  Corpora with a lattice the device builder does not take (a cycle, more than 96 states) go to the host builder."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import hip_env, set_hip_option  # noqa: E402,F401

from carmel_amd import synth
from carmel_amd.model import Corpus, Wfst

pytestmark = pytest.mark.gpu


def _build(w, c, gpu, device_tables=None):
    """gpu: the device builder where it takes the corpus / False: the host builder, all of it (its own counting sorts for the
    posterior slots and the transposition tables too, unless device_tables says otherwise)"""
    from carmel_amd._capi import check, lib, ptr
    from carmel_amd.trainer import HipForwardBackward
    set_hip_option("gpu_build", "1" if gpu else "0")
    if device_tables is None:
        device_tables = gpu
    set_hip_option("device_tables", "1" if device_tables else "0")
    try:
        fb = HipForwardBackward(w, c)
    finally:
        set_hip_option("gpu_build", None)
        set_hip_option("device_tables", None)
    fp = np.zeros(16, np.uint64)
    check(lib.carmel_hip_debug_lattice_fingerprint(fb.h, ptr(fp)), "fingerprint")
    ls = fb.lattice_stats
    stats = tuple(int(getattr(ls, n)) for n in ("n_pairs", "n_pairs_kept", "explored_states", "explored_arcs", "kept_states",
                                                 "kept_arcs", "last_pair_explored_states", "last_pair_kept_states",
                                                 "last_pair_kept_arcs"))
    lp, wlp = fb.estimate(per_pair=True)
    res = dict(fp=fp, stats=stats, has=fb.has_deriv.copy(), counts=fb.counts(), pair_lp=fb.pair_logprob.copy(), lp=lp,
               seconds=ls.build_seconds)
    fb.maximize(1.0)
    res["w1"] = fb.weights()
    fb.close()
    return res


def _same(a, b, atomics=False):
    """atomics: the corpus has arcs whose items fill several buckets -- their pieces are added to the count atomically, in an
    order that changes from run to run, so counts and weights agree to rounding only (the tables still byte for byte)"""
    assert a["stats"] == b["stats"]
    assert np.array_equal(a["has"], b["has"])
    assert np.array_equal(a["fp"], b["fp"]), (a["fp"], b["fp"])
    assert np.array_equal(a["pair_lp"], b["pair_lp"]) and a["lp"] == b["lp"]
    if atomics:
        np.testing.assert_allclose(a["counts"], b["counts"], rtol=1e-12)
        np.testing.assert_allclose(np.exp(a["w1"]), np.exp(b["w1"]), rtol=1e-12)
    else:
        assert np.array_equal(a["counts"], b["counts"])
        assert np.array_equal(a["w1"], b["w1"])


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(n_sym=3, deg=10, n_states=30)), (3, dict(n_sym=64, deg=20, n_states=500, lo=5, hi=40)),
                                     (4, dict(n_sym=2, deg=6, n_states=12, hi=9, n_pairs=80)), (5, dict(n_sym=6, deg=12, n_states=2000, n_pairs=3000, hi=16))])
def test_small_corpora(seed, kw):
    from test_gpu_parity import ambiguous
    w, c = ambiguous(seed, **kw)
    rng = np.random.default_rng(seed)
    c.weight[:] = rng.uniform(0.5, 3.0, c.n_pairs)
    # a few pairs that have no derivation, an empty pair, a one-sided pair
    extra = Corpus.from_lists([([2, 3, 2], [3]), ([], []), ([2], []), ([3, 3, 3, 3, 3, 3, 3, 3], [2, 2])], np.array([1.0, 2.0, 0.5, 1.5]))
    c = Corpus(np.concatenate([c.in_off, c.in_off[-1] + extra.in_off[1:]]), np.concatenate([c.in_sym, extra.in_sym]),
               np.concatenate([c.out_off, c.out_off[-1] + extra.out_off[1:]]), np.concatenate([c.out_sym, extra.out_sym]),
               np.concatenate([c.weight, extra.weight]))
    _same(_build(w, c, False), _build(w, c, True))


def test_config2_and_a_slice_of_config4():
    for name, n in (("c2", None), ("c4", 200000)):
        w, c = synth.make_config(name, n_pairs=n)
        host, dev = _build(w, c, False), _build(w, c, True)
        _same(host, dev)
        if name == "c4":  # (the 50 000 pairs of c2 take both builders under 0.1 s)
            assert dev["seconds"] < host["seconds"]


def test_corpora_outside_its_scope_go_to_the_host_builder(oracle):
    """a cyclic lattice, and lattices wider than the one-per-lane layout: CARMEL_HIP_GPU_BUILD=1 must give what the host
    builder gives (it IS the host builder then)"""
    # *e*:*e* loop -> cyclic derivation lattices (derivations.h:726-728)
    src = np.array([0, 0, 1, 1], np.uint32)
    dst = np.array([1, 2, 0, 2], np.uint32)
    isym = np.array([0, 2, 0, 3], np.uint32)
    osym = np.array([0, 2, 0, 3], np.uint32)
    w = Wfst(3, 2, src, dst, isym, osym, np.log([0.5, 0.5, 0.4, 0.6]))
    c = Corpus.from_lists([([2], [2]), ([3], [3])])
    _same(_build(w, c, False), _build(w, c, True))
    w = synth.random_wfst(20000, 12, n_sym=4, p_eps=0.1, seed=5)
    c = synth.random_walk_corpus(w, 100, min_arcs=5, max_arcs=16, seed=5, out_degree=12)  # thousands of states per lattice
    a, b = _build(w, c, False), _build(w, c, True)
    _same(a, b)
    assert a["stats"][4] / a["stats"][1] > 96


@pytest.mark.parametrize("name", ["cyclic", "wide", "long", "tagging-small", "hub"])
def test_host_layouts_get_their_tables_from_the_device(name, capfd, hipopt):
    """the corpora the device builder leaves to the host (bundles of cyclic lattices, one-per-wavefront lattices, mixtures, a hub
    arc with buckets of its own): the host lays them out and the DEVICE sorts the posterior slots by arc and builds the
    transposition tables (gpu_tables_for_host_layout) -- the image the host's own counting sorts leave, byte for byte, and the
    same counts"""
    hipopt.set("timing", "1")
    if name == "cyclic":
        rng = np.random.default_rng(3)
        w = synth.random_wfst(40, 5, n_sym=4, p_eps=0.25, seed=3)  # *e*:*e* arcs: cycles in the lattices
        c = synth.random_walk_corpus(w, 300, min_arcs=3, max_arcs=10, seed=3, out_degree=5)
    elif name == "wide":
        w = synth.random_wfst(20000, 12, n_sym=4, p_eps=0.1, seed=5)
        c = synth.random_walk_corpus(w, 100, min_arcs=5, max_arcs=16, seed=5, out_degree=12)
    elif name == "long":
        w, c = synth.make_config("long", n_pairs=300)
    elif name == "tagging-small":
        w, c = _tagging(2)
    else:
        w = synth.random_wfst(3, 2, n_sym=2, p_eps=0.0, seed=9)  # every pair crosses the same few arcs thousands of times over
        c = synth.random_walk_corpus(w, 4000, min_arcs=20, max_arcs=40, seed=9, out_degree=2)
    host = _build(w, c, False, device_tables=False)
    capfd.readouterr()
    hyb = _build(w, c, False, device_tables=True)
    err = capfd.readouterr().err
    assert "slots by arc (left to the device)" in err and "items of the host layout" in err
    _same(host, hyb, atomics=name == "hub")


def _tagging(reps):
    from oracle import binding as ob
    from conftest import GOLDEN
    g = lambda n: open(os.path.join(GOLDEN, n)).read()
    oc = ob.OracleCascade([g("tagging.fsa"), g("tagging.fst")])
    a = oc.composed().arrays()
    w = Wfst(a["n_states"], a["final"], a["src"], a["dst"], a["isym"], a["osym"], a["logw"], a["group"])
    ca = oc.corpus(g("tagging.data") * reps).arrays()
    return w, Corpus(ca["in_off"], ca["in_sym"], ca["out_off"], ca["out_sym"], ca["weight"])


@pytest.mark.parametrize("name", ["c4a", "tagging", "forced"])
def test_windowed_corpora_are_built_on_the_device(name, capfd, hipopt):
    """round 3: corpora with WINDOWED lane groups (lattices of up to 1 023 states whose arcs span few states of the
    topological numbering: the tagging cascade, config c4a) -- larger per-pair capacities for the exploration, the span and
    the ring of every lattice, plain groups then windowed groups, parked-value rows -- leave the host builder's image, byte
    for byte.  `forced`: windows forced onto small random lattices (CARMEL_HIP_LANE_WINDOW_MIN)."""
    hipopt.set("timing", "1")
    if name == "c4a":
        w, c = synth.make_config("c4a", n_pairs=30000)
    elif name == "tagging":
        w, c = _tagging(6)
    else:
        hipopt.set("lane_window_min", "6")
        w = synth.random_wfst(40, 4, n_sym=4, p_eps=0.1, seed=73)
        c = synth.random_walk_corpus(w, 3000, min_arcs=4, max_arcs=50, seed=73, out_degree=4)
    host = _build(w, c, False)
    capfd.readouterr()
    dev = _build(w, c, True)
    err = capfd.readouterr().err
    assert "lattices built on the GPU" in err, "the device builder gave the corpus back to the host"
    _same(host, dev)
    from carmel_amd.trainer import HipForwardBackward
    fb = HipForwardBackward(w, c)
    assert fb.lattice_stats.n_windowed_pairs > 0
    fb.close()
    # (no timing claim on 6 030 pairs: both builders take ~0.03 s there and a 256-thread host wins as often as not;
    # test_config2_and_a_slice_of_config4 compares the two on 200 000 pairs)


@pytest.mark.parametrize("name", ["c4a", "tagging", "plain", "cyclic", "hub"])
def test_tile_weights_from_the_table_are_the_same_bits(name, capfd, hipopt):
    """the tile passes' two sources of weights -- X, written bucket by bucket by the first pass (CARMEL_HIP_TILE_GATHER=0; what a
    WFST of few items an arc or a table beyond the cache gets), and the WFST's table through the arc of every tile-major item
    (t_t_arc; no first pass) -- are the same values at the same positions: fused-lane layouts with and without windows, the tile
    sweep's layout, bundles beside lanes, a hub arc with buckets of its own"""
    hipopt.set("timing", "1")
    hipopt.set("trans_runs", "0")  # (run-length indices keep X)
    if name == "c4a":
        w, c = synth.make_config("c4a", n_pairs=30000)
    elif name == "tagging":
        w, c = _tagging(6)
    elif name == "plain":
        w, c = synth.make_config("c2", n_pairs=20000)
    elif name == "cyclic":
        w = synth.random_wfst(40, 5, n_sym=4, p_eps=0.25, seed=3)
        c = synth.random_walk_corpus(w, 300, min_arcs=3, max_arcs=10, seed=3, out_degree=5)
    else:
        w = synth.random_wfst(3, 2, n_sym=2, p_eps=0.0, seed=9)
        c = synth.random_walk_corpus(w, 4000, min_arcs=20, max_arcs=40, seed=9, out_degree=2)
    res = {}
    for g in ("1", "0"):
        hipopt.set("tile_gather", g)
        capfd.readouterr()
        res[g] = _build(w, c, name in ("c4a", "tagging", "plain"))
        assert ("tile weights from the table" in capfd.readouterr().err) == (g == "1")
    _same(res["1"], res["0"], atomics=name == "hub")
