"""CPU: the product's host-side lattice builder and HBM layout against the oracle (no GPU compute)."""
import numpy as np
import pytest

from carmel_amd import synth
from carmel_amd.model import Corpus, Wfst
import helpers as H
from helpers import host_lattices, numpy_sweep


def ambiguous(seed, n_states=40, deg=8, n_sym=4, n_pairs=60, p_eps=0.15):
    w = synth.random_wfst(n_states, deg, n_sym=n_sym, p_eps=p_eps, seed=seed)
    c = synth.random_walk_corpus(w, n_pairs, min_arcs=3, max_arcs=12, seed=seed, out_degree=deg)
    return w, c


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lattice_matches_oracle_structure(oracle, seed):
    w, c = ambiguous(seed)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    img = host_lattices(w, c, small_pairs=8, small_states=512, lane_states=0)
    r = oracle.estimate(ow, oc)
    assert np.array_equal(img["has_deriv"].astype(bool), r["has_deriv"])
    # exploration statistics are the reference's (derivations.h:191-247): same DFS, same counts
    assert img["explored_states"] == int(r["stats"][0])
    assert img["explored_arcs"] == int(r["stats"][1])
    # per pair: identical multiset of (WFST arc id) uses and identical state counts, cycles excepted (the product
    # additionally drops states that cannot reach the goal, which the reference keeps with zero backward mass)
    by_pair = {}
    for b in img["bundles"]:
        if b["n_pairs"] != 1:
            continue
        pid = int(img["pair_id"][b["pair_base"]])
        arcs = img["out_arcs"][int(b["out_base"]):int(b["out_base"]) + int(b["n_arcs"]), 1]
        by_pair[pid] = (int(b["n_states"]), np.sort(arcs))
    for pid, (ns, arcs) in list(by_pair.items())[:20]:
        L = oracle.lattice(ow, oc, pid)
        if L["n_back_edges"] == 0:
            assert ns == L["n_states"]
            assert np.array_equal(arcs, np.sort(L["arcid"]))


@pytest.mark.parametrize("seed,small_pairs,lane_states", [(1, 8, 0), (2, 64, 0), (5, 3, 0), (1, 8, 96), (2, 8, 20),
                                                          (6, 8, 12)])
def test_layout_sweep_matches_oracle_estep(oracle, seed, small_pairs, lane_states):
    w, c = ambiguous(seed, n_pairs=150)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    img = host_lattices(w, c, small_pairs=small_pairs, small_states=1024, lane_states=lane_states)
    if lane_states:
        assert len(img["lane_groups"]) > 0
        plain = np.concatenate([img["lane_nstates"][int(g["pair_base"]):int(g["pair_base"]) + 64] for g in img["lane_groups"]
                                if not g["window"]] or [np.zeros(1, np.uint32)])
        assert int(plain.max()) <= lane_states  # windowed groups take larger lattices (LaneGroup::window)
    r = oracle.estimate(ow, oc)
    counts, plp = numpy_sweep(img, w.logw, c.n_pairs)
    ok = r["has_deriv"]
    np.testing.assert_allclose(plp[ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9)
    ref = np.exp(r["counts_ln"])
    np.testing.assert_allclose(counts, ref, rtol=1e-8, atol=1e-12)
    # size-independent property: the final state has no out-arcs, so every derivation enters it exactly once:
    # expected counts of the arcs INTO it sum to the corpus weight
    assert abs(counts[w.dst == w.final].sum() - c.weight[ok].sum()) < 1e-9 * c.n_pairs


def test_levels_are_topological(oracle):
    w, c = ambiguous(9, n_states=30, deg=10, n_sym=3, n_pairs=40)
    img = host_lattices(w, c, small_pairs=16, small_states=2048, lane_states=0)
    for b in img["bundles"]:
        if b["flags"] & 1:
            continue
        ns, ob = int(b["n_states"]), int(b["off_base"])
        lv = img["level_off"][int(b["level_base"]):int(b["level_base"]) + int(b["n_levels"]) + 1]
        assert lv[0] == 0 and lv[-1] == ns and np.all(np.diff(lv.astype(np.int64)) > 0)
        level_of = np.searchsorted(lv, np.arange(ns), side="right") - 1
        ioff = img["in_off"][ob:ob + ns + 1]
        ia = img["in_arcs"][int(b["in_base"]):int(b["in_base"]) + int(b["n_arcs"])]
        for s in range(ns):
            src = ia[ioff[s]:ioff[s + 1], 0]
            if len(src):
                assert level_of[src].max() + 1 == level_of[s]  # longest-path level


def test_empty_and_no_derivation_pairs(oracle):
    # 3-state toy: (0 -a:x-> 1 -b:y-> 2), final 2; pairs: good, wrong symbol, empty/empty
    w = Wfst(3, 2, [0, 1], [1, 2], [2, 3], [2, 3], np.log([1.0, 1.0]))
    c = Corpus.from_lists([([2, 3], [2, 3]), ([2, 2], [2, 3]), ([], [])])
    img = host_lattices(w, c)
    assert img["has_deriv"].tolist() == [1, 0, 0]
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    assert oracle.estimate(ow, oc)["has_deriv"].tolist() == [True, False, False]


def test_cyclic_lattice_reference_order(oracle):
    # *e*:*e* 2-cycle between states 1 and 2 on the way to the goal: the reference drops the back-edge paths
    # (derivations.h:726-728); the product must drop the same ones
    src = [0, 1, 1, 2, 2]
    dst = [1, 2, 3, 1, 3]
    isym = [2, 0, 3, 0, 3]
    osym = [2, 0, 3, 0, 3]
    w = Wfst(4, 3, src, dst, isym, osym, np.log([1.0, 0.3, 0.7, 0.4, 0.6]))
    c = Corpus.from_lists([([2, 3], [2, 3])])
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    L = oracle.lattice(ow, oc, 0)
    assert L["n_back_edges"] > 0
    img = host_lattices(w, c)
    assert img["n_cyclic"] == 1
    r = oracle.estimate(ow, oc)
    counts, plp = numpy_sweep(img, w.logw, 1)
    np.testing.assert_allclose(plp[0], r["pair_logprob"][0], rtol=1e-12)
    np.testing.assert_allclose(counts, np.exp(r["counts_ln"]), rtol=1e-10, atol=1e-300)


def test_blocked_transposition_tables():
    """the two-pass LDS-blocked transposition moves exactly what the random gather / segmented sum would"""
    from carmel_amd import synth
    w = synth.random_wfst(3000, 6, seed=5)
    c = synth.random_walk_corpus(w, 90000, min_arcs=3, max_arcs=8, seed=5, out_degree=6)
    img = H.host_lattices(w, c, threads=4)
    tr = img["transpose"]
    assert len(tr["split_arcs"]) >= 1
    assert tr["n_items"] == img["total_arcs"] and len(tr["buckets"]) > 0
    n_post = tr["n_post"]
    # arc of every valid position, from the arc-sorted slot list
    arc_at = np.full(n_post, -1, np.int64)
    for a in range(tr["n_arcs"]):
        arc_at[tr["slot_pos"][int(tr["arc_off"][a]):int(tr["arc_off"][a + 1])]] = a
    valid = arc_at >= 0
    assert valid.sum() == tr["n_items"]
    # buckets tile the arc-sorted order; a hub arc (here: those out of the start state, > 16384 uses) is split
    assert int(tr["buckets"]["n_items"].sum()) == tr["n_items"]
    assert (tr["buckets"]["n_items"] <= 16384).all() and (tr["buckets"]["n_arcs"] <= 16384).all()
    uses = tr["arc_off"][1:] - tr["arc_off"][:-1]
    assert len(tr["split_arcs"]) == int((uses > 16384).sum())
    assert int(((tr["buckets"]["flags"] & 2) != 0).sum()) >= int((uses > 2048).sum())
    rng = np.random.default_rng(0)
    logw = rng.normal(size=tr["n_arcs"])
    wc = H.transpose_weights(tr, logw, n_post)
    assert np.array_equal(wc[valid], logw[arc_at[valid]])
    post = rng.random(n_post)
    got = H.transpose_counts(tr, post)
    want = np.bincount(arc_at[valid], weights=post[valid], minlength=tr["n_arcs"])
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    # the items of (bucket, tile) are one run in both orders: t_src is increasing inside a tile
    for t in range(len(tr["tile_base"]) - 1):
        seg = tr["t_src"][int(tr["tile_base"][t]):int(tr["tile_base"][t + 1])].astype(np.int64)
        assert (np.diff(seg) > 0).all()


@pytest.mark.parametrize("seed,window,wmin", [(1, 64, 4), (2, 32, 4), (3, 16, 6), (4, 8, 4), (5, 64, 40)])
def test_windowed_lane_layout(oracle, hipopt, seed, window, wmin):
    """windowed lane groups (LaneGroup::window): lattices whose arcs span fewer than `window` states of the topological
    numbering keep only a ring of that many rows; the host restatement of the sweep poisons stale ring rows, so an arc that
    reached outside the ring would surface as a NaN.  Counts and probabilities are the oracle's."""
    hipopt.set("lane_window", str(window))
    hipopt.set("lane_window_min", str(wmin))
    w = synth.random_wfst(12 + 5 * seed, 3 + seed % 3, n_sym=3 + seed, p_eps=0.1, seed=70 + seed)
    c = synth.random_walk_corpus(w, 150, min_arcs=4, max_arcs=30 + 10 * seed, seed=70 + seed, out_degree=3 + seed % 3)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    img = host_lattices(w, c, small_pairs=8, small_states=1024, lane_states=96)
    wins = [int(g["window"]) for g in img["lane_groups"] if g["window"]]
    assert wins and max(wins) <= window
    r = oracle.estimate(ow, oc)
    counts, plp = numpy_sweep(img, w.logw, c.n_pairs)
    ok = r["has_deriv"]
    assert not np.isnan(counts).any()
    np.testing.assert_allclose(plp[ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(counts, np.exp(r["counts_ln"]), rtol=1e-8, atol=1e-12)


def test_layout_and_tables_do_not_depend_on_the_thread_count():
    """the host builder's threads (derivations per pair, slots by arc, buckets, the tile-major counting sort) produce the bytes
    one thread produces -- what the GPU builder is compared with (tests/test_lattice_gpu.py) must not depend on the host"""
    w = synth.random_wfst(50000, 20, seed=3)
    c = synth.random_walk_corpus(w, 26000, min_arcs=5, max_arcs=40, seed=10, out_degree=20)
    a, b = host_lattices(w, c, threads=1), host_lattices(w, c, threads=8)
    ta, tb = a["transpose"], b["transpose"]
    assert ta["n_items"] > 2 * (1 << 18)  # several threads in the tile-major sort
    for k in ("tile_base", "b_arc", "b_rank", "b_src", "t_pos", "t_src", "split_arcs", "arc_off", "slot_pos"):
        assert np.array_equal(ta[k], tb[k]), k
    assert ta["buckets"].tobytes() == tb["buckets"].tobytes()
    for k in ("lane_fwd", "lane_bwd", "lane_pair", "lane_nstates", "lane_groups"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("name,pairs", [("toya", None), ("c4a", 400), ("long", 6)])
def test_clustered_workloads_layout_matches_oracle(oracle, name, pairs):
    """the clustered (ambiguous) transducers of bench.py's c4a / long workloads: every lattice state is a real log-semiring
    sum (members^2 arcs between neighbouring positions); host builder + layout (windowed lane groups, bundles) swept in
    numpy against the oracle's E-step"""
    w, c = synth.make_config(name, n_pairs=pairs)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    img = host_lattices(w, c, small_pairs=8, small_states=1024, lane_states=96)
    assert img["total_arcs"] / img["total_states"] > (2.0 if name != "long" else 6.0)
    r = oracle.estimate(ow, oc)
    assert r["has_deriv"].all() and img["has_deriv"].all()
    counts, plp = numpy_sweep(img, w.logw, c.n_pairs)
    np.testing.assert_allclose(plp, r["pair_logprob"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(counts, np.exp(r["counts_ln"]), rtol=1e-8, atol=1e-12)
    assert abs(counts[w.dst == w.final].sum() - c.n_pairs) < 1e-9 * c.n_pairs


@pytest.mark.parametrize("seed,lane_states", [(1, 0), (2, 0), (3, 12), (4, 0), (5, 20)])
def test_wave_layout_on_ambiguous_lattices(oracle, hipopt, seed, lane_states):
    """the one-lattice-per-wavefront layout (WaveDesc) forced onto small ambiguous corpora (every acyclic lattice no lane
    takes, however narrow): rows of 64 records that never straddle a level, forward records pointing at their arc's backward
    position; swept in numpy exactly as sweep_wave_kernel walks it, against the oracle -- and the blocked transposition
    tables must cover the wave slots (weights in, posteriors out) exactly"""
    hipopt.set("wave_min_width", "0")
    w = synth.random_wfst(14 + 4 * seed, 4 + seed % 3, n_sym=3 + seed % 2, p_eps=0.12, seed=90 + seed)
    c = synth.random_walk_corpus(w, 120, min_arcs=3, max_arcs=14 + 4 * seed, seed=90 + seed, out_degree=4 + seed % 3)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    img = host_lattices(w, c, small_pairs=8, small_states=1024, lane_states=lane_states)
    wv = img["waves"]
    assert len(wv["descs"]) > 10 and wv["slot_base"] % H.TRANS_TILE == 0
    cyc = sum(1 for b in img["bundles"] if b["flags"] & 1)
    assert len(img["bundles"]) == cyc  # bundles are left with the cyclic lattices only
    r = oracle.estimate(ow, oc)
    counts, plp = numpy_sweep(img, w.logw, c.n_pairs)
    ok = r["has_deriv"]
    np.testing.assert_allclose(plp[ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(counts, np.exp(r["counts_ln"]), rtol=1e-8, atol=1e-12)
    # transposition: every wave slot gets its arc's weight, and the posteriors written there come back to its arc
    tr = img["transpose"]
    n_post = tr["n_post"]
    arc_at = np.full(n_post, -1, np.int64)
    for a in range(tr["n_arcs"]):
        arc_at[tr["slot_pos"][int(tr["arc_off"][a]):int(tr["arc_off"][a + 1])]] = a
    nb = len(wv["bwd"])
    wslots = arc_at[wv["slot_base"]:wv["slot_base"] + nb]
    want = np.where(wv["bwd_arc"] == 0xffffffff, -1, wv["bwd_arc"].astype(np.int64))
    assert np.array_equal(wslots, want)
    rng = np.random.default_rng(seed)
    logw = rng.normal(size=tr["n_arcs"])
    wc = H.transpose_weights(tr, logw, wv["slot_base"] + nb)
    valid = want >= 0
    assert np.array_equal(wc[wv["slot_base"]:][valid], logw[want[valid]])
    post = rng.random(n_post)
    got = H.transpose_counts(tr, post)
    v = arc_at >= 0
    np.testing.assert_allclose(got, np.bincount(arc_at[v], weights=post[v], minlength=tr["n_arcs"]), rtol=1e-12, atol=0)


FUSED_TILE = 1024  # lattice.hpp: LANE_FUSED_TILE


def test_tile_sweep_layout(hipopt):
    """LatticeSet::tile_sweep (lattice.hpp, TILE_SWEEP_*): a corpus of small plain lane lattices is laid out in tiles of 8192
    positions that no lane group straddles, with room in a workgroup's LDS for the values of a tile's groups; the tables of
    the blocked transposition are built on those tiles and move exactly what they move on the five-kernel layout; the sweep
    itself (numpy model over the lane groups) gives the same counts on either layout."""
    from carmel_amd import synth
    w = synth.random_wfst(3000, 6, seed=11)
    c = synth.random_walk_corpus(w, 30000, min_arcs=3, max_arcs=40, seed=11, out_degree=6)
    img = H.host_lattices(w, c, threads=4)
    tr = img["transpose"]
    T, tg, g = tr["tile"], tr["tile_group"], img["lane_groups"]
    assert T == 8192 and len(tg) >= 2 and tg[0] == 0 and tg[-1] == len(g) and (np.diff(tg.astype(np.int64)) > 0).all()
    assert (np.diff(tg.astype(np.int64)) <= 16).all()
    for t in range(len(tg) - 1):
        gs = g[int(tg[t]):int(tg[t + 1])]
        lo, hi = gs["stream_base"].astype(np.int64), gs["stream_base"].astype(np.int64) + gs["maxlen"].astype(np.int64) * 64
        assert lo.min() >= t * T and hi.max() <= (t + 1) * T and (lo[1:] == hi[:-1]).all()
        need = np.maximum(gs["max_states"], gs["maxlen"] + 1).astype(np.int64)
        assert (gs["window"] == 0).all() and (gs["maxlen"] <= 48).all() and need.sum() <= 126
        # a group's values start where the previous group's end
        assert (gs["spill_row"].astype(np.int64) == np.concatenate([[0], np.cumsum(need)[:-1]])).all()
    n_post = tr["n_post"]
    assert n_post % T == 0 and len(tr["tile_base"]) - 1 == n_post // T
    rng = np.random.default_rng(1)
    logw = rng.normal(size=tr["n_arcs"])
    arc_at = np.full(n_post, -1, np.int64)
    for a in range(tr["n_arcs"]):
        arc_at[tr["slot_pos"][int(tr["arc_off"][a]):int(tr["arc_off"][a + 1])]] = a
    valid = arc_at >= 0
    wc = H.transpose_weights(tr, logw, n_post)
    assert np.array_equal(wc[valid], logw[arc_at[valid]])
    post = rng.random(n_post)
    np.testing.assert_allclose(H.transpose_counts(tr, post), np.bincount(arc_at[valid], weights=post[valid], minlength=tr["n_arcs"]),
                               rtol=1e-12, atol=0)
    counts, plp = numpy_sweep(img, w.logw, c.n_pairs)
    hipopt.set("tile_sweep", "0")
    for fused, tile in (("1", FUSED_TILE), ("0", 16384)):  # without the tile sweep: the fused-lane layout, or (A/B) the round-1 tiles
        hipopt.set("lane_fused", fused)
        img0 = H.host_lattices(w, c, threads=4)
        assert img0["transpose"]["tile"] == tile and len(img0["transpose"]["tile_group"]) == 0
        counts0, plp0 = numpy_sweep(img0, w.logw, c.n_pairs)
        assert np.array_equal(plp, plp0)
        np.testing.assert_allclose(counts, counts0, rtol=1e-12, atol=0)


def test_tile_sweep_layout_is_for_small_plain_lattices_only(hipopt):
    """a corpus with one lattice beyond 48 arcs, a windowed group or a one-per-wavefront lattice is not laid out for the tile
    sweep: a lanes-only corpus gets the fused-lane layout instead (LatticeSet::lane_fused, next test)"""
    from carmel_amd import synth
    w = synth.random_wfst(3000, 6, seed=12)
    c = synth.random_walk_corpus(w, 3000, min_arcs=3, max_arcs=60, seed=12, out_degree=6)
    tr = H.host_lattices(w, c, threads=4)["transpose"]
    assert tr["tile"] == FUSED_TILE and len(tr["tile_group"]) == 0
    hipopt.set("lane_fused", "0")
    tr = H.host_lattices(w, c, threads=4)["transpose"]
    assert tr["tile"] == 16384 and len(tr["tile_group"]) == 0


def test_fused_lane_layout(oracle, hipopt):
    """LatticeSet::lane_fused (lattice.hpp, LANE_FUSED_TILE): a corpus of lane lattices the tile sweep does not take -- windowed
    groups, lattices above 48 arcs -- is laid out with every lane group on a tile of 1024 positions (16 rows of its 64 lanes) of
    its own, so that the wavefront sweeping a group owns whole tiles of the transposition and sends its posteriors out itself.
    The tables move what they move on the 16384-position layout; the numpy sweep gives the oracle's counts on both."""
    from carmel_amd import synth
    hipopt.set("lane_window_min", "12")  # windows on small lattices too
    w = synth.clustered_wfst(3 * 40 + 1, 12, members=3, seed=8)
    c = synth.clustered_walk_corpus(w, 700, 12, members=3, min_arcs=3, max_arcs=30, seed=8)
    img = H.host_lattices(w, c, threads=4)
    tr, g = img["transpose"], img["lane_groups"]
    T = FUSED_TILE
    assert tr["tile"] == T and len(tr["tile_group"]) == 0 and len(img["bundles"]) == 0 and len(img["waves"]["descs"]) == 0
    assert (g["window"] > 0).any() and (g["window"] == 0).any()
    sb = g["stream_base"].astype(np.int64)
    assert (sb % T == 0).all() and (np.diff(sb) >= g["maxlen"][:-1].astype(np.int64) * 64).all()
    assert tr["n_post"] % T == 0 and len(tr["tile_base"]) - 1 == tr["n_post"] // T
    # every item of a tile belongs to the one group that owns the tile
    owner = np.searchsorted(sb, np.arange(len(tr["tile_base"]) - 1) * T, side="right") - 1
    for t in (0, len(owner) // 2, len(owner) - 1):
        i0, i1 = int(tr["tile_base"][t]), int(tr["tile_base"][t + 1])
        p = t * T + tr["t_pos"][i0:i1].astype(np.int64)
        assert (p >= sb[owner[t]]).all() and (p < sb[owner[t]] + int(g["maxlen"][owner[t]]) * 64).all()
    n_post = tr["n_post"]
    rng = np.random.default_rng(2)
    logw = rng.normal(size=tr["n_arcs"])
    arc_at = np.full(n_post, -1, np.int64)
    for a in range(tr["n_arcs"]):
        arc_at[tr["slot_pos"][int(tr["arc_off"][a]):int(tr["arc_off"][a + 1])]] = a
    valid = arc_at >= 0
    assert np.array_equal(H.transpose_weights(tr, logw, n_post)[valid], logw[arc_at[valid]])
    post = rng.random(n_post)
    np.testing.assert_allclose(H.transpose_counts(tr, post), np.bincount(arc_at[valid], weights=post[valid], minlength=tr["n_arcs"]),
                               rtol=1e-12, atol=0)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    r = oracle.estimate(ow, oc)
    counts, plp = numpy_sweep(img, w.logw, c.n_pairs)
    ok = r["has_deriv"]
    np.testing.assert_allclose(plp[ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(counts, np.exp(r["counts_ln"]), rtol=1e-8, atol=1e-12)
    hipopt.set("lane_fused", "0")
    img0 = H.host_lattices(w, c, threads=4)
    assert img0["transpose"]["tile"] == 16384
    counts0, plp0 = numpy_sweep(img0, w.logw, c.n_pairs)
    assert np.array_equal(plp, plp0)
    np.testing.assert_allclose(counts, counts0, rtol=1e-12, atol=0)
