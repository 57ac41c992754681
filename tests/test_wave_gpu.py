"""GPU: the one-lattice-per-wavefront sweep (sweep_wave_kernel; csrc/lattice.hpp WaveDesc) through the C-ABI against the
oracle -- forced onto small ambiguous corpora of every shape (CARMEL_HIP_WAVE_MIN_WIDTH=0: however narrow), on weighted
pairs, on lattices with zero-weight arcs, and where it is the natural choice: few long, wide lattices."""
import numpy as np
import pytest

from carmel_amd import synth
from carmel_amd.model import Corpus

pytestmark = pytest.mark.gpu


def _check(oracle, w, c, iters=3, expect_waves=True, rtol=1e-7):
    from carmel_amd.trainer import HipForwardBackward
    fb = HipForwardBackward(w, c)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    ow.normalize(0, 0.0)
    for it in range(iters):
        lp, wlp = fb.estimate(per_pair=True)
        r = oracle.estimate(ow, oc)
        ok = r["has_deriv"]
        assert np.array_equal(ok, fb.has_deriv.astype(bool))
        np.testing.assert_allclose(fb.pair_logprob[ok], r["pair_logprob"][ok], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(fb.counts(), np.exp(r["counts_ln"]), rtol=rtol, atol=1e-12)
        assert lp == pytest.approx(r["sum_logprob"], rel=1e-10, abs=1e-9)
        assert wlp == pytest.approx(r["sum_weighted_logprob"], rel=1e-10, abs=1e-9)
        fb.maximize(1.0)
        ow.set_logw(fb.weights())
    fb.close()


@pytest.mark.parametrize("ring", ["1", "0"])
@pytest.mark.parametrize("seed,lane_states", [(1, "0"), (2, "0"), (3, "12"), (4, "0"), (5, "20"), (6, "0")])
def test_forced_wave_sweep_matches_oracle(oracle, hipopt, seed, lane_states, ring):
    """ring = 1: the ring form where a lattice allows it (values in a ring of LDS slots, forward values parked in HBM);
    ring = 0: every lattice with all of its values in LDS"""
    hipopt.set("wave_min_width", "0")
    hipopt.set("wave_ring", ring)
    hipopt.set("lane_states", lane_states)
    w = synth.random_wfst(14 + 4 * seed, 4 + seed % 3, n_sym=3 + seed % 2, p_eps=0.12, seed=90 + seed)
    c = synth.random_walk_corpus(w, 400, min_arcs=3, max_arcs=14 + 4 * seed, seed=90 + seed, out_degree=4 + seed % 3)
    rng = np.random.default_rng(seed)
    c.weight[:] = rng.uniform(0.5, 3.0, c.n_pairs)  # pair weights: beta'[goal] = ln weight - ln p
    if seed == 6:  # zero-weight arcs: -inf terms must neither poison a state's sum nor count
        w.logw[rng.random(w.n_arcs) < 0.15] = -np.inf
        # (pairs left without a path of non-zero weight divide by a zero probability in the reference: keep the others)
        ow = oracle.OracleWfst.from_arrays(w)
        ow.normalize(0, 0.0)
        r = oracle.estimate(ow, oracle.OracleCorpus.from_arrays(c))
        keep = np.nonzero(r["has_deriv"] & np.isfinite(r["pair_logprob"]))[0]
        assert 50 < len(keep) < c.n_pairs
        c = c.subset(keep)
    _check(oracle, w, c)


def test_wide_levels_span_several_rows(oracle, hipopt):
    """levels with more than 64 arcs (several rows per level) and states with many in-arcs: a dense little transducer"""
    hipopt.set("wave_min_width", "0")
    hipopt.set("lane_states", "0")
    w = synth.clustered_wfst(12 * 4 + 1, 48, members=12, n_sym=6, n_in_sym=3, seed=21)
    c = synth.clustered_walk_corpus(w, 60, 48, members=12, min_arcs=3, max_arcs=25, seed=21)
    _check(oracle, w, c, iters=2)


@pytest.mark.parametrize("ring", ["1", "0"])
def test_long_workload_slice_uses_the_wave_sweep(oracle, hipopt, ring):
    """bench.py --config long: these lattices go one per wavefront by the builder's own rule"""
    from carmel_amd.trainer import HipForwardBackward
    hipopt.set("wave_ring", ring)
    w, c = synth.make_config("long", n_pairs=40)
    fb = HipForwardBackward(w, c)
    assert fb.lattice_stats.n_windowed_pairs == 0 and fb.lattice_stats.n_bundles == 40
    fb.close()
    _check(oracle, w, c, iters=2)


@pytest.mark.parametrize("ring", ["1", "0"])
@pytest.mark.parametrize("seed", [2, 3, 5])
def test_gathered_and_laid_out_weights_give_the_same_bits(hipopt, capfd, seed, ring):
    """the sweep's two sources of weights -- the WFST's table through the records' arc ids (tables the caches hold: the default
    here) and wcache, written in lattice order by the transposition's weight pass (CARMEL_HIP_WAVE_GATHER=0; what a larger
    table gets) -- and its two ways out for the posteriors -- `post` and the tile pass, or every posterior straight to its item's
    place in XC (CARMEL_HIP_WAVE_XC=1: forced here, chosen where a row's items are neighbours in XC) -- are the same numbers in
    the same places of the same sums"""
    from carmel_amd.trainer import HipForwardBackward
    hipopt.set("timing", "1")
    hipopt.set("wave_min_width", "0")
    hipopt.set("wave_ring", ring)
    hipopt.set("lane_states", "12" if seed == 3 else "0")  # (seed 3: lane lattices beside the waves)
    if seed == 5:
        w = synth.clustered_wfst(12 * 4 + 1, 48, members=12, n_sym=6, n_in_sym=3, seed=21)  # levels of several rows
        c = synth.clustered_walk_corpus(w, 60, 48, members=12, min_arcs=3, max_arcs=25, seed=21)
    else:
        w = synth.random_wfst(14 + 4 * seed, 4 + seed % 3, n_sym=3 + seed % 2, p_eps=0.12, seed=90 + seed)
        c = synth.random_walk_corpus(w, 400, min_arcs=3, max_arcs=14 + 4 * seed, seed=90 + seed, out_degree=4 + seed % 3)
    runs = []
    for g, x in (("1", "1"), ("0", "0"), ("1", "0"), ("0", "1")):
        hipopt.set("wave_gather", g)
        hipopt.set("wave_xc", x)
        capfd.readouterr()
        fb = HipForwardBackward(w, c)
        assert ("wave posteriors straight to XC" in capfd.readouterr().err) == (x == "1")
        assert fb.lattice_stats.n_bundles > 0  # (wave lattices are counted with the bundles)
        assert bool(fb.weight_source & 2) == (g == "1") and bool(fb.weight_source & 4) == (x == "1")
        out = []
        for _ in range(3):
            lp, wlp = fb.estimate(per_pair=True)
            out.append((lp, wlp, fb.pair_logprob.copy(), fb.counts().copy()))
            fb.maximize(1.0)
        out.append(fb.weights().copy())
        fb.close()
        runs.append(out)
    for other in runs[1:]:
        for a, b in zip(runs[0][:-1], other[:-1]):
            assert a[0] == b[0] and a[1] == b[1]
            assert np.array_equal(a[2], b[2])
            # (an arc whose items lie in several buckets is summed with one atomic per bucket: the last bits may differ)
            assert (a[3] != b[3]).sum() <= 16 and np.allclose(a[3], b[3], rtol=1e-13, atol=0)
        np.testing.assert_allclose(runs[0][-1], other[-1], rtol=1e-12, atol=0)


def test_waves_refuse_the_gather_formulation(hipopt):
    """CARMEL_HIP_TRANSPOSE=0 (the A/B switch of the lane corpora) has no weight pass: wave lattices say so instead of sweeping
    over weights nobody wrote"""
    from carmel_amd.trainer import HipForwardBackward
    hipopt.set("transpose", "0")
    w, c = synth.make_config("long", n_pairs=8)
    with pytest.raises(Exception, match="blocked transposition"):
        HipForwardBackward(w, c)
