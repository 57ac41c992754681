"""GPU: carmel --matrix-fb (carmel_hip_set_matrix_fb, csrc/matrix_fb.hip) against the oracle's restatement of
forward_backward::matrix_compute / estimate_matrix (oracle/matrix.hpp; train.cc:698-860), against the lattice sweep of the same
trainer, and through the front end on the reference's recorded tutorial runs."""
import json
import os
import re

import numpy as np
import pytest

from carmel_amd import synth
from carmel_amd.model import Corpus, Wfst, NORM_CONDITIONAL
from test_matrix_fb_oracle import model
from test_cli_gpu import ITER, NUM, run

pytestmark = pytest.mark.gpu


def _fb(*a, **k):
    from carmel_amd.trainer import HipForwardBackward
    return HipForwardBackward(*a, **k)


def sig6(x):
    return float("%.6g" % x)


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(n_sym=3, deg=10, n_states=20)), (3, dict(n_sym=2, deg=6, n_states=12, hi=12, p_eps=0.4)),
                                     (4, dict(n_sym=8, deg=5, n_states=60, p_eps=0.1)), (5, dict(n_states=300, deg=12, n_sym=5, n_pairs=1500, hi=14))])
def test_matrix_estep_against_the_oracle_and_the_lattices(oracle, seed, kw):
    w, c = model(seed, **kw)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    ow.normalize(NORM_CONDITIONAL, 0.0)
    mat = oracle.estimate_matrix(ow, oc)
    fb = _fb(w, c)
    lp_l, wlp_l = fb.estimate(per_pair=True)
    pl_l, counts_l = fb.pair_logprob.copy(), fb.counts().copy()
    fb.set_matrix_fb(True)
    lp, wlp = fb.estimate(per_pair=True)
    pl, counts = fb.pair_logprob.copy(), fb.counts().copy()
    # the oracle
    ok = np.isfinite(mat["pair_logprob"])
    assert np.array_equal(np.isfinite(pl), ok)
    np.testing.assert_allclose(pl[ok], mat["pair_logprob"][ok], rtol=1e-12, atol=1e-12)
    assert lp == pytest.approx(mat["sum_logprob"], rel=1e-12) and wlp == pytest.approx(mat["sum_weighted_logprob"], rel=1e-12)
    np.testing.assert_allclose(counts, np.exp(mat["counts_ln"]), rtol=1e-9, atol=1e-300)
    # the lattice sweep of the same trainer
    np.testing.assert_allclose(pl[ok], pl_l[ok], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(counts, counts_l, rtol=1e-9, atol=1e-300)
    # M-step and a second E-step on the matrix path; reproducible bit for bit
    fb.maximize(1.0)
    lp2, _ = fb.estimate()
    c2 = fb.counts().copy()
    lp3, _ = fb.estimate()
    assert lp2 >= lp - 1e-9 * abs(lp) and lp3 == lp2 and np.array_equal(c2, fb.counts())
    # and back to the lattices
    fb.set_matrix_fb(False)
    lp4, _ = fb.estimate()
    assert lp4 == pytest.approx(lp2, rel=1e-12)
    fb.close()


def test_matrix_training_replays_the_recorded_tutorial_run(oracle, golden_dir):
    """carmel -t epron-jpron.data epron-jpron.fst (commands.trace:7-77) with every E-step on the dense matrix"""
    from carmel_amd.trainer import TrainOpts, train
    gold = json.load(open(os.path.join(golden_dir, "trace_expected.json")))["epron-jpron"]
    ow = oracle.OracleWfst.parse(open(os.path.join(golden_dir, "epron-jpron.fst")).read())
    ow.reduce()
    oc = oracle.OracleCorpus.parse(ow, open(os.path.join(golden_dir, "epron-jpron.data")).read())
    a, ca = ow.arrays(), oc.arrays()
    w = Wfst(a["n_states"], a["final"], a["src"], a["dst"], a["isym"], a["osym"], a["logw"], a["group"])
    c = Corpus(ca["in_off"], ca["in_sym"], ca["out_off"], ca["out_sym"], ca["weight"])
    fb = _fb(w, c)
    fb.set_matrix_fb(True)
    best, trace = train(fb, TrainOpts())
    assert len(trace) == len(gold["iters"]) == 5
    for t, g in zip(trace, gold["iters"]):
        assert sig6(t["log2_prob"]) == g["log2_prob"]
        assert sig6(t["log2_ppx_example"]) == g["log2_ppx_example"]
        assert t["new_best"] == g["new_best"]
    for t, g in zip(trace[2:], gold["iters"][2:]):
        assert t["last_change"] == pytest.approx(g["max_dweight"], rel=1e-9)
    ow.set_logw(fb.weights())
    gl, el = ow.write(full=False, onearc=False).strip().split("\n"), gold["final_wfst"].strip().split("\n")
    assert len(gl) == len(el)
    for x, y in zip(gl, el):
        assert NUM.sub("#", x) == NUM.sub("#", y)
        for u, v in zip(NUM.findall(x), NUM.findall(y)):
            assert float(u) == pytest.approx(float(v), rel=1e-9)
    fb.close()


def test_matrix_fb_command_line(golden_dir, tmp_path):
    """$CARMEL --matrix-fb -t ... and --matrix-fb --train-cascade -HJ cipher.data cipher.wfsa cipher.fst: the reference's log
    line (train.cc:382-383) and the recorded iterations (commands.trace:7-19, 6903-6952; first 6 of the cascade's)"""
    g = lambda n: os.path.join(golden_dir, n)
    gold = json.load(open(g("trace_expected.json")))
    rc, out, err = run(["--matrix-fb", "-t", g("epron-jpron.data"), g("epron-jpron.fst")])
    assert rc == 0, err
    assert "Using (input,state,output) full matrix, not derivation lattice.  Usually slower." in err
    its = ITER.findall(err)
    assert len(its) == 5
    for got, gg in zip(its, gold["epron-jpron"]["iters"]):
        assert float(got[1]) == gg["log2_prob"] and float(got[5]) == gg["log2_ppx_example"] and bool(got[6]) == gg["new_best"]
    rc, out, err = run(["--matrix-fb", "--train-cascade", "-HJ", "-M", "6", g("cipher.data"), g("cipher.wfsa"), g("cipher.fst")],
                       env={"CARMEL_TRAINED_DIR": str(tmp_path)})
    assert rc == 0, err
    assert "Using (input,state,output) full matrix" in err
    its = ITER.findall(err)
    assert len(its) == 6
    for got, gg in zip(its, gold["cipher"]["iters"]):
        assert float(got[1]) == gg["log2_prob"] and float(got[5]) == gg["log2_ppx_example"]


def test_matrix_fb_refuses_an_epsilon_cycle():
    from carmel_amd._capi import CarmelHipError
    w = synth.random_wfst(12, 4, n_sym=3, p_eps=0.0, seed=9)
    w.isym[1] = w.osym[1] = 0      # state 0 -> x
    x = int(w.dst[1])
    k = np.flatnonzero(w.src == x)[1]
    w.isym[k] = w.osym[k] = 0
    w.dst[k] = 0                   # x -> state 0: a cycle of *e*:*e* arcs
    c = synth.random_walk_corpus(w, 20, min_arcs=2, max_arcs=6, seed=9, out_degree=4)
    fb = _fb(w, c)
    with pytest.raises(CarmelHipError) as e:
        fb.set_matrix_fb(True)
    assert "cycle" in str(e.value)
    fb.estimate()  # the lattices are still there
    fb.close()
