"""CPU: the oracle's restatement of carmel --matrix-fb (oracle/matrix.hpp; train.cc:698-860) against its restatement of
the derivation lattices (oracle/deriv.hpp, which the recorded tutorial runs pin) and against the recorded trace itself."""
import json
import math
import os

import numpy as np
import pytest

from carmel_amd import synth


def acyclic_eps(w):
    """random_wfst may draw *e*:*e* arcs that close a cycle; the matrix walk needs a topological order of the epsilon
    graph (train.cc:339-357), so *e*:*e* arcs that do not go forward in state number get an output symbol"""
    both = (w.isym == 0) & (w.osym == 0) & (w.dst <= w.src)
    w.osym[both] = synth.FIRST_SYM
    return w


def model(seed, n_states=30, deg=8, n_sym=4, n_pairs=60, p_eps=0.25, lo=2, hi=9):
    w = acyclic_eps(synth.random_wfst(n_states, deg, n_sym=n_sym, p_eps=p_eps, seed=seed))
    c = synth.random_walk_corpus(w, n_pairs, min_arcs=lo, max_arcs=hi, seed=seed, out_degree=deg)
    rng = np.random.default_rng(seed)
    c.weight[:] = rng.uniform(0.5, 3.0, c.n_pairs)
    return w, c


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(n_sym=3, deg=10, n_states=20)), (3, dict(n_sym=2, deg=6, n_states=12, hi=12, p_eps=0.4)),
                                     (4, dict(n_sym=8, deg=5, n_states=60, p_eps=0.1))])
def test_matrix_walk_sums_over_the_lattices_derivations(oracle, seed, kw):
    w, c = model(seed, **kw)
    ow, oc = oracle.OracleWfst.from_arrays(w), oracle.OracleCorpus.from_arrays(c)
    lat = oracle.estimate(ow, oc)
    mat = oracle.estimate_matrix(ow, oc)
    assert mat["eps_back_edges"] == 0
    assert (w.isym == 0).any() and ((w.isym == 0) & (w.osym == 0)).any()
    assert np.array_equal(np.isfinite(mat["pair_logprob"]), lat["has_deriv"])
    ok = lat["has_deriv"]
    np.testing.assert_allclose(mat["pair_logprob"][ok], lat["pair_logprob"][ok], rtol=1e-12, atol=1e-12)
    assert mat["sum_logprob"] == pytest.approx(lat["sum_logprob"], rel=1e-12)
    assert mat["sum_weighted_logprob"] == pytest.approx(lat["sum_weighted_logprob"], rel=1e-12)
    fin = np.isfinite(lat["counts_ln"])
    assert np.array_equal(np.isfinite(mat["counts_ln"]), fin)
    np.testing.assert_allclose(mat["counts_ln"][fin], lat["counts_ln"][fin], rtol=0, atol=1e-10)


def test_matrix_walk_on_the_tutorial_transducer(oracle, golden_dir):
    """carmel -t epron-jpron.data epron-jpron.fst (commands.trace:7-19): the first iteration's recorded corpus probability
    is a sum over derivations at the transducer's initial weights -- the dense walk must land on the same number"""
    gold = json.load(open(os.path.join(golden_dir, "trace_expected.json")))["epron-jpron"]
    ow = oracle.OracleWfst.parse(open(os.path.join(golden_dir, "epron-jpron.fst")).read())
    ow.reduce()
    oc = oracle.OracleCorpus.parse(ow, open(os.path.join(golden_dir, "epron-jpron.data")).read())
    ow.normalize(0, 0.0)  # WFST::train normalises first (train.cc:509), carmel's default method is conditional
    mat = oracle.estimate_matrix(ow, oc)
    assert mat["eps_back_edges"] == 0
    assert float("%.6g" % (mat["sum_weighted_logprob"] / math.log(2))) == gold["iters"][0]["log2_prob"]
    lat = oracle.estimate(ow, oc)
    fin = np.isfinite(lat["counts_ln"])
    np.testing.assert_allclose(mat["counts_ln"][fin], lat["counts_ln"][fin], rtol=0, atol=1e-10)
