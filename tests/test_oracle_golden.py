"""CPU: the oracle (CPU restatement of the reference) against the reference's own recorded runs
(carmel/carmel-tutorial/commands.trace, parsed into tests/golden/trace_expected.json by make_golden.py)."""
import json
import math
import os
import re

import numpy as np
import pytest


def sig6(x):
    """the reference prints these with 6 significant digits (default ostream precision)"""
    return float("%.6g" % x)


def _gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "trace_expected.json")))


def _rd(golden_dir, name):
    return open(os.path.join(golden_dir, name)).read()


def test_epron_jpron_trace_and_final_weights(oracle, golden_dir):
    gold = _gold(golden_dir)["epron-jpron"]
    ow = oracle.OracleWfst.parse(_rd(golden_dir, "epron-jpron.fst"))
    ow.reduce()
    assert ow.dims()[:2] == (57, 154)  # commands.trace:1-2
    oc = oracle.OracleCorpus.parse(ow, _rd(golden_dir, "epron-jpron.data"))
    best, rows = oracle.train(ow, oc)
    assert len(rows) == 5
    for r, g in zip(rows, gold["iters"]):
        assert sig6(r["log2_prob"]) == g["log2_prob"]
        assert sig6(r["log2_ppx_symbol"]) == g["log2_ppx_symbol"]
        assert sig6(r["log2_ppx_example"]) == g["log2_ppx_example"]
        assert bool(r["new_best"]) == g["new_best"]
        assert int(r["n_example"]) == g["n_example"]
    ratios = [g["rel_ppx_ratio"] for g in gold["iters"][1:]]
    for r, g in zip(rows[1:], ratios):
        gl = float(g[2:]) if g.startswith("e^") else math.log(float(g))
        assert r["rel_ppx_ratio_ln"] == pytest.approx(gl, rel=1e-9)
    for r, g in zip(rows[2:], gold["iters"][2:]):
        assert r["last_change"] == pytest.approx(g["max_dweight"], rel=1e-12)
    got = ow.write().strip().split("\n")
    exp = gold["final_wfst"].strip().split("\n")
    num = re.compile(r"(?<![\w\"])(\d+\.\d+(?:e[-+]\d+)?|\d+e[-+]\d+)(?![\w\"])")
    assert len(got) == len(exp)
    for x, y in zip(got, exp):
        assert num.sub("#", x) == num.sub("#", y)
        for u, v in zip(num.findall(x), num.findall(y)):
            assert float(u) == pytest.approx(float(v), rel=1e-13)


def test_cipher_cascade_trace_and_trained_channel(oracle, golden_dir):
    gold = _gold(golden_dir)["cipher"]
    rows, texts, dims = oracle.train_cascade_text(
        [_rd(golden_dir, "cipher.wfsa"), _rd(golden_dir, "cipher.fst")], _rd(golden_dir, "cipher.data"))
    assert dims == (gold["composed"]["states"], gold["composed"]["arcs"])
    assert len(rows) == len(gold["iters"]) == 22
    for r, g in zip(rows, gold["iters"]):
        assert sig6(r["log2_prob"]) == g["log2_prob"]
        assert sig6(r["log2_ppx_example"]) == g["log2_ppx_example"]
        assert bool(r["new_best"]) == g["new_best"]
    for r, g in zip(rows[1:], gold["iters"][1:]):
        assert r["rel_ppx_ratio_ln"] == pytest.approx(math.log(float(g["rel_ppx_ratio"])), rel=1e-9)

    def weights(txt):
        d = {}
        for s, t, i, o, w in re.findall(r'\((\S+) \((\S+) (\S+) (\S+) ([^()! ]+)!?\)\)', txt):
            d[(s, t, i, o)] = math.exp(float(w[2:])) if w.startswith("e^") else float(w)
        return d
    for k, name in enumerate(["cipher.wfsa.trained", "cipher.fst.trained"]):
        a, b = weights(texts[k]), weights(_rd(golden_dir, name))
        assert set(a) == set(b) and len(a) > 500
        for key in b:
            assert a[key] == pytest.approx(b[key], rel=1e-9, abs=1e-300)


@pytest.mark.timeout(900)
def test_tagging_cascade_trace(oracle, golden_dir):
    gold = _gold(golden_dir)["tagging"]
    rows, _, dims = oracle.train_cascade_text(
        [_rd(golden_dir, "tagging.fsa"), _rd(golden_dir, "tagging.fst")], _rd(golden_dir, "tagging.data"),
        max_iter=3)
    assert dims == (gold["composed"]["states"], gold["composed"]["arcs"])  # 46 states / 400994 arcs
    for r, g in zip(rows, gold["iters"][:3]):
        assert sig6(r["log2_prob"]) == g["log2_prob"]
        assert int(r["n_example"]) == g["n_example"]


def test_logweight_identities(oracle):
    # graehl/shared/weight.h:938-947: "1", "e^0", "0ln", "0log" all parse to the same weight
    texts = ["0\n(0 (0 a b %s))" % s for s in ("1", "e^0", "0ln", "0log")]
    ws = [oracle.OracleWfst.parse(t, always_named=False).arrays()["logw"][0] for t in texts]
    assert ws == [0.0, 0.0, 0.0, 0.0]
    # locked and tied suffixes, *e* defaults, comment lines (carmel/doc/FORMATS)
    w = oracle.OracleWfst.parse("%% comment\n2\n(0 (1 \"a\" 0.5!) (1 *e* \"b\" 0.25!7) (2))\n(1 (2 x))\n", False)
    a = w.arrays()
    assert a["group"].tolist() == [0, 7, 0xFFFFFFFF, 0xFFFFFFFF]
    assert a["isym"][2] == 0 and a["osym"][2] == 0 and a["logw"][2] == 0.0


def test_crp_tagging_bookkeeping_against_the_reference_output(oracle, golden_dir):
    """carmel --crp -M 6000 tagging.data tagging.fsa tagging.fst (commands:33) wrote tagging.{fsa,fst}.trained: the only
    reference-held OUTPUT of the sampler (tests/crp_pin.py says what in it does not depend on the random stream).  The
    oracle's sampler -- min_prior 0.01 (gibbs.cc:390-397), prior = alpha * p0 * |group| (gibbs.hpp:589-592), time-averaged
    counts (gibbs.hpp:626-638, delta_sum.hpp:49-106), probs_to_cascade (gibbs.cc:66-76) -- is run twice with its own
    uniforms; where its two runs agree it must have written what the reference wrote, to 1e-11."""
    import crp_pin
    g = lambda n: open(os.path.join(golden_dir, n)).read()
    runs = []
    for seed, sweeps in ((3, 60), (4, 90)):
        oc = oracle.OracleCascade([g("tagging.fsa"), g("tagging.fst")])  # (a run leaves its result in the cascade: p0 of the next)
        corpus = oc.corpus(g("tagging.data"))
        r = oracle.gibbs_run(oc, corpus, seed, normby="CC", priors=[0.0, 0.0], iters=sweeps, burnin=0)
        runs.append((oc.write_member(0, r["param_logw"]), oc.write_member(1, r["param_logw"])))
    res = crp_pin.check((g("tagging.fsa.crp-trained"), g("tagging.fst.crp-trained")), runs[1], runs[0], min_corr=0.8,
                        inputs=(g("tagging.fsa"), g("tagging.fst"), g("tagging.data")))
    # beyond the closed form only what neither chain ever chose coincides (a first word that may be JJS once in 10^5 sweeps)
    assert res["fsa_closed_form"] <= res["fsa_equal_to_reference"] <= res["fsa_closed_form"] + 80
    print(res)
