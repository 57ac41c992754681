"""GPU: the carmel-compatible command line end to end on the reference's tutorial commands
(carmel/carmel-tutorial/commands), compared with the recorded trace and the committed *.trained files."""
import json
import math
import os
import re
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "carmel_amd", "bin", "carmel")
NUM = re.compile(r"(?<![\w\"])(\d+\.\d+(?:e[-+]\d+)?|\d+e[-+]\d+)(?![\w\"])")
ITER = re.compile(r"i=(\d+) \(rate=1\): probability=2\^(\S+) per-symbol-perplexity\(N=(\d+)\)=2\^(\S+) "
                  r"per-example-perplexity\(N=(\d+)\)=2\^(\S+)( \(new best\))?")


def run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=e)
    return p.returncode, p.stdout, p.stderr


def test_epron_jpron_command(golden_dir):
    """$CARMEL -t epron-jpron.data epron-jpron.fst   (commands:10; trace lines 7-77)"""
    gold = json.load(open(os.path.join(golden_dir, "trace_expected.json")))["epron-jpron"]
    rc, out, err = run(["-t", os.path.join(golden_dir, "epron-jpron.data"), os.path.join(golden_dir, "epron-jpron.fst")])
    assert rc == 0, err
    its = ITER.findall(err)
    assert len(its) == 5
    for got, g in zip(its, gold["iters"]):
        assert float(got[1]) == g["log2_prob"] and int(got[2]) == g["n_symbol"] and float(got[3]) == g["log2_ppx_symbol"]
        assert int(got[4]) == g["n_example"] and float(got[5]) == g["log2_ppx_example"] and bool(got[6]) == g["new_best"]
    assert "Converged - maximum weight change less than 0.0001 after 5 iterations." in err
    gl, el = out.strip().split("\n"), gold["final_wfst"].strip().split("\n")
    assert len(gl) == len(el)
    for x, y in zip(gl, el):
        assert NUM.sub("#", x) == NUM.sub("#", y)
        for u, v in zip(NUM.findall(x), NUM.findall(y)):
            assert float(u) == pytest.approx(float(v), rel=1e-9)


def test_cipher_cascade_command(golden_dir, tmp_path):
    """$CARMEL --train-cascade -HJ cipher.data cipher.wfsa cipher.fst   (commands:24; trace lines 6903-6952)"""
    gold = json.load(open(os.path.join(golden_dir, "trace_expected.json")))["cipher"]
    g = lambda n: os.path.join(golden_dir, n)
    rc, out, err = run(["--train-cascade", "-HJ", g("cipher.data"), g("cipher.wfsa"), g("cipher.fst")],
                       env={"CARMEL_TRAINED_DIR": str(tmp_path)})
    assert rc == 0, err
    assert "(57 states / 11511 arcs)" in err
    its = ITER.findall(err)
    assert len(its) == 22
    for got, gg in zip(its, gold["iters"]):
        assert float(got[1]) == gg["log2_prob"] and float(got[5]) == gg["log2_ppx_example"] and bool(got[6]) == gg["new_best"]
    assert "Converged - per-example perplexity ratio exceeds 0.999 after 22 iterations." in err

    def weights(txt):
        d = {}
        for s, t, i, o, x in re.findall(r'\((\S+) \((\S+) (\S+) (\S+) ([^()! ]+)!?\)\)', txt):
            d[(s, t, i, o)] = math.exp(float(x[2:])) if x.startswith("e^") else float(x)
        return d
    for name in ("cipher.wfsa", "cipher.fst"):
        got = weights(open(os.path.join(str(tmp_path), name + ".trained")).read())
        exp = weights(open(g(name + ".trained")).read())
        assert set(got) == set(exp) and len(exp) > 500
        for k in exp:
            assert got[k] == pytest.approx(exp[k], rel=1e-7, abs=1e-300)


def test_flags_joint_and_max_iter(golden_dir, oracle):
    """-j (joint normalisation) -M 3 against the oracle's run of the same command"""
    g = lambda n: os.path.join(golden_dir, n)
    rc, out, err = run(["-t", "-j", "-M", "3", g("epron-jpron.data"), g("epron-jpron.fst")])
    assert rc == 0, err
    ow = oracle.OracleWfst.parse(open(g("epron-jpron.fst")).read())
    ow.reduce()
    oc = oracle.OracleCorpus.parse(ow, open(g("epron-jpron.data")).read())
    best, rows = oracle.train(ow, oc, norm_group=1, max_iter=3)
    its = ITER.findall(err)
    assert len(its) == len(rows)
    for got, r in zip(its, rows):
        assert float(got[1]) == float("%.6g" % r["log2_prob"])
    exp = ow.write().strip().split("\n")
    gl = out.strip().split("\n")
    assert len(gl) == len(exp)
    for x, y in zip(gl, exp):
        assert NUM.sub("#", x) == NUM.sub("#", y)
        for u, v in zip(NUM.findall(x), NUM.findall(y)):
            assert float(u) == pytest.approx(float(v), rel=1e-7)


def test_crp_command(golden_dir, tmp_path, oracle):
    """carmel --crp -M 30 --burnin=10 --priors=0.5,0.1 cipher.data cipher.wfsa cipher.fst: the front end drives the
    exact sampler; its *.trained files carry the time-averaged probabilities the oracle computes from the same uniforms"""
    import ctypes as C
    import numpy as np
    from carmel_amd._capi import lib
    g = lambda n: os.path.join(golden_dir, n)
    rc, out, err = run(["--crp", "-M", "30", "--burnin=10", "--priors=0.5,0.1", "-R", "5", "-HJ", g("cipher.data"),
                        g("cipher.wfsa"), g("cipher.fst")], env={"CARMEL_TRAINED_DIR": str(tmp_path)})
    assert rc == 0, err
    lines = [l for l in err.split("\n") if l.startswith("Gibbs i=")]
    assert len(lines) == 31 and "cache-model prob=2^" in lines[0] and "per-block-ppx(N=10)" in lines[0]
    oc = oracle.OracleCascade([open(g("cipher.wfsa")).read(), open(g("cipher.fst")).read()])
    ref = oracle.gibbs_run(oc, oc.corpus(open(g("cipher.data")).read()),
                           lambda i, b, s: lib.carmel_hip_gibbs_uniform(5, i, b, s), normby="CC", priors=[0.5, 0.1],
                           iters=30, burnin=10)
    logged = [float(re.search(r"prob=2\^(\S+)", l).group(1)) for l in lines]
    for a, b in zip(logged, ref["iter_logprob"] / math.log(2)):
        assert a == float("%.6g" % b)
    exp_txt = oc.write_member(1, ref["param_logw"])
    got_txt = open(os.path.join(str(tmp_path), "cipher.fst.trained")).read()
    gl, el = got_txt.strip().split("\n"), exp_txt.strip().split("\n")
    assert len(gl) == len(el)
    for x, y in zip(gl, el):
        assert NUM.sub("#", x) == NUM.sub("#", y)
        for u, v in zip(NUM.findall(x), NUM.findall(y)):
            assert float(u) == pytest.approx(float(v), rel=1e-9)
