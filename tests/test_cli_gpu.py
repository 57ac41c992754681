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
