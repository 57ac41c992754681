"""GPU: composition with the product construction on the device (carmel --gpu-compose -> carmel_hip_compose,
csrc/compose.hip) against the host composer and the oracle: the composed transducer must be the same arc for arc, in the
same order, with the same state numbers (compose.cc's LIFO discovery order) -- and, under --train-cascade, the same
chains, i.e. the same training run."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "carmel_amd", "bin", "carmel")


def run(args, env=None):
    p = subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                       env=dict(os.environ, **(env or {})))
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("seed", range(40))
@pytest.mark.parametrize("threshold", ["32", "2"])
def test_random_transducers(oracle, tmp_path, seed, threshold):
    """both of compose.cc's code paths: -T 32 (small states: nested loops in list order) and -T 2 (the larger state is
    indexed: matches newest first)"""
    from test_cli_host import random_fst_text
    rng = np.random.default_rng(seed)
    mid = ["x", "y", "z"][:int(rng.integers(2, 4))]
    a = random_fst_text(rng, int(rng.integers(2, 7)), int(rng.integers(2, 14)), ["a", "b", "c"], mid, float(rng.uniform(0, 0.5)))
    b = random_fst_text(rng, int(rng.integers(2, 7)), int(rng.integers(2, 14)), mid, ["u", "v"], float(rng.uniform(0, 0.5)))
    pa, pb = str(tmp_path / "a.fst"), str(tmp_path / "b.fst")
    open(pa, "w").write(a)
    open(pb, "w").write(b)
    rc_h, out_h, err_h = run(["-HJ", "-q", "-T", threshold, pa, pb])
    rc_d, out_d, err_d = run(["-HJ", "-q", "-T", threshold, "--gpu-compose", pa, pb])
    assert rc_d == rc_h, err_d
    assert out_d == out_h
    if threshold == "32" and rc_h == 0:
        oc = oracle.OracleCascade([a, b], remember=False)
        assert out_d == oc.composed().write(full=True, onearc=True)


@pytest.mark.parametrize("a,b,states,arcs", [("cipher.wfsa", "cipher.fst", 57, 11511), ("tagging.fsa", "tagging.fst", 46, 400994),
                                             ("chain.1", "chain.2", None, None)])
def test_tutorial_cascades(golden_dir, a, b, states, arcs):
    pa, pb = os.path.join(golden_dir, a), os.path.join(golden_dir, b)
    rc, out_h, err = run(["-HJ", "-q", pa, pb])
    assert rc == 0, err
    rc, out_d, err = run(["-HJ", "--gpu-compose", pa, pb], env={"CARMEL_TIMING": "1"})
    assert rc == 0, err
    assert out_d == out_h
    assert "timing: composition on the GPU" in err
    if states:
        assert "(%d states / %d arcs)" % (states, arcs) in err  # commands.trace:5866, 6903


def test_three_members_and_training(golden_dir, tmp_path):
    """a three-member cascade (the second composition's left operand already carries chains) trained with and without
    the device composition: the same iterations, the same trained members"""
    g = lambda n: os.path.join(golden_dir, n)
    ident = tmp_path / "ident.fst"
    syms = sorted(set(re.findall(r'"([^"]+)"\)\)', open(g("cipher.fst")).read())))  # the channel's output alphabet
    ident.write_text("0\n" + "".join('(0 (0 "%s" "%s" 0.9))\n(0 (0 "%s" "%s" 0.1))\n' % (c, c, d, c)
                                      for c, d in zip(syms, syms[1:] + syms[:1])))  # a noisy copy: every symbol is mostly itself
    outs = []
    for extra in ([], ["--gpu-compose"]):
        d = tmp_path / ("o%d" % len(outs))
        d.mkdir()
        rc, out, err = run(["--train-cascade", "-HJ", "-M", "4"] + extra + [g("cipher.data"), g("cipher.wfsa"), g("cipher.fst"), str(ident)],
                           env={"CARMEL_TRAINED_DIR": str(d)})
        assert rc == 0, err
        its = [l for l in err.split("\n") if l.startswith("i=")]
        outs.append((its, "".join(open(str(d / f)).read() for f in sorted(os.listdir(str(d))))))
    assert len(outs[0][0]) == 4 and outs[0][0] == outs[1][0]
    # the trained members: equal up to the order in which the atomic adds of chain_scatter land (last digits)
    num = re.compile(r"(?<![\w\"])(\d+\.\d+(?:e[-+]\d+)?|\d+e[-+]\d+)(?![\w\"])")
    la, lb = outs[0][1].split("\n"), outs[1][1].split("\n")
    assert len(la) == len(lb) > 1000
    for x, y in zip(la, lb):
        assert num.sub("#", x) == num.sub("#", y)
        for u, v in zip(num.findall(x), num.findall(y)):
            assert float(u) == pytest.approx(float(v), rel=1e-9, abs=1e-300)
