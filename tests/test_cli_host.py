"""CPU: the carmel-compatible front end's host side (file reader, writer, composition) against the oracle.
`carmel a b` (no -t) composes and prints on the host; nothing here needs a GPU."""
import os
import subprocess

import pytest

from conftest import ROOT

CLI = os.path.join(ROOT, "carmel_amd", "bin", "carmel")


def run(*args):
    p = subprocess.run([CLI] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("name", ["epron-jpron.fst", "cipher.wfsa", "cipher.fst", "train.a", "train.a.w", "train.a.u",
                                  "wfst3", "wfst3c", "chain.1", "chain.2", "tagging.fsa"])
@pytest.mark.parametrize("flags", ["-HJ", "-J", ""])
def test_parse_write_roundtrip_matches_oracle(oracle, golden_dir, name, flags):
    path = os.path.join(golden_dir, name)
    rc, out, err = run(*([flags] if flags else []), path)
    assert rc == 0, err
    ow = oracle.OracleWfst.parse(open(path).read())
    ow.reduce()
    assert out == ow.write(full="J" in flags, onearc="H" in flags)


@pytest.mark.parametrize("a,b,states,arcs", [("cipher.wfsa", "cipher.fst", 57, 11511),
                                             ("tagging.fsa", "tagging.fst", 46, 400994)])
def test_composition_is_the_references(oracle, golden_dir, a, b, states, arcs):
    """state/arc counts are the reference's (commands.trace:5866, 6903) and the composed transducer is, arc for arc
    and in the same order, the oracle's restatement of compose.cc"""
    pa, pb = os.path.join(golden_dir, a), os.path.join(golden_dir, b)
    rc, out, err = run("-c", "-q", pa, pb)
    assert rc == 0, err
    assert "Number of states in result: %d" % states in out and "Number of arcs in result: %d" % arcs in out
    rc, out, err = run("-HJ", "-q", pa, pb)
    oc = oracle.OracleCascade([open(pa).read(), open(pb).read()], remember=False)  # plain `carmel a b`: no chains
    assert out == oc.composed().write(full=True, onearc=True)


def test_three_way_chain(oracle, golden_dir):
    pa, pb = os.path.join(golden_dir, "chain.1"), os.path.join(golden_dir, "chain.2")
    rc, out, err = run("-HJ", "-q", pa, pb, pb)
    # chain.2 composed twice may be empty; the front end must say so like carmel does (exit -3) or print a result
    assert rc in (0, 253)


def test_bad_file_and_usage():
    rc, out, err = run("/nonexistent/file")
    assert rc != 0 and "could not be opened" in err
    rc, out, err = run()
    assert rc != 0


def test_training_without_gpu_fails_loudly(golden_dir):
    import carmel_amd._capi as capi
    if capi.lib.carmel_hip_device_count() > 0:
        pytest.skip("GPU present")
    rc, out, err = run("-t", os.path.join(golden_dir, "epron-jpron.data"), os.path.join(golden_dir, "epron-jpron.fst"))
    assert rc != 0 and "ERROR" in err and out == ""


def random_fst_text(rng, n_states, n_arcs, in_syms, out_syms, p_eps):
    """a small transducer in carmel's text format: named states, some *e* labels, some locked arcs"""
    names = ["q%d" % i for i in range(n_states)]
    lines = [names[-1]]
    arcs = [(0, int(rng.integers(1, n_states)))] + [(int(rng.integers(0, n_states)), int(rng.integers(0, n_states)))
                                                     for _ in range(n_arcs - 1)]
    arcs += [(s, n_states - 1) for s in range(n_states - 1) if rng.random() < 0.5]  # ways into the final state
    for s, d in arcs:
        i = "*e*" if rng.random() < p_eps else str(rng.choice(in_syms))
        o = "*e*" if rng.random() < p_eps else str(rng.choice(out_syms))
        w = "%.4f" % rng.uniform(0.05, 1.0)
        lock = "!" if rng.random() < 0.15 else ""
        lines.append("(%s (%s %s %s %s%s))" % (names[s], names[d], i, o, w, lock))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("dash_a", [False, True])
@pytest.mark.parametrize("seed", range(40))
def test_composition_of_random_transducers(oracle, tmp_path, seed, dash_a):
    """compose.cc's three-state epsilon filter (and, with -a, its two-state filter with mediate states, :219-313), LIFO
    state discovery, arc prepending and reduction on random inputs: the front end's composition must be the oracle's,
    arc for arc, in the same order (bit-exact arc indexing), groups included (a plain composition keeps the group of a
    copied epsilon arc and gives none to a paired one, cascade.h:566-592)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    mid = ["x", "y", "z"][:int(rng.integers(2, 4))]
    a = random_fst_text(rng, int(rng.integers(2, 7)), int(rng.integers(2, 14)), ["a", "b", "c"], mid, float(rng.uniform(0, 0.5)))
    b = random_fst_text(rng, int(rng.integers(2, 7)), int(rng.integers(2, 14)), mid, ["u", "v"], float(rng.uniform(0, 0.5)))
    pa, pb = str(tmp_path / "a.fst"), str(tmp_path / "b.fst")
    open(pa, "w").write(a)
    open(pb, "w").write(b)
    rc, out, err = run("-HJ", "-q", *((["-a"] if dash_a else []) + [pa, pb]))
    try:
        oc = oracle.OracleCascade([a, b], remember=False, dash_a=dash_a)
    except RuntimeError:
        # both say so: an unusable transducer file (-2, carmel.cc:1213) or an empty composition (-3, carmel.cc:1336-1341)
        assert (rc == 256 - 3 and "Empty or invalid result of composition" in err) or (rc == 256 - 2 and "Bad format" in err)
        return
    assert rc == 0, err
    assert out == oc.composed().write(full=True, onearc=True), (a, b)


def test_load_fem_param(golden_dir, tmp_path, oracle):
    """--load-fem-param=FILE (carmel.cc:790-799; cascade.h:180-202): the members' weights, in arc order, from a file
    such as --fem-param writes; too few weights is an error"""
    import re
    path = os.path.join(golden_dir, "train.a.w")
    n = len(oracle.OracleWfst.parse(open(path).read()).arrays()["logw"])
    ws = [0.5 / (k + 1) for k in range(n)]
    pf = str(tmp_path / "params")
    open(pf, "w").write("".join("%r\n" % w for w in ws))
    rc, out, err = run("-HJ", "-d", "--load-fem-param=" + pf, path)
    assert rc == 0, err
    got = [float(x) for x in re.findall(r" ([0-9.e+-]+)!?\)\)", out)]
    assert got == pytest.approx(ws, rel=1e-12)
    open(pf, "w").write("0.5\n")
    rc, out, err = run("-HJ", "--load-fem-param=" + pf, path)
    assert rc != 0 and "doesn't have enough params" in err


def test_unimplemented_switches_are_refused(golden_dir):
    """a switch or option outside the training path exits -12 with a message instead of being accepted and ignored
    (the verdict of round 1: `-a`, `-B`, `-k n` were swallowed; carmel.cc:1137 is the reference's own -12)"""
    path = os.path.join(golden_dir, "train.a.w")
    for args in (["-k", "3", path], ["-g", "5", path], ["-v", path], ["-N", "0", path],
                 ["--project-left", path], ["-tx", path, path]):
        rc, out, err = run(*args)
        assert rc == 256 - 12, (args, rc, err)
        assert "not implemented" in err and out == ""
    rc, out, err = run("-h")
    assert rc == 0


@pytest.mark.parametrize("flags,suffix", [("-ZB", "log"), ("-Z2", "ln"), ("-Z", None)])
def test_weight_output_bases(golden_dir, oracle, flags, suffix):
    """-B / -2 (carmel.cc:76-101; weight.h:476-486): a weight in log form is written `x log` (base 10) or `x ln` instead
    of e^x; the reader takes all three spellings back (weight.h:503-528) to the same weights"""
    import math
    import re
    path = os.path.join(golden_dir, "train.a.w")
    rc, out, err = run(flags, "-HJ", path)
    assert rc == 0, err
    ws = oracle.OracleWfst.parse(open(path).read())
    ws.reduce()
    want = ws.arrays()["logw"]
    toks = re.findall(r" (\S+)\)\)", out)
    assert len(toks) == len(want)
    for tok, lw in zip(toks, want):
        if suffix == "log":
            assert tok.endswith("log") and float(tok[:-3]) * math.log(10) == pytest.approx(lw, abs=1e-12)
        elif suffix == "ln":
            assert tok.endswith("ln") and float(tok[:-2]) == pytest.approx(lw, abs=1e-12)
        else:
            assert tok.startswith("e^") and float(tok[2:]) == pytest.approx(lw, abs=1e-12)
    back = oracle.OracleWfst.parse(out)  # the oracle's reader parses what the front end wrote
    assert back.arrays()["logw"] == pytest.approx(want, abs=1e-12)


@pytest.mark.parametrize("seed", range(12))
def test_normby_normalises_the_inputs_before_composing(oracle, tmp_path, seed):
    """fem_in -> fem_normby (carmel.cc:778-808): with --normby every INPUT transducer is normalised -- priors added, locked
    arcs reserving their weight, tied groups sharing one -- before anything is composed; --write-loaded=suffix writes them,
    --number-from=N first gives every arc its own tie group.  The host-side normalisation (the same fst.cc:86-244 the GPU
    M-step implements) against the oracle's, on random transducers with locked and tied arcs."""
    import re
    import numpy as np
    rng = np.random.default_rng(500 + seed)
    a = random_fst_text(rng, int(rng.integers(2, 7)), int(rng.integers(3, 16)), ["a", "b", "c"], ["x", "y"], float(rng.uniform(0, 0.3)))
    b = random_fst_text(rng, int(rng.integers(2, 7)), int(rng.integers(3, 16)), ["x", "y"], ["u", "v"], float(rng.uniform(0, 0.3)))
    # tie some unlocked arcs of b into groups 7 and 8
    lines = b.split("\n")
    for k in range(1, len(lines)):
        if lines[k].endswith("))") and not lines[k].endswith("!))") and rng.random() < 0.3:
            lines[k] = lines[k][:-2] + "!%d))" % int(rng.integers(7, 9))
    b = "\n".join(lines)
    pa, pb = str(tmp_path / "a.fst"), str(tmp_path / "b.fst")
    open(pa, "w").write(a)
    open(pb, "w").write(b)
    normby = "".join(rng.choice(["J", "C", "N"], 2))
    priors = [float(rng.choice([0.0, 0.25, 1.5])), float(rng.choice([0.0, 0.1]))]
    env = dict(os.environ, CARMEL_TRAINED_DIR=str(tmp_path))
    p = subprocess.run([CLI, "-HJm", "-q", "--normby=" + normby, "--priors=%r,%r" % tuple(priors), "--write-loaded=loaded", pa, pb],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env)
    if p.returncode == 256 - 2:  # a random file whose final state has no arc: both readers refuse it
        with pytest.raises((RuntimeError, ValueError)):
            oracle.OracleWfst.parse(a), oracle.OracleWfst.parse(b)
        return
    assert p.returncode in (0, 256 - 3), p.stderr   # (the composition itself may be empty)
    assert "Normalizing input transducers by --normby=" + normby in p.stderr
    num = re.compile(r"(?<![A-Za-z_*])-?\d+\.?\d*(?:e[+-]?\d+)?(?![A-Za-z_])")
    for path, text, ch, pr in ((pa, a, normby[0], priors[0]), (pb, b, normby[1], priors[1])):
        ow = oracle.OracleWfst.parse(text)
        if ch != "N":
            ow.normalize(1 if ch == "J" else 0, pr)
        want = ow.write(full=True, onearc=True)
        got = open(os.path.join(str(tmp_path), os.path.basename(path) + ".loaded")).read()
        gl, wl = got.strip().split("\n"), want.strip().split("\n")
        assert len(gl) == len(wl)
        for x, y in zip(gl, wl):
            assert num.sub("#", x) == num.sub("#", y), (x, y)
            for u, v in zip(num.findall(x), num.findall(y)):
                assert float(u) == pytest.approx(float(v), rel=1e-9, abs=1e-300)
    # --number-from: consecutive group ids over the inputs, in file order
    p = subprocess.run([CLI, "-HJm", "-q", "--number-from=5", "--write-loaded=num", pa, pb], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, env=env)
    assert "Assigning unique group ids to each arc in input cascade starting at 5" in p.stderr
    ids = []
    for path in (pa, pb):
        ids += [int(x) for x in re.findall(r"!(\d+)\)\)", open(os.path.join(str(tmp_path), os.path.basename(path) + ".num")).read())]
    assert ids == list(range(5, 5 + len(ids))) and len(ids) > 0


def test_fem_early_param_and_inert_gibbs_options(golden_dir, tmp_path):
    """--fem-early-param=FILE: the input transducers' weights as loaded (after --load-fem-param / --normby), one per arc in
    the members' arc order (carmel.cc:801, 810-817 -- where the reference, by a slip, writes to --fem-param's file);
    --sample-prob / --no-prob / --cache-prob are accepted and change nothing, as in carmel (gibbs_opts.hpp:240, 255-258)"""
    pa, pb = os.path.join(golden_dir, "cipher.wfsa"), os.path.join(golden_dir, "cipher.fst")
    out = str(tmp_path / "early")
    rc, so, err = run("-q", "-c", "--fem-early-param=" + out, "--sample-prob", "--no-prob", "--cache-prob", pa, pb)
    assert rc == 0, err
    got = [float(x[2:]) if x.startswith("e^") else float(x) for x in open(out).read().split()]
    rc, full, _ = run("-HJ", pa)
    rc, full2, _ = run("-HJ", pb)
    assert len(got) == full.count("(") // 2 + full2.count("(") // 2 - 0  # one weight per arc of both inputs
