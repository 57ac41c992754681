"""GPU: the forest-em command line (carmel_amd/bin/forest-em) on the reference's sample inputs and on synthetic forests,
against the oracle's EM over the same files; plus the host-side text reader on CPU (test_forest_text_cpu below is not
GPU-marked)."""
import math
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CLI = os.path.join(ROOT, "carmel_amd", "bin", "forest-em")


def run(args):
    p = subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    return p.returncode, p.stdout, p.stderr


def parse_vec(txt):
    toks = txt.strip().strip("()").split()
    out = []
    for t in toks:
        out.append(math.exp(float(t[2:])) if t.startswith("e^") else float(t))
    return np.array(out)


def oracle_em(oracle, ftxt, ntxt, max_iter, rel_eps=1.0 / 65536, delta_eps=1.0 / 65536, init=None, restarts=0, seed=1,
              ones=False, random_set=False, keep=None):
    """em.hpp:107-216 at rate 1, driven from Python over the oracle's estimate / maximize; random restarts
    (FForests::randomize -> NormalizeGroups::init_random) draw from the library's counter-based generator"""
    of = oracle.OracleForests(ftxt, ntxt)
    if init is not None:
        of.set_weights(init)
    else:
        of.init_rule_weights(ones=ones)  # forest-em.hpp:297-318: uniform per norm group unless -u
    if random_set:
        from carmel_amd._capi import lib
        of.randomize([1.0 - lib.carmel_hip_gibbs_uniform(seed, 0, r, 0) for r in range(of.n_rules)])
    best, best_w, very_first, trace = -np.inf, of.weights(), True, []
    for restart in range(restarts + 1):
        last, first = -np.inf, True
        for _ in range(max_iter):
            alp = of.estimate()[0]
            trace.append(alp)
            if alp > best or very_first:
                best, best_w = alp, of.weights()
            very_first = False
            rel = np.inf if first else (alp - last) / max(abs(last), 1e-5)
            first = False
            if rel < rel_eps:
                break
            if of.maximize() <= delta_eps:
                break
            last = alp
        if restart < restarts:
            from carmel_amd._capi import lib
            of.randomize([1.0 - lib.carmel_hip_gibbs_uniform(seed, restart + 1, r, 0) for r in range(of.n_rules)])
    # FForests::restore_best (forest-em.hpp:660-671) acts only when restarts were asked for (save_best_enable, :363):
    # otherwise the parameters stay as the last M-step left them
    if restarts > 0:
        of.set_weights(best_w)
    else:
        best_w = of.weights()
    if keep is not None:
        keep.append(of)
    return best, best_w, trace


@pytest.mark.gpu
def test_forest_em_cli_on_reference_sample(oracle, golden_dir, tmp_path):
    """forest-em -f sample/forests -n sample/norm (forest-em/Makefile:62-67): parameters equal the oracle's EM"""
    f, n = os.path.join(golden_dir, "fem.forests"), os.path.join(golden_dir, "fem.norm")
    out = tmp_path / "params"
    rc, so, err = run(["-f", f, "-n", n, "-o", str(out), "-i", "30", "-d", "1.52587890625e-05"])
    assert rc == 0, err
    assert "5 forests, 60 nodes" in err
    best, bw, trace = oracle_em(oracle, open(f).read(), open(n).read(), 30)
    got = parse_vec(out.read_text())
    assert len(got) == len(bw) - 1
    np.testing.assert_allclose(got, np.exp(bw[1:]), rtol=1e-9, atol=1e-300)
    its = re.findall(r"^i=(\d+) average log-prob=(\S+)", err, re.M)
    assert len(its) == len(trace)
    for (i, v), t in zip(its, trace):
        assert float(v) == pytest.approx(t, rel=1e-5)


@pytest.mark.gpu
def test_forest_em_cli_random_restarts(oracle, golden_dir, tmp_path):
    """forest-em -r 2 (em.hpp:199-206): two more starts from random parameters, the best start's parameters win"""
    f, n = os.path.join(golden_dir, "fem.forests"), os.path.join(golden_dir, "fem.norm")
    out = tmp_path / "params"
    rc, so, err = run(["-f", f, "-n", n, "-o", str(out), "-i", "12", "-r", "2", "--random-seed=5", "-d", "1.52587890625e-05"])
    assert rc == 0, err
    assert err.count("Random restart") == 2
    best, bw, trace = oracle_em(oracle, open(f).read(), open(n).read(), 12, restarts=2, seed=5)
    its = re.findall(r"^i=(\d+) average log-prob=(\S+)", err, re.M)
    assert len(its) == len(trace)
    for (i, v), t in zip(its, trace):
        assert float(v) == pytest.approx(t, rel=1e-5)
    np.testing.assert_allclose(parse_vec(out.read_text()), np.exp(bw[1:]), rtol=1e-9, atol=1e-300)


@pytest.mark.gpu
def test_forest_em_cli_checkpoints_on_watch_iterations(oracle, golden_dir, tmp_path):
    """-x PREFIX -c -W 2 (forest-em-params.hpp:138-145; FForests::maximize's tail forest-em.hpp:638-653, dump_params :172-189):
    after the first W + 1 M-steps of a (re)start and every W-th after, the parameters and the counts they were normalised from
    go to PREFIX.{params,counts}.restart.R.iteration.I; -X / -Y report how many counts exceed the thresholds.  Every dump is the
    oracle's state after that M-step; a random restart starts a new series"""
    f, n = os.path.join(golden_dir, "fem.forests"), os.path.join(golden_dir, "fem.norm")
    pre = str(tmp_path / "ck")
    rc, so, err = run(["-f", f, "-n", n, "-o", str(tmp_path / "o"), "-i", "7", "-r", "1", "--random-seed=5", "-e", "-1", "-d", "-1",
                       "-x", pre, "-c", "-W", "2", "-X", "0.5", "-Y", "2"])
    assert rc == 0, err
    from carmel_amd._capi import lib
    of = oracle.OracleForests(open(f).read(), open(n).read())
    of.init_rule_weights()
    reports = re.findall(r"\(out of (\d+) parameters, (\d+) had count > (\S+), and (\d+) had prob > (\S+)\)", err)
    k = 0
    for restart in (1, 2):
        for m in range(7):
            cnt = np.exp(of.estimate()[1])  # (the oracle's counts are logarithms)
            of.maximize()
            watch = m <= 2 or m % 2 == 0
            for kind in ("params", "counts"):
                path = "%s.%s.restart.%d.iteration.%d" % (pre, kind, restart, m + 1)
                assert os.path.exists(path) == watch, path
            if watch:
                tag = ".restart.%d.iteration.%d" % (restart, m + 1)
                np.testing.assert_allclose(parse_vec(open(pre + ".params" + tag).read()), np.exp(of.weights()[1:]), rtol=1e-9, atol=1e-300)
                np.testing.assert_allclose(parse_vec(open(pre + ".counts" + tag).read()), cnt[1:], rtol=1e-9, atol=1e-300)
                total, n_count, _, n_prob, _ = reports[k]
                assert int(total) == of.n_rules - 1
                assert int(n_count) == int((cnt[1:] >= 0.5).sum()) and int(n_prob) == int((cnt[1:] >= 2).sum())
                k += 1
        of.randomize([1.0 - lib.carmel_hip_gibbs_uniform(5, restart, r, 0) for r in range(of.n_rules)])
    assert k == len(reports) and err.count("Writing trained parameters to " + pre) == k
    # -V 2 / -Z 3: on the same iterations the Viterbi derivation of every second forest (the lines -v writes for those forests under
    # the parameters of that E-step) and the empty per-forest counts of every third
    rc, so, err = run(["-f", f, "-n", n, "-i", "3", "-e", "-1", "-d", "-1", "-x", pre + "v", "-V", "2", "-Z", "3", "-W", "1"])
    assert rc == 0, err
    of = oracle.OracleForests(open(f).read(), open(n).read())
    of.init_rule_weights()
    for m in range(3):
        vit = open("%sv.viterbi.restart.1.iteration.%d" % (pre, m + 1)).read().split("\n")[:-1]
        assert len(vit) == of.n_forests // 2
        num = re.compile(r"^(\S+)/(\S+)=(\S+)% (.*)$")
        for k, line in enumerate(vit):
            a, b = num.match(line).groups(), num.match(of.viterbi_line(2 * k + 1)[0].rstrip("\n")).groups()
            assert a[3] == b[3]  # the tree
            for u, v in zip(a[:3], b[:3]):
                assert parse_vec(u)[0] == pytest.approx(parse_vec(v)[0], rel=1e-9)
        assert open("%sv.per_forest_counts.restart.1.iteration.%d" % (pre, m + 1)).read() == "()\n" * (of.n_forests // 3)
        of.estimate()
        of.maximize()
    # without a prefix -c writes nothing (forest-em-params.cpp:43-47)
    rc, so, err = run(["-f", f, "-n", n, "-i", "3", "-c"])
    assert rc == 0 and "Writing trained parameters" not in err


def _fmt_weight(lw):
    """weight.h:468-489 at the stream's defaults: the number while it fits a double comfortably, e^x otherwise; 15 digits"""
    if lw == -np.inf:
        return "0"
    return "%.15g" % math.exp(lw) if -82.0 < lw < 82.0 else "e^%.15g" % lw


@pytest.mark.gpu
@pytest.mark.parametrize("names", [False, True])
def test_forest_em_cli_watch_rule(oracle, golden_dir, tmp_path, names):
    """-w RULE -D DEPTH [-R names] (forest-em-params.hpp:134-137, 150-151; FForests::watch_report, forest-em.hpp:583-616): on watch
    iterations, after the M-step, the top DEPTH rules by weight of the normalisation group that holds RULE -- or that their order
    has not changed.  Restated here over the oracle's weights after every M-step: the same ranking (std::partial_sort's, ties
    aside: the weights of a rank are compared), the same lines (weight at 15 digits, padded to column 15, the rule's description
    or -- without -R -- its number less one, FileLines::getline of no file), the same 'no change' notes; a rule in no group and
    a names file that is too short are refused as the reference refuses them"""
    f, n = os.path.join(golden_dir, "fem.forests"), os.path.join(golden_dir, "fem.norm")
    of = oracle.OracleForests(open(f).read(), open(n).read())
    of.init_rule_weights()
    groups = [[int(x) for x in g.split()] for g in re.findall(r"\(([^()]*)\)", open(n).read()) if g.strip()]
    group = max(groups, key=len)
    rule, depth = group[-1], 3
    args = ["-f", f, "-n", n, "-i", "6", "-e", "-1", "-d", "-1", "-W", "2", "-w", str(rule), "-D", str(depth)]
    label = lambda r: str(r - 1)
    if names:
        rf = tmp_path / "rules"
        rf.write_text("".join("rule number %d -> x\n" % (r + 1) for r in range(of.n_rules - 1)))
        args += ["-R", str(rf)]
        label = lambda r: "rule number %d -> x" % r
    rc, so, err = run(args)
    assert rc == 0, err
    lines = err.split("\n")
    members = list(group)
    d = min(depth, len(members))
    pos = 0
    n_reports = n_same = 0
    for m in range(6):
        of.estimate()
        of.maximize()
        if not (m <= 2 or m % 2 == 0):
            continue
        w = of.weights()
        # (an unchanged ranking's note ends without a newline, as the reference's does: the next iteration's line follows it)
        it = [k for k in range(pos, len(lines)) if re.search(r"(^|\))i=%d " % (m + 1), lines[k])][0]
        pos = it + 1
        top = sorted(members, key=lambda r: -w[r])[:d]
        if all(w[members[k]] >= w[members[k + 1]] for k in range(d - 1)):
            assert lines[it + 1].startswith(" (no change in rank order of top %d rules)" % d), lines[it:it + 3]
            n_same += 1
            continue
        assert lines[it + 1] == "" and lines[it + 2] == "New top %d rules for normalization group:" % d, lines[it:it + 4]
        got = lines[it + 3:it + 3 + d]
        for k, line in enumerate(got):
            mt = re.match(r"^(\S+) +(.*) \(id = (\d+)\)$", line)
            assert mt, line
            r = int(mt.group(3))
            assert r in members and mt.group(2) == label(r)
            assert w[r] == pytest.approx(w[top[k]], rel=1e-9)  # (the k-th weight; equal weights may swap places)
            assert parse_vec(mt.group(1))[0] == pytest.approx(math.exp(w[r]), rel=1e-9)
            head = "\n" + mt.group(1) + " "
            assert line.startswith((head + " " * max(0, 15 - len(head)))[1:] + label(r))
        # (the report leaves the group's list in its new order: the next 'no change' is judged against it)
        rest = [r for r in members if r not in [int(re.search(r"id = (\d+)", x).group(1)) for x in got]]
        members = [int(re.search(r"id = (\d+)", x).group(1)) for x in got] + rest
        n_reports += 1
    assert n_reports >= 1 and n_reports + n_same == 4  # (M-steps 1, 2, 3 and 5: the first W + 1 and every W-th after)
    rc, so, err = run(["-f", f, "-n", n, "-i", "2", "-w", str(of.n_rules + 7)])
    assert rc != 0 and "Couldn't find rule %d in any normalization groups." % (of.n_rules + 7) in err
    if names:
        short = tmp_path / "short"
        short.write_text("only one\n")
        rc, so, err = run(["-f", f, "-n", n, "-i", "2", "-w", str(rule), "-R", str(short)])
        assert rc != 0 and "Not enough lines in rule names file (%d expected, got 1)" % (of.n_rules - 1) in err


@pytest.mark.gpu
@pytest.mark.parametrize("header", [False, True])
def test_forest_em_cli_rules_by_id(golden_dir, tmp_path, header):
    """-b RULES -B OUT [-F field -C field] (forest-em-params.hpp:158-165, forest-em-params.cpp:135-143): RULES copied with
    " emprob=<weight> emcount=<count>" behind every id=N at a word boundary -- the numbers -o and -O print for rule N -- under a
    $$$ header line that carries the version and the command line (an existing header line is kept and extended).  insert_byid's
    state machine (io.hpp:653-709) is restated here with its quirks: id= after two blanks, or inside a word, is not an id"""
    f, n = os.path.join(golden_dir, "fem.forests"), os.path.join(golden_dir, "fem.norm")
    body = "id=1 first rule\nS(x0:NP) -> x0 ### id=3 more\nxid=4 no boundary\ntwo  id=5 blanks\ntab\tid=6\nidd=2 i id=2\nlast id=7"
    rules = tmp_path / "rules"
    rules.write_text(("$$$ filetype=rule version=0.9 made by hand\n" if header else "") + body)
    o1, o2, ob = (str(tmp_path / x) for x in ("o", "O", "B"))
    args = ["-f", f, "-n", n, "-i", "4", "-o", o1, "-O", o2, "-b", str(rules), "-B", ob, "-F", "p", "-C", "c count"]
    rc, so, err = run(args)
    assert rc == 0, err
    w = open(o1).read().strip().strip("()").split()
    c = open(o2).read().strip().strip("()").split()
    got = open(ob).read()
    head, rest = got.split("\n", 1)
    cmd = " ".join([CLI] + [('"%s"' % a if " " in a else a) for a in args])
    tail = " forest-em-version= {{ {v20}}} floating-point-precision= {{ {double}}} forest-em-cmdline= {{ {" + cmd + "}}}"
    assert head == ("$$$ filetype=rule version=0.9 made by hand" if header else "$$$ filetype=rule version=1.0") + tail
    out, st, N = "", "wait_i", 0
    fields = lambda k: " p=%s c count=%s" % (w[k - 1], c[k - 1]) if 0 < k <= len(w) else ""
    for ch in body:
        space = ch in " \n\t"
        if st == "wait_space":
            st = "wait_i" if space else st
        elif st == "wait_i":
            st = "seen_i" if ch == "i" else "wait_space"
        elif st == "seen_i":
            st = "seen_id" if ch == "d" else "wait_space"
        elif st == "seen_id":
            st, N = ("scan", 0) if ch == "=" else ("wait_space", N)
        else:
            if ch.isdigit():
                N = N * 10 + int(ch)
            else:
                out += fields(N)
                st = "wait_i" if space else "wait_space"
        out += ch
    if st == "scan":
        out += fields(N)
    assert rest == out
    assert " p=" in out.split("\n")[0] and "id=3 p=" in out and "xid=4 no" in out and "two  id=5 blanks" in out and "tab\tid=6 p=" in out
    assert out.endswith("last id=7 p=%s c count=%s" % (w[6], c[6]))
    rc, so, err = run(["-f", f, "-n", n, "-i", "1", "-B", ob])
    assert rc != 0 and "Must provide byid-rule-file." in err


@pytest.mark.gpu
def test_forest_em_cli_options(oracle, tmp_path):
    """initial parameters (-I), add-k smoothing, prior counts, counts output (-O), human probs (-H)"""
    from test_forest_gpu import synth_forests
    ftxt, ntxt = synth_forests(40, 30, seed=5)
    (tmp_path / "f").write_text(ftxt)
    (tmp_path / "n").write_text(ntxt)
    rng = np.random.default_rng(2)
    init = rng.uniform(0.1, 1.0, oracle.OracleForests(ftxt, ntxt).n_rules - 1)  # (a shorter file is an error: below)
    (tmp_path / "i").write_text("(" + " ".join("%.17g" % v for v in init) + ")\n")
    rc, so, err = run(["-f", str(tmp_path / "f"), "-n", str(tmp_path / "n"), "-I", str(tmp_path / "i"), "-o",
                       str(tmp_path / "o"), "-O", str(tmp_path / "c"), "-i", "1", "-H"])
    assert rc == 0, err
    of = oracle.OracleForests(ftxt, ntxt)
    lw = np.zeros(of.n_rules)
    lw[1:1 + len(init)] = np.log(init)
    of.set_weights(lw)
    avg, counts, _ = of.estimate()
    got_c = parse_vec((tmp_path / "c").read_text())
    np.testing.assert_allclose(got_c, np.exp(counts[1:1 + len(got_c)]), rtol=1e-9, atol=1e-300)  # the oracle hands back ln counts
    # one iteration = one estimate and one M-step (em.hpp:126-196; no restarts: nothing restores the first parameters)
    of.maximize()
    np.testing.assert_allclose(parse_vec((tmp_path / "o").read_text())[:len(init)], np.exp(of.weights()[1:1 + len(init)]), rtol=1e-9)
    assert "e^" not in (tmp_path / "o").read_text()
    # FForests::init_rule_weights (forest-em.hpp:299-301): an initial parameter file that does not cover every rule
    (tmp_path / "short").write_text("(" + " ".join("%.17g" % v for v in init[:5]) + ")\n")
    rc, so, err = run(["-f", str(tmp_path / "f"), "-n", str(tmp_path / "n"), "-I", str(tmp_path / "short"), "-i", "1"])
    assert rc != 0 and "Initial params file wasn't large enough" in err
    rc, so, err = run(["-f", str(tmp_path / "f"), "-i", "1"])  # forest-em-params.cpp:59-60
    assert rc != 0 and "Missing normgroups-file" in err


@pytest.mark.gpu
def test_forest_em_cli_initial_parameters_and_final_outputs(oracle, golden_dir, tmp_path):
    """Without -I every norm group starts uniform and a rule of no group at weight 0 (forest-em.hpp:297-318), with -u every
    parameter at 1, with --random-set at random fractions normalised per group; and the reference's final iteration
    (forest-em-params.cpp:125-132): -v the Viterbi derivation of every forest as `best/sum=percent% tree` (forest.hpp:507-632),
    -S the inside sums, -E the per-forest counts -- which the reference prints as `()` (forest-em.hpp:383-389)."""
    from test_forest_gpu import synth_forests
    cases = [(open(os.path.join(golden_dir, "fem.forests")).read(), open(os.path.join(golden_dir, "fem.norm")).read())]
    cases += [synth_forests(40, 30, seed=s) for s in (5, 6)]
    for ci, (ftxt, ntxt) in enumerate(cases):
        (tmp_path / "f").write_text(ftxt)
        (tmp_path / "n").write_text(ntxt)
        for flags, kw in (([], {}), (["-u"], dict(ones=True)), (["--random-set", "--random-seed=9"], dict(random_set=True, seed=9))):
            rc, so, err = run(["-f", str(tmp_path / "f"), "-n", str(tmp_path / "n"), "-o", str(tmp_path / "o"), "-i", "6",
                               "-d", "1.52587890625e-05", "-v", str(tmp_path / "v"), "-S", str(tmp_path / "s"), "-E", str(tmp_path / "e")] + flags)
            assert rc == 0, err
            keep = []
            best, bw, trace = oracle_em(oracle, ftxt, ntxt, 6, keep=keep, **kw)
            its = re.findall(r"^i=(\d+) average log-prob=(\S+)", err, re.M)
            assert len(its) == len(trace), (ci, flags)
            for (i, v), t in zip(its, trace):
                assert float(v) == pytest.approx(t, rel=1e-5), (ci, flags)
            of = keep[0]
            vit = (tmp_path / "v").read_text().split("\n")[:-1]
            ins = (tmp_path / "s").read_text().split("\n")[:-1]
            assert (tmp_path / "e").read_text() == "()\n" * of.n_forests
            assert len(vit) == len(ins) == of.n_forests
            _, _, pf = of.estimate()
            for f in range(of.n_forests):
                line, best_ln = of.viterbi_line(f)
                got_head, got_tree = vit[f].split("% ", 1)
                ref_head, ref_tree = line.split("% ", 1)
                assert got_tree == ref_tree, (ci, flags, f)
                gb, rest = got_head.split("/")
                gs, gp = rest.split("=")
                assert parse_vec("(" + gb + ")")[0] == pytest.approx(math.exp(best_ln), rel=1e-9)
                assert parse_vec("(" + gs + ")")[0] == pytest.approx(math.exp(pf[f]), rel=1e-9)
                assert float(gp) == pytest.approx(100 * math.exp(best_ln - pf[f]), rel=1e-4)
                assert parse_vec("(" + ins[f] + ")")[0] == pytest.approx(math.exp(pf[f]), rel=1e-9)


@pytest.mark.gpu
def test_forest_em_cli_prior_inference(oracle, tmp_path):
    """forest-em --crp=N --prior-inference-stddev=S --prior-inference-local --prior-inference-show: the log says what each
    inferring sweep decided, as the oracle decides on the same uniforms; the written parameters are the oracle's"""
    from carmel_amd._capi import lib
    from test_forest_gpu import synth_forests
    ftxt, ntxt = synth_forests(40, 30, seed=5)
    (tmp_path / "f").write_text(ftxt)
    (tmp_path / "n").write_text(ntxt)
    rc, so, err = run(["-f", str(tmp_path / "f"), "-n", str(tmp_path / "n"), "-o", str(tmp_path / "o"), "--crp=10", "--burnin=3",
                       "--const-alpha=0.3", "--prior-inference-stddev=0.1", "--prior-inference-local", "--prior-inference-show",
                       "--random-seed=23", "--outsample-file=" + str(tmp_path / "samples")])
    assert rc == 0, err
    of = oracle.OracleForests(ftxt, ntxt)
    of.init_rule_weights()
    ref = of.gibbs(lambda i, b, st: lib.carmel_hip_gibbs_uniform(23, i, b, st), 10, burnin=3, alpha=0.3,
                   prior_inference=dict(stddev=0.1, local=True))
    lines = [l for l in (so + err).split("\n") if re.match(r"i=\d+ ", l)]
    assert len(lines) == 11
    tr = ref["prior_trace"]
    for i, l in enumerate(lines):
        said = "accepted" if " accepted new priors with p1=" in l else "rejected" if " rejected new priors with p1=" in l else None
        assert said == (None if tr[i, 0] == 0 else ("accepted" if tr[i, 1] else "rejected")), l
    final = re.search(r"Final prior-scale=\[(.*)\]", so + err).group(1).split()
    np.testing.assert_allclose([float(x) for x in final], ref["prior_cumulative"], rtol=1e-5)
    got = parse_vec((tmp_path / "o").read_text())
    np.testing.assert_allclose(got, np.exp(of.weights()[1:1 + len(got)]), rtol=1e-9, atol=1e-300)
    # --outsample-file (forest-em.hpp:768-787): the final sample, one forest per line, rule ids in the order sampled
    samples = [[int(x) for x in l.split()] for l in (tmp_path / "samples").read_text().split("\n")[:-1]]
    assert samples == ref["samples"]


def test_forest_text_cpu(golden_dir):
    """the command line parses its inputs and fails loudly without a GPU (no CPU fallback)"""
    if not os.path.exists(CLI):
        pytest.skip("forest-em not built")
    from carmel_amd._capi import lib
    if lib.carmel_hip_device_count() > 0:
        pytest.skip("has a GPU")
    rc, so, err = run(["-f", os.path.join(golden_dir, "fem.forests"), "-n", os.path.join(golden_dir, "fem.norm")])
    assert rc != 0
    assert "5 forests, 60 nodes, 16 parameters in 2 normalization groups." in err
    assert "ERROR: carmel_hip_forests_create" in err
    bad = run(["-f", os.path.join(golden_dir, "fem.norm")])
    assert bad[0] != 0 and "forest 1, character" in bad[2]


@pytest.mark.gpu
def test_forest_em_cli_parallel_sampler(oracle, tmp_path):
    """forest-em --crp=N --crp-parallel: the throughput mode through the front end -- valid derivations for every forest
    (--outsample-file), a reproducible run, normalised parameters, and the sample's probability in the region the exact
    chain reaches on the same files"""
    from test_forest_gpu import synth_forests
    ftxt, ntxt = synth_forests(300, 40, seed=8)
    (tmp_path / "f").write_text(ftxt)
    (tmp_path / "n").write_text(ntxt)
    of = oracle.OracleForests(ftxt, ntxt)
    outs = {}
    for tag, flags in (("par", ["--crp-parallel"]), ("par2", ["--crp-parallel"]), ("exact", [])):
        rc, so, err = run(["-f", str(tmp_path / "f"), "-n", str(tmp_path / "n"), "-o", str(tmp_path / ("o" + tag)), "--crp=40",
                           "--burnin=10", "--const-alpha=0.3", "--random-seed=5", "--outsample-file=" + str(tmp_path / ("s" + tag))] + flags)
        assert rc == 0, err
        lp = [float(x) for x in re.findall(r"sample log-prob=(\S+)", so + err)]
        assert len(lp) == 41
        outs[tag] = (parse_vec((tmp_path / ("o" + tag)).read_text()), (tmp_path / ("s" + tag)).read_text(), lp)
    assert outs["par"][1] == outs["par2"][1] and outs["par"][2] == outs["par2"][2]
    samples = [[int(x) for x in l.split()] for l in outs["par"][1].split("\n")[:-1]]
    assert len(samples) == of.n_forests and all(len(s) > 0 for s in samples)
    w = outs["par"][0]
    for g in range(of.n_groups):
        mem = np.asarray(of.group_rule[int(of.group_off[g]):int(of.group_off[g + 1])])
        assert w[mem - 1].sum() == pytest.approx(1.0, abs=1e-9)
    a, b = np.mean(outs["par"][2][-10:]), np.mean(outs["exact"][2][-10:])
    # (a coarse band: on 300 forests the all-at-once sweep is a visibly different chain from the sequential one -- the bias
    # and its decay with corpus size are measured in test_bench_workloads_gpu.py)
    assert abs(a - b) < 0.12 * abs(b)


@pytest.mark.gpu
def test_forest_em_cli_crp_restarts(oracle, tmp_path):
    """forest-em --crp=N --crp-restarts=R [--crp-argmax-final]: the front end's lines run by run ("(random restart r of R):"),
    the kept run, its parameters and sample -- the oracle's chains on the shared uniforms (run r, sweep i: sweep r * (N + 1) + i)"""
    from carmel_amd._capi import lib
    from test_forest_gpu import synth_forests
    ftxt, ntxt = synth_forests(60, 30, seed=4)
    (tmp_path / "f").write_text(ftxt)
    (tmp_path / "n").write_text(ntxt)
    N, R = 12, 4
    for flags, stat in (([], lambda t: t.sum()), (["--crp-argmax-final"], lambda t: t[-1])):
        rc, so, err = run(["-f", str(tmp_path / "f"), "-n", str(tmp_path / "n"), "-o", str(tmp_path / "o"), "--crp=%d" % N, "--burnin=4",
                           "--const-alpha=0.3", "--random-seed=5", "--crp-restarts=%d" % R, "--outsample-file=" + str(tmp_path / "s")] + flags)
        assert rc == 0, err
        assert [int(x) for x in re.findall(r"\(random restart (\d+) of %d\)" % R, err)] == list(range(R + 1))
        lp = np.array([float(x) for x in re.findall(r"sample log-prob=(\S+)", err)]).reshape(R + 1, N + 1)
        best = None
        for r in range(R + 1):
            of = oracle.OracleForests(ftxt, ntxt)
            of.init_rule_weights()
            ref = of.gibbs(lambda i, b, s, r=r: lib.carmel_hip_gibbs_uniform(5, r * (N + 1) + i, b, s), N, burnin=4, alpha=0.3)
            np.testing.assert_allclose(lp[r], ref["iter_logprob"], rtol=1e-5)
            st = stat(ref["iter_logprob"][4:])
            if best is None or st > best[0]:
                best = (st, r, ref["samples"], of.weights().copy())
        assert "Kept run %d of %d" % (best[1], R) in err
        np.testing.assert_allclose(parse_vec((tmp_path / "o").read_text()), np.exp(best[3][1:]), rtol=1e-9, atol=1e-300)
        assert [[int(x) for x in l.split()] for l in (tmp_path / "s").read_text().split("\n")[:-1]] == best[2]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(width=9, burnin=3), dict(width=5, final_counts=True)])
def test_forest_em_cli_final_tables(oracle, tmp_path, kw):
    """forest-em --crp=N --print-counts-to=.. --print-norms-to=.. [--width=W] [--print-file=F] (gibbs_base::print_all at the end of
    FForests::run_gibbs, forest-em.hpp:719; gibbs.hpp:970-1078): the kept run's norm sums and averaged counts, parameters = rule ids,
    norm ids = the groups' numbers from ONE, a locked rule (negative --alpha entry) LOCKED.  The reference holds no output of
    these switches: the text is restated in Python (test_cli_gpu._print_width) over the ORACLE's finalized counts"""
    from carmel_amd._capi import lib
    from test_cli_gpu import _print_width
    from test_forest_gpu import synth_forests
    ftxt, ntxt = synth_forests(50, 30, seed=6)
    (tmp_path / "f").write_text(ftxt)
    (tmp_path / "n").write_text(ntxt)
    of = oracle.OracleForests(ftxt, ntxt)
    of.init_rule_weights()
    N, width, burnin = 8, kw.get("width", 7), kw.get("burnin", 0)
    alphas = np.full(of.n_rules, 0.3)
    alphas[3::7] = -1.0  # some locked rules
    (tmp_path / "a").write_text("(" + " ".join("%.17g" % a for a in alphas) + ")\n")
    args = ["-f", str(tmp_path / "f"), "-n", str(tmp_path / "n"), "--crp=%d" % N, "--burnin=%d" % burnin, "--alpha=" + str(tmp_path / "a"),
            "--random-seed=5", "--print-counts-to=4294967295", "--print-norms-to=4294967295", "--print-file=" + str(tmp_path / "t")]
    if "width" in kw:
        args.append("--width=%d" % width)
    if kw.get("final_counts"):
        args.append("--final-counts")
    rc, so, err = run(args)
    assert rc == 0, err
    ref = of.gibbs(lambda i, b, s: lib.carmel_hip_gibbs_uniform(5, i, b, s), N, burnin=burnin, alpha=0.3, alphas=alphas,
                   final_counts=bool(kw.get("final_counts")))
    x, norm = ref["final"][:, 0], ref["final"][:, 1].astype(int)
    prob = np.exp(of.weights())
    b = N if kw.get("final_counts") else min(burnin, N)
    t = float(N - b)
    nnorm = norm.max() + 2
    ns = np.zeros(nnorm)
    np.add.at(ns, norm[norm >= 0] + 1, x[norm >= 0])
    want = "\n# final best gibbs run (start #0 t=%g):\n" % t
    want += "\n# group\tnormalization group sums i=%d t=%g\n(\n" % (N + 1, t) + "".join(" %g\n" % v for v in ns) + ")\n"
    want += "\n#id\tgroup\tcount\tprob\t\n"
    for r in range(of.n_rules):
        has = norm[r] >= 0
        want += "%d\t%s\t%s\t%s\n" % (r, norm[r] + 1 if has else "LOCKED", _print_width((x[r] if has else 0.0) / (t + 1), width), _print_width(prob[r], width))
    want += "\n"
    got = (tmp_path / "t").read_text()
    bad = [(k, a, c) for k, (a, c) in enumerate(zip(got.split("\n"), want.split("\n"))) if a != c][:5]
    assert not bad and len(got) == len(want), bad
    assert any("LOCKED" in l for l in got.split("\n")[10:])
