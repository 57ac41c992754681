"""CPU: the forest-em oracle (oracle/forest.hpp) pinned by brute force.  The reference ships forest inputs but no
expected outputs, so inside probabilities and expected rule counts are checked against an
independent exhaustive enumeration of every derivation, on the reference's own unit-test forest strings
(forest-em/forest.hpp:1041-1043) and on forest-em/sample/forests."""
import math
import os
import re

import numpy as np
import pytest

REF_TEST_FORESTS = ["1", "#1(1 #1 (2 #1 (1 1)))", "(1 4)", "(OR (1 4) (1 3))", "(OR (1 4 4) (2 3 4) (2 4 3) (1 5))",
                    "(OR (1 #1(4) #1) (2 #2(3) #1) (2 #1 #2) (1 5))"]


def parse_tree(s):
    """independent little parser: returns nested ('OR'|rule, [children]) with back-references expanded"""
    toks = re.findall(r"#\d+\(|#\d+|\(|\)|OR|\d+", s)
    pos = [0]
    defs = {}

    def node():
        t = toks[pos[0]]
        pos[0] += 1
        m = re.match(r"#(\d+)(\()?$", t)
        if m and not m.group(2):
            return defs[int(m.group(1))]
        if m or t == "(":
            label = toks[pos[0]]
            pos[0] += 1
            kids = []
            n = [label if label == "OR" else int(label), kids]
            if m:
                defs[int(m.group(1))] = n
            while toks[pos[0]] != ")":
                kids.append(node())
            pos[0] += 1
            return n
        return [int(t), []]
    return node()


def derivations(n):
    """all derivations as lists of rule ids"""
    label, kids = n
    if label == "OR":
        out = []
        for k in kids:
            out.extend(derivations(k))
        return out
    out = [[label]]
    for k in kids:
        out = [a + b for a in out for b in derivations(k)]
    return out


def brute(forest_text, w):
    ds = derivations(parse_tree(forest_text))
    ps = [math.prod(w[r] for r in d) for d in ds]
    z = sum(ps)
    counts = np.zeros(len(w))
    for d, p in zip(ds, ps):
        for r in d:
            counts[r] += p / z
    return z, counts


@pytest.mark.parametrize("text", REF_TEST_FORESTS[2:] + ["(OR #1(OR 1 1) #1 (2 3) (3 3))"])
def test_inside_and_counts_match_exhaustive_enumeration(oracle, text):
    rng = np.random.default_rng(len(text))
    f = oracle.OracleForests(text, "()")
    w = rng.uniform(0.1, 0.9, f.n_rules)
    f.set_weights(np.log(w))
    avg, counts_ln, pf = f.estimate()
    z, counts = brute(text, w)
    assert pf[0] == pytest.approx(math.log(z), rel=1e-12)
    np.testing.assert_allclose(np.exp(counts_ln)[1:], counts[1:], rtol=1e-12, atol=1e-300)


def test_sample_forests_file(oracle, golden_dir):
    text = open(os.path.join(golden_dir, "fem.forests")).read()
    norm = open(os.path.join(golden_dir, "fem.norm")).read()
    f = oracle.OracleForests(text, norm)
    assert (f.n_forests, f.n_groups) == (5, 2)
    rng = np.random.default_rng(0)
    w = rng.uniform(0.1, 0.9, f.n_rules)
    f.set_weights(np.log(w))
    avg, counts_ln, pf = f.estimate()
    # the five forests of the file, one per top-level s-expression
    depth, start, forests = 0, None, []
    for i, ch in enumerate(text):
        if ch == "(":
            if depth == 0 and start is None:
                start = i
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                forests.append(text[start:i + 1])
                start = None
    tot = np.zeros(f.n_rules)
    for k, ft in enumerate(forests):
        z, c = brute(ft, w)
        assert pf[k] == pytest.approx(math.log(z), rel=1e-12)
        tot += c
    np.testing.assert_allclose(np.exp(counts_ln)[1:], tot[1:], rtol=1e-12, atol=1e-300)
    # M-step: groups ((1 2 7) (3 4 5 6)); rules outside every group keep their weights (forest-em.README)
    before = f.weights().copy()
    f.maximize()
    after = np.exp(f.weights())
    assert after[[1, 2, 7]].sum() == pytest.approx(1.0, rel=1e-12)
    assert after[[3, 4, 5, 6]].sum() == pytest.approx(1.0, rel=1e-12)
    np.testing.assert_allclose(after[[3, 4, 5, 6]], tot[[3, 4, 5, 6]] / tot[[3, 4, 5, 6]].sum(), rtol=1e-12)
    np.testing.assert_array_equal(f.weights()[8:], before[8:])


def test_em_increases_likelihood(oracle, golden_dir):
    f = oracle.OracleForests(open(os.path.join(golden_dir, "fem.forests")).read(),
                             open(os.path.join(golden_dir, "fem.norm")).read())
    f.maximize()  # start from normalised weights
    prev = -1e300
    for _ in range(6):
        avg, _, _ = f.estimate()
        assert avg >= prev - 1e-12
        prev = avg
        f.maximize()


def test_fem_export_forests_carry_the_lattice_probabilities(oracle, golden_dir):
    """carmel --fem-forest (cascade.h:117-165 as restated in oracle/fem.hpp): each pair's derivation lattice written
    as a forest.  Two independent restatements must agree: the inside probability of forest p under the cascade's
    own arc weights (forest reader + inside_rec) is the pair's forward probability in the composed transducer
    (derivations + compute_fb), and every arc id of a norm group belongs to one member's (state, input) group."""
    rd = lambda n: open(os.path.join(golden_dir, n)).read()
    for names, normby in ((("cipher.wfsa", "cipher.fst"), "NC"), (("chain.1", "chain.2"), "CC")):
        corpus_name = "cipher.data" if names[0].startswith("cipher") else "chain.corpus"
        oc = oracle.OracleCascade([rd(n) for n in names])
        w = oc.composed()
        corp = oc.corpus(rd(corpus_name))
        ftxt = oracle.fem_export(oc, corp, 0, normby)
        ntxt = oracle.fem_export(oc, corp, 1, normby)
        ptxt = oracle.fem_export(oc, corp, 2, normby)
        params = [float(t[2:]) if t.startswith("e^") else (math.log(float(t)) if float(t) > 0 else -math.inf)
                  for t in ptxt.split()]
        assert len(params) == oc.n_params
        np.testing.assert_allclose(params, oc.param_logw, rtol=1e-12, atol=1e-12)
        est = oracle.estimate(w, corp)
        kept = est["pair_logprob"][est["has_deriv"]]
        of = oracle.OracleForests(ftxt, ntxt)
        assert of.n_forests == len(kept)
        lw = np.zeros(max(of.n_rules, len(params) + 1))[:of.n_rules]
        lw[1:min(of.n_rules, len(params) + 1)] = params[:of.n_rules - 1]
        of.set_weights(lw)
        _, _, per_forest = of.estimate()
        np.testing.assert_allclose(per_forest, kept, rtol=1e-10, atol=1e-10)
        ids = [int(x) for x in re.findall(r"\d+", ntxt)]
        assert len(ids) == len(set(ids)) and max(ids) <= oc.n_params


def best_tree(n, w):
    """independent Viterbi over the parsed tree: (probability, printed tree); an OR node keeps its FIRST best child"""
    label, kids = n
    if label == "OR":
        best = None
        for k in kids:
            c = best_tree(k, w)
            if best is None or c[0] > best[0]:
                best = c
        return best
    p, parts = w[label], []
    for k in kids:
        cp, ct = best_tree(k, w)
        p *= cp
        parts.append(ct)
    return p, (str(label) if not kids else "(%d %s)" % (label, " ".join(parts)))


@pytest.mark.parametrize("text", REF_TEST_FORESTS[2:] + ["(OR #1(OR 1 1) #1 (2 3) (3 3))", "(OR (1 (OR 2 3) (OR (4 5) 6)) (2 #7(OR 3 (5 6)) #7))"])
def test_viterbi_matches_exhaustive_enumeration(oracle, text):
    """forest-em -v (forest.hpp:507-632): the best derivation's probability is the maximum over all derivations, the printed
    tree is an independent recursion's (first best child on ties: the weights below make ties on purpose), and the line reads
    best/sum=percent% tree"""
    for tie in (False, True):
        rng = np.random.default_rng(len(text))
        f = oracle.OracleForests(text, "()")
        w = np.full(f.n_rules, 0.5) if tie else rng.uniform(0.1, 0.9, f.n_rules)
        f.set_weights(np.log(w))
        line, best_ln = f.viterbi_line(0, mode=2)  # never-log weights
        ds = derivations(parse_tree(text))
        ps = [math.prod(w[r] for r in d) for d in ds]
        assert math.exp(best_ln) == pytest.approx(max(ps), rel=1e-12)
        p, tree = best_tree(parse_tree(text), w)
        head, got_tree = line.split("% ", 1)
        # (two derivations that differ in the order of their factors tie up to rounding in a random draw: the tree is then
        # decided by the last bit; with the weights all 1/2 every product is exact and the FIRST best child must win)
        n_best = sum(1 for q in ps if abs(q - max(ps)) <= 1e-12 * max(ps))
        if tie or n_best == 1:
            assert got_tree == tree
        assert math.prod(w[int(r)] for r in re.findall(r"\d+", got_tree)) == pytest.approx(max(ps), rel=1e-12)
        b, rest = head.split("/")
        z, pct = rest.split("=")
        assert float(b) == pytest.approx(p, rel=1e-12) and float(z) == pytest.approx(sum(ps), rel=1e-12)
        assert float(pct) == pytest.approx(100 * p / sum(ps), rel=1e-5)


def test_initial_parameters_and_random_sets(oracle, golden_dir):
    """FForests::init_rule_weights / randomize (forest-em.hpp:297-318, 393-399; normalize.hpp:212-238): uniform per norm group
    with the rules of no group at weight zero, all ones with -u, random fractions divided by their group's sum"""
    ftxt = "(OR (1 4) (2 3) (5 6))\n(OR (7 1) (2 8))\n"
    ntxt = "((1 2 5) (3 4) (6))"
    f = oracle.OracleForests(ftxt, ntxt)
    f.init_rule_weights()
    w = np.exp(f.weights())
    np.testing.assert_allclose(w[[1, 2, 5]], 1 / 3)
    np.testing.assert_allclose(w[[3, 4]], 1 / 2)
    assert w[6] == 1.0 and w[7] == 0.0 and w[8] == 0.0  # 7 and 8 are in no group
    f.init_rule_weights(ones=True)
    np.testing.assert_allclose(np.exp(f.weights()), 1.0)
    fr = np.array([0, .2, .3, .9, .1, .5, .7, .4, .4])
    f.randomize(fr)
    w = np.exp(f.weights())
    np.testing.assert_allclose(w[[1, 2, 5]], fr[[1, 2, 5]] / fr[[1, 2, 5]].sum())
    np.testing.assert_allclose(w[[3, 4]], fr[[3, 4]] / fr[[3, 4]].sum())
    assert w[6] == pytest.approx(1.0) and w[7] == pytest.approx(1.0) and w[8] == pytest.approx(1.0)  # untouched since -u


def test_exact_chain_settles_on_the_enumerated_stationary_distribution(oracle):
    """The sampler restatement pinned on something other than itself: on a corpus small enough to enumerate (5 forests,
    1 600 joint assignments) the transition matrix of one sweep is built from the definition of a block's proposal
    (tests/forest_enum.py; forest-em.hpp:750-766, gibbs.hpp:589-592, 769-792) and the time-averaged rule probabilities of the
    oracle's chain must be its stationary expectations: 16 chains of 8 000 sweeps, per rule z = (mean - exact) / standard
    error over chains, |z| < 4.9 (Student t, 15 degrees of freedom, overall level 1e-3 over 8 rules, two-sided).  The
    stale-count sweep's stationary distribution (a different chain) must lie outside that band: the test discriminates."""
    import math
    from carmel_amd._capi import lib
    from forest_enum import TOY_FORESTS, TOY_NORM, toy_setup
    of = oracle.OracleForests(TOY_FORESTS, TOY_NORM)
    lw = np.log(np.random.default_rng(3).uniform(0.2, 1.0, of.n_rules))
    derivs, exact = toy_setup(of.node_off, of.label, of.ref, of.next, of.n_rules, of.group_off, of.group_rule, lw, 0.5)
    chains = []
    for seed in range(16):
        of.set_weights(lw)
        of.gibbs(lambda i, b, s: lib.carmel_hip_gibbs_uniform(5 + 7919 * seed, i, b, s), 8000, burnin=500, alpha=0.5)
        chains.append(np.exp(of.weights()[1:]))
    chains = np.asarray(chains)
    mean, se = chains.mean(0), chains.std(0, ddof=1) / math.sqrt(len(chains))
    z = (mean - exact[False][1:]) / se
    assert np.all(np.abs(z) < 4.9), z
    assert np.max(np.abs((mean - exact[True][1:]) / se)) > 6.0
