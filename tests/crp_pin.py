"""The reference-held vector for the Gibbs sampler's bookkeeping.

carmel-tutorial/tagging.{fsa,fst}.trained were written last by `carmel --crp -M 6000 tagging.data tagging.fsa tagging.fst`
(commands:33; the EM run of commands:19 wrote them first and was overwritten).  They are the reference's own output of
   priors --priors unset -> min_prior 0.01 per parameter        (gibbs.cc:390-397)
   prior pseudo-count = alpha * p0 * |group|, p0 uniform here     (gibbs.cc:114-186; gibbs.hpp:589-592)
   final_counts time-averaged over the 6001 samples, burn-in 0    (gibbs.hpp:626-638; delta_sum.hpp:49-106)
   probs_to_cascade: count / norm-group sum                       (gibbs.cc:66-76)
The sampled sequence depends on Boost's random stream, but part of the output does not:
 * a parameter whose count is the same in every derivation of every block (arcs no derivation can use, arcs every
   derivation must use), in a norm group whose total is the same in every sample, has ONE possible time average; e.g. every
   never-used arc out of state 0 of the tag-bigram acceptor reads 0.01 / (1005 + 44 * 0.01) = 9.94589433481821e-06;
 * a norm group's never-sampled members all read 0.01 / (T + n * 0.01), T the time-averaged group total, n its size -- so
   T can be read back off the file, and the T of the acceptor's groups must sum to the 25 120 arcs every sample uses
   (24 115 tokens + 1005 sentence ends), whatever was sampled.
The rest (which tags the chain settled on) is distributional and the chain mixes slowly between tag labellings: two
chains agree on the bulk, not parameter by parameter; the bound below is on the correlation of the large parameters.

closed_form(...) works the first kind out from the files alone -- no sampler involved: position i of a sentence admits the
tags A_i the lexicon lists for its word; a derivation is any tag sequence with t_i in A_i (the acceptor has every bigram).
check(...) first holds the REFERENCE's file to that closed form (so the reading of the file is itself tested), then the
sampler under test."""
import re

import numpy as np

_TOK = re.compile(r'"(?:[^"\\]|\\.)*"|[()]|[^\s()]+')
_NUM = re.compile(r'^(e\^)?-?[0-9.]+(e-?[0-9]+)?!?$')


def parse_wfst(txt):
    """{(src, dst, in, out): weight} and the arcs in file order, of a carmel transducer text (wfstio.cc:341-506; the subset
    the tutorial files use: `(src (dst [in [out]] [w]) ...)`, several arcs per line or one)"""
    toks = _TOK.findall(txt)
    i, arcs, order = 1, {}, []
    while i < len(toks):
        assert toks[i] == "("
        src = toks[i + 1]
        i += 2
        while toks[i] == "(":
            j, f = i + 1, []
            while toks[j] != ")":
                f.append(toks[j])
                j += 1
            dst, rest, w = f[0], f[1:], "1"
            if rest and _NUM.match(rest[-1]):
                w = rest.pop()
            a = rest[0] if rest else "*e*"
            b = rest[1] if len(rest) > 1 else a
            w = w.rstrip("!")
            k = (src, dst, a, b)
            arcs[k] = float(np.exp(float(w[2:]))) if w.startswith("e^") else float(w)
            order.append(k)
            i = j + 1
        assert toks[i] == ")"
        i += 1
    return arcs, order


def _groups(order):
    g = {}
    for k in order:  # conditional normalisation: one group per (source state, input symbol) (fst.h:1362-1446)
        g.setdefault((k[0], k[2]), []).append(k)
    return g


def group_totals(arcs, order, prior=0.01):
    """time-averaged total of every norm group that has a floor (its minimum shared by >= 2 members: never-sampled ones)"""
    out = {}
    for gk, ks in _groups(order).items():
        v = sorted(arcs[k] for k in ks)
        if len(v) >= 2 and v[0] > 0 and abs(v[1] - v[0]) <= 1e-12 * v[0]:
            out[gk] = prior / v[0] - len(ks) * prior
    return out


def closed_form(fsa_text, fst_text, data_text, prior=0.01):
    """({acceptor arc: value}, {lexicon arc: value}) for every parameter whose time-averaged probability is the same whatever
    was sampled: (c + prior) / (T + n * prior), c its count in every derivation, T its norm group's total in every sample"""
    fsa, fsa_order = parse_wfst(fsa_text)
    fst, fst_order = parse_wfst(fst_text)
    tags_of = {}
    for (_, _, tag, word) in fst_order:
        tags_of.setdefault(word, set()).add(tag.strip('"'))
    sents = [_TOK.findall(l) for l in data_text.split("\n")[1::2] if l.strip()]
    start, final = fsa_order[0][0], fsa_text.split()[0]
    free = set()      # tags (acceptor states) some position could or could not take: their group totals move
    total = {}        # tag -> tokens it must take
    pair_free, pair_n = set(), {}
    word_n = {}
    for words in sents:
        adm = [{start}] + [tags_of[w] for w in words] + [{final}]
        for w in words:
            word_n[w] = word_n.get(w, 0) + 1
        for a in adm[:-1]:
            if len(a) > 1:
                free |= a
            else:
                t = next(iter(a))
                total[t] = total.get(t, 0) + 1
        for a, b in zip(adm, adm[1:]):
            if len(a) == 1 and len(b) == 1:
                k = (next(iter(a)), next(iter(b)))
                pair_n[k] = pair_n.get(k, 0) + 1
            else:
                pair_free |= {(x, y) for x in a for y in b}
    out_fsa, out_fst = {}, {}
    num = {}  # the trained files name states by index: first seen first, sources before destinations (wfstio.cc:356-368, 594-625)
    for k in fsa_order:
        for name in k[:2]:
            num.setdefault(name, str(len(num)))
    for gk, ks in _groups(fsa_order).items():
        src = gk[0]
        if src in free:
            continue
        for k in ks:
            if (src, k[1]) not in pair_free:
                out_fsa[(num[k[0]], num[k[1]]) + k[2:]] = (pair_n.get((src, k[1]), 0) + prior) / (total.get(src, 0) + len(ks) * prior)
    for gk, ks in _groups(fst_order).items():
        tag = gk[1].strip('"')
        if tag in free:
            continue
        for k in ks:  # every word this tag can emit has no other tag, or does not occur
            out_fst[k] = (word_n.get(k[3], 0) + prior) / (total.get(tag, 0) + len(ks) * prior)
    return out_fsa, out_fst


def check(ref_texts, run_a, run_b, min_corr, inputs, n_arcs_per_sample=25120.0, min_fixed_totals=8):
    """ref_texts / run_a / run_b: (acceptor text, transducer text) as the reference / two runs of the sampler under test
    wrote them; inputs: (tagging.fsa, tagging.fst, tagging.data).  Returns a dict of what was found, after asserting it."""
    res = {}
    cf = closed_form(*inputs)
    for m, name in enumerate(("fsa", "fst")):
        ref, _ = parse_wfst(ref_texts[m])
        a, _ = parse_wfst(run_a[m])
        b, _ = parse_wfst(run_b[m])
        for k, v in cf[m].items():
            assert abs(ref[k] - v) <= 1e-12 * v, ("the reference's file", name, k, ref[k], v)
            assert abs(a[k] - v) <= 1e-12 * v and abs(b[k] - v) <= 1e-12 * v, (name, k, a[k], b[k], v)
        res[name + "_closed_form"] = len(cf[m])
    assert res["fsa_closed_form"] >= 200 and res["fst_closed_form"] >= 10, res
    for m, name in enumerate(("fsa", "fst")):
        ref, _ = parse_wfst(ref_texts[m])
        a, order = parse_wfst(run_a[m])
        b, _ = parse_wfst(run_b[m])
        assert set(ref) == set(a) == set(b), name
        keys = [k for k in order if ref[k] > 0 or a[k] > 0]
        rv, av, bv = (np.array([d[k] for k in keys]) for d in (ref, a, b))
        rel = np.abs(rv - av) / np.maximum(rv, 1e-300)
        res[name + "_equal_to_reference"] = int((rel <= 1e-11).sum())  # (nothing outside the closed form coincides)
        big = (rv > 1e-3) | (av > 1e-3)
        res[name + "_corr"] = float(np.corrcoef(rv[big], av[big])[0, 1])
        assert res[name + "_corr"] >= min_corr, (name, res[name + "_corr"])
        tr, ta = group_totals(ref, order), group_totals(a, order)
        res[name + "_groups_with_floor"] = len(tr)
        both = [g for g in tr if g in ta]
        d = np.array([abs(tr[g] - ta[g]) / max(tr[g], 1.0) for g in both])
        res[name + "_groups_fixed_total"] = int((d <= 1e-9).sum())
        res[name + "_median_total_diff"] = float(np.median(d))
        if name == "fsa":  # every row of the tag-bigram acceptor has never-sampled members
            assert len(tr) == len(ta) == len(_groups(order)) == 45
            assert abs(sum(tr.values()) - n_arcs_per_sample) <= 1e-6, sum(tr.values())
            assert abs(sum(ta.values()) - n_arcs_per_sample) <= 1e-6, sum(ta.values())
            start = [k for k in keys if k[0] == "0"]
            floor = 0.01 / (1005 + len(start) * 0.01)
            assert len(start) == 44 and sum(1 for k in start if abs(ref[k] - floor) <= 1e-12 * floor) >= 10
        assert res[name + "_median_total_diff"] <= (0.08 if name == "fsa" else 0.16), res
    # (closed form: in the tag-bigram acceptor the arcs out of the start state that no sentence can take and the rows of the
    # tags whose words are unambiguous; in the lexicon the 7 tags all of whose words have one tag)
    assert res["fsa_groups_fixed_total"] >= min_fixed_totals, res
    return res
