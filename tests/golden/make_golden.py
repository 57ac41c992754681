#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference tree (run in the build container only; /root/reference does
not exist on the GPU box).  What is copied is DATA: the reference's own tutorial/test fixtures (transducer
and corpus files) and the numeric content of its recorded run trace
(carmel/carmel-tutorial/commands.trace), parsed into JSON.  No reference source code is copied.
"""
import json, os, re, shutil, sys

REF = "/root/reference/carmel"
HERE = os.path.dirname(os.path.abspath(__file__))
TUT = os.path.join(REF, "carmel-tutorial")

FIXTURES = [
    ("carmel-tutorial/epron-jpron.data", "epron-jpron.data"),
    ("carmel-tutorial/epron-jpron.fst", "epron-jpron.fst"),
    ("carmel-tutorial/cipher.data", "cipher.data"),
    ("carmel-tutorial/cipher.wfsa", "cipher.wfsa"),
    ("carmel-tutorial/cipher.fst", "cipher.fst"),
    ("carmel-tutorial/cipher.fst.trained", "cipher.fst.trained"),
    ("carmel-tutorial/cipher.wfsa.trained", "cipher.wfsa.trained"),
    ("carmel-tutorial/tagging.data", "tagging.data"),
    ("carmel-tutorial/tagging.fsa", "tagging.fsa"),
    ("carmel-tutorial/tagging.fst", "tagging.fst"),
    # written twice by the tutorial's command script (EM at commands:19, then `carmel --crp -M 6000` at commands:33): the
    # committed copies are the SAMPLER's output -- every never-sampled arc out of state 0 reads 0.01 / (1005 + 44 * 0.01).
    # The reference-held vector for the Gibbs bookkeeping: tests/crp_pin.py
    ("carmel-tutorial/tagging.fsa.trained", "tagging.fsa.crp-trained"),
    ("carmel-tutorial/tagging.fst.trained", "tagging.fst.crp-trained"),
    ("test/train.a", "train.a"),
    ("test/train.a.w", "train.a.w"),
    ("test/train.a.u", "train.a.u"),
    ("sample/wfst3", "wfst3"),
    ("sample/wfst3c", "wfst3c"),
    ("sample/chain.1", "chain.1"),
    ("sample/chain.2", "chain.2"),
    ("sample/chain.corpus", "chain.corpus"),
]
FOREST_FIXTURES = [("sample/forests", "fem.forests"), ("sample/norm", "fem.norm"), ("sample/forest", "fem.forest")]

ITER = re.compile(
    r"i=(\d+) \(rate=([^)]*)\): probability=2\^(\S+) per-output-symbol-perplexity\(N=(\d+)\)=2\^(\S+) "
    r"per-example-perplexity\(N=(\d+)\)=2\^(\S+)( \(new best\))?"
    r"(?: \(relative-perplexity-ratio=([^)]+)\))?(?:, max\{d\(weight\)\}=(\S+))?")


def parse_iters(lines):
    out = []
    for ln in lines:
        ln = ln.lstrip(".0123456789\n") if ln.startswith(".") else ln
        m = ITER.search(ln)
        if m:
            rr = m.group(9)
            out.append({
                "iter": int(m.group(1)), "log2_prob": float(m.group(3)), "n_symbol": int(m.group(4)),
                "log2_ppx_symbol": float(m.group(5)), "n_example": int(m.group(6)),
                "log2_ppx_example": float(m.group(7)), "new_best": bool(m.group(8)),
                "rel_ppx_ratio": rr, "max_dweight": (float(m.group(10)) if m.group(10) else None)})
    return out


def sample_corpus(arcs, final, n_pairs, seed):
    """config 1 (BASELINE.json configs[0]): a corpus of n_pairs output strings read off random start->final
    paths of a small *e*-input transducer, each arc chosen in proportion to its weight (what `carmel -g` does,
    fst.cc:24-79, with this script's own numpy stream).  arcs: {state: [(dest, out_symbol_or_None, weight)]}.
    Format: blank input line, output line (cipher/epsilon-string-pairs)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lines = []
    for _ in range(n_pairs):
        s, out = 0, []
        while s != final:
            ws = np.array([a[2] for a in arcs[s]])
            k = int(rng.choice(len(ws), p=ws / ws.sum()))
            if arcs[s][k][1]:
                out.append(arcs[s][k][1])
            s = arcs[s][k][0]
        lines += ["", " ".join(out)]
    return "\n".join(lines) + "\n"


def main():
    # config 1 corpora, sampled from the reference's own toy transducers test/train.a.w and sample/wfst3
    open(os.path.join(HERE, "train.a.w.corpus100"), "w").write(sample_corpus(
        {0: [(2, None, .1), (1, "b", .6), (0, "a", .3)], 1: [(2, "a", .3), (0, "b", .7)]}, 2, 100, 1))
    open(os.path.join(HERE, "wfst3.corpus100"), "w").write(sample_corpus(
        {0: [(2, None, .1), (1, "b", .6), (0, "a", .3), (4, "a", .1)], 1: [(2, "a", .3), (0, "b", .7)], 4: [(2, "c", 1.)]},
        2, 100, 2))
    for src, dst in FIXTURES:
        shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
    for src, dst in FOREST_FIXTURES:
        shutil.copyfile(os.path.join(REF, "..", "forest-em", src), os.path.join(HERE, dst))
    tr = open(os.path.join(TUT, "commands.trace"), errors="replace").read().split("\n")
    # line ranges (1-based) cited in SURVEY.md section 8c
    gold = {
        "epron-jpron": {"command": "carmel -t epron-jpron.data epron-jpron.fst", "trace_lines": "7-19",
                        "iters": parse_iters(tr[6:19]), "final_wfst": "\n".join(tr[19:77]) + "\n",
                        "composed": None},
        "tagging": {"command": "carmel --train-cascade -HJ tagging.data tagging.fsa tagging.fst",
                    "trace_lines": "5866-5890", "iters": parse_iters(tr[5865:5890]),
                    "composed": {"states": 46, "arcs": 400994}},
        "cipher": {"command": "carmel --train-cascade -HJ cipher.data cipher.wfsa cipher.fst",
                   "trace_lines": "6903-6952", "iters": parse_iters(tr[6902:6952]),
                   "composed": {"states": 57, "arcs": 11511}},
    }
    # carmel --crp -M 6000 tagging.data tagging.fsa tagging.fst (commands:33; trace lines 6976-12990).  The lattice
    # statistics are deterministic; note what derivations::statistics holds (derivations.h:197-210, 617-618, 687): "pre"
    # arcs accumulate over all pairs, while pre.states / post.states / post.arcs are ASSIGNED per pair, i.e. they are
    # the LAST pair's.  The 6001 per-sweep "sample prob" lines depend on the reference's Boost random stream.
    crp = []
    for ln in tr[6988:12990]:
        m = re.match(r"Gibbs i=(\d+) 0\.+ sample prob=2\^(\S+) per-point-ppx\(N=(\d+)\)=2\^(\S+) per-block-ppx\(N=(\d+)\)", ln)
        if m:
            crp.append(float(m.group(2)))
            n_sym, n_blocks = int(m.group(3)), int(m.group(5))
    assert len(crp) == 6001
    m = re.search(r"Pre pruning: \((\d+) states, (\d+) arcs\)\nPost pruning: \((\d+) states, (\d+) arcs\)", "\n".join(tr[6980:6988]))
    gold["tagging-crp"] = {"command": "carmel --crp -M 6000 tagging.data tagging.fsa tagging.fst", "trace_lines": "6976-12990",
                           "pre_states_last_pair": int(m.group(1)), "pre_arcs_all_pairs": int(m.group(2)),
                           "post_states_last_pair": int(m.group(3)), "post_arcs_last_pair": int(m.group(4)),
                           "n_symbols": n_sym, "n_blocks": n_blocks, "log2_sample_prob": crp,
                           "burned_in_avg_log2": 214371}
    for k, v in gold.items():
        print(k, len(v.get("iters", v.get("log2_sample_prob"))), "iterations")
    json.dump(gold, open(os.path.join(HERE, "trace_expected.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
