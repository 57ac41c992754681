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


def main():
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
    for k, v in gold.items():
        print(k, len(v["iters"]), "iterations")
    json.dump(gold, open(os.path.join(HERE, "trace_expected.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
