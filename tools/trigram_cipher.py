#!/usr/bin/env python3
"""Decipherment under a character TRIGRAM model (carmel/sample/decipher/plain.tri.wfsa's shape): a locked acceptor with
one state per character pair (27^2 + start states, 27^3 arcs *e*:"C") over the 27x27 substitution channel, and a corpus
of enciphered lines.  The composed transducer has ~730 states: the workgroup-per-pair unrolled sweep.
usage: python3 tools/trigram_cipher.py N_LINES OUT_DIR"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from carmel_amd.synth import CIPHER_PLAIN


def main():
    n_lines, out = int(sys.argv[1]), sys.argv[2]
    rng = np.random.default_rng(5)
    n = len(CIPHER_PLAIN)
    tri = rng.dirichlet(np.full(n, 0.3), size=(n, n))  # P(c | a, b)
    big = rng.dirichlet(np.full(n, 0.3), size=n)
    name = lambda a, b: "%s%s" % (CIPHER_PLAIN[a], CIPHER_PLAIN[b])
    lm = ["END"]
    for b in range(n):  # the line starts with the separator "_" (index 0), then a bigram step
        lm.append('(START (S_%s *e* "%s" %.12g!))' % (CIPHER_PLAIN[b], CIPHER_PLAIN[b], 1.0 if b == 0 else 1e-9))
    for b in range(n):
        for c in range(n):
            lm.append('(S_%s (%s *e* "%s" %.12g!))' % (CIPHER_PLAIN[b], name(b, c), CIPHER_PLAIN[c], big[b, c]))
    for a in range(n):
        for b in range(n):
            scale = 0.9 if b == 0 else 1.0
            for c in range(n):
                lm.append('(%s (%s *e* "%s" %.12g!))' % (name(a, b), name(b, c), CIPHER_PLAIN[c], tri[a, b, c] * scale))
            if b == 0:
                lm.append("(%s (END *e* *e* 0.1!))" % name(a, b))
    low = [s.lower() if s != "_" else "_" for s in CIPHER_PLAIN]
    ch = ["0"] + ['(0 (0 "%s" "%s"))' % (CIPHER_PLAIN[a], low[c]) for a in range(n) for c in range(n)]
    key = np.concatenate([[0], rng.permutation(n - 1) + 1])
    lines = []
    for L in rng.integers(30, 81, size=n_lines):
        a, b, seq = 0, 0, [0]
        b = int(rng.choice(n, p=big[0]))
        seq.append(b)
        a, b = 0, b
        for _ in range(int(L) - 3):
            c = int(rng.choice(n, p=tri[a, b]))
            seq.append(c)
            a, b = b, c
        seq.append(0)
        lines.append("\n" + " ".join('"%s"' % low[key[c]] for c in seq))
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, "lm3.wfsa"), "w").write("\n".join(lm) + "\n")
    open(os.path.join(out, "ch.fst"), "w").write("\n".join(ch) + "\n")
    open(os.path.join(out, "corpus"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
