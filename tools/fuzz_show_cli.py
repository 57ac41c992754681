#!/usr/bin/env python3
"""Prints the log lines of both command lines for one seed of the random-cascade CLI test (what a fuzz failure looked like):
    python tools/fuzz_show_cli.py 9006"""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_cli_gpu as C  # noqa: E402

seed = int(sys.argv[1])
real_run = subprocess.run
logs = []


def spy(cmd, *a, **k):
    p = real_run(cmd, *a, **k)
    logs.append((os.path.basename(cmd[0]), p.stderr if isinstance(p.stderr, str) else ""))
    return p


subprocess.run = spy
with tempfile.TemporaryDirectory() as d:
    try:
        C.test_random_cascades_train_like_the_oracle(pathlib.Path(d), seed)
        print("passed")
    except BaseException as e:  # noqa: BLE001
        print("FAILED:", type(e).__name__)
for name, err in logs:
    print("==", name)
    print("\n".join(l for l in err.split("\n") if l.startswith(("i=", "Converged", "Maximum"))))
