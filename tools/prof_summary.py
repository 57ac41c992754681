#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats) from a rocprofv3 results .db into a small CSV that
can be committed under profiles/.  usage: prof_summary.py results.db out.csv [note]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select * from top_kernels"))
    cols = [d[0] for d in cur.description]
    with open(out, "w", newline="") as f:
        if note:
            f.write("# %s\n" % note)
        w = csv.writer(f)
        w.writerow(cols)
        for r in rows:
            w.writerow(r)
    print(open(out).read())


if __name__ == "__main__":
    main()
