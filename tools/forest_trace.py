#!/usr/bin/env python3
"""CARMEL_HIP_FOREST_TRACE dump: per wave {start, after own-sample table, after inside, after walk (max over lanes), end,
maxlen, n_lanes, -} in shader cycles."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 0] > 0]
ml = a[:, 5]
for lo, hi in ((0, 60), (60, 120), (120, 180), (180, 400)):
    s = (ml >= lo) & (ml < hi)
    if not s.any():
        continue
    d = a[s]
    print("maxlen %3d..%3d waves %5d  table %8d  inside %8d  walk %8d  cheap %8d  total %8d cycles (medians)" % (
        lo, hi, s.sum(), *[np.median(d[:, i + 1] - d[:, i]) for i in range(4)], np.median(d[:, 4] - d[:, 0])))

tot = a[:, 4] - a[:, 0]
o = np.argsort(-tot)[:6]
print("slowest waves (cycles): total, table, inside, walk, cheap, maxlen, lanes")
for i in o:
    print("  ", tot[i], *[int(a[i, k + 1] - a[i, k]) for k in range(4)], a[i, 5], a[i, 6])
print("pct 50/90/99/100 of total:", np.percentile(tot, [50, 90, 99, 100]).astype(int))
