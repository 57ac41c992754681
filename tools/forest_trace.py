#!/usr/bin/env python3
"""CARMEL_HIP_FOREST_TRACE dump of forest_sample_multi_kernel: per workgroup {start, after staging (tables + proposal
probabilities), after the inside pass, after the walk, -, most nodes, greatest height, most visits} in shader cycles."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print("workgroups %d, span %d cycles" % (len(a), a[:, 3].max() - t0))
n = a[:, 5]
for lo, hi in ((0, 30), (30, 60), (60, 90), (90, 130), (130, 1000)):
    s = (n >= lo) & (n < hi)
    if not s.any():
        continue
    d = a[s]
    print("nodes %3d..%3d wgs %5d  staging %7d  inside %7d  walk %7d  total %7d cycles (medians)  height %d visits %d" % (
        lo, hi, s.sum(), *[np.median(d[:, i + 1] - d[:, i]) for i in range(3)], np.median(d[:, 3] - d[:, 0]),
        np.median(d[:, 6]), np.median(d[:, 7])))
tot = a[:, 3] - a[:, 0]
print("pct 50/90/99/100 of total:", np.percentile(tot, [50, 90, 99, 100]).astype(int))
print("start offsets pct 50/90/100:", np.percentile(a[:, 0] - t0, [50, 90, 100]).astype(int))
