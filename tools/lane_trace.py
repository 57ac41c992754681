#!/usr/bin/env python3
"""Analyse a CARMEL_HIP_LANE_TRACE dump: per-wave s_memtime stamps
[0] start [1] after forward [2] end [3] hw_id<<32 | maxlen [4..9] forward super-iteration starts [10..15] backward."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
st, mid, en = a[:, 0], a[:, 1], a[:, 2]
maxlen = a[:, 3] & 0xffffffff
print("waves", len(a))
sel = maxlen >= 33
print("waves with maxlen >= 33:", sel.sum(), " median cycles between stamps:")
b = a[sel]
names = ["start->fwd it0 (prologue)", "fwd it0", "fwd it1", "fwd it2(+tail) -> mid", "mid->bwd it0 (prologue)", "bwd it0", "bwd it1", "bwd it2 -> end"]
seq = [b[:, 0], b[:, 4], b[:, 5], b[:, 6], b[:, 1], b[:, 10], b[:, 11], b[:, 12], b[:, 2]]
for i, n in enumerate(names):
    d = seq[i + 1] - seq[i]
    print("  %-28s p10 %7d  p50 %7d  p90 %7d" % (n, *np.percentile(d, [10, 50, 90])))
print("lifetime p50", int(np.median(en[sel] - st[sel])))
t0 = st.min()
print("kernel span (cycles, all classes incl. gaps):", int(en.max() - t0))
