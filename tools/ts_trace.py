"""per-phase cycles of tile_sweep_kernel (CARMEL_HIP_LANE_TRACE=<file> during a bench run): python tools/ts_trace.py <file>"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], np.uint64).reshape(-1, 16)
a = a[a[:, 0] > 0]
ng = (a[:, 4] >> np.uint64(32)).astype(np.int64)
ni = (a[:, 4] & np.uint64(0xffffffff)).astype(np.int64)
ml = (a[:, 5] & np.uint64(0xffffffff)).astype(np.int64)
ld, sw, st = a[:, 1].astype(np.int64), a[:, 2].astype(np.int64), a[:, 3].astype(np.int64)
print("tiles", len(a), "span of start stamps", int(a[:, 0].max() - a[:, 0].min()))
print("median cycles: load %d  sweep %d  store %d   (items per tile %d, groups %d)" % (np.median(ld), np.median(sw), np.median(st), np.median(ni), np.median(ng)))
for lo, hi in ((0, 8), (9, 16), (17, 24), (25, 32), (33, 40), (41, 48)):
    m = (ml >= lo) & (ml <= hi)
    if m.any():
        w = a[m][:, 8:12].astype(np.int64)
        print("  maxlen %2d-%2d: tiles %5d groups %4.1f items %5d | load %6d sweep %6d store %6d | per-wave sweep max %6d" % (
            lo, hi, m.sum(), ng[m].mean(), ni[m].mean(), np.median(ld[m]), np.median(sw[m]), np.median(st[m]), np.median(w.max(1))))
