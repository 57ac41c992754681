import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from test_unrolled_gpu import one_tape
from carmel_amd.trainer import HipForwardBackward
for S, deg, n in ((729, 27, 20000), (200, 20, 20000)):
    w, c = one_tape(3, n_states=S, deg=deg, n_sym=27, n_pairs=n, lo=30, hi=80, eps_arcs=False)
    fb = HipForwardBackward(w, c)
    ls = fb.lattice_stats
    fb.estimate(); fb.maximize(1.0)
    t0 = time.perf_counter()
    for _ in range(3):
        fb.estimate(); fb.maximize(1.0)
    dt = (time.perf_counter() - t0) / 3
    print("S=%d arcs=%d pairs=%d: unrolled=%s lattice arcs %.3g, %.1f ms per iteration, %.3g arc-updates/s" % (
        S, w.n_arcs, n, ls.n_bundles == 0, ls.kept_arcs, dt * 1e3, ls.kept_arcs / dt))
    fb.close()
