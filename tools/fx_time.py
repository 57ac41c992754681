#!/usr/bin/env python3
"""Times the exact (sequential) forest chain -- forest_exact_kernel -- on config 5's generator.
usage: fx_time.py [forests] [sweeps]; CARMEL_HIP_FOREST_EXACT_CLK=1 adds the kernel's per-phase cycle counts (stderr),
CARMEL_HIP_FOREST_EXACT_HOST=1 runs the host-driven loop it replaces."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import carmel_amd
from carmel_amd import synth
from carmel_amd.forests import HipForests

carmel_amd.options_from_env()

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
node_off, label, ref, nxt, n_rules, goff, grule = synth.random_forests(nf)
lw = np.log(np.random.default_rng(4).uniform(0.05, 1.0, n_rules))
hf = HipForests(node_off, label, ref, nxt, n_rules, lw, goff, grule)
hf.maximize()
hf.gibbs(0, alpha=0.1, seed=4, mode=0)  # warm-up: one sweep
hf.set_weights(lw)
hf.maximize()
t0 = time.perf_counter()
hf.gibbs(sweeps - 1, alpha=0.1, seed=4, mode=0)
dt = time.perf_counter() - t0
print("exact chain: %d forests, %d nodes: %.1f ms per sweep, %.2f us per forest, %.3g node-updates/s"
      % (nf, len(label), 1e3 * dt / sweeps, 1e6 * dt / sweeps / nf, len(label) * sweeps / dt))
hf.close()
