#!/usr/bin/env python3
"""bench_forest.py — BASELINE.json configs[4]: forest-em on a synthetic packed forest (SURVEY.md section 8d C5:
10^5 forests x ~50 nodes, 5*10^5 parameters, alpha 0.1).  Reports, on one MI355X:
  * EM: ms per iteration (estimate = inside + outside + counts, then maximize)
  * Gibbs (--crp): sweeps per second of the parallel stale-count schedule, and forest-nodes resampled per second
The driver's contract lives in bench.py (EM over WFST lattices); this script only adds the config-5 numbers that
DESIGN.md quotes.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--forests", type=int, default=100000)
    ap.add_argument("--rules", type=int, default=500000)
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--em-iters", type=int, default=10)
    args = ap.parse_args()
    import numpy as np
    from carmel_amd import synth
    from carmel_amd.forests import HipForests
    t0 = time.time()
    node_off, label, ref, nxt, n_rules, goff, grule = synth.random_forests(args.forests, n_rules=args.rules)
    t_gen = time.time() - t0
    rng = np.random.default_rng(4)
    lw = np.log(rng.uniform(0.05, 1.0, n_rules))
    hf = HipForests(node_off, label, ref, nxt, n_rules, lw, goff, grule)
    hf.maximize()  # start from normalised weights
    for _ in range(2):
        hf.estimate()
        hf.maximize()
    t0 = time.perf_counter()
    for _ in range(args.em_iters):
        avg = hf.estimate()
        hf.maximize()
    em_ms = 1e3 * (time.perf_counter() - t0) / args.em_iters
    hf.gibbs(3, alpha=0.1, seed=4, mode=1)  # warm-up
    t0 = time.perf_counter()
    hf.gibbs(args.sweeps, burnin=args.sweeps // 4, alpha=0.1, seed=4, mode=1)
    dt = time.perf_counter() - t0
    n_nodes = int(len(label))
    print(json.dumps({
        "workload": "c5: %d forests / %d nodes / %d parameters (synthetic, seed 4)" % (args.forests, n_nodes, args.rules),
        "em_ms_per_iteration": em_ms, "em_avg_logprob": avg,
        "gibbs_mode": "parallel stale-count sweep (mode 1)", "gibbs_sweeps": args.sweeps + 1,
        "gibbs_sweeps_per_s": (args.sweeps + 1) / dt, "gibbs_forest_nodes_per_s": (args.sweeps + 1) * n_nodes / dt,
        "gibbs_last_cheap_logprob": float(hf.iter_cheap_logprob[-1]), "synth_gen_s": t_gen}))
    hf.close()


if __name__ == "__main__":
    main()
