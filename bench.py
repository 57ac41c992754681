#!/usr/bin/env python3
"""bench.py — EM arc-weight training throughput on MI355X (one rank per GPU, RCCL all-reduce of arc counts).

One "step" = one EM iteration of the hot path: forward_backward::estimate (forward sweep, backward sweep,
expected-count accumulation over every derivation lattice of this rank's corpus shard) + the all-reduce of the
per-arc count vector across ranks (N > 1) + forward_backward::maximize (normalisation of all arc weights).
Inputs (transducer, lattices) are resident in HBM before the timed region.

Workload (config.workload): by default BASELINE.json configs[3] shape — synthetic 1M-state / 10M-arc transducer,
1M training pairs PER GPU (weak scaling: the corpus grows with N, the model — and therefore the 80 MB count
all-reduce — does not).  `--scaling strong` keeps the corpus at 1M pairs in total and gives every rank 1/N of it.
`--config c2` selects configs[1] (100k states / 2M arcs / 50k pairs).  For N > 1 the all-reduce is the library's own
(carmel_hip_allreduce_counts: RCCL enqueued on the trainer's stream between the count pass and the M-step, no host
synchronisation inside a step); torch.distributed only carries the communicator id, the barrier and the max-over-ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(lattice_arcs, lattice_states):
    """48 B per lattice arc (8 B arc record re-read in each of the three passes, one 8 B
    gather of logw[arc], one 16 B read-modify-write of counts[arc]) + 16 B per lattice state (alpha and beta
    written once).  The M-step's 16 B per WFST arc belongs to the M-step kernels, not to the sweep kernel."""
    return 48.0 * lattice_arcs + 16.0 * lattice_states


ESTEP_KERNELS = ("trans_w_bucket_kernel", "trans_w_tile_kernel", "sweep_lane_kernel", "sweep_bundle_kernel",
                 "sweep_serial_kernel", "trans_c_tile_kernel", "trans_c_bucket_kernel", "zero_list_kernel",
                 "scalars_partial_kernel", "scalars_final_kernel", "count_reduce_kernel", "count_reduce_hot_kernel")


def pmc_traffic(config, walk_arcs, n_pairs):
    """HBM-side bytes per E-step from the committed rocprofv3 PMC passes of THIS command (tools/pmc_traffic.sh: FETCH_SIZE
    and WRITE_SIZE in separate runs, summarised per kernel by tools/pmc_summary.py into profiles/).  Correction per
    MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies each 128-byte request at 64 B, so it is doubled (calibrated here on
    known byte counts, profiles/r1_pmc_calibration.txt: streaming reads of 4/8/16 B per lane all report exactly half;
    WRITE_SIZE is exact).  None when no profile of this workload is committed."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_%s.json" % config)
    if not os.path.exists(path) or walk_arcs != "5,40":
        return None
    d = json.load(open(path))
    if d.get("pairs_per_gpu") != n_pairs:
        return None
    tot = 0.0
    for name, k in d["kernels"].items():
        if any(e in name for e in ESTEP_KERNELS) and k["fetch_kb_per_launch"] is not None:
            per_step = k["launches"] / d["estep_count"]
            tot += per_step * (2.0 * k["fetch_kb_per_launch"] + (k["write_kb_per_launch"] or 0.0)) * 1024.0
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=["c2", "c4", "toy"])
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU (default: the config's)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --pairs per GPU; strong: --pairs in total, sharded over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=200000)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0: host cores, at most 64)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--walk-arcs", default="5,40", help="min,max arcs of the random walks (SURVEY 8d: 5,40; other values are experiments)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched through torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the EM hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from carmel_amd import synth
    from carmel_amd.trainer import HipForwardBackward

    n_states, deg, npairs, seed = synth.CONFIGS[args.config]
    if args.pairs:
        npairs = args.pairs
    t0 = time.time()
    w = synth.random_wfst(n_states, deg, seed=seed)  # same model on every rank
    lo, hi = (int(v) for v in args.walk_arcs.split(","))
    if args.scaling == "strong" and world > 1:  # one corpus, every rank takes its contiguous block of pairs
        c = synth.random_walk_corpus(w, npairs, min_arcs=lo, max_arcs=hi, seed=seed, out_degree=deg).shard(rank, world)
    else:
        c = synth.random_walk_corpus(w, npairs, min_arcs=lo, max_arcs=hi, seed=seed + 7919 * rank, out_degree=deg)  # this rank's shard
    t_gen = time.time() - t0
    fb = HipForwardBackward(w, c, device=local_rank, host_threads=args.host_threads)
    ls = fb.lattice_stats
    comm = None
    if world > 1:
        from carmel_amd.trainer import HipComm
        ids = [HipComm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = HipComm(local_rank, rank, world, ids[0])

    def step():
        fb.estimate_async()
        if comm is not None:
            fb.allreduce_counts(comm)  # stream-ordered: count pass -> all-reduce -> M-step
        return fb.maximize(1.0)

    def fence():
        fb.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(fb.last_kernel_ms())
    fence()
    dt = time.perf_counter() - t0
    lp, wlp, n_swept = fb.read_scalars()
    t = torch.tensor([dt, float(ls.kept_arcs), float(ls.kept_states)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        total_arcs, total_states = float(tsum[1]), float(tsum[2])
    else:
        total_arcs, total_states = float(ls.kept_arcs), float(ls.kept_states)
    iters_per_s = args.steps / dt
    value = iters_per_s * total_arcs

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        alg = algorithmic_bytes(float(ls.kept_arcs), float(ls.kept_states))
        achieved = alg / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "arc-updates/sec (EM iterations/sec x derivation-lattice arcs swept per iteration)",
            "value": value, "unit": "arc-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: synthetic %d-state / %d-arc WFST, %d training pairs %s (random walks of "
                                   "%s arcs), conditional normalisation, cached lattices" %
                                   (args.config, w.n_states, w.n_arcs, npairs, "in total, sharded" if args.scaling == "strong"
                                    and world > 1 else "per GPU", args.walk_arcs.replace(",", "-")),
                       "pairs_per_gpu": c.n_pairs, "wfst_arcs": int(w.n_arcs), "wfst_states": int(w.n_states),
                       "lattice_arcs_per_gpu": int(ls.kept_arcs), "lattice_states_per_gpu": int(ls.kept_states),
                       "bundles_per_gpu": int(ls.n_bundles), "parallelism": "corpus-sharded x%d, RCCL all-reduce of %d f64 "
                       "counts per iteration on the trainer's stream" % (world, w.n_arcs + 4)},
            "iters_per_s": iters_per_s,
            "wfst_arcs_x_iters_per_s": iters_per_s * w.n_arcs,
            "ln_corpus_prob_last": lp,
            "lattice_build_s": ls.build_seconds, "synth_gen_s": t_gen,
            "roofline": {"bound": "hbm", "kernel": "E-step = trans_w_bucket + trans_w_tile (weights to lattice order) + "
                         "sweep_lane_kernel + trans_c_tile + trans_c_bucket (posteriors to per-arc counts), timed "
                         "together with HIP events on the trainer's stream",
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.config, args.walk_arcs, c.n_pairs),
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": k_ms},
        }
        if not args.no_cpu_baseline:
            from oracle import binding as ob  # CPU restatement of the reference: the checker, timed as the baseline
            # One EM iteration of the reference costs a FIXED part (clear the count table, maximize: O(|WFST arcs|)) plus
            # a part per lattice arc (the per-pair sweeps).  Timing a sample of the corpus and dividing by its lattice
            # arcs would charge the fixed part to the sample; instead both parts are measured on a bounded sample (the
            # E-step on a quarter and on all of its cached lattices, maximize on its own) and the iteration time of the
            # FULL rank-0 shard follows as fixed + per_arc * lattice arcs of the shard.
            ns = min(args.cpu_sample_pairs, c.n_pairs)
            cs = c.shard(0, max(1, c.n_pairs // ns)) if ns < c.n_pairs else c
            nthreads = args.cpu_threads or min(64, len(os.sched_getaffinity(0)))
            ow, oc = ob.OracleWfst.from_arrays(w), ob.OracleCorpus.from_arrays(cs)
            r = ob.bench_em_fit(ow, oc, iters=2, threads=nthreads)
            full_arcs = float(ls.kept_arcs)

            def leg(d):
                sec = d["fixed_sec"] + d["sec_per_arc"] * full_arcs
                return {"value": full_arcs / sec, "sec_per_iter_full_shard": sec, "fixed_sec": d["fixed_sec"],
                        "sec_per_lattice_arc": d["sec_per_arc"], "estep_sec_sample": d["estep_all"],
                        "estep_sec_quarter_sample": d["estep_quarter"], "maximize_sec": d["maximize"]}
            one = leg(r["serial"])
            out["cpu_baseline"] = dict(one, unit="arc-updates/s", cores=1, kind="port",
                sample="first %d pairs of rank 0's shard (%d lattice arcs), same transducer; scalar oracle over cached "
                       "lattices, lattice build (%.1f s) excluded; value = lattice arcs of the full shard / (fixed_sec + "
                       "sec_per_lattice_arc * those arcs), the two terms measured apart (E-step on a quarter and on all "
                       "of the sample, maximize on its own; 2 repetitions each)" %
                       (cs.n_pairs, int(r["arcs_all"]), r["build_sec"]))
            if r["threaded"]:
                out["cpu_baseline"]["all_cores"] = dict(leg(r["threaded"]), unit="arc-updates/s", cores=nthreads, kind="port",
                    note="the same oracle with OpenMP over pairs (atomic adds into one linear count table) and over states "
                         "in maximize; reference carmel itself is single-threaded")
        print(json.dumps(out))
    fence()
    fb.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
