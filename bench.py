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
    import hashlib
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_%s.json" % config)
    if not os.path.exists(path) or walk_arcs != "5,40":
        return None
    d = json.load(open(path))
    if d.get("pairs_per_gpu") != n_pairs:
        return None
    # the counters belong to one build of the kernels: after any change to kernels.hip, or under an A/B library or a
    # CARMEL_HIP_* switch, the committed figure says nothing about this run
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "carmel_amd", "csrc", "kernels.hip")
    if d.get("kernels_hip_sha16") != hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]:
        return None
    if os.environ.get("CARMEL_HIP_LIB") or any(k.startswith("CARMEL_HIP_") for k in os.environ):
        return None
    tot = 0.0
    for name, k in d["kernels"].items():
        if any(e in name for e in ESTEP_KERNELS) and k["fetch_kb_per_launch"] is not None:
            per_step = k["launches"] / d["estep_count"]
            tot += per_step * (2.0 * k["fetch_kb_per_launch"] + (k["write_kb_per_launch"] or 0.0)) * 1024.0
    return tot


F64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix rate (AMD data sheet; the microarchitecture guide lists no f64 peak)


def _replicas(value):
    """c3 / c5 at N > 1: independent replicas, one per GPU (no data-path collective: SURVEY 8e -- the sampler does not
    shard exactly, and config 3's model is 758 parameters); the job's value is the sum over the ranks"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return value, 1
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl")
    t = torch.tensor([value], dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    dist.destroy_process_group()
    return float(t[0]), world


def other_configs(args):
    """BASELINE.json configs[2] (cipher cascade, 200k lines) and configs[4] (forest-em, 5M-node packed forest) under the
    same output contract.  A step is one EM iteration (c3) / one Gibbs sweep over all forests (c5)."""
    import re
    import subprocess
    import tempfile
    import numpy as np
    from carmel_amd import synth
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.config == "amb":
        # ambiguous lattices (config 4's are 99.99 % single-path chains, so it never exercises the log-semiring sum): the
        # reference's own tagging cascade (carmel-tutorial/tagging.*: tag bigram model o tag->word lexicon, 46 states /
        # 400 994 composed arcs; a sentence's lattice is positions x candidate tags), its 1005-sentence corpus repeated
        d = tempfile.mkdtemp(prefix="amb_")
        g = lambda n: os.path.join(ROOT, "tests", "golden", n)
        reps = max(1, args.pairs // 1005) if args.pairs else 400  # 402 000 pairs: enough wavefronts to fill 256 CUs
        open(os.path.join(d, "corpus"), "w").write(open(g("tagging.data")).read() * reps)
        iters = args.steps + args.warmup
        cmd = [os.path.join(ROOT, "carmel_amd", "bin", "carmel"), "--gpu=%d" % local_rank, "--train-cascade", "-HJ", "-M", str(iters),
               "-X", "1.1", "-e", "0", os.path.join(d, "corpus"), g("tagging.fsa"), g("tagging.fst")]
        p = subprocess.run(cmd, env=dict(os.environ, CARMEL_TIMING="1", CARMEL_TRAINED_DIR=d), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, universal_newlines=True)
        if p.returncode != 0:
            sys.exit(p.stderr[-2000:])
        lat = re.search(r"timing: lattices pairs_kept=(\d+) states=(\d+) arcs=(\d+) layout=(\w+) device_bytes=(\d+) build_seconds=(\S+)", p.stderr)
        est = [float(x) for x in re.findall(r"timing: i=\d+ estimate (\S+) ms", p.stderr)][args.warmup:]
        ker = [float(x) for x in re.findall(r"estimate \S+ ms \(kernels (\S+) ms\)", p.stderr)][args.warmup:]
        mx = [float(x) for x in re.findall(r"timing: i=\d+ maximize (\S+) ms", p.stderr)][args.warmup:]
        arcs, states = float(lat.group(3)), float(lat.group(2))
        ms = (sum(est) + sum(mx)) / max(len(est), 1)
        k_ms = sum(ker) / max(len(ker), 1)
        alg = algorithmic_bytes(arcs, states)
        value, world = _replicas(arcs / (ms * 1e-3))
        out = {"metric": "arc-updates/sec (EM iterations/sec x derivation-lattice arcs swept per iteration)", "value": value,
               "unit": "arc-updates/s", "n_gpus": world, "steps": len(est), "warmup": args.warmup, "ms_per_step": ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "carmel-tutorial tagging.* x%d" % reps,
               "config": {"workload": "amb: tagging cascade (tag bigram model o lexicon, 46 states / 400994 composed arcs), the "
                                      "tutorial's 1005 sentences x %d = %d pairs, carmel --train-cascade through the front end; "
                                      "ambiguous lattices (positions x candidate tags)" % (reps, int(lat.group(1))),
                          "lattice_arcs_per_gpu": int(arcs), "lattice_states_per_gpu": int(states), "lattice_layout": lat.group(4),
                          "parallelism": "replicas x%d" % world},
               "lattice_build_s": float(lat.group(6)), "estep_ms": sum(est) / max(len(est), 1), "mstep_ms": sum(mx) / max(len(mx), 1),
               "roofline": {"bound": "hbm", "kernel": "E-step (weights to lattice order, lane / bundle sweeps, posteriors to counts), HIP "
                            "events on the trainer's stream", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg,
                            "kernel_ms": k_ms}}
        if not args.no_cpu_baseline and rank == 0:
            # the trace of the reference itself: 25.9-26.5 s per iteration on this cascade (commands.trace:5868-5889,
            # unknown hardware, no derivation caching); here the oracle with cached derivations on the 1005 sentences
            oc = [os.path.join(ROOT, "oracle", "oracle_carmel"), "--train-cascade", "-HJ", "-:", "-X", "1.1", "-e", "0", "-M"]
            tail = [g("tagging.data"), g("tagging.fsa"), g("tagging.fst")]
            dts = []
            for it in ("1", "3"):
                t0 = time.time()
                q = subprocess.run(oc + [it] + tail, env=dict(os.environ, ORACLE_TRAINED_DIR=d), stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, universal_newlines=True)
                dts.append(time.time() - t0)
            if q.returncode == 0 and dts[1] > dts[0]:
                per_iter = (dts[1] - dts[0]) / 2.0
                out["cpu_baseline"] = {"value": arcs / reps / per_iter, "unit": "arc-updates/s", "cores": 1, "kind": "port",
                                       "sample": "the oracle's command line on the 1005 sentences (one repetition), cached "
                                                 "derivations: (time of 3 iterations - time of 1) / 2 = %.2f s per iteration "
                                                 "(the reference's own trace: 26 s per iteration without caching)" % per_iter}
    elif args.config == "c3":
        d = tempfile.mkdtemp(prefix="c3_")
        lm, ch, co = synth.cipher_files(args.lines)
        for name, txt in (("lm.wfsa", lm), ("ch.fst", ch), ("corpus", co)):
            open(os.path.join(d, name), "w").write(txt)
        iters = args.steps + args.warmup
        cmd = [os.path.join(ROOT, "carmel_amd", "bin", "carmel"), "--gpu=%d" % local_rank, "--train-cascade", "--normby=NC", "-HJ", "-M", str(iters), "-X", "1.1",
               "-e", "0", os.path.join(d, "corpus"), os.path.join(d, "lm.wfsa"), os.path.join(d, "ch.fst")]
        p = subprocess.run(cmd, env=dict(os.environ, CARMEL_TIMING="1", CARMEL_TRAINED_DIR=d), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, universal_newlines=True)
        if p.returncode != 0:
            sys.exit(p.stderr[-2000:])
        lat = re.search(r"timing: lattices pairs_kept=(\d+) states=(\d+) arcs=(\d+) layout=(\w+)", p.stderr)
        est = [float(x) for x in re.findall(r"timing: i=\d+ estimate (\S+) ms", p.stderr)][args.warmup:]
        ker = [float(x) for x in re.findall(r"estimate \S+ ms \(kernels (\S+) ms\)", p.stderr)][args.warmup:]
        mx = [float(x) for x in re.findall(r"timing: i=\d+ maximize (\S+) ms", p.stderr)][args.warmup:]
        arcs = float(lat.group(3))
        ms = (sum(est) + sum(mx)) / max(len(est), 1)
        k_ms = sum(ker) / max(len(ker), 1)
        # the unrolled sweep never stores a lattice: per lattice arc it does one multiply-add forwards, and backwards a
        # multiply-add, the posterior (two multiplies) and its accumulation -- 8 f64 flops -- out of L2-resident tables.
        # In its dense form (dense.hpp: weight(s -> s', c) = A[s][s'] * B[c][s']) a position of a string is two S x S
        # vector-matrix products (forward, backward): 4 S^2 flops, S the real state count (the padding is not priced).
        dn = re.search(r"timing: dense sweep S=(\d+) padded=(\d+) .* positions=(\d+)", p.stderr)
        if dn:
            flops = 4.0 * float(dn.group(1)) ** 2 * float(dn.group(3))
            kname = ("dense_mfma_kernel (v_mfma_f64_16x16x4_f64: per string position two %sx%s vector-matrix products, priced at "
                     "4 S^2 flops per position against the f64 vector peak; tools/f64_rate.hip measures 49.7 TFLOP/s for the f64 "
                     "matrix instruction and 65.8 for v_fma_f64 on this part)" % (dn.group(1), dn.group(1)))
        else:
            flops = 8.0 * arcs
            kname = ("unrolled_sweep_kernel (f64 vector FMAs on L2-resident tables; priced against the f64 rate, 8 flops per "
                     "lattice arc -- it moves 2 B of HBM per string position)")
        value, world = _replicas(arcs / (ms * 1e-3))
        out = {"metric": "arc-updates/sec (EM iterations/sec x derivation-lattice arcs swept per iteration)", "value": value,
               "unit": "arc-updates/s", "n_gpus": world, "steps": len(est), "warmup": args.warmup, "ms_per_step": ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "c3: cipher cascade, character bigram LM (29 states, locked) o 27x27 substitution channel, "
                                      "%d lines of 30-80 symbols, carmel --train-cascade --normby=NC through the front end; "
                                      "lattices unrolled over string positions, never stored" % args.lines,
                          "lattice_arcs_per_gpu": int(arcs), "lattice_layout": lat.group(4), "parallelism": "replicas x%d" % world},
               "roofline": {"bound": "mfma", "kernel": kname,
                            "achieved": flops / (k_ms * 1e-3) / 1e12, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": flops / (k_ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS, "traffic": None, "kernel_ms": k_ms}}
        if not args.no_cpu_baseline and rank == 0:
            n = 150
            lm2, ch2, co2 = synth.cipher_files(n)
            for name, txt in (("lm2.wfsa", lm2), ("ch2.fst", ch2), ("corpus2", co2)):
                open(os.path.join(d, name), "w").write(txt)
            oc = [os.path.join(ROOT, "oracle", "oracle_carmel"), "--train-cascade", "--normby=NC", "-HJ", "-:", "-M"]
            tail = [os.path.join(d, "corpus2"), os.path.join(d, "lm2.wfsa"), os.path.join(d, "ch2.fst")]
            dts = []
            for it in ("1", "3"):  # the difference of two runs leaves the per-iteration time (composition, build cancel)
                t0 = time.time()
                q = subprocess.run(oc + [it, "-X", "1.1", "-e", "0"] + tail, env=dict(os.environ, ORACLE_TRAINED_DIR=d),
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
                dts.append(time.time() - t0)
            if q.returncode == 0 and dts[1] > dts[0]:
                per_iter = (dts[1] - dts[0]) / 2.0
                out["cpu_baseline"] = {"value": arcs * n / args.lines / per_iter, "unit": "arc-updates/s", "cores": 1, "kind": "port",
                                       "sample": "the oracle's command line on %d lines of the same model, cached derivations: "
                                                 "(time of 3 iterations - time of 1) / 2 = %.2f s per iteration" % (n, per_iter)}
    else:
        from carmel_amd.forests import HipForests
        node_off, label, ref, nxt, n_rules, goff, grule = synth.random_forests(args.forests)
        rng = np.random.default_rng(4)
        lw = np.log(rng.uniform(0.05, 1.0, n_rules))
        hf = HipForests(node_off, label, ref, nxt, n_rules, lw, goff, grule, device=local_rank)
        hf.maximize()
        hf.gibbs(max(args.warmup, 1), alpha=0.1, seed=4 + rank, mode=1)
        t0 = time.perf_counter()
        hf.gibbs(args.steps, burnin=args.steps // 4, alpha=0.1, seed=4 + rank, mode=1)
        dt = time.perf_counter() - t0
        sweeps = args.steps + 1
        n_nodes = float(len(label))
        n_sampled = float(np.mean([len(hf.sample(f)) for f in range(0, args.forests, max(1, args.forests // 2000))])) * args.forests
        ms = 1e3 * dt / sweeps
        alg = 40.0 * n_nodes + 64.0 * n_sampled  # node record, proposal gather, inside write + read; two count RMWs per sampled rule
        value, world = _replicas(n_nodes * sweeps / dt)
        out = {"metric": "forest-node updates/sec (Gibbs sweeps/sec x nodes of the packed forests; carmel's arc-updates for "
                         "forest-em)", "value": value, "unit": "node-updates/s", "n_gpus": world, "steps": sweeps,
               "warmup": max(args.warmup, 1), "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "c5: forest-em --crp on %d synthetic packed forests (%d nodes, %d parameters, alpha 0.1), "
                                      "parallel stale-count sweeps" % (args.forests, int(n_nodes), n_rules - 1),
                          "sampled_rules_per_sweep": int(n_sampled), "parallelism": "replicas x%d (independent chains)" % world},
               "sweeps_per_s": sweeps / dt,
               "roofline": {"bound": "hbm", "kernel": "one sweep = forest_proposal + forest_sample (per launch class) + forest_recount; "
                            "timed as wall time per sweep around carmel_hip_forests_gibbs (launch gaps included)",
                            "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg}}
        hf.close()
        if not args.no_cpu_baseline and rank == 0:
            from oracle import binding as ob
            from carmel_amd._capi import lib
            nf = min(args.forests, 3000)
            txt = synth.forests_to_text(node_off, label, ref, nxt, 0, nf)
            norm = "(" + " ".join("(" + " ".join(str(int(r)) for r in grule[int(goff[g]):int(goff[g + 1])]) + ")"
                                  for g in range(len(goff) - 1)) + ")"
            of = ob.OracleForests(txt, norm)
            of.set_weights(lw[:of.n_rules] if of.n_rules <= len(lw) else np.concatenate([lw, np.zeros(of.n_rules - len(lw))]))
            t0 = time.time()
            of.gibbs(lambda i, b, st: lib.carmel_hip_gibbs_uniform(4, i, b, st), 9, burnin=2, alpha=0.1)
            cdt = time.time() - t0
            out["cpu_baseline"] = {"value": float(of.n_nodes) * 10 / cdt, "unit": "node-updates/s", "cores": 1, "kind": "port",
                                   "sample": "the first %d forests (%d nodes), same parameters; 10 exact (sequential) sweeps of the "
                                             "scalar oracle, uniforms through a callback" % (nf, of.n_nodes)}
    if rank == 0:
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20; c5: 1000 sweeps, the length BASELINE.json's config names")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=["c2", "c4", "toy", "c3", "c5", "amb"])
    ap.add_argument("--lines", type=int, default=200000, help="c3: corpus lines")
    ap.add_argument("--forests", type=int, default=100000, help="c5: forests")
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU (default: the config's)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --pairs per GPU; strong: --pairs in total, sharded over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=200000)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0: host cores, at most 64)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--walk-arcs", default="5,40", help="min,max arcs of the random walks (SURVEY 8d: 5,40; other values are experiments)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 1000 if args.config == "c5" else 20

    if args.config in ("c3", "c5", "amb"):
        return other_configs(args)
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
    # CARMEL_HIP_COMM=host (tests on a one-GPU box): every rank on GPU 0, sums staged through shared memory, gloo for control
    one_device = os.environ.get("CARMEL_HIP_COMM") == "host"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctl = "cpu" if one_device else "cuda"  # where the few control values of the collectives below live

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
    comm, ext_counts, exchange = None, None, "none"
    if world > 1:
        # the library's own exchange (RCCL on the trainer's stream).  Should the communicator not come up on every rank,
        # all ranks fall back together to torch.distributed on the trainer's device count buffer (round 1's path: two host
        # synchronisations per step) -- the line then says so in config.parallelism
        from carmel_amd.trainer import HipComm
        ok = 0 if os.environ.get("BENCH_FORCE_TORCH_EXCHANGE") else 1  # (test hook for the fallback below)
        try:
            ids = [HipComm.unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            ids, ok = [None], 0
            sys.stderr.write("bench.py: library communicator unavailable (%s)\n" % e)
        dist.broadcast_object_list(ids, src=0)
        if ids[0] is None:
            ok = 0
        if ok:
            try:
                comm = HipComm(local_rank, rank, world, ids[0])
            except Exception as e:  # noqa: BLE001
                ok = 0
                sys.stderr.write("bench.py: rank %d could not join the library communicator (%s)\n" % (rank, e))
        flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:  # every rank joined: one exchange end to end before anything is timed
            try:
                fb.estimate_async()
                fb.allreduce_counts(comm)
                fb.synchronize()
            except Exception as e:  # noqa: BLE001
                ok = 0
                sys.stderr.write("bench.py: rank %d: the library's all-reduce failed (%s)\n" % (rank, e))
            flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            if comm is not None:
                comm.close()
            comm = None
            ext_counts = torch.zeros(int(w.n_arcs) + 4, dtype=torch.float64, device="cuda")
            fb.use_external_counts(ext_counts.data_ptr())
            exchange = "torch.distributed all-reduce of %d f64 counts per iteration (library communicator unavailable), two host synchronisations per step" % (w.n_arcs + 4)
        else:
            exchange = "RCCL all-reduce of %d f64 counts per iteration on the trainer's stream" % (w.n_arcs + 4)

    def step():
        fb.estimate_async()
        if comm is not None:
            fb.allreduce_counts(comm)  # stream-ordered: count pass -> all-reduce -> M-step
        elif ext_counts is not None:
            fb.synchronize()
            dist.all_reduce(ext_counts)
            torch.cuda.synchronize()
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
    t = torch.tensor([dt, float(ls.kept_arcs), float(ls.kept_states)], dtype=torch.float64, device=ctl)
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
                       "bundles_per_gpu": int(ls.n_bundles), "parallelism": "corpus-sharded x%d, %s" % (world, exchange)},
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
                         "traffic_source": "profiles/pmc_traffic_%s.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "command (tools/pmc_traffic.sh), committed; null when kernels.hip changed since, the "
                                           "workload differs or a CARMEL_HIP_* switch is set -- not measured by this run" % args.config,
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
