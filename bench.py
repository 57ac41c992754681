#!/usr/bin/env python3
"""bench.py — EM arc-weight training throughput on MI355X (one rank per GPU, RCCL exchange of arc counts).

One "step" = one EM iteration of the hot path: forward_backward::estimate (forward sweep, backward sweep,
expected-count accumulation over every derivation lattice of this rank's corpus shard) + the exchange of the
per-arc count vector across ranks (N > 1) + forward_backward::maximize (normalisation of all arc weights).
Inputs (transducer, lattices) are resident in HBM before the timed region.

Headline workload (config.workload): BASELINE.json configs[3] shape — synthetic 1M-state / 10M-arc transducer,
1M training pairs PER GPU (weak scaling: the corpus grows with N, the model — and therefore the exchanged count
vector — does not).  `--scaling strong` keeps the corpus at 1M pairs in total and gives every rank 1/N of it.
`--config NAME` makes another workload the headline.

At N = 1 the same process then runs the other workloads one after another and attaches each to the line as
`"secondary": {name: {...}}` with its own `roofline`, `cpu_baseline` and `kernel_ms` (`--no-secondary` skips them):
  c4a   config 4's sizes on a clustered transducer whose lattices are ambiguous (3 in-arcs per lattice state)
  amb   the tutorial's tagging cascade x400 through the carmel front end
  c2    BASELINE.json configs[1]
  long  5 000 pairs with lattices of 320-4 800 states (the one-lattice-per-wavefront path)
  c3    configs[2]: the cipher cascade, 200 000 lines, through the front end
  c5    configs[4]: forest-em --crp, parallel stale-count sweeps AND the exact (sequential) chain
  crp   carmel --crp on the tutorial's tagging cascade: the exact chain and the stale-count parallel sweep

The line printed is COMPACT (the driver keeps the last 2 000 characters): the headline in full, every secondary as
{ms_per_step, kernel_ms, frac, traffic, cpu, parity, value}; the complete objects go to --full-out (a file).

The `cpu_baseline` leg of every synthetic EM workload also CHECKS the GPU: the oracle's per-pair ln p and per-arc
counts on the timed sample are compared with a GPU trainer's on the same pairs (rtol 1e-7); `parity_checked_pairs`
says how many pairs that was, and a mismatch fails the run.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
F64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix rate (AMD data sheet; the microarchitecture guide lists no f64 peak)
METRIC = "arc-updates/sec (EM iterations/sec x derivation-lattice arcs swept per iteration)"


def algorithmic_bytes(lattice_arcs, lattice_states):
    """48 B per lattice arc (8 B arc record re-read in each of the three passes, one 8 B
    gather of logw[arc], one 16 B read-modify-write of counts[arc]) + 16 B per lattice state (alpha and beta
    written once).  The M-step's 16 B per WFST arc belongs to the M-step kernels, not to the sweep kernel."""
    return 48.0 * lattice_arcs + 16.0 * lattice_states


ESTEP_KERNELS = ("trans_w_bucket_kernel", "trans_w_tile_kernel", "trans_w_tile_small_kernel", "trans_c_tile_small_kernel", "tile_sweep_kernel", "sweep_lane_kernel", "sweep_bundle_kernel", "sweep_wave_kernel",
                 "sweep_serial_kernel", "trans_c_tile_kernel", "trans_c_bucket_kernel", "zero_list_kernel",
                 "scalars_partial_kernel", "scalars_final_kernel", "count_reduce_kernel", "count_reduce_hot_kernel")


def pmc_traffic(config, walk_arcs, n_pairs):
    """HBM-side bytes per E-step from the committed rocprofv3 PMC passes of THIS command (tools/pmc_traffic.sh: FETCH_SIZE
    and WRITE_SIZE in separate runs, summarised per kernel by tools/pmc_summary.py into profiles/).  Correction per
    MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies each 128-byte request at 64 B, so it is doubled (calibrated here on
    known byte counts, profiles/r1_pmc_calibration.txt: streaming reads of 4/8/16 B per lane all report exactly half;
    WRITE_SIZE is exact).  None when no profile of this workload is committed."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % config)
    if not os.path.exists(path) or walk_arcs:
        return None
    d = json.load(open(path))
    if d.get("pairs_per_gpu") != n_pairs:
        return None
    # the counters belong to one build of the kernels: after any change to kernels.hip, or under an A/B library or a
    # CARMEL_HIP_* switch, the committed figure says nothing about this run
    csrc = os.path.join(ROOT, "carmel_amd", "csrc")  # (the E-step's device code: kernels.hip, tile_sweep.hip, sweep_math.hpp)
    sha = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in ("kernels.hip", "tile_sweep.hip", "sweep_math.hpp")))
    if d.get("kernels_hip_sha16") != sha.hexdigest()[:16]:
        return None
    if os.environ.get("CARMEL_HIP_LIB") or any(k.startswith("CARMEL_HIP_") for k in os.environ):
        return None
    tot = 0.0
    for name, k in d["kernels"].items():
        if any(e in name for e in ESTEP_KERNELS) and k["fetch_kb_per_launch"] is not None:
            per_step = k["launches"] / d["estep_count"]
            tot += per_step * (2.0 * k["fetch_kb_per_launch"] + (k["write_kb_per_launch"] or 0.0)) * 1024.0
    return tot


SWEEP_KERNELS = ("forest_proposal_kernel", "forest_sample_kernel", "forest_sample_multi_kernel", "forest_recount_kernel",
                 "forest_commit_kernel", "forest_gibbs_kernel")


def pmc_traffic_c5(n_forests):
    """HBM-side bytes per parallel sweep of config 5 from the committed PMC passes (tools/pmc_traffic.sh c5 -> profiles/
    pmc_traffic_c5.json), FETCH_SIZE doubled as for the EM workloads; None unless the profile is of this forest.hip and size"""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_traffic_c5.json")
    if not os.path.exists(path) or any(k.startswith("CARMEL_HIP_") for k in os.environ):
        return None
    d = json.load(open(path))
    src = os.path.join(ROOT, "carmel_amd", "csrc", "forest.hip")
    if d.get("forests") != n_forests or not d.get("sweep_count") or \
            d.get("forest_hip_sha16") != hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]:
        return None
    tot = 0.0
    for name, k in d["kernels"].items():
        if any(e in name for e in SWEEP_KERNELS) and k["fetch_kb_per_launch"] is not None:
            tot += k["launches"] / d["sweep_count"] * (2.0 * k["fetch_kb_per_launch"] + (k["write_kb_per_launch"] or 0.0)) * 1024.0
    return tot


CRP_KERNELS = ("gibbs_lane_kernel", "gibbs_lane_recount_kernel", "gibbs_normsum_kernel", "gibbs_commit_kernel", "gibbs_exact_wave_kernel",
               "gibbs_reg_wave_kernel", "gibbs_recount_tables_kernel")


def pmc_traffic_crp(n_blocks):
    """HBM-side bytes per parallel sweep of `--config crp` from the committed PMC passes (tools/gibbs_profile.sh -> profiles/
    pmc_traffic_crp.json), FETCH_SIZE doubled as for the EM workloads; None unless the profile is of these sources and this corpus"""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_traffic_crp.json")
    if not os.path.exists(path) or any(k.startswith("CARMEL_HIP_") for k in os.environ):
        return None
    d = json.load(open(path))
    csrc = os.path.join(ROOT, "carmel_amd", "csrc")
    sha = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in ("gibbs_lane.hip", "gibbs_exact.hip", "gibbs.hip"))).hexdigest()[:16]
    if d.get("blocks") != n_blocks or not d.get("sweep_count") or d.get("gibbs_sha16") != sha:
        return None
    tot = 0.0
    for name, k in d["kernels"].items():
        if any(e in name for e in CRP_KERNELS) and k["fetch_kb_per_launch"] is not None:
            tot += k["launches"] / d["sweep_count"] * (2.0 * k["fetch_kb_per_launch"] + (k["write_kb_per_launch"] or 0.0)) * 1024.0
    return tot


def _replicas(value):
    """c3 / c5 / amb as the headline at N > 1: independent replicas, one per GPU (no data-path collective: SURVEY 8e --
    the sampler does not shard exactly, and config 3's model is 758 parameters); the job's value is the sum over the ranks"""
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


WORKLOAD_TEXT = {
    "c2": "synthetic %d-state / %d-arc WFST, %d training pairs %s (random walks of %s arcs), conditional normalisation, cached lattices",
    "c4": "synthetic %d-state / %d-arc WFST, %d training pairs %s (random walks of %s arcs), conditional normalisation, cached lattices",
    "toy": "synthetic %d-state / %d-arc WFST, %d training pairs %s (random walks of %s arcs), conditional normalisation, cached lattices",
    "c4a": "config 4's sizes with AMBIGUOUS lattices: synthetic clustered %d-state / %d-arc WFST (clusters of 3 member states; a "
           "string fixes the cluster sequence, not the members: a lattice is positions x 3 states with 9 arcs between neighbouring "
           "positions, 3 in-arcs per state, every state a real log-semiring sum), %d training pairs %s (walks of %s moves), "
           "conditional normalisation, cached lattices",
    "long": "few LONG lattices: synthetic clustered %d-state / %d-arc WFST (clusters of 8 members: 64 arcs between neighbouring "
            "positions), %d training pairs %s (walks of %s moves: lattices of 320-4800 states), conditional normalisation, "
            "cached lattices",
    "toya": "small clustered %d-state / %d-arc WFST, %d training pairs %s (walks of %s moves)",
    "mix": "ONE corpus of all three lattice classes: synthetic %d-state / %d-arc WFST of three regions (single paths like config 4's; "
           "c4a's clusters of 3; `long`'s clusters of 8), %d training pairs %s -- 90 %% short single paths, 9 %% ambiguous windowed "
           "lattices, 1 %% long wide ones (walks of %s moves), interleaved in corpus order; conditional normalisation, cached lattices",
    "toymix": "small mixed-class %d-state / %d-arc WFST, %d training pairs %s (walks of %s moves)",
}


def cpu_leg(ob, w, cs, full_arcs, nthreads, device, parity=True):
    """cpu_baseline of a synthetic EM workload: the oracle (CPU restatement of the reference) timed on the sample `cs` of
    the corpus -- and, on that same sample, the checker of the GPU: a fresh trainer's per-pair ln p and per-arc counts
    against the oracle's.

    One EM iteration of the reference costs a FIXED part (clear the count table, maximize: O(|WFST arcs|)) plus a part
    per lattice arc (the per-pair sweeps).  Timing a sample of the corpus and dividing by its lattice arcs would charge
    the fixed part to the sample; instead both parts are measured on the sample (the E-step on a quarter and on all of
    its cached lattices, maximize on its own) and the iteration time of the FULL rank-0 shard follows as
    fixed + per_arc * lattice arcs of the shard."""
    import numpy as np
    from carmel_amd.trainer import HipForwardBackward
    ow, oc = ob.OracleWfst.from_arrays(w), ob.OracleCorpus.from_arrays(cs)
    r = ob.bench_em_fit(ow, oc, iters=2, threads=nthreads, check=parity)

    def leg(d):
        sec = d["fixed_sec"] + d["sec_per_arc"] * full_arcs
        return {"value": full_arcs / sec, "sec_per_iter_full_shard": sec, "fixed_sec": d["fixed_sec"],
                # the E-step alone -- the region `roofline` describes (maximize and the clearing of the count table left out)
                "estep_value": 1.0 / d["sec_per_arc"] if d["sec_per_arc"] > 0 else None,
                "sec_per_lattice_arc": d["sec_per_arc"], "estep_sec_sample": d["estep_all"],
                "estep_sec_quarter_sample": d["estep_quarter"], "maximize_sec": d["maximize"]}
    out = dict(leg(r["serial"]), unit="arc-updates/s", cores=1, kind="port",
               sample="first %d pairs of rank 0's shard (%d lattice arcs), same transducer; scalar oracle over cached "
                      "lattices, lattice build (%.1f s) excluded; value = lattice arcs of the full shard / (fixed_sec + "
                      "sec_per_lattice_arc * those arcs), the two terms measured apart (E-step on a quarter and on all "
                      "of the sample, maximize on its own; 2 repetitions each)" % (cs.n_pairs, int(r["arcs_all"]), r["build_sec"]))
    if r["threaded"]:
        out["all_cores"] = dict(leg(r["threaded"]), unit="arc-updates/s", cores=nthreads, kind="port",
                                note="the same oracle with OpenMP over pairs (atomic adds into one linear count table) and over "
                                     "states in maximize; reference carmel itself is single-threaded")
    checked = 0
    if parity:
        # the GPU on the very pairs the oracle was timed on, from the same initial weights (both normalise first, train.cc:509)
        fb2 = HipForwardBackward(w, cs, device=device)
        fb2.estimate(per_pair=True)
        gc = fb2.counts()
        keep = fb2.has_deriv.astype(bool)
        glp = fb2.pair_logprob[keep]
        fb2.close()
        olp = r["pair_logprob"][:int(keep.sum())]
        oc_lin = np.exp(r["counts_ln"])
        np.testing.assert_allclose(glp, olp, rtol=1e-7, atol=1e-9, err_msg="bench.py: GPU per-pair ln p differs from the oracle's on the cpu_baseline sample")
        np.testing.assert_allclose(gc, oc_lin, rtol=1e-7, atol=1e-12, err_msg="bench.py: GPU expected counts differ from the oracle's on the cpu_baseline sample")
        checked = int(keep.sum())
        out["parity"] = {"checked_pairs": checked, "checked_arcs": int(len(gc)), "rtol": 1e-7,
                         "max_abs_err_ln_p": float(np.max(np.abs(glp - olp))) if checked else 0.0,
                         "max_abs_err_counts": float(np.max(np.abs(gc - oc_lin)))}
    return out, checked


def run_em(name, args, dist=None, rank=0, world=1, local_rank=0, one_device=False, headline=True):
    """one synthetic EM workload (c2 / c4 / c4a / long / toy); returns the result object (rank 0) or None"""
    import numpy as np
    import torch
    import carmel_amd
    carmel_amd.options_from_env()  # (bench.py is a front end: CARMEL_HIP_<KEY> in its environment are the library's options)
    from carmel_amd import synth
    from carmel_amd.trainer import HipForwardBackward
    ctl = "cpu" if one_device else "cuda"  # where the few control values of the collectives below live
    walk = tuple(int(v) for v in args.walk_arcs.split(",")) if (args.walk_arcs and headline) else None
    t0 = time.time()
    npairs = args.pairs if (args.pairs and headline) else None
    if args.scaling == "strong" and world > 1:  # one corpus, every rank takes its contiguous block of pairs
        w, c = synth.make_config(name, n_pairs=npairs, walk=walk)
        c = c.shard(rank, world)
    else:
        w, c = synth.make_config(name, n_pairs=npairs, rank=rank, walk=walk)  # this rank's shard, same model on every rank
    t_gen = time.time() - t0
    fb = HipForwardBackward(w, c, device=local_rank, host_threads=args.host_threads)
    ls = fb.lattice_stats
    comm, ext_counts, exchange, xinfo = None, None, "none", None
    loopback = world == 1 and headline and not args.no_exchange_loopback and name in ("c4", "c2", "toy")
    if world > 1 or loopback:
        # the library's own exchange (csrc/exchange.cpp: RCCL -- or the transport of --comm-plugin -- on the communicator's
        # stream beside the count pass).  Should the communicator not come up on every rank, all ranks fall back together
        # to torch.distributed on the trainer's device count buffer (round 1's path: two host synchronisations per step)
        # -- the line then says so in config.parallelism
        from carmel_amd.trainer import HipComm
        ok = 0 if os.environ.get("BENCH_FORCE_TORCH_EXCHANGE") else 1  # (test hook for the fallback below)
        ids = [None]
        try:
            if rank == 0:
                ids = ["%s_%d_%d" % (os.path.basename(args.comm_plugin), os.getpid(), int(time.time()))] if args.comm_plugin else [HipComm.unique_id()]
        except Exception as e:  # noqa: BLE001
            ids, ok = [None], 0
            sys.stderr.write("bench.py: library communicator unavailable (%s)\n" % e)
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        if ids[0] is None:
            ok = 0
        if ok:
            try:
                comm = (HipComm.custom(args.comm_plugin, ids[0], local_rank, rank, world) if args.comm_plugin
                        else HipComm(local_rank, rank, world, ids[0]))
            except Exception as e:  # noqa: BLE001
                ok = 0
                sys.stderr.write("bench.py: rank %d could not join the library communicator (%s)\n" % (rank, e))
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag[0])
        if ok == 1 and not loopback:  # every rank joined: plan, and one exchange end to end before anything is timed
            form = {"sharded": "auto"}.get(args.exchange, args.exchange)
            if form == "auto" and world > 1:
                # the direct form rests on the transport's point-to-point groups: one such group between all ranks, checked
                # (carmel_hip_comm_selftest); a transport that fails it on any rank keeps the ring collectives
                try:
                    comm.selftest()
                    p2p = 1
                except Exception as e:  # noqa: BLE001
                    p2p = 0
                    sys.stderr.write("bench.py: rank %d: point-to-point self-test failed (%s)\n" % (rank, e))
                flag = torch.tensor([p2p], dtype=torch.int32, device=ctl)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag[0]) == 0:
                    form = "collectives"
            try:
                xinfo = fb.exchange_plan(comm, args.exchange_chunks, form=form)
                fb.estimate_async()
                fb.allreduce_counts(comm)
            except Exception as e:  # noqa: BLE001
                ok = 0
                sys.stderr.write("bench.py: rank %d: the library's exchange failed at enqueue (%s)\n" % (rank, e))
            # agree BEFORE anybody waits on a stream: a rank whose enqueue failed never joins the collective, and its
            # peers would wait for it forever (round-2 advisor finding)
            flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag[0])
            if ok == 1:
                fb.maximize(1.0)
                fb.synchronize()
            elif comm is not None:
                comm.abort()  # (drops the plan it was given: the trainer runs unplanned from here on)
                comm = None
        if ok == 0 and not loopback:
            if comm is not None:
                comm.close()
            comm = None
            ext_counts = torch.zeros(int(w.n_arcs) + 4, dtype=torch.float64, device="cuda")
            fb.use_external_counts(ext_counts.data_ptr())
            exchange = "torch.distributed all-reduce of %d f64 counts per iteration (library communicator unavailable), two host synchronisations per step" % (w.n_arcs + 4)
        elif not loopback:
            if xinfo["form"] == "direct":
                exchange = ("%s: the counts straight to their owners in %d arc-range chunks beside the count pass (one group of sends / "
                            "receives per chunk, %.1f MB out per rank, summed in rank order), M-step on 1/%d of the arcs, the weights "
                            "straight to every peer chunk by chunk into the next count pass (%.1f MB); no small collective"
                            % (comm.transport, xinfo["n_chunks"], xinfo["bytes_reduce_scatter"] / 1e6, world, xinfo["bytes_all_gather"] / 1e6))
            elif xinfo["sharded"]:
                exchange = ("%s: reduce-scatter of the counts in %d arc-range chunks beside the count pass (%.1f MB out per rank), M-step on "
                            "1/%d of the arcs, all-gather of the weights chunk by chunk into the next count pass (%.1f MB), one small "
                            "all-reduce (%.0f KB)" % (comm.transport, xinfo["n_chunks"], xinfo["bytes_reduce_scatter"] / 1e6, world,
                                                      xinfo["bytes_all_gather"] / 1e6, xinfo["bytes_all_reduce"] / 1e3))
            else:
                exchange = "%s: all-reduce of %d f64 counts per iteration on the trainer's stream, replicated M-step" % (comm.transport, w.n_arcs + 4)

    def step():
        fb.estimate_async()
        if comm is not None and not loopback:
            fb.allreduce_counts(comm)  # stream-ordered: count pass (-> reduce-scatters beside it) -> M-step (-> all-gathers)
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

    steps = args.steps if headline else args.secondary_steps
    warmup = args.warmup if headline else min(args.warmup, 3)
    for _ in range(warmup):
        step()
    fence()
    kernel_ms, step_end = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()  # (ends when the host has the M-step's largest change: every step's end is a point on the host's clock)
        step_end.append(time.perf_counter())
        kernel_ms.append(fb.last_kernel_ms())
    fence()
    dt = time.perf_counter() - t0
    step_ms = 1e3 * np.diff(np.array([t0] + step_end))
    lp, wlp, n_swept = fb.read_scalars()
    # the exchange on its own, and what of it the step does not hide: exchange_ms = one iteration's collectives back to
    # back on the communicator's stream (carmel_hip_exchange_measure); exposed_exchange_ms = the step time with the exchange
    # minus the step time of the same trainer without it (a few extra steps, outside the timed region).  At N = 1 both
    # are measured over a loopback communicator of one rank (the collectives still run; they move nothing between GPUs).
    xch = None
    if comm is not None and ext_counts is None:
        def timed(fn, n):
            fence()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            fence()
            return 1e3 * (time.perf_counter() - t1) / n
        n_x = max(3, min(10, steps))
        if loopback:
            xinfo = fb.exchange_plan(comm, args.exchange_chunks, form={"sharded": "auto"}.get(args.exchange, args.exchange))

        def with_x():
            fb.estimate_async()
            fb.allreduce_counts(comm)
            fb.maximize(1.0)

        def without_x():
            fb.estimate_async()
            fb.maximize(1.0)
        with_x()
        ms_with = timed(with_x, n_x)
        x_ms = fb.exchange_measure(5)
        fb.exchange_clear()
        without_x()
        ms_without = timed(without_x, n_x)
        tt = torch.tensor([ms_with, ms_without, x_ms], dtype=torch.float64, device=ctl)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        xch = {"world": world, "loopback": bool(loopback), "sharded": xinfo["sharded"], "form": xinfo["form"], "n_chunks": xinfo["n_chunks"],
               "transport": comm.transport, "exchange_ms": float(tt[2]), "exposed_exchange_ms": max(0.0, float(tt[0]) - float(tt[1])),
               "ms_per_step_with_exchange": float(tt[0]), "ms_per_step_without_exchange": float(tt[1]),
               "bytes_reduce_scatter_per_rank": xinfo["bytes_reduce_scatter"], "bytes_all_gather_per_rank": xinfo["bytes_all_gather"],
               "bytes_all_reduce": xinfo["bytes_all_reduce"]}
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
    iters_per_s = steps / dt
    value = iters_per_s * total_arcs
    out = None
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        alg = algorithmic_bytes(float(ls.kept_arcs), float(ls.kept_states))
        achieved = alg / (k_ms * 1e-3) / 1e9
        shape = (w.n_states, w.n_arcs, c.n_pairs if not (args.scaling == "strong" and world > 1) else c.n_pairs * world,
                 "in total, sharded" if args.scaling == "strong" and world > 1 else "per GPU",
                 ("%d-%d" % walk) if walk else {"c4a": "5-40", "long": "40-600", "toya": "3-14", "mix": "5-40 / 5-40 / 40-600", "toymix": "3-12 / 3-12 / 8-30"}.get(name, "5-40"))
        out = {
            "metric": METRIC, "value": value, "unit": "arc-updates/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * dt / steps, "ms_per_step_median": float(np.median(step_ms)),
            "ms_per_step_p10": float(np.percentile(step_ms, 10)), "ms_per_step_p90": float(np.percentile(step_ms, 90)),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: " % name + WORKLOAD_TEXT[name] % shape,
                       "pairs_per_gpu": c.n_pairs, "wfst_arcs": int(w.n_arcs), "wfst_states": int(w.n_states),
                       "lattice_arcs_per_gpu": int(ls.kept_arcs), "lattice_states_per_gpu": int(ls.kept_states),
                       "in_arcs_per_lattice_state": float(ls.kept_arcs) / max(1.0, float(ls.kept_states)),
                       "lattice_layout": fb.layout_description(),
                       "parallelism": "corpus-sharded x%d, %s" % (world, exchange)},
            "iters_per_s": iters_per_s,
            "wfst_arcs_x_iters_per_s": iters_per_s * w.n_arcs,
            "ln_corpus_prob_last": lp,
            "lattice_build_s": ls.build_seconds, "synth_gen_s": t_gen,
            "kernel_ms": k_ms,
            "roofline": {"bound": "hbm", "kernel": ("E-step = trans_w_bucket + tile_sweep (a tile's weights in, its lane sweeps out of "
                         "LDS, its posteriors out: one persistent kernel) + trans_c_bucket (posteriors to per-arc counts), timed "
                         "together with HIP events on the trainer's stream") if fb.tile_sweep_tiles else
                         ("E-step = %strans_w_tile_small (weights to lattice order, 1024-position tiles%s) + sweep_lane<XC> " % (
                             ("", ", fetched from the WFST's table: no bucket pass") if fb.weight_source & 1 else ("trans_w_bucket + ", "")) +
                          "(the backward pass stages a tile's posteriors in LDS and writes XC itself: no posterior array, no tile pass "
                          "back) + trans_c_bucket (posteriors to per-arc counts), timed together with HIP events on the trainer's "
                          "stream") if fb.fused_lane_tiles else
                         ("E-step = sweep_wave kernels (weights gathered from the WFST's table: no weight pass%s + "
                          "trans_c_bucket (posteriors to per-arc counts), timed together with HIP events on the trainer's stream") % (
                              "; posteriors straight to the count pass's input: no posterior array, no tile pass)" if fb.weight_source & 4
                              else ") + trans_c_tile")
                         if fb.weight_source & 2 and not ls.n_windowed_pairs and ls.n_bundles == ls.n_pairs_kept else
                         "E-step = trans_w_bucket + trans_w_tile (weights to lattice order) + "
                         "sweep_lane / sweep_wave kernels + trans_c_tile + trans_c_bucket (posteriors to per-arc counts), timed "
                         "together with HIP events on the trainer's stream",
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(name, walk, c.n_pairs),
                         "traffic_source": "profiles/pmc_traffic_%s.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "command (tools/pmc_traffic.sh), committed; null when kernels.hip changed since, the "
                                           "workload differs or a CARMEL_HIP_* switch is set -- not measured by this run" % name,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": k_ms, "kernel_ms_median": float(np.median(kernel_ms)),
                         # SURVEY 8(d)'s WHOLE formula -- the M-step's 16 B per WFST arc included -- over the whole step
                         "frac_iteration": (alg + 16.0 * w.n_arcs) / (1e-3 * 1e3 * dt / steps) / 1e9 / HBM_PEAK_GBS},
        }
        if xch:
            out["exchange_ms"], out["exposed_exchange_ms"], out["exchange"] = xch["exchange_ms"], xch["exposed_exchange_ms"], xch
    fence()
    fb.close()
    if comm is not None:
        comm.close()
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import binding as ob  # CPU restatement of the reference: the checker, timed as the baseline
        cap = args.cpu_sample_pairs if headline else {"c2": 50000, "c4a": 20000, "long": 40, "mix": 4000}.get(name, 20000)
        ns = min(cap, c.n_pairs)
        cs = c.shard(0, max(1, c.n_pairs // ns)) if ns < c.n_pairs else c
        nthreads = args.cpu_threads or min(64, len(os.sched_getaffinity(0)))
        out["cpu_baseline"], out["parity_checked_pairs"] = cpu_leg(ob, w, cs, float(ls.kept_arcs), nthreads, local_rank)
    return out


def _front_end_run(cmd, d, steps, warmup):
    p = subprocess.run(cmd, env=dict(os.environ, CARMEL_TIMING="1", CARMEL_TRAINED_DIR=d), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True)
    if p.returncode != 0:
        raise RuntimeError("front end failed: " + p.stderr[-2000:])
    lat = re.search(r"timing: lattices pairs_kept=(\d+) states=(\d+) arcs=(\d+) layout=(\w+)(?: device_bytes=(\d+) build_seconds=(\S+))?", p.stderr)
    est = [float(x) for x in re.findall(r"timing: i=\d+ estimate (\S+) ms", p.stderr)][warmup:]
    ker = [float(x) for x in re.findall(r"estimate \S+ ms \(kernels (\S+) ms\)", p.stderr)][warmup:]
    mx = [float(x) for x in re.findall(r"timing: i=\d+ maximize (\S+) ms", p.stderr)][warmup:]
    return p, lat, est, ker, mx


def _oracle_per_iter(oc, tail, d):
    """(time of 3 iterations - time of 1) / 2 of the oracle's command line: composition and lattice build cancel"""
    dts = []
    for it in ("1", "3"):
        t0 = time.time()
        q = subprocess.run(oc + [it] + tail, env=dict(os.environ, ORACLE_TRAINED_DIR=d), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, universal_newlines=True)
        dts.append(time.time() - t0)
    return (dts[1] - dts[0]) / 2.0 if (q.returncode == 0 and dts[1] > dts[0]) else None


def run_amb(args, steps, warmup, local_rank=0, rank=0):
    """ambiguous lattices on real data: the reference's own tagging cascade (carmel-tutorial/tagging.*: tag bigram model o
    tag->word lexicon, 46 states / 400 994 composed arcs; a sentence's lattice is positions x candidate tags), its
    1005-sentence corpus repeated"""
    d = tempfile.mkdtemp(prefix="amb_")
    g = lambda n: os.path.join(ROOT, "tests", "golden", n)
    reps = max(1, args.pairs // 1005) if (args.pairs and args.config == "amb") else 400  # 402 000 pairs: enough wavefronts to fill 256 CUs
    open(os.path.join(d, "corpus"), "w").write(open(g("tagging.data")).read() * reps)
    cmd = [os.path.join(ROOT, "carmel_amd", "bin", "carmel"), "--gpu=%d" % local_rank, "--train-cascade", "-HJ", "-M", str(steps + warmup),
           "-X", "1.1", "-e", "0", os.path.join(d, "corpus"), g("tagging.fsa"), g("tagging.fst")]
    p, lat, est, ker, mx = _front_end_run(cmd, d, steps, warmup)
    arcs, states = float(lat.group(3)), float(lat.group(2))
    ms = (sum(est) + sum(mx)) / max(len(est), 1)
    k_ms = sum(ker) / max(len(ker), 1)
    alg = algorithmic_bytes(arcs, states)
    value, world = _replicas(arcs / (ms * 1e-3)) if args.config == "amb" else (arcs / (ms * 1e-3), 1)
    out = {"metric": METRIC, "value": value, "unit": "arc-updates/s", "n_gpus": world, "steps": len(est), "warmup": warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "carmel-tutorial tagging.* x%d" % reps,
           "config": {"workload": "amb: tagging cascade (tag bigram model o lexicon, 46 states / 400994 composed arcs), the "
                                  "tutorial's 1005 sentences x %d = %d pairs, carmel --train-cascade through the front end; "
                                  "ambiguous lattices (positions x candidate tags)" % (reps, int(lat.group(1))),
                      "lattice_arcs_per_gpu": int(arcs), "lattice_states_per_gpu": int(states), "lattice_layout": lat.group(4),
                      "parallelism": "replicas x%d" % world},
           "lattice_build_s": float(lat.group(6)), "estep_ms": sum(est) / max(len(est), 1), "mstep_ms": sum(mx) / max(len(mx), 1),
           "kernel_ms": k_ms,
           "roofline": {"bound": "hbm", "kernel": "E-step (weights to lattice order, lane / wave / bundle sweeps, posteriors to counts), HIP "
                        "events on the trainer's stream", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("amb", None, 1005 * reps), "algorithmic_bytes_per_launch": alg,
                        "kernel_ms": k_ms}}
    if not args.no_cpu_baseline and rank == 0:
        # the trace of the reference itself: 25.9-26.5 s per iteration on this cascade (commands.trace:5868-5889,
        # unknown hardware, no derivation caching); here the oracle with cached derivations on the 1005 sentences
        oc = [os.path.join(ROOT, "oracle", "oracle_carmel"), "--train-cascade", "-HJ", "-:", "-X", "1.1", "-e", "0", "-M"]
        per_iter = _oracle_per_iter(oc, [g("tagging.data"), g("tagging.fsa"), g("tagging.fst")], d)
        if per_iter:
            out["cpu_baseline"] = {"value": arcs / reps / per_iter, "unit": "arc-updates/s", "cores": 1, "kind": "port",
                                   "sample": "the oracle's command line on the 1005 sentences (one repetition), cached "
                                             "derivations: (time of 3 iterations - time of 1) / 2 = %.2f s per iteration "
                                             "(the reference's own trace: 26 s per iteration without caching)" % per_iter}
    return out


def run_crp(args, local_rank=0, rank=0, reps_parallel=100, sweeps_parallel=40, sweeps_exact=60):
    """`carmel --crp` (gibbs.cc:306-371, derivations.h:345-375, gibbs.hpp:769-877) through the front end on the tutorial's
    tagging cascade: `exact` = the reference's chain (blocks strictly in order, one workgroup; 1005 blocks), the headline of
    this workload = the stale-count parallel sweep (--crp-parallel) on the corpus x reps_parallel.  A step is one sweep.
    Bytes per sweep (the lattice model of SURVEY 8(d) restricted to what a sampling sweep touches): per
    lattice arc its 8 B record once (proposal weight) + 16 B per chain element (count and norm sum of its parameter; 2
    elements on this cascade) + the 8 B proposal weight written and read back by the backward sweep; per lattice state 8 B
    (beta); per sampled parameter 64 B (count and norm sum read-modify-written twice: old sample out, new sample in)."""
    d = tempfile.mkdtemp(prefix="crp_")
    g = lambda n: os.path.join(ROOT, "tests", "golden", n)
    exe = os.path.join(ROOT, "carmel_amd", "bin", "carmel")
    pat = re.compile(r"timing: gibbs mode=(\w+) sweeps=(\d+) blocks=(\d+) lattice_states=(\d+) lattice_arcs=(\d+) sampled_params=(\d+) seconds=(\S+)")

    def leg(reps, sweeps, extra):
        corpus = g("tagging.data")
        if reps > 1:
            corpus = os.path.join(d, "corpus%d" % reps)
            open(corpus, "w").write(open(g("tagging.data")).read() * reps)
        # two runs, n and 3n sweeps: composition, lattice construction and the first sample cancel in the difference
        res = []
        for k in (1, 3):
            p = subprocess.run([exe, "--gpu=%d" % local_rank, "--crp", "-M", str(k * sweeps), "-R", "7"] + extra + [corpus, g("tagging.fsa"), g("tagging.fst")],
                               env=dict(os.environ, CARMEL_TIMING="1", CARMEL_TRAINED_DIR=d), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               universal_newlines=True)
            m = pat.search(p.stderr)
            if p.returncode != 0 or not m:
                raise RuntimeError("front end failed: " + p.stderr[-1500:])
            res.append(m)
        n_sw = int(res[1].group(2)) - int(res[0].group(2))
        sec = float(res[1].group(7)) - float(res[0].group(7))
        m = res[1]
        arcs, states, sampled = float(m.group(5)), float(m.group(4)), float(m.group(6))
        ms = 1e3 * sec / n_sw
        alg = arcs * (8.0 + 2 * 16.0 + 16.0) + states * 8.0 + sampled * 64.0
        return {"blocks": int(m.group(3)), "lattice_arcs": int(arcs), "lattice_states": int(states), "sampled_params_per_sweep": int(sampled),
                "sweeps": n_sw, "ms_per_step": ms, "value": arcs / (ms * 1e-3), "algorithmic_bytes_per_launch": alg,
                "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    par = leg(reps_parallel, sweeps_parallel, ["--crp-parallel"])
    ex = leg(1, sweeps_exact, [])
    # the reference's chain 64 times over (--crp-restarts=63: independent runs, gibbs.hpp:880-914) as 64 concurrent wavefronts
    ex64 = leg(1, max(10, sweeps_exact // 3), ["--crp-restarts=63"])
    out = {"metric": "lattice-arc updates/sec of Gibbs sweeps (sweeps/sec x derivation-lattice arcs)", "value": par["value"], "unit": "arc-updates/s",
           "n_gpus": 1, "steps": par["sweeps"], "warmup": 0, "ms_per_step": par["ms_per_step"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "carmel-tutorial tagging.* x%d" % reps_parallel,
           "config": {"workload": "crp: carmel --crp --crp-parallel on the tagging cascade, %d blocks (the tutorial's 1005 sentences x %d), "
                                  "through the front end; `exact`: the reference's chain on the 1005 blocks" % (par["blocks"], reps_parallel),
                      "lattice_arcs_per_gpu": par["lattice_arcs"], "lattice_states_per_gpu": par["lattice_states"]},
           "kernel_ms": par["ms_per_step"],
           "roofline": {"bound": "hbm", "kernel": "one sweep = gibbs_lane_kernel (64 lattices a wavefront, one per lane; round 6) + gibbs_lane_recount + gibbs_normsum + gibbs_commit; wall time per sweep "
                        "(difference of two runs of the front end: launch gaps included)", "achieved": par["frac"] * HBM_PEAK_GBS,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": par["frac"], "traffic": pmc_traffic_crp(par["blocks"]),
                        "algorithmic_bytes_per_launch": par["algorithmic_bytes_per_launch"]},
           "exact": {k: ex[k] for k in ("blocks", "sweeps", "ms_per_step", "value", "frac", "lattice_arcs")}}
    out["exact"]["chains64"] = {"runs": 64, "ms_per_chain_sweep": ex64["ms_per_step"], "value": ex64["value"], "unit": "arc-updates/s",
                                "aggregate_over_one_chain": ex["ms_per_step"] / ex64["ms_per_step"],
                                "note": "carmel --crp --crp-restarts=63: the 64 runs side by side, a wavefront each (GxArgs::n_chains); value = all chains' sweeps"}
    out["exact"]["unit"] = "arc-updates/s"
    out["exact"]["note"] = "gibbs_exact_wave_kernel: one wavefront, blocks strictly in order (the reference's chain; latency-bound by construction)"
    if not args.no_cpu_baseline and rank == 0:
        from oracle import binding as ob
        rd = lambda n: open(g(n)).read()
        oc = ob.OracleCascade([rd("tagging.fsa"), rd("tagging.fst")])
        corp = oc.corpus(rd("tagging.data"))
        ob.gibbs_run(oc, corp, 7, normby="CC", priors=[0.0, 0.0], iters=1, burnin=0)
        n = 120
        t0 = time.time()
        ob.gibbs_run(oc, corp, 7, normby="CC", priors=[0.0, 0.0], iters=n - 1, burnin=0)
        cdt = time.time() - t0
        out["cpu_baseline"] = {"value": ex["lattice_arcs"] * n / cdt, "unit": "arc-updates/s", "cores": 1, "kind": "port",
                               "sample": "the oracle's sampler (the reference's chain) on the 1005 sentences, %d sweeps, its own counter-based "
                                         "uniforms: %.1f ms per sweep" % (n, 1e3 * cdt / n)}
    return out


def run_c3(args, steps, warmup, local_rank=0, rank=0):
    """BASELINE.json configs[2]: the cipher cascade (character bigram LM o substitution channel) through the front end"""
    from carmel_amd import synth
    d = tempfile.mkdtemp(prefix="c3_")
    lm, ch, co = synth.cipher_files(args.lines)
    for name, txt in (("lm.wfsa", lm), ("ch.fst", ch), ("corpus", co)):
        open(os.path.join(d, name), "w").write(txt)
    cmd = [os.path.join(ROOT, "carmel_amd", "bin", "carmel"), "--gpu=%d" % local_rank, "--train-cascade", "--normby=NC", "-HJ", "-M", str(steps + warmup), "-X", "1.1",
           "-e", "0", os.path.join(d, "corpus"), os.path.join(d, "lm.wfsa"), os.path.join(d, "ch.fst")]
    p, lat, est, ker, mx = _front_end_run(cmd, d, steps, warmup)
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
    value, world = _replicas(arcs / (ms * 1e-3)) if args.config == "c3" else (arcs / (ms * 1e-3), 1)
    out = {"metric": METRIC, "value": value, "unit": "arc-updates/s", "n_gpus": world, "steps": len(est), "warmup": warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "c3: cipher cascade, character bigram LM (29 states, locked) o 27x27 substitution channel, "
                                  "%d lines of 30-80 symbols, carmel --train-cascade --normby=NC through the front end; "
                                  "lattices unrolled over string positions, never stored" % args.lines,
                      "lattice_arcs_per_gpu": int(arcs), "lattice_layout": lat.group(4), "parallelism": "replicas x%d" % world},
           "kernel_ms": k_ms,
           "roofline": {"bound": "mfma", "kernel": kname,
                        "achieved": flops / (k_ms * 1e-3) / 1e12, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (k_ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS, "traffic": None, "kernel_ms": k_ms}}
    if not args.no_cpu_baseline and rank == 0:
        n = 150
        lm2, ch2, co2 = synth.cipher_files(n)
        for name, txt in (("lm2.wfsa", lm2), ("ch2.fst", ch2), ("corpus2", co2)):
            open(os.path.join(d, name), "w").write(txt)
        oc = [os.path.join(ROOT, "oracle", "oracle_carmel"), "--train-cascade", "--normby=NC", "-HJ", "-:", "-X", "1.1", "-e", "0", "-M"]
        per_iter = _oracle_per_iter(oc, [os.path.join(d, "corpus2"), os.path.join(d, "lm2.wfsa"), os.path.join(d, "ch2.fst")], d)
        if per_iter:
            out["cpu_baseline"] = {"value": arcs * n / args.lines / per_iter, "unit": "arc-updates/s", "cores": 1, "kind": "port",
                                   "sample": "the oracle's command line on %d lines of the same model, cached derivations: "
                                             "(time of 3 iterations - time of 1) / 2 = %.2f s per iteration" % (n, per_iter)}
    return out


def run_c5(args, steps, warmup, local_rank=0, rank=0, exact_sweeps=1):
    """BASELINE.json configs[4]: forest-em --crp sweeps over synthetic packed forests.  A step is one Gibbs sweep over all
    forests: the parallel stale-count sweep (the throughput mode) and -- `exact` -- the reference's sequential chain"""
    import numpy as np
    import carmel_amd
    carmel_amd.options_from_env()  # (bench.py is a front end: CARMEL_HIP_<KEY> in its environment are the library's options)
    from carmel_amd import synth
    from carmel_amd.forests import HipForests
    node_off, label, ref, nxt, n_rules, goff, grule = synth.random_forests(args.forests)
    rng = np.random.default_rng(4)
    lw = np.log(rng.uniform(0.05, 1.0, n_rules))
    hf = HipForests(node_off, label, ref, nxt, n_rules, lw, goff, grule, device=local_rank)
    hf.maximize()
    hf.gibbs(max(warmup, 1), alpha=0.1, seed=4 + rank, mode=1)
    t0 = time.perf_counter()
    hf.gibbs(steps, burnin=steps // 4, alpha=0.1, seed=4 + rank, mode=1)
    dt = time.perf_counter() - t0
    sweeps = steps + 1
    n_nodes = float(len(label))
    n_sampled = float(np.mean([len(hf.sample(f)) for f in range(0, args.forests, max(1, args.forests // 2000))])) * args.forests
    ms = 1e3 * dt / sweeps
    alg = 40.0 * n_nodes + 64.0 * n_sampled  # node record, proposal gather, inside write + read; two count RMWs per sampled rule
    value, world = _replicas(n_nodes * sweeps / dt) if args.config == "c5" else (n_nodes * sweeps / dt, 1)
    out = {"metric": "forest-node updates/sec (Gibbs sweeps/sec x nodes of the packed forests; carmel's arc-updates for "
                     "forest-em)", "value": value, "unit": "node-updates/s", "n_gpus": world, "steps": sweeps,
           "warmup": max(warmup, 1), "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "c5: forest-em --crp on %d synthetic packed forests (%d nodes, %d parameters, alpha 0.1), "
                                  "parallel stale-count sweeps" % (args.forests, int(n_nodes), n_rules - 1),
                      "sampled_rules_per_sweep": int(n_sampled), "parallelism": "replicas x%d (independent chains)" % world},
           "sweeps_per_s": sweeps / dt, "kernel_ms": ms,
           "roofline": {"bound": "hbm", "kernel": "one sweep = forest_proposal + forest_sample (per launch class) + forest_recount; "
                        "timed as wall time per sweep around carmel_hip_forests_gibbs (launch gaps included)",
                        "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic_c5(args.forests), "algorithmic_bytes_per_launch": alg}}
    if exact_sweeps:
        # the reference's chain (forests strictly in order): a parity device, latency-bound by construction
        hf.set_weights(lw)
        hf.maximize()
        t0 = time.perf_counter()
        hf.gibbs(exact_sweeps - 1, alpha=0.1, seed=4 + rank, mode=0)  # (iter + 1 sweeps: the initial sample counts)
        edt = time.perf_counter() - t0
        out["exact"] = {"sweeps": exact_sweeps, "ms_per_step": 1e3 * edt / exact_sweeps, "value": n_nodes * exact_sweeps / edt,
                        "unit": "node-updates/s", "note": "mode 0: the reference's sequential chain over all %d forests" % args.forests}
        # --crp-restarts=63: 64 such chains side by side (a wavefront each, one launch per sweep for all of them)
        hf.set_weights(lw)
        hf.maximize()
        t0 = time.perf_counter()
        hf.gibbs(exact_sweeps - 1, alpha=0.1, seed=4 + rank, mode=0, restarts=63)
        edt64 = time.perf_counter() - t0
        v64 = n_nodes * 64 * exact_sweeps / edt64
        out["exact"]["chains64"] = {"runs": 64, "ms_per_chain_sweep": 1e3 * edt64 / (64 * exact_sweeps), "value": v64, "unit": "node-updates/s",
                                    "aggregate_over_one_chain": v64 / out["exact"]["value"],
                                    "note": "forest-em --crp-restarts=63: 64 independent exact chains in one launch per sweep"}
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
        of.gibbs(4, 9, burnin=2, alpha=0.1)  # (the oracle's own counter-based uniforms: no Python call per draw)
        cdt = time.time() - t0
        out["cpu_baseline"] = {"value": float(of.n_nodes) * 10 / cdt, "unit": "node-updates/s", "cores": 1, "kind": "port",
                               "sample": "the first %d forests (%d nodes), same parameters; 10 exact (sequential) sweeps of the "
                                         "scalar oracle" % (nf, of.n_nodes)}
    return out


def compact_line(out):
    """the line the driver records: the contract's keys of the headline (texts shortened), every secondary as a handful of
    numbers.  The driver's record keeps the last 2 000 characters of stdout: all of this fits."""
    def short(t, n):
        t = str(t)
        return t if len(t) <= n else t[:n - 1] + "~"

    def rf(r):
        if not r:
            return None
        o = {"bound": r.get("bound"), "achieved": _r4(r.get("achieved")), "peak": r.get("peak"), "unit": r.get("unit"),
             "frac": _r4(r.get("frac")), "traffic": _r4(r.get("traffic"))}
        if "frac_iteration" in r:
            o["frac_iteration"] = _r4(r["frac_iteration"])
        return o

    def cb(c):
        return None if not c else {"value": _r4(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"),
                                   "sample": short(c.get("sample", ""), 90)}

    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median", "ms_per_step_p10", "ms_per_step_p90",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: (_r4(out[k]) if isinstance(out.get(k), float) else out.get(k)) for k in keep if k in out}
    line["metric"] = short(line.get("metric", ""), 90)
    cfg = out.get("config", {})
    line["config"] = {"workload": short(cfg.get("workload", ""), 150), "parallelism": cfg.get("parallelism")}
    line["kernel_ms"] = _r4(out.get("kernel_ms"))
    line["roofline"] = rf(out.get("roofline"))
    line["cpu_baseline"] = cb(out.get("cpu_baseline"))
    if "parity_checked_pairs" in out:
        line["parity_checked_pairs"] = out["parity_checked_pairs"]
    if "exposed_exchange_ms" in out:
        line["exposed_exchange_ms"] = _r4(out["exposed_exchange_ms"])
    sec = {}
    for name, r in (out.get("secondary") or {}).items():
        if "error" in r:
            sec[name] = {"error": short(r["error"], 80)}
            continue
        e = {}
        if "exact" in r:  # the samplers: the reference's chain (the default mode) first, the stale-count sweep (non-default) after it
            e["exact_ms"] = _r4(r["exact"].get("ms_per_step"))
            e["exact_value"] = _r4(r["exact"].get("value"))
            if "chains64" in r["exact"]:
                e["exact_x64"] = _r4(r["exact"]["chains64"].get("aggregate_over_one_chain"))
            e["parallel_ms"] = _r4(r.get("ms_per_step"))
        e.update({"ms": _r4(r.get("ms_per_step")), "k_ms": _r4(r.get("kernel_ms")), "frac": _r4((r.get("roofline") or {}).get("frac")),
                  "traffic": _r4((r.get("roofline") or {}).get("traffic")), "cpu": _r4((r.get("cpu_baseline") or {}).get("value")),
                  "value": _r4(r.get("value"))})
        if "parity_checked_pairs" in r:
            e["parity"] = r["parity_checked_pairs"]
        sec[name] = e
    if sec:
        line["secondary"] = sec
    line["bench_wall_s"] = _r4(out.get("bench_wall_s"))
    return line


def _r4(x):
    """four significant digits (None stays None)"""
    if x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.4g" % x)
    except (TypeError, ValueError):
        return x


def _self_launch(n):
    """`python bench.py --gpus N` without torch.distributed.run: one child process per rank with the launcher's environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), rank 0's stdout passed through; returns the exit code.
    A rank that fails ends the others (they would wait in a collective for ever)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            c = p.poll()
            if c is None:
                continue
            alive.remove(p)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                for q in alive:  # exact PIDs we started
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 200 (c3: 20); c5: 1000 sweeps, the length BASELINE.json's config names")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=["c2", "c4", "c4a", "long", "mix", "toy", "toya", "toymix", "c3", "c5", "amb", "crp"])
    ap.add_argument("--lines", type=int, default=200000, help="c3: corpus lines")
    ap.add_argument("--forests", type=int, default=100000, help="c5: forests")
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU of the headline workload (default: the config's)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --pairs per GPU; strong: --pairs in total, sharded over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="the headline workload only")
    ap.add_argument("--secondary", default="c4a,amb,c2,long,mix,c3,c5,crp", help="which workloads follow the headline at N = 1")
    ap.add_argument("--full-out", default=None, help="file for the complete JSON (default: gpurun_out/bench_full.json when that directory "
                                                      "exists or can be made, else none); stdout carries the compact line")
    ap.add_argument("--secondary-steps", type=int, default=40)
    ap.add_argument("--cpu-sample-pairs", type=int, default=200000)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0: host cores, at most 64)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--comm-plugin", default=None, help="a transport library for the exchange instead of RCCL (carmel_hip_comm_create_custom), "
                    "e.g. tests/native/libhosttransport.so: every rank then runs on GPU 0 (several ranks on a one-GPU box)")
    ap.add_argument("--exchange", default="sharded", choices=["sharded", "allreduce", "collectives", "direct"],
                    help="form of the count exchange at N > 1 (sharded: direct point-to-point groups where the transport has them, else the collectives)")
    ap.add_argument("--exchange-chunks", type=int, default=0, help="arc-range chunks of the sharded exchange (0: the library's default, 4; at most 16)")
    ap.add_argument("--no-exchange-loopback", action="store_true", help="N = 1: skip the loopback measurement of the exchange")
    ap.add_argument("--walk-arcs", default=None, help="min,max arcs of the headline's random walks (default: the config's; other values are experiments)")
    args = ap.parse_args()
    if args.steps is None:
        # (the EM headline's timed region used to be 20 steps = 9 ms of a c4 run: round-5 verdict.  200 steps, median and p10 / p90 on the line)
        args.steps = 1000 if args.config == "c5" else 20 if args.config in ("c3", "amb", "crp") else 200

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (no launcher): start the N ranks ourselves.  Nothing in this process has touched
        # a GPU yet (no HIP call, torch not imported), and the ranks are CHILD processes -- never an exec of this one.
        sys.exit(_self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    t_start = time.time()
    if args.config == "crp":
        out = run_crp(args, local_rank, rank)
    elif args.config in ("c3", "c5", "amb"):
        out = {"c3": run_c3, "c5": run_c5, "amb": run_amb}[args.config](args, args.steps, args.warmup, local_rank, rank)
    else:
        if world != args.gpus:
            args.gpus = world
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: the EM hot path has no CPU fallback")
        # --comm-plugin (tests on a one-GPU box): every rank on GPU 0, the sums through the plugin's transport, gloo for control
        one_device = bool(args.comm_plugin)
        if one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if one_device:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        out = run_em(args.config, args, dist, rank, world, local_rank, one_device, headline=True)
        if world > 1:
            dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not args.no_secondary:
            sec = {}
            for name in [s for s in args.secondary.split(",") if s and s != args.config]:
                t0 = time.time()
                try:
                    if name == "amb":
                        r = run_amb(args, args.secondary_steps, 3, local_rank, rank)
                    elif name == "c3":
                        r = run_c3(args, args.secondary_steps, 3, local_rank, rank)
                    elif name == "c5":
                        r = run_c5(args, 1000, 3, local_rank, rank, exact_sweeps=3)
                    elif name == "crp":
                        r = run_crp(args, local_rank, rank)
                    else:
                        r = run_em(name, args, None, 0, 1, local_rank, False, headline=False)
                except Exception as e:  # noqa: BLE001  (a secondary must not take the headline down with it)
                    r = {"error": "%s: %s" % (type(e).__name__, e), "traceback": traceback.format_exc()[-1500:]}
                r["wall_s"] = time.time() - t0
                sec[name] = r
            out["secondary"] = sec
        out["bench_wall_s"] = time.time() - t_start
        full = args.full_out
        if full is None:
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                full = os.path.join(ROOT, "gpurun_out", "bench_full.json")
            except OSError:
                full = None
        if full:
            try:
                with open(full, "w") as fh:
                    fh.write(json.dumps(out) + "\n")
            except OSError:
                pass
        print(json.dumps(compact_line(out)))


if __name__ == "__main__":
    main()
