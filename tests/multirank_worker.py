"""Worker of tests/test_multirank_gpu.py: one rank of a corpus-sharded EM run through the PRODUCT's N > 1 path -- the
trainer's device count buffer handed to the caller (carmel_hip_use_external_counts), estimate_async, an all-reduce of
that buffer, carmel_hip_read_scalars, carmel_hip_maximize.  Both ranks may share one GPU, so the all-reduce is staged
through the host over gloo (RCCL refuses two ranks on one device); `--rccl` uses carmel_hip_allreduce_counts instead
(one GPU per rank, or world 1).

`--plugin=LIB.so:SESSION` (with --rccl) makes the library's communicator over that transport instead of RCCL
(carmel_hip_comm_create_custom; tests/native/libhosttransport.so lets the ranks share one GPU); `--plan[=K]` plans the
exchange (carmel_hip_exchange_plan: sharded where the model allows it), `--plan-allreduce` plans its all-reduce form;
`--disagree` makes rank 1 keep explicit lattices first (the ranks' layouts differ: the plan must refuse, then every rank
rebuilds with explicit lattices).

usage: multirank_worker.py RANK WORLD PORT OUT.npy MODE(synth|synth-big|cipher|cipher-explicit|dense) [--rccl] [...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(mode, rank, world):
    import carmel_amd
    carmel_amd.options_from_env()  # (the parent test exported its switches: a front end translates them, the library does not read them)
    from carmel_amd import synth
    from carmel_amd.model import Corpus, Wfst
    from carmel_amd.trainer import HipForwardBackward
    if mode == "synth":
        w = synth.random_wfst(60, 8, n_sym=4, p_eps=0.15, seed=3)
        c = synth.random_walk_corpus(w, 301, min_arcs=3, max_arcs=12, seed=3, out_degree=8)  # odd size: ragged shards
        return w, HipForwardBackward(w, c.shard(rank, world), device=0)
    if mode == "synth-big":  # enough arcs for the sharded exchange (world * 512 at least), groups straddling piece boundaries
        w = synth.random_wfst(3001, 7, n_sym=5, p_eps=0.15, seed=4)
        c = synth.random_walk_corpus(w, 4001, min_arcs=3, max_arcs=14, seed=4, out_degree=7)
        fb = HipForwardBackward(w, c.shard(rank, world), device=0)
        if os.environ.get("CARMEL_HIP_TILE_GATHER") in ("0", "1"):  # (the test that forces the tile passes' source of weights)
            assert (fb.weight_source & 1) == int(os.environ["CARMEL_HIP_TILE_GATHER"])
        return w, fb
    if mode == "waves":  # one-per-wavefront lattices: weights gathered from the table, posteriors straight to the count pass's input
        w, c = synth.make_config("long", n_pairs=36)
        fb = HipForwardBackward(w, c.shard(rank, world), device=0)
        assert fb.weight_source & 6 == 6, "expected gathering wave sweeps that write XC"
        return w, fb
    from oracle import binding as ob
    g = lambda n: open(os.path.join(ROOT, "tests", "golden", n)).read()
    if mode == "dense":  # config 3's shape: the unrolled sweep in its rank-1 dense form (dense.hpp), counts per channel parameter
        from carmel_amd.model import NORM_CONDITIONAL, NORM_NONE
        lm, ch, co = synth.cipher_files(90, min_len=5, max_len=30, seed=5)
        oc = ob.OracleCascade([lm, ch])
        a = oc.composed().arrays()
        w = Wfst(a["n_states"], a["final"], a["src"], a["dst"], a["isym"], a["osym"], a["logw"], a["group"])
        ca = oc.corpus(co).arrays()
        c = Corpus(ca["in_off"], ca["in_sym"], ca["out_off"], ca["out_sym"], ca["weight"])
        fb = HipForwardBackward(w, c.shard(rank, world), cascade=oc.as_dict([NORM_NONE, NORM_CONDITIONAL], [0.0, 0.0]), device=0)
        from carmel_amd._capi import lib
        assert os.environ.get("CARMEL_HIP_UNROLLED") == "0" or lib.carmel_hip_lattice_layout(fb.h) == 2, "expected the dense layout"
        return w, fb
    oc = ob.OracleCascade([g("cipher.wfsa"), g("cipher.fst")])
    a = oc.composed().arrays()
    w = Wfst(a["n_states"], a["final"], a["src"], a["dst"], a["isym"], a["osym"], a["logw"], a["group"])
    ca = oc.corpus(g("cipher.data")).arrays()
    c = Corpus(ca["in_off"], ca["in_sym"], ca["out_off"], ca["out_sym"], ca["weight"])
    from carmel_amd.model import NORM_CONDITIONAL
    return w, HipForwardBackward(w, c.shard(rank, world), cascade=oc.as_dict([NORM_CONDITIONAL, NORM_CONDITIONAL], [0.0, 0.0]),
                                 device=0)


def main():
    rank, world, port, out, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    rccl = "--rccl" in sys.argv
    if mode == "cipher-explicit":
        os.environ["CARMEL_HIP_UNROLLED"] = "0"  # (build() translates the environment into the library's options, after torch is up)
        mode = "cipher"
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w, fb = build(mode, rank, world)
    plugin = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--plugin=")]
    plan = [a for a in sys.argv if a.startswith("--plan")]
    counts = None
    if not plan:  # (a planned exchange works on the trainer's own buffer)
        counts = torch.zeros(w.n_arcs + 4, dtype=torch.float64, device="cuda:0")
        fb.use_external_counts(counts.data_ptr())
    comm = None
    if rccl:
        from carmel_amd.trainer import HipComm
        if plugin:
            path, session = plugin[0].rsplit(":", 1)
            comm = HipComm.custom(path, session, 0, rank, world)
        else:
            ids = [HipComm.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            comm = HipComm(0, rank, world, ids[0])
    if comm is not None and "--selftest" in sys.argv:
        comm.selftest()
        comm.selftest(100003)
    info = None
    if plan:
        if "--disagree" in sys.argv:
            if rank == 1:  # this rank keeps explicit lattices, the others unroll: the plan must refuse on EVERY rank
                fb.set_layout_policy(False)
                fb.rebuild_lattices()
            try:
                fb.exchange_plan(comm)
                raise SystemExit("the plan accepted ranks with different lattice layouts")
            except RuntimeError as e:
                assert "different layouts" in str(e), str(e)
            fb.set_layout_policy(False)
            fb.rebuild_lattices()
        k = plan[0].split("=", 1)
        form = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--form=")]
        info = fb.exchange_plan(comm, int(k[1]) if len(k) > 1 and k[0] == "--plan" else 0, force_allreduce=plan[0] == "--plan-allreduce",
                                form=form[0].replace("auto-direct", "auto") if form else "auto")
        if form:  # ("auto-direct": the library's own choice must be the direct form)
            assert info["form"] == form[0].replace("auto-", ""), info
    logs = []
    for it in range(4):
        if fb.cascade is not None and it > 0:
            fb.save_counts()
        fb.estimate_async()
        if comm is not None:
            fb.allreduce_counts(comm)
        elif world > 1:
            fb.synchronize()
            host = counts.cpu()
            dist.all_reduce(host)
            counts.copy_(host)
            torch.cuda.synchronize()
        lp, wlp, n = fb.read_scalars()
        logs += [lp, wlp, float(n)]
        fb.maximize(1.0)
    if "--check-counts" in sys.argv:  # after a sharded exchange the whole count vector must still be there for the asking
        fb.estimate_async()
        fb.allreduce_counts(comm)
        cnt = fb.counts()
        logs += [float(cnt.sum()), float((cnt * np.arange(len(cnt))).sum())]
    if rank == 0:
        np.save(out, np.concatenate([fb.weights(), logs, [1.0 if (info and info["sharded"]) else 0.0]]))
        if info:  # what one iteration's exchange moves (carmel_hip_exchange_info), beside the results
            import json
            json.dump(info, open(out + ".info.json", "w"))
    fb.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
