"""Worker of tests/test_multirank_gpu.py: one rank of a corpus-sharded EM run through the PRODUCT's N > 1 path -- the
trainer's device count buffer handed to the caller (carmel_hip_use_external_counts), estimate_async, an all-reduce of
that buffer, carmel_hip_read_scalars, carmel_hip_maximize.  Both ranks may share one GPU, so the all-reduce is staged
through the host over gloo (RCCL refuses two ranks on one device); `--rccl` uses carmel_hip_allreduce_counts instead
(one GPU per rank, or world 1).

usage: multirank_worker.py RANK WORLD PORT OUT.npy MODE(synth|cipher|cipher-explicit|dense) [--rccl]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(mode, rank, world):
    from carmel_amd import synth
    from carmel_amd.model import Corpus, Wfst
    from carmel_amd.trainer import HipForwardBackward
    if mode == "synth":
        w = synth.random_wfst(60, 8, n_sym=4, p_eps=0.15, seed=3)
        c = synth.random_walk_corpus(w, 301, min_arcs=3, max_arcs=12, seed=3, out_degree=8)  # odd size: ragged shards
        return w, HipForwardBackward(w, c.shard(rank, world), device=0)
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
        assert lib.carmel_hip_lattice_layout(fb.h) == 2, "expected the dense layout"
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
        os.environ["CARMEL_HIP_UNROLLED"] = "0"
        mode = "cipher"
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w, fb = build(mode, rank, world)
    counts = torch.zeros(w.n_arcs + 4, dtype=torch.float64, device="cuda:0")
    fb.use_external_counts(counts.data_ptr())
    comm = None
    if rccl:
        from carmel_amd.trainer import HipComm
        ids = [HipComm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        comm = HipComm(0, rank, world, ids[0])
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
    if rank == 0:
        np.save(out, np.concatenate([fb.weights(), logs]))
    fb.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
