"""CPU, world_size 2, gloo: the data-parallel exchange of the N>1 path — corpus shards, one all-reduce of the
per-arc count vector (+3 scalars) per iteration, replicated M-step — reproduces the single-process result.
The per-shard E-step here comes from the oracle (no GPU in this container); on the GPU box bench.py drives the same
exchange with RCCL over the trainers' device buffers."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from carmel_amd import synth
    from oracle import binding as ob
    w = synth.random_wfst(40, 8, n_sym=4, p_eps=0.15, seed=3)
    c = synth.random_walk_corpus(w, 101, min_arcs=3, max_arcs=10, seed=3, out_degree=8)  # odd size: ragged shards
    shard = c.shard(rank, world)
    ow = ob.OracleWfst.from_arrays(w)
    ow.normalize(0, 0.0)
    logs = []
    for _ in range(3):
        r = ob.estimate(ow, ob.OracleCorpus.from_arrays(shard))
        buf = torch.zeros(w.n_arcs + 4, dtype=torch.float64)  # the layout of carmel_hip_counts_dev
        buf[:w.n_arcs] = torch.from_numpy(np.exp(r["counts_ln"]))
        buf[w.n_arcs] = r["sum_logprob"]
        buf[w.n_arcs + 1] = r["sum_weighted_logprob"]
        buf[w.n_arcs + 2] = float(r["has_deriv"].sum())
        dist.all_reduce(buf)
        logs.append(float(buf[w.n_arcs]))
        counts = buf[:w.n_arcs].numpy()
        ow.set_logw(np.log(np.maximum(counts, 1e-300)) + np.where(counts > 0, 0.0, -np.inf))
        ow.normalize(0, 0.0)  # replicated M-step
    if rank == 0:
        np.save(out, np.concatenate([ow.arrays()["logw"], logs]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_single_process(tmp_path, oracle):
    port = 29500 + (os.getpid() % 2000)
    out = str(tmp_path / "w.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    from carmel_amd import synth
    w = synth.random_wfst(40, 8, n_sym=4, p_eps=0.15, seed=3)
    c = synth.random_walk_corpus(w, 101, min_arcs=3, max_arcs=10, seed=3, out_degree=8)
    ow = oracle.OracleWfst.from_arrays(w)
    ow.normalize(0, 0.0)
    logs = []
    for _ in range(3):
        r = oracle.estimate(ow, oracle.OracleCorpus.from_arrays(c))
        logs.append(r["sum_logprob"])
        counts = np.exp(r["counts_ln"])
        ow.set_logw(np.log(np.maximum(counts, 1e-300)) + np.where(counts > 0, 0.0, -np.inf))
        ow.normalize(0, 0.0)
    ref = np.concatenate([ow.arrays()["logw"], logs])
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(got))
    np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-9)


def test_shards_partition_the_corpus():
    from carmel_amd import synth
    w = synth.random_wfst(30, 6, n_sym=4, seed=5)
    c = synth.random_walk_corpus(w, 37, seed=5, out_degree=6)
    for world in (1, 2, 3, 8):
        tot_in = tot_pairs = 0
        for r in range(world):
            s = c.shard(r, world)
            tot_pairs += s.n_pairs
            tot_in += int(s.in_off[-1])
            assert len(s.in_sym) == int(s.in_off[-1]) and len(s.out_sym) == int(s.out_off[-1])
        assert tot_pairs == c.n_pairs and tot_in == int(c.in_off[-1])
