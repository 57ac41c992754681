"""Flat-array containers for a transducer and a training corpus, in the layout the C-ABI takes.

Arc order is the reference's arc-id order: state-major, each state's arcs in list order
(carmel/src/derivations.h:86-101).  Weights are natural logs (graehl/shared/weight.h:132-135).
"""
import numpy as np

NO_GROUP = 0xFFFFFFFF
LOCKED_GROUP = 0
EPS = 0

NORM_CONDITIONAL, NORM_JOINT, NORM_NONE = 0, 1, 2


class Wfst(object):
    def __init__(self, n_states, final, src, dst, isym, osym, logw, group=None):
        self.n_states = int(n_states)
        self.final = int(final)
        self.src = np.ascontiguousarray(src, dtype=np.uint32)
        self.dst = np.ascontiguousarray(dst, dtype=np.uint32)
        self.isym = np.ascontiguousarray(isym, dtype=np.uint32)
        self.osym = np.ascontiguousarray(osym, dtype=np.uint32)
        self.logw = np.ascontiguousarray(logw, dtype=np.float64)
        if group is None:
            group = np.full(len(self.src), NO_GROUP, dtype=np.uint32)
        self.group = np.ascontiguousarray(group, dtype=np.uint32)
        n = len(self.src)
        assert all(len(a) == n for a in (self.dst, self.isym, self.osym, self.logw, self.group))
        assert n == 0 or np.all(np.diff(self.src.astype(np.int64)) >= 0), "arcs must be state-major"

    @property
    def n_arcs(self):
        return len(self.src)


class Corpus(object):
    """training_corpus (carmel/src/train.h:134-189) as CSR symbol arrays"""

    def __init__(self, in_off, in_sym, out_off, out_sym, weight=None):
        self.in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        self.in_sym = np.ascontiguousarray(in_sym, dtype=np.uint32)
        self.out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
        self.out_sym = np.ascontiguousarray(out_sym, dtype=np.uint32)
        n = len(self.in_off) - 1
        if weight is None:
            weight = np.ones(n)
        self.weight = np.ascontiguousarray(weight, dtype=np.float64)
        assert len(self.out_off) == n + 1 and len(self.weight) == n

    @property
    def n_pairs(self):
        return len(self.in_off) - 1

    @classmethod
    def from_lists(cls, pairs, weights=None):
        ins = [np.asarray(p[0], dtype=np.uint32) for p in pairs]
        outs = [np.asarray(p[1], dtype=np.uint32) for p in pairs]
        io = np.concatenate([[0], np.cumsum([len(x) for x in ins])]).astype(np.uint64)
        oo = np.concatenate([[0], np.cumsum([len(x) for x in outs])]).astype(np.uint64)
        isym = np.concatenate(ins) if ins and io[-1] else np.zeros(0, np.uint32)
        osym = np.concatenate(outs) if outs and oo[-1] else np.zeros(0, np.uint32)
        return cls(io, isym, oo, osym, weights)

    def subset(self, idx):
        idx = np.asarray(idx)
        return Corpus.from_lists(
            [(self.in_sym[int(self.in_off[i]):int(self.in_off[i + 1])],
              self.out_sym[int(self.out_off[i]):int(self.out_off[i + 1])]) for i in idx], self.weight[idx])

    def shard(self, rank, world):
        """contiguous block split of the pairs (corpus sharding for data-parallel EM)"""
        n = self.n_pairs
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        io = self.in_off[lo:hi + 1] - self.in_off[lo]
        oo = self.out_off[lo:hi + 1] - self.out_off[lo]
        return Corpus(io, self.in_sym[int(self.in_off[lo]):int(self.in_off[hi])], oo,
                      self.out_sym[int(self.out_off[lo]):int(self.out_off[hi])], self.weight[lo:hi])

    def stats(self, keep=None):
        """n_pairs, totalEmpiricalWeight, n_input, n_output over the kept pairs (train.h:151-168)"""
        li = np.diff(self.in_off.astype(np.int64))
        lo = np.diff(self.out_off.astype(np.int64))
        if keep is None:
            keep = np.ones(self.n_pairs, dtype=bool)
        keep = np.asarray(keep, dtype=bool)
        return dict(n_pairs=int(keep.sum()), total_weight=float(self.weight[keep].sum()),
                    n_input=float(li[keep].sum()), n_output=float(lo[keep].sum()))
