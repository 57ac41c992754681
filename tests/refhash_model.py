"""A plain-Python model of the reference's chained hash table as far as ITERATION ORDER goes (graehl/shared/2hash.h: init :437-448,
insert :503-517, rehash_pow2 :583-600, HashIter :188-240; hash of an unsigned key: hash_functions.hpp:239-302 in its default
branch, state.h:16-22).  carmel enumerates a CONDITIONAL transducer's normalisation groups by walking such a table per state
(fst.h:1362-1446 over State::index, built by State::indexBy, state.h:158-199), so the ORDER of the groups -- Gibbs norm ids, the
lines of --fem-norm -- is this walk.  Third, independent restatement beside oracle/refhash.hpp and csrc/host/refhash.hpp: the tests
hold the two against it."""
import numpy as np


def uint32_hash(a):
    a = (a * 2654435769) & 0xFFFFFFFF  # golden_ratio_fraction_32 (hash_functions.hpp:41)
    return a ^ (a >> 16)


def pow2bound(request):  # 2hash.h:68-74
    mask = 2
    while mask < request:
        mask <<= 1
    return mask


class RefHashTable:
    """keys only; a chain is a Python list, head first"""

    def __init__(self, sz=8, load=0.9):
        siz = 4 if sz < 4 else pow2bound(sz)  # MINHASHSIZE = 4
        self.cnt = 0
        self.grow_at = max(2, int(np.float32(load) * np.float32(siz)))  # (unsigned)(mLoad * siz), float arithmetic
        self.mask = siz - 1
        self.table = [[] for _ in range(siz)]

    def insert(self, key):
        hv = uint32_hash(key)
        b = hv & self.mask
        if key in self.table[b]:
            return False
        self.cnt += 1
        if self.cnt >= self.grow_at:
            self._grow()
            b = hv & self.mask
        self.table[b].insert(0, key)  # new nodes go to the head of their chain
        return True

    def _grow(self):
        old, old_n = self.table, self.mask + 1
        n = 2 * old_n
        self.mask = n - 1
        self.table = [[] for _ in range(n)]
        for chain in old:  # old buckets in order, each chain head to tail; a moved node becomes the head of its new chain
            for k in chain:
                self.table[uint32_hash(k) & self.mask].insert(0, k)
        self.grow_at = int(np.float32(np.float32(self.grow_at) * np.float32(n)) / np.float32(old_n)) + 1

    def keys(self):  # HashIter: buckets in order, a chain head to tail
        return [k for chain in self.table for k in chain]


def conditional_groups(arcs_in):
    """the (input symbol) groups of one state in NormGroupIter's order, each as the list of arc positions (in the state's arc
    list) in the order the reference walks them: State::indexBy pushes every arc onto the FRONT of its symbol's list"""
    t = RefHashTable(len(arcs_in))
    for sym in arcs_in:
        t.insert(sym)
    out = []
    for sym in t.keys():
        out.append([j for j in range(len(arcs_in) - 1, -1, -1) if arcs_in[j] == sym])
    return out
