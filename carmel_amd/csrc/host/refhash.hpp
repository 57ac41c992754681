// refhash.hpp -- in which ORDER the reference walks the keys of one of its hash tables.
// carmel numbers a CONDITIONAL transducer's normalisation groups -- the norm ids of its Gibbs sampler (gibbs.cc:114-186), the
// lines of --fem-norm (cascade.h:85-116) -- by walking, state after state, State::index: a HashTable<UnsignedKey, List<HalfArc>>
// keyed by input symbol (fst.h:1362-1446; state.h:158-199).  Group MEMBERSHIP does not depend on that walk, group NUMBERING does,
// and a drop-in has to number them the same way.  This header replays the table's life -- graehl/shared/2hash.h: construction
// with the state's arc count :437-448, one insert per arc in list order :503-517, the doublings :583-600 -- on a flat array of
// (key, next) nodes and reports the keys in the order HashIter (:188-240) visits them.  Hash: uint32_hash of the symbol id
// (state.h:16-22; hash_functions.hpp:239-302, default branch).
#pragma once
#include <cstdint>
#include <vector>

namespace carmel_host {

struct RefKeyWalk {
  struct Node {
    uint32_t key;
    int next;
  };
  std::vector<Node> nodes;
  std::vector<int> head;  // per bucket: first node of its chain, -1 = empty
  unsigned grow_at = 2;

  static uint32_t hash(uint32_t a) {
    a *= 2654435769u;  // golden_ratio_fraction_32
    return a ^ (a >> 16);
  }
  // a table made for `expected` entries: the smallest power of two >= expected, 4 at least (MINHASHSIZE)
  explicit RefKeyWalk(unsigned expected) {
    unsigned n = 2;
    while (n < expected) n <<= 1;
    if (expected < 4) n = 4;
    head.assign(n, -1);
    grow_at = (unsigned)(0.9f * (float)n);
    if (grow_at < 2) grow_at = 2;
  }
  void double_table() {
    const unsigned old_n = (unsigned)head.size(), n = 2 * old_n;
    std::vector<int> old;
    old.swap(head);
    head.assign(n, -1);
    for (unsigned b = 0; b < old_n; ++b)
      for (int p = old[b], nx; p >= 0; p = nx) {  // head to tail; each node moves to the front of its new chain
        nx = nodes[(size_t)p].next;
        const unsigned nb = hash(nodes[(size_t)p].key) & (n - 1);
        nodes[(size_t)p].next = head[nb];
        head[nb] = p;
      }
    grow_at = unsigned((float(grow_at) * (float)n) / (float)old_n) + 1;
  }
  void insert(uint32_t key) {
    const uint32_t hv = hash(key);
    unsigned b = hv & ((unsigned)head.size() - 1);
    for (int p = head[b]; p >= 0; p = nodes[(size_t)p].next)
      if (nodes[(size_t)p].key == key) return;
    if (nodes.size() + 1 >= grow_at) {
      double_table();
      b = hv & ((unsigned)head.size() - 1);
    }
    nodes.push_back(Node{key, head[b]});
    head[b] = (int)nodes.size() - 1;
  }
  std::vector<uint32_t> order() const {  // buckets ascending, a chain head to tail
    std::vector<uint32_t> r;
    for (int h : head)
      for (int p = h; p >= 0; p = nodes[(size_t)p].next) r.push_back(nodes[(size_t)p].key);
    return r;
  }
};

// the input symbols of a state's arcs (in arc-list order) -> the symbols in the order the reference enumerates their groups
inline std::vector<uint32_t> conditional_group_order(const std::vector<uint32_t>& arc_in) {
  RefKeyWalk t((unsigned)arc_in.size());
  for (uint32_t s : arc_in) t.insert(s);
  return t.order();
}

}  // namespace carmel_host
