// refhash.hpp (oracle) -- TEST INFRASTRUCTURE: the iteration order of the reference's chained hash table, restated.
// graehl/shared/2hash.h: init :437-448 (power-of-two bucket count, at least MINHASHSIZE = 4, growAt = (unsigned)(0.9f * buckets),
// at least 2), insert :503-517 (a new key becomes the HEAD of its bucket's chain; the table doubles when the entry count reaches
// growAt), rehash_pow2 :583-600 (old buckets in order, each chain head to tail, every node pushed onto the head of its new chain;
// growAt = unsigned(float(growAt) * new / old) + 1), HashIter :188-240 (buckets in order, a chain head to tail).  Hash of an
// unsigned key (state.h:16-22): uint32_hash, hash_functions.hpp:239-302 in its default branch: a *= 2654435769; a ^= a >> 16.
// carmel's CONDITIONAL normalisation groups of a state are enumerated by walking such a table (fst.h:1362-1446, State::index
// built by State::indexBy, state.h:158-199: `NEW Index(size)`, then index[arc.in].push_front(arc) in arc-list order).
#pragma once
#include <cstdint>
#include <list>
#include <vector>

namespace oracle {

inline uint32_t ref_uint32_hash(uint32_t a) {
  a *= 2654435769u;
  a ^= a >> 16;
  return a;
}

class RefHashKeys {  // keys only
  std::vector<std::list<uint32_t> > table_;
  unsigned cnt_ = 0, grow_at_ = 2;
  static unsigned pow2_bound(unsigned request) {  // 2hash.h:68-74
    unsigned mask = 2;
    for (; mask < request; mask <<= 1) {
    }
    return mask;
  }
  void grow() {
    std::vector<std::list<uint32_t> > old;
    old.swap(table_);
    const unsigned old_n = (unsigned)old.size(), n = 2 * old_n;
    table_.assign(n, std::list<uint32_t>());
    for (auto& chain : old)
      for (uint32_t k : chain) table_[ref_uint32_hash(k) & (n - 1)].push_front(k);
    grow_at_ = unsigned((float(grow_at_) * n) / old_n) + 1;
  }

 public:
  explicit RefHashKeys(unsigned sz = 8) {
    const unsigned n = sz < 4 ? 4 : pow2_bound(sz);
    grow_at_ = (unsigned)(0.9f * n);
    if (grow_at_ < 2) grow_at_ = 2;
    table_.assign(n, std::list<uint32_t>());
  }
  bool insert(uint32_t key) {
    const uint32_t hv = ref_uint32_hash(key);
    size_t b = hv & (table_.size() - 1);
    for (uint32_t k : table_[b])
      if (k == key) return false;
    if (++cnt_ >= grow_at_) {
      grow();
      b = hv & (table_.size() - 1);
    }
    table_[b].push_front(key);
    return true;
  }
  std::vector<uint32_t> keys() const {
    std::vector<uint32_t> r;
    for (auto& chain : table_)
      for (uint32_t k : chain) r.push_back(k);
    return r;
  }
};

}  // namespace oracle
