// compose.hip — the product construction of WFST composition on the GPU.
//
// What it computes is WFST::set_compose with the default 3-state epsilon filter
// (/root/reference/carmel/src/compose.cc:163-531; filter states :315-324):
//   composite state (qa, qb, filter); from it, in the reference's own emission order (which of the three code paths
//   runs depends on the sizes of the two operand states and carmel -T, compose.cc:330-498):
//     a:x of A with x:c of B            -> a:c   to (qa', qb', 0), weight wa * wb          ("both", incl. x = *e* on both)
//     a:*e* of A alone (filter != 2)    -> a:*e* to (qa', qb , 1)
//     *e*:c of B alone (filter != 1)    -> *e*:c to (qa , qb', 2)
// How: level-synchronous frontier expansion.  Every composite state of the frontier is expanded by one thread that
// walks the outer operand state's arcs in list order and looks the matching arcs of the other state up in a per-state
// symbol index (binary search); destinations are looked up / inserted in a device hash table (64-bit keys, atomicCAS),
// which hands out temporary state ids.  Two passes per level -- count, exclusive scan (hipcub), emit -- give every state
// a contiguous run of arcs in emission order.  Each arc carries its PROVENANCE (which arc of A and/or of B it was built
// from): that is what cascade_parameters::record / record1 / record2 receive (cascade.h:507-599).
//
// What stays on the host (host/compose.hpp, Composer::from_device): the reference numbers composite states in the order
// a LIFO work list discovers them (compose.cc:193, 326-328) and creates chain ids in emission order -- both are
// sequential definitions, O(arcs) list walks over the device's output; the arc matching, the filter logic, the weights
// and the state discovery -- the part that grows with the product of the operands -- happen here.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <memory>
#include <type_traits>
#include <cstring>
#include <string>
#include <vector>
#include "engine.hpp"

namespace {

#define C_NONE 0xffffffffu

struct Operand {           // one transducer in CSR, arcs in list (file) order, plus the per-state symbol index
  const uint64_t* off;     // n_states + 1
  const uint32_t* in;
  const uint32_t* out;
  const uint32_t* dst;
  const double* logw;
  const uint32_t* ix_sym;  // per state (same offsets): the join symbol (A: output, B: input) ascending ...
  const uint32_t* ix_pos;  // ... and within a symbol the arc position (within the state) DESCENDING: the reference's
                           // per-symbol lists are built by push_front (state.h:158-199)
  uint32_t n_states;
};

struct Table {             // open addressing, key 0 = empty
  unsigned long long* keys;
  uint32_t* vals;
  uint64_t mask;
};

struct ComposeArgs {
  Operand A, B;
  const uint32_t* a2b;     // A output symbol -> B input symbol (C_NONE: no such symbol in B)
  const uint32_t* b2a;
  uint32_t n_a2b, n_b2a, threshold;
  // frontier
  uint32_t lo, hi;
  // composite states by temporary id
  uint32_t* st_qa;
  uint32_t* st_qb;
  uint8_t* st_f;
  uint64_t* st_off;        // first arc of the state
  // arcs
  uint64_t arc_base;       // first arc of this level
  const uint64_t* level_off;  // exclusive scan of the frontier's arc counts
  uint32_t* arc_in;
  uint32_t* arc_out;
  uint32_t* arc_dst;
  double* arc_logw;
  uint32_t* arc_ka;
  uint32_t* arc_kb;
  uint64_t* counts;        // count pass: arcs per frontier state
  Table table;
  uint32_t* n_states;      // device counter of composite states
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// [first, last) of the entries of state q whose join symbol is sym, in the state's symbol index
__device__ __forceinline__ void bucket_of(const Operand& X, uint32_t q, uint32_t sym, uint64_t& first, uint64_t& last) {
  uint64_t lo = X.off[q], hi = X.off[q + 1];
  if (sym == C_NONE) {
    first = last = lo;
    return;
  }
  uint64_t a = lo, b = hi;
  while (a < b) {  // lower bound
    const uint64_t m = (a + b) >> 1;
    if (X.ix_sym[m] < sym) a = m + 1; else b = m;
  }
  first = a;
  b = hi;
  while (a < b) {  // upper bound
    const uint64_t m = (a + b) >> 1;
    if (X.ix_sym[m] <= sym) a = m + 1; else b = m;
  }
  last = a;
}

// the expansion of one composite state in the reference's order; E(in, out, qa', qb', filter', logw, ka, kb)
template <class E>
__device__ __forceinline__ void expand(const ComposeArgs& G, uint32_t qa, uint32_t qb, int f, E&& emit) {
  const Operand& A = G.A;
  const Operand& B = G.B;
  const uint64_t a0 = A.off[qa], a1 = A.off[qa + 1], b0 = B.off[qb], b1 = B.off[qb + 1];
  const uint64_t na = a1 - a0, nb = b1 - b0;
  const bool a_bigger = na > nb;
  const uint64_t big = a_bigger ? na : nb;
  uint64_t e0, e1;
  if (!(big > G.threshold && a_bigger)) {
    // A walked in list order; B's matches through its index -- newest first when the reference itself uses the index
    // (compose.cc:339-385), in list order when it scans (both states small, :437-487)
    const bool newest_first = big > G.threshold;
    uint64_t be0, be1;
    bucket_of(B, qb, 0u, be0, be1);
    for (uint64_t ka = a0; ka < a1; ++ka) {
      const uint32_t o = A.out[ka];
      if (o == 0u) {
        if (f != 2) emit(A.in[ka], 0u, A.dst[ka], qb, 1, A.logw[ka], (uint32_t)(ka - a0), C_NONE);
        if (f == 0)
          for (uint64_t j = 0; j < be1 - be0; ++j) {
            const uint64_t kb = b0 + B.ix_pos[newest_first ? be0 + j : be1 - 1 - j];
            emit(A.in[ka], B.out[kb], A.dst[ka], B.dst[kb], 0, A.logw[ka] + B.logw[kb], (uint32_t)(ka - a0), (uint32_t)(kb - b0));
          }
      } else {
        bucket_of(B, qb, o < G.n_a2b ? G.a2b[o] : C_NONE, e0, e1);
        for (uint64_t j = 0; j < e1 - e0; ++j) {
          const uint64_t kb = b0 + B.ix_pos[newest_first ? e0 + j : e1 - 1 - j];
          emit(A.in[ka], B.out[kb], A.dst[ka], B.dst[kb], 0, A.logw[ka] + B.logw[kb], (uint32_t)(ka - a0), (uint32_t)(kb - b0));
        }
      }
    }
    if (f != 1)
      for (uint64_t j = 0; j < be1 - be0; ++j) {
        const uint64_t kb = b0 + B.ix_pos[newest_first ? be0 + j : be1 - 1 - j];
        emit(0u, B.out[kb], qa, B.dst[kb], 2, B.logw[kb], C_NONE, (uint32_t)(kb - b0));
      }
  } else {
    // A is the larger, indexed state: B walked in list order, A's matches newest first (compose.cc:386-436)
    uint64_t ae0, ae1;
    bucket_of(A, qa, 0u, ae0, ae1);
    for (uint64_t kb = b0; kb < b1; ++kb) {
      const uint32_t i = B.in[kb];
      if (i == 0u) {
        if (f != 1) emit(0u, B.out[kb], qa, B.dst[kb], 2, B.logw[kb], C_NONE, (uint32_t)(kb - b0));
        if (f == 0)
          for (uint64_t j = ae0; j < ae1; ++j) {
            const uint64_t ka = a0 + A.ix_pos[j];
            emit(A.in[ka], B.out[kb], A.dst[ka], B.dst[kb], 0, A.logw[ka] + B.logw[kb], (uint32_t)(ka - a0), (uint32_t)(kb - b0));
          }
      } else {
        bucket_of(A, qa, i < G.n_b2a ? G.b2a[i] : C_NONE, e0, e1);
        for (uint64_t j = e0; j < e1; ++j) {
          const uint64_t ka = a0 + A.ix_pos[j];
          emit(A.in[ka], B.out[kb], A.dst[ka], B.dst[kb], 0, A.logw[ka] + B.logw[kb], (uint32_t)(ka - a0), (uint32_t)(kb - b0));
        }
      }
    }
    if (f != 2)
      for (uint64_t j = ae0; j < ae1; ++j) {
        const uint64_t ka = a0 + A.ix_pos[j];
        emit(A.in[ka], 0u, A.dst[ka], qb, 1, A.logw[ka], (uint32_t)(ka - a0), C_NONE);
      }
  }
}

__global__ __launch_bounds__(256) void compose_count_kernel(ComposeArgs G) {
  const uint32_t s = G.lo + blockIdx.x * 256 + threadIdx.x;
  if (s >= G.hi) return;
  uint64_t n = 0;
  expand(G, G.st_qa[s], G.st_qb[s], (int)G.st_f[s], [&](uint32_t, uint32_t, uint32_t, uint32_t, int, double, uint32_t, uint32_t) { ++n; });
  G.counts[s - G.lo] = n;
}

// the temporary id of composite state (qa, qb, f): looked up, or inserted (the inserting thread draws the next id)
__device__ __forceinline__ uint32_t state_id(const ComposeArgs& G, uint32_t qa, uint32_t qb, int f) {
  const unsigned long long key = ((unsigned long long)qa * G.B.n_states + qb) * 3ull + (unsigned long long)f + 1ull;
  uint64_t h = mix64(key) & G.table.mask;
  // No lane ever spins on another lane's progress inside a branch: a lane that finds the key present but its id not
  // yet published just goes round the (wave-uniform) loop again, and the inserting lane publishes within one pass of
  // the loop body -- wave64 lanes run in lockstep, an inner wait loop could starve the very lane it waits for.
  for (;;) {
    unsigned long long cur = G.table.keys[h];
    if (cur == 0ull) {
      cur = atomicCAS(G.table.keys + h, 0ull, key);
      if (cur == 0ull) {  // ours
        const uint32_t id = atomicAdd(G.n_states, 1u);
        G.st_qa[id] = qa;
        G.st_qb[id] = qb;
        G.st_f[id] = (uint8_t)f;
        __threadfence();
        atomicExch(G.table.vals + h, id);
        return id;
      }
    }
    if (cur == key) {
      const uint32_t v = atomicAdd(G.table.vals + h, 0u);
      if (v != C_NONE) return v;
      continue;  // the inserter has not published the id yet: look again
    }
    h = (h + 1) & G.table.mask;
  }
}

__global__ __launch_bounds__(256) void compose_emit_kernel(ComposeArgs G) {
  const uint32_t s = G.lo + blockIdx.x * 256 + threadIdx.x;
  if (s >= G.hi) return;
  uint64_t at = G.arc_base + G.level_off[s - G.lo];
  G.st_off[s] = at;
  expand(G, G.st_qa[s], G.st_qb[s], (int)G.st_f[s],
         [&](uint32_t in, uint32_t out, uint32_t qa, uint32_t qb, int f, double lw, uint32_t ka, uint32_t kb) {
           G.arc_in[at] = in;
           G.arc_out[at] = out;
           G.arc_logw[at] = lw;
           G.arc_ka[at] = ka;
           G.arc_kb[at] = kb;
           G.arc_dst[at] = state_id(G, qa, qb, f);
           ++at;
         });
}

__global__ void table_reinsert_kernel(Table T, const uint32_t* st_qa, const uint32_t* st_qb, const uint8_t* st_f, uint32_t n,
                                      uint32_t nb) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n) return;
  const unsigned long long key = ((unsigned long long)st_qa[id] * nb + st_qb[id]) * 3ull + (unsigned long long)st_f[id] + 1ull;
  uint64_t h = mix64(key) & T.mask;
  for (;;) {
    if (atomicCAS(T.keys + h, 0ull, key) == 0ull) {
      T.vals[h] = id;
      return;
    }
    h = (h + 1) & T.mask;
  }
}

template <class T>
hipError_t grow(DevBuf<T>& b, size_t need, size_t used, hipStream_t s) {
  if (need <= b.n) return hipSuccess;
  size_t cap = b.n ? b.n : 1024;
  while (cap < need) cap *= 2;
  T* np = nullptr;
  hipError_t e = hipMalloc((void**)&np, cap * sizeof(T));
  if (e != hipSuccess) return e;
  if (used) e = hipMemcpyAsync(np, b.p, used * sizeof(T), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (b.p) (void)hipFree(b.p);
  b.p = np;
  b.n = cap;
  return e;
}

}  // namespace

struct carmel_hip_composition {
  int device = 0;
  uint32_t n_states = 0;
  uint64_t n_arcs = 0;
  DevBuf<uint32_t> st_qa, st_qb, arc_in, arc_out, arc_dst, arc_ka, arc_kb;
  DevBuf<uint8_t> st_f;
  DevBuf<uint64_t> st_off;
  DevBuf<double> arc_logw;
  double seconds = 0;
  uint32_t levels = 0;
};

// per state, arc positions sorted by (join symbol ascending, position descending)
static void symbol_index(uint32_t n_states, const uint64_t* off, const uint32_t* sym, std::vector<uint32_t>& ix_sym,
                         std::vector<uint32_t>& ix_pos) {
  const uint64_t n = off[n_states];
  ix_sym.resize(n);
  ix_pos.resize(n);
  std::vector<std::pair<uint32_t, uint32_t> > tmp;
  for (uint32_t q = 0; q < n_states; ++q) {
    const uint64_t a = off[q], b = off[q + 1];
    tmp.clear();
    for (uint64_t k = a; k < b; ++k) tmp.emplace_back(sym[k], (uint32_t)(k - a));
    std::sort(tmp.begin(), tmp.end(), [](const std::pair<uint32_t, uint32_t>& x, const std::pair<uint32_t, uint32_t>& y) {
      return x.first != y.first ? x.first < y.first : x.second > y.second;
    });
    for (uint64_t k = a; k < b; ++k) {
      ix_sym[k] = tmp[k - a].first;
      ix_pos[k] = tmp[k - a].second;
    }
  }
}

extern "C" {

int carmel_hip_compose(carmel_hip_composition** out, int device, uint32_t a_states, const uint64_t* a_off, const uint32_t* a_in,
                       const uint32_t* a_out, const uint32_t* a_dst, const double* a_logw, uint32_t b_states,
                       const uint64_t* b_off, const uint32_t* b_in, const uint32_t* b_out, const uint32_t* b_dst,
                       const double* b_logw, const uint32_t* a2b, uint32_t n_a2b, const uint32_t* b2a, uint32_t n_b2a,
                       uint32_t index_threshold) {
  if (!out || !a_off || !b_off || !a2b || !b2a || !a_states || !b_states) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail(CARMEL_HIP_ERR_HIP, "no HIP device: composition on the GPU has no CPU fallback (the front end composes "
                                                 "on the host unless asked to use the GPU)");
  HIPCHK(hipSetDevice(device));
  const auto t0 = std::chrono::steady_clock::now();
  hipStream_t s = nullptr;
  const uint64_t na = a_off[a_states], nb = b_off[b_states];
  std::vector<uint32_t> asym, apos, bsym, bpos;
  symbol_index(a_states, a_off, a_out, asym, apos);
  symbol_index(b_states, b_off, b_in, bsym, bpos);
  DevBuf<uint64_t> d_aoff, d_boff;
  DevBuf<uint32_t> d_ain, d_aout, d_adst, d_bin, d_bout, d_bdst, d_asym, d_apos, d_bsym, d_bpos, d_a2b, d_b2a;
  DevBuf<double> d_alw, d_blw;
#define UP(buf, ptr, n) HIPCHK((buf).upload(std::vector<typename std::remove_pointer<decltype((buf).p)>::type>((ptr), (ptr) + (n)), s))
  UP(d_aoff, a_off, (size_t)a_states + 1);
  UP(d_boff, b_off, (size_t)b_states + 1);
  UP(d_ain, a_in, na);
  UP(d_aout, a_out, na);
  UP(d_adst, a_dst, na);
  UP(d_alw, a_logw, na);
  UP(d_bin, b_in, nb);
  UP(d_bout, b_out, nb);
  UP(d_bdst, b_dst, nb);
  UP(d_blw, b_logw, nb);
  UP(d_a2b, a2b, n_a2b);
  UP(d_b2a, b2a, n_b2a);
#undef UP
  HIPCHK(d_asym.upload(asym, s));
  HIPCHK(d_apos.upload(apos, s));
  HIPCHK(d_bsym.upload(bsym, s));
  HIPCHK(d_bpos.upload(bpos, s));
  HIPCHK(hipStreamSynchronize(s));
  std::unique_ptr<carmel_hip_composition> C(new carmel_hip_composition());
  C->device = device;
  ComposeArgs G;
  std::memset(&G, 0, sizeof G);
  G.A = Operand{d_aoff.p, d_ain.p, d_aout.p, d_adst.p, d_alw.p, d_asym.p, d_apos.p, a_states};
  G.B = Operand{d_boff.p, d_bin.p, d_bout.p, d_bdst.p, d_blw.p, d_bsym.p, d_bpos.p, b_states};
  G.a2b = d_a2b.p;
  G.b2a = d_b2a.p;
  G.n_a2b = n_a2b;
  G.n_b2a = n_b2a;
  G.threshold = index_threshold;
  DevBuf<unsigned long long> keys;
  DevBuf<uint32_t> vals, counter;
  DevBuf<uint64_t> counts, level_off;
  DevBuf<char> scan_tmp;
  HIPCHK(counter.alloc(1));
  uint64_t cap = 1ull << 16;
  auto rebuild_table = [&](uint64_t want, uint32_t n_states) -> int {
    while (cap < want) cap <<= 1;
    HIPCHK(keys.alloc(cap));
    HIPCHK(vals.alloc(cap));
    HIPCHK(hipMemsetAsync(keys.p, 0, cap * sizeof(unsigned long long), s));
    HIPCHK(hipMemsetAsync(vals.p, 0xff, cap * sizeof(uint32_t), s));
    G.table = Table{keys.p, vals.p, cap - 1};
    if (n_states)
      hipLaunchKernelGGL(table_reinsert_kernel, dim3((n_states + 255) / 256), dim3(256), 0, s, G.table, C->st_qa.p, C->st_qb.p,
                         C->st_f.p, n_states, b_states);
    HIPCHK(hipGetLastError());
    return CARMEL_HIP_OK;
  };
  // state 0 = (0, 0, 0)
  HIPCHK(grow(C->st_qa, 1024, 0, s));
  HIPCHK(grow(C->st_qb, 1024, 0, s));
  HIPCHK(grow(C->st_f, 1024, 0, s));
  HIPCHK(grow(C->st_off, 1025, 0, s));
  HIPCHK(hipMemsetAsync(C->st_qa.p, 0, 4, s));
  HIPCHK(hipMemsetAsync(C->st_qb.p, 0, 4, s));
  HIPCHK(hipMemsetAsync(C->st_f.p, 0, 1, s));
  uint32_t n_states = 1;
  HIPCHK(hipMemcpyAsync(counter.p, &n_states, 4, hipMemcpyHostToDevice, s));
  {
    int rc = rebuild_table(cap, 1);
    if (rc) return rc;
  }
  uint32_t lo = 0, hi = 1;
  uint64_t n_arcs = 0;
  while (lo < hi) {
    const uint32_t nf = hi - lo;
    if (counts.n < nf) HIPCHK(counts.alloc((size_t)nf * 2));
    if (level_off.n < nf) HIPCHK(level_off.alloc((size_t)nf * 2));
    G.lo = lo;
    G.hi = hi;
    G.st_qa = C->st_qa.p;
    G.st_qb = C->st_qb.p;
    G.st_f = C->st_f.p;
    G.counts = counts.p;
    hipLaunchKernelGGL(compose_count_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, G);
    size_t tmp_bytes = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts.p, level_off.p, (int)nf, s));
    if (scan_tmp.n < tmp_bytes) HIPCHK(scan_tmp.alloc(tmp_bytes));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, tmp_bytes, counts.p, level_off.p, (int)nf, s));
    uint64_t last_off = 0, last_cnt = 0;
    HIPCHK(hipMemcpyAsync(&last_off, level_off.p + (nf - 1), 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&last_cnt, counts.p + (nf - 1), 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const uint64_t level_arcs = last_off + last_cnt;
    if (n_arcs + level_arcs >= 0xfffffff0ull) return fail(CARMEL_HIP_ERR_UNSUPPORTED, "composition has more than 2^32 arcs");
    // room for this level's arcs, for the states they may discover, and a table at most half full
    HIPCHK(grow(C->arc_in, n_arcs + level_arcs, n_arcs, s));
    HIPCHK(grow(C->arc_out, n_arcs + level_arcs, n_arcs, s));
    HIPCHK(grow(C->arc_dst, n_arcs + level_arcs, n_arcs, s));
    HIPCHK(grow(C->arc_ka, n_arcs + level_arcs, n_arcs, s));
    HIPCHK(grow(C->arc_kb, n_arcs + level_arcs, n_arcs, s));
    HIPCHK(grow(C->arc_logw, n_arcs + level_arcs, n_arcs, s));
    const uint64_t max_states = (uint64_t)n_states + level_arcs;
    HIPCHK(grow(C->st_qa, max_states, n_states, s));
    HIPCHK(grow(C->st_qb, max_states, n_states, s));
    HIPCHK(grow(C->st_f, max_states, n_states, s));
    HIPCHK(grow(C->st_off, max_states + 1, n_states, s));
    if (2 * max_states > cap) {
      int rc = rebuild_table(2 * max_states, n_states);
      if (rc) return rc;
    }
    G.st_qa = C->st_qa.p;
    G.st_qb = C->st_qb.p;
    G.st_f = C->st_f.p;
    G.st_off = C->st_off.p;
    G.arc_base = n_arcs;
    G.level_off = level_off.p;
    G.arc_in = C->arc_in.p;
    G.arc_out = C->arc_out.p;
    G.arc_dst = C->arc_dst.p;
    G.arc_logw = C->arc_logw.p;
    G.arc_ka = C->arc_ka.p;
    G.arc_kb = C->arc_kb.p;
    G.n_states = counter.p;
    hipLaunchKernelGGL(compose_emit_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, G);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&n_states, counter.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    n_arcs += level_arcs;
    lo = hi;
    hi = n_states;
    ++C->levels;
  }
  HIPCHK(hipMemcpyAsync(C->st_off.p + n_states, &n_arcs, 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  C->n_states = n_states;
  C->n_arcs = n_arcs;
  C->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *out = C.release();
  return CARMEL_HIP_OK;
}

uint64_t carmel_hip_composition_states(carmel_hip_composition* c) { return c ? c->n_states : 0; }
uint64_t carmel_hip_composition_arcs(carmel_hip_composition* c) { return c ? c->n_arcs : 0; }
double carmel_hip_composition_seconds(carmel_hip_composition* c) { return c ? c->seconds : 0; }

int carmel_hip_composition_export(carmel_hip_composition* c, uint64_t* state_off, uint32_t* state_qa, uint32_t* state_qb,
                                  uint8_t* state_filter, uint32_t* arc_in, uint32_t* arc_out, uint32_t* arc_dst,
                                  double* arc_logw, uint32_t* arc_ka, uint32_t* arc_kb) {
  if (!c) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  const size_t ns = c->n_states, na = c->n_arcs;
#define DN(dst, buf, n) \
  if (dst) HIPCHK(hipMemcpy(dst, (buf).p, (n) * sizeof(*(buf).p), hipMemcpyDeviceToHost))
  DN(state_off, c->st_off, ns + 1);
  DN(state_qa, c->st_qa, ns);
  DN(state_qb, c->st_qb, ns);
  DN(state_filter, c->st_f, ns);
  DN(arc_in, c->arc_in, na);
  DN(arc_out, c->arc_out, na);
  DN(arc_dst, c->arc_dst, na);
  DN(arc_logw, c->arc_logw, na);
  DN(arc_ka, c->arc_ka, na);
  DN(arc_kb, c->arc_kb, na);
#undef DN
  return CARMEL_HIP_OK;
}

int carmel_hip_composition_free(carmel_hip_composition* c) {
  if (c) {
    (void)hipSetDevice(c->device);
    delete c;
  }
  return CARMEL_HIP_OK;
}

}  // extern "C"
