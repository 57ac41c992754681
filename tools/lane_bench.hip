// Microbenchmark (tools/, not part of the product): build the lane-sweep forward pass up feature by feature on
// synthetic chain lattices to see which ingredient costs what.
//   hipcc --offload-arch=gfx950 -O3 tools/lane_bench.hip -o lane_bench && ./lane_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define NEG_INF (-__builtin_huge_val())
static const uint32_t LAST = 0x80000000u, VALID = 0x40000000u, SMASK = 0x3ffu;

struct Lse {
  double m, acc;
  __device__ __forceinline__ void init() { m = NEG_INF; acc = 0.0; }
  __device__ __forceinline__ void add(double x) {
    if (x == NEG_INF) return;
    if (x <= m) acc += exp(x - m);
    else { acc = (m == NEG_INF) ? 1.0 : acc * exp(m - x) + 1.0; m = x; }
  }
  __device__ __forceinline__ double value() const { return acc == 1.0 ? m : (acc > 0.0 ? m + log(acc) : NEG_INF); }
};

// FEAT bits: 1 chain logic with Lse, 2 LDS column, 4 gather (else sequential weights), 8 wcache store
template <int FEAT, int R>
__global__ __launch_bounds__(64) void fwd_k(const uint2* __restrict__ recs, const double* __restrict__ logw,
                                             const double* __restrict__ seqw, double* __restrict__ wcache,
                                             double* __restrict__ out, uint32_t rows) {
  extern __shared__ double lds[];
  constexpr int U = 4, W = R / 2;
  const int lane = threadIdx.x;
  const uint2* f = recs + (size_t)blockIdx.x * rows * 64 + lane;
  const double* sw = seqw + (size_t)blockIdx.x * rows * 64 + lane;
  double* wc = wcache + (size_t)blockIdx.x * rows * 64 + lane;
  double* col = lds + lane;
  const uint32_t lastk = rows - 1;
  col[0] = 0.0;
  uint2 rq[R][U];
  double wq[R][U];
#pragma unroll
  for (int j = 0; j < R; ++j)
#pragma unroll
    for (int u = 0; u < U; ++u) { uint32_t k = j * U + u; rq[j][u] = f[(size_t)(k < rows ? k : lastk) * 64]; }
#pragma unroll
  for (int j = 0; j < W; ++j)
#pragma unroll
    for (int u = 0; u < U; ++u) wq[j][u] = (FEAT & 16) ? wc[(size_t)((rq[j][u].x >> 10) & 0xfffff) * 64] : (FEAT & 4) ? logw[rq[j][u].y] : sw[(size_t)(j * U + u) * 64];
  Lse acc; acc.init();
  uint32_t d = 1;
  double prev = 0.0, sum = 0.0;
  for (uint32_t k0 = 0; k0 + R * U <= rows; k0 += R * U) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t kb = k0 + j * U;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t kk = kb + W * U + u;
        wq[(j + W) % R][u] = (FEAT & 16) ? wc[(size_t)((rq[(j + W) % R][u].x >> 10) & 0xfffff) * 64] : (FEAT & 4) ? logw[rq[(j + W) % R][u].y] : sw[(size_t)(kk < rows ? kk : lastk) * 64];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t x = rq[j][u].x;
        const double w = wq[j][u];
        if (FEAT & 8) wc[(size_t)((x >> 10) & 0xfffff) * 64] = w;
        if (FEAT & 1) {
          const uint32_t src = x & SMASK;
          const double a_src = (src + 1 == d || !(FEAT & 2)) ? prev : col[src * 64];
          acc.add((x & VALID) ? a_src + w : NEG_INF);
          if (x & LAST) {
            prev = acc.value();
            if (FEAT & 2) col[d * 64] = prev;
            ++d;
            acc.init();
          }
        } else {
          sum += w + (double)x;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { uint32_t k = kb + R * U + u; rq[j][u] = f[(size_t)(k < rows ? k : lastk) * 64]; }
    }
  }
  out[(size_t)blockIdx.x * 64 + lane] = prev + sum;
}

// backward pass: x words (4 B) + weights (8 B) streamed, beta chain, posterior exp + store
template <int FEAT, int R>
__global__ __launch_bounds__(64) void bwd_k(const uint32_t* __restrict__ xs, const double* __restrict__ wcache,
                                             double* __restrict__ post, double* __restrict__ out, uint32_t rows) {
  extern __shared__ double lds[];
  constexpr int U = 4;
  const int lane = threadIdx.x;
  const uint32_t* b = xs + (size_t)blockIdx.x * rows * 64 + lane;
  const double* wc = wcache + (size_t)blockIdx.x * rows * 64 + lane;
  double* po = post + (size_t)blockIdx.x * rows * 64 + lane;
  double* col = lds + lane;
  const uint32_t lastk = rows - 1;
  uint32_t xq[R][U];
  double wq[R][U];
#pragma unroll
  for (int j = 0; j < R; ++j)
#pragma unroll
    for (int u = 0; u < U; ++u) { uint32_t k = j * U + u; size_t kk = (size_t)(k < rows ? k : lastk) * 64; xq[j][u] = b[kk]; wq[j][u] = wc[kk]; }
  Lse acc; acc.init();
  uint32_t s = rows - 1;
  double next = 0.0, al = -1.0;
  for (uint32_t k0 = 0; k0 + R * U <= rows; k0 += R * U) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t kb = k0 + j * U;
      double arg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t x = xq[j][u];
        const uint32_t dst = x & SMASK;
        const double b_dst = (dst == s + 1 || !(FEAT & 2)) ? next : col[dst * 64];
        const double t = (x & VALID) ? wq[j][u] + b_dst : NEG_INF;
        acc.add(t);
        arg[u] = al + t;
        if (x & LAST) {
          next = acc.value();
          if (FEAT & 2) col[s * 64] = next;
          acc.init();
          if (s > 0) { --s; if (FEAT & 2) al = col[s * 64]; }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double pu = (FEAT & 1) ? exp(arg[u]) : arg[u];
        if (FEAT & 8) po[(size_t)(kb + u) * 64] = pu; else next += pu * 1e-300;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { uint32_t k = kb + R * U + u; size_t kk = (size_t)(k < rows ? k : lastk) * 64; xq[j][u] = b[kk]; wq[j][u] = wc[kk]; }
    }
  }
  out[(size_t)blockIdx.x * 64 + lane] = next;
}

// forward (weights from wcache at the record's position) and backward in ONE kernel, as the product kernel does
template <int R>
__global__ __launch_bounds__(64) void fused_k(const uint2* __restrict__ recs, const uint32_t* __restrict__ xs,
                                               const double* __restrict__ wcache, double* __restrict__ post,
                                               double* __restrict__ out, uint32_t rows) {
  extern __shared__ double lds[];
  constexpr int U = 4, W = R / 2;
  const int lane = threadIdx.x;
  const uint2* f = recs + (size_t)blockIdx.x * rows * 64 + lane;
  const uint32_t* b = xs + (size_t)blockIdx.x * rows * 64 + lane;
  const double* wc = wcache + (size_t)blockIdx.x * rows * 64 + lane;
  double* po = post + (size_t)blockIdx.x * rows * 64 + lane;
  double* col = lds + lane;
  const uint32_t lastk = rows - 1;
  col[0] = 0.0;
  double prev = 0.0;
  {
    uint2 rq[R][U];
    double wq[R][U];
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u) { uint32_t k = j * U + u; rq[j][u] = f[(size_t)(k < rows ? k : lastk) * 64]; }
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u) wq[j][u] = wc[(size_t)((rq[j][u].x >> 10) & 0xfffff) * 64];
    Lse acc; acc.init();
    uint32_t d = 1;
    for (uint32_t k0 = 0; k0 + R * U <= rows; k0 += R * U) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const uint32_t kb = k0 + j * U;
#pragma unroll
        for (int u = 0; u < U; ++u) wq[(j + W) % R][u] = wc[(size_t)((rq[(j + W) % R][u].x >> 10) & 0xfffff) * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t x = rq[j][u].x;
          const double w = wq[j][u];
          const uint32_t src = x & SMASK;
          const double a_src = (src + 1 == d) ? prev : col[src * 64];
          acc.add((x & VALID) ? a_src + w : NEG_INF);
          if (x & LAST) { prev = acc.value(); col[d * 64] = prev; ++d; acc.init(); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { uint32_t k = kb + R * U + u; rq[j][u] = f[(size_t)(k < rows ? k : lastk) * 64]; }
      }
    }
  }
  double next = -prev;
  col[rows * 64] = next;
  {
    uint32_t xq[R][U];
    double wq[R][U];
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u) { uint32_t k = j * U + u; size_t kk = (size_t)(k < rows ? k : lastk) * 64; xq[j][u] = b[kk]; wq[j][u] = wc[kk]; }
    Lse acc; acc.init();
    uint32_t s = rows - 1;
    double al = col[s * 64];
    for (uint32_t k0 = 0; k0 + R * U <= rows; k0 += R * U) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const uint32_t kb = k0 + j * U;
        double arg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t x = xq[j][u];
          const uint32_t dst = x & SMASK;
          const double b_dst = (dst == s + 1) ? next : col[dst * 64];
          const double t = (x & VALID) ? wq[j][u] + b_dst : NEG_INF;
          acc.add(t);
          arg[u] = al + t;
          if (x & LAST) { next = acc.value(); col[s * 64] = next; acc.init(); if (s > 0) { --s; al = col[s * 64]; } }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) po[(size_t)(kb + u) * 64] = exp(arg[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) { uint32_t k = kb + R * U + u; size_t kk = (size_t)(k < rows ? k : lastk) * 64; xq[j][u] = b[kk]; wq[j][u] = wc[kk]; }
      }
    }
  }
  out[(size_t)blockIdx.x * 64 + lane] = next;
}

int main() {
  const uint32_t rows = 32, NB = 15625;  // 1M lattices of 32 arcs
  const size_t N = (size_t)NB * rows * 64, T = 10000000;
  std::vector<uint2> h(N);
  std::mt19937 rng(1);
  for (uint32_t b = 0; b < NB; ++b)
    for (uint32_t k = 0; k < rows; ++k)
      for (uint32_t l = 0; l < 64; ++l)
        h[((size_t)b * rows + k) * 64 + l] = make_uint2(k | ((rows - 1 - k) << 10) | VALID | LAST, rng() % T);
  uint2* recs; double *logw, *seqw, *wc, *out;
  CK(hipMalloc(&recs, N * 8)); CK(hipMalloc(&logw, T * 8)); CK(hipMalloc(&seqw, N * 8)); CK(hipMalloc(&wc, N * 8));
  CK(hipMalloc(&out, (size_t)NB * 64 * 8));
  CK(hipMemcpy(recs, h.data(), N * 8, hipMemcpyHostToDevice));
  CK(hipMemset(logw, 0, T * 8)); CK(hipMemset(seqw, 0, N * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s %8.3f ms  %6.1f G records/s\n", name, ms / 5, N / (ms / 5 * 1e-3) / 1e9);
    CK(hipGetLastError());
  };
  const size_t lds = 33 * 512;
#define RUN(FEAT, R, label) time(label, [&] { hipLaunchKernelGGL((fwd_k<FEAT, R>), dim3(NB), dim3(64), lds, 0, recs, logw, seqw, wc, out, rows); });
  RUN(0, 4, "records + sequential weights, plain sum        R=4")
  RUN(1, 4, "+ chain logic (Lse, register bypass only)      R=4")
  RUN(3, 4, "+ LDS column                                   R=4")
  RUN(7, 4, "+ gather of weights                            R=4")
  RUN(15, 4, "+ wcache store (coalesced, reversed row)       R=4")
  RUN(4, 4, "records + gather, plain sum                    R=4")
  RUN(4, 8, "records + gather, plain sum                    R=8")
  RUN(19, 4, "records + weights from wcache at record pos (reversed) R=4")
  {
    std::vector<uint32_t> hx(N);
    for (uint32_t b = 0; b < NB; ++b) for (uint32_t k = 0; k < rows; ++k) for (uint32_t l = 0; l < 64; ++l)
      hx[((size_t)b * rows + k) * 64 + l] = (rows - 1 - k + 1 > rows - 1 ? rows - 1 : rows - k) | VALID | LAST;
    uint32_t* xs; double* post;
    CK(hipMalloc(&xs, N * 4)); CK(hipMalloc(&post, N * 8));
    CK(hipMemcpy(xs, hx.data(), N * 4, hipMemcpyHostToDevice));
#define RUNB(FEAT, R, label) time(label, [&] { hipLaunchKernelGGL((bwd_k<FEAT, R>), dim3(NB), dim3(64), lds, 0, xs, seqw, post, out, rows); });
    RUNB(0, 4, "bwd: x + weights streamed, chain only (no LDS/exp/store)  R=4")
    RUNB(2, 4, "bwd: + LDS column                                         R=4")
    RUNB(3, 4, "bwd: + exp                                                R=4")
    RUNB(11, 4, "bwd: + posterior store                                    R=4")
    RUNB(8, 4, "bwd: store without exp/LDS                                R=4")
    time("fused forward(wcache) + backward, one kernel              R=4", [&] { hipLaunchKernelGGL((fused_k<4>), dim3(NB), dim3(64), lds + 512, 0, recs, xs, seqw, post, out, rows); });
    time("fused forward(wcache) + backward, one kernel              R=2", [&] { hipLaunchKernelGGL((fused_k<2>), dim3(NB), dim3(64), lds + 512, 0, recs, xs, seqw, post, out, rows); });
  }
  RUN(15, 2, "everything                                     R=2")
  RUN(15, 8, "everything                                     R=8")
  return 0;
}
