// NOTE: written against the lane kernel of profile v9; LaneArgs has grown since (4-byte forward records, wcache) and this
// driver no longer sets every field -- kept for the record of the experiment, rebuild the argument block before use.
// Microbenchmark (tools/, not part of the product): the PRODUCT lane-sweep kernel on the synthetic chain lattices
// of lane_bench.hip (1M lattices x 32 arcs), to separate "kernel code" from "data / allocation environment".
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Icarmel_amd/csrc tools/lane_bench_real.hip -o tools/lane_bench_real
#include "../carmel_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace carmel_hip;
int main() {
  const uint32_t rows = 32, NB = 15625, T = 10000000;
  const size_t N = (size_t)NB * rows * 64;
  std::vector<uint2> hf(N);
  std::vector<uint32_t> hb(N);
  std::mt19937 rng(1);
  for (uint32_t b = 0; b < NB; ++b)
    for (uint32_t k = 0; k < rows; ++k)
      for (uint32_t l = 0; l < 64; ++l) {
        size_t i = ((size_t)b * rows + k) * 64 + l;
        hf[i] = make_uint2(k | ((rows - 1 - k) << LANE_POS_SHIFT) | LANE_VALID | LANE_LAST, rng() % T);  // src = k
        hb[i] = (rows - k) | LANE_VALID | LANE_LAST;  // backward row k: source state rows-1-k, destination rows-k
      }
  std::vector<LaneGroup> hg(NB);
  for (uint32_t b = 0; b < NB; ++b) { hg[b] = LaneGroup{(uint64_t)b * rows * 64, rows, 64, b * 64, rows + 1, 0}; }
  std::vector<uint32_t> hpair((size_t)NB * 64), hns((size_t)NB * 64, rows + 1);
  for (size_t i = 0; i < hpair.size(); ++i) hpair[i] = (uint32_t)i;
  LaneArgs A{};
  uint2* fwd; uint32_t *bwd, *lp, *ns; LaneGroup* gr; double *llw, *logw, *post, *wc, *sc, *plp;
  CK(hipMalloc(&fwd, N * 8)); CK(hipMalloc(&bwd, N * 4)); CK(hipMalloc(&gr, NB * sizeof(LaneGroup)));
  CK(hipMalloc(&lp, hpair.size() * 4)); CK(hipMalloc(&ns, hns.size() * 4)); CK(hipMalloc(&llw, hns.size() * 8));
  CK(hipMalloc(&logw, (size_t)T * 8)); CK(hipMalloc(&post, N * 8)); CK(hipMalloc(&wc, N * 8)); CK(hipMalloc(&sc, 64));
  CK(hipMalloc(&plp, hns.size() * 8));
  CK(hipMemcpy(fwd, hf.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(bwd, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(gr, hg.data(), NB * sizeof(LaneGroup), hipMemcpyHostToDevice));
  CK(hipMemcpy(lp, hpair.data(), hpair.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ns, hns.data(), hns.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(llw, 0, hns.size() * 8)); CK(hipMemset(logw, 0, (size_t)T * 8)); CK(hipMemset(wc, 0, N * 8)); CK(hipMemset(sc, 0, 64));
  A.groups = gr; A.fwd = fwd; A.bwd = bwd; A.lane_pair = lp; A.lane_nstates = ns; A.lane_logw = llw; A.logw = logw;
  A.post = post; A.wcache = wc; A.scalars = sc; A.pair_logprob = plp; A.first_group = 0; A.trace = nullptr;
  LatticeSet::LaneClass lc{0, NB, rows + 1};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int pre = 0; pre < 2; ++pre) {
    A.pre_weights = pre;
    for (int w = 0; w < 2; ++w) CK(launch_lane_sweep(A, lc, 0));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) CK(launch_lane_sweep(A, lc, 0));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("product sweep_lane_kernel, pre_weights=%d: %8.3f ms  %6.1f G records/s\n", pre, ms / 5, N / (ms / 5 * 1e-3) / 1e9);
  }
  double h[3]; CK(hipMemcpy(h, sc, 24, hipMemcpyDeviceToHost));
  printf("scalars %g %g %g\n", h[0], h[1], h[2]);
  return 0;
}
