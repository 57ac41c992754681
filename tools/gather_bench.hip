// Microbenchmark (tools/, not part of the product): what does a random 8-byte access stream cost on MI355X?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/gather_bench.hip -o gather_bench && ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void gather_k(const double* __restrict__ tab, const float* __restrict__ tabf, const uint32_t* __restrict__ idx,
                         double* __restrict__ out, uint64_t n, uint32_t mask) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t i = idx[k] & mask;
    double v;
    if (MODE == 0) v = tab[i];
    else if (MODE == 1) v = __builtin_nontemporal_load(tab + i);
    else if (MODE == 2) v = tabf[i];
    else v = __builtin_nontemporal_load(tabf + i);
    out[k] = v;
  }
}
template <int MODE>
__global__ void scatter_k(double* __restrict__ tab, const uint32_t* __restrict__ idx, const double* __restrict__ val,
                          uint64_t n, uint32_t mask) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t i = idx[k] & mask;
    if (MODE == 0) unsafeAtomicAdd(tab + i, val[k]);
    else if (MODE == 1) tab[i] = val[k];
    else __builtin_nontemporal_store(val[k], tab + i);
  }
}
int main() {
  const uint64_t N = 22500000, T = 10000000;
  std::vector<uint32_t> h(N);
  std::mt19937 rng(1);
  for (auto& x : h) x = rng() % T;
  uint32_t* idx; double *tab, *out; float* tabf;
  CK(hipMalloc(&idx, N * 4)); CK(hipMalloc(&tab, T * 8)); CK(hipMalloc(&tabf, T * 4)); CK(hipMalloc(&out, N * 8));
  CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(tab, 0, T * 8)); CK(hipMemset(tabf, 0, T * 4)); CK(hipMemset(out, 0, N * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.2f G/s\n", name, ms / 5, N / (ms / 5 * 1e-3) / 1e9);
  };
  dim3 g(256 * 16), b(256);
  uint32_t full = 0xffffffffu;
  time("gather f64 plain, 80 MB table", [&] { hipLaunchKernelGGL(gather_k<0>, g, b, 0, 0, tab, tabf, idx, out, N, full); });
  time("gather f64 nontemporal", [&] { hipLaunchKernelGGL(gather_k<1>, g, b, 0, 0, tab, tabf, idx, out, N, full); });
  time("gather f32 plain, 40 MB table", [&] { hipLaunchKernelGGL(gather_k<2>, g, b, 0, 0, tab, tabf, idx, out, N, full); });
  time("gather f32 nontemporal", [&] { hipLaunchKernelGGL(gather_k<3>, g, b, 0, 0, tab, tabf, idx, out, N, full); });
  time("gather f64, 2 MB window (L2-resident)", [&] { hipLaunchKernelGGL(gather_k<0>, g, b, 0, 0, tab, tabf, idx, out, N, 0x3ffffu); });
  time("gather f64, 16 MB window", [&] { hipLaunchKernelGGL(gather_k<0>, g, b, 0, 0, tab, tabf, idx, out, N, 0x1fffffu); });
  time("atomic add f64, 80 MB table", [&] { hipLaunchKernelGGL(scatter_k<0>, g, b, 0, 0, tab, idx, out, N, full); });
  time("atomic add f64, 2 MB window", [&] { hipLaunchKernelGGL(scatter_k<0>, g, b, 0, 0, tab, idx, out, N, 0x3ffffu); });
  time("atomic add f64, 256 KB window", [&] { hipLaunchKernelGGL(scatter_k<0>, g, b, 0, 0, tab, idx, out, N, 0x7fffu); });
  time("plain store f64 random, 80 MB", [&] { hipLaunchKernelGGL(scatter_k<1>, g, b, 0, 0, tab, idx, out, N, full); });
  time("nontemporal store f64 random, 80 MB", [&] { hipLaunchKernelGGL(scatter_k<2>, g, b, 0, 0, tab, idx, out, N, full); });
  time("plain store f64 random, 2 MB window", [&] { hipLaunchKernelGGL(scatter_k<1>, g, b, 0, 0, tab, idx, out, N, 0x3ffffu); });
  return 0;
}
