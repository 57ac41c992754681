// Microbenchmark (tools/, not part of the product): how fast can one-wave workgroups stream private contiguous
// regions (the lane-sweep access pattern: wave w reads rows base_w + k*64*sizeof(T), one row per instruction)?
//   hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o stream_bench && ./stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <typename T> __device__ inline double tsum(T v);
template <> __device__ inline double tsum(uint32_t v) { return (double)v; }
template <> __device__ inline double tsum(uint2 v) { return (double)(v.x + v.y); }
template <> __device__ inline double tsum(uint4 v) { return (double)(v.x + v.y + v.z + v.w); }

// each block (1 wave) reads ROWS rows of 64*sizeof(T) bytes from its own region, D rows in flight
template <typename T, int D>
__global__ __launch_bounds__(64) void stream_k(const T* __restrict__ src, double* __restrict__ out, uint32_t rows) {
  extern __shared__ double lds[];
  const T* p = src + (size_t)blockIdx.x * rows * 64 + threadIdx.x;
  T q[D];
#pragma unroll
  for (int j = 0; j < D; ++j) q[j] = p[(size_t)(j < (int)rows ? j : rows - 1) * 64];
  double acc = 0;
  uint32_t k0 = 0;
  for (; k0 + D <= rows; k0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      acc += tsum(q[j]);
      uint32_t k = k0 + D + j;
      q[j] = p[(size_t)(k < rows ? k : rows - 1) * 64];
    }
  }
  if (acc == 123.456) lds[threadIdx.x] = acc;
  if (threadIdx.x == 0) out[blockIdx.x] = acc + (acc == 123.456 ? lds[1] : 0);
}

// same bytes, but a "flat" grid-stride float4 copy-like read for reference
__global__ void flat_k(const uint4* __restrict__ src, double* __restrict__ out, size_t n) {
  double acc = 0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) acc += tsum(src[k]);
  if (acc == 123.456) out[0] = acc;
}

int main() {
  const size_t BYTES = 512ull << 20;
  void* src; double* out;
  CK(hipMalloc(&src, BYTES)); CK(hipMalloc(&out, 1 << 24));
  CK(hipMemset(src, 1, BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, size_t bytes, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %8.3f ms  %7.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    CK(hipGetLastError());
  };
  time("flat grid-stride uint4 read, 4096x256", BYTES, [&] { hipLaunchKernelGGL(flat_k, dim3(4096), dim3(256), 0, 0, (const uint4*)src, out, BYTES / 16); });
  for (int ldskb : {4, 20, 60}) {
    for (uint32_t rows : {40u, 160u}) {
      char nm[128];
#define RUN(T, D)                                                                                                  \
  {                                                                                                                \
    size_t per = (size_t)rows * 64 * sizeof(T);                                                                    \
    unsigned nb = (unsigned)(BYTES / per);                                                                         \
    snprintf(nm, sizeof nm, "1-wave blocks, %2zu B/lane, %2d rows in flight, %3u rows, LDS %2d KB", sizeof(T), D, rows, ldskb); \
    time(nm, (size_t)nb * per, [&] { hipLaunchKernelGGL((stream_k<T, D>), dim3(nb), dim3(64), ldskb * 1024, 0, (const T*)src, out, rows); }); \
  }
      RUN(uint32_t, 8) RUN(uint2, 4) RUN(uint2, 8) RUN(uint2, 16) RUN(uint4, 4) RUN(uint4, 8) RUN(uint4, 16)
    }
  }
  return 0;
}
