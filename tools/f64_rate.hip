// f64 issue rates on gfx950: v_mfma_f64_16x16x4_f64 and v_fma_f64, independent chains, all SIMDs busy.
// build + run (through gpurun): hipcc --offload-arch=gfx950 -O3 tools/f64_rate.hip -o /tmp/f64_rate && /tmp/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void mfma_k(double* out, int iters, double a, double b) {
  d4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = d4{0, 0, 0, 0};
  double x = a + threadIdx.x * 1e-9, y = b;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[c], 0, 0, 0);
  double s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void fma_k(double* out, int iters, double a, double b) {
  double acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = c;
  double x = a + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = fma(acc[c], x, b);
  double s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double* out;
  hipMalloc(&out, 8 * 256 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, grid = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  float ms;
  auto run = [&](auto kern, const char* name, double flops_per_thread_iter) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 100, 1.0000001, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    double fl = flops_per_thread_iter * iters * (double)grid * 256;
    printf("%-28s %8.3f ms  %7.2f TFLOP/s\n", name, ms, fl / (ms * 1e-3) / 1e12);
  };
  // one MFMA 16x16x4 = 2 * 16*16*4 flops per wave = 32 per lane
  run(mfma_k<1>, "mfma_f64 1 chain", 32.0 * 1);
  run(mfma_k<4>, "mfma_f64 4 chains", 32.0 * 4);
  run(mfma_k<8>, "mfma_f64 8 chains", 32.0 * 8);
  run(fma_k<4>, "fma_f64 4 chains", 2.0 * 4);
  run(fma_k<16>, "fma_f64 16 chains", 2.0 * 16);
  return 0;
}
