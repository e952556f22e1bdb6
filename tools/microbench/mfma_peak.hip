// Microbenchmark: what fp32 MFMA rate does the chip SUSTAIN, and at which shader clock?  256 x NB workgroups x 4 waves (1, 2 or 4 waves per
// SIMD), every wave a register-only stream of v_mfma_f32_16x16x4_f32 on 8 independent accumulators (no memory, no VALU): the roof a
// GEMM loop can reach.  Reports TFLOP/s from hipEvents and the shader clock from s_memtime / s_memrealtime inside the kernel, for a short
// (~50 us, like one dense layer) and a long (~2 ms) run.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/microbench/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 a = {1.f + lane, 0.5f, 0.25f, 0.125f}, b = {1.f, 2.f, 3.f, 4.f};
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[q], 0, 0, 0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].w;
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, sizeof(float) * 4096 * 256));
  CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nb : {1, 2, 4}) {
    for (int iters : {100, 4000}) {
      const int grid = 256 * nb;
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, iters, out, clk);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const int reps = iters > 1000 ? 3 : 20;
      for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, iters, out, clk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(2 * grid);
      CK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
      double mhz = 0; for (int i = 0; i < grid; ++i) mhz += (double)h[2 * i] / ((double)h[2 * i + 1] / 100.0);
      const double flop = (double)grid * 4 * iters * 32 * 2048.0 * reps;
      printf("%d wave(s) per SIMD, %5d iterations: %8.1f us per launch  %6.1f TFLOP/s  shader clock %.0f MHz (in-kernel)  -> MFMA issue %.3f of the clock\n", nb, iters,
             ms * 1e3 / reps, flop / (ms * 1e-3) / 1e12, mhz / grid, (flop / (ms * 1e-3)) / (1024.0 * 64.0 * (mhz / grid) * 1e6 ));
    }
  }
  return 0;
}
