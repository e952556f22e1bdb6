// Microbenchmark: do MFMA work of one wave and VALU / transcendental work of ANOTHER wave on the same SIMD overlap?
// 256 workgroups x 8 waves (2 per SIMD).  Waves 0-3 = "matrix" waves (v_mfma_f32_16x16x4_f32 from registers), waves 4-7 =
// "vector" waves (BN + swish on 4 floats per iteration: v_pk_fma, v_exp, v_rcp, ...).  Three runs: matrix waves only,
// vector waves only, both.  overlap = (t_m + t_v - t_both) / min(t_m, t_v): 1 = the shorter one is free, 0 = they serialize.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap mfma_valu_overlap.hip && /tmp/overlap [mfma_iters] [valu_iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void k(int mode, int mfma_iters, int valu_iters, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool matrix = wave < 4;
  if (matrix && (mode & 1)) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 a = {1.f + lane, 0.5f, 0.25f, 0.125f}, b = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[q], 0, 0, 0);
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
  } else if (!matrix && (mode & 2)) {
    f32x4 y = {0.001f * lane, 0.002f * lane, -0.001f * lane, 0.5f};
    const f32x4 sc = {1.01f, 0.99f, 1.02f, 0.98f}, sh = {0.1f, -0.1f, 0.2f, -0.2f};
    for (int it = 0; it < valu_iters; ++it) {
      f32x4 z = y * sc + sh;
      z.x = z.x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * z.x));
      z.y = z.y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * z.y));
      z.z = z.z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * z.z));
      z.w = z.w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * z.w));
      y = z + y * 0.5f;
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = y.x + y.y + y.z + y.w;
  }
}

int main(int argc, char** argv) {
  const int mi = argc > 1 ? atoi(argv[1]) : 2000, vi = argc > 2 ? atoi(argv[2]) : 4000;
  float* out; CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float t[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    float best = 1e9f;
    for (int it = 0; it < 12; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, mi, vi, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2 && ms < best) best = ms;
    }
    t[mode] = best * 1000.f;
  }
  const float mn = t[1] < t[2] ? t[1] : t[2];
  printf("mfma_iters %d (x16 MFMA) valu_iters %d (x4 swish): matrix only %.1f us (%.1f TF/s on half the waves), vector only %.1f us, both %.1f us -> overlap %.2f\n",
         mi, vi, t[1], 256.0 * 4 * mi * 16 * 2048.0 / (t[1] * 1e6), t[2], t[3], (t[1] + t[2] - t[3]) / mn);
  return 0;
}
