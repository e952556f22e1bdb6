// Microbenchmark: what sustains the fp32 MFMA pipe in the shape of mbconv_block_kernel's streaming phases?
// 256 workgroups x 8 waves; each wave issues chunks of NT*MT*4 v_mfma_f32_16x16x4_f32 with
//   mode 0: operands in registers only            mode 1: + B operand from LDS (ds_read_b128 per m-tile per chunk)
//   mode 2: + A operand streamed from global (one float4 per n-tile per chunk, D-deep ring), every workgroup the same stream
//   mode 3: as 2 but each workgroup streams its own copy of the weights (no sharing in L2)
// Prints achieved TFLOP/s.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_stream mfma_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int NT, int MODE, int D, int EPI = 0, int KC = 7>
__global__ __launch_bounds__(512) void k(const float* __restrict__ W, size_t wstride_wg, int chunks, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * 1024; i += 512) lds[i] = 0.001f * (i & 15);
  __syncthreads();
  const float* w = W + (size_t)blockIdx.x * wstride_wg + (size_t)wave * 65536 + lane * 4;   // wave-private stream region (256 KB apart)
  f32x4 acc[NT][MT];
  for (int q = 0; q < NT; ++q) for (int m = 0; m < MT; ++m) acc[q][m] = (f32x4){0, 0, 0, 0};
  f32x4 ring[D][NT];
  f32x4 wreg[NT], xreg[MT];
  for (int q = 0; q < NT; ++q) wreg[q] = (f32x4){1.f, 0.5f, 0.25f, 0.125f};
  for (int m = 0; m < MT; ++m) xreg[m] = (f32x4){1.f, 2.f, 3.f, 4.f};
  auto load = [&](int j, f32x4 (&r)[NT]) {
    for (int q = 0; q < NT; ++q) r[q] = *reinterpret_cast<const f32x4*>(w + ((size_t)(j % 240) * NT + q) * 256);
  };
  if (MODE >= 2) for (int d = 0; d < D; ++d) load(d, ring[d]);
  f32x4 pend[NT * MT]; int pend_n = 0, cj = 0;
  for (int i = 0; i < NT * MT; ++i) pend[i] = (f32x4){0, 0, 0, 0};
  const f32x4 sc = {1.01f, 0.99f, 1.02f, 0.98f}, sh = {0.1f, -0.1f, 0.2f, -0.2f};
  auto swish = [](float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * x)); };
  for (int j0 = 0; j0 < chunks; j0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int j = j0 + d;
      f32x4 x[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        x[m] = (MODE >= 1) ? *reinterpret_cast<const f32x4*>(lds + ((size_t)((j & 15) * MT + m) * 64 + lane) * 4) : xreg[m];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x4f32((MODE >= 2) ? ring[d][q][s] : wreg[q][s], x[m][s], acc[q][m], 0, 0, 0);
      if (MODE >= 2) { load(j + D, ring[d]); __builtin_amdgcn_sched_barrier(0); }
      if (EPI == 2 && pend_n > 0) {             // one parked accumulator per chunk, in the shadow of the MFMAs just issued
        f32x4 y = pend[0] * sc + sh;
        y.x = swish(y.x); y.y = swish(y.y); y.z = swish(y.z); y.w = swish(y.w);
        *reinterpret_cast<f32x4*>(lds + 8192 + ((size_t)(pend_n & 3) * 64 + lane) * 4) = y;
#pragma unroll
        for (int i = 0; i + 1 < NT * MT; ++i) pend[i] = pend[i + 1];
        --pend_n;
      }
      if (EPI != 0 && ++cj == KC) {
        cj = 0;
        if (EPI == 1) {
#pragma unroll
          for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              f32x4 y = acc[q][m] * sc + sh;
              y.x = swish(y.x); y.y = swish(y.y); y.z = swish(y.z); y.w = swish(y.w);
              *reinterpret_cast<f32x4*>(lds + 8192 + ((size_t)(q * MT + m) * 64 + lane) * 4) = y;
            }
        } else {
#pragma unroll
          for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) pend[q * MT + m] = acc[q][m];
          pend_n = NT * MT;
        }
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[q][m] = (f32x4){0, 0, 0, 0};
      }
    }
  }
  f32x4 t = {0, 0, 0, 0};
  for (int q = 0; q < NT; ++q) for (int m = 0; m < MT; ++m) t += acc[q][m];
  if (t.x == 12345.678f) out[blockIdx.x * 512 + tid] = t.x + t.y + t.z + t.w;
}

template <int MT, int NT, int MODE, int D, int EPI = 0, int KC = 7>
void run(const char* name, const float* W, float* out, int nwg) {
  const int chunks = 960;
  const size_t stride = (MODE == 3) ? (size_t)8 * 65536 : 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MT, NT, MODE, D, EPI, KC>), dim3(nwg), dim3(512), 64 * 1024, 0, W, stride, chunks, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)nwg * 8 * chunks * NT * MT * 4 * 2048.0;
  printf("%-44s MT=%d NT=%d D=%d: %7.3f ms  %6.1f TFLOP/s\n", name, MT, NT, D, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
  const int nwg = 256;
  float *W, *out;
  const size_t wfloats = (size_t)nwg * 8 * 65536 + 240 * 4 * 256 + 1024;
  hipMalloc(&W, wfloats * 4); hipMalloc(&out, (size_t)nwg * 512 * 4);
  hipMemset(W, 0, wfloats * 4);
  run<3, 1, 0, 4>("registers only", W, out, nwg);
  run<3, 1, 1, 4>("+ LDS B operand", W, out, nwg);
  run<3, 1, 2, 4>("+ global A stream (shared by all WGs)", W, out, nwg);
  run<3, 1, 3, 4>("+ global A stream (private per WG)", W, out, nwg);
  run<1, 2, 0, 4>("registers only", W, out, nwg);
  run<1, 2, 1, 4>("+ LDS B operand", W, out, nwg);
  run<1, 2, 2, 4>("+ global A stream (shared by all WGs)", W, out, nwg);
  run<1, 2, 2, 8>("+ global A stream (shared), deeper ring", W, out, nwg);
  run<1, 2, 3, 4>("+ global A stream (private per WG)", W, out, nwg);
  run<2, 4, 2, 3>("dense-like tile, global A stream", W, out, nwg);
  run<3, 1, 2, 4, 1, 7>("4x3 expand: + BN/swish epilogue per 7 chunks", W, out, nwg);
  run<3, 1, 2, 4, 2, 7>("4x3 expand: epilogue deferred into next run", W, out, nwg);
  run<1, 2, 2, 4, 1, 12>("2x2 expand: + epilogue per 12 chunks", W, out, nwg);
  run<1, 2, 2, 4, 2, 12>("2x2 expand: epilogue deferred", W, out, nwg);
  return 0;
}
