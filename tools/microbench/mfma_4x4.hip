// Layout and rate check of v_mfma_f32_4x4x1_16b_f32 on gfx950 (round 6: the squeeze-excite FCs of 4 clips want a 4-wide N, not 16).
// Hypothesis (CDNA3 ISA, 16 blocks of 4x4x1): lane l = 4 * blk + r.  A operand: a(i = r, blk); B operand: b(j = r, blk);
// D VGPR v of lane l: D(i = v, j = r, blk) = sum_k a(i = v, blk) * b(j = r, blk).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_4x4 mfma_4x4.hip && /tmp/mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void one(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + l], b[64 + l], acc, 0, 0, 0);      // second k step: accumulates
  for (int v = 0; v < 4; ++v) d[v * 64 + l] = acc[v];
}
// K random steps, operands from memory as float4 (the shape of the real use): every (lane, VGPR) against the hypothesis
__global__ void many(const f32x4* a4, const f32x4* b4, int nq, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < nq; ++q) {
    const f32x4 a = a4[q * 64 + l], b = b4[q * 64 + l];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], b[e], acc, 0, 0, 0);
  }
  for (int v = 0; v < 4; ++v) d[v * 64 + l] = acc[v];
}
__global__ __launch_bounds__(256) void rate(int iters, float* out) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const float a = 1.0f + threadIdx.x, b = 0.5f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q & 3], 0, 0, 0);
  out[blockIdx.x * 256 + threadIdx.x] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
}
int main() {
  float ha[128], hb[128], hd[256];
  for (int i = 0; i < 128; ++i) { ha[i] = 1.0f + 0.37f * i; hb[i] = 2.0f - 0.11f * i; }
  float *a, *b, *d, *o;
  hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&d, sizeof hd); hipMalloc(&o, 256 * 256 * 4);
  hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      const int blk = l / 4;
      const float want = fmaf(ha[64 + 4 * blk + v], hb[64 + l], ha[4 * blk + v] * hb[l]);
      if (fabsf(hd[v * 64 + l] - want) > 1e-4f * fabsf(want)) { if (bad < 8) printf("lane %d vgpr %d: got %g want %g\n", l, v, hd[v * 64 + l], want); ++bad; }
    }
  printf("layout hypothesis: %s (%d mismatches of 256)\n", bad ? "WRONG" : "confirmed", bad);
  {
    const int nq = 11;
    std::vector<float> A(nq * 256), Bv(nq * 256);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : A) v = rnd(); for (auto& v : Bv) v = rnd();
    float *da, *db; hipMalloc(&da, A.size() * 4); hipMalloc(&db, Bv.size() * 4);
    hipMemcpy(da, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, Bv.data(), Bv.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(many, dim3(1), dim3(64), 0, 0, (const f32x4*)da, (const f32x4*)db, nq, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
      const int blk = l / 4, r = l % 4;
      double h1 = 0, h2 = 0;
      for (int q = 0; q < nq; ++q) for (int e = 0; e < 4; ++e) {
        h1 += (double)A[(q * 64 + 4 * blk + v) * 4 + e] * Bv[(q * 64 + l) * 4 + e];          // d(i = v, j = r) = a(lane 4 blk + v) b(lane l)
        h2 += (double)A[(q * 64 + l) * 4 + e] * Bv[(q * 64 + 4 * blk + v) * 4 + e];          // the transpose
      }
      if (std::fabs(hd[v * 64 + l] - h1) > 1e-4) ++bad1;
      if (std::fabs(hd[v * 64 + l] - h2) > 1e-4) ++bad2;
      (void)r;
    }
    printf("44 random steps: hypothesis (A by VGPR, B by lane) %d mismatches, transpose %d mismatches of 256\n", bad1, bad2);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0); hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, iters, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
  }
  printf("4x4x1 back to back, one wave per SIMD: %.2f ns per MFMA (16x16x4 measures ~14.1 ns = 32 cycles; 4x4x1 is 256 MACs against 1024)\n", best * 1e6 / (8.0 * iters));
  return 0;
}
