// Microbenchmark (round 6): what ONE instruction of each class costs a gfx950 SIMD, alone and beside fp32 MFMAs.
//   256 workgroups x 8 waves (2 per SIMD).  Waves 0-3 = "matrix" waves, waves 4-7 = "vector" waves.
//   For each vector class X in {v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_exp_f32, v_rcp_f32, v_mad_i32_i16, v_pk_add_i16, ds_read_b128}
//   and each matrix instruction M in {16x16x4 f32, 32x32x2 f32, 16x16x16 bf16}:
//     t_x   vector waves alone (cycles per instruction = t_x * clock / count)
//     t_m   matrix waves alone
//     t_xm  both at once, on the same SIMDs            -> overlap = (t_x + t_m - t_xm) / min(t_x, t_m)
//     t_s   ONE wave per SIMD issuing {1 MFMA, k x X} per iteration (same-wave shadow): extra cycles per X beside the MFMA
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_costs issue_costs.hip && /tmp/issue_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { X_FMA, X_PKFMA, X_PKMUL, X_EXP, X_RCP, X_MADI16, X_PKADDI16, X_LDS, X_N };
static const char* xname[X_N] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32", "v_mad_i32_i16", "v_pk_add_i16", "ds_read_b128"};
enum { M_16x4, M_32x2, M_BF16, M_N };
static const char* mname[M_N] = {"mfma_f32_16x16x4_f32", "mfma_f32_32x32x2_f32", "mfma_f32_16x16x16_bf16"};

// eight independent instructions of class X (no dependencies between them or across calls: sources are loop-invariant)
template <int X>
__device__ __forceinline__ void eight(float (&r)[8], const float (&a)[8], f32x2 (&r2)[8], const f32x2 (&a2)[8], const float* lds, f32x4 (&l4)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // every instruction updates its own register in place (eight independent chains): the allocator cannot fold the results into one register
    // (a write-after-write on a transcendental result makes hipcc insert s_nop wait states)
    if (X == X_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
    if (X == X_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r2[i]) : "v"(a2[(i + 1) & 7]), "v"(a2[(i + 2) & 7]));
    if (X == X_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r2[i]) : "v"(a2[(i + 1) & 7]));
    if (X == X_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
    if (X == X_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
    if (X == X_MADI16) asm volatile("v_mad_i32_i16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
    if (X == X_PKADDI16) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r[i]) : "v"(a[(i + 1) & 7]));
    if (X == X_LDS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l4[i]) : "v"((unsigned)(size_t)lds), "n"(i * 1024));
  }
  if (X == X_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // eight reads in flight, one wait
}

template <int M>
struct Mat {
  f32x4 acc4[4]; f32x16 acc16[2];
  __device__ __forceinline__ void init() {
    for (int i = 0; i < 4; ++i) acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
  }
  // four MFMAs on independent accumulators (two for the 32x32 form: 16 passes each, so two already fill the pipe)
  __device__ __forceinline__ void four(float a, float b, bf16x8 ha, bf16x8 hb) {
    if (M == M_16x4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc4[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[q], 0, 0, 0);
    } else if (M == M_32x2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc16[q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[q & 1], 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc4[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc4[q], 0, 0, 0);
    }
  }
  __device__ __forceinline__ float fold() { return acc4[0].x + acc4[1].y + acc4[2].z + acc4[3].w + acc16[0][3] + acc16[1][7]; }
};

// mode bit 0: matrix waves run, bit 1: vector waves run, bit 2: same-wave interleave (waves 0-3 only: 4 MFMAs + 8*k X per iteration)
template <int X, int M>
__global__ __launch_bounds__(512) void k(int mode, int iters, int kx, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = 0.001f * i;
  __syncthreads();
  float a[8], r[8]; f32x2 a2[8], r2[8]; f32x4 l4[8];
  for (int i = 0; i < 8; ++i) { a[i] = 0.5f + 0.01f * (lane + i); r[i] = 0.f; a2[i] = (f32x2){a[i], a[i] + 0.25f}; r2[i] = (f32x2){0.f, 0.f}; l4[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(0.01f * (lane + i)); hb[i] = (__bf16)(0.02f * (lane - i)); }
  const float* lp = lds + lane * 4;
  Mat<M> mt; mt.init();
  const bool matrix = wave < 4;
  if (mode & 4) {
    if (matrix) {
      for (int it = 0; it < iters; ++it) {
        mt.four(a[0], a[1], ha, hb);
        for (int q = 0; q < kx; ++q) eight<X>(r, a, r2, a2, lp, l4);
      }
    }
  } else if (matrix && (mode & 1)) {
    for (int it = 0; it < iters; ++it) { mt.four(a[0], a[1], ha, hb); mt.four(a[2], a[3], ha, hb); }
  } else if (!matrix && (mode & 2)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) eight<X>(r, a, r2, a2, lp, l4);
    }
  }
  float s = mt.fold();
  for (int i = 0; i < 8; ++i) s += r[i] + r2[i].x + r2[i].y + l4[i].x + l4[i].w;
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

static float* g_out;
template <int X, int M>
static float run(int mode, int iters, int kx) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 7; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<X, M>), dim3(256), dim3(512), 0, 0, mode, iters, kx, g_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2 && ms < best) best = ms;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return best * 1000.f;
}

template <int X, int M>
static void report(int iters) {
  const float tm = run<X, M>(1, iters, 0), tx = run<X, M>(2, iters, 0), txm = run<X, M>(3, iters, 0);
  // matrix alone: 8 MFMAs per iteration; vector alone: 32 X per iteration.  Cycle figures assume the matrix-alone run issues the 16x16x4 form
  // back to back at 32 cycles each (the clock it implies is printed).
  const float mn = tm < tx ? tm : tx;
  printf("%-24s | %-14s | matrix %8.1f us  vector %8.1f us (%6.2f ns per X per wave)  both %8.1f us  overlap %5.2f", mname[M], xname[X], tm, tx,
         tx * 1000.f / (32.f * iters), txm, (tm + tx - txm) / mn);
  // same-wave shadow: 4 MFMAs + 8 k X per iteration, k = 0, 1, 2, 4
  float ts[4]; const int ks[4] = {0, 1, 2, 4};
  for (int i = 0; i < 4; ++i) ts[i] = run<X, M>(4, iters, ks[i]);
  printf("  | same wave, 4 MFMA + {0,8,16,32} X per iteration: %7.1f %7.1f %7.1f %7.1f us\n", ts[0], ts[1], ts[2], ts[3]);
}

template <int M>
static void all_x(int iters) {
  report<X_FMA, M>(iters); report<X_PKFMA, M>(iters); report<X_PKMUL, M>(iters); report<X_EXP, M>(iters); report<X_RCP, M>(iters);
  report<X_MADI16, M>(iters); report<X_PKADDI16, M>(iters); report<X_LDS, M>(iters);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  CK(hipMalloc(&g_out, 256 * 512 * sizeof(float)));
  printf("iters %d: matrix waves issue 8 MFMAs per iteration, vector waves 32 X per iteration; one wave of each kind per SIMD\n", iters);
  all_x<M_16x4>(iters);
  all_x<M_32x2>(iters);
  all_x<M_BF16>(iters);
  return 0;
}
