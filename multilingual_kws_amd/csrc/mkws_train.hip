// mkws_train.hip -- training-mode operators for `backprop_into_embedding=True`
// (multilingual_kws/embedding/transfer_learning.py:94-112: the reference un-freezes the whole nested EfficientNet, so
// Keras runs it with training=True: BatchNormalization on batch statistics with moving-average updates, per-block
// drop-connect, gradients into every kernel / bias / gamma / beta, Adam over all of them).
//
// The reference's training graph lives in Keras (Python); here the graph (forward tape + backward sweep) is host
// code in multilingual_kws_amd/embedding_trainer.py and every numerical operator is one of the C-ABI entry points
// below (include/mkws.h, "training operators").  All tensors are NHWC float32 viewed as row-major [M, C]; parameter
// tensors keep their Keras layouts (conv kernels HWIO = [K, N] for 1x1, depthwise [kh, kw, C], dense [in, out]), so
// the flat parameter / gradient buffers are the weight blob itself.
// Reductions that cross workgroups are FIXED-ORDER (round 3): every workgroup writes its partial sums to a caller-provided scratch
// arena (mkws_op_set_scratch) and a small second launch folds them in index order -- no atomics, no memsets, so a training step
// with the same inputs, masks and parameters is bit-reproducible (round 2 used fp32 atomics).
#include "mkws_common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

#pragma clang fp contract(fast)

namespace mkws {

using f32x4 = __attribute__((ext_vector_type(4))) float;

enum TrainAct { TA_NONE = 0, TA_SWISH = 1, TA_RELU = 2, TA_SELU = 3, TA_SIGMOID = 4 };
constexpr float kSeluScale = 1.0507009873554805f, kSeluAlpha = 1.6732632423543772f;

// 1 / (1 + 2^(-x log2 e)) on the hardware exp2 / reciprocal (1 ulp each), the formulation of the inference kernels (mkws_embed.hip sigmoidf_).
// Round 4: with expf() and an IEEE division the BatchNorm launches of a 512-clip step were bound by this function, not by memory.
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float act_fwd(float y, int act) {
  switch (act) {
    case TA_SWISH: return y * sigm(y);
    case TA_RELU: return y > 0.0f ? y : 0.0f;
    case TA_SELU: return kSeluScale * (y > 0.0f ? y : kSeluAlpha * expm1f(y));
    case TA_SIGMOID: return sigm(y);
    default: return y;
  }
}
// d act(y) / dy
__device__ __forceinline__ float act_grad(float y, int act) {
  switch (act) {
    case TA_SWISH: { const float s = sigm(y); return s * (1.0f + y * (1.0f - s)); }
    case TA_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case TA_SELU: return kSeluScale * (y > 0.0f ? 1.0f : kSeluAlpha * expf(y));
    case TA_SIGMOID: { const float s = sigm(y); return s * (1.0f - s); }
    default: return 1.0f;
  }
}

// ------------------------------------------------------------------------------------------------
// Generic fp32 GEMM on the MFMA (v_mfma_f32_16x16x4_f32, exact fp32): C[M,N] (+)= op(A)[M,K] . op(B)[K,N].
// 64x64 block tile, 4 waves (2x2), each wave 2x2 MFMA tiles; K step 16 through LDS.  Operands are read with guarded
// scalar loads, so any M, N, K, leading dimension and both transposes work (1x1-conv forward: NN; input gradient
// dZ . W^T: NT; weight gradient X^T . dZ: TN with the long reduction split over blockIdx.z and fp32 atomics).
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void train_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
                                                         int K, int lda, int ldb, int ldc, int accumulate, int ksplit, float* __restrict__ part,
                                                         const float* __restrict__ bias, int act, float* __restrict__ Act, float* __restrict__ stats) {
  // Two LDS stages; the next K step's operands travel global -> registers while the current step's MFMAs run (round 2 loaded,
  // synchronised and multiplied one K step at a time: a full memory latency per 16 columns of K, 44-61 us per call at batch 64).
  __shared__ float As[2][64][17];
  __shared__ float Bs[2][16][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kper = ((K + ksplit - 1) / ksplit + 15) / 16 * 16;
  const int kbeg = blockIdx.z * kper, kend = (kbeg + kper < K) ? kbeg + kper : K;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                      // A tile: 64 x 16
      const int e = tid + 256 * i;
      const int r = TA ? (e & 63) : (e >> 4), kk = TA ? (e >> 6) : (e & 15);
      const int m = m0 + r, k = k0 + kk;
      ra[i] = (m < M && k < kend) ? (TA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k]) : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                      // B tile: 16 x 64
      const int e = tid + 256 * i;
      const int kk = TB ? (e & 15) : (e >> 6), cn = TB ? (e >> 4) : (e & 63);
      const int n = n0 + cn, k = k0 + kk;
      rb[i] = (n < N && k < kend) ? (TB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n]) : 0.0f;
    }
  };
  auto stash = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      As[st][TA ? (e & 63) : (e >> 4)][TA ? (e >> 6) : (e & 15)] = ra[i];
      Bs[st][TB ? (e & 15) : (e >> 6)][TB ? (e >> 4) : (e & 63)] = rb[i];
    }
  };
  if (kbeg < kend) {
    fetch(kbeg);
    stash(0);
  }
  __syncthreads();
  int st = 0;
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    const bool more = k0 + 16 < kend;
    if (more) fetch(k0 + 16);
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[st][wm * 32 + i * 16 + (lane & 15)][ks + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[st][ks + (lane >> 4)][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash(st ^ 1);
    __syncthreads();
    st ^= 1;
  }
  if (stats) {
    // BatchNorm chunk statistics of this 64-row tile in the epilogue (unsplit GEMMs in front of a training-mode BN: the separate statistics
    // launch and its pass over Z go away): per column the mean of the tile's valid rows and the sum of squared deviations from it, two
    // passes over the accumulators.  Order: rows of a lane, the four lane groups by butterfly, the two row halves (waves) in order.
    __shared__ float s_cs[2][64], s_cq[2][64];
    const int nrows = (M - m0 < 64) ? M - m0 : 64;
    float mu[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float t = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (wm * 32 + i * 16 + 4 * (lane >> 4) + r < nrows) t += acc[i][j][r];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      if ((lane >> 4) == 0) s_cs[wm][wn * 32 + j * 16 + (lane & 15)] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn * 32 + j * 16 + (lane & 15);
      mu[j] = (s_cs[0][col] + s_cs[1][col]) / (float)nrows;
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (wm * 32 + i * 16 + 4 * (lane >> 4) + r < nrows) { const float d = acc[i][j][r] - mu[j]; q += d * d; }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if ((lane >> 4) == 0) s_cq[wm][col] = q;
    }
    __syncthreads();
    if (wm == 0 && (lane >> 4) == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 32 + j * 16 + (lane & 15), n = n0 + col;
        if (n < N) {
          stats[((size_t)blockIdx.y * 2 + 0) * N + n] = mu[j];
          stats[((size_t)blockIdx.y * 2 + 1) * N + n] = s_cq[0][col] + s_cq[1][col];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + 4 * (lane >> 4) + r, n = n0 + wn * 32 + j * 16 + (lane & 15);
        if (m < M && n < N) {
          if (ksplit > 1) {
            part[((size_t)blockIdx.z * M + m) * N + n] = acc[i][j][r];       // raw slice sums; gemm_reduce_kernel folds them in order
          } else {
            float* dst = C + (size_t)m * ldc + n;
            const float v = accumulate ? *dst + acc[i][j][r] : acc[i][j][r];
            *dst = v;
            if (Act) Act[(size_t)m * ldc + n] = act_fwd(v + bias[n], act);      // fused dense / SE epilogue: C keeps the pre-activation for the backward pass
          }
        }
      }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the NN / NT forms without LDS.  The generic kernel above stages both operands through LDS with one scalar load per element
// and a barrier per 16 columns of K; on the step's shapes it reaches 12-70 TFLOP/s where the inference GEMM (pw_gemm_kernel,
// mkws_embed.hip) reaches 121.  This kernel is that one's scheme on unpacked operands: every operand fragment goes global / L2 ->
// registers through buffer loads with a uniform descriptor, a 32-bit per-lane byte offset and the K-chunk offset in an SGPR (the K loop
// is MFMAs, loads and counted waits only), a D-deep register ring over the K chunks, no barrier anywhere.
//   A [M, K] row-major (X, or dZ): lane (g, c) of a row tile reads A[row c][16 j + 4 g .. + 3], one float4.
//   B, TB = false: W [K, N] row-major (forward): lane (g, c) reads W[16 j + 4 g + s][col c], s = 0 .. 3 -- four dword loads, each 64
//      contiguous bytes per lane group; W is small and sits in L1 / L2.
//   B, TB = true: B [N, K] row-major (dX = dZ . W^T with W [Cin, Cout]): lane (g, c) reads B[col c][16 j + 4 g .. + 3], one float4.
//   MFMA step s of a chunk covers k = 16 j + 4 g + s on both operands.  Accumulation order of an output element: K chunks ascending,
//   steps ascending inside a chunk, the four lane groups inside a step as the matrix core adds them -- a function of the shapes only.
// Rows / columns outside the matrices: the descriptors carry the exact byte sizes, out-of-range loads return 0, nothing is stored there.
// The K tail (K % 16 in {4, 8, 12}) is peeled: the A fragment of lane groups past K is zeroed (its address holds the next row's data).
// A workgroup = 4 waves = 64 * MT rows x 16 * NT columns; with MT = 1 its rows are one 64-row BatchNorm chunk and the epilogue can leave
// the chunk statistics (mean, M2) as the generic kernel does.  Needs K, N, lda, ldb, ldc % 4 == 0 and 16-byte aligned A / C / part / Act
// (activations and scratch: host-checked).  B is a WEIGHT: a view into the trainer's flat parameter buffer at its blob offset, which for most
// tensors is 8 bytes off a 16-byte boundary -- the TB = true fragment of B is therefore read as BV-float pieces (BV = 4: one float4; 2: two
// float2; 1: four dwords), the widest the base allows; TB = false reads dwords anyway; the bias is read element by element.
template <bool TB, int MT, int NT, int BV>
__global__ __launch_bounds__(256) void train_gemm2_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int lda,
                                                          int ldb, int ldc, int accumulate, int ksplit, float* __restrict__ part, const float* __restrict__ bias,
                                                          int act, float* __restrict__ Act, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int m0 = blockIdx.y * (64 * MT) + wave * (16 * MT);
  const int n0 = blockIdx.x * (16 * NT);
  const int KC = (K + 15) >> 4;
  int jbeg = 0, jend = KC;
  if (ksplit > 1) {
    const int per = (KC + ksplit - 1) / ksplit;
    jbeg = blockIdx.z * per;
    jend = (jbeg + per < KC) ? jbeg + per : KC;
    if (jbeg > jend) jbeg = jend;
  }
  const unsigned bytesA = (unsigned)(((size_t)(M - 1) * lda + K) * sizeof(float));
  const unsigned bytesB = (unsigned)(TB ? ((size_t)(N - 1) * ldb + K) * sizeof(float) : ((size_t)(K - 1) * ldb + N) * sizeof(float));
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, bytesB, 0x00020000);
  unsigned aoff[MT], boff[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + mt * 16 + c;
    aoff[mt] = (m < M) ? (unsigned)(((size_t)m * lda + 4 * g) * sizeof(float)) : 0xFFFFFFF0u;       // (past the last row: out of range, reads 0)
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + nt * 16 + c;
    if (TB) boff[nt] = (n < N) ? (unsigned)(((size_t)n * ldb + 4 * g) * sizeof(float)) : 0xFFFFFFF0u;
    else boff[nt] = (n < N) ? (unsigned)(((size_t)(4 * g) * ldb + n) * sizeof(float)) : 0xFFFFFFF0u;
  }
  const unsigned brow = (unsigned)ldb * (unsigned)sizeof(float);                   // TB = false: bytes between two k rows of W
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int D = 3;
  f32x4 aq[D][MT], bq[D][NT];
  auto load = [&](int j, f32x4 (&av)[MT], f32x4 (&bv)[NT]) {
    const unsigned kx = 64u * (unsigned)j;                                          // 16 floats of K per chunk
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) av[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff[mt], kx, 0));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (TB) {
        if constexpr (BV == 4) {
          bv[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, boff[nt], kx, 0));
        } else if constexpr (BV == 2) {
          typedef float f32x2_ __attribute__((ext_vector_type(2)));
          const f32x2_ lo = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(rB, boff[nt], kx, 0));
          const f32x2_ hi = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(rB, boff[nt], kx + 8u, 0));
          bv[nt] = (f32x4){lo.x, lo.y, hi.x, hi.y};
        } else {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) bv[nt][s4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, boff[nt], kx + 4u * (unsigned)s4, 0));
        }
      } else {
        const unsigned k0 = 16u * (unsigned)j * brow;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bv[nt][s4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, boff[nt], k0 + (unsigned)s4 * brow, 0));
      }
    }
  };
  auto compute = [&](const f32x4 (&av)[MT], const f32x4 (&bv)[NT]) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[nt][s4], av[mt][s4], acc[mt][nt], 0, 0, 0);
  };
  const bool ktail = (K & 15) != 0;                                                  // uniform
  const int jpipe_end = (ktail && jend == KC && jend > jbeg) ? jend - 1 : jend;
  const int n = jpipe_end - jbeg;
  if (n >= D) {
#pragma unroll
    for (int d = 0; d < D; ++d) load(jbeg + d, aq[d], bq[d]);
    int j = jbeg;
    for (; j + 2 * D <= jpipe_end; j += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        compute(aq[d], bq[d]);
        load(j + D + d, aq[d], bq[d]);
        __builtin_amdgcn_sched_barrier(0);       // keep each reload behind its slot's MFMAs (see pw_gemm_kernel)
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      compute(aq[d], bq[d]);
      if (j + D + d < jpipe_end) load(j + D + d, aq[d], bq[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
    j += D;
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (j + d < jpipe_end) compute(aq[d], bq[d]);
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < n) load(jbeg + d, aq[d], bq[d]);
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < n) compute(aq[d], bq[d]);
  }
  if (jpipe_end != jend) {                       // peeled K-tail chunk: lane groups whose four columns lie past K contribute nothing
    load(jend - 1, aq[0], bq[0]);
    if (16 * (jend - 1) + 4 * g >= K) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) aq[0][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bq[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    compute(aq[0], bq[0]);
  }
  // lane (g, c) holds rows m0 + mt * 16 + c, columns n0 + nt * 16 + 4 g .. + 3
  if (MT == 1 && stats) {
    // BatchNorm chunk statistics of this workgroup's 64 rows (see train_gemm_kernel): per column the mean of the valid rows and the sum of
    // squared deviations from it, two passes over the accumulators.  Order: the 16 rows of a wave by butterfly, the four waves in order.
    __shared__ float s_cs[4][16 * NT], s_cq[4][16 * NT];
    const int r0 = blockIdx.y * 64;
    const int nrows = (M - r0 < 64) ? M - r0 : 64;
    const bool rowok = m0 + c < M;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 t = rowok ? acc[0][nt] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int off = 1; off < 16; off <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] += __shfl_xor(t[r], off);
      if (c == 0) *reinterpret_cast<f32x4*>(&s_cs[wave][nt * 16 + 4 * g]) = t;
    }
    __syncthreads();
    f32x4 mu[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = nt * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) mu[nt][r] = (((s_cs[0][col + r] + s_cs[1][col + r]) + s_cs[2][col + r]) + s_cs[3][col + r]) / (float)nrows;
      f32x4 q = {0.f, 0.f, 0.f, 0.f};
      if (rowok) { const f32x4 dlt = acc[0][nt] - mu[nt]; q = dlt * dlt; }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] += __shfl_xor(q[r], off);
      if (c == 0) *reinterpret_cast<f32x4*>(&s_cq[wave][col]) = q;
    }
    __syncthreads();
    if (wave == 0 && c == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = nt * 16 + 4 * g, nn = n0 + col;
        if (nn < N) {
          f32x4 qq;
#pragma unroll
          for (int r = 0; r < 4; ++r) qq[r] = ((s_cq[0][col + r] + s_cq[1][col + r]) + s_cq[2][col + r]) + s_cq[3][col + r];
          *reinterpret_cast<f32x4*>(stats + ((size_t)blockIdx.y * 2 + 0) * N + nn) = mu[nt];
          *reinterpret_cast<f32x4*>(stats + ((size_t)blockIdx.y * 2 + 1) * N + nn) = qq;
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + mt * 16 + c;
    if (m >= M) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nn = n0 + nt * 16 + 4 * g;
      if (nn >= N) continue;
      if (ksplit > 1) {
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.z * M + m) * N + nn) = acc[mt][nt];       // raw slice sums; gemm_reduce_kernel folds them in order
      } else {
        float* dst = C + (size_t)m * ldc + nn;
        f32x4 v = acc[mt][nt];
        if (accumulate) v = *reinterpret_cast<const f32x4*>(dst) + v;
        *reinterpret_cast<f32x4*>(dst) = v;
        if (Act) {
          f32x4 y;
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = act_fwd(v[r] + bias[nn + r], act);
          *reinterpret_cast<f32x4*>(Act + (size_t)m * ldc + nn) = y;
        }
      }
    }
  }
}

// The TN form (weight gradient dW [Kin, N] = X^T [Kin, R] . dZ [R, N], R = rows of the layer: up to 256 000 at 512 clips) on the same
// scheme.  The reduction index is the ROW of both operands, so a lane's four k values sit in four different rows: lane (g, c) reads
// X[16 j + 4 g + s][k0 + c] and dZ[16 j + 4 g + s][n0 + c], s = 0 .. 3 -- dword loads, each 64 contiguous bytes per lane group, every byte
// of both operands fetched once per output block.  A workgroup owns a KT x NT block of 16 x 16 output tiles and ONE slice of the rows
// (blockIdx.z); its four waves walk interleaved 16-row chunks of the slice and their partial tiles meet in LDS, added in wave order; the
// slice sums go to part[slice] and are folded in slice order by the launch that follows (gemm_reduce_kernel / fold_defer): fixed order
// everywhere.  The LDS-staged kernel ran the step's tall shapes (32 x 16 outputs over 256 000 rows: 48 MB of operands) at 2-10 TFLOP/s,
// i.e. 110-127 us for what the memory system delivers in ~15.
template <int KT, int NT>
__global__ __launch_bounds__(256) void train_gemm_tn2_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int lda,
                                                             int ldb, int ldc, int accumulate, int ksplit, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float s_tn[];                     // [3 waves][KT * NT][64 lanes][4]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int k0 = blockIdx.y * (16 * KT), n0 = blockIdx.x * (16 * NT);            // output block: rows k0.. of dW (columns of A), columns n0..
  // rows (the reduction, length K) of this slice, in 16-row chunks dealt round-robin to the four waves
  const int RC = (K + 15) >> 4;
  const int per = (RC + ksplit - 1) / ksplit;
  int jbeg = blockIdx.z * per, jend = (jbeg + per < RC) ? jbeg + per : RC;
  if (jbeg > jend) jbeg = jend;
  const unsigned bytesA = (unsigned)(((size_t)(K - 1) * lda + M) * sizeof(float));
  const unsigned bytesB = (unsigned)(((size_t)(K - 1) * ldb + N) * sizeof(float));
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, bytesB, 0x00020000);
  unsigned aoff[KT], boff[NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k = k0 + kt * 16 + c;
    aoff[kt] = (k < M) ? (unsigned)(((size_t)(4 * g) * lda + k) * sizeof(float)) : 0xFFFFFFF0u;       // (columns past the matrix: out of range, read 0)
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + nt * 16 + c;
    boff[nt] = (n < N) ? (unsigned)(((size_t)(4 * g) * ldb + n) * sizeof(float)) : 0xFFFFFFF0u;
  }
  const unsigned arow = (unsigned)lda * (unsigned)sizeof(float), brow = (unsigned)ldb * (unsigned)sizeof(float);
  f32x4 acc[KT][NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int D = 3;
  f32x4 aq[D][KT], bq[D][NT];
  // rows past the end of the matrices lie past the descriptors' ranges as long as the LAST row is the last thing in the buffer; a row
  // tail inside a chunk (K % 16 != 0) therefore reads zeros on both operands
  auto load = [&](int j, f32x4 (&av)[KT], f32x4 (&bv)[NT]) {
    const unsigned ra = 16u * (unsigned)j * arow, rb = 16u * (unsigned)j * brow;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) av[kt][s4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, aoff[kt], ra + (unsigned)s4 * arow, 0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[nt][s4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, boff[nt], rb + (unsigned)s4 * brow, 0));
    }
  };
  auto compute = [&](const f32x4 (&av)[KT], const f32x4 (&bv)[NT]) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[nt][s4], av[kt][s4], acc[kt][nt], 0, 0, 0);
  };
  // this wave's chunks: jbeg + wave, + 4, ...
  const int first = jbeg + wave;
  const int n = (jend > first) ? (jend - first + 3) / 4 : 0;
  if (n >= D) {
#pragma unroll
    for (int d = 0; d < D; ++d) load(first + 4 * d, aq[d], bq[d]);
    int i = 0;
    for (; i + 2 * D <= n; i += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        compute(aq[d], bq[d]);
        load(first + 4 * (i + D + d), aq[d], bq[d]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      compute(aq[d], bq[d]);
      if (i + D + d < n) load(first + 4 * (i + D + d), aq[d], bq[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
    i += D;
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (i + d < n) compute(aq[d], bq[d]);
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < n) load(first + 4 * d, aq[d], bq[d]);
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < n) compute(aq[d], bq[d]);
  }
  // the four waves' partial tiles, added in wave order by wave 0
  if (wave > 0) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(s_tn + ((size_t)((wave - 1) * KT * NT + kt * NT + nt) * 64 + lane) * 4) = acc[kt][nt];
  }
  __syncthreads();
  if (wave != 0) return;
  // lane (g, c) holds dW rows k0 + kt * 16 + c, columns n0 + nt * 16 + 4 g .. + 3
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k = k0 + kt * 16 + c;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 v = acc[kt][nt];
#pragma unroll
      for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const f32x4*>(s_tn + ((size_t)(w * KT * NT + kt * NT + nt) * 64 + lane) * 4);
      const int nn = n0 + nt * 16 + 4 * g;
      if (k >= M || nn >= N) continue;
      if (ksplit > 1) {
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.z * M + k) * N + nn) = v;
      } else {
        float* dst = C + (size_t)k * ldc + nn;                   // (a weight gradient: a view into the flat gradient buffer, not 16-byte aligned)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r] = accumulate ? dst[r] + v[r] : v[r];
      }
    }
  }
}

// ---- second stages: out[i] (+)= sum over slices z of part[z][i], fixed order ----
// A fold of many slices into FEW elements is pure latency when one thread walks all the slices of its element (a 256-slice fold of a 32 x 16
// weight gradient: two workgroups, 57-230 us of dependent loads, several times the GEMM that produced the slices).  So an element is shared
// by S threads: thread `sub` adds the slices [sub * per, (sub + 1) * per) in order, eight loads in flight, and the S sub-sums meet in LDS and
// are added in sub order.  S is a function of the element and slice counts only (fold_subs), the same in the immediate and in the deferred
// launch of a fold: bit-identical results either way, bit-reproducible from run to run.
__host__ __device__ inline int fold_subs(long n, int chunks) { return (chunks >= 32 && n <= 16384) ? 16 : ((chunks >= 16 && n <= 262144) ? 4 : 1); }
__host__ __device__ inline int fold_grid(long n, int S) { return (int)((n + 256 / S - 1) / (256 / S)); }

// block = 256 threads = (256 / S) element lanes x S subs; returns the element index of this thread (-1: none) and, in sub 0, its sum
__device__ __forceinline__ long fold_sum(const float* __restrict__ part, int chunks, size_t stride, long n, int block, int S, float* s_buf, float* sum) {
  const int E = 256 / S, e = (int)threadIdx.x % E, sub = (int)threadIdx.x / E;
  const long i = (long)block * E + e;
  const int per = (chunks + S - 1) / S;
  const int z0 = sub * per, z1 = (z0 + per < chunks) ? z0 + per : chunks;
  float v = 0.0f;
  if (i < n) {
    int z = z0;
    for (; z + 8 <= z1; z += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(z + u) * stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; z < z1; ++z) v += part[(size_t)z * stride + i];
  }
  if (S > 1) {                                   // (S is uniform over the workgroup)
    s_buf[threadIdx.x] = v;
    __syncthreads();
    if (sub == 0) {
      v = s_buf[e];
      for (int q = 1; q < S; ++q) v += s_buf[q * E + e];
    }
  }
  *sum = v;
  return (i < n && sub == 0) ? i : -1;
}

// C (+)= part[0] + part[1] + ... (fixed order); grid = fold_grid(M * N, S)
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ part, int ksplit, float* __restrict__ C, int M, int N, int ldc, int accumulate,
                                                          const float* __restrict__ bias, int act, float* __restrict__ Act, int S) {
  __shared__ float s_buf[256];
  const long total = (long)M * N;
  float v;
  const long i = fold_sum(part, ksplit, (size_t)total, total, (int)blockIdx.x, S, s_buf, &v);
  if (i < 0) return;
  const size_t o = (size_t)(i / N) * (size_t)ldc + (size_t)(i % N);
  if (accumulate) v += C[o];
  C[o] = v;
  if (Act) Act[o] = act_fwd(v + bias[i % N], act);
}

// out[i] (+)= scale * sum over chunks of part[chunk][i]  (fixed order): second stage of every cross-workgroup reduction below; grid = fold_grid(n, S)
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ part, int chunks, int n, float* __restrict__ out, float scale, int accumulate, int S) {
  __shared__ float s_buf[256];
  float v;
  const long i = fold_sum(part, chunks, (size_t)n, n, (int)blockIdx.x, S, s_buf, &v);
  if (i < 0) return;
  v *= scale;
  out[i] = accumulate ? out[i] + v : v;
}

// Several deferred second stages in ONE launch (mkws_op_fold_defer): block b serves the descriptor whose block range holds b.  Each
// output element is the same fixed-order sum the single-descriptor kernels (fold_partials_kernel, gemm_reduce_kernel without an epilogue)
// compute: bit-identical results, one launch instead of up to kMaxFolds.
constexpr int kMaxFolds = 24;
struct FoldDesc { const float* part; float* out; int chunks, n, N, ldc; float scale; int accumulate; int block0; int S; };
struct FoldBatch { FoldDesc d[kMaxFolds]; int n; };
__global__ __launch_bounds__(256) void fold_batch_kernel(FoldBatch fb) {
  __shared__ float s_buf[256];
  int k = 0;
  for (int j = 1; j < fb.n; ++j)
    if ((int)blockIdx.x >= fb.d[j].block0) k = j;
  const FoldDesc& d = fb.d[k];
  float v;
  const long i = fold_sum(d.part, d.chunks, (size_t)d.n, d.n, (int)blockIdx.x - d.block0, d.S, s_buf, &v);
  if (i < 0) return;
  v *= d.scale;
  const size_t o = (size_t)(i / d.N) * d.ldc + (i % d.N);
  d.out[o] = d.accumulate ? d.out[o] + v : v;
}

// ------------------------------------------------------------------------------------------------
// Thread layout of the row-streaming BatchNorm launches: a workgroup owns a 64-channel slab x a chunk of rows, 16 channel quads x 16 row lanes.
// A slab with fewer live quads (C = 16, 32, 96's second slab, ...) hands its idle quad lanes to the rows instead: 8 quads x 32 row lanes, or 4 x 64
// (round 4: the stem / block-1a / 2a layers -- the largest tensors of a step, C = 32 / 16 / 96 -- ran these launches with a quarter to a half of
// their threads masked off).  Full slabs keep 16 x 16: same sums, same order as before.
struct BnLanes { int QL, RL, ql, rl; };
__device__ __forceinline__ BnLanes bn_lanes(int C) {
  const int live = (C - (int)blockIdx.x * 64 + 3) / 4;          // live quads of this slab (> 0 by the grid)
  BnLanes b;
  b.QL = live > 8 ? 16 : (live > 4 ? 8 : 4);
  b.RL = 256 / b.QL;
  b.ql = (int)threadIdx.x & (b.QL - 1);
  b.rl = (int)threadIdx.x / b.QL;
  return b;
}

// Batch statistics of Z [M, C] per channel, two levels, fixed order.
// Level 1 (grid: 64-channel slabs x row chunks, block = 64 channel lanes x 4 row lanes): every workgroup computes the mean of ITS
// rows and the sum of squared deviations from that mean (two passes over its own rows, which sit in L2 after the first), and
// writes (mean_k, M2_k) to part[chunk][2][C].  Level 2 folds the chunks in index order with Chan's parallel-variance update
// (no E[z^2] - mean^2 cancellation) and, in the same launch, updates the moving statistics as Keras' fused BatchNorm does.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ Z, float* __restrict__ part, int M, int C) {
  // block = 16 channel quads (64 channels, one float4 per thread and row) x 16 row lanes.  ONE pass with shifted data: d = z - K
  // with K = the chunk's first row (any value near the mean removes the cancellation of the raw sum-of-squares formula);
  // mean_k = K + sum(d)/n,  M2_k = sum(d^2) - sum(d)^2/n.  The 16 row lanes fold in lane order.
  __shared__ float s1[256][4], s2[256][4];
  const BnLanes L = bn_lanes(C);
  const int ql = L.ql, rl = L.rl;
  const int c0 = blockIdx.x * 64 + 4 * ql;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = (r0 + per < M) ? r0 + per : M;
  const int n = r1 > r0 ? r1 - r0 : 0;
  const bool ok = c0 < C && n > 0;                     // C % 4 == 0
  f32x4 K = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    K = *reinterpret_cast<const f32x4*>(Z + (size_t)r0 * C + c0);
#pragma unroll 4      // four rows' loads in flight per thread (the adds keep their order: no fast-math)
    for (int r = r0 + rl; r < r1; r += L.RL) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(Z + (size_t)r * C + c0) - K;
      a1 += d;
      a2 += d * d;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { s1[rl * L.QL + ql][i] = a1[i]; s2[rl * L.QL + ql][i] = a2[i]; }
  __syncthreads();
  if (rl == 0 && ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t1 = s1[ql][i], t2 = s2[ql][i];
      for (int l = 1; l < L.RL; ++l) { t1 += s1[l * L.QL + ql][i]; t2 += s2[l * L.QL + ql][i]; }
      part[((size_t)blockIdx.y * 2 + 0) * C + c0 + i] = K[i] + t1 / (float)n;
      part[((size_t)blockIdx.y * 2 + 1) * C + c0 + i] = t2 - t1 * t1 / (float)n;
    }
  }
}

// Chan's parallel-variance fold of the chunk statistics [k0, k1) of one channel, in chunk order -> (count, mean, M2)
__device__ __forceinline__ void bn_fold_range(const float* __restrict__ part, int chunks, int M, int C, int c, int k0, int k1, float* n_out, float* mean_out,
                                              float* m2_out, int stat_rows) {
  const int per = stat_rows > 0 ? stat_rows : (M + chunks - 1) / chunks;      // rows per chunk: the producer's tiling (GEMM tiles of 64, depthwise chunks of 128), or even shares
  float n = 0.0f, mu = 0.0f, m2 = 0.0f;
  for (int k = k0; k < k1; ++k) {
    const int r0 = k * per, r1 = (r0 + per < M) ? r0 + per : M;
    const float nk = (float)(r1 > r0 ? r1 - r0 : 0);
    if (nk <= 0.0f) continue;
    const float muk = part[((size_t)k * 2 + 0) * C + c], m2k = part[((size_t)k * 2 + 1) * C + c];
    const float d = muk - mu, nn = n + nk;
    mu += d * (nk / nn);
    m2 += m2k + d * d * (n * nk / nn);
    n = nn;
  }
  *n_out = n; *mean_out = mu; *m2_out = m2;
}
// The statistics of the 64 channels of slab `slab`, by all 256 threads of a workgroup: thread group q = tid / 64 folds the q-th
// quarter of the chunks, then lane group 0 combines the four results in order.  The grouping is a function of `chunks` only, so
// every workgroup (and bn_stats_finalize_kernel) produces the same bits.  Results in s_mean / s_var [64]; needs a barrier after.
__device__ __forceinline__ void bn_fold_slab(const float* __restrict__ part, int chunks, int M, int C, int slab, float (*s_q)[64][3], float* s_mean, float* s_var,
                                             int stat_rows = 0) {
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = slab * 64 + cl;
  const int per4 = (chunks + 3) / 4;
  float n = 0.0f, mu = 0.0f, m2 = 0.0f;
  if (c < C) bn_fold_range(part, chunks, M, C, c, q * per4, (q + 1) * per4 < chunks ? (q + 1) * per4 : chunks, &n, &mu, &m2, stat_rows);
  s_q[q][cl][0] = n; s_q[q][cl][1] = mu; s_q[q][cl][2] = m2;
  __syncthreads();
  if (q == 0) {
    n = s_q[0][cl][0]; mu = s_q[0][cl][1]; m2 = s_q[0][cl][2];
    for (int j = 1; j < 4; ++j) {
      const float nk = s_q[j][cl][0];
      if (nk <= 0.0f) continue;
      const float d = s_q[j][cl][1] - mu, nn = n + nk;
      mu += d * (nk / nn);
      m2 += s_q[j][cl][2] + d * d * (n * nk / nn);
      n = nn;
    }
    s_mean[cl] = mu;
    s_var[cl] = m2 / (float)M;
  }
}

__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int chunks, int M, int C, float* __restrict__ mean,
                                                                float* __restrict__ var, float* __restrict__ mmean, float* __restrict__ mvar, float momentum) {
  __shared__ float s_q[4][64][3], s_mean[64], s_var[64];
  bn_fold_slab(part, chunks, M, C, blockIdx.x, s_q, s_mean, s_var);
  __syncthreads();
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x >= 64 || c >= C) return;
  const float mu = s_mean[threadIdx.x], v = s_var[threadIdx.x];
  mean[c] = mu;
  var[c] = v;                                          // biased variance: what the normalisation uses
  if (mmean) {                                         // moving averages: the variance enters Bessel-corrected (Keras' fused BN)
    const float bessel = M > 1 ? (float)M / (float)(M - 1) : 1.0f;
    mmean[c] = momentum * mmean[c] + (1.0f - momentum) * mu;
    mvar[c] = momentum * mvar[c] + (1.0f - momentum) * v * bessel;
  }
}

// Training-mode BN forward, second launch: every workgroup (64-channel slab x row chunk) first folds the chunk statistics of ITS
// 64 channels (same fixed order everywhere, so all workgroups agree bit for bit), then normalises / activates its rows; the
// workgroups of row chunk 0 also publish mean / var and update the moving statistics.  Saves the separate finalize launch.
__global__ __launch_bounds__(256) void bn_train_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ part, int chunks, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int act, float momentum, float* __restrict__ mmean,
                                                           float* __restrict__ mvar, float* __restrict__ mean, float* __restrict__ var, float* __restrict__ A, int M,
                                                           int C, const float* __restrict__ res, const float* __restrict__ row_scale, int group, int stat_rows) {
  __shared__ float s_q[4][64][3], s_sh[64], s_var[64], s_sc[64];
  const BnLanes L = bn_lanes(C);
  const int ql = L.ql, rl = L.rl;
  bn_fold_slab(part, chunks, M, C, blockIdx.x, s_q, s_sh, s_var, stat_rows);          // s_sh = mean
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < C) {
      const float mu = s_sh[threadIdx.x], v = s_var[threadIdx.x];
      s_sc[threadIdx.x] = rsqrtf(v + eps);             // xhat = (z - mu) * inv is formed exactly as bn_act_fwd_kernel does
      if (blockIdx.y == 0) {
        mean[c] = mu;
        var[c] = v;
        const float bessel = M > 1 ? (float)M / (float)(M - 1) : 1.0f;
        mmean[c] = momentum * mmean[c] + (1.0f - momentum) * mu;
        mvar[c] = momentum * mvar[c] + (1.0f - momentum) * v * bessel;
      }
    }
  }
  __syncthreads();
  const int c0 = blockIdx.x * 64 + 4 * ql;
  if (c0 >= C) return;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = (r0 + per < M) ? r0 + per : M;
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c0), bt = *reinterpret_cast<const f32x4*>(beta + c0);
  f32x4 inv, mu;
#pragma unroll
  for (int i = 0; i < 4; ++i) { inv[i] = s_sc[4 * ql + i]; mu[i] = s_sh[4 * ql + i]; }
#pragma unroll 4      // four rows' loads in flight per thread (the adds keep their order: no fast-math)
  for (int r = r0 + rl; r < r1; r += L.RL) {
    const f32x4 z = *reinterpret_cast<const f32x4*>(Z + (size_t)r * C + c0);
    f32x4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = act_fwd(g[i] * ((z[i] - mu[i]) * inv[i]) + bt[i], act);
    if (res) {                                         // residual branch of an MBConv block: out = keep[row / group] * BN(Z) + shortcut (mkws_op_bn_train_fwd_res)
      const float ks = row_scale ? row_scale[r / group] : 1.0f;
      const f32x4 x = *reinterpret_cast<const f32x4*>(res + (size_t)r * C + c0);
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = y[i] * ks + x[i];
    }
    *reinterpret_cast<f32x4*>(A + (size_t)r * C + c0) = y;
  }
}

// Per-channel sums over the rows of X [M, C] -> part[chunk][C] (MODE 0), or the same after an in-place elementwise backward
// of a bias + activation (MODE 1: X <- X * act'(Zb + bias), the dense / SE layers' backward)
template <int MODE>
__global__ __launch_bounds__(256) void col_sum_partial_kernel(float* __restrict__ X, const float* __restrict__ Zb, const float* __restrict__ bias, int act,
                                                              float* __restrict__ part, int M, int C) {
  __shared__ float s[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = (r0 + per < M) ? r0 + per : M;
  float acc = 0.0f;
  if (c < C) {
    const float bc = MODE == 1 ? bias[c] : 0.0f;
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 4) {
      const size_t i = (size_t)r * C + c;
      float x = X[i];
      if (MODE == 1) { x *= act_grad(Zb[i] + bc, act); X[i] = x; }
      acc += x;
    }
  }
  s[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < C) part[(size_t)blockIdx.y * C + c] = (s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]);
}

// A = act(gamma * (Z - mean) * rsqrt(var + eps) + beta)
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ mean, const float* __restrict__ var,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
                                                         float* __restrict__ A, size_t total, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float xh = (Z[i] - mean[c]) * rsqrtf(var[c] + eps);
    A[i] = act_fwd(gamma[c] * xh + beta[c], act);
  }
}

// backward step 1: dY = dA * act'(y) in place, and this row chunk's sums of dY and dY * xhat -> part[chunk][2][C]
// (block = 16 channel quads x 16 row lanes, float4 rows; lanes fold in lane order)
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ Z, const float* __restrict__ mean, const float* __restrict__ var,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
                                                                float* __restrict__ dA, float* __restrict__ part /*[chunks][2][C]*/, int M, int C,
                                                                const float* __restrict__ src, const float* __restrict__ row_scale,
                                                                const float* __restrict__ bcast, float bscale, int group) {
  __shared__ float s1[256][4], s2[256][4];
  const BnLanes L = bn_lanes(C);
  const int ql = L.ql, rl = L.rl;
  const int c0 = blockIdx.x * 64 + 4 * ql;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = (r0 + per < M) ? r0 + per : M;
  const bool ok = c0 < C;
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), vv = *reinterpret_cast<const f32x4*>(var + c0);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c0), b = *reinterpret_cast<const f32x4*>(beta + c0);
    f32x4 inv;
#pragma unroll
    for (int i = 0; i < 4; ++i) inv[i] = rsqrtf(vv[i] + eps);
#pragma unroll 4      // four rows' loads in flight per thread (the adds keep their order: no fast-math)
    for (int r = r0 + rl; r < r1; r += L.RL) {
      const size_t o = (size_t)r * C + c0;
      const f32x4 z = *reinterpret_cast<const f32x4*>(Z + o);
      // incoming gradient (mkws_op_bn_act_bwd_ex): src (or dA itself; neither = zero) * row_scale[row / group] + bcast[row / group][c] * bscale
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      if (src) d = *reinterpret_cast<const f32x4*>(src + o);
      if (row_scale) { const float ks = row_scale[r / group]; d = d * ks; }
      if (bcast) { const f32x4 bv = *reinterpret_cast<const f32x4*>(bcast + (size_t)(r / group) * C + c0); d = d + bv * bscale; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (z[i] - mu[i]) * inv[i];
        d[i] *= act_grad(g[i] * xh + b[i], act);
        a1[i] += d[i];
        a2[i] += d[i] * xh;
      }
      *reinterpret_cast<f32x4*>(dA + o) = d;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { s1[rl * L.QL + ql][i] = a1[i]; s2[rl * L.QL + ql][i] = a2[i]; }
  __syncthreads();
  if (rl == 0 && ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t1 = s1[ql][i], t2 = s2[ql][i];
      for (int l = 1; l < L.RL; ++l) { t1 += s1[l * L.QL + ql][i]; t2 += s2[l * L.QL + ql][i]; }
      part[((size_t)blockIdx.y * 2 + 0) * C + c0 + i] = t1;
      part[((size_t)blockIdx.y * 2 + 1) * C + c0 + i] = t2;
    }
  }
}

// backward step 2: dZ = gamma * inv * (dY - sum(dY)/M - xhat * sum(dY xhat)/M) in place.  Every workgroup (64-channel slab x row
// chunk) folds the chunk sums of ITS channels in chunk order first; the workgroups of row chunk 0 write dgamma / dbeta.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ Z, const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ gamma, float eps, float* __restrict__ dY, const float* __restrict__ part,
                                                           int chunks, float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C) {
  __shared__ float s_a[64], s_b[64], s_q[4][64][2];
  const BnLanes L = bn_lanes(C);
  const int ql = L.ql, rl = L.rl;
  {
    // thread group q folds the q-th quarter of the chunks, group 0 adds the four results in order (a function of `chunks` only)
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const int per4 = (chunks + 3) / 4, k0 = q * per4, k1 = (k0 + per4 < chunks) ? k0 + per4 : chunks;
    float t1 = 0.0f, t2 = 0.0f;
    if (c < C)
      for (int k = k0; k < k1; ++k) { t1 += part[((size_t)k * 2 + 0) * C + c]; t2 += part[((size_t)k * 2 + 1) * C + c]; }
    s_q[q][cl][0] = t1; s_q[q][cl][1] = t2;
    __syncthreads();
    if (q == 0 && c < C) {
      t1 = (s_q[0][cl][0] + s_q[1][cl][0]) + (s_q[2][cl][0] + s_q[3][cl][0]);
      t2 = (s_q[0][cl][1] + s_q[1][cl][1]) + (s_q[2][cl][1] + s_q[3][cl][1]);
      s_a[cl] = t1;
      s_b[cl] = t2;
      if (blockIdx.y == 0) { dbeta[c] = t1; dgamma[c] = t2; }
    }
  }
  __syncthreads();
  const int c0 = blockIdx.x * 64 + 4 * ql;
  if (c0 >= C) return;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = (r0 + per < M) ? r0 + per : M;
  const float invM = 1.0f / (float)M;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), vv = *reinterpret_cast<const f32x4*>(var + c0), g = *reinterpret_cast<const f32x4*>(gamma + c0);
  f32x4 inv, sa, sb;
#pragma unroll
  for (int i = 0; i < 4; ++i) { inv[i] = rsqrtf(vv[i] + eps); sa[i] = s_a[4 * ql + i] * invM; sb[i] = s_b[4 * ql + i] * invM; }
#pragma unroll 4      // four rows' loads in flight per thread (the adds keep their order: no fast-math)
  for (int r = r0 + rl; r < r1; r += L.RL) {
    const size_t o = (size_t)r * C + c0;
    const f32x4 z = *reinterpret_cast<const f32x4*>(Z + o);
    f32x4 d = *reinterpret_cast<const f32x4*>(dY + o);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (z[i] - mu[i]) * inv[i];
      d[i] = g[i] * inv[i] * (d[i] - sa[i] - xh * sb[i]);
    }
    *reinterpret_cast<f32x4*>(dY + o) = d;
  }
}

// Both backward steps in ONE launch for layers with few rows (M <= kBnSmallRows: the 4x3 and 2x2 image layers of a 64-clip step, 33 of
// the 49 BatchNorms; at 2 048-2 304 rows it measured ~17 us SLOWER per layer than the two launches): a workgroup owns 8 channels (two
// quads) of ALL rows -- 128 row lanes stride the rows --, so the column sums it needs never leave it: pass 1 writes dY and sums, a workgroup barrier, pass 2 re-reads its own dY (L2) and writes dZ.  At this size
// the two-launch form is two launch / drain latencies around ~3 us of work each.  Sums: the rows of a lane in row order, the 32 row
// lanes of a wave by butterfly, the four waves in order -- fixed.
constexpr int kBnSmallRows = 1024;
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
                                                           float* __restrict__ dA, int M, int C, const float* __restrict__ src,
                                                           const float* __restrict__ row_scale, const float* __restrict__ bcast, float bscale, int group,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float s_w[4][2][8];
  const int tid = threadIdx.x, ql = tid & 1, rl = tid >> 1, wave = tid >> 6;
  const int c0 = blockIdx.x * 8 + 4 * ql;
  const bool ok = c0 < C;
  const int cc = ok ? c0 : 0;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cc), vv = *reinterpret_cast<const f32x4*>(var + cc);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cc), b = *reinterpret_cast<const f32x4*>(beta + cc);
  f32x4 inv;
#pragma unroll
  for (int i = 0; i < 4; ++i) inv[i] = rsqrtf(vv[i] + eps);
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
#pragma unroll 6
    for (int r = rl; r < M; r += 128) {
      const size_t o = (size_t)r * C + c0;
      const f32x4 z = *reinterpret_cast<const f32x4*>(Z + o);
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      if (src) d = *reinterpret_cast<const f32x4*>(src + o);
      if (row_scale) { const float ks = row_scale[r / group]; d = d * ks; }
      if (bcast) { const f32x4 bv = *reinterpret_cast<const f32x4*>(bcast + (size_t)(r / group) * C + c0); d = d + bv * bscale; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (z[i] - mu[i]) * inv[i];
        d[i] *= act_grad(g[i] * xh + b[i], act);
        a1[i] += d[i];
        a2[i] += d[i] * xh;
      }
      *reinterpret_cast<f32x4*>(dA + o) = d;
    }
  }
#pragma unroll
  for (int m = 2; m < 64; m <<= 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) { a1[i] += __shfl_xor(a1[i], m); a2[i] += __shfl_xor(a2[i], m); }
  if ((tid & 63) < 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_w[wave][ql][i] = a1[i]; s_w[wave][ql][4 + i] = a2[i]; }
  }
  __syncthreads();
  f32x4 sa, sb;
  const float invM = 1.0f / (float)M;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t1 = ((s_w[0][ql][i] + s_w[1][ql][i]) + s_w[2][ql][i]) + s_w[3][ql][i];
    const float t2 = ((s_w[0][ql][4 + i] + s_w[1][ql][4 + i]) + s_w[2][ql][4 + i]) + s_w[3][ql][4 + i];
    if (rl == 0 && ok) { dbeta[c0 + i] = t1; dgamma[c0 + i] = t2; }
    sa[i] = t1 * invM; sb[i] = t2 * invM;
  }
  if (!ok) return;
#pragma unroll 6
  for (int r = rl; r < M; r += 128) {
    const size_t o = (size_t)r * C + c0;
    const f32x4 z = *reinterpret_cast<const f32x4*>(Z + o);
    f32x4 d = *reinterpret_cast<const f32x4*>(dA + o);        // this thread's own dY of pass 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (z[i] - mu[i]) * inv[i];
      d[i] = g[i] * inv[i] * (d[i] - sa[i] - xh * sb[i]);
    }
    *reinterpret_cast<f32x4*>(dA + o) = d;
  }
}

// moving = momentum * moving + (1 - momentum) * batch  (variance: Bessel-corrected, Keras' fused BN)
__global__ void bn_moving_kernel(float* __restrict__ mmean, float* __restrict__ mvar, const float* __restrict__ mean, const float* __restrict__ var,
                                 float momentum, float bessel, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    mmean[c] = momentum * mmean[c] + (1.0f - momentum) * mean[c];
    mvar[c] = momentum * mvar[c] + (1.0f - momentum) * var[c] * bessel;
  }
}

// ------------------------------------------------------------------------------------------------
// depthwise conv (explicit TF/Keras padding pt / pl), raw output
__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Z, int B, int H, int Wd,
                                                     int C, int k, int s, int pt, int pl, int Ho, int Wo) {
  const int cq = C / 4;
  const size_t total = (size_t)B * Ho * Wo * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i % cq);
    size_t p = i / cq;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int b = (int)(p / Ho);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ii = 0; ii < k; ++ii) {
      const int ih = oh * s - pt + ii;
      if (ih < 0 || ih >= H) continue;
      for (int jj = 0; jj < k; ++jj) {
        const int iw = ow * s - pl + jj;
        if (iw < 0 || iw >= Wd) continue;
        acc += *reinterpret_cast<const f32x4*>(X + (((size_t)b * H + ih) * Wd + iw) * C + 4 * q) * *reinterpret_cast<const f32x4*>(W + (size_t)(ii * k + jj) * C + 4 * q);
      }
    }
    *reinterpret_cast<f32x4*>(Z + i * 4) = acc;
  }
}

// The same convolution with the BatchNorm chunk statistics of its output (rows = output positions) in the same launch: grid (64-channel
// slabs, chunks of 128 rows), block = 16 channel quads x 16 row lanes, eight rows per thread kept in registers for the second pass
// (mean of the chunk, then squared deviations from it; lanes fold in lane order).  part[chunk][2][C] as bn_stats_partial_kernel writes it.
template <int KS>
__global__ __launch_bounds__(256) void dw_fwd_stats_kernel(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Z,
                                                           float* __restrict__ part, int B, int H, int Wd, int C, int s, int pt, int pl, int Ho, int Wo) {
  __shared__ float s1[16][16][4];
  const int ql = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + 4 * ql;
  const int M = B * Ho * Wo;
  const int r0 = blockIdx.y * 128, r1 = (r0 + 128 < M) ? r0 + 128 : M;
  const int n = r1 - r0;
  const bool ok = c0 < C;
  const int cc = ok ? c0 : 0;
  // the taps of this thread's channel quad stay in registers; a row's KS*KS input loads are issued together (clamped addresses, the
  // padding taps are replaced by zero after the load), so a thread pays one memory latency per row, not one per tap
  f32x4 wv[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) wv[t] = *reinterpret_cast<const f32x4*>(W + (size_t)t * C + cc);
  f32x4 z[8];
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int t = 0; t < 8; ++t) {
    const int r = r0 + rl + 16 * t;
    const bool live = ok && r < r1;
    int p = live ? r : r0;
    const int ow = p % Wo; p /= Wo;
    const int oh = p % Ho;
    const int b = p / Ho;
    const float* xb = X + (size_t)b * H * Wd * C + cc;
    f32x4 xv[KS * KS];
#pragma unroll
    for (int ii = 0; ii < KS; ++ii)
#pragma unroll
      for (int jj = 0; jj < KS; ++jj) {
        const int ih = oh * s - pt + ii, iw = ow * s - pl + jj;
        const bool in = ih >= 0 && ih < H && iw >= 0 && iw < Wd;
        const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= Wd ? Wd - 1 : iw);
        const f32x4 ld = *reinterpret_cast<const f32x4*>(xb + ((size_t)ihc * Wd + iwc) * C);
        // a padding tap is SELECTED away, not multiplied by zero: the clamped address may hold a non-finite activation, and Inf * 0 = NaN
        // would leak into an output that dw_fwd_kernel (which skips the tap) keeps finite
        xv[ii * KS + jj] = in ? ld : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KS * KS; ++q) acc += xv[q] * wv[q];               // same products in the same order as dw_fwd_kernel (padding taps add +-0)
    z[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live) {
      *reinterpret_cast<f32x4*>(Z + (size_t)r * C + c0) = acc;
      z[t] = acc;
      a1 += acc;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s1[rl][ql][i] = a1[i];
  __syncthreads();
  f32x4 mu;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t1 = s1[0][ql][i];
    for (int l = 1; l < 16; ++l) t1 += s1[l][ql][i];
    mu[i] = t1 / (float)n;
  }
  __syncthreads();
  f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if (r0 + rl + 16 * t < r1) { const f32x4 d = z[t] - mu; a2 += d * d; }
#pragma unroll
  for (int i = 0; i < 4; ++i) s1[rl][ql][i] = a2[i];
  __syncthreads();
  if (rl == 0 && ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t2 = s1[0][ql][i];
      for (int l = 1; l < 16; ++l) t2 += s1[l][ql][i];
      part[((size_t)blockIdx.y * 2 + 0) * C + c0 + i] = mu[i];
      part[((size_t)blockIdx.y * 2 + 1) * C + c0 + i] = t2;
    }
  }
}

// dX[b,ih,iw,c] = sum_{i,j} dZ[b,oh,ow,c] W[i,j,c] over the outputs that read this input
__global__ __launch_bounds__(256) void dw_bwd_input_kernel(const float* __restrict__ dZ, const float* __restrict__ W, float* __restrict__ dX, int B, int H,
                                                           int Wd, int C, int k, int s, int pt, int pl, int Ho, int Wo) {
  const int cq = C / 4;
  const size_t total = (size_t)B * H * Wd * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i % cq);
    size_t p = i / cq;
    const int iw = (int)(p % Wd); p /= Wd;
    const int ih = (int)(p % H);
    const int b = (int)(p / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ii = 0; ii < k; ++ii) {
      const int t = ih + pt - ii;
      if (t < 0 || t % s != 0) continue;
      const int oh = t / s;
      if (oh >= Ho) continue;
      for (int jj = 0; jj < k; ++jj) {
        const int u = iw + pl - jj;
        if (u < 0 || u % s != 0) continue;
        const int ow = u / s;
        if (ow >= Wo) continue;
        acc += *reinterpret_cast<const f32x4*>(dZ + (((size_t)b * Ho + oh) * Wo + ow) * C + 4 * q) * *reinterpret_cast<const f32x4*>(W + (size_t)(ii * k + jj) * C + 4 * q);
      }
    }
    *reinterpret_cast<f32x4*>(dX + i * 4) = acc;
  }
}

// dW[i,j,c] = sum_{b,oh,ow} dZ[b,oh,ow,c] X[b, oh*s-pt+i, ow*s-pl+j, c]: block = 16 channel quads x 16 position lanes (round 2 used
// 64 x 4: with 24..60 quads per layer most lanes idled), grid (quad slabs of 16, position chunks); the position lanes fold in lane
// order and every workgroup stores its chunk's slab of partial sums (folded in chunk order by fold_partials_kernel).
template <int KS>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const float* __restrict__ X, const float* __restrict__ dZ, float* __restrict__ part /*[chunks][KS*KS*C]*/,
                                                            int B, int H, int Wd, int C, int s, int pt, int pl, int Ho, int Wo) {
  // thread = (channel quad ql of 16, position lane pl16 of 16); a workgroup owns 16 quads x one chunk of output positions.
  // Round 4: the 25 tap sums used to leave the workgroup one tap at a time (2 barriers + a 16-deep serial sum each: 50 barriers for a
  // 5x5 kernel, 42.7 us per call at 64 clips).  Now the four position lanes of a wave fold by two DPP-style shuffles, the four waves
  // through LDS with ONE barrier, and the chunks are finer (launcher), so that the grid covers the chip.  Fixed order throughout.
  __shared__ float red[4][KS * KS][16][4];
  const int ql = threadIdx.x & 15, pl16 = threadIdx.x >> 4;
  const int q = blockIdx.x * 16 + ql;
  const bool qok = q * 4 < C;
  const int npos = B * Ho * Wo;
  const int per = (npos + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = (p0 + per < npos) ? p0 + per : npos;
  f32x4 acc[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (qok) {
    for (int p = p0 + pl16; p < p1; p += 16) {
      const int ow = p % Wo, oh = (p / Wo) % Ho, b = p / (Wo * Ho);
      const f32x4 dz = *reinterpret_cast<const f32x4*>(dZ + (size_t)p * C + 4 * q);
#pragma unroll
      for (int ii = 0; ii < KS; ++ii) {
        const int ih = oh * s - pt + ii;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int jj = 0; jj < KS; ++jj) {
          const int iw = ow * s - pl + jj;
          if (iw < 0 || iw >= Wd) continue;
          acc[ii * KS + jj] += dz * *reinterpret_cast<const f32x4*>(X + (((size_t)b * H + ih) * Wd + iw) * C + 4 * q);
        }
      }
    }
  }
  // lanes ql, ql + 16, ql + 32, ql + 48 of a wave hold the same quad: (l0 + l1) + (l2 + l3), then the four waves in wave order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[t][r];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      acc[t][r] = v;
    }
    if (lane < 16) *reinterpret_cast<f32x4*>(&red[wave][t][ql][0]) = acc[t];
  }
  __syncthreads();
  float* slab = part + (size_t)blockIdx.y * KS * KS * C;
  for (int i = threadIdx.x; i < KS * KS * 64; i += 256) {
    const int t = i >> 6, qq = (i >> 2) & 15, r = i & 3;
    const int ch = (blockIdx.x * 16 + qq) * 4 + r;
    if (ch < C) slab[(size_t)t * C + ch] = (red[0][t][qq][r] + red[1][t][qq][r]) + (red[2][t][qq][r] + red[3][t][qq][r]);
  }
}

// ------------------------------------------------------------------------------------------------
// stem: Rescaling(1/255) + Normalization + ZeroPadding2D(((1,1),(0,1))) + Conv2D(32, 3, s2, valid), raw output [B,25,20,32]
__device__ __forceinline__ float stem_in(const float* img, int ih, int iw, float nm, float ns) {
  return (ih >= 0 && ih < 49 && iw < 40) ? __fdiv_rn(img[ih * 40 + iw] * (1.0f / 255.0f) - nm, ns) : 0.0f;
}
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ spec, const float* __restrict__ W /*[9][32]*/, float nm, float ns,
                                                       float* __restrict__ Z, int B) {
  const size_t total = (size_t)B * 500 * 8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i & 7);
    const size_t pix = i >> 3;
    const int b = (int)(pix / 500), r = (int)(pix % 500), oh = r / 20, ow = r % 20;
    const float* img = spec + (size_t)b * 1960;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) acc += *reinterpret_cast<const f32x4*>(W + (ii * 3 + jj) * 32 + 4 * q) * stem_in(img, oh * 2 - 1 + ii, ow * 2 + jj, nm, ns);
    *reinterpret_cast<f32x4*>(Z + pix * 32 + 4 * q) = acc;
  }
}
__global__ __launch_bounds__(256) void stem_bwd_weight_kernel(const float* __restrict__ spec, const float* __restrict__ dZ, float nm, float ns,
                                                              float* __restrict__ part /*[gridDim.x][9][32]*/, int B) {
  __shared__ float red[32][8][4];
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;            // 8 quads x 32 pixel lanes
  const int npix = B * 500;
  const int per = (npix + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int p = p0 + pl; p < p1; p += 32) {
    const int b = p / 500, r = p % 500, oh = r / 20, ow = r % 20;
    const float* img = spec + (size_t)b * 1960;
    const f32x4 dz = *reinterpret_cast<const f32x4*>(dZ + (size_t)p * 32 + 4 * q);
#pragma unroll
    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) acc[ii * 3 + jj] += dz * stem_in(img, oh * 2 - 1 + ii, ow * 2 + jj, nm, ns);
  }
  float* slab = part + (size_t)blockIdx.x * 288;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[pl][q][r] = acc[t][r];
    __syncthreads();
    if (pl == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = red[0][q][r];
        for (int l = 1; l < 32; ++l) v += red[l][q][r];
        slab[t * 32 + 4 * q + r] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// squeeze-excite pieces and small elementwise operators
// mean[b,c] = (1/HW) sum_hw A[b,hw,c]
__global__ __launch_bounds__(256) void pool_hw_kernel(const float* __restrict__ A, float* __restrict__ mean, int B, int HW, int C) {
  // grid (channel-quad slabs of 64, clips); block = 64 quads x 4 position lanes (coalesced rows, 4x the parallelism of one thread
  // per (clip, quad)); the lanes fold in fixed order
  __shared__ float red[4][64][4];
  const int ql = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + ql, b = blockIdx.y;
  const bool ok = q * 4 < C;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (ok)
    for (int p = pl; p < HW; p += 4) s += *reinterpret_cast<const f32x4*>(A + ((size_t)b * HW + p) * C + 4 * q);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[pl][ql][r] = s[r];
  __syncthreads();
  if (pl == 0 && ok) {
    f32x4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = (red[0][ql][r] + red[1][ql][r]) + (red[2][ql][r] + red[3][ql][r]);
    *reinterpret_cast<f32x4*>(mean + (size_t)b * C + 4 * q) = t * (1.0f / (float)HW);
  }
}
// out[b,hw,c] = A[b,hw,c] * g[b,c]
__global__ __launch_bounds__(256) void scale_channels_kernel(const float* __restrict__ A, const float* __restrict__ g, float* __restrict__ out, int B, int HW, int C) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t b = i / ((size_t)HW * C);
    out[i] = A[i] * g[b * C + c];
  }
}
// dA[b,hw,c] = dOut * g[b,c];  dg[b,c] = sum_hw dOut * A
__global__ __launch_bounds__(256) void se_bwd_kernel(const float* __restrict__ A, const float* __restrict__ g, const float* __restrict__ dOut, float* __restrict__ dA,
                                                     float* __restrict__ dg, int B, int HW, int C) {
  __shared__ float red[4][64][4];
  const int ql = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + ql, b = blockIdx.y;
  const bool ok = q * 4 < C;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (size_t)b * C + 4 * q);
    for (int p = pl; p < HW; p += 4) {
      const size_t o = ((size_t)b * HW + p) * C + 4 * q;
      const f32x4 d = *reinterpret_cast<const f32x4*>(dOut + o);
      s += d * *reinterpret_cast<const f32x4*>(A + o);
      *reinterpret_cast<f32x4*>(dA + o) = d * gv;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[pl][ql][r] = s[r];
  __syncthreads();
  if (pl == 0 && ok) {
    f32x4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = (red[0][ql][r] + red[1][ql][r]) + (red[2][ql][r] + red[3][ql][r]);
    *reinterpret_cast<f32x4*>(dg + (size_t)b * C + 4 * q) = t;
  }
}
// ------------------------------------------------------------------------------------------------
// The squeeze-excite branch of one block in two launches forward and three backward.  The operator-by-operator form above costs 4-5
// launches forward (pool, two dense layers -- the first split over K with a fold --, excite multiply) and 7 backward per block; at batch
// 64 every one of them is a few microseconds of work behind ~5 us of launch / drain latency.  Here a workgroup owns (clip, slab of 128
// channels): it pools / gates / scales its slab of the clip's [HW, C] activation and multiplies it with its slab of the two tiny dense
// layers (C x se, se <= 48); the sums over C cross slabs through a [B, slabs, se] scratch that the next launch folds in slab order.
// (One workgroup per clip -- one launch per direction -- was built first: 51 us / 33 us per call, the clip's workgroup walks 2 x 220 KB
// of weights alone, behind a CU's ~41 GB/s L2 stream.)  All sums keep a fixed order.
constexpr int kSeMaxC = 1152, kSeMaxSe = 48, kSeSlab = 128;
__device__ __forceinline__ float wave_sum_fixed(float v) {          // butterfly: the same order on every call
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// forward 1: mean = pool(A) for the slab; Zpart[b, slab, n] = sum_{c in slab} mean[c] Wr[c, n]
__global__ __launch_bounds__(256) void se_fwd_pool_kernel(const float* __restrict__ A, const float* __restrict__ Wr /*[C,se]*/, float* __restrict__ mean,
                                                          float* __restrict__ Zpart, int HW, int C, int se) {
  __shared__ __attribute__((aligned(16))) float s_red[256 * 4];
  __shared__ __attribute__((aligned(16))) float s_mean[kSeSlab];
  __shared__ float s_part[16][64];
  const int tid = threadIdx.x, slab = blockIdx.x, b = blockIdx.y, nslab = gridDim.x;
  const int c0 = slab * kSeSlab, SC = (C - c0 < kSeSlab) ? C - c0 : kSeSlab;
  const int QS = SC >> 2, PL = 256 / QS;
  const float* Ab = A + (size_t)b * HW * C + c0;
  const int q = tid % QS, pl = tid / QS;
  if (pl < PL) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int p = pl; p < HW; p += PL) s += *reinterpret_cast<const f32x4*>(Ab + (size_t)p * C + 4 * q);
    *reinterpret_cast<f32x4*>(s_red + (size_t)(pl * QS + q) * 4) = s;
  }
  __syncthreads();
  if (tid < QS) {
    f32x4 t = *reinterpret_cast<const f32x4*>(s_red + (size_t)tid * 4);
    for (int l = 1; l < PL; ++l) t += *reinterpret_cast<const f32x4*>(s_red + (size_t)(l * QS + tid) * 4);
    t = t * (1.0f / (float)HW);
    *reinterpret_cast<f32x4*>(s_mean + 4 * tid) = t;
    *reinterpret_cast<f32x4*>(mean + (size_t)b * C + c0 + 4 * tid) = t;
  }
  __syncthreads();
  const int NL = (se <= 16) ? 16 : (se <= 32) ? 32 : 64, KL = 256 / NL;
  const int n = tid % NL, kl = tid / NL;
  float v = 0.0f;
  if (n < se) {
#pragma unroll 8
    for (int c = kl; c < SC; c += KL) v += s_mean[c] * Wr[(size_t)(c0 + c) * se + n];
  }
  s_part[kl][n] = v;
  __syncthreads();
  if (tid < se) {
    float t = s_part[0][tid];
    for (int l = 1; l < KL; ++l) t += s_part[l][tid];
    Zpart[((size_t)b * nslab + slab) * se + tid] = t;
  }
}
// forward 2: Yr = sum_slab Zpart + br; R = swish(Yr); G = sigmoid(R We + be) for the slab; out = A * G
__global__ __launch_bounds__(256) void se_fwd_gate_kernel(const float* __restrict__ A, const float* __restrict__ Zpart, const float* __restrict__ br,
                                                          const float* __restrict__ We /*[se,C]*/, const float* __restrict__ be, float* __restrict__ Yr,
                                                          float* __restrict__ R, float* __restrict__ G, float* __restrict__ out, int HW, int C, int se) {
  __shared__ float s_R[kSeMaxSe];
  __shared__ float s_half[2][kSeSlab];
  __shared__ __attribute__((aligned(16))) float s_G[kSeSlab];
  const int tid = threadIdx.x, slab = blockIdx.x, b = blockIdx.y, nslab = gridDim.x;
  const int c0 = slab * kSeSlab, SC = (C - c0 < kSeSlab) ? C - c0 : kSeSlab;
  if (tid < se) {
    float y = Zpart[(size_t)b * nslab * se + tid];
    for (int l = 1; l < nslab; ++l) y += Zpart[((size_t)b * nslab + l) * se + tid];
    y += br[tid];
    const float r = act_fwd(y, TA_SWISH);
    s_R[tid] = r;
    if (slab == 0) { Yr[(size_t)b * se + tid] = y; R[(size_t)b * se + tid] = r; }
  }
  __syncthreads();
  {
    const int c = tid & (kSeSlab - 1), h = tid >> 7;
    const int nh = (se + 1) / 2, n0 = h * nh, n1 = (n0 + nh < se) ? n0 + nh : se;
    float z = 0.0f;
    if (c < SC) {
#pragma unroll 8
      for (int n = n0; n < n1; ++n) z += s_R[n] * We[(size_t)n * C + c0 + c];
    }
    s_half[h][c] = z;
  }
  __syncthreads();
  if (tid < SC) {
    const float g = sigm(be[c0 + tid] + (s_half[0][tid] + s_half[1][tid]));
    G[(size_t)b * C + c0 + tid] = g;
    s_G[tid] = g;
  }
  __syncthreads();
  const int QS = SC >> 2;
  const float* Ab = A + (size_t)b * HW * C + c0;
  float* ob = out + (size_t)b * HW * C + c0;
#pragma unroll 4
  for (int i = tid; i < HW * QS; i += 256) {
    const int p = i / QS, q = i - p * QS;
    *reinterpret_cast<f32x4*>(ob + (size_t)p * C + 4 * q) = *reinterpret_cast<const f32x4*>(Ab + (size_t)p * C + 4 * q) * *reinterpret_cast<const f32x4*>(s_G + 4 * q);
  }
}
// backward 1: dA = dOut * G; dG = sum_hw dOut * A; dYg = dG * G (1 - G) for the slab; dRpart[b, slab, n] = sum_{c in slab} dYg[c] We[n, c]
__global__ __launch_bounds__(256) void se_bwd_gate_kernel(const float* __restrict__ A, const float* __restrict__ G, const float* __restrict__ dOut,
                                                          const float* __restrict__ We, float* __restrict__ dA, float* __restrict__ dYg,
                                                          float* __restrict__ dRpart, int HW, int C, int se) {
  __shared__ __attribute__((aligned(16))) float s_red[256 * 4];
  __shared__ __attribute__((aligned(16))) float s_g[kSeSlab];
  const int tid = threadIdx.x, slab = blockIdx.x, b = blockIdx.y, nslab = gridDim.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = slab * kSeSlab, SC = (C - c0 < kSeSlab) ? C - c0 : kSeSlab;
  const int QS = SC >> 2, PL = 256 / QS;
  const float* Ab = A + (size_t)b * HW * C + c0;
  const float* Db = dOut + (size_t)b * HW * C + c0;
  float* dAb = dA + (size_t)b * HW * C + c0;
  const float* Gb = G + (size_t)b * C + c0;
  const int q = tid % QS, pl = tid / QS;
  if (pl < PL) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(Gb + 4 * q);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int p = pl; p < HW; p += PL) {
      const size_t o = (size_t)p * C + 4 * q;
      const f32x4 d = *reinterpret_cast<const f32x4*>(Db + o);
      s += d * *reinterpret_cast<const f32x4*>(Ab + o);
      *reinterpret_cast<f32x4*>(dAb + o) = d * gv;
    }
    *reinterpret_cast<f32x4*>(s_red + (size_t)(pl * QS + q) * 4) = s;
  }
  __syncthreads();
  if (tid < QS) {
    f32x4 t = *reinterpret_cast<const f32x4*>(s_red + (size_t)tid * 4);
    for (int l = 1; l < PL; ++l) t += *reinterpret_cast<const f32x4*>(s_red + (size_t)(l * QS + tid) * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(Gb + 4 * tid);
    t = t * gv * ((f32x4){1.f, 1.f, 1.f, 1.f} - gv);
    *reinterpret_cast<f32x4*>(s_g + 4 * tid) = t;
    *reinterpret_cast<f32x4*>(dYg + (size_t)b * C + c0 + 4 * tid) = t;
  }
  __syncthreads();
  const float g0 = (lane < SC) ? s_g[lane] : 0.0f, g1 = (lane + 64 < SC) ? s_g[lane + 64] : 0.0f;
  for (int n = wave; n < se; n += 4) {
    const float* w = We + (size_t)n * C + c0;
    float v = ((lane < SC) ? g0 * w[lane] : 0.0f) + ((lane + 64 < SC) ? g1 * w[lane + 64] : 0.0f);
    v = wave_sum_fixed(v);
    if (lane == 0) dRpart[((size_t)b * nslab + slab) * se + n] = v;
  }
}
// backward 2: dYr = (sum_slab dRpart) * swish'(Yr); dmean[c] = sum_n dYr[n] Wr[c, n] for the slab
__global__ __launch_bounds__(128) void se_bwd_squeeze_kernel(const float* __restrict__ dRpart, const float* __restrict__ Yr, const float* __restrict__ Wr,
                                                             float* __restrict__ dYr, float* __restrict__ dmean, int C, int se) {
  __shared__ float s_dr[kSeMaxSe];
  const int tid = threadIdx.x, slab = blockIdx.x, b = blockIdx.y, nslab = gridDim.x;
  const int c0 = slab * kSeSlab, SC = (C - c0 < kSeSlab) ? C - c0 : kSeSlab;
  if (tid < se) {
    float v = dRpart[(size_t)b * nslab * se + tid];
    for (int l = 1; l < nslab; ++l) v += dRpart[((size_t)b * nslab + l) * se + tid];
    v *= act_grad(Yr[(size_t)b * se + tid], TA_SWISH);
    s_dr[tid] = v;
    if (slab == 0) dYr[(size_t)b * se + tid] = v;
  }
  __syncthreads();
  if (tid < SC) {
    const float* w = Wr + (size_t)(c0 + tid) * se;
    float v = 0.0f;
#pragma unroll 8
    for (int n = 0; n < se; ++n) v += s_dr[n] * w[n];
    dmean[(size_t)b * C + c0 + tid] = v;
  }
}
// backward, weights (sums over the batch, row order): dWe = R^T dYg, dbe = colsum(dYg), dWr = mean^T dYr, dbr = colsum(dYr).
// grid (64-channel slabs, groups of 8 squeezed units); block = 64 channels x 4 unit lanes, two units per thread.
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ mean, const float* __restrict__ R, const float* __restrict__ dYg,
                                                       const float* __restrict__ dYr, float* __restrict__ dWr, float* __restrict__ dbr, float* __restrict__ dWe,
                                                       float* __restrict__ dbe, int B, int C, int se, int rows_per_chunk) {
  // blockIdx.z = chunk of batch rows.  One chunk: the outputs are the gradients themselves; several: every output pointer is the base of a
  // [chunks][n] array of partial sums that a fixed-order fold adds up (host: se_wgrad_launch).
  __shared__ float s_R[64][8], s_d[64][8];
  const int tid = threadIdx.x, cl = tid & 63, nl = tid >> 6;
  const int c = blockIdx.x * 64 + cl, n0 = blockIdx.y * 8;
  const bool okc = c < C;
  const int rb = blockIdx.z * rows_per_chunk, re = (rb + rows_per_chunk < B) ? rb + rows_per_chunk : B;
  dWe += (size_t)blockIdx.z * se * C; dWr += (size_t)blockIdx.z * se * C; dbe += (size_t)blockIdx.z * C; dbr += (size_t)blockIdx.z * se;
  float ae[2] = {0.f, 0.f}, ar[2] = {0.f, 0.f}, sbe = 0.0f, sbr = 0.0f;
  for (int b0 = rb; b0 < re; b0 += 64) {
    const int nb = (re - b0 < 64) ? re - b0 : 64;
    __syncthreads();
    for (int i = tid; i < 64 * 8; i += 256) {
      const int r = i >> 3, j = i & 7;
      const bool ok = r < nb && n0 + j < se;
      s_R[r][j] = ok ? R[(size_t)(b0 + r) * se + n0 + j] : 0.0f;
      s_d[r][j] = ok ? dYr[(size_t)(b0 + r) * se + n0 + j] : 0.0f;
    }
    __syncthreads();
    if (okc) {
#pragma unroll 8      // eight rows' loads in flight (the adds keep their order)
      for (int r = 0; r < nb; ++r) {
        const float g = dYg[(size_t)(b0 + r) * C + c], m = mean[(size_t)(b0 + r) * C + c];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          ae[j] += s_R[r][nl * 2 + j] * g;
          ar[j] += m * s_d[r][nl * 2 + j];
        }
        sbe += g;
      }
    }
    if (blockIdx.x == 0 && tid < 8)
      for (int r = 0; r < nb; ++r) sbr += s_d[r][tid];
  }
  if (okc) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + nl * 2 + j;
      if (n < se) { dWe[(size_t)n * C + c] = ae[j]; dWr[(size_t)c * se + n] = ar[j]; }
    }
    if (blockIdx.y == 0 && nl == 0) dbe[c] = sbe;
  }
  if (blockIdx.x == 0 && tid < 8 && n0 + tid < se) dbr[n0 + tid] = sbr;
}
// X[b,hw,c] += v[b,c] * scale   (gradient of a mean over hw; also: broadcast add)
__global__ __launch_bounds__(256) void add_bcast_kernel(float* __restrict__ X, const float* __restrict__ v, float scale, int B, int HW, int C) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t b = i / ((size_t)HW * C);
    X[i] += v[b * C + c] * scale;
  }
}
// A = act(Z + bias)
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ bias, int act, float* __restrict__ A, size_t total, int N) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) A[i] = act_fwd(Z[i] + bias[i % N], act);
}
// out[b, :] = a[b, :] * s[b] + c[b, :]   (c may be NULL)  -- drop-connect + residual add, and its backward
__global__ __launch_bounds__(256) void row_scale_add_kernel(const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ c, float* __restrict__ out,
                                                            int B, size_t per_row) {
  const size_t total = (size_t)B * per_row;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float v = a[i] * s[i / per_row];
    out[i] = c ? v + c[i] : v;
  }
}
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] += alpha * x[i];
}
// Keras Adam over a flat buffer (same arithmetic as head_adam_kernel)
__global__ __launch_bounds__(256) void train_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                         float lr_t, float b1, float b2, float eps, float grad_scale) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
    m[i] = mi; v[i] = vi;
    p[i] -= (mi * lr_t) / (sqrtf(vi) + eps);
  }
}

// Sparse categorical cross-entropy FROM LOGITS over N classes (the 761-way classifier the reference trains the embedding with,
// train_multilingual_embedding.py:84-93): one workgroup per row.  rowstat[b] = {-log softmax(z)[y], argmax == y}, and in place
// z <- (softmax(z) - onehot(y)) * inv_batch (the gradient of the MEAN loss).  Row reductions fold in a fixed tree.
__global__ __launch_bounds__(256) void softmax_ce_kernel(float* __restrict__ Z, const int* __restrict__ labels, float* __restrict__ rowstat, int N, float inv_batch) {
  __shared__ float s_v[256];
  __shared__ int s_i[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* z = Z + (size_t)b * N;
  float mx = -3.0e38f;
  int arg = 0;
  for (int i = tid; i < N; i += 256) { const float v = z[i]; if (v > mx) { mx = v; arg = i; } }      // first maximum of this thread's (ascending) indices
  s_v[tid] = mx; s_i[tid] = arg;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st) {
      const float o = s_v[tid + st]; const int oi = s_i[tid + st];
      if (o > s_v[tid] || (o == s_v[tid] && oi < s_i[tid])) { s_v[tid] = o; s_i[tid] = oi; }           // ties -> lowest index, like argmax
    }
    __syncthreads();
  }
  mx = s_v[0]; arg = s_i[0];
  __syncthreads();
  float sum = 0.0f;
  for (int i = tid; i < N; i += 256) sum += expf(z[i] - mx);
  s_v[tid] = sum;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st) s_v[tid] += s_v[tid + st];
    __syncthreads();
  }
  sum = s_v[0];
  const int y = labels[b];
  const float inv = 1.0f / sum;
  if (tid == 0) {
    rowstat[2 * b] = (logf(sum) + mx) - z[y];
    rowstat[2 * b + 1] = (arg == y) ? 1.0f : 0.0f;
  }
  __syncthreads();                                       // z[y] read above before anyone overwrites it
  for (int i = tid; i < N; i += 256) z[i] = (expf(z[i] - mx) * inv - (i == y ? 1.0f : 0.0f)) * inv_batch;
}
// stats[0] = sum of row losses, stats[1] = number of correct rows (rows folded in index order)
__global__ void rowstat_fold_kernel(const float* __restrict__ rowstat, int B, float* __restrict__ stats) {
  if (threadIdx.x < 2 && blockIdx.x == 0) {
    float v = 0.0f;
    for (int b = 0; b < B; ++b) v += rowstat[2 * b + threadIdx.x];
    stats[threadIdx.x] = v;
  }
}

// Adam whose step index lives in device memory: a captured hipGraph replays the same launch every step, so lr_t cannot be a kernel
// argument.  step_inc_kernel runs once per step before the Adam launches that share the counter.
__global__ void step_inc_kernel(int* __restrict__ step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }
__global__ __launch_bounds__(256) void train_adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                             float lr, float b1, float b2, float eps, const int* __restrict__ step, float grad_scale) {
  __shared__ float s_lr;
  if (threadIdx.x == 0) {
    const int t = *step;
    s_lr = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
  }
  __syncthreads();
  const float lr_t = s_lr;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
    m[i] = mi; v[i] = vi;
    p[i] -= (mi * lr_t) / (sqrtf(vi) + eps);
  }
}

inline int grid_for(size_t total, int cap = 8192) {
  size_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

}  // namespace mkws

using namespace mkws;

#define MKWS_REQ(cond, ...) do { if (!(cond)) return fail(MKWS_ERR_INVALID_ARG, __VA_ARGS__); } while (0)

// Context of the training operators: the scratch arena of the fixed-order reductions and the queue of deferred second stages.  Every
// host thread has a default context (mkws_op_set_scratch); a trainer that wants its state independent of the thread it happens to run
// on -- two trainers on one thread, one trainer handed from thread to thread -- owns a context (mkws_train_ctx_create) and binds it
// (mkws_train_ctx_bind) before its mkws_op_* calls.
struct mkws_train_ctx {
  float* scratch = nullptr;          // caller-owned device memory
  size_t floats = 0;
  size_t bump = 0;                   // floats held by queued folds: while folds are queued the arena is handed out from a bump pointer
  int defer = 0;
  mkws::FoldBatch fb;                // fb.n descriptors queued
  int fold_blocks = 0;
  hipEvent_t ev[8] = {};             // mkws_op_stream_wait: created on first use, used round-robin
  int evpos = 0;
};
namespace {
thread_local mkws_train_ctx g_default_ctx;
thread_local mkws_train_ctx* g_ctx = nullptr;
inline mkws_train_ctx& ctx() { return g_ctx ? *g_ctx : g_default_ctx; }
inline float* scratch(size_t floats) { mkws_train_ctx& c = ctx(); return (c.scratch && floats <= c.floats) ? c.scratch : nullptr; }   // capacity check (base of the arena)
inline int fold_flush(hipStream_t s) {
  mkws_train_ctx& c = ctx();
  if (c.fb.n > 0) hipLaunchKernelGGL(fold_batch_kernel, dim3(c.fold_blocks), dim3(256), 0, s, c.fb);
  c.fb.n = 0; c.fold_blocks = 0; c.bump = 0;
  return MKWS_OK;
}
// every operator takes its scratch through scratch_at(), which flushes the queued folds when the arena is full
inline float* scratch_at(size_t floats, hipStream_t s) {
  mkws_train_ctx& c = ctx();
  if (!c.scratch || floats > c.floats) return nullptr;
  if (c.bump + floats > c.floats) fold_flush(s);
  return c.scratch + c.bump;
}
// second stage out[(i / N) * ldc + i % N] (+)= scale * sum_z part[z][i], i < n: queued when deferral is on, launched otherwise (returns false)
inline bool fold_defer(const float* part, float* out, int chunks, int n, int N, int ldc, float scale, int accumulate, hipStream_t s) {
  mkws_train_ctx& c = ctx();
  if (!c.defer) return false;
  if (c.fb.n == kMaxFolds) { fold_flush(s); return false; }          // (the caller's partials sit at the old bump position: fold them now)
  FoldDesc& d = c.fb.d[c.fb.n++];
  d.part = part; d.out = out; d.chunks = chunks; d.n = n; d.N = N; d.ldc = ldc; d.scale = scale; d.accumulate = accumulate; d.block0 = c.fold_blocks;
  d.S = fold_subs(n, chunks);
  c.fold_blocks += fold_grid(n, d.S);
  c.bump += ((size_t)chunks * n + 63) & ~(size_t)63;
  return true;
}
inline void launch_fold_partials(const float* part, int chunks, int n, float* out, float scale, int accumulate, hipStream_t s) {
  const int S = fold_subs(n, chunks);
  hipLaunchKernelGGL(fold_partials_kernel, dim3(fold_grid(n, S)), dim3(256), 0, s, part, chunks, n, out, scale, accumulate, S);
}
inline void launch_gemm_reduce(const float* part, int ksplit, float* C, int M, int N, int ldc, int accumulate, const float* bias, int act, float* Act, hipStream_t s) {
  const int S = fold_subs((long)M * N, ksplit);
  hipLaunchKernelGGL(gemm_reduce_kernel, dim3(fold_grid((long)M * N, S)), dim3(256), 0, s, part, ksplit, C, M, N, ldc, accumulate, bias, act, Act, S);
}
constexpr int kBnMaxChunks = 256;          // chunk statistics one BatchNorm launch folds per channel (a producer with more chunks keeps the separate statistics launch)
constexpr int kBnMaxGemmTiles = 160;       // the same for GEMM row tiles (64 rows each: more, smaller chunks than the statistics kernel would make)
const bool g_bn_small = [] { const char* e = getenv("MKWS_TRAIN_BN_SMALL"); return !(e && e[0] == '0'); }();      // A/B switch of bn_small_bwd_kernel
inline int row_chunks(int M, int cap) { int c = (M + 127) / 128; if (c > cap) c = cap; if (c < 1) c = 1; return c; }
}  // namespace

extern "C" {

int mkws_op_set_scratch(float* d_scratch, size_t floats) {
  g_ctx = nullptr;                                                   // the thread's default context, and binds it
  mkws_train_ctx& c = g_default_ctx;
  if (d_scratch != c.scratch || floats != c.floats) { c.fb.n = 0; c.fold_blocks = 0; c.bump = 0; }      // a new arena: nothing is queued in it
  c.scratch = d_scratch;
  c.floats = d_scratch ? floats : 0;
  return MKWS_OK;
}

int mkws_train_ctx_create(float* d_scratch, size_t floats, mkws_train_ctx** out) {
  if (!out) return fail(MKWS_ERR_INVALID_ARG, "train_ctx_create: out is NULL");
  *out = nullptr;
  if (!d_scratch || floats == 0) return fail(MKWS_ERR_INVALID_ARG, "train_ctx_create: a context owns a scratch arena (device memory, > 0 floats)");
  mkws_train_ctx* c = new (std::nothrow) mkws_train_ctx();
  if (!c) return fail(MKWS_ERR_ALLOC, "train_ctx_create: out of host memory");
  c->scratch = d_scratch; c->floats = floats;
  *out = c;
  return MKWS_OK;
}

void mkws_train_ctx_destroy(mkws_train_ctx* c) {
  if (!c) return;
  if (g_ctx == c) g_ctx = nullptr;                                   // (other threads must not have it bound: like any handle)
  for (hipEvent_t e : c->ev)
    if (e) (void)hipEventDestroy(e);
  delete c;
}

int mkws_train_ctx_bind(mkws_train_ctx* c) {
  g_ctx = c;                                                         // NULL = back to the thread's default context
  return MKWS_OK;
}

int mkws_op_stream_wait(void* waiting_stream, void* signalling_stream) {
  // everything queued on signalling_stream so far happens before whatever is queued on waiting_stream from here on (an event record + a
  // stream wait; a wait refers to the record that preceded it, so the events are reused round-robin; capturable: fork / join of a hipGraph)
  mkws_train_ctx& c = ctx();
  hipEvent_t& e = c.ev[c.evpos];
  c.evpos = (c.evpos + 1) % 8;
  if (!e) MKWS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  MKWS_HIP(hipEventRecord(e, static_cast<hipStream_t>(signalling_stream)));
  MKWS_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(waiting_stream), e, 0));
  return MKWS_OK;
}

int mkws_op_fold_defer(int enable, void* stream) {
  mkws_train_ctx& c = ctx();
  if (enable) { c.fb.n = 0; c.fold_blocks = 0; c.bump = 0; }      // a new pass: whatever an aborted one left queued is dropped, not folded
  else fold_flush(static_cast<hipStream_t>(stream));
  c.defer = enable ? 1 : 0;
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_fold_flush(void* stream) {
  fold_flush(static_cast<hipStream_t>(stream));
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

// Which GEMM kernels mkws_op_gemm / dense_fwd / conv_bn_fwd launch (mkws_op_set_option; the environment gives the initial values):
//   "gemm_ring"    (MKWS_TRAIN_GEMM2, default 1): NN / NT on train_gemm2_kernel; 0 = everything on the LDS-staged kernel
//   "gemm_ring_tn" (MKWS_TRAIN_GEMM_TN2, default 0): weight gradients on train_gemm_tn2_kernel: 0 none, 1 = outputs of at most 256 tiles over
//                  at least 4096 rows (the big-image layers), 2 = all.  Faster on every shape of the step in isolation and when the step runs on
//                  one stream (1.14 -> 0.65 ms per 512-clip step), but it is off by default: in the trainer's two-stream step, where the weight
//                  gradients run NEXT TO the input-gradient chain, the step is bound by what the two streams move together, and the faster,
//                  greedier launch costs the other stream what it gains (7.39 -> 7.48 ms at 512 clips; profiles/r05_notes.md)
static int g_gemm2 = [] { const char* e = getenv("MKWS_TRAIN_GEMM2"); return (e && e[0] == '0') ? 0 : 1; }();

static int g_tn2 = [] { const char* e = getenv("MKWS_TRAIN_GEMM_TN2"); return e ? atoi(e) : 0; }();
static const int g_tn2_wgs = [] { const char* e = getenv("MKWS_TRAIN_TN2_WGS"); const int v = e ? atoi(e) : 1024; return v > 0 ? v : 1024; }();   // workgroups a TN launch aims for
static bool tn2_wanted(int M, int N, int K) {
  if (g_tn2 == 0) return false;
  if (g_tn2 == 2) return true;
  return (size_t)((M + 15) / 16) * ((N + 15) / 16) <= 256 && K >= 4096;
}

static int gemm_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA, int transB, int accumulate, int ksplit,
                     const float* bias, int act, float* Act, hipStream_t s, float* stats = nullptr, bool* stats_done = nullptr) {
  MKWS_REQ(A && B && C, "gemm: NULL operand");
  MKWS_REQ(M > 0 && N > 0 && K > 0 && ksplit >= 0, "gemm: bad dimensions");
  const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
  if (ksplit == 0) {
    // auto: split a long reduction of a small grid until the launch has ~256 workgroups, at least 64 columns of K per slice, as far
    // as the scratch arena reaches.  A function of the shapes and the arena size only: reproducible.
    ksplit = 1;
    {
      // Measured rules (round 4, tools/gemm_ksplit.py; times include the fold launch):
      int few = 1, many = 1;
      if (tiles <= 128 && K >= 192) {                    // a handful of tiles (weight gradients, 64-clip layers): one workgroup per CU, slices of
        few = (256 + tiles - 1) / tiles;                 // >= 64 columns of K (4b project at 64 clips, K = 480: 23.7 -> 10.4 us; K = 240: 13.5 -> 9.4)
        if (few > K / 64) few = K / 64;
      }
      if (tiles < 1024 && K >= 384) {                    // up to four workgroups per CU, at most 8 slices of >= 96 columns: a 512-row dense layer
        many = (1024 + tiles - 1) / tiles;               // is 256 tiles = ONE wave per SIMD walking 128 K steps alone (dense_1 97 -> 69 us with 4
        if (many > 8) many = 8;                          // slices, 6b project 30 -> 22 with 8, 4b project at 512 clips 25.4 -> 21 with 5; K = 240
        if (many > K / 96) many = K / 96;                // loses: 13.8 -> 14.5)
      }
      ksplit = few > many ? few : many;
      while (ksplit > 1 && !scratch((size_t)ksplit * M * N)) --ksplit;
      if (ksplit < 1) ksplit = 1;
    }
  }
  float* part = nullptr;
  if (ksplit > 1) {
    part = scratch_at((size_t)ksplit * M * N, s);
    MKWS_REQ(part, "gemm: ksplit = %d needs %zu floats of scratch (mkws_op_set_scratch)", ksplit, (size_t)ksplit * M * N);
  }
  const dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
  // chunk statistics ride in the epilogue of an unsplit NN GEMM whose row tiles are few enough for the BatchNorm launch to fold
  float* st = (stats && ksplit == 1 && !transA && !transB && !accumulate && (M + 63) / 64 <= kBnMaxGemmTiles) ? stats : nullptr;
  if (stats_done) *stats_done = st != nullptr;
  // NN / NT on the register-ring kernel (train_gemm2_kernel) whenever the operands allow float4 accesses
  // (activations and scratch must allow float4 accesses; weights -- B of NN / NT, C of TN, the bias -- are views at blob offsets and need not)
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool aligned = al16(A) && al16(part) && al16(Act) && al16(st) && (transA ? al16(B) : al16(C));
  const int bvec = (reinterpret_cast<uintptr_t>(B) & 15) == 0 ? 4 : ((reinterpret_cast<uintptr_t>(B) & 7) == 0 ? 2 : 1);
  const size_t spanA = ((size_t)(M - 1) * lda + K) * sizeof(float), spanB = (transB ? (size_t)(N - 1) * ldb + K : (size_t)(K - 1) * ldb + N) * sizeof(float);
  if (g_gemm2 && !transA && aligned && ((K | N | lda | ldb | ldc) & 3) == 0 && spanA < 0xFFFFFF00ull && spanB < 0xFFFFFF00ull) {
    const int T = (N + 15) / 16;
    int nt = T <= 4 ? T : 4;                          // n-tiles per wave: the fewest idle tile slots of 4 / 3 / 2, the larger on ties
    if (T > 4) {
      int best = 1 << 30;
      for (int cand = 4; cand >= 2; --cand) {
        const int waste = (T + cand - 1) / cand * cand - T;
        if (waste < best) { best = waste; nt = cand; }
      }
    }
    const int nblk = (T + nt - 1) / nt;
    // two row tiles per wave (every B fragment feeds both) while the launch still has ~2 workgroups per CU; BatchNorm statistics need 64-row workgroups
    const int mt = (!st && (size_t)((M + 127) / 128) * nblk * ksplit >= 512) ? 2 : 1;
    const dim3 grid2(nblk, (M + 64 * mt - 1) / (64 * mt), ksplit);
#define MKWS_TG2(TB_, MT_, NT_, BV_) hipLaunchKernelGGL((train_gemm2_kernel<TB_, MT_, NT_, BV_>), grid2, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, accumulate, ksplit, \
                                                        part, bias, act, ksplit > 1 ? nullptr : Act, st)
#define MKWS_TG2_NT(TB_, MT_, BV_) do { switch (nt) { case 1: MKWS_TG2(TB_, MT_, 1, BV_); break; case 2: MKWS_TG2(TB_, MT_, 2, BV_); break; \
                                                      case 3: MKWS_TG2(TB_, MT_, 3, BV_); break; default: MKWS_TG2(TB_, MT_, 4, BV_); break; } } while (0)
#define MKWS_TG2_BV(MT_) do { if (bvec == 4) MKWS_TG2_NT(true, MT_, 4); else if (bvec == 2) MKWS_TG2_NT(true, MT_, 2); else MKWS_TG2_NT(true, MT_, 1); } while (0)
    if (transB) { if (mt == 2) MKWS_TG2_BV(2); else MKWS_TG2_BV(1); }
    else { if (mt == 2) MKWS_TG2_NT(false, 2, 4); else MKWS_TG2_NT(false, 1, 4); }
#undef MKWS_TG2_BV
#undef MKWS_TG2_NT
#undef MKWS_TG2
    if (ksplit > 1) launch_gemm_reduce(part, ksplit, C, M, N, ldc, accumulate, bias, act, Act, s);
    MKWS_HIP(hipGetLastError());
    return MKWS_OK;
  }
  // TN (weight gradients) on the register-ring kernel: small outputs over long reductions (the tall shapes of the big-image layers)
  if (g_gemm2 && transA && !transB && aligned && !Act && ((M | N | lda | ldb | ldc) & 3) == 0 && tn2_wanted(M, N, K) &&
      ((size_t)(K - 1) * lda + M) * sizeof(float) < 0xFFFFFF00ull && ((size_t)(K - 1) * ldb + N) * sizeof(float) < 0xFFFFFF00ull) {
    const int TK = (M + 15) / 16, TN = (N + 15) / 16;
    const int kt = TK >= 2 ? 2 : 1;
    int nt = TN <= 4 ? TN : 4;
    if (TN > 4) {
      int best = 1 << 30;
      for (int cand = 4; cand >= 2; --cand) {
        const int waste = (TN + cand - 1) / cand * cand - TN;
        if (waste < best) { best = waste; nt = cand; }
      }
    }
    const int blocks = ((TK + kt - 1) / kt) * ((TN + nt - 1) / nt);
    // slices of the rows: ~4 workgroups per CU, at least 1024 rows (16 chunks per wave) each, as far as the scratch arena reaches
    int slices = (g_tn2_wgs + blocks - 1) / blocks;
    const int RC = (K + 15) / 16;
    if (slices > (RC + 63) / 64) slices = (RC + 63) / 64;
    if (slices < 1) slices = 1;
    while (slices > 1 && !scratch((size_t)slices * M * N)) --slices;
    float* part2 = nullptr;
    if (slices > 1) {
      part2 = scratch_at((size_t)slices * M * N, s);
      MKWS_REQ(part2, "gemm: %d row slices need %zu floats of scratch (mkws_op_set_scratch)", slices, (size_t)slices * M * N);
    }
    const dim3 grid3((TN + nt - 1) / nt, (TK + kt - 1) / kt, slices);
    const size_t lds = (size_t)3 * kt * nt * 256 * sizeof(float);
#define MKWS_TN2(KT_, NT_) hipLaunchKernelGGL((train_gemm_tn2_kernel<KT_, NT_>), grid3, dim3(256), lds, s, A, B, C, M, N, K, lda, ldb, ldc, accumulate, slices, part2)
#define MKWS_TN2_NT(KT_) do { switch (nt) { case 1: MKWS_TN2(KT_, 1); break; case 2: MKWS_TN2(KT_, 2); break; case 3: MKWS_TN2(KT_, 3); break; default: MKWS_TN2(KT_, 4); break; } } while (0)
    if (kt == 2) MKWS_TN2_NT(2); else MKWS_TN2_NT(1);
#undef MKWS_TN2_NT
#undef MKWS_TN2
    if (slices > 1 && !fold_defer(part2, C, slices, M * N, N, ldc, 1.0f, accumulate, s))
      launch_gemm_reduce(part2, slices, C, M, N, ldc, accumulate, nullptr, 0, nullptr, s);
    MKWS_HIP(hipGetLastError());
    return MKWS_OK;
  }
#define MKWS_TG(TA_, TB_) hipLaunchKernelGGL((train_gemm_kernel<TA_, TB_>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, accumulate, ksplit, part, bias, act, \
                                             ksplit > 1 ? nullptr : Act, st)
  if (!transA && !transB) MKWS_TG(false, false);
  else if (!transA) MKWS_TG(false, true);
  else if (!transB) MKWS_TG(true, false);
  else MKWS_TG(true, true);
#undef MKWS_TG
  // the fold of a weight-gradient GEMM (transA, no epilogue) may wait for the batch launch: nobody reads dW before the optimizer / all-reduce
  if (ksplit > 1 && !(transA && !Act && fold_defer(part, C, ksplit, M * N, N, ldc, 1.0f, accumulate, s)))
    launch_gemm_reduce(part, ksplit, C, M, N, ldc, accumulate, bias, act, Act, s);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_set_option(const char* name, int value) {
  MKWS_REQ(name, "op_set_option: name is NULL");
  if (strcmp(name, "gemm_ring") == 0) { g_gemm2 = value != 0; return MKWS_OK; }
  if (strcmp(name, "gemm_ring_tn") == 0) { MKWS_REQ(value >= 0 && value <= 2, "gemm_ring_tn: 0, 1 or 2"); g_tn2 = value; return MKWS_OK; }
  return fail(MKWS_ERR_INVALID_ARG, "unknown training-operator option '%s'", name);
}

int mkws_op_get_option(const char* name) {
  MKWS_REQ(name, "op_get_option: name is NULL");
  if (strcmp(name, "gemm_ring") == 0) return g_gemm2;
  if (strcmp(name, "gemm_ring_tn") == 0) return g_tn2;
  return fail(MKWS_ERR_INVALID_ARG, "unknown training-operator option '%s'", name);
}

int mkws_op_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA, int transB, int accumulate, int ksplit,
                 void* stream) {
  return gemm_impl(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, accumulate, ksplit, nullptr, 0, nullptr, static_cast<hipStream_t>(stream));
}

int mkws_op_dense_fwd(const float* X, const float* W, const float* bias, int act, float* Z, float* A, int M, int N, int K, void* stream) {
  MKWS_REQ(X && W && bias && Z && A, "dense_fwd: NULL operand");
  return gemm_impl(X, W, Z, M, N, K, K, N, N, 0, 0, 0, 0, bias, act, A, static_cast<hipStream_t>(stream));
}

static int bn_stats_impl(const float* Z, int M, int C, float* mean, float* var, float* mmean, float* mvar, float momentum, hipStream_t s) {
  MKWS_REQ(C % 4 == 0, "bn_stats: C must be a multiple of 4");
  const int chunks = row_chunks(M, 128);
  float* part = scratch_at((size_t)chunks * 2 * C, s);
  MKWS_REQ(part, "bn_stats: needs %zu floats of scratch (mkws_op_set_scratch)", (size_t)chunks * 2 * C);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, s, Z, part, M, C);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, s, part, chunks, M, C, mean, var, mmean, mvar, momentum);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_bn_stats(const float* Z, int M, int C, float* mean, float* var, void* stream) {
  MKWS_REQ(Z && mean && var && M > 0 && C > 0, "bn_stats: bad arguments");
  return bn_stats_impl(Z, M, C, mean, var, nullptr, nullptr, 0.0f, static_cast<hipStream_t>(stream));
}

int mkws_op_bn_train_fwd(const float* Z, int M, int C, const float* gamma, const float* beta, float eps, int act, float momentum, float* moving_mean,
                         float* moving_var, float* mean, float* var, float* A, void* stream) {
  return mkws_op_bn_train_fwd_res(Z, M, C, gamma, beta, eps, act, momentum, moving_mean, moving_var, mean, var, A, nullptr, nullptr, 1, stream);
}

int mkws_op_bn_train_fwd_res(const float* Z, int M, int C, const float* gamma, const float* beta, float eps, int act, float momentum, float* moving_mean,
                             float* moving_var, float* mean, float* var, float* A, const float* res, const float* row_scale, int group, void* stream) {
  MKWS_REQ(Z && gamma && beta && moving_mean && moving_var && mean && var && A && M > 0 && C > 0, "bn_train_fwd: bad arguments");
  MKWS_REQ(group > 0 && (res || !row_scale), "bn_train_fwd_res: row_scale needs a residual input and a positive group");
  MKWS_REQ(C % 4 == 0, "bn_train_fwd: C must be a multiple of 4");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int chunks = row_chunks(M, 128);
  float* part = scratch_at((size_t)chunks * 2 * C, s);
  MKWS_REQ(part, "bn_train_fwd: needs %zu floats of scratch (mkws_op_set_scratch)", (size_t)chunks * 2 * C);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, s, Z, part, M, C);
  hipLaunchKernelGGL(bn_train_fwd_kernel, dim3((C + 63) / 64, row_chunks(M, 256)), dim3(256), 0, s, Z, part, chunks, gamma, beta, eps, act, momentum, moving_mean,
                     moving_var, mean, var, A, M, C, res, row_scale, group, 0);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

// 1x1 convolution (NN GEMM) + training-mode BatchNorm (+ activation, + residual branch) as ONE operator: when the GEMM is unsplit and has
// at most kBnMaxGemmTiles row tiles, its epilogue leaves the BatchNorm chunk statistics (one chunk = one 64-row tile) and the BatchNorm is a
// single launch; otherwise the three-launch sequence of mkws_op_gemm + mkws_op_bn_train_fwd_res.
int mkws_op_conv_bn_fwd(const float* X, const float* W, float* Z, int M, int N, int K, const float* gamma, const float* beta, float eps, int act, float momentum,
                        float* moving_mean, float* moving_var, float* mean, float* var, float* A, const float* res, const float* row_scale, int group,
                        void* stream) {
  MKWS_REQ(X && W && Z && gamma && beta && moving_mean && moving_var && mean && var && A && M > 0 && N > 0 && K > 0, "conv_bn_fwd: bad arguments");
  MKWS_REQ(group > 0 && (res || !row_scale), "conv_bn_fwd: row_scale needs a residual input and a positive group");
  MKWS_REQ(N % 4 == 0, "conv_bn_fwd: N must be a multiple of 4");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles = (M + 63) / 64;
  float* part = (tiles <= kBnMaxGemmTiles) ? scratch_at((size_t)tiles * 2 * N, s) : nullptr;
  bool fused = false;
  if (int rc = gemm_impl(X, W, Z, M, N, K, K, N, N, 0, 0, 0, 0, nullptr, 0, nullptr, s, part, &fused)) return rc;
  if (!fused) return mkws_op_bn_train_fwd_res(Z, M, N, gamma, beta, eps, act, momentum, moving_mean, moving_var, mean, var, A, res, row_scale, group, stream);
  hipLaunchKernelGGL(bn_train_fwd_kernel, dim3((N + 63) / 64, row_chunks(M, 256)), dim3(256), 0, s, Z, part, tiles, gamma, beta, eps, act, momentum, moving_mean,
                     moving_var, mean, var, A, M, N, res, row_scale, group, 64);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_bn_act_fwd(const float* Z, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act, float* A, int M, int C,
                       void* stream) {
  MKWS_REQ(Z && mean && var && gamma && beta && A && M > 0 && C > 0, "bn_act_fwd: bad arguments");
  const size_t total = (size_t)M * C;
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), Z, mean, var, gamma, beta, eps, act, A, total, C);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_bn_act_bwd(const float* Z, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act, float* dA, float* dgamma,
                       float* dbeta, float* scratch, int M, int C, void* stream) {
  MKWS_REQ(scratch, "bn_act_bwd: bad arguments");       // (the 2*C-float argument of round 2's atomics path; kept in the signature, unused)
  return mkws_op_bn_act_bwd_ex(Z, mean, var, gamma, beta, eps, act, dA, dA, nullptr, nullptr, 0.0f, 1, dgamma, dbeta, M, C, stream);
}

int mkws_op_bn_act_bwd_ex(const float* Z, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act, float* dA,
                          const float* src, const float* row_scale, const float* bcast, float bscale, int group, float* dgamma, float* dbeta, int M, int C,
                          void* stream) {
  MKWS_REQ(group > 0 && (src || bcast), "bn_act_bwd_ex: needs an incoming gradient (src and / or bcast) and a positive group");
  MKWS_REQ(Z && mean && var && gamma && beta && dA && dgamma && dbeta && M > 0 && C > 0, "bn_act_bwd: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  MKWS_REQ(C % 4 == 0, "bn_act_bwd: C must be a multiple of 4");
  if (M <= kBnSmallRows && g_bn_small) {                            // few rows: both steps in one launch, a workgroup per 8 channels
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3((C + 7) / 8), dim3(256), 0, s, Z, mean, var, gamma, beta, eps, act, dA, M, C, src, row_scale, bcast, bscale, group,
                       dgamma, dbeta);
    MKWS_HIP(hipGetLastError());
    return MKWS_OK;
  }
  const int chunks = row_chunks(M, 128);
  float* part = scratch_at((size_t)chunks * 2 * C, s);
  MKWS_REQ(part, "bn_act_bwd: needs %zu floats of scratch (mkws_op_set_scratch)", (size_t)chunks * 2 * C);
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, s, Z, mean, var, gamma, beta, eps, act, dA, part, M, C, src, row_scale, bcast,
                     bscale, group);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((C + 63) / 64, row_chunks(M, 256)), dim3(256), 0, s, Z, mean, var, gamma, eps, dA, part, chunks, dgamma, dbeta, M, C);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_bn_update_moving(float* moving_mean, float* moving_var, const float* mean, const float* var, float momentum, int M, int C, void* stream) {
  MKWS_REQ(moving_mean && moving_var && mean && var && M > 0 && C > 0, "bn_update_moving: bad arguments");
  const float bessel = M > 1 ? (float)M / (float)(M - 1) : 1.0f;
  hipLaunchKernelGGL(bn_moving_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), moving_mean, moving_var, mean, var, momentum, bessel, C);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_dwconv_fwd(const float* X, const float* W, float* Z, int B, int H, int Wd, int C, int k, int s, int pt, int pl, int Ho, int Wo, void* stream) {
  MKWS_REQ(X && W && Z && B > 0 && C % 4 == 0 && (k == 3 || k == 5) && (s == 1 || s == 2), "dwconv_fwd: bad arguments");
  hipLaunchKernelGGL(dw_fwd_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), X, W, Z, B, H, Wd, C, k, s, pt, pl, Ho, Wo);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

// Depthwise convolution + training-mode BatchNorm + activation as ONE operator: up to kBnMaxChunks chunks of 128 output rows the convolution
// launch leaves the chunk statistics itself (two launches in all), above that the three-launch sequence.
int mkws_op_dwconv_bn_fwd(const float* X, const float* W, float* Z, int B, int H, int Wd, int C, int k, int s, int pt, int pl, int Ho, int Wo, const float* gamma,
                          const float* beta, float eps, int act, float momentum, float* moving_mean, float* moving_var, float* mean, float* var, float* A,
                          void* stream) {
  MKWS_REQ(X && W && Z && gamma && beta && moving_mean && moving_var && mean && var && A, "dwconv_bn_fwd: NULL operand");
  MKWS_REQ(B > 0 && C % 4 == 0 && (k == 3 || k == 5) && (s == 1 || s == 2), "dwconv_bn_fwd: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int M = B * Ho * Wo, chunks = (M + 127) / 128;
  if (chunks > kBnMaxChunks) {
    if (int rc = mkws_op_dwconv_fwd(X, W, Z, B, H, Wd, C, k, s, pt, pl, Ho, Wo, stream)) return rc;
    return mkws_op_bn_train_fwd_res(Z, M, C, gamma, beta, eps, act, momentum, moving_mean, moving_var, mean, var, A, nullptr, nullptr, 1, stream);
  }
  float* part = scratch_at((size_t)chunks * 2 * C, st);
  MKWS_REQ(part, "dwconv_bn_fwd: needs %zu floats of scratch (mkws_op_set_scratch)", (size_t)chunks * 2 * C);
  if (k == 3) hipLaunchKernelGGL((dw_fwd_stats_kernel<3>), dim3((C + 63) / 64, chunks), dim3(256), 0, st, X, W, Z, part, B, H, Wd, C, s, pt, pl, Ho, Wo);
  else hipLaunchKernelGGL((dw_fwd_stats_kernel<5>), dim3((C + 63) / 64, chunks), dim3(256), 0, st, X, W, Z, part, B, H, Wd, C, s, pt, pl, Ho, Wo);
  hipLaunchKernelGGL(bn_train_fwd_kernel, dim3((C + 63) / 64, row_chunks(M, 256)), dim3(256), 0, st, Z, part, chunks, gamma, beta, eps, act, momentum, moving_mean,
                     moving_var, mean, var, A, M, C, nullptr, nullptr, 1, 128);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_dwconv_bwd(const float* X, const float* W, const float* dZ, float* dX, float* dW, int B, int H, int Wd, int C, int k, int s, int pt, int pl, int Ho,
                       int Wo, void* stream) {
  MKWS_REQ(X && W && dZ && (dW || dX) && B > 0 && C % 4 == 0 && (k == 3 || k == 5) && (s == 1 || s == 2), "dwconv_bwd: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dX) hipLaunchKernelGGL(dw_bwd_input_kernel, dim3(grid_for((size_t)B * H * Wd * (C / 4))), dim3(256), 0, st, dZ, W, dX, B, H, Wd, C, k, s, pt, pl, Ho, Wo);
  if (!dW) {                                                         // input gradient only (the weight gradient is a second call, possibly on another stream)
    MKWS_HIP(hipGetLastError());
    return MKWS_OK;
  }
  // chunks of output positions: 32 positions each (two per position lane) until the grid has ~512 workgroups, at most 256 chunks
  const int xb = (C / 4 + 15) / 16;
  int chunks = (B * Ho * Wo + 31) / 32;
  const int cap = (512 + xb - 1) / xb < 256 ? (512 + xb - 1) / xb : 256;
  if (chunks > cap) chunks = cap;
  if (chunks < 1) chunks = 1;
  float* part = scratch_at((size_t)chunks * k * k * C, st);
  MKWS_REQ(part, "dwconv_bwd: needs %zu floats of scratch (mkws_op_set_scratch)", (size_t)chunks * k * k * C);
  const dim3 grid(xb, chunks);
  if (k == 3) hipLaunchKernelGGL((dw_bwd_weight_kernel<3>), grid, dim3(256), 0, st, X, dZ, part, B, H, Wd, C, s, pt, pl, Ho, Wo);
  else hipLaunchKernelGGL((dw_bwd_weight_kernel<5>), grid, dim3(256), 0, st, X, dZ, part, B, H, Wd, C, s, pt, pl, Ho, Wo);
  if (!fold_defer(part, dW, chunks, k * k * C, k * k * C, k * k * C, 1.0f, 0, st))
    launch_fold_partials(part, chunks, k * k * C, dW, 1.0f, 0, st);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_stem_fwd(const float* spec, const float* W, float norm_mean, float norm_std, float* Z, int B, void* stream) {
  MKWS_REQ(spec && W && Z && B > 0, "stem_fwd: bad arguments");
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid_for((size_t)B * 500 * 8)), dim3(256), 0, static_cast<hipStream_t>(stream), spec, W, norm_mean, norm_std, Z, B);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_stem_bwd_weight(const float* spec, const float* dZ, float norm_mean, float norm_std, float* dW, int B, void* stream) {
  MKWS_REQ(spec && dZ && dW && B > 0, "stem_bwd_weight: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int blocks = (B * 500 + 255) / 256; if (blocks > 256) blocks = 256;
  float* part = scratch_at((size_t)blocks * 288, s);
  MKWS_REQ(part, "stem_bwd_weight: needs %zu floats of scratch (mkws_op_set_scratch)", (size_t)blocks * 288);
  hipLaunchKernelGGL(stem_bwd_weight_kernel, dim3(blocks), dim3(256), 0, s, spec, dZ, norm_mean, norm_std, part, B);
  if (!fold_defer(part, dW, blocks, 288, 288, 288, 1.0f, 0, s)) launch_fold_partials(part, blocks, 288, dW, 1.0f, 0, s);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_pool_hw(const float* A, float* mean, int B, int HW, int C, void* stream) {
  MKWS_REQ(A && mean && B > 0 && HW > 0 && C % 4 == 0, "pool_hw: bad arguments");
  hipLaunchKernelGGL(pool_hw_kernel, dim3((C / 4 + 63) / 64, B), dim3(256), 0, static_cast<hipStream_t>(stream), A, mean, B, HW, C);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_scale_channels(const float* A, const float* g, float* out, int B, int HW, int C, void* stream) {
  MKWS_REQ(A && g && out && B > 0 && HW > 0 && C > 0, "scale_channels: bad arguments");
  hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for((size_t)B * HW * C)), dim3(256), 0, static_cast<hipStream_t>(stream), A, g, out, B, HW, C);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_se_bwd(const float* A, const float* g, const float* dOut, float* dA, float* dg, int B, int HW, int C, void* stream) {
  MKWS_REQ(A && g && dOut && dA && dg && B > 0 && HW > 0 && C % 4 == 0, "se_bwd: bad arguments");
  hipLaunchKernelGGL(se_bwd_kernel, dim3((C / 4 + 63) / 64, B), dim3(256), 0, static_cast<hipStream_t>(stream), A, g, dOut, dA, dg, B, HW, C);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

// the squeeze-excite parameter gradients: batch rows in chunks of 64 over blockIdx.z; with more than one chunk the kernel writes partial
// sums to the scratch arena and four fixed-order folds (deferred with the others when deferral is on) produce the gradients
static int se_wgrad_launch(const float* mean, const float* R, const float* dYg, const float* dYr, float* dWr, float* dbr, float* dWe, float* dbe, int B, int C, int se,
                           hipStream_t s) {
  const int chunks = (B + 63) / 64;
  const dim3 grid((C + 63) / 64, (se + 7) / 8, chunks);
  if (chunks == 1) {
    hipLaunchKernelGGL(se_wgrad_kernel, grid, dim3(256), 0, s, mean, R, dYg, dYr, dWr, dbr, dWe, dbe, B, C, se, 64);
    return MKWS_OK;
  }
  auto r64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t nW = (size_t)se * C, sW = r64(chunks * nW), sbe = r64((size_t)chunks * C), sbr = r64((size_t)chunks * se);
  mkws_train_ctx& c = ctx();
  if (c.defer && c.fb.n + 4 > kMaxFolds) fold_flush(s);              // all four folds join the queue together (or none)
  float* base = scratch_at(2 * sW + sbe + sbr, s);
  MKWS_REQ(base, "se_wgrad: needs %zu floats of scratch (mkws_op_set_scratch)", 2 * sW + sbe + sbr);
  float* pWr = base; float* pWe = pWr + sW; float* pbe = pWe + sW; float* pbr = pbe + sbe;
  hipLaunchKernelGGL(se_wgrad_kernel, grid, dim3(256), 0, s, mean, R, dYg, dYr, pWr, pbr, pWe, pbe, B, C, se, 64);
  struct { float* part; float* out; int n; } f[4] = {{pWr, dWr, (int)nW}, {pWe, dWe, (int)nW}, {pbe, dbe, C}, {pbr, dbr, se}};
  for (auto& d : f)
    if (!fold_defer(d.part, d.out, chunks, d.n, d.n, d.n, 1.0f, 0, s))
      launch_fold_partials(d.part, chunks, d.n, d.out, 1.0f, 0, s);
  return MKWS_OK;
}

int mkws_op_se_fwd(const float* A, const float* Wr, const float* br, const float* We, const float* be, float* mean, float* Yr, float* R, float* G, float* out,
                   float* work, int B, int HW, int C, int se, void* stream) {
  MKWS_REQ(A && Wr && br && We && be && mean && Yr && R && G && out && work, "se_fwd: NULL operand");
  MKWS_REQ(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= kSeMaxC && se > 0 && se <= kSeMaxSe, "se_fwd: needs C %% 4 == 0, C <= %d, se <= %d (got C = %d, se = %d)",
           kSeMaxC, kSeMaxSe, C, se);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((C + kSeSlab - 1) / kSeSlab, B);
  hipLaunchKernelGGL(se_fwd_pool_kernel, grid, dim3(256), 0, s, A, Wr, mean, work, HW, C, se);
  hipLaunchKernelGGL(se_fwd_gate_kernel, grid, dim3(256), 0, s, A, work, br, We, be, Yr, R, G, out, HW, C, se);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_se_bwd_fused(const float* A, const float* G, const float* dOut, const float* mean, const float* Yr, const float* R, const float* Wr, const float* We,
                         float* dA, float* dmean, float* dYg, float* dYr, float* dWr, float* dbr, float* dWe, float* dbe, float* work, int B, int HW, int C, int se,
                         void* stream) {
  MKWS_REQ(A && G && dOut && mean && Yr && R && Wr && We && dA && dmean && dYg && dYr && work, "se_bwd_fused: NULL operand");
  MKWS_REQ((dWr && dbr && dWe && dbe) || (!dWr && !dbr && !dWe && !dbe), "se_bwd_fused: the four parameter gradients come together (all or none)");
  MKWS_REQ(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= kSeMaxC && se > 0 && se <= kSeMaxSe, "se_bwd_fused: needs C %% 4 == 0, C <= %d, se <= %d (got C = %d, se = %d)",
           kSeMaxC, kSeMaxSe, C, se);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((C + kSeSlab - 1) / kSeSlab, B);
  hipLaunchKernelGGL(se_bwd_gate_kernel, grid, dim3(256), 0, s, A, G, dOut, We, dA, dYg, work, HW, C, se);
  hipLaunchKernelGGL(se_bwd_squeeze_kernel, grid, dim3(128), 0, s, work, Yr, Wr, dYr, dmean, C, se);
  if (dWr)
    if (int rc = se_wgrad_launch(mean, R, dYg, dYr, dWr, dbr, dWe, dbe, B, C, se, s)) return rc;
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_se_wgrad(const float* mean, const float* R, const float* dYg, const float* dYr, float* dWr, float* dbr, float* dWe, float* dbe, int B, int C, int se,
                     void* stream) {
  MKWS_REQ(mean && R && dYg && dYr && dWr && dbr && dWe && dbe, "se_wgrad: NULL operand");
  MKWS_REQ(B > 0 && C > 0 && se > 0 && se <= kSeMaxSe, "se_wgrad: bad dimensions");
  if (int rc = se_wgrad_launch(mean, R, dYg, dYr, dWr, dbr, dWe, dbe, B, C, se, static_cast<hipStream_t>(stream))) return rc;
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_add_bcast(float* X, const float* v, float scale, int B, int HW, int C, void* stream) {
  MKWS_REQ(X && v && B > 0 && HW > 0 && C > 0, "add_bcast: bad arguments");
  hipLaunchKernelGGL(add_bcast_kernel, dim3(grid_for((size_t)B * HW * C)), dim3(256), 0, static_cast<hipStream_t>(stream), X, v, scale, B, HW, C);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_bias_act_fwd(const float* Z, const float* bias, int act, float* A, int M, int N, void* stream) {
  MKWS_REQ(Z && bias && A && M > 0 && N > 0, "bias_act_fwd: bad arguments");
  const size_t total = (size_t)M * N;
  hipLaunchKernelGGL(bias_act_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), Z, bias, act, A, total, N);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_bias_act_bwd(const float* Z, const float* bias, int act, float* dA, float* dbias, int M, int N, void* stream) {
  MKWS_REQ(Z && bias && dA && dbias && M > 0 && N > 0, "bias_act_bwd: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int chunks = (M + 63) / 64; if (chunks > 64) chunks = 64; if (chunks < 1) chunks = 1;
  float* part = scratch_at((size_t)chunks * N, s);
  MKWS_REQ(part, "bias_act_bwd: needs %zu floats of scratch (mkws_op_set_scratch)", (size_t)chunks * N);
  // one pass: dA <- dA * act'(Z + bias) in place and this chunk's column sums; then the chunks fold in order into dbias
  hipLaunchKernelGGL((col_sum_partial_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, dA, Z, bias, act, part, M, N);
  if (!fold_defer(part, dbias, chunks, N, N, N, 1.0f, 0, s)) launch_fold_partials(part, chunks, N, dbias, 1.0f, 0, s);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_row_scale_add(const float* a, const float* row_scale, const float* c, float* out, int B, int64_t per_row, void* stream) {
  MKWS_REQ(a && row_scale && out && B > 0 && per_row > 0, "row_scale_add: bad arguments");
  hipLaunchKernelGGL(row_scale_add_kernel, dim3(grid_for((size_t)B * per_row)), dim3(256), 0, static_cast<hipStream_t>(stream), a, row_scale, c, out, B, (size_t)per_row);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_axpy(float* y, const float* x, float alpha, int64_t n, void* stream) {
  MKWS_REQ(y && x && n > 0, "axpy: bad arguments");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, static_cast<hipStream_t>(stream), y, x, alpha, (size_t)n);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_softmax_ce(float* logits, const int32_t* labels, int B, int N, float* rowstat, float* stats, void* stream) {
  MKWS_REQ(logits && labels && rowstat && stats && B > 0 && N > 0, "softmax_ce: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(B), dim3(256), 0, s, logits, labels, rowstat, N, 1.0f / (float)B);
  hipLaunchKernelGGL(rowstat_fold_kernel, dim3(1), dim3(64), 0, s, rowstat, B, stats);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_step_inc(int* d_step, void* stream) {
  MKWS_REQ(d_step, "step_inc: NULL counter");
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_step);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_adam_dev(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, const int* d_step,
                     float grad_scale, void* stream) {
  MKWS_REQ(params && grads && m && v && n > 0 && d_step, "adam_dev: bad arguments");
  hipLaunchKernelGGL(train_adam_dev_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, static_cast<hipStream_t>(stream), params, grads, m, v, (size_t)n, lr, beta1, beta2, eps,
                     d_step, grad_scale);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_op_adam(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step_t, float grad_scale,
                 void* stream) {
  MKWS_REQ(params && grads && m && v && n > 0 && step_t >= 1, "adam: bad arguments");
  const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, step_t)) / (1.0 - std::pow((double)beta1, step_t));
  hipLaunchKernelGGL(train_adam_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, static_cast<hipStream_t>(stream), params, grads, m, v, (size_t)n, (float)lr_t, beta1, beta2,
                     eps, grad_scale);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

}  // extern "C"
