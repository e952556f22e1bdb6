// mkws_head.hip -- the few-shot head on the frozen embedding: Dense(18,tanh) -> Dense(3,softmax),
// sparse categorical cross-entropy, backward, Keras Adam.
//
// Replaces the trainable part of multilingual_kws/embedding/transfer_learning.py:47-59 and the
// per-step work of xfer.fit (:86-93).  Semantics: SURVEY.md Appendix C.1-C.2.
//   parameter vector: W1[in,hid] | b1[hid] | W2[hid,cls] | b2[cls]   (Keras [in,out] kernels)
// The gradient lives in one flat device buffer so the data-parallel host code can all-reduce it with
// a single RCCL call (74 028 B for 1024/18/3).  All reductions are fixed-order (no atomics), so a
// step is bit-reproducible.
#include "mkws_common.h"

#include <cmath>
#include <new>

namespace mkws {

constexpr int kMaxHidden = 32;
constexpr int kMaxClasses = 8;
constexpr int kGradSplits = 32;     // row slices of the dW1 partial sums: 4 x 32 workgroups (8 slices left 7 of 8 CUs idle: 64 us per step)

struct HeadDims { int in, hid, cls; };

// one wave per row: h = tanh(x W1 + b1); z = h W2 + b2; p = softmax(z).
// TRAIN: also per-row loss, correctness, dz = (p - onehot)/B and dpre = (dz W2^T) * (1 - h^2).
// Parameter blocks of up to kHeadsPerLaunch heads, passed by value (multi-keyword serving: blockIdx.y = head).
constexpr int kHeadsPerLaunch = 64;
struct HeadTable { const float* p[kHeadsPerLaunch]; };

// Inference forward on the fp32 matrix cores (mkws_head_forward and the multi-keyword mkws_heads_forward).
// The per-row kernel below spends its time on latency: 50 heads x 256 windows took 94 us for 0.47 GFLOP, one wave per SIMD with
// ~300 live registers.  Here a wave owns 16 rows x (up to) 32 hidden units of ONE head: z1^T[hid, rows] = W1^T[hid, in] . x^T[in, rows]
// as NT = ceil(hid / 16) tiles of v_mfma_f32_16x16x4_f32 (A = weights, B = activations, exactly as in pw_gemm_kernel), K walked in
// ascending chunks of 16, so a row's result depends on nothing but that row and the head's parameters (bit-identical across batch
// sizes, and between the single-head and the multi-head entry point).  Lane (g, c) supplies W1[16j + 4g + s][16 nt + c] (four
// dword loads 72 B apart: the Keras [in, hid] layout is kept, the heads stay trainable in place) and x[row c][16j + 4g .. + 3]
// (one float4).  The four waves of a workgroup share ONE row tile and split K (wave w walks chunks [w * KC/4, (w+1) * KC/4)): a
// single window then waits for 16 dependent chunk steps instead of 64 (batch-1 serving: 15 -> ~5 us), and 256 windows give 800
// workgroups instead of 200; the four partial tiles meet in LDS and are added in wave order, so the split is the same for every
// batch size.  Epilogue (wave 0) in registers: tanh, the hid x cls second layer as per-lane partial sums folded across the four lane
// groups by two xor-shuffles, softmax on every lane, lane group 0 stores.  Needs in % 16 == 0 and hid <= 32.
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NT, bool MULTI>
__global__ __launch_bounds__(256) void head_fwd_mfma_kernel(HeadDims d, const float* __restrict__ params, const float* __restrict__ x, int B,
                                                            float* __restrict__ probs, HeadTable table) {
  __shared__ float s_part[3][NT][64][4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int row0 = blockIdx.x * 16;
  if (MULTI) {
    params = table.p[blockIdx.y];
    probs += (size_t)blockIdx.y * B * d.cls;
  }
  const float* W1 = params;
  const float* b1 = W1 + (size_t)d.in * d.hid;
  const float* W2 = b1 + d.hid;
  const float* b2 = W2 + (size_t)d.hid * d.cls;
  const int row = row0 + c;
  // operands as buffer descriptors + 32-bit lane offsets + the K-chunk offset in an SGPR: the steady state issues no VALU address
  // arithmetic (see pw_gemm_kernel).  Columns past hid re-read column 0: their accumulator rows are never looked at.
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W1), 0, 0x7fffffff, 0x00020000);
  const unsigned xoff = (unsigned)(((size_t)(row < B ? row : B - 1) * d.in + 4 * g) * sizeof(float));
  unsigned woff[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = (16 * nt + c < d.hid) ? 16 * nt + c : 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) woff[nt][s] = (unsigned)(((4 * g + s) * d.hid + col) * sizeof(float));
  }
  const unsigned wchunk = (unsigned)(16 * d.hid * sizeof(float));
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int KCall = d.in / 16;
  const int kper = (KCall + 3) / 4;                      // K chunks per wave
  const int kc0 = wave * kper;
  const int KC = (kc0 + kper <= KCall) ? kper : (KCall > kc0 ? KCall - kc0 : 0);
  constexpr int D = 4;                           // chunks in flight per wave
  f32x4 xq[D];
  float wq[D][NT][4];
  auto load = [&](int j, f32x4& xv, float (&wv)[NT][4]) {
    xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, xoff, 64u * (unsigned)(kc0 + j), 0));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < 4; ++s) wv[nt][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW, woff[nt][s], wchunk * (unsigned)(kc0 + j), 0));
  };
  auto compute = [&](const f32x4& xv, const float (&wv)[NT][4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][s], xv[s], acc[nt], 0, 0, 0);
  };
  if (KC >= D) {                                 // unconditional prologue / branch-free steady state / drain (hipcc then counts vmcnt)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) load(dd, xq[dd], wq[dd]);
    int j = 0;
    for (; j + 2 * D <= KC; j += D) {
#pragma unroll
      for (int dd = 0; dd < D; ++dd) {
        compute(xq[dd], wq[dd]);
        load(j + D + dd, xq[dd], wq[dd]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      compute(xq[dd], wq[dd]);
      if (j + D + dd < KC) load(j + D + dd, xq[dd], wq[dd]);
      __builtin_amdgcn_sched_barrier(0);
    }
    j += D;
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
      if (j + dd < KC) compute(xq[dd], wq[dd]);
  } else {
    for (int j = 0; j < KC; ++j) {
      load(j, xq[0], wq[0]);
      compute(xq[0], wq[0]);
    }
  }
  // the K slices of waves 1..3 join wave 0's in wave order
  if (wave > 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_part[wave - 1][nt][lane][r] = acc[nt][r];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[nt][r] += s_part[w][nt][lane][r];
  // acc[nt][r] = pre-activation of hidden unit 16 nt + 4 g + r for row c
  float z[kMaxClasses];
#pragma unroll
  for (int k = 0; k < kMaxClasses; ++k) z[k] = 0.0f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int u = 16 * nt + 4 * g + r;
      if (u < d.hid) {
        const float h = tanhf(acc[nt][r] + b1[u]);
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k)
          if (k < d.cls) z[k] += h * W2[u * d.cls + k];
      }
    }
  float zmax = -3.0e38f;
#pragma unroll
  for (int k = 0; k < kMaxClasses; ++k) {
    if (k < d.cls) {
      z[k] += __shfl_xor(z[k], 16, 64);
      z[k] += __shfl_xor(z[k], 32, 64);
      z[k] += b2[k];
      zmax = fmaxf(zmax, z[k]);
    }
  }
  float e[kMaxClasses], esum = 0.0f;
#pragma unroll
  for (int k = 0; k < kMaxClasses; ++k) {
    e[k] = (k < d.cls) ? expf(z[k] - zmax) : 0.0f;
    esum += e[k];
  }
  const float inv = 1.0f / esum;
  if (g == 0 && row < B) {
#pragma unroll
    for (int k = 0; k < kMaxClasses; ++k)
      if (k < d.cls) probs[(size_t)row * d.cls + k] = e[k] * inv;
  }
}

// R rows per wave: every W1 value a lane loads is used for R rows (the 50-head serving launch re-read each head's 73 KB of W1 once
// per row and wave: 121 us per 256 windows).  The arithmetic of one row does not depend on R, so the results are bit-identical.
template <bool TRAIN, bool MULTI = false, int R = 1>
__global__ __launch_bounds__(256) void head_rows_kernel(HeadDims d, const float* __restrict__ params, const float* __restrict__ x,
                                                        const int32_t* __restrict__ labels, int B, float* __restrict__ probs,
                                                        float* __restrict__ hbuf /*[B,hid]*/, float* __restrict__ dz /*[B,cls]*/,
                                                        float* __restrict__ dpre /*[B,hid]*/, float* __restrict__ rowstat /*[B,2]*/,
                                                        HeadTable table = HeadTable()) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= B) return;
  if (MULTI) {                      // head blockIdx.y: its own parameters, its own [B, cls] slab of the output
    params = table.p[blockIdx.y];
    probs += (size_t)blockIdx.y * B * d.cls;
  }
  const float* W1 = params;
  const float* b1 = W1 + (size_t)d.in * d.hid;
  const float* W2 = b1 + d.hid;
  const float* b2 = W2 + (size_t)d.hid * d.cls;
  float accs[R][kMaxHidden];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < kMaxHidden; ++j) accs[r][j] = 0.0f;
  const float* xr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) xr[r] = x + (size_t)((row0 + r < B) ? row0 + r : B - 1) * d.in;
  const bool even = (d.hid & 1) == 0;              // rows of W1 are then 8-byte aligned: float2 loads
  // k steps in flight (their loads issue together; the adds keep their order).  Round 5: the training launch (R = 1, one wave per row) is
  // pure latency -- 16 k steps of 1 + 9 loads each -- so it keeps 8 of them in flight instead of 4
  constexpr int KU = (R == 1) ? (TRAIN ? 8 : 4) : 2;
#pragma unroll KU
  for (int k = lane; k < d.in; k += 64) {
    float xv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) xv[r] = xr[r][k];
    const float* w = W1 + (size_t)k * d.hid;
    float wv[kMaxHidden];
    if (even) {
      const float2* w2 = reinterpret_cast<const float2*>(w);
#pragma unroll
      for (int j = 0; j < kMaxHidden / 2; ++j) {
        const float2 t = (2 * j < d.hid) ? w2[j] : make_float2(0.0f, 0.0f);
        wv[2 * j] = t.x; wv[2 * j + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < kMaxHidden; ++j) wv[j] = (j < d.hid) ? w[j] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < kMaxHidden; ++j)
        if (j < d.hid) accs[r][j] += xv[r] * wv[j];
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < kMaxHidden; ++j) {
      if (j < d.hid) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) accs[r][j] += __shfl_xor(accs[r][j], off, 64);
      }
    }
  // every lane now holds the full sums.  The row is finished IN PARALLEL: lane j takes hidden unit j (one tanhf per lane instead
  // of hid serial ones on lane 0), the logits are butterfly sums over the lanes, every lane evaluates the softmax of the few classes.
  const bool hj = lane < d.hid;
  const float b1l = hj ? b1[lane] : 0.0f;
  float w2l[kMaxClasses];
#pragma unroll
  for (int c = 0; c < kMaxClasses; ++c) w2l[c] = (hj && c < d.cls) ? W2[lane * d.cls + c] : 0.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r;
    if (row >= B) break;
    float a = 0.0f;
#pragma unroll
    for (int j = 0; j < kMaxHidden; ++j) a = (lane == j) ? accs[r][j] : a;
    const float h = hj ? tanhf(a + b1l) : 0.0f;
    float z[kMaxClasses];
    float zmax = -3.0e38f;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) {
      if (c < d.cls) {
        float p = h * w2l[c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off, 64);
        z[c] = p + b2[c];
        zmax = fmaxf(zmax, z[c]);
      } else {
        z[c] = 0.0f;
      }
    }
    float e[kMaxClasses], esum = 0.0f;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) {
      e[c] = (c < d.cls) ? expf(z[c] - zmax) : 0.0f;
      esum += e[c];
    }
    const float inv = 1.0f / esum;
    if (probs && lane == 0) {
#pragma unroll
      for (int c = 0; c < kMaxClasses; ++c)
        if (c < d.cls) probs[(size_t)row * d.cls + c] = e[c] * inv;
    }
    if (TRAIN) {
      const int y = labels[row];
      int best = 0;
      float zy = 0.0f;
#pragma unroll
      for (int c = 0; c < kMaxClasses; ++c) {
        if (c < d.cls) {
          if (z[c] > z[best]) best = c;
          if (c == y) zy = z[c];
        }
      }
      if (lane == 0) {
        // -log softmax(z)[y], computed from the logits (what Keras does for a softmax-activated output)
        rowstat[2 * row] = (logf(esum) + zmax) - zy;
        rowstat[2 * row + 1] = (best == y) ? 1.0f : 0.0f;
      }
      const float invB = 1.0f / (float)B;
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < kMaxClasses; ++c) {
        const float dzc = (c < d.cls) ? (e[c] * inv - (c == y ? 1.0f : 0.0f)) * invB : 0.0f;
        if (c < d.cls && lane == 0) dz[(size_t)row * d.cls + c] = dzc;
        s += dzc * w2l[c];
      }
      if (hj) {
        hbuf[(size_t)row * d.hid + lane] = h;
        dpre[(size_t)row * d.hid + lane] = s * (1.0f - h * h);
      }
    }
  }
}

// partial dW1: block (kx, split) handles 256 input features x rows [split*rows_per, +rows_per)
__global__ __launch_bounds__(256) void head_dw1_partial_kernel(HeadDims d, const float* __restrict__ x, const float* __restrict__ dpre,
                                                               int B, int rows_per, float* __restrict__ partial /*[splits][in*hid]*/) {
  __shared__ float s_dp[64 * kMaxHidden];
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = (r0 + rows_per < B) ? r0 + rows_per : B;
  float acc[kMaxHidden];
#pragma unroll
  for (int j = 0; j < kMaxHidden; ++j) acc[j] = 0.0f;
  for (int rb = r0; rb < r1; rb += 64) {
    const int nr = (r1 - rb < 64) ? r1 - rb : 64;
    __syncthreads();
    for (int i = threadIdx.x; i < nr * d.hid; i += 256) s_dp[i] = dpre[(size_t)rb * d.hid + i];
    __syncthreads();
    if (k < d.in) {
#pragma unroll 8      // eight rows' loads in flight (the sums keep their row order): the plain loop paid one L2 latency per row
      for (int r = 0; r < nr; ++r) {
        const float xv = x[(size_t)(rb + r) * d.in + k];
#pragma unroll
        for (int j = 0; j < kMaxHidden; ++j)
          if (j < d.hid) acc[j] += xv * s_dp[r * d.hid + j];
      }
    }
  }
  if (k < d.in) {
    float* dst = partial + (size_t)blockIdx.y * d.in * d.hid + (size_t)k * d.hid;
#pragma unroll
    for (int j = 0; j < kMaxHidden; ++j)
      if (j < d.hid) dst[j] = acc[j];
  }
}

// final reduce: grads = [sum of partials | db1 | dW2 | db2 | sum loss, sum correct]; the two statistics ride
// behind the gradient so that a data-parallel step is ONE all-reduce of nparams + 2 floats; `stats` (optional)
// receives a copy of them
__global__ __launch_bounds__(256) void head_grad_finish_kernel(HeadDims d, const float* __restrict__ partial, int splits,
                                                               const float* __restrict__ hbuf, const float* __restrict__ dz,
                                                               const float* __restrict__ dpre, const float* __restrict__ rowstat, int B,
                                                               float* __restrict__ grads, float* __restrict__ stats) {
  const int nW1 = d.in * d.hid;
  const int nsmall = d.hid + d.hid * d.cls + d.cls;
  const int nbW1 = (nW1 + 255) / 256;
  if ((int)blockIdx.x < nbW1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nW1) {
      float s = 0.0f;
#pragma unroll 8      // (loads of eight slices in flight, added in slice order)
      for (int p = 0; p < splits; ++p) s += partial[(size_t)p * nW1 + i];
      grads[i] = s;
    }
    return;
  }
  // the nsmall + 2 sums over the batch: one WAVE each, lanes stride the rows, fixed-order butterfly (one THREAD each walked
  // the 512 rows with dependent loads: 181 us per step)
  const int t = ((int)blockIdx.x - nbW1) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= nsmall + 2) return;
  float s = 0.0f;
  if (t < d.hid) {                                    // db1[j] = sum_b dpre[b][j]
    for (int b = lane; b < B; b += 64) s += dpre[(size_t)b * d.hid + t];
  } else if (t < d.hid + d.hid * d.cls) {              // dW2[j][c] = sum_b h[b][j] dz[b][c]
    const int j = (t - d.hid) / d.cls, c = (t - d.hid) % d.cls;
    for (int b = lane; b < B; b += 64) s += hbuf[(size_t)b * d.hid + j] * dz[(size_t)b * d.cls + c];
  } else if (t < nsmall) {                            // db2[c]
    const int c = t - d.hid - d.hid * d.cls;
    for (int b = lane; b < B; b += 64) s += dz[(size_t)b * d.cls + c];
  } else {                                            // loss sum, correct count
    for (int b = lane; b < B; b += 64) s += rowstat[2 * b + (t - nsmall)];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) {
    grads[nW1 + t] = s;
    if (t >= nsmall && stats) stats[t - nsmall] = s;
  }
}

// gradient of the mean loss w.r.t. the head's input rows: dX[b,k] = sum_j dpre[b,j] * W1[k,j]   (backprop_into_embedding)
__global__ __launch_bounds__(256) void head_input_grad_kernel(HeadDims d, const float* __restrict__ params, const float* __restrict__ dpre, int B,
                                                              float* __restrict__ dX) {
  const size_t total = (size_t)B * d.in;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % d.in);
    const size_t b = i / d.in;
    const float* w = params + (size_t)k * d.hid;
    const float* dp = dpre + b * d.hid;
    float s = 0.0f;
    for (int j = 0; j < d.hid; ++j) s += dp[j] * w[j];
    dX[i] = s;
  }
}

// Keras Adam (optimizer_v2/adam.py): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps)
__global__ __launch_bounds__(256) void head_adam_kernel(float* __restrict__ params, const float* __restrict__ grads, float* __restrict__ m,
                                                        float* __restrict__ v, int n, float lr_t, float beta1, float beta2, float eps,
                                                        float grad_scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float g = grads[i] * grad_scale;
  const float mi = m[i] + (g - m[i]) * (1.0f - beta1);
  const float vi = v[i] + (g * g - v[i]) * (1.0f - beta2);
  m[i] = mi;
  v[i] = vi;
  params[i] -= (mi * lr_t) / (sqrtf(vi) + eps);
}

// The same update with the step index read from device memory (a captured hipGraph replays one launch for every step)
__global__ __launch_bounds__(256) void head_adam_dev_kernel(float* __restrict__ params, const float* __restrict__ grads, float* __restrict__ m,
                                                            float* __restrict__ v, int n, float lr, float beta1, float beta2, float eps,
                                                            const int* __restrict__ step, float grad_scale) {
  __shared__ float s_lr;
  if (threadIdx.x == 0) {
    const int t = *step;
    s_lr = (float)((double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t)));
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float lr_t = s_lr;
  const float g = grads[i] * grad_scale;
  const float mi = m[i] + (g - m[i]) * (1.0f - beta1);
  const float vi = v[i] + (g * g - v[i]) * (1.0f - beta2);
  m[i] = mi;
  v[i] = vi;
  params[i] -= (mi * lr_t) / (sqrtf(vi) + eps);
}

}  // namespace mkws

using namespace mkws;

struct mkws_head {
  HeadDims d;
  int max_batch = 0, nparams = 0;
  float* d_state = nullptr;   // params | grads | m | v
  float* d_work = nullptr;    // hbuf | dz | dpre | rowstat | partial
  float *params, *grads, *m, *v;
  float *hbuf, *dz, *dpre, *rowstat, *partial;
};

extern "C" {

int mkws_head_create(int in_dim, int hidden, int classes, int max_batch, mkws_head** out) {
  if (!out) return fail(MKWS_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (in_dim <= 0 || hidden <= 0 || classes <= 1 || max_batch <= 0) return fail(MKWS_ERR_INVALID_ARG, "head dims must be positive (classes >= 2)");
  if (hidden > kMaxHidden || classes > kMaxClasses)
    return fail(MKWS_ERR_UNSUPPORTED, "head kernels support hidden <= %d and classes <= %d", kMaxHidden, kMaxClasses);
  int rc = require_device();
  if (rc != MKWS_OK) return rc;
  mkws_head* hd = new (std::nothrow) mkws_head();
  if (!hd) return fail(MKWS_ERR_ALLOC, "out of host memory");
  hd->d = {in_dim, hidden, classes};
  hd->max_batch = max_batch;
  hd->nparams = in_dim * hidden + hidden + hidden * classes + classes;
  const size_t P = ((size_t)hd->nparams + 2 + 63) & ~size_t(63);      // + 2: the loss / accuracy sums behind the gradient
  if (hipMalloc(reinterpret_cast<void**>(&hd->d_state), 4 * P * sizeof(float)) != hipSuccess) { delete hd; return fail(MKWS_ERR_ALLOC, "hipMalloc failed"); }
  (void)hipMemset(hd->d_state, 0, 4 * P * sizeof(float));
  hd->params = hd->d_state; hd->grads = hd->d_state + P; hd->m = hd->d_state + 2 * P; hd->v = hd->d_state + 3 * P;
  const size_t mb = (size_t)max_batch;
  const size_t nwork = mb * hidden * 2 + mb * classes + mb * 2 + (size_t)kGradSplits * in_dim * hidden + 64;
  if (hipMalloc(reinterpret_cast<void**>(&hd->d_work), nwork * sizeof(float)) != hipSuccess) {
    (void)hipFree(hd->d_state); delete hd; return fail(MKWS_ERR_ALLOC, "hipMalloc failed");
  }
  float* w = hd->d_work;
  hd->hbuf = w; w += mb * hidden; hd->dpre = w; w += mb * hidden; hd->dz = w; w += mb * classes; hd->rowstat = w; w += mb * 2; hd->partial = w;
  *out = hd;
  return MKWS_OK;
}

void mkws_head_destroy(mkws_head* hd) {
  if (!hd) return;
  if (hd->d_state) (void)hipFree(hd->d_state);
  if (hd->d_work) (void)hipFree(hd->d_work);
  delete hd;
}

int mkws_head_param_count(const mkws_head* hd) { return hd ? hd->nparams : fail(MKWS_ERR_INVALID_ARG, "head handle is NULL"); }
int mkws_head_grad_count(const mkws_head* hd) { return hd ? hd->nparams + 2 : fail(MKWS_ERR_INVALID_ARG, "head handle is NULL"); }
float* mkws_head_params(mkws_head* hd) { return hd ? hd->params : nullptr; }
float* mkws_head_grads(mkws_head* hd) { return hd ? hd->grads : nullptr; }
int mkws_head_state_floats(const mkws_head* hd) { return hd ? (int)(4 * (hd->grads - hd->params)) : fail(MKWS_ERR_INVALID_ARG, "head handle is NULL"); }

int mkws_head_set_params(mkws_head* hd, const float* h_params, int n, void* stream) {
  if (!hd || !h_params) return fail(MKWS_ERR_INVALID_ARG, "NULL argument");
  if (n != hd->nparams) return fail(MKWS_ERR_INVALID_ARG, "expected %d parameters, got %d", hd->nparams, n);
  // ordered on the caller's stream like every other entry point (a null-stream copy could overtake a
  // loss_grad / adam_step still in flight on a non-blocking stream); h_params is pageable host memory, so
  // the stream is drained before returning
  hipStream_t s = static_cast<hipStream_t>(stream);
  MKWS_HIP(hipMemcpyAsync(hd->params, h_params, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
  const size_t P = ((size_t)hd->nparams + 2 + 63) & ~size_t(63);
  MKWS_HIP(hipMemsetAsync(hd->grads, 0, 3 * P * sizeof(float), s));
  MKWS_HIP(hipStreamSynchronize(s));
  return MKWS_OK;
}

int mkws_head_get_params(mkws_head* hd, float* h_params, int n, void* stream) {
  if (!hd || !h_params) return fail(MKWS_ERR_INVALID_ARG, "NULL argument");
  if (n != hd->nparams) return fail(MKWS_ERR_INVALID_ARG, "expected %d parameters, got %d", hd->nparams, n);
  hipStream_t s = static_cast<hipStream_t>(stream);
  MKWS_HIP(hipMemcpyAsync(h_params, hd->params, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
  MKWS_HIP(hipStreamSynchronize(s));
  return MKWS_OK;
}

int mkws_head_forward(mkws_head* hd, const float* d_emb, int B, float* d_probs, void* stream) {
  if (!hd) return fail(MKWS_ERR_INVALID_ARG, "head handle is NULL");
  if (B < 0) return fail(MKWS_ERR_INVALID_ARG, "negative batch");
  if (B == 0) return MKWS_OK;
  if (!d_emb || !d_probs) return fail(MKWS_ERR_INVALID_ARG, "NULL buffer");
  if (hd->d.in % 16 == 0) {        // matrix-core path (the same kernel serves mkws_heads_forward: results are bit-identical between the two)
    if (hd->d.hid <= 16)
      hipLaunchKernelGGL((head_fwd_mfma_kernel<1, false>), dim3((B + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), hd->d, hd->params, d_emb, B, d_probs, HeadTable());
    else
      hipLaunchKernelGGL((head_fwd_mfma_kernel<2, false>), dim3((B + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), hd->d, hd->params, d_emb, B, d_probs, HeadTable());
  } else {
    hipLaunchKernelGGL((head_rows_kernel<false>), dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), hd->d, hd->params, d_emb,
                       nullptr, B, d_probs, nullptr, nullptr, nullptr, nullptr);
  }
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_heads_forward(mkws_head* const* heads, int n_heads, const float* d_emb, int B, float* d_probs, void* stream) {
  if (!heads || n_heads < 0) return fail(MKWS_ERR_INVALID_ARG, "heads is NULL or n_heads < 0");
  if (B < 0) return fail(MKWS_ERR_INVALID_ARG, "negative batch");
  if (n_heads == 0 || B == 0) return MKWS_OK;
  if (!d_emb || !d_probs) return fail(MKWS_ERR_INVALID_ARG, "NULL buffer");
  for (int i = 0; i < n_heads; ++i) {
    if (!heads[i]) return fail(MKWS_ERR_INVALID_ARG, "head %d is NULL", i);
    if (heads[i]->d.in != heads[0]->d.in || heads[i]->d.hid != heads[0]->d.hid || heads[i]->d.cls != heads[0]->d.cls)
      return fail(MKWS_ERR_INVALID_ARG, "head %d has different dimensions than head 0", i);
  }
  const HeadDims d = heads[0]->d;
  for (int h0 = 0; h0 < n_heads; h0 += kHeadsPerLaunch) {
    const int n = (n_heads - h0 < kHeadsPerLaunch) ? n_heads - h0 : kHeadsPerLaunch;
    HeadTable t;
    for (int i = 0; i < kHeadsPerLaunch; ++i) t.p[i] = heads[h0 + (i < n ? i : 0)]->params;
    if (d.in % 16 == 0) {
      if (d.hid <= 16)
        hipLaunchKernelGGL((head_fwd_mfma_kernel<1, true>), dim3((B + 15) / 16, n), dim3(256), 0, static_cast<hipStream_t>(stream), d, nullptr, d_emb, B,
                           d_probs + (size_t)h0 * B * d.cls, t);
      else
        hipLaunchKernelGGL((head_fwd_mfma_kernel<2, true>), dim3((B + 15) / 16, n), dim3(256), 0, static_cast<hipStream_t>(stream), d, nullptr, d_emb, B,
                           d_probs + (size_t)h0 * B * d.cls, t);
    } else {
      hipLaunchKernelGGL((head_rows_kernel<false, true, 4>), dim3((B + 15) / 16, n), dim3(256), 0, static_cast<hipStream_t>(stream), d, nullptr, d_emb,
                         nullptr, B, d_probs + (size_t)h0 * B * d.cls, nullptr, nullptr, nullptr, nullptr, t);
    }
  }
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_head_loss_grad(mkws_head* hd, const float* d_emb, const int32_t* d_labels, int B, float* d_stats, void* stream) {
  if (!hd) return fail(MKWS_ERR_INVALID_ARG, "head handle is NULL");
  if (B <= 0 || B > hd->max_batch) return fail(MKWS_ERR_INVALID_ARG, "batch %d outside [1, max_batch=%d]", B, hd->max_batch);
  if (!d_emb || !d_labels) return fail(MKWS_ERR_INVALID_ARG, "NULL buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL((head_rows_kernel<true>), dim3((B + 3) / 4), dim3(256), 0, s, hd->d, hd->params, d_emb, d_labels, B, nullptr,
                     hd->hbuf, hd->dz, hd->dpre, hd->rowstat);
  const int rows_per = (B + kGradSplits - 1) / kGradSplits;
  const int splits = (B + rows_per - 1) / rows_per;
  hipLaunchKernelGGL(head_dw1_partial_kernel, dim3((hd->d.in + 255) / 256, splits), dim3(256), 0, s, hd->d, d_emb, hd->dpre, B, rows_per, hd->partial);
  const int total = hd->nparams + 2;
  const int fin_blocks = (hd->d.in * hd->d.hid + 255) / 256 + (total - hd->d.in * hd->d.hid + 3) / 4;
  hipLaunchKernelGGL(head_grad_finish_kernel, dim3(fin_blocks), dim3(256), 0, s, hd->d, hd->partial, splits, hd->hbuf, hd->dz, hd->dpre,
                     hd->rowstat, B, hd->grads, d_stats);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_head_input_grad(mkws_head* hd, float* d_dx, int B, void* stream) {
  if (!hd || !d_dx) return fail(MKWS_ERR_INVALID_ARG, "NULL argument");
  if (B <= 0 || B > hd->max_batch) return fail(MKWS_ERR_INVALID_ARG, "batch %d outside [1, max_batch=%d]", B, hd->max_batch);
  const size_t total = (size_t)B * hd->d.in;
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(head_input_grad_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), hd->d, hd->params, hd->dpre, B, d_dx);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_head_adam_step(mkws_head* hd, float lr, float beta1, float beta2, float eps, int step_t, float grad_scale, void* stream) {
  if (!hd) return fail(MKWS_ERR_INVALID_ARG, "head handle is NULL");
  if (step_t < 1) return fail(MKWS_ERR_INVALID_ARG, "Adam step index starts at 1");
  const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, step_t)) / (1.0 - std::pow((double)beta1, step_t));
  hipLaunchKernelGGL(head_adam_kernel, dim3((hd->nparams + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), hd->params, hd->grads,
                     hd->m, hd->v, hd->nparams, (float)lr_t, beta1, beta2, eps, grad_scale);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_head_adam_step_dev(mkws_head* hd, float lr, float beta1, float beta2, float eps, const int* d_step, float grad_scale, void* stream) {
  if (!hd || !d_step) return fail(MKWS_ERR_INVALID_ARG, "head handle / step counter is NULL");
  hipLaunchKernelGGL(head_adam_dev_kernel, dim3((hd->nparams + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), hd->params, hd->grads,
                     hd->m, hd->v, hd->nparams, lr, beta1, beta2, eps, d_step, grad_scale);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

}  // extern "C"
