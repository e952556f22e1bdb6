// mkws_embed.hip -- EfficientNet-B0 (49x40x1) + GAP + Dense 2048/2048/1024 forward on gfx950, fp32.
//
// Replaces `embedding.predict(x)` of the Keras model defined at
// multilingual_kws/train_multilingual_embedding.py:58-83 (cut at "dense_2" by
// multilingual_kws/embedding/transfer_learning.py:36-43).  Layer table: SURVEY.md Appendix B.
//
// Layout: activations NHWC fp32, viewed as row-major [M = B*H*W, C].  Kernels:
//   stem_kernel      3x3 s2 conv on the 1-channel spectrogram (+Rescaling, Normalization, BN, swish)
//   pw_gemm_kernel   every 1x1 conv and Dense layer: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) GEMM
//                    computed TRANSPOSED (weights are the MFMA "A" operand, activation rows the "B"
//                    operand) so each lane ends up with 4 consecutive output channels of one row and
//                    stores a float4.  Operands go HBM/L2 -> registers directly as float4: the
//                    reduction index k is permuted (lane group g of chunk j owns k = 16j+4g..+3) and
//                    the weights are pre-packed on the host in exactly that order, so no LDS
//                    transpose is needed.  Epilogue fuses BN scale/shift (or bias), activation,
//                    the SE excite gate on the input side, and the residual add.
//   dw_kernel        depthwise kxk (explicit TF/Keras padding) + BN + swish + SE squeeze sums
//   se_kernel        SE reduce FC + swish + expand FC + sigmoid -> per-(clip, channel) gate
//   mean_hw_kernel   global average pool
#include "mkws_common.h"
#include "mkws_embed_arch.h"

#include <cmath>
#include <new>
#include <string>
#include <vector>

namespace mkws {

using f32x4 = __attribute__((ext_vector_type(4))) float;

enum Act { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2, ACT_SELU = 3, ACT_SIGMOID = 4 };

__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_SWISH: return swishf_(v);
    case ACT_RELU: return v > 0.0f ? v : 0.0f;
    case ACT_SELU: return 1.0507009873554805f * (v > 0.0f ? v : 1.6732632423543772f * expm1f(v));
    case ACT_SIGMOID: return sigmoidf_(v);
    default: return v;
  }
}

// ------------------------------------------------------------------------------------------------
// stem: spec [B,49,40] -> [B,25,20,32]; ZeroPadding2D(((1,1),(0,1))) + Conv2D(32,3,s2,valid) + BN + swish
// 8 threads per output pixel, each 4 output channels (float4 store, fully coalesced).
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ spec, const float* __restrict__ w /*[9][32]*/,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float norm_mean, float norm_std, float* __restrict__ out, int B) {
  constexpr int H = kInH, W = kInW, Ho = 25, Wo = 20;
  const int q = threadIdx.x & 7;
  f32x4 wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const f32x4*>(w + t * 32 + q * 4);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + q * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + q * 4);
  const long total = (long)B * Ho * Wo;
  for (long pix = (long)blockIdx.x * 32 + (threadIdx.x >> 3); pix < total; pix += (long)gridDim.x * 32) {
    const int b = (int)(pix / (Ho * Wo));
    const int r = (int)(pix % (Ho * Wo));
    const int oh = r / Wo, ow = r % Wo;
    const float* img = spec + (size_t)b * H * W;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ih = oh * 2 - 1 + i;          // pad top 1
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int iw = ow * 2 + j;            // pad left 0
        float v = 0.0f;
        if (ih >= 0 && ih < H && iw < W) v = __fdiv_rn(img[ih * W + iw] * (1.0f / 255.0f) - norm_mean, norm_std);
        acc += wk[i * 3 + j] * v;
      }
    }
    f32x4 y = acc * sc + sh;
    y.x = swishf_(y.x); y.y = swishf_(y.y); y.z = swishf_(y.z); y.w = swishf_(y.w);
    *reinterpret_cast<f32x4*>(out + pix * 32 + q * 4) = y;
  }
}

// ------------------------------------------------------------------------------------------------
// 1x1 conv / dense GEMM.  Y[m, n] = act((sum_k X[m,k] * gate[m/HW, k] * W[k,n]) * scale[n] + shift[n]) + R[m,n]
// Packed weights: Wp[((nt*KC + j)*4 + g)*64 + c*4 + s] = W[16j + 4g + s][16nt + c]  (zero padded).
// Block = 4 waves; wave w owns rows [blockIdx.x*128 + 32w, +32) and NT n-tiles starting at blockIdx.y*NT.
struct GemmArgs {
  const float* X; int ldx;
  const float* Wp; const float* scale; const float* shift;
  const float* gate; int HW;          // gate [B, K] (K = ldg), rows of one clip = HW
  const float* R; int ldr;
  float* Y; int ldy;
  int M, K, N, KC, NTtot, act;
};

template <int NT, bool GATE>
__global__ __launch_bounds__(256) void pw_gemm_kernel(GemmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int m0 = blockIdx.x * 128 + wave * 32;
  const int nt0 = blockIdx.y * NT;
  if (m0 >= a.M) return;

  const float* xrow[2];
  const float* grow[2];
  bool rowok[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + mt * 16 + c;
    rowok[mt] = m < a.M;
    const int mm = rowok[mt] ? m : (a.M - 1);
    xrow[mt] = a.X + (size_t)mm * a.ldx + 4 * g;
    grow[mt] = GATE ? (a.gate + (size_t)(mm / a.HW) * a.K + 4 * g) : nullptr;
  }
  const float* wbase = a.Wp + ((size_t)nt0 * a.KC * 4 + g) * 64 + c * 4;

  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 xb[2], wb[NT];
  auto load = [&](int j, f32x4 (&xv)[2], f32x4 (&wv)[NT]) {
    const bool kok = (16 * j + 4 * g) < a.K;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kok && rowok[mt]) {
        v = *reinterpret_cast<const f32x4*>(xrow[mt] + 16 * j);
        if (GATE) v *= *reinterpret_cast<const f32x4*>(grow[mt] + 16 * j);
      }
      xv[mt] = v;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (nt0 + nt < a.NTtot) v = *reinterpret_cast<const f32x4*>(wbase + ((size_t)nt * a.KC + j) * 256);
      wv[nt] = v;
    }
  };
  load(0, xb, wb);
  for (int j = 0; j < a.KC; ++j) {
    f32x4 xn[2], wn[NT];
    if (j + 1 < a.KC) load(j + 1, xn, wn);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[nt][s], xb[mt][s], acc[mt][nt], 0, 0, 0);
    if (j + 1 < a.KC) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) xb[mt] = xn[mt];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wb[nt] = wn[nt];
    }
  }
  // epilogue: lane (g, c) holds rows m = m0 + mt*16 + c, channels n = 16*(nt0+nt) + 4g .. +3
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (nt0 + nt) * 16 + 4 * g;
    if (n >= a.N) continue;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (!rowok[mt]) continue;
      const size_t m = (size_t)(m0 + mt * 16 + c);
      f32x4 y = acc[mt][nt] * sc + sh;
      if (a.act != ACT_NONE) {
        y.x = apply_act(y.x, a.act); y.y = apply_act(y.y, a.act); y.z = apply_act(y.z, a.act); y.w = apply_act(y.w, a.act);
      }
      if (a.R) y += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
      *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = y;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// depthwise conv: X [B,H,W,C] -> Y [B,Ho,Wo,C], + BN + swish, and per-(b,c) sums of Y for SE squeeze.
// block = one clip x CQB channel quads; thread = (pixel lane tp, quad tq), tq fastest (coalesced).
template <int KS, int S>
__global__ __launch_bounds__(256) void dw_kernel(const float* __restrict__ X, const float* __restrict__ Wd /*[KS*KS][C]*/,
                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                 float* __restrict__ Y, float* __restrict__ sums /*[B,C]*/,
                                                 int H, int W, int C, int Ho, int Wo, int pt, int pl, int CQB, int P) {
  __shared__ f32x4 s_red[256];
  const int b = blockIdx.x;
  const int tq = threadIdx.x % CQB, tp = threadIdx.x / CQB;
  const int cq = blockIdx.y * CQB + tq;
  const bool active = tp < P && cq * 4 < C;
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const int c0 = cq * 4;
    f32x4 wk[KS * KS];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) wk[t] = *reinterpret_cast<const f32x4*>(Wd + (size_t)t * C + c0);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c0);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c0);
    const float* xin = X + (size_t)b * H * W * C + c0;
    float* yout = Y + (size_t)b * Ho * Wo * C + c0;
    for (int p = tp; p < Ho * Wo; p += P) {
      const int oh = p / Wo, ow = p % Wo;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const int ih = oh * S - pt + i;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          const int iw = ow * S - pl + j;
          if (iw < 0 || iw >= W) continue;
          acc += *reinterpret_cast<const f32x4*>(xin + ((size_t)ih * W + iw) * C) * wk[i * KS + j];
        }
      }
      f32x4 y = acc * sc + sh;
      y.x = swishf_(y.x); y.y = swishf_(y.y); y.z = swishf_(y.z); y.w = swishf_(y.w);
      *reinterpret_cast<f32x4*>(yout + (size_t)p * C) = y;
      ssum += y;
    }
  }
  s_red[threadIdx.x] = ssum;
  __syncthreads();
  if (tp == 0 && cq * 4 < C && threadIdx.x < CQB) {
    f32x4 t = s_red[tq];
    for (int k = 1; k < P; ++k) t += s_red[k * CQB + tq];
    *reinterpret_cast<f32x4*>(sums + (size_t)b * C + cq * 4) = t;
  }
}

// ------------------------------------------------------------------------------------------------
// SE: mean = sums/HW; r = swish(mean @ Wr + br); gate = sigmoid(r @ We + be).  4 clips per block.
__global__ __launch_bounds__(256) void se_kernel(const float* __restrict__ sums, float inv_hw, const float* __restrict__ Wr /*[C][se]*/,
                                                 const float* __restrict__ br, const float* __restrict__ We /*[se][C]*/,
                                                 const float* __restrict__ be, float* __restrict__ gate, int B, int C, int se) {
  extern __shared__ __attribute__((aligned(16))) float s_se[];
  float* s_mean = s_se;            // [4][C]
  float* s_r = s_se + 4 * C;       // [4][64]
  const int b0 = blockIdx.x * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * C; i += 256) {
    const int bb = i / C, cc = i % C;
    s_mean[i] = (b0 + bb < B) ? sums[(size_t)(b0 + bb) * C + cc] * inv_hw : 0.0f;
  }
  __syncthreads();
  if (lane < se) {
    const float* mrow = s_mean + wave * C;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int cc = 0;
    for (; cc + 3 < C; cc += 4) {
      acc0 += mrow[cc] * Wr[(size_t)cc * se + lane];
      acc1 += mrow[cc + 1] * Wr[(size_t)(cc + 1) * se + lane];
      acc2 += mrow[cc + 2] * Wr[(size_t)(cc + 2) * se + lane];
      acc3 += mrow[cc + 3] * Wr[(size_t)(cc + 3) * se + lane];
    }
    for (; cc < C; ++cc) acc0 += mrow[cc] * Wr[(size_t)cc * se + lane];
    s_r[wave * 64 + lane] = swishf_(((acc0 + acc1) + (acc2 + acc3)) + br[lane]);
  }
  __syncthreads();
  for (int cc = tid; cc < C; cc += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j = 0; j < se; ++j) {
      const float w = We[(size_t)j * C + cc];
      a0 += s_r[j] * w; a1 += s_r[64 + j] * w; a2 += s_r[128 + j] * w; a3 += s_r[192 + j] * w;
    }
    const float bias = be[cc];
    if (b0 + 0 < B) gate[(size_t)(b0 + 0) * C + cc] = sigmoidf_(a0 + bias);
    if (b0 + 1 < B) gate[(size_t)(b0 + 1) * C + cc] = sigmoidf_(a1 + bias);
    if (b0 + 2 < B) gate[(size_t)(b0 + 2) * C + cc] = sigmoidf_(a2 + bias);
    if (b0 + 3 < B) gate[(size_t)(b0 + 3) * C + cc] = sigmoidf_(a3 + bias);
  }
}

// global average pool: X [B, HW, C] -> Y [B, C]
__global__ __launch_bounds__(256) void mean_hw_kernel(const float* __restrict__ X, float* __restrict__ Y, int B, int HW, int C) {
  const int cq = C / 4;
  const long total = (long)B * cq;
  const float inv = 1.0f / (float)HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int b = (int)(i / cq), q = (int)(i % cq);
    const float* p = X + (size_t)b * HW * C + q * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < HW; ++k) s += *reinterpret_cast<const f32x4*>(p + (size_t)k * C);
    *reinterpret_cast<f32x4*>(Y + (size_t)b * C + q * 4) = s * inv;
  }
}

// ================================================================================================
// host side
// ================================================================================================
struct GemmLayer {   // device pointers into the weight blob
  const float* Wp = nullptr; const float* scale = nullptr; const float* shift = nullptr;
  int K = 0, N = 0, KC = 0, NTtot = 0;
};
struct DwLayer { const float* Wd = nullptr; const float* scale = nullptr; const float* shift = nullptr; };
struct SeLayer { const float* Wr = nullptr; const float* br = nullptr; const float* We = nullptr; const float* be = nullptr; int se = 0; };

struct BlockPlan {
  MBConvSpec spec;
  int ce, H, W, Ho, Wo, pt, pl;
  bool has_expand, residual;
  GemmLayer expand, project;
  DwLayer dw;
  SeLayer se;
};

}  // namespace mkws

using namespace mkws;

struct mkws_embed {
  int max_batch = 0;
  int device = 0;
  float* d_weights = nullptr;     // packed device weights
  float* d_ws = nullptr;          // workspace
  // workspace carve (floats per clip in parentheses)
  float *bufA = nullptr, *bufB = nullptr;   // block in/out ping-pong (16000)
  float *bufE = nullptr;                    // expand output (48000)
  float *bufD = nullptr;                    // depthwise output (18720)
  float *sums = nullptr, *gate = nullptr;   // (1152 each)
  float *gap = nullptr, *d0 = nullptr, *d1 = nullptr;   // (1280, 2048, 2048)
  // plan
  const float *stem_w = nullptr, *stem_scale = nullptr, *stem_shift = nullptr;
  float norm_mean = 0.f, norm_std = 1.f;
  BlockPlan blocks[kNumBlocks];
  GemmLayer top, dense0, dense1, dense2;
  int topH = 0, topW = 0;
};

namespace {

// ---- host-side packing -------------------------------------------------------------------------------
struct Packer {
  std::vector<float> buf;
  size_t add(const float* p, size_t n) {
    size_t off = (buf.size() + 3) & ~size_t(3);
    buf.resize(off + n, 0.0f);
    if (p) memcpy(buf.data() + off, p, n * sizeof(float));
    return off;
  }
  size_t reserve(size_t n) { return add(nullptr, n); }
};

struct GemmOff { size_t Wp, scale, shift; int K, N, KC, NTtot; };

// W: Keras [K, N] row-major.  scale/shift: per output channel (already folded).
GemmOff pack_gemm(Packer& pk, const float* W, int K, int N, const std::vector<float>& scale, const std::vector<float>& shift) {
  GemmOff o;
  o.K = K; o.N = N; o.KC = (K + 15) / 16; o.NTtot = (N + 15) / 16;
  const size_t n = (size_t)o.NTtot * o.KC * 256;
  o.Wp = pk.reserve(n);
  float* dst = pk.buf.data() + o.Wp;
  for (int nt = 0; nt < o.NTtot; ++nt)
    for (int j = 0; j < o.KC; ++j)
      for (int g = 0; g < 4; ++g)
        for (int c = 0; c < 16; ++c)
          for (int s = 0; s < 4; ++s) {
            const int k = 16 * j + 4 * g + s, col = 16 * nt + c;
            dst[(((size_t)nt * o.KC + j) * 4 + g) * 64 + c * 4 + s] = (k < K && col < N) ? W[(size_t)k * N + col] : 0.0f;
          }
  const int Np = o.NTtot * 16;
  std::vector<float> sc(Np, 0.0f), sh(Np, 0.0f);
  for (int i = 0; i < N; ++i) { sc[i] = scale[i]; sh[i] = shift[i]; }
  o.scale = pk.add(sc.data(), Np);
  o.shift = pk.add(sh.data(), Np);
  return o;
}

void fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, int C, std::vector<float>* scale, std::vector<float>* shift) {
  scale->resize(C); shift->resize(C);
  for (int i = 0; i < C; ++i) {
    const float s = gamma[i] / std::sqrt(var[i] + kBnEps);
    (*scale)[i] = s;
    (*shift)[i] = beta[i] - mean[i] * s;
  }
}

void correct_pad(int H, int W, int k, int* pt, int* pb, int* pl, int* pr) {
  const int c = k / 2;
  *pt = c - (1 - H % 2); *pb = c; *pl = c - (1 - W % 2); *pr = c;
}

int pick_cqb(int cq) {   // largest divisor of cq that is <= 64
  for (int d = (cq < 64 ? cq : 64); d >= 1; --d) if (cq % d == 0) return d;
  return 1;
}

template <bool GATE>
void launch_gemm_nt(int NT, dim3 grid, hipStream_t s, const GemmArgs& a) {
  switch (NT) {
    case 1: hipLaunchKernelGGL((pw_gemm_kernel<1, GATE>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((pw_gemm_kernel<2, GATE>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((pw_gemm_kernel<3, GATE>), grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((pw_gemm_kernel<4, GATE>), grid, dim3(256), 0, s, a); break;
    case 5: hipLaunchKernelGGL((pw_gemm_kernel<5, GATE>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((pw_gemm_kernel<6, GATE>), grid, dim3(256), 0, s, a); break;
  }
}

void launch_gemm(hipStream_t s, const GemmLayer& L, const float* X, int ldx, int M, int act, const float* gate, int HW,
                 const float* R, int ldr, float* Y, int ldy) {
  GemmArgs a;
  a.X = X; a.ldx = ldx; a.Wp = L.Wp; a.scale = L.scale; a.shift = L.shift; a.gate = gate; a.HW = HW > 0 ? HW : 1;
  a.R = R; a.ldr = ldr; a.Y = Y; a.ldy = ldy; a.M = M; a.K = L.K; a.N = L.N; a.KC = L.KC; a.NTtot = L.NTtot; a.act = act;
  const int nblk = (L.NTtot + 5) / 6;
  const int NT = (L.NTtot + nblk - 1) / nblk;
  dim3 grid((M + 127) / 128, (L.NTtot + NT - 1) / NT);
  if (gate) launch_gemm_nt<true>(NT, grid, s, a);
  else launch_gemm_nt<false>(NT, grid, s, a);
}

void launch_dw(hipStream_t s, const BlockPlan& b, const float* X, float* Y, float* sums, int B) {
  const int cq = b.ce / 4;
  const int CQB = pick_cqb(cq);
  int P = 256 / CQB;
  const int npix = b.Ho * b.Wo;
  if (P > npix) P = npix;
  dim3 grid(B, cq / CQB);
#define MKWS_DW(KS, S) hipLaunchKernelGGL((dw_kernel<KS, S>), grid, dim3(256), 0, s, X, b.dw.Wd, b.dw.scale, b.dw.shift, Y, sums, \
                                          b.H, b.W, b.ce, b.Ho, b.Wo, b.pt, b.pl, CQB, P)
  if (b.spec.kernel == 3 && b.spec.stride == 1) MKWS_DW(3, 1);
  else if (b.spec.kernel == 3) MKWS_DW(3, 2);
  else if (b.spec.stride == 1) MKWS_DW(5, 1);
  else MKWS_DW(5, 2);
#undef MKWS_DW
}

// Runs the network; stops after `stop` (nullptr = run everything).  On stop, *tap_src/*tap_count describe
// the buffer holding that stage's output.
int run_forward(mkws_embed* em, const float* d_spec, int B, float* d_emb, hipStream_t s, const char* stop,
                const float** tap_src, size_t* tap_count) {
  auto hit = [&](const std::string& name, const float* p, size_t n) {
    if (stop && name == stop) { *tap_src = p; *tap_count = n; return true; }
    return false;
  };
  {
    const long pix = (long)B * 500;
    int grid = (int)((pix + 31) / 32);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(stem_kernel, dim3(grid), dim3(256), 0, s, d_spec, em->stem_w, em->stem_scale, em->stem_shift,
                       em->norm_mean, em->norm_std, em->bufA, B);
  }
  if (hit("stem", em->bufA, (size_t)B * 500 * kStemCh)) return MKWS_OK;
  float* cur = em->bufA;
  float* nxt = em->bufB;
  for (int i = 0; i < kNumBlocks; ++i) {
    const BlockPlan& b = em->blocks[i];
    const std::string p = std::string("block") + b.spec.name;
    const int Min = B * b.H * b.W, Mout = B * b.Ho * b.Wo;
    const float* dw_in = cur;
    if (b.has_expand) {
      launch_gemm(s, b.expand, cur, b.spec.in_ch, Min, ACT_SWISH, nullptr, 0, nullptr, 0, em->bufE, b.ce);
      dw_in = em->bufE;
      if (hit(p + "_expand", em->bufE, (size_t)Min * b.ce)) return MKWS_OK;
    }
    launch_dw(s, b, dw_in, em->bufD, em->sums, B);
    if (hit(p + "_dw", em->bufD, (size_t)Mout * b.ce)) return MKWS_OK;
    hipLaunchKernelGGL(se_kernel, dim3((B + 3) / 4), dim3(256), (4 * b.ce + 256) * sizeof(float), s, em->sums,
                       1.0f / (float)(b.Ho * b.Wo), b.se.Wr, b.se.br, b.se.We, b.se.be, em->gate, B, b.ce, b.se.se);
    if (hit(p + "_gate", em->gate, (size_t)B * b.ce)) return MKWS_OK;
    launch_gemm(s, b.project, em->bufD, b.ce, Mout, ACT_NONE, em->gate, b.Ho * b.Wo, b.residual ? cur : nullptr,
                b.spec.out_ch, nxt, b.spec.out_ch);
    if (hit(p, nxt, (size_t)Mout * b.spec.out_ch)) return MKWS_OK;
    float* t = cur; cur = nxt; nxt = t;
  }
  const int HWt = em->topH * em->topW;
  launch_gemm(s, em->top, cur, em->top.K, B * HWt, ACT_SWISH, nullptr, 0, nullptr, 0, em->bufE, kTopCh);
  if (hit("top", em->bufE, (size_t)B * HWt * kTopCh)) return MKWS_OK;
  {
    const long total = (long)B * (kTopCh / 4);
    hipLaunchKernelGGL(mean_hw_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, em->bufE, em->gap, B, HWt, kTopCh);
  }
  if (hit("gap", em->gap, (size_t)B * kTopCh)) return MKWS_OK;
  launch_gemm(s, em->dense0, em->gap, kTopCh, B, ACT_RELU, nullptr, 0, nullptr, 0, em->d0, kDense0);
  if (hit("dense", em->d0, (size_t)B * kDense0)) return MKWS_OK;
  launch_gemm(s, em->dense1, em->d0, kDense0, B, ACT_RELU, nullptr, 0, nullptr, 0, em->d1, kDense1);
  if (hit("dense_1", em->d1, (size_t)B * kDense1)) return MKWS_OK;
  float* out = d_emb ? d_emb : em->d0;
  launch_gemm(s, em->dense2, em->d1, kDense1, B, ACT_SELU, nullptr, 0, nullptr, 0, out, kEmbDim);
  if (hit("dense_2", out, (size_t)B * kEmbDim)) return MKWS_OK;
  if (stop) return fail(MKWS_ERR_INVALID_ARG, "unknown stage '%s'", stop);
  return MKWS_OK;
}

}  // namespace

extern "C" {

size_t mkws_embed_weight_count(void) {
  const auto v = enumerate_tensors();
  return v.back().offset + v.back().count;
}

int mkws_embed_weight_manifest(char* dst, size_t cap) {
  const auto v = enumerate_tensors();
  std::string s = "{\"tensors\": [";
  for (size_t i = 0; i < v.size(); ++i) {
    if (i) s += ", ";
    s += "{\"name\": \"" + v[i].name + "\", \"shape\": [";
    for (size_t d = 0; d < v[i].shape.size(); ++d) { if (d) s += ", "; s += std::to_string(v[i].shape[d]); }
    s += "], \"offset\": " + std::to_string(v[i].offset) + ", \"count\": " + std::to_string(v[i].count) + "}";
  }
  s += "]}";
  if (dst && cap > 0) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(dst, s.data(), n);
    dst[n] = 0;
  }
  return (int)s.size();
}

int mkws_embed_create(const float* h, size_t n_floats, int max_batch, mkws_embed** out) {
  if (!h || !out) return fail(MKWS_ERR_INVALID_ARG, "weights/out is NULL");
  *out = nullptr;
  if (max_batch <= 0) return fail(MKWS_ERR_INVALID_ARG, "max_batch must be positive");
  if (n_floats != mkws_embed_weight_count())
    return fail(MKWS_ERR_BAD_WEIGHTS, "weight blob has %zu floats, architecture needs %zu", n_floats, mkws_embed_weight_count());
  int rc = require_device();
  if (rc != MKWS_OK) return rc;
  const auto tens = enumerate_tensors();
  auto T = [&](const std::string& name) -> const float* {
    for (const auto& t : tens) if (t.name == name) return h + t.offset;
    return nullptr;
  };
  for (size_t i = 0; i < n_floats; ++i)
    if (!std::isfinite(h[i])) return fail(MKWS_ERR_BAD_WEIGHTS, "weight blob has a non-finite value at %zu", i);

  mkws_embed* em = new (std::nothrow) mkws_embed();
  if (!em) return fail(MKWS_ERR_ALLOC, "out of host memory");
  em->max_batch = max_batch;
  (void)hipGetDevice(&em->device);
  Packer pk;
  std::vector<float> sc, sh;

  // stem
  em->norm_mean = *T("normalization/mean");
  {
    const float sd = std::sqrt(*T("normalization/variance"));
    em->norm_std = sd > 1e-7f ? sd : 1e-7f;
  }
  const size_t o_stem_w = pk.add(T("stem_conv/kernel"), 9 * kStemCh);
  fold_bn(T("stem_bn/gamma"), T("stem_bn/beta"), T("stem_bn/moving_mean"), T("stem_bn/moving_variance"), kStemCh, &sc, &sh);
  const size_t o_stem_sc = pk.add(sc.data(), kStemCh), o_stem_sh = pk.add(sh.data(), kStemCh);

  struct BlockOff { GemmOff expand, project; size_t dw_w, dw_sc, dw_sh, se_wr, se_br, se_we, se_be; } bo[kNumBlocks];
  int H = 25, W = 20;
  for (int i = 0; i < kNumBlocks; ++i) {
    BlockPlan& b = em->blocks[i];
    b.spec = kBlocks[i];
    const std::string p = std::string("block") + b.spec.name;
    b.ce = b.spec.in_ch * b.spec.expand;
    b.has_expand = b.spec.expand != 1;
    b.residual = (b.spec.stride == 1 && b.spec.in_ch == b.spec.out_ch);
    b.H = H; b.W = W;
    if (b.spec.stride == 2) {
      int pt, pb, pl, pr;
      correct_pad(H, W, b.spec.kernel, &pt, &pb, &pl, &pr);
      b.pt = pt; b.pl = pl;
      b.Ho = (H + pt + pb - b.spec.kernel) / 2 + 1;
      b.Wo = (W + pl + pr - b.spec.kernel) / 2 + 1;
    } else {
      b.pt = b.pl = b.spec.kernel / 2;
      b.Ho = H; b.Wo = W;
    }
    const int se = se_channels(b.spec);
    b.se.se = se;
    std::vector<float> ones;
    if (b.has_expand) {
      fold_bn(T(p + "_expand_bn/gamma"), T(p + "_expand_bn/beta"), T(p + "_expand_bn/moving_mean"), T(p + "_expand_bn/moving_variance"), b.ce, &sc, &sh);
      bo[i].expand = pack_gemm(pk, T(p + "_expand_conv/kernel"), b.spec.in_ch, b.ce, sc, sh);
    }
    bo[i].dw_w = pk.add(T(p + "_dwconv/depthwise_kernel"), (size_t)b.spec.kernel * b.spec.kernel * b.ce);
    fold_bn(T(p + "_bn/gamma"), T(p + "_bn/beta"), T(p + "_bn/moving_mean"), T(p + "_bn/moving_variance"), b.ce, &sc, &sh);
    bo[i].dw_sc = pk.add(sc.data(), b.ce); bo[i].dw_sh = pk.add(sh.data(), b.ce);
    bo[i].se_wr = pk.add(T(p + "_se_reduce/kernel"), (size_t)b.ce * se);
    bo[i].se_br = pk.add(T(p + "_se_reduce/bias"), se);
    bo[i].se_we = pk.add(T(p + "_se_expand/kernel"), (size_t)se * b.ce);
    bo[i].se_be = pk.add(T(p + "_se_expand/bias"), b.ce);
    fold_bn(T(p + "_project_bn/gamma"), T(p + "_project_bn/beta"), T(p + "_project_bn/moving_mean"), T(p + "_project_bn/moving_variance"), b.spec.out_ch, &sc, &sh);
    bo[i].project = pack_gemm(pk, T(p + "_project_conv/kernel"), b.ce, b.spec.out_ch, sc, sh);
    H = b.Ho; W = b.Wo;
  }
  em->topH = H; em->topW = W;
  fold_bn(T("top_bn/gamma"), T("top_bn/beta"), T("top_bn/moving_mean"), T("top_bn/moving_variance"), kTopCh, &sc, &sh);
  const GemmOff o_top = pack_gemm(pk, T("top_conv/kernel"), kBlocks[kNumBlocks - 1].out_ch, kTopCh, sc, sh);
  auto dense = [&](const char* name, int K, int N) {
    std::vector<float> one(N, 1.0f), bias(T(std::string(name) + "/bias"), T(std::string(name) + "/bias") + N);
    return pack_gemm(pk, T(std::string(name) + "/kernel"), K, N, one, bias);
  };
  const GemmOff o_d0 = dense("dense", kTopCh, kDense0);
  const GemmOff o_d1 = dense("dense_1", kDense0, kDense1);
  const GemmOff o_d2 = dense("dense_2", kDense1, kEmbDim);

  // upload
  const size_t wbytes = pk.buf.size() * sizeof(float);
  if (hipMalloc(reinterpret_cast<void**>(&em->d_weights), wbytes) != hipSuccess) { delete em; return fail(MKWS_ERR_ALLOC, "hipMalloc(%zu) for weights failed", wbytes); }
  if (hipMemcpy(em->d_weights, pk.buf.data(), wbytes, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(em->d_weights); delete em; return fail(MKWS_ERR_HIP, "weight upload failed");
  }
  const float* d = em->d_weights;
  auto G = [&](const GemmOff& o) { GemmLayer L; L.Wp = d + o.Wp; L.scale = d + o.scale; L.shift = d + o.shift; L.K = o.K; L.N = o.N; L.KC = o.KC; L.NTtot = o.NTtot; return L; };
  em->stem_w = d + o_stem_w; em->stem_scale = d + o_stem_sc; em->stem_shift = d + o_stem_sh;
  for (int i = 0; i < kNumBlocks; ++i) {
    BlockPlan& b = em->blocks[i];
    if (b.has_expand) b.expand = G(bo[i].expand);
    b.project = G(bo[i].project);
    b.dw.Wd = d + bo[i].dw_w; b.dw.scale = d + bo[i].dw_sc; b.dw.shift = d + bo[i].dw_sh;
    b.se.Wr = d + bo[i].se_wr; b.se.br = d + bo[i].se_br; b.se.We = d + bo[i].se_we; b.se.be = d + bo[i].se_be;
  }
  em->top = G(o_top); em->dense0 = G(o_d0); em->dense1 = G(o_d1); em->dense2 = G(o_d2);

  // workspace
  const size_t per_clip = 16000 * 2 + 48000 + 18720 + 1152 * 2 + 1280 + 2048 * 2;
  const size_t ws = per_clip * (size_t)max_batch + 64;
  if (hipMalloc(reinterpret_cast<void**>(&em->d_ws), ws * sizeof(float)) != hipSuccess) {
    (void)hipFree(em->d_weights); delete em; return fail(MKWS_ERR_ALLOC, "hipMalloc(%zu) for workspace failed", ws * sizeof(float));
  }
  float* w = em->d_ws;
  const size_t mb = (size_t)max_batch;
  em->bufA = w; w += 16000 * mb; em->bufB = w; w += 16000 * mb; em->bufE = w; w += 48000 * mb; em->bufD = w; w += 18720 * mb;
  em->sums = w; w += 1152 * mb; em->gate = w; w += 1152 * mb; em->gap = w; w += 1280 * mb; em->d0 = w; w += 2048 * mb; em->d1 = w;
  *out = em;
  return MKWS_OK;
}

void mkws_embed_destroy(mkws_embed* em) {
  if (!em) return;
  if (em->d_weights) (void)hipFree(em->d_weights);
  if (em->d_ws) (void)hipFree(em->d_ws);
  delete em;
}

int mkws_embed_forward(mkws_embed* em, const float* d_spec, int B, float* d_emb, void* stream) {
  if (!em) return fail(MKWS_ERR_INVALID_ARG, "embed handle is NULL");
  if (B < 0 || B > em->max_batch) return fail(MKWS_ERR_INVALID_ARG, "batch %d outside [0, max_batch=%d]", B, em->max_batch);
  if (B == 0) return MKWS_OK;
  if (!d_spec || !d_emb) return fail(MKWS_ERR_INVALID_ARG, "d_spec/d_emb is NULL");
  const float* src; size_t cnt;
  int rc = run_forward(em, d_spec, B, d_emb, static_cast<hipStream_t>(stream), nullptr, &src, &cnt);
  if (rc != MKWS_OK) return rc;
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_embed_forward_tap(mkws_embed* em, const float* d_spec, int B, const char* stage, float* d_dst, size_t cap_floats, void* stream) {
  if (!em || !stage) return fail(MKWS_ERR_INVALID_ARG, "embed handle/stage is NULL");
  if (B <= 0 || B > em->max_batch) return fail(MKWS_ERR_INVALID_ARG, "batch %d outside [1, max_batch=%d]", B, em->max_batch);
  if (!d_spec || !d_dst) return fail(MKWS_ERR_INVALID_ARG, "d_spec/d_dst is NULL");
  const float* src = nullptr; size_t cnt = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = run_forward(em, d_spec, B, nullptr, s, stage, &src, &cnt);
  if (rc != MKWS_OK) return rc;
  MKWS_HIP(hipGetLastError());
  if (cnt > cap_floats) return fail(MKWS_ERR_INVALID_ARG, "stage '%s' has %zu floats, destination holds %zu", stage, cnt, cap_floats);
  MKWS_HIP(hipMemcpyAsync(d_dst, src, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
  return (int)cnt;
}

}  // extern "C"
