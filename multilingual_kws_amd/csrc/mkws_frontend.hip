// mkws_frontend.hip -- batched TFLite-Micro "audio_microfrontend" on gfx950, bit-exact.
//
// Replaces the per-clip CPU op call of multilingual_kws/embedding/input_data.py:19-35.
// Arithmetic spec: SURVEY.md Appendix A (validated against upstream TF's known-answer vectors).
//
// Mapping (CDNA4, wave = 64 lanes):
//   * one workgroup per clip, NWAVES waves; wave w takes frames w, w+NWAVES, ...
//   * per frame, ONE WAVE does window -> block-shift -> 256-point complex radix-4 fixed-point FFT
//     (4 stages x 64 butterflies = exactly one butterfly per lane per stage) -> real-FFT post-pass
//     -> |X|^2 -> 40 triangular mel sums (uint64) -> rounded sqrt.  The 1 KB of FFT state lives in
//     LDS; window coefficients and twiddles are per-lane constants held in registers for the
//     whole clip.  Samples are read straight from HBM already in base-4 digit-reversed order, so
//     the first FFT stage runs out of registers.
//   * the only sequential part -- the noise-estimate recurrence over frames -- is a 49-step scan
//     by 40 lanes of wave 0 over the LDS-staged [frames x channels] tile; noise subtraction, PCAN
//     gain and the log LUT are then element-wise over the tile by all lanes.
//   * HBM traffic = the audio once (+ 160-sample frame overlaps that hit L2) and the [49,40] tile
//     out: 71 840 B per clip for fp32 input, the algorithmic minimum.
#include "mkws_common.h"
#include "mkws_frontend_tables.h"

#include <algorithm>
#include <new>
#include <vector>

namespace mkws {

// ------------------------------------------------------------------------------------------------
struct FrontendParams {
  // device tables
  const int16_t* window_coef;   // [512], zero padded
  const uint32_t* tw;           // [256] packed (re | im<<16)
  const uint32_t* stw;          // [128] packed
  const int16_t* out_start;     // [64] per-LANE filterbank tasks (see LaneConst)
  const int16_t* out_len;       // [64]
  const int16_t* out_off;       // [64]
  const int16_t* task_ch;       // [64] channel a helper lane (>= C) adds to, else -1
  const int16_t* task_helped;   // [64] 1 if a helper lane contributes to this lane's channel
  const int16_t* out_coef;      // [ncoef]
  const int16_t* pcan_lut;      // [128]
  const uint16_t* log_lut;      // [132]
  int ncoef;
  int fast48;                   // every channel's filterbank weights are >= 0 and add up to <= 2^16: mel sums < 2^48, 16-bit multiplies suffice
  int lane_off, nm;             // out_coef[lane_off + ((j >> 1) * 64 + lane) * 2 + (j & 1)], j < nm (nm % 4 == 0): lane `lane`'s tap list, zero padded to the longest
  int window_size, window_step, num_channels;
  int smoothing_bits, enable_pcan, enable_log, scale_shift, snr_shift, correction_bits;
  uint32_t even_smoothing, odd_smoothing, min_signal_remaining;
};

struct cpx { int r, i; };

__device__ __forceinline__ int sext16(int v) { return (int)(short)v; }
__device__ __forceinline__ int sround(int v) { return sext16((v + 16384) >> 15); }
__device__ __forceinline__ uint32_t pack(int r, int i) { return ((uint32_t)r & 0xFFFFu) | ((uint32_t)i << 16); }
__device__ __forceinline__ cpx unpack(uint32_t u) { cpx c; c.r = sext16((int)u); c.i = ((int)u) >> 16; return c; }

// LDS traffic between FFT stages is wave-private: DS operations of one wave execute in program
// order, so only the compiler has to be kept from reordering across the exchange.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Packed int16 complex (x = re, y = im; the same bits as the LDS word pack() builds).  kissfft's FIXED_POINT=16
// arithmetic stores every intermediate to int16, i.e. wraps mod 2^16 at every add -- exactly what the packed
// 16-bit VALU ops do (v_pk_add/sub_i16); the rounded products are 32-bit dot products (v_dot2c_i32_i16).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 mk2(int r, int i) { s16x2 v; v.x = (short)r; v.y = (short)i; return v; }
__device__ __forceinline__ uint32_t bits2(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 from2(uint32_t u) { return __builtin_bit_cast(s16x2, u); }
// C_FIXDIV: sround(v * (32767/div)) on both components (|result| <= |v|/div: no wrap).
// (c*k + 2^14) >> 15 == (c*2k + 2^15) >> 16, i.e. the HIGH half of a 32-bit multiply-add: v_mad_i32_i16 reads either 16-bit
// half of the packed value directly (op_sel) and one v_perm_b32 packs the two high halves -- 3 instructions per complex value
// instead of unpack (2) + mad (2) + shift (2) + pack (2).  2k = 16382 / 32766 still fits int16; |c*2k| < 2^31.
__device__ __forceinline__ s16x2 fixdiv_pk(s16x2 v, int k) {
  const uint32_t u = bits2(v);
  const int k2 = 2 * k, half = 32768;
  int lo, hi;
  asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(lo) : "v"(u), "v"(k2), "v"(half));
  asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(hi) : "v"(u), "v"(k2), "v"(half));
  return from2(__builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060302u));
}
// twiddle t as the two dot-product operands of C_MUL: (t.r, -t.i) and (t.i, t.r)   (|t| <= 32767: negation is exact)
struct tw2 { s16x2 a, b; };
__device__ __forceinline__ tw2 mktw(cpx t) { tw2 w; w.a = mk2(t.r, -t.i); w.b = mk2(t.i, t.r); return w; }
// (v_dot2_i32_i16 with a separate addend operand: the builtin selects the accumulate form v_dot2c, which costs a v_mov of
// the rounding constant per product)
__device__ __forceinline__ int dot2_i16(s16x2 a, s16x2 b, int c) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(bits2(a)), "v"(bits2(b)), "v"(c));
  return d;
}
__device__ __forceinline__ s16x2 cmul_pk(s16x2 a, tw2 t) {
  const int rnd = 16384;
  return mk2(dot2_i16(a, t.a, rnd) >> 15, dot2_i16(a, t.b, rnd) >> 15);
}
// max over the 64 lanes of a non-negative value, on the VALU's DPP paths (no LDS round trips): quads, half rows, rows,
// then rows 1,3 <- lane 15 of rows 0,2 and rows 2,3 <- lane 31; the result sits in lane 63
__device__ __forceinline__ int wave_max_nonneg(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false));   // row_half_mirror
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));   // row_mirror
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));   // row_bcast15
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));   // row_bcast31
  return __builtin_amdgcn_readlane(v, 63);
}

// kissfft kf_bfly4 (forward), one butterfly, on packed values.
__device__ __forceinline__ void bfly4(s16x2& F0, s16x2& F1, s16x2& F2, s16x2& F3, tw2 t1, tw2 t2, tw2 t3) {
  F0 = fixdiv_pk(F0, 8191); F1 = fixdiv_pk(F1, 8191); F2 = fixdiv_pk(F2, 8191); F3 = fixdiv_pk(F3, 8191);
  const s16x2 s0 = cmul_pk(F1, t1), s1 = cmul_pk(F2, t2), s2 = cmul_pk(F3, t3);
  const s16x2 s5 = F0 - s1, a0 = F0 + s1, s3 = s0 + s2, s4 = s0 - s2;
  const s16x2 rot = mk2(s4.y, -(int)s4.x);             // -j * s4
  F2 = a0 - s3; F0 = a0 + s3;
  F1 = s5 + rot; F3 = s5 - rot;
}

// bits.h Sqrt64/Sqrt32: floor sqrt, then +1 when the remainder exceeds the root (round to nearest),
// except at the saturation points 0xFFFF (32-bit path) / 0xFFFFFFFF.
__device__ __forceinline__ uint32_t sqrt64_round(uint64_t x) {
  if (x == 0) return 0;
  uint64_t r = (uint64_t)__builtin_sqrt((double)x);
  if (r > 0xFFFFFFFFull) r = 0xFFFFFFFFull;
  while (r * r > x) --r;
  while (r < 0xFFFFFFFFull && (r + 1) * (r + 1) <= x) ++r;
  const uint64_t rem = x - r * r;
  const uint64_t sat = ((x >> 32) == 0) ? 0xFFFFull : 0xFFFFFFFFull;
  if (rem > r && r != sat) ++r;
  return (uint32_t)r;
}

// The same rounded root for x < 2^48 (the mel sums of every configuration whose channel weights add up to <= 2^16: host-checked,
// FrontendParams::fast48) without 64-bit or double arithmetic.  r0 = trunc(sqrtf(float(x))) is within 4 of the integer root (2^-24 from
// each of the two conversions / the fma, 1 ulp from v_sqrt_f32, root < 2^24), so d = x - r0^2 fits an int32 and its LOW words suffice;
// k = floor(d / (2 r0)) is the correction up to one (exact for roots < 4096, where sqrtf is exact; the k^2 term is < 0.01 above), fixed by
// one exact step on t(k) = k (2 r0 + k) <= d < t(k + 1).  Checked against the integer definition on 7e5 values incl. every boundary
// class with r0 forced off by -4..+4 (profiles/r04_notes.md section 7).
__device__ __forceinline__ uint32_t sqrt48_round(uint64_t x) {
  const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
  const float xf = __builtin_fmaf((float)hi, 4294967296.0f, (float)lo);
  int r0 = (int)__builtin_amdgcn_sqrtf(xf);
  r0 = min(max(r0, 1), 0xFFFFFF);
  const int d = (int)(lo - __umul24((uint32_t)r0, (uint32_t)r0));
  const int two = 2 * r0;
  int k = (int)__builtin_floorf((float)d * __builtin_amdgcn_rcpf((float)two));
  int t = k * (two + k);
  {
    const int up = t + two + 2 * k + 1, dn = t - (two + 2 * k - 1);
    const bool dec = t > d, inc = !dec && up <= d;
    t = dec ? dn : (inc ? up : t);
    k = dec ? k - 1 : (inc ? k + 1 : k);
  }
  uint32_t r = (uint32_t)(r0 + k);
  const uint32_t rem = (uint32_t)(d - t);
  const uint32_t sat = (hi == 0) ? 0xFFFFu : 0xFFFFFFFFu;
  if (rem > r && r != sat) ++r;
  return x == 0 ? 0u : r;
}

// pcan_gain_control.c WideDynamicFunction
__device__ __forceinline__ int wide_dynamic(uint32_t x, const int16_t* lut) {
  if (x <= 2) return lut[x];
  const int interval = 32 - __clz((int)x);
  const int16_t* l = lut + 4 * interval - 6;
  const int frac = (int)(((interval < 11) ? (x << (11 - interval)) : (x >> (interval - 11))) & 0x3FF);
  int result = ((int)l[2] * frac) >> 5;
  result += (int)((uint32_t)(int)l[1] << 5);
  result *= frac;
  result = (result + (1 << 14)) >> 15;
  result += l[0];
  return sext16(result);
}

// log_scale.c Log (with Log2FractionPart)
__device__ __forceinline__ uint32_t log_scale(uint32_t x, int scale_shift, const uint16_t* lut) {
  const uint32_t integer = 31 - __clz((int)x);
  int frac = (int)(x - (1u << integer));
  if (integer < 16) frac <<= (16 - integer); else frac >>= (integer - 16);
  const uint32_t seg = (uint32_t)frac >> 9;
  const int c0 = lut[seg], c1 = lut[seg + 1];
  const int rel = ((c1 - c0) * (frac - (int)(seg << 9))) >> 16;
  const uint32_t fraction = (uint32_t)(frac + c0 + rel);
  const uint32_t log2v = (integer << 16) + fraction;
  const uint32_t loge = (uint32_t)((45426ull * log2v + 32768u) >> 16);
  return ((loge << scale_shift) + 32768u) >> 16;
}

// Per-lane constants that do not change across the frames of a clip.
struct LaneConst {
  int coef[8];        // window coefficients of this lane's 8 samples
  tw2 twB[3], twC[3], twD[3];
  tw2 st1, st2;       // super twiddles for k = lane+1 and k = lane+65
  int n0;             // base-4 digit reversal of the lane id (3 digits)
  int toff[4];        // sample offset of pair j inside the frame, clamped into the window (loads are unconditional)
  int tsel[4];        // 2: both samples inside the window, 1: only the first (odd window), 0: none (zero padding)
  // mel filterbank task of this lane: bins [fb_start, +fb_len) with coefficients from fb_off; lanes < C own channel = lane,
  // lanes >= C may help the channel fb_ch (second half of a long tap list); fb_helped: a helper adds to this lane's channel
  int fb_start, fb_len, fb_off, fb_ch, fb_helped;
  // LDS word index of every FFT-state access of this lane, SWIZZLED: z[i] lives at word i ^ ((i >> 3) & 31).  In the natural layout the
  // exchanges of stage B (stride-16 groups of four) and C (stride-64 groups of sixteen) put 8 / 4 lanes on one bank (8 / 4 passes per
  // 64-lane access instead of 2); with this XOR every access pattern of the transform -- stage A's 4 lane + q, B, C, D's lane + 64 q and the
  // post-pass's k / 256 - k -- is two lanes per bank (searched over XOR / padding families, tools/fft_lds_swizzle.py).  The indices do not
  // change from frame to frame: 20 registers of the 128 a wave may hold at four waves per SIMD.
  int ia[4], ib[4], ic[4], id[4], ip[4];
};
__device__ __forceinline__ int fft_swz(int i) { return i ^ ((i >> 3) & 31); }

__device__ __forceinline__ void init_lane_const(const FrontendParams& p, int lane, LaneConst& L) {
  const int d0 = lane & 3, d1 = (lane >> 2) & 3, d2 = (lane >> 4) & 3;
  L.n0 = d0 * 16 + d1 * 4 + d2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = L.n0 + 64 * j;
    L.coef[2 * j] = p.window_coef[2 * n];
    L.coef[2 * j + 1] = p.window_coef[2 * n + 1];
  }
  {  // stage B: m = 4, fstride = 16; stage C: m = 16, fstride = 4; stage D: m = 64, fstride = 1
    const int kB = lane & 3, kC = lane & 15, kD = lane;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      L.twB[q] = mktw(unpack(p.tw[kB * 16 * (q + 1)]));
      L.twC[q] = mktw(unpack(p.tw[kC * 4 * (q + 1)]));
      L.twD[q] = mktw(unpack(p.tw[kD * (q + 1)]));
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = 2 * (L.n0 + 64 * j);
    L.tsel[j] = (t + 1 < p.window_size) ? 2 : (t < p.window_size ? 1 : 0);
    L.toff[j] = (L.tsel[j] == 2) ? t : p.window_size - 2;     // the clamped pair ends at the window's last sample
  }
  L.st1 = mktw(unpack(p.stw[lane]));
  L.st2 = mktw(unpack(p.stw[lane + 64]));
  L.fb_start = p.out_start[lane]; L.fb_len = p.out_len[lane]; L.fb_off = p.out_off[lane];
  L.fb_ch = p.task_ch[lane]; L.fb_helped = p.task_helped[lane];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    L.ia[q] = fft_swz(4 * lane + q);
    L.ib[q] = fft_swz((lane >> 2) * 16 + (lane & 3) + 4 * q);
    L.ic[q] = fft_swz((lane >> 4) * 64 + (lane & 15) + 16 * q);
    L.id[q] = fft_swz(lane + 64 * q);
  }
  L.ip[0] = fft_swz(lane + 1); L.ip[1] = fft_swz(255 - lane); L.ip[2] = fft_swz(lane + 65); L.ip[3] = fft_swz(191 - lane);
}

// window.c + fft.c + kiss_fftr + filterbank.c for ONE frame by ONE wave.
//   x[8]   : this lane's 8 int16 samples (as ints), samples 2n,2n+1 for n = n0 + 64 j
//   fftbuf : wave-private LDS uint32[256];  ebuf: wave-private LDS uint32[256]
//   sig_out: where lane c < C writes channel c of this frame (LDS or global)
__device__ __forceinline__ void frame_to_sig(const FrontendParams& p, const LaneConst& L, int lane,
                                             const int (&x)[8], uint32_t* fftbuf, uint32_t* ebuf,
                                             const int16_t* s_coef, uint32_t* sig_out) {
  // ---- window (A.1) and block exponent (A.2), on packed int16 pairs ----
  // w = int16((x * coef) >> 12) is bits [12..27] of the product = the high half of (product << 4): one v_perm_b32 packs a pair
  s16x2 W[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t lo = (uint32_t)(x[2 * j] * L.coef[2 * j]) << 4, hi = (uint32_t)(x[2 * j + 1] * L.coef[2 * j + 1]) << 4;
    W[j] = from2(__builtin_amdgcn_perm(hi, lo, 0x07060302u));
  }
  s16x2 m2 = mk2(0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // int16 negate: -(-32768) stays -32768, as upstream's (int16_t)(-x).  Done on the unsigned lanes: wrap-around is defined there
    // (a signed vector subtract that overflows is not), same v_pk_sub instruction
    using u16x2 = __attribute__((ext_vector_type(2))) unsigned short;
    const s16x2 neg = __builtin_bit_cast(s16x2, (u16x2)(0) - __builtin_bit_cast(u16x2, W[j]));
    m2 = __builtin_elementwise_max(m2, __builtin_elementwise_max(W[j], neg));
  }
  const int mx = wave_max_nonneg(max((int)m2.x, (int)m2.y));
  const int shift = (mx == 0) ? 15 : (15 - (32 - __clz(mx)));
  // ---- stage A (m = 1): butterflies on z[n0 + 64 j], twiddle (32767, 0) ----
  // fft.c FftCompute: (int16)((uint16)w << shift), both halves at once
  const s16x2 sh2 = mk2(shift, shift);
  s16x2 F0 = W[0] << sh2, F1 = W[1] << sh2, F2 = W[2] << sh2, F3 = W[3] << sh2;
  const tw2 one = mktw(cpx{32767, 0});
  bfly4(F0, F1, F2, F3, one, one, one);
  fftbuf[L.ia[0]] = bits2(F0);
  fftbuf[L.ia[1]] = bits2(F1);
  fftbuf[L.ia[2]] = bits2(F2);
  fftbuf[L.ia[3]] = bits2(F3);
  wave_lds_sync();
  // ---- stage B (m = 4) ----
  {
    F0 = from2(fftbuf[L.ib[0]]); F1 = from2(fftbuf[L.ib[1]]); F2 = from2(fftbuf[L.ib[2]]); F3 = from2(fftbuf[L.ib[3]]);
    bfly4(F0, F1, F2, F3, L.twB[0], L.twB[1], L.twB[2]);
    fftbuf[L.ib[0]] = bits2(F0); fftbuf[L.ib[1]] = bits2(F1);
    fftbuf[L.ib[2]] = bits2(F2); fftbuf[L.ib[3]] = bits2(F3);
    wave_lds_sync();
  }
  // ---- stage C (m = 16) ----
  {
    F0 = from2(fftbuf[L.ic[0]]); F1 = from2(fftbuf[L.ic[1]]); F2 = from2(fftbuf[L.ic[2]]); F3 = from2(fftbuf[L.ic[3]]);
    bfly4(F0, F1, F2, F3, L.twC[0], L.twC[1], L.twC[2]);
    fftbuf[L.ic[0]] = bits2(F0); fftbuf[L.ic[1]] = bits2(F1);
    fftbuf[L.ic[2]] = bits2(F2); fftbuf[L.ic[3]] = bits2(F3);
    wave_lds_sync();
  }
  // ---- stage D (m = 64) ----
  {
    F0 = from2(fftbuf[L.id[0]]); F1 = from2(fftbuf[L.id[1]]); F2 = from2(fftbuf[L.id[2]]); F3 = from2(fftbuf[L.id[3]]);
    bfly4(F0, F1, F2, F3, L.twD[0], L.twD[1], L.twD[2]);
    fftbuf[L.id[0]] = bits2(F0); fftbuf[L.id[1]] = bits2(F1);
    fftbuf[L.id[2]] = bits2(F2); fftbuf[L.id[3]] = bits2(F3);
    wave_lds_sync();
  }
  // ---- real-FFT post-pass (kiss_fftr) + energy, bins k and 256-k for k = lane+1, lane+65 ----
  uint32_t ebig = 0;                                      // bit 31: some energy of this frame is exactly 2^31 (the value upstream sign-extends)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = lane + 1 + 64 * h;
    const tw2 st = h ? L.st2 : L.st1;
    const s16x2 t = from2(fftbuf[L.ip[2 * h + 1]]);                  // z[256 - k]
    const s16x2 fpk = fixdiv_pk(from2(fftbuf[L.ip[2 * h]]), 16383);   // z[k]
    const s16x2 fpnk = fixdiv_pk(mk2(t.x, -(int)t.y), 16383);
    const s16x2 f1 = fpk + fpnk, f2 = fpk - fpnk;
    const s16x2 tw = cmul_pk(f2, st);
    // the halving adds are done in int (no int16 wrap before the shift), as in kiss_fftr
    const int ar = ((int)f1.x + (int)tw.x) >> 1, ai = ((int)f1.y + (int)tw.y) >> 1;
    const int br = ((int)f1.x - (int)tw.x) >> 1, bi = ((int)tw.y - (int)f1.y) >> 1;
    // FilterbankConvertFftComplexToEnergy: uint32 r*r + i*i (can reach exactly 2^31)
    const uint32_t e1 = (uint32_t)(ar * ar) + (uint32_t)(ai * ai), e2 = (uint32_t)(br * br) + (uint32_t)(bi * bi);
    if (k != 128) ebuf[k] = e1;
    ebuf[256 - k] = e2;   // k == 128: the second write wins upstream
    ebig |= e1 | e2;
  }
  wave_lds_sync();
  // ---- mel filterbank (uint64 sums) + rounded sqrt ----
  // Lane c < C owns channel c; the tap lists of the longest channels are split in two and the lanes >= C take the second
  // halves (the loop runs max-length iterations for the whole wave: 14 instead of 28 for the 40-channel configuration).
  // uint64 sums wrap mod 2^64, so the split is exact.
  // Round 4: when no energy of the frame is 2^31 (wave-uniform test; the sign-extension quirk cannot fire) and the weights allow it
  // (fast48), the sums run on FULL-RATE 16-bit multiply-adds -- coef * e = coef * e.lo16 + (coef * e.hi16 << 16), two uint32 accumulators
  // per lane (<= 14 taps x 2^12 x 2^16 < 2^32) -- instead of quarter-rate v_mad_i64_i32, and the root on sqrt48_round.  Same integers.
  // Every lane walks the SAME number of taps (its list is zero padded to the longest, p.nm): a uniform loop of four taps per trip,
  // 6 LDS reads and 8 multiply-adds, no per-lane bounds (the energies read past a short list -- possibly past this wave's buffer, still
  // inside the workgroup's LDS -- meet zero coefficients).
  const bool fast = p.fast48 && __builtin_amdgcn_ballot_w64((ebig >> 31) != 0) == 0;
  if (fast) {
    uint32_t alo = 0, ahi = 0;
    const uint32_t* eb = ebuf + L.fb_start;
    const uint32_t* lc = reinterpret_cast<const uint32_t*>(s_coef + p.lane_off) + lane;      // coefficient PAIR m of lane l at word m * 64 + l: conflict-free
    for (int j0 = 0; j0 < p.nm; j0 += 4) {
      const uint32_t c01 = lc[(j0 >> 1) * 64], c23 = lc[((j0 >> 1) + 1) * 64];
      const uint32_t e0 = eb[j0], e1 = eb[j0 + 1], e2 = eb[j0 + 2], e3 = eb[j0 + 3];
      asm("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(alo) : "v"(e0), "v"(c01));
      asm("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(ahi) : "v"(e0), "v"(c01));
      asm("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(alo) : "v"(e1), "v"(c01));
      asm("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,1,0,0]" : "+v"(ahi) : "v"(e1), "v"(c01));
      asm("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(alo) : "v"(e2), "v"(c23));
      asm("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(ahi) : "v"(e2), "v"(c23));
      asm("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(alo) : "v"(e3), "v"(c23));
      asm("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,1,0,0]" : "+v"(ahi) : "v"(e3), "v"(c23));
    }
    uint64_t acc = (uint64_t)alo + ((uint64_t)ahi << 16);
    if (L.fb_ch >= 0) { fftbuf[2 * L.fb_ch] = (uint32_t)acc; fftbuf[2 * L.fb_ch + 1] = (uint32_t)(acc >> 32); }   // helper: FFT state is dead
    wave_lds_sync();
    if (lane < p.num_channels) {
      if (L.fb_helped) acc += (uint64_t)fftbuf[2 * lane] | ((uint64_t)fftbuf[2 * lane + 1] << 32);
      sig_out[lane] = sqrt48_round(acc) >> shift;
    }
  } else {
    uint64_t acc = 0;
    for (int j = 0; j < L.fb_len; ++j) {
      // upstream multiplies (uint64_t)(int32 energy): sign-extends the one value 2^31
      const uint64_t e = (uint64_t)(int64_t)(int32_t)ebuf[L.fb_start + j];
      acc += (uint64_t)(int64_t)s_coef[L.fb_off + j] * e;
    }
    if (L.fb_ch >= 0) { fftbuf[2 * L.fb_ch] = (uint32_t)acc; fftbuf[2 * L.fb_ch + 1] = (uint32_t)(acc >> 32); }   // helper: FFT state is dead
    wave_lds_sync();
    if (lane < p.num_channels) {
      if (L.fb_helped) acc += (uint64_t)fftbuf[2 * lane] | ((uint64_t)fftbuf[2 * lane + 1] << 32);
      sig_out[lane] = sqrt64_round(acc) >> shift;
    }
  }
}

template <typename T> struct AudioLoad;
template <> struct AudioLoad<float> {
  // input_data.py:23: audio * 32768 -> int16 (truncation toward zero; saturating)
  static __device__ __forceinline__ int cvt(float a) {
    const float v = fminf(fmaxf(a * 32768.0f, -32768.0f), 32767.0f);
    return (int)v;
  }
  static __device__ __forceinline__ void pair(const float* p, bool aligned, int& a, int& b) {
    if (aligned) { const float2 v = *reinterpret_cast<const float2*>(p); a = cvt(v.x); b = cvt(v.y); }
    else { a = cvt(p[0]); b = cvt(p[1]); }
  }
  static __device__ __forceinline__ int one(const float* p) { return cvt(*p); }
};
template <> struct AudioLoad<int16_t> {
  static __device__ __forceinline__ void pair(const int16_t* p, bool aligned, int& a, int& b) {
    if (aligned) { const uint32_t v = *reinterpret_cast<const uint32_t*>(p); a = sext16((int)v); b = ((int)v) >> 16; }
    else { a = p[0]; b = p[1]; }
  }
  static __device__ __forceinline__ int one(const int16_t* p) { return *p; }
};

// Loads this lane's 8 samples of the frame starting at `frame` (window_size valid samples).
template <typename T>
__device__ __forceinline__ void load_frame(const T* frame, int window_size, bool aligned, const LaneConst& L, int (&x)[8]) {
  // branch-free: every lane issues its 4 pair loads (clamped into the window), the zero padding of the
  // 512-point FFT frame is applied by selects (window_size >= 2; host sets `aligned` only for even windows)
  (void)window_size;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int a, b;
    AudioLoad<T>::pair(frame + L.toff[j], aligned, a, b);
    x[2 * j] = (L.tsel[j] == 2) ? a : (L.tsel[j] == 1 ? b : 0);
    x[2 * j + 1] = (L.tsel[j] == 2) ? b : 0;
  }
}

// noise_reduction.c (given the already-scanned estimate) + pcan_gain_control.c + log_scale.c for one
// (frame, channel) element.
__device__ __forceinline__ uint32_t finish_element(const FrontendParams& p, uint32_t s, uint32_t est,
                                                   const int16_t* s_pcan, const uint16_t* s_log) {
  const uint32_t su = s << p.smoothing_bits;
  const uint32_t e = est < su ? est : su;
  const uint32_t floor_ = (uint32_t)(((uint64_t)s * p.min_signal_remaining) >> 14);
  const uint32_t sub = (su - e) >> p.smoothing_bits;
  uint32_t v = sub > floor_ ? sub : floor_;
  if (p.enable_pcan) {
    const uint32_t gain = (uint32_t)wide_dynamic(est, s_pcan);
    const uint32_t snr = (uint32_t)(((uint64_t)v * gain) >> p.snr_shift);
    v = (snr < 8192u) ? ((snr * snr) >> 20) : ((snr >> 6) - 64u);
  }
  if (p.enable_log) {
    v = (p.correction_bits < 0) ? (v >> (-p.correction_bits)) : (v << p.correction_bits);
    v = (v > 1) ? log_scale(v, p.scale_shift, s_log) : 0;
  }
  return v < 0xFFFFu ? v : 0xFFFFu;
}

// ------------------------------------------------------------------------------------------------
// Fused kernel: one workgroup per clip.
// dynamic LDS: [NWAVES][512] u32 (fft+energy) | sig [F*C] u32 | est [F*C] u32 | coef i16 | pcan i16[128] | log u16[132]
template <typename T, int NWAVES>
__device__ __forceinline__ void clip_body(const FrontendParams& p, const T* __restrict__ audio, int n_samples, int num_frames, int aligned,
                                          float* __restrict__ spec, uint16_t* __restrict__ raw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int C = p.num_channels;
  const int FC = num_frames * C;
  uint32_t* s_fft = reinterpret_cast<uint32_t*>(smem);
  uint32_t* s_sig = s_fft + NWAVES * 512;
  uint32_t* s_est = s_sig + FC;
  int16_t* s_coef = reinterpret_cast<int16_t*>(s_est + FC);
  int16_t* s_pcan = s_coef + ((p.ncoef + 7) & ~7);
  uint16_t* s_log = reinterpret_cast<uint16_t*>(s_pcan + 128);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t clip = blockIdx.x;

  for (int i = tid; i < p.ncoef; i += NWAVES * 64) s_coef[i] = p.out_coef[i];
  for (int i = tid; i < 128; i += NWAVES * 64) s_pcan[i] = p.pcan_lut[i];
  for (int i = tid; i < 132; i += NWAVES * 64) s_log[i] = p.log_lut[i];
  LaneConst L;
  init_lane_const(p, lane, L);
  __syncthreads();

  const T* clip_audio = audio + clip * (size_t)n_samples;
  uint32_t* fftbuf = s_fft + wave * 512;
  uint32_t* ebuf = fftbuf + 256;
  int x[8];
  int f = wave;
  if (f < num_frames) load_frame<T>(clip_audio + (size_t)f * p.window_step, p.window_size, aligned != 0, L, x);
  for (; f < num_frames; f += NWAVES) {
    int xn[8];
    const int fn = f + NWAVES;
    if (fn < num_frames) load_frame<T>(clip_audio + (size_t)fn * p.window_step, p.window_size, aligned != 0, L, xn);
    frame_to_sig(p, L, lane, x, fftbuf, ebuf, s_coef, s_sig + f * C);
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = xn[q];
  }
  __syncthreads();
  // noise-estimate recurrence over frames (noise_reduction.c), 40 lanes of wave 0
  if (wave == 0 && lane < C) {
    const uint32_t sm = (lane & 1) ? p.odd_smoothing : p.even_smoothing;
    const uint32_t om = (1u << 14) - sm;
    uint32_t est = 0;
    for (int t = 0; t < num_frames; ++t) {
      const uint32_t su = s_sig[t * C + lane] << p.smoothing_bits;
      est = (uint32_t)((((uint64_t)su * sm) + ((uint64_t)est * om)) >> 14);
      s_est[t * C + lane] = est;
    }
  }
  __syncthreads();
  const float scale = 10.0f / 256.0f;
  for (int i = tid; i < FC; i += NWAVES * 64) {
    const uint32_t v = finish_element(p, s_sig[i], s_est[i], s_pcan, s_log);
    if (spec) spec[clip * FC + i] = (float)v * scale;
    if (raw) raw[clip * FC + i] = (uint16_t)v;
  }
}

template <typename T, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void frontend_clip_kernel(FrontendParams p, const T* __restrict__ audio, int n_samples, int num_frames, int aligned,
                                                                      float* __restrict__ spec, uint16_t* __restrict__ raw) {
  clip_body<T, NWAVES>(p, audio, n_samples, num_frames, aligned, spec, raw);
}
// The same body compiled for EIGHT waves per SIMD (64 VGPRs: hipcc keeps ~20 loop-invariant lane constants in scratch and reloads them
// once per frame -- 20 L1 hits against a ~10 000-cycle frame), so that four clips with eight waves each share a CU.
template <typename T, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void frontend_clip_kernel_w8(FrontendParams p, const T* __restrict__ audio,
                                                                                                                 int n_samples, int num_frames, int aligned,
                                                                                                                 float* __restrict__ spec,
                                                                                                                 uint16_t* __restrict__ raw) {
  clip_body<T, NWAVES>(p, audio, n_samples, num_frames, aligned, spec, raw);
}

// ------------------------------------------------------------------------------------------------
// Streaming split (batch_streaming_analysis.py:99-117): frame-level work once per hop, then one
// scan per window.
template <typename T>
__global__ __launch_bounds__(256) void frontend_frames_kernel(FrontendParams p, const T* __restrict__ audio, int total_frames,
                                                              int aligned, uint32_t* __restrict__ sig /*[total_frames, C]*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* s_fft = reinterpret_cast<uint32_t*>(smem);
  int16_t* s_coef = reinterpret_cast<int16_t*>(s_fft + 4 * 512);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < p.ncoef; i += 256) s_coef[i] = p.out_coef[i];
  LaneConst L;
  init_lane_const(p, lane, L);
  __syncthreads();
  uint32_t* fftbuf = s_fft + wave * 512;
  for (int f = blockIdx.x * 4 + wave; f < total_frames; f += gridDim.x * 4) {
    int x[8];
    load_frame<T>(audio + (size_t)f * p.window_step, p.window_size, aligned != 0, L, x);
    frame_to_sig(p, L, lane, x, fftbuf, fftbuf + 256, s_coef, sig + (size_t)f * p.num_channels);
  }
}

// one workgroup (256 threads) per window: scan + finish over frames [w*hop_frames, +frames_per_window)
__global__ __launch_bounds__(256) void frontend_windows_kernel(FrontendParams p, const uint32_t* __restrict__ sig, int hop_frames,
                                                               int frames_per_window, float* __restrict__ spec,
                                                               uint16_t* __restrict__ raw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int C = p.num_channels, FC = frames_per_window * C;
  uint32_t* s_sig = reinterpret_cast<uint32_t*>(smem);
  uint32_t* s_est = s_sig + FC;
  int16_t* s_pcan = reinterpret_cast<int16_t*>(s_est + FC);
  uint16_t* s_log = reinterpret_cast<uint16_t*>(s_pcan + 128);
  const int tid = threadIdx.x;
  const size_t win = blockIdx.x;
  const uint32_t* src = sig + win * (size_t)hop_frames * C;
  for (int i = tid; i < FC; i += 256) s_sig[i] = src[i];
  for (int i = tid; i < 128; i += 256) s_pcan[i] = p.pcan_lut[i];
  for (int i = tid; i < 132; i += 256) s_log[i] = p.log_lut[i];
  __syncthreads();
  if (tid < C) {
    const uint32_t sm = (tid & 1) ? p.odd_smoothing : p.even_smoothing;
    const uint32_t om = (1u << 14) - sm;
    uint32_t est = 0;
    for (int t = 0; t < frames_per_window; ++t) {
      const uint32_t su = s_sig[t * C + tid] << p.smoothing_bits;
      est = (uint32_t)((((uint64_t)su * sm) + ((uint64_t)est * om)) >> 14);
      s_est[t * C + tid] = est;
    }
  }
  __syncthreads();
  const float scale = 10.0f / 256.0f;
  for (int i = tid; i < FC; i += 256) {
    const uint32_t v = finish_element(p, s_sig[i], s_est[i], s_pcan, s_log);
    if (spec) spec[win * FC + i] = (float)v * scale;
    if (raw) raw[win * FC + i] = (uint16_t)v;
  }
}

}  // namespace mkws

// ================================================================================================
// C-ABI
// ================================================================================================
using namespace mkws;

struct mkws_frontend {
  mkws_frontend_cfg cfg;
  FrontendTables tab;
  FrontendParams prm;
  void* d_blob = nullptr;
  uint32_t* d_stream_sig = nullptr;   // [max_frames, C] workspace for the streaming split
  int max_samples = 0;
  int max_frames = 0;
  int device = 0;
};

extern "C" {

int mkws_abi_version(void) { return MKWS_ABI_VERSION; }
const char* mkws_last_error(void) { return mkws::err_buf(); }
const char* mkws_build_arch(void) { return "gfx950"; }

void mkws_frontend_default_cfg(mkws_frontend_cfg* c) {
  if (!c) return;
  c->sample_rate = 16000; c->window_size_ms = 30; c->window_step_ms = 20; c->num_channels = 40;
  c->upper_band_limit = 7500.0f; c->lower_band_limit = 125.0f; c->smoothing_bits = 10;
  c->even_smoothing = 0.025f; c->odd_smoothing = 0.06f; c->min_signal_remaining = 0.05f;
  c->enable_pcan = 1; c->pcan_strength = 0.95f; c->pcan_offset = 80.0f; c->gain_bits = 21;
  c->enable_log = 1; c->scale_shift = 6;
}

int mkws_frontend_host_table(const mkws_frontend_cfg* cfg, int which, void* dst, size_t cap) {
  if (!cfg) return fail(MKWS_ERR_INVALID_ARG, "cfg is NULL");
  FrontendTables t;
  int rc = build_frontend_tables(*cfg, &t);
  if (rc != MKWS_OK) return rc;
  const void* src = nullptr;
  size_t n = 0;
  int32_t scalars[8] = {t.window_size, t.window_step, t.fft_size, t.start_index, t.end_index, t.num_weights, t.snr_shift, t.correction_bits};
  switch (which) {
    case MKWS_FT_WINDOW_COEF: src = t.window_coef.data(); n = t.window_coef.size() * 2; break;
    case MKWS_FT_TWIDDLES: src = t.twiddles.data(); n = t.twiddles.size() * 2; break;
    case MKWS_FT_SUPER_TWIDDLES: src = t.super_twiddles.data(); n = t.super_twiddles.size() * 2; break;
    case MKWS_FT_FB_WEIGHTS: src = t.fb_weights.data(); n = t.fb_weights.size() * 2; break;
    case MKWS_FT_FB_UNWEIGHTS: src = t.fb_unweights.data(); n = t.fb_unweights.size() * 2; break;
    case MKWS_FT_FB_FREQ_STARTS: src = t.fb_freq_starts.data(); n = t.fb_freq_starts.size() * 2; break;
    case MKWS_FT_FB_WEIGHT_STARTS: src = t.fb_weight_starts.data(); n = t.fb_weight_starts.size() * 2; break;
    case MKWS_FT_FB_WIDTHS: src = t.fb_widths.data(); n = t.fb_widths.size() * 2; break;
    case MKWS_FT_PCAN_LUT: src = t.pcan_lut.data(); n = t.pcan_lut.size() * 2; break;
    case MKWS_FT_LOG_LUT: src = t.log_lut.data(); n = t.log_lut.size() * 2; break;
    case MKWS_FT_SCALARS: src = scalars; n = sizeof(scalars); break;
    default: return fail(MKWS_ERR_INVALID_ARG, "unknown table id %d", which);
  }
  if (dst && cap > 0) memcpy(dst, src, n < cap ? n : cap);
  return (int)n;
}

int mkws_frontend_num_frames(const mkws_frontend_cfg* cfg, int n_samples) {
  if (!cfg || cfg->sample_rate <= 0) return fail(MKWS_ERR_INVALID_ARG, "cfg is NULL or invalid");
  const int size = cfg->window_size_ms * cfg->sample_rate / 1000;
  const int step = cfg->window_step_ms * cfg->sample_rate / 1000;
  if (size <= 0 || step <= 0) return fail(MKWS_ERR_INVALID_ARG, "window/step of zero samples");
  if (n_samples < size) return 0;
  return (n_samples - size) / step + 1;
}

int mkws_frontend_create(const mkws_frontend_cfg* cfg, int max_samples, mkws_frontend** out) {
  if (!cfg || !out) return fail(MKWS_ERR_INVALID_ARG, "cfg/out is NULL");
  *out = nullptr;
  if (max_samples <= 0) return fail(MKWS_ERR_INVALID_ARG, "max_samples must be positive");
  mkws_frontend* fe = new (std::nothrow) mkws_frontend();
  if (!fe) return fail(MKWS_ERR_ALLOC, "out of host memory");
  fe->cfg = *cfg;
  int rc = build_frontend_tables(*cfg, &fe->tab);
  if (rc != MKWS_OK) { delete fe; return rc; }
  const FrontendTables& t = fe->tab;
  if (t.fft_size != 512) {
    delete fe;
    return fail(MKWS_ERR_UNSUPPORTED, "HIP frontend implements the 512-point FFT (window of 257..512 samples); got window %d -> fft %d",
                t.window_size, t.fft_size);
  }
  if (cfg->num_channels > 64) { delete fe; return fail(MKWS_ERR_UNSUPPORTED, "num_channels %d > 64", cfg->num_channels); }
  rc = require_device();
  if (rc != MKWS_OK) { delete fe; return rc; }
  (void)hipGetDevice(&fe->device);
  // pack one device blob
  const int C = cfg->num_channels;
  // device coefficient array = the compact lists (padded to a multiple of 8) + the per-LANE lists zero padded to the longest one (nm),
  // which the 16-bit filterbank path walks with a uniform trip count; both are staged into LDS together
  const size_t ncoef_compact = (t.out_coef.size() + 7) & ~size_t(7);
  int lane_len[64], lane_src[64];
  for (int l = 0; l < 64; ++l) { lane_len[l] = 0; lane_src[l] = 0; }
  {
    for (int c = 0; c < C; ++c) { lane_len[c] = t.out_len[c]; lane_src[c] = t.out_off[c]; }
    std::vector<int> order(C);
    for (int c = 0; c < C; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return t.out_len[a] > t.out_len[b]; });
    for (int k = 0, l = C; k < C && l < 64; ++k, ++l) {          // the same split as the task table below
      const int c = order[k], len = t.out_len[c];
      if (len < 2) break;
      const int first = (len + 1) / 2;
      lane_len[c] = first; lane_len[l] = len - first; lane_src[l] = t.out_off[c] + first;
    }
  }
  int nm = 4;
  for (int l = 0; l < 64; ++l) nm = std::max(nm, (lane_len[l] + 3) & ~3);
  const size_t ncoef = ncoef_compact + (size_t)64 * nm;
  auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
  size_t o_win = 0, o_tw = al(o_win + 512 * 2), o_stw = al(o_tw + 256 * 4), o_os = al(o_stw + 128 * 4), o_ol = al(o_os + 64 * 2),
         o_oo = al(o_ol + 64 * 2), o_tc = al(o_oo + 64 * 2), o_th = al(o_tc + 64 * 2), o_oc = al(o_th + 64 * 2), o_pc = al(o_oc + (ncoef + 8) * 2), o_lg = al(o_pc + 128 * 2), total = al(o_lg + 132 * 2);
  std::vector<unsigned char> h(total, 0);
  memcpy(h.data() + o_win, t.window_coef.data(), t.window_coef.size() * 2);
  uint32_t* tw = reinterpret_cast<uint32_t*>(h.data() + o_tw);
  for (int i = 0; i < 256; ++i) tw[i] = ((uint32_t)(uint16_t)t.twiddles[2 * i]) | ((uint32_t)(uint16_t)t.twiddles[2 * i + 1] << 16);
  uint32_t* stw = reinterpret_cast<uint32_t*>(h.data() + o_stw);
  for (int i = 0; i < 128; ++i) stw[i] = ((uint32_t)(uint16_t)t.super_twiddles[2 * i]) | ((uint32_t)(uint16_t)t.super_twiddles[2 * i + 1] << 16);
  {
    // per-lane filterbank tasks: lanes < C their channel; the 64 - C spare lanes take the second half of the longest tap lists
    int16_t* ts = reinterpret_cast<int16_t*>(h.data() + o_os);
    int16_t* tl = reinterpret_cast<int16_t*>(h.data() + o_ol);
    int16_t* to = reinterpret_cast<int16_t*>(h.data() + o_oo);
    int16_t* tc = reinterpret_cast<int16_t*>(h.data() + o_tc);
    int16_t* th = reinterpret_cast<int16_t*>(h.data() + o_th);
    for (int l = 0; l < 64; ++l) { ts[l] = 0; tl[l] = 0; to[l] = 0; tc[l] = -1; th[l] = 0; }
    for (int c = 0; c < C; ++c) { ts[c] = t.out_start[c]; tl[c] = t.out_len[c]; to[c] = t.out_off[c]; }
    std::vector<int> order(C);
    for (int c = 0; c < C; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return t.out_len[a] > t.out_len[b]; });
    for (int k = 0, l = C; k < C && l < 64; ++k, ++l) {
      const int c = order[k], len = t.out_len[c];
      if (len < 2) break;
      const int first = (len + 1) / 2;
      tl[c] = static_cast<int16_t>(first);
      ts[l] = static_cast<int16_t>(t.out_start[c] + first); tl[l] = static_cast<int16_t>(len - first); to[l] = static_cast<int16_t>(t.out_off[c] + first);
      tc[l] = static_cast<int16_t>(c); th[c] = 1;
    }
  }
  memcpy(h.data() + o_oc, t.out_coef.data(), t.out_coef.size() * 2);
  {
    int16_t* lt = reinterpret_cast<int16_t*>(h.data() + o_oc) + ncoef_compact;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < lane_len[l]; ++j) lt[((size_t)(j >> 1) * 64 + l) * 2 + (j & 1)] = t.out_coef[lane_src[l] + j];      // pair-major, lane-minor
  }
  memcpy(h.data() + o_pc, t.pcan_lut.data(), t.pcan_lut.size() * 2);
  memcpy(h.data() + o_lg, t.log_lut.data(), t.log_lut.size() * 2);
  if (hipMalloc(&fe->d_blob, total) != hipSuccess) { delete fe; return fail(MKWS_ERR_ALLOC, "hipMalloc(%zu) failed", total); }
  if (hipMemcpy(fe->d_blob, h.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(fe->d_blob); delete fe; return fail(MKWS_ERR_HIP, "table upload failed");
  }
  unsigned char* d = static_cast<unsigned char*>(fe->d_blob);
  FrontendParams& p = fe->prm;
  p.window_coef = reinterpret_cast<int16_t*>(d + o_win);
  p.tw = reinterpret_cast<uint32_t*>(d + o_tw);
  p.stw = reinterpret_cast<uint32_t*>(d + o_stw);
  p.out_start = reinterpret_cast<int16_t*>(d + o_os);
  p.out_len = reinterpret_cast<int16_t*>(d + o_ol);
  p.out_off = reinterpret_cast<int16_t*>(d + o_oo);
  p.out_coef = reinterpret_cast<int16_t*>(d + o_oc);
  p.task_ch = reinterpret_cast<int16_t*>(d + o_tc);
  p.task_helped = reinterpret_cast<int16_t*>(d + o_th);
  p.pcan_lut = reinterpret_cast<int16_t*>(d + o_pc);
  p.log_lut = reinterpret_cast<uint16_t*>(d + o_lg);
  p.ncoef = (int)ncoef;
  {
    // fast48: all weights non-negative and every channel's list (both lane halves together) adds up to <= 2^16, so that a mel sum of
    // energies < 2^31 stays below 2^47 and a lane's 16-bit partial sums below 2^32
    bool ok48 = true;
    for (int c = 0; c < C && ok48; ++c) {
      long sum = 0;
      for (int j = 0; j < t.out_len[c]; ++j) { const int w = t.out_coef[t.out_off[c] + j]; if (w < 0) ok48 = false; sum += w; }
      if (sum > 65536) ok48 = false;
    }
    p.fast48 = (ok48 && nm <= 32) ? 1 : 0;
    p.lane_off = (int)ncoef_compact; p.nm = nm;
  }
  p.window_size = t.window_size; p.window_step = t.window_step; p.num_channels = C;
  p.smoothing_bits = cfg->smoothing_bits; p.enable_pcan = cfg->enable_pcan ? 1 : 0; p.enable_log = cfg->enable_log ? 1 : 0;
  p.scale_shift = cfg->scale_shift; p.snr_shift = t.snr_shift; p.correction_bits = t.correction_bits;
  p.even_smoothing = t.even_smoothing; p.odd_smoothing = t.odd_smoothing; p.min_signal_remaining = t.min_signal_remaining;
  fe->max_samples = max_samples;
  fe->max_frames = mkws_frontend_num_frames(cfg, max_samples);
  *out = fe;
  return MKWS_OK;
}

void mkws_frontend_destroy(mkws_frontend* fe) {
  if (!fe) return;
  if (fe->d_blob) (void)hipFree(fe->d_blob);
  if (fe->d_stream_sig) (void)hipFree(fe->d_stream_sig);
  delete fe;
}

}  // extern "C"

namespace {

// Waves per clip: four.  A clip's frames are independent until the scan and one wave works on one frame, so more waves per clip look like
// free parallelism -- measured in round 4 (tools/gpu/r4_fe.sh, 1024 clips, same call): 60.0 us with four waves per clip; 66.4 with five
// (90 VGPRs allow five per SIMD, but a 5-wave workgroup does not spread evenly over four SIMDs); 70.6 with eight at eight waves per SIMD
// (frontend_clip_kernel_w8: 64 VGPRs, 20 scratch reloads per frame); 83.8 with ten.  MKWS_FRONTEND_WAVES=5 / 8 / 10 keeps the experiment
// runnable; results do not depend on it.
int clip_waves(int B) {
  (void)B;
  static const int forced = [] { const char* e = getenv("MKWS_FRONTEND_WAVES"); return e ? atoi(e) : 0; }();
  return (forced == 5 || forced == 8 || forced == 10) ? forced : 4;
}

size_t clip_lds_bytes(const mkws_frontend* fe, int frames, int waves) {
  const size_t FC = (size_t)frames * fe->prm.num_channels;
  return (size_t)waves * 512 * 4 + 2 * FC * 4 + ((fe->prm.ncoef + 7) & ~7) * 2 + 128 * 2 + 132 * 2 + 16;
}

template <typename T>
int frontend_forward_impl(mkws_frontend* fe, const T* d_audio, int B, int n_samples, float* d_spec, uint16_t* d_raw, void* stream) {
  if (!fe) return fail(MKWS_ERR_INVALID_ARG, "frontend handle is NULL");
  if (B < 0 || n_samples < 0) return fail(MKWS_ERR_INVALID_ARG, "negative batch or sample count");
  if (n_samples > fe->max_samples) return fail(MKWS_ERR_INVALID_ARG, "n_samples %d exceeds max_samples %d given at create", n_samples, fe->max_samples);
  const int frames = mkws_frontend_num_frames(&fe->cfg, n_samples);
  if (B == 0 || frames == 0) return MKWS_OK;   // empty input -> empty output, like the op
  if (!d_spec && !d_raw) return fail(MKWS_ERR_INVALID_ARG, "both outputs are NULL");
  if (!d_audio) return fail(MKWS_ERR_INVALID_ARG, "d_audio is NULL");
  int waves = clip_waves(B);
  if (clip_lds_bytes(fe, frames, waves) > 64 * 1024) waves = 4;
  const size_t lds = clip_lds_bytes(fe, frames, waves);
  if (lds > 64 * 1024)
    return fail(MKWS_ERR_UNSUPPORTED, "%d frames per clip need %zu B of LDS; use mkws_frontend_stream_f32 for long audio", frames, lds);
  const int aligned = ((n_samples % 2) == 0 && (fe->prm.window_step % 2) == 0 && (fe->prm.window_size % 2) == 0 &&
                       (reinterpret_cast<uintptr_t>(d_audio) % (2 * sizeof(T))) == 0) ? 1 : 0;
#define MKWS_FE_CLIP(W_) hipLaunchKernelGGL((frontend_clip_kernel<T, W_>), dim3(B), dim3(W_ * 64), lds, static_cast<hipStream_t>(stream), fe->prm, d_audio, n_samples, \
                                            frames, aligned, d_spec, d_raw)
  if (waves == 10) MKWS_FE_CLIP(10);
  else if (waves == 8)
    hipLaunchKernelGGL((frontend_clip_kernel_w8<T, 8>), dim3(B), dim3(8 * 64), lds, static_cast<hipStream_t>(stream), fe->prm, d_audio, n_samples, frames, aligned,
                       d_spec, d_raw);
  else if (waves == 5) MKWS_FE_CLIP(5);
  else MKWS_FE_CLIP(4);
#undef MKWS_FE_CLIP
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

}  // namespace

extern "C" {

int mkws_frontend_forward_f32(mkws_frontend* fe, const float* d_audio, int B, int n_samples, float* d_spec, uint16_t* d_raw, void* stream) {
  return frontend_forward_impl<float>(fe, d_audio, B, n_samples, d_spec, d_raw, stream);
}

int mkws_frontend_forward_i16(mkws_frontend* fe, const int16_t* d_audio, int B, int n_samples, float* d_spec, uint16_t* d_raw, void* stream) {
  return frontend_forward_impl<int16_t>(fe, d_audio, B, n_samples, d_spec, d_raw, stream);
}

int mkws_frontend_stream_f32(mkws_frontend* fe, const float* d_audio, int n_samples, int window_samples, int hop_samples,
                             float* d_spec, uint16_t* d_raw, int max_windows, void* stream) {
  if (!fe) return fail(MKWS_ERR_INVALID_ARG, "frontend handle is NULL");
  if (n_samples < 0 || window_samples <= 0 || hop_samples <= 0) return fail(MKWS_ERR_INVALID_ARG, "bad sample counts");
  if (n_samples > fe->max_samples) return fail(MKWS_ERR_INVALID_ARG, "n_samples %d exceeds max_samples %d", n_samples, fe->max_samples);
  const FrontendParams& p = fe->prm;
  if (hop_samples % p.window_step != 0)
    return fail(MKWS_ERR_UNSUPPORTED, "hop of %d samples is not a multiple of the %d-sample frame step (frame sharing needs that)", hop_samples, p.window_step);
  const int fpw = mkws_frontend_num_frames(&fe->cfg, window_samples);
  if (fpw <= 0 || n_samples < window_samples) return 0;
  const int num_windows = 1 + (n_samples - window_samples) / hop_samples;
  if (num_windows > max_windows) return fail(MKWS_ERR_INVALID_ARG, "%d windows exceed max_windows %d", num_windows, max_windows);
  if (!d_audio || (!d_spec && !d_raw)) return fail(MKWS_ERR_INVALID_ARG, "NULL buffer");
  const int hop_frames = hop_samples / p.window_step;
  const int total_frames = (num_windows - 1) * hop_frames + fpw;
  const size_t lds2 = 2 * (size_t)fpw * p.num_channels * 4 + 128 * 2 + 132 * 2 + 16;
  if (lds2 > 64 * 1024) return fail(MKWS_ERR_UNSUPPORTED, "window of %d frames needs %zu B LDS", fpw, lds2);
  if (!fe->d_stream_sig) {
    const size_t n = (size_t)(fe->max_frames > 0 ? fe->max_frames : 1) * p.num_channels * 4;
    if (hipMalloc(reinterpret_cast<void**>(&fe->d_stream_sig), n) != hipSuccess) return fail(MKWS_ERR_ALLOC, "hipMalloc(%zu) failed", n);
  }
  const int aligned = ((p.window_step % 2) == 0 && (p.window_size % 2) == 0 && (reinterpret_cast<uintptr_t>(d_audio) % 8) == 0) ? 1 : 0;
  const size_t lds1 = 4 * 512 * 4 + ((p.ncoef + 7) & ~7) * 2 + 16;
  int grid1 = (total_frames + 3) / 4;
  if (grid1 > 4096) grid1 = 4096;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL((frontend_frames_kernel<float>), dim3(grid1), dim3(256), lds1, s, p, d_audio, total_frames, aligned, fe->d_stream_sig);
  MKWS_HIP(hipGetLastError());
  hipLaunchKernelGGL(frontend_windows_kernel, dim3(num_windows), dim3(256), lds2, s, p, fe->d_stream_sig, hop_frames, fpw, d_spec, d_raw);
  MKWS_HIP(hipGetLastError());
  return num_windows;
}

}  // extern "C"
