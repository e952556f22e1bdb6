// Device-side helpers shared by the embedding kernels' translation units (mkws_embed.hip, mkws_embed_rows.hip):
// vector types, the hardware-transcendental sigmoid / swish, the buffer-descriptor weight stream, the argument block of the
// whole-block kernels.  Include AFTER `#pragma clang fp contract(fast)` (the epilogue arithmetic may fuse in both files alike).
#pragma once
#include <hip/hip_runtime.h>

namespace mkws {

using f32x4 = __attribute__((ext_vector_type(4))) float;

enum Act { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2, ACT_SELU = 3, ACT_SIGMOID = 4 };

// sigmoid on the hardware transcendentals: v_exp_f32 + v_rcp_f32 (1 ulp each) -- 4 instructions, vs ~20 for
// expf + IEEE division; swish sits in every epilogue and was the largest VALU cost of the fused kernels.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// Four at once, written on the vector so that the multiply / add around the two transcendentals become packed instructions
// (v_pk_mul_f32 / v_pk_add_f32: half the VALU issue of the scalar form; MFMA and VALU work serialize on a SIMD,
// tools/microbench/mfma_valu_overlap.hip, so every VALU instruction saved in an epilogue is MFMA time gained).  Same operations
// in the same order as sigmoidf_ / swishf_: bit-identical.
__device__ __forceinline__ f32x4 sigmoid4_(f32x4 v) {
  f32x4 t = v * -1.4426950408889634f;
  t.x = __builtin_amdgcn_exp2f(t.x); t.y = __builtin_amdgcn_exp2f(t.y); t.z = __builtin_amdgcn_exp2f(t.z); t.w = __builtin_amdgcn_exp2f(t.w);
  t = t + 1.0f;
  t.x = __builtin_amdgcn_rcpf(t.x); t.y = __builtin_amdgcn_rcpf(t.y); t.z = __builtin_amdgcn_rcpf(t.z); t.w = __builtin_amdgcn_rcpf(t.w);
  return t;
}
__device__ __forceinline__ f32x4 swish4_(f32x4 v) { return v * sigmoid4_(v); }
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 swish2_(f32x2 v) {          // the same operations per element as swish4_ / swishf_
  f32x2 t = v * -1.4426950408889634f;
  t.x = __builtin_amdgcn_exp2f(t.x); t.y = __builtin_amdgcn_exp2f(t.y);
  t = t + 1.0f;
  t.x = __builtin_amdgcn_rcpf(t.x); t.y = __builtin_amdgcn_rcpf(t.y);
  return v * t;
}
struct WBuf {
  __amdgpu_buffer_rsrc_t r; unsigned voff;
  __device__ __forceinline__ WBuf(const float* base, unsigned lane_off_floats)
      : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000)), voff(lane_off_floats * 4u) {}
  __device__ __forceinline__ f32x4 ld(size_t idx) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, (unsigned)(idx * 4u), 0));
  }
};

struct MidArgs {
  const float* X; int Cin;
  const float* WpE; const float* scE; const float* shE; int NTtotE;
  const float* Wd; const float* scD; const float* shD;
  const float* Wr; const float* br; const float* We; const float* be; int se;
  const float* WpP; const float* scP; const float* shP;
  float* Y; int Cout; int residual;
  float* dbg_dw; float* dbg_gate;
  int B;
#ifdef MKWS_FRONT_TIMING
  unsigned long long* dbg_t;
  int ablate;                      // mbconv_rows_kernel, timing build: MKWS_ABLATE bit mask (phases it skips: wrong results, honest timing of the rest)
#endif
};


}  // namespace mkws
