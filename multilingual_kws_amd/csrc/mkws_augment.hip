// mkws_augment.hip -- training-batch assembly on the device.
//
// Replaces the per-clip tf.data map of AudioDataset.augment / random_timeshift /
// random_background_sample / add_background (multilingual_kws/embedding/input_data.py:141-157,
// 227-304) and spec_augment (:306-369).  The random DRAWS stay on the host (a few scalars per clip,
// numpy Generator; the reference uses tf.random.Generator, so only the distributions can match);
// the sample-level work -- gathers from the resident waveform banks, zero-filled time shifts, RMS-
// matched background mixing with clipping, SpecAugment masking -- runs here, one workgroup per clip.
#include "mkws_common.h"

namespace mkws {

struct AugItem {          // == mkws_augment_item
  int32_t mode;           // 0 keep(shift) | 1 silence(bg*vol) | 2 mix(shifted fg + rms-matched bg*vol, clipped)
  int32_t bank;           // which waveform bank the foreground comes from (0 = targets, 1 = unknown)
  int32_t src;            // row in that bank
  int32_t shift;          // out[t] = src[t - shift], zero outside
  int32_t bg_idx, bg_off; // background track and offset of the n-sample slice
  float bg_vol;
  int32_t reserved;
};

__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// One workgroup per clip; thread t owns samples t, t + 256, ...  Each pass walks them EIGHT at a time: the eight (guarded) loads of a
// thread are issued together and consumed in order -- the sums keep the sequential order of a one-sample-at-a-time loop, bit for bit, but a
// thread pays one memory latency per eight samples instead of one per sample (round 5: the plain loops made this launch ~60 us per 1 024 clips,
// more than the micro-frontend that consumes its output).
constexpr int kAugUnroll = 8;

__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ bank0, const float* __restrict__ bank1,
                                                      const float* __restrict__ bg, long bg_stride, const AugItem* __restrict__ items,
                                                      int n, float* __restrict__ out) {
  __shared__ float s_red[4];
  const AugItem it = items[blockIdx.x];
  float* o = out + (size_t)blockIdx.x * n;
  const float* src = (it.bank == 0 ? bank0 : bank1) + (size_t)it.src * n;
  const float* b = bg ? bg + (size_t)it.bg_idx * bg_stride + it.bg_off : nullptr;
  const int shift = it.shift;
  // foreground sample of output position t (zero-filled time shift) / background sample, 0 past the clip
  auto fg_at = [&](int t) { const int s = t - shift; return (t < n && s >= 0 && s < n) ? src[s] : 0.0f; };
  auto bg_at = [&](int t) { return t < n ? b[t] : 0.0f; };
  if (it.mode == 1) {          // random_background_sample(background_volume)
    for (int t0 = threadIdx.x; t0 < n; t0 += 256 * kAugUnroll) {
      float g[kAugUnroll];
#pragma unroll
      for (int u = 0; u < kAugUnroll; ++u) g[u] = bg_at(t0 + 256 * u);
#pragma unroll
      for (int u = 0; u < kAugUnroll; ++u)
        if (t0 + 256 * u < n) o[t0 + 256 * u] = g[u] * it.bg_vol;
    }
    return;
  }
  if (it.mode == 0) {          // random_timeshift
    for (int t0 = threadIdx.x; t0 < n; t0 += 256 * kAugUnroll) {
      float f[kAugUnroll];
#pragma unroll
      for (int u = 0; u < kAugUnroll; ++u) f[u] = fg_at(t0 + 256 * u);
#pragma unroll
      for (int u = 0; u < kAugUnroll; ++u)
        if (t0 + 256 * u < n) o[t0 + 256 * u] = f[u];
    }
    return;
  }
  // add_background(shifted foreground, background slice, volume)
  float sf = 0.0f, sb = 0.0f;
  for (int t0 = threadIdx.x; t0 < n; t0 += 256 * kAugUnroll) {
    float f[kAugUnroll], g[kAugUnroll];
#pragma unroll
    for (int u = 0; u < kAugUnroll; ++u) { f[u] = fg_at(t0 + 256 * u); g[u] = bg_at(t0 + 256 * u); }
#pragma unroll
    for (int u = 0; u < kAugUnroll; ++u) { sf += f[u] * f[u]; sb += g[u] * g[u]; }      // (positions past the clip add +0)
  }
  const float fg_rms = sqrtf(block_sum(sf, s_red) / (float)n);
  const float bg_rms = sqrtf(block_sum(sb, s_red) / (float)n);
  const float snr = bg_rms > 0.0f ? fg_rms / bg_rms : 0.0f;
  for (int t0 = threadIdx.x; t0 < n; t0 += 256 * kAugUnroll) {
    float f[kAugUnroll], g[kAugUnroll];
#pragma unroll
    for (int u = 0; u < kAugUnroll; ++u) { f[u] = fg_at(t0 + 256 * u); g[u] = bg_at(t0 + 256 * u); }
#pragma unroll
    for (int u = 0; u < kAugUnroll; ++u) {
      const float v = (g[u] * snr) * it.bg_vol + f[u];
      if (t0 + 256 * u < n) o[t0 + 256 * u] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
  }
}

// masks [B,8] = {f0 start, f0 size, f1 start, f1 size, t0 start, t0 size, t1 start, t1 size}; size 0 = unused
__global__ __launch_bounds__(256) void specaug_kernel(float* __restrict__ spec, const int32_t* __restrict__ masks, int F, int C) {
  const int32_t* m = masks + (size_t)blockIdx.x * 8;
  float* s = spec + (size_t)blockIdx.x * F * C;
  for (int i = threadIdx.x; i < F * C; i += 256) {
    const int f = i / C, c = i % C;
    const bool z = (c >= m[0] && c < m[0] + m[1]) || (c >= m[2] && c < m[2] + m[3]) ||
                   (f >= m[4] && f < m[4] + m[5]) || (f >= m[6] && f < m[6] + m[7]);
    if (z) s[i] = 0.0f;
  }
}

// any number of masks per axis: masks [B][2*(NF+NT)] = NF x {channel start, size} then NT x {frame start, size}
__global__ __launch_bounds__(256) void specaug_n_kernel(float* __restrict__ spec, const int32_t* __restrict__ masks, int NF, int NT, int F, int C) {
  const int32_t* m = masks + (size_t)blockIdx.x * 2 * (NF + NT);
  float* s = spec + (size_t)blockIdx.x * F * C;
  for (int i = threadIdx.x; i < F * C; i += 256) {
    const int f = i / C, c = i % C;
    bool z = false;
    for (int k = 0; k < NF; ++k) z |= (c >= m[2 * k] && c < m[2 * k] + m[2 * k + 1]);
    for (int k = 0; k < NT; ++k) z |= (f >= m[2 * (NF + k)] && f < m[2 * (NF + k)] + m[2 * (NF + k) + 1]);
    if (z) s[i] = 0.0f;
  }
}

}  // namespace mkws

using namespace mkws;

extern "C" {

int mkws_augment_batch(const float* d_bank0, const float* d_bank1, const float* d_bg, int64_t bg_stride,
                       const mkws_augment_item* d_items, int B, int n_samples, float* d_out, void* stream) {
  static_assert(sizeof(AugItem) == sizeof(mkws_augment_item), "item layout");
  if (B < 0 || n_samples <= 0) return fail(MKWS_ERR_INVALID_ARG, "bad batch/sample count");
  if (B == 0) return MKWS_OK;
  if (!d_bank0 || !d_items || !d_out) return fail(MKWS_ERR_INVALID_ARG, "NULL buffer");
  hipLaunchKernelGGL(augment_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), d_bank0, d_bank1 ? d_bank1 : d_bank0, d_bg,
                     (long)bg_stride, reinterpret_cast<const AugItem*>(d_items), n_samples, d_out);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_specaug_apply(float* d_spec, const int32_t* d_masks, int B, int frames, int channels, void* stream) {
  if (B < 0 || frames <= 0 || channels <= 0) return fail(MKWS_ERR_INVALID_ARG, "bad shape");
  if (B == 0) return MKWS_OK;
  if (!d_spec || !d_masks) return fail(MKWS_ERR_INVALID_ARG, "NULL buffer");
  hipLaunchKernelGGL(specaug_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), d_spec, d_masks, frames, channels);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

int mkws_specaug_apply_n(float* d_spec, const int32_t* d_masks, int n_freq, int n_time, int B, int frames, int channels, void* stream) {
  if (B < 0 || frames <= 0 || channels <= 0 || n_freq < 0 || n_time < 0) return fail(MKWS_ERR_INVALID_ARG, "bad shape / mask counts");
  if (B == 0 || n_freq + n_time == 0) return MKWS_OK;
  if (!d_spec || !d_masks) return fail(MKWS_ERR_INVALID_ARG, "NULL buffer");
  hipLaunchKernelGGL(specaug_n_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), d_spec, d_masks, n_freq, n_time, frames, channels);
  MKWS_HIP(hipGetLastError());
  return MKWS_OK;
}

}  // extern "C"
