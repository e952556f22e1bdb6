// See mkws_frontend_tables.h.  Spec: SURVEY.md Appendix A.1 (window), A.3 (kissfft twiddles),
// A.4 (filterbank), A.6 (noise-reduction constants), A.7 (PCAN LUT), A.8 (log LUT).
#include "mkws_frontend_tables.h"

#include <cmath>

#include "mkws_common.h"

namespace mkws {
namespace {

constexpr int kWindowBits = 12;
constexpr int kFilterbankBits = 12;
constexpr int kNoiseReductionBits = 14;
constexpr int kPcanSnrBits = 12;
constexpr int kWideDynamicBits = 32;

int bit_length(uint32_t x) {
  int n = 0;
  while (x) { ++n; x >>= 1; }
  return n;
}

// A.4: float in, double log1p, rounded back to float -- as the upstream `float FreqToMel(float)`.
float mel_of(float hz) { return static_cast<float>(1127.0 * std::log1p(static_cast<double>(hz) / 700.0)); }

// A.3: kissfft's kf_cexp for FIXED_POINT=16.
void fixed_cexp(double phase, int16_t* re, int16_t* im) {
  *re = static_cast<int16_t>(std::floor(0.5 + 32767.0 * std::cos(phase)));
  *im = static_cast<int16_t>(std::floor(0.5 + 32767.0 * std::sin(phase)));
}

}  // namespace

int build_frontend_tables(const mkws_frontend_cfg& c, FrontendTables* t) {
  if (c.sample_rate <= 0 || c.window_size_ms <= 0 || c.window_step_ms <= 0 || c.num_channels <= 0)
    return fail(MKWS_ERR_INVALID_ARG, "frontend cfg: sample_rate/window/step/channels must be positive");
  if (!(c.lower_band_limit >= 0.0f) || !(c.upper_band_limit > c.lower_band_limit))
    return fail(MKWS_ERR_INVALID_ARG, "frontend cfg: need 0 <= lower_band_limit < upper_band_limit");
  if (c.smoothing_bits < 0 || c.smoothing_bits > 16 || c.gain_bits < 0 || c.gain_bits > 30 ||
      c.scale_shift < 0 || c.scale_shift > 16)
    return fail(MKWS_ERR_INVALID_ARG, "frontend cfg: smoothing_bits/gain_bits/scale_shift out of range");

  // ---- A.1 window ------------------------------------------------------------------------------
  t->window_size = c.window_size_ms * c.sample_rate / 1000;
  t->window_step = c.window_step_ms * c.sample_rate / 1000;
  if (t->window_size < 2 || t->window_step < 1)
    return fail(MKWS_ERR_INVALID_ARG, "frontend cfg: window of %d samples / step %d", t->window_size, t->window_step);
  t->window_coef.resize(t->window_size);
  {
    const float arg = static_cast<float>(M_PI * 2.0 / static_cast<float>(t->window_size));
    for (int i = 0; i < t->window_size; ++i) {
      const float v = static_cast<float>(0.5 - 0.5 * std::cos(static_cast<double>(arg) * (i + 0.5)));
      t->window_coef[i] = static_cast<int16_t>(std::floor(static_cast<double>(v * static_cast<float>(1 << kWindowBits)) + 0.5));
    }
  }

  // ---- A.3 FFT size and twiddles ------------------------------------------------------------------
  t->fft_size = 1;
  while (t->fft_size < t->window_size) t->fft_size <<= 1;
  t->ncfft = t->fft_size / 2;
  t->twiddles.resize(2 * t->ncfft);
  for (int i = 0; i < t->ncfft; ++i) {
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    fixed_cexp(-2 * pi * i / t->ncfft, &t->twiddles[2 * i], &t->twiddles[2 * i + 1]);
  }
  t->super_twiddles.resize(2 * (t->ncfft / 2));
  for (int i = 0; i < t->ncfft / 2; ++i)
    fixed_cexp(-3.14159265358979323846264338327 * (static_cast<double>(i + 1) / t->ncfft + .5),
               &t->super_twiddles[2 * i], &t->super_twiddles[2 * i + 1]);

  // ---- A.4 mel filterbank ----------------------------------------------------------------------------
  const int C = c.num_channels, C1 = C + 1;
  const int spectrum_size = t->fft_size / 2 + 1;
  std::vector<float> center(C1);
  const float mel_lo = mel_of(c.lower_band_limit);
  const float mel_hi = mel_of(c.upper_band_limit);
  {
    const float span = mel_hi - mel_lo;
    const float spacing = span / static_cast<float>(C1);
    for (int i = 0; i < C1; ++i) center[i] = mel_lo + (spacing * static_cast<float>(i + 1));
  }
  const float hz_per_bin = static_cast<float>(0.5 * c.sample_rate / (static_cast<float>(spectrum_size) - 1));
  t->start_index = static_cast<int>(1.5 + c.lower_band_limit / hz_per_bin);
  t->end_index = 0;
  std::vector<int> astart(C1), awidth(C1);
  t->fb_freq_starts.assign(C1, 0);
  t->fb_weight_starts.assign(C1, 0);
  t->fb_widths.assign(C1, 0);
  {
    int cur = t->start_index, running = 0;
    bool zeros_inserted = false;
    for (int ch = 0; ch < C1; ++ch) {
      int f = cur;
      while (f < 2 * spectrum_size && mel_of(static_cast<float>(f) * hz_per_bin) <= center[ch]) ++f;
      const int width = f - cur;
      astart[ch] = cur;
      awidth[ch] = width;
      if (width == 0) {
        // empty channel: upstream points it at 4 zero weights placed at the front of the arrays
        t->fb_freq_starts[ch] = 0;
        t->fb_weight_starts[ch] = 0;
        t->fb_widths[ch] = 4;
        if (!zeros_inserted) {
          zeros_inserted = true;
          for (int j = 0; j < ch; ++j) t->fb_weight_starts[j] += 4;
          running += 4;
        }
      } else {
        const int aligned = (cur / 2) * 2;
        const int padded = (((cur - aligned + width) - 1) / 4 + 1) * 4;
        t->fb_freq_starts[ch] = static_cast<int16_t>(aligned);
        t->fb_weight_starts[ch] = static_cast<int16_t>(running);
        t->fb_widths[ch] = static_cast<int16_t>(padded);
        running += padded;
      }
      cur = f;
    }
    t->num_weights = running;
  }
  t->fb_weights.assign(t->num_weights, 0);
  t->fb_unweights.assign(t->num_weights, 0);
  std::vector<int16_t> wbin(spectrum_size + 1, 0), ubin(spectrum_size + 1, 0);  // per-bin quantised weights
  for (int ch = 0; ch < C1; ++ch) {
    const float denom = (ch == 0) ? mel_lo : center[ch - 1];
    const int offset = astart[ch] - t->fb_freq_starts[ch];
    for (int j = 0; j < awidth[ch]; ++j) {
      const int bin = astart[ch] + j;
      const float w = (center[ch] - mel_of(static_cast<float>(bin) * hz_per_bin)) / (center[ch] - denom);
      const int16_t W = static_cast<int16_t>(std::floor(static_cast<double>(w * static_cast<float>(1 << kFilterbankBits)) + 0.5));
      const int16_t U = static_cast<int16_t>(std::floor((1.0 - static_cast<double>(w)) * (1 << kFilterbankBits) + 0.5));
      const int idx = t->fb_weight_starts[ch] + offset + j;
      t->fb_weights[idx] = W;
      t->fb_unweights[idx] = U;
      if (bin <= spectrum_size) { wbin[bin] = W; ubin[bin] = U; }
    }
    if (awidth[ch] > 0 && astart[ch] + awidth[ch] > t->end_index) t->end_index = astart[ch] + awidth[ch];
  }
  if (t->end_index >= spectrum_size)
    return fail(MKWS_ERR_INVALID_ARG, "frontend cfg: filterbank end_index %d is above the spectrum size %d", t->end_index, spectrum_size);
  // per-output-channel flattened coefficients (unweights of channel c's bins, weights of channel c+1's)
  t->out_start.resize(C); t->out_len.resize(C); t->out_off.resize(C);
  t->out_coef.clear();
  for (int o = 0; o < C; ++o) {
    t->out_start[o] = static_cast<int16_t>(astart[o]);
    t->out_len[o] = static_cast<int16_t>(awidth[o] + awidth[o + 1]);
    t->out_off[o] = static_cast<int16_t>(t->out_coef.size());
    for (int j = 0; j < awidth[o]; ++j) t->out_coef.push_back(ubin[astart[o] + j]);
    for (int j = 0; j < awidth[o + 1]; ++j) t->out_coef.push_back(wbin[astart[o + 1] + j]);
  }

  // ---- A.6 noise-reduction constants (float * int -> uint16 truncation) ------------------------------
  t->even_smoothing = static_cast<uint16_t>(c.even_smoothing * static_cast<float>(1 << kNoiseReductionBits));
  t->odd_smoothing = static_cast<uint16_t>(c.odd_smoothing * static_cast<float>(1 << kNoiseReductionBits));
  t->min_signal_remaining = static_cast<uint16_t>(c.min_signal_remaining * static_cast<float>(1 << kNoiseReductionBits));

  // ---- A.7 PCAN gain LUT ------------------------------------------------------------------------------
  t->correction_bits = bit_length(static_cast<uint32_t>(t->fft_size)) - 1 - (kFilterbankBits / 2);
  t->snr_shift = c.gain_bits - t->correction_bits - kPcanSnrBits;
  t->pcan_lut.assign(4 * kWideDynamicBits - 3, 0);
  if (c.enable_pcan) {
    if (t->snr_shift < 0 || t->snr_shift > 40)
      return fail(MKWS_ERR_INVALID_ARG, "frontend cfg: gain_bits %d gives snr_shift %d", c.gain_bits, t->snr_shift);
    const int input_bits = c.smoothing_bits - t->correction_bits;
    if (input_bits < 0 || input_bits > 31)
      return fail(MKWS_ERR_INVALID_ARG, "frontend cfg: smoothing_bits %d too small for fft size %d", c.smoothing_bits, t->fft_size);
    auto gain = [&](uint32_t x) -> int16_t {
      const float xf = static_cast<float>(x) / static_cast<float>(1u << input_bits);
      const float g = static_cast<float>(1u << c.gain_bits) * powf(xf + c.pcan_offset, -c.pcan_strength);
      if (g > 32767.0f) return 32767;
      return static_cast<int16_t>(g + 0.5f);
    };
    t->pcan_lut[0] = gain(0);
    t->pcan_lut[1] = gain(1);
    for (int interval = 2; interval <= kWideDynamicBits; ++interval) {
      const uint32_t x0 = 1u << (interval - 1);
      const uint32_t x1 = x0 + (x0 >> 1);
      const uint32_t x2 = (interval == kWideDynamicBits) ? x0 + (x0 - 1) : 2 * x0;
      const int32_t y0 = gain(x0), y1 = gain(x1), y2 = gain(x2);
      const int32_t a1 = 4 * (y1 - y0) - (y2 - y0);
      const int32_t a2 = (y2 - y0) - a1;
      t->pcan_lut[4 * interval - 6] = static_cast<int16_t>(y0);
      t->pcan_lut[4 * interval - 5] = static_cast<int16_t>(a1);
      t->pcan_lut[4 * interval - 4] = static_cast<int16_t>(a2);
    }
  }

  // ---- A.8 log LUT: round(2^16 * (log2(1 + k/128) - k/128)), k = 0..128, plus a trailing 0 ------------
  t->log_lut.assign(130, 0);
  for (int k = 0; k <= 128; ++k)
    t->log_lut[k] = static_cast<uint16_t>(std::floor(65536.0 * (std::log2(1.0 + k / 128.0) - k / 128.0) + 0.5));
  return MKWS_OK;
}

}  // namespace mkws
