/*
 * microfrontend_oracle.c -- CPU restatement of the TFLite-Micro "audio_microfrontend" feature
 * extractor, the arithmetic behind multilingual_kws/embedding/input_data.py:19-35
 * (to_micro_spectrogram).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The shipped product (multilingual_kws_amd/, libmkws_hip.so) never links, imports
 * or calls anything in this file.
 *
 * Provenance.  The reference repo contains no arithmetic for this path: input_data.py:25-33 calls
 * TensorFlow's AudioMicrofrontend op, which is an un-vendored third-party dependency
 *   tensorflow (docker/Dockerfile:1 pins 2.7.0):
 *     tensorflow/lite/experimental/microfrontend/lib/{window,fft,filterbank,noise_reduction,
 *       pcan_gain_control,log_scale,frontend}{,_util}.c, log_lut.c, bits.h
 *     third_party/kissfft (kiss_fft.c / tools/kiss_fftr.c, FIXED_POINT=16)
 * TensorFlow is not installable here (no wheel, no network), so this file restates the published
 * algorithm from its specification.  Each function names the upstream
 * routine it restates.  Types are followed exactly (int16 wrap, int32 products, uint64
 * accumulators, C float vs double evaluation in table construction).
 *
 * Pinning.  The reference has no tests for this path, but it holds one OUTPUT of the real op: the tutorial notebook's cell 13 renders
 * input_data.file2spec (= TF's AudioMicrofrontend op with its defaults) for three clips whose WAV bytes the notebook embeds.  This
 * restatement reproduces that rendering cell for cell (tests/test_tutorial_png_pin.py: all 5 880 cells carry the colour of the value
 * computed here; one colour step ~ 2.5 raw integer units, and every wrong op default recolours cells) -- pinned to reference-held data
 * at that resolution.  Below a colour step, bit-exactness rests on upstream TensorFlow's own unit-test constants
 * (window_test.cc, noise_reduction_test.cc, frontend_test.cc / audio_microfrontend_op_test.py:
 * the 4x2 known answer {{479,425},{436,378},{410,350},{391,325}}) and on the SURVEY.md Appendix D
 * checksums; see tests/test_oracle_frontend.py and tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define kFrontendWindowBits 12
#define kFilterbankBits 12
#define kNoiseReductionBits 14
#define kPcanSnrBits 12
#define kPcanOutputBits 6
#define kWideDynamicFunctionBits 32
#define kWideDynamicFunctionLUTSize (4 * kWideDynamicFunctionBits - 3)
#define kLogScaleLog2 16
#define kLogScale 65536
#define kLogSegmentsLog2 7
#define kLogCoeff 45426

typedef struct {
  int32_t sample_rate;
  int32_t window_size_ms;
  int32_t window_step_ms;
  int32_t num_channels;
  float upper_band_limit;
  float lower_band_limit;
  int32_t smoothing_bits;
  float even_smoothing;
  float odd_smoothing;
  float min_signal_remaining;
  int32_t enable_pcan;
  float pcan_strength;
  float pcan_offset;
  int32_t gain_bits;
  int32_t enable_log;
  int32_t scale_shift;
} mfo_config;

typedef struct { int16_t r, i; } cpx16;

typedef struct {
  mfo_config cfg;
  /* window (window_util.c) */
  int window_size, window_step;
  int16_t* window_coef;
  /* fft (fft_util.c + kiss_fftr_alloc) */
  int fft_size, ncfft;
  cpx16* twiddles;       /* ncfft */
  cpx16* super_twiddles; /* ncfft/2 */
  int factors[64];
  /* filterbank (filterbank_util.c) */
  int start_index, end_index, num_weights;
  int16_t *chan_freq_starts, *chan_weight_starts, *chan_widths;
  int16_t *weights, *unweights;
  /* noise reduction */
  uint16_t even_smoothing, odd_smoothing, min_signal_remaining;
  /* pcan */
  int16_t gain_lut[kWideDynamicFunctionLUTSize];
  int snr_shift;
  int correction_bits;
  /* log */
  uint16_t log_lut[130];
} mfo_state;

/* ---- bits.h ------------------------------------------------------------------------------- */
static int msb32(uint32_t x) { return x ? 32 - __builtin_clz(x) : 0; }
static int msb64(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; }

/* bits.h Sqrt32: restoring integer square root, then round to nearest (saturating at 0xFFFF). */
static uint32_t sqrt32(uint32_t num) {
  if (num == 0) return 0;
  uint32_t res = 0;
  int max_bit_number = 32 - msb32(num);
  max_bit_number |= 1;
  uint32_t bit = 1u << (31 - max_bit_number);
  int iterations = (31 - max_bit_number) / 2 + 1;
  while (iterations--) {
    if (num >= res + bit) {
      num -= res + bit;
      res = (res >> 1) + bit;
    } else {
      res >>= 1;
    }
    bit >>= 2;
  }
  if (num > res && res != 0xFFFF) ++res;
  return res;
}

/* bits.h Sqrt64. */
static uint32_t sqrt64(uint64_t num) {
  if ((num >> 32) == 0) return sqrt32((uint32_t)num);
  uint64_t res = 0;
  int max_bit_number = 64 - msb64(num);
  max_bit_number |= 1;
  uint64_t bit = 1ull << (63 - max_bit_number);
  int iterations = (63 - max_bit_number) / 2 + 1;
  while (iterations--) {
    if (num >= res + bit) {
      num -= res + bit;
      res = (res >> 1) + bit;
    } else {
      res >>= 1;
    }
    bit >>= 2;
  }
  if (num > res && res != 0xFFFFFFFFull) ++res;
  return (uint32_t)res;
}

/* ---- kissfft FIXED_POINT=16 helpers (_kiss_fft_guts.h) ------------------------------------- */
static inline int16_t sround(int32_t v) { return (int16_t)((v + (1 << 14)) >> 15); }
static inline int16_t fixdiv(int16_t v, int k) { return sround((int32_t)v * (32767 / k)); }
static inline cpx16 cmul(cpx16 a, cpx16 b) {
  cpx16 m;
  m.r = sround((int32_t)a.r * b.r - (int32_t)a.i * b.i);
  m.i = sround((int32_t)a.r * b.i + (int32_t)a.i * b.r);
  return m;
}
static inline cpx16 cadd(cpx16 a, cpx16 b) { cpx16 c = {(int16_t)(a.r + b.r), (int16_t)(a.i + b.i)}; return c; }
static inline cpx16 csub(cpx16 a, cpx16 b) { cpx16 c = {(int16_t)(a.r - b.r), (int16_t)(a.i - b.i)}; return c; }

/* kiss_fft.c kf_factor */
static void kf_factor(int n, int* facbuf) {
  int p = 4;
  double floor_sqrt = floor(sqrt((double)n));
  do {
    while (n % p) {
      switch (p) {
        case 4: p = 2; break;
        case 2: p = 3; break;
        default: p += 2; break;
      }
      if (p > floor_sqrt) p = n;
    }
    n /= p;
    *facbuf++ = p;
    *facbuf++ = n;
  } while (n > 1);
}

/* kiss_fft.c kf_bfly2 */
static void kf_bfly2(cpx16* Fout, size_t fstride, const mfo_state* st, int m) {
  cpx16* Fout2 = Fout + m;
  const cpx16* tw1 = st->twiddles;
  do {
    Fout->r = fixdiv(Fout->r, 2); Fout->i = fixdiv(Fout->i, 2);
    Fout2->r = fixdiv(Fout2->r, 2); Fout2->i = fixdiv(Fout2->i, 2);
    cpx16 t = cmul(*Fout2, *tw1);
    tw1 += fstride;
    *Fout2 = csub(*Fout, t);
    *Fout = cadd(*Fout, t);
    ++Fout2; ++Fout;
  } while (--m);
}

/* kiss_fft.c kf_bfly4 (forward) */
static void kf_bfly4(cpx16* Fout, size_t fstride, const mfo_state* st, size_t m) {
  const cpx16 *tw1, *tw2, *tw3;
  cpx16 s[6];
  size_t k = m;
  const size_t m2 = 2 * m, m3 = 3 * m;
  tw3 = tw2 = tw1 = st->twiddles;
  do {
    Fout[0].r = fixdiv(Fout[0].r, 4); Fout[0].i = fixdiv(Fout[0].i, 4);
    Fout[m].r = fixdiv(Fout[m].r, 4); Fout[m].i = fixdiv(Fout[m].i, 4);
    Fout[m2].r = fixdiv(Fout[m2].r, 4); Fout[m2].i = fixdiv(Fout[m2].i, 4);
    Fout[m3].r = fixdiv(Fout[m3].r, 4); Fout[m3].i = fixdiv(Fout[m3].i, 4);
    s[0] = cmul(Fout[m], *tw1);
    s[1] = cmul(Fout[m2], *tw2);
    s[2] = cmul(Fout[m3], *tw3);
    s[5] = csub(Fout[0], s[1]);
    Fout[0] = cadd(Fout[0], s[1]);
    s[3] = cadd(s[0], s[2]);
    s[4] = csub(s[0], s[2]);
    Fout[m2] = csub(Fout[0], s[3]);
    tw1 += fstride; tw2 += fstride * 2; tw3 += fstride * 3;
    Fout[0] = cadd(Fout[0], s[3]);
    Fout[m].r = (int16_t)(s[5].r + s[4].i);
    Fout[m].i = (int16_t)(s[5].i - s[4].r);
    Fout[m3].r = (int16_t)(s[5].r - s[4].i);
    Fout[m3].i = (int16_t)(s[5].i + s[4].r);
    ++Fout;
  } while (--k);
}

/* kiss_fft.c kf_work (in_stride == 1; only radices 2 and 4 occur for power-of-two sizes) */
static void kf_work(cpx16* Fout, const cpx16* f, size_t fstride, const int* factors, const mfo_state* st) {
  cpx16* Fout_beg = Fout;
  const int p = *factors++;
  const int m = *factors++;
  const cpx16* Fout_end = Fout + p * m;
  if (m == 1) {
    do { *Fout = *f; f += fstride; } while (++Fout != Fout_end);
  } else {
    do { kf_work(Fout, f, fstride * p, factors, st); f += fstride; } while ((Fout += m) != Fout_end);
  }
  Fout = Fout_beg;
  if (p == 2) kf_bfly2(Fout, fstride, st, m);
  else kf_bfly4(Fout, fstride, st, (size_t)m);
}

/* tools/kiss_fftr.c kiss_fftr: real FFT of fft_size int16 -> ncfft+1 complex bins */
static void kiss_fftr(const mfo_state* st, const int16_t* timedata, cpx16* freqdata, cpx16* tmpbuf) {
  const int ncfft = st->ncfft;
  kf_work(tmpbuf, (const cpx16*)timedata, 1, st->factors, st);
  cpx16 tdc = tmpbuf[0];
  tdc.r = fixdiv(tdc.r, 2); tdc.i = fixdiv(tdc.i, 2);
  freqdata[0].r = (int16_t)(tdc.r + tdc.i);
  freqdata[ncfft].r = (int16_t)(tdc.r - tdc.i);
  freqdata[ncfft].i = freqdata[0].i = 0;
  for (int k = 1; k <= ncfft / 2; ++k) {
    cpx16 fpk = tmpbuf[k];
    cpx16 fpnk;
    fpnk.r = tmpbuf[ncfft - k].r;
    fpnk.i = (int16_t)(-tmpbuf[ncfft - k].i);
    fpk.r = fixdiv(fpk.r, 2); fpk.i = fixdiv(fpk.i, 2);
    fpnk.r = fixdiv(fpnk.r, 2); fpnk.i = fixdiv(fpnk.i, 2);
    cpx16 f1k = cadd(fpk, fpnk);
    cpx16 f2k = csub(fpk, fpnk);
    cpx16 tw = cmul(f2k, st->super_twiddles[k - 1]);
    freqdata[k].r = (int16_t)((f1k.r + tw.r) >> 1);
    freqdata[k].i = (int16_t)((f1k.i + tw.i) >> 1);
    freqdata[ncfft - k].r = (int16_t)((f1k.r - tw.r) >> 1);
    freqdata[ncfft - k].i = (int16_t)((tw.i - f1k.i) >> 1);
  }
}

/* ---- table construction -------------------------------------------------------------------- */
/* filterbank_util.c FreqToMel: float in, double math, float out */
static float freq_to_mel(float freq) { return 1127.0 * log1p(freq / 700.0); }

/* pcan_gain_control_util.c PcanGainLookupFunction */
static int16_t pcan_gain_lookup(const mfo_config* c, int32_t input_bits, uint32_t x) {
  const float x_as_float = ((float)x) / ((uint32_t)1 << input_bits);
  const float gain_as_float = ((uint32_t)1 << c->gain_bits) * powf(x_as_float + c->pcan_offset, -(c->pcan_strength));
  if (gain_as_float > 0x7FFF) return 0x7FFF;
  return (int16_t)(gain_as_float + 0.5f);
}

void mfo_destroy(mfo_state* st) {
  if (!st) return;
  free(st->window_coef); free(st->twiddles); free(st->super_twiddles);
  free(st->chan_freq_starts); free(st->chan_weight_starts); free(st->chan_widths);
  free(st->weights); free(st->unweights);
  free(st);
}

/* frontend_util.c FrontendPopulateState and the *_util.c PopulateState routines it calls. */
mfo_state* mfo_create(const mfo_config* cfg) {
  mfo_state* st = (mfo_state*)calloc(1, sizeof(mfo_state));
  st->cfg = *cfg;
  /* window_util.c WindowPopulateState */
  st->window_size = cfg->window_size_ms * cfg->sample_rate / 1000;
  st->window_step = cfg->window_step_ms * cfg->sample_rate / 1000;
  st->window_coef = (int16_t*)malloc(sizeof(int16_t) * st->window_size);
  {
    const float arg = M_PI * 2.0 / ((float)st->window_size);
    for (int i = 0; i < st->window_size; ++i) {
      float float_value = 0.5 - (0.5 * cos(arg * (i + 0.5)));
      st->window_coef[i] = floor(float_value * (1 << kFrontendWindowBits) + 0.5);
    }
  }
  /* fft_util.c FftPopulateState + kiss_fftr_alloc */
  st->fft_size = 1;
  while (st->fft_size < st->window_size) st->fft_size <<= 1;
  st->ncfft = st->fft_size / 2;
  st->twiddles = (cpx16*)malloc(sizeof(cpx16) * st->ncfft);
  for (int i = 0; i < st->ncfft; ++i) {
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    double phase = -2 * pi * i / st->ncfft;
    st->twiddles[i].r = floor(.5 + 32767 * cos(phase));
    st->twiddles[i].i = floor(.5 + 32767 * sin(phase));
  }
  st->super_twiddles = (cpx16*)malloc(sizeof(cpx16) * (st->ncfft / 2 + 1));
  for (int i = 0; i < st->ncfft / 2; ++i) {
    double phase = -3.14159265358979323846264338327 * ((double)(i + 1) / st->ncfft + .5);
    st->super_twiddles[i].r = floor(.5 + 32767 * cos(phase));
    st->super_twiddles[i].i = floor(.5 + 32767 * sin(phase));
  }
  kf_factor(st->ncfft, st->factors);

  /* filterbank_util.c FilterbankPopulateState */
  {
    const int nch1 = cfg->num_channels + 1;
    const int spectrum_size = st->fft_size / 2 + 1;
    const int index_alignment = 2; /* kFilterbankIndexAlignment(4) / sizeof(int16_t) */
    st->chan_freq_starts = (int16_t*)malloc(sizeof(int16_t) * nch1);
    st->chan_weight_starts = (int16_t*)malloc(sizeof(int16_t) * nch1);
    st->chan_widths = (int16_t*)malloc(sizeof(int16_t) * nch1);
    float* center = (float*)malloc(sizeof(float) * nch1);
    int16_t* actual_starts = (int16_t*)malloc(sizeof(int16_t) * nch1);
    int16_t* actual_widths = (int16_t*)malloc(sizeof(int16_t) * nch1);
    /* CalculateCenterFrequencies(num_channels_plus_1, ...) */
    const float mel_low = freq_to_mel(cfg->lower_band_limit);
    const float mel_hi = freq_to_mel(cfg->upper_band_limit);
    const float mel_span = mel_hi - mel_low;
    const float mel_spacing = mel_span / ((float)nch1);
    for (int i = 0; i < nch1; ++i) center[i] = mel_low + (mel_spacing * (i + 1));

    const float hz_per_sbin = 0.5 * cfg->sample_rate / ((float)spectrum_size - 1);
    st->start_index = 1.5 + cfg->lower_band_limit / hz_per_sbin;
    st->end_index = 0;
    int chan_freq_index_start = st->start_index;
    int weight_index_start = 0;
    int needs_zeros = 0;
    for (int chan = 0; chan < nch1; ++chan) {
      int freq_index = chan_freq_index_start;
      while (freq_to_mel((freq_index)*hz_per_sbin) <= center[chan]) ++freq_index;
      const int width = freq_index - chan_freq_index_start;
      actual_starts[chan] = chan_freq_index_start;
      actual_widths[chan] = width;
      if (width == 0) {
        st->chan_freq_starts[chan] = 0;
        st->chan_weight_starts[chan] = 0;
        st->chan_widths[chan] = 4;
        if (!needs_zeros) {
          needs_zeros = 1;
          for (int j = 0; j < chan; ++j) st->chan_weight_starts[j] += 4;
          weight_index_start += 4;
        }
      } else {
        const int aligned_start = (chan_freq_index_start / index_alignment) * index_alignment;
        const int aligned_width = (chan_freq_index_start - aligned_start + width);
        const int padded_width = (((aligned_width - 1) / 4) + 1) * 4;
        st->chan_freq_starts[chan] = aligned_start;
        st->chan_weight_starts[chan] = weight_index_start;
        st->chan_widths[chan] = padded_width;
        weight_index_start += padded_width;
      }
      chan_freq_index_start = freq_index;
    }
    st->num_weights = weight_index_start;
    st->weights = (int16_t*)calloc(weight_index_start, sizeof(int16_t));
    st->unweights = (int16_t*)calloc(weight_index_start, sizeof(int16_t));
    for (int chan = 0; chan < nch1; ++chan) {
      int frequency = actual_starts[chan];
      const int num_frequencies = actual_widths[chan];
      const int frequency_offset = frequency - st->chan_freq_starts[chan];
      const int weight_start = st->chan_weight_starts[chan];
      const float denom_val = (chan == 0) ? mel_low : center[chan - 1];
      for (int j = 0; j < num_frequencies; ++j, ++frequency) {
        const float weight = (center[chan] - freq_to_mel(frequency * hz_per_sbin)) / (center[chan] - denom_val);
        const int weight_index = weight_start + frequency_offset + j;
        /* QuantizeFilterbankWeights */
        st->weights[weight_index] = floor(weight * (1 << kFilterbankBits) + 0.5);
        st->unweights[weight_index] = floor((1.0 - weight) * (1 << kFilterbankBits) + 0.5);
      }
      if (num_frequencies > 0 && frequency > st->end_index) st->end_index = frequency;
    }
    free(center); free(actual_starts); free(actual_widths);
    if (st->end_index >= spectrum_size) { mfo_destroy(st); return NULL; }
  }

  /* noise_reduction_util.c NoiseReductionPopulateState */
  st->even_smoothing = cfg->even_smoothing * (1 << kNoiseReductionBits);
  st->odd_smoothing = cfg->odd_smoothing * (1 << kNoiseReductionBits);
  st->min_signal_remaining = cfg->min_signal_remaining * (1 << kNoiseReductionBits);

  /* frontend_util.c: input_correction_bits; pcan_gain_control_util.c PcanGainControlPopulateState */
  st->correction_bits = msb32((uint32_t)st->fft_size) - 1 - (kFilterbankBits / 2);
  if (cfg->enable_pcan) {
    st->snr_shift = cfg->gain_bits - st->correction_bits - kPcanSnrBits;
    const int32_t input_bits = cfg->smoothing_bits - st->correction_bits;
    int16_t* lut = st->gain_lut;
    lut[0] = pcan_gain_lookup(cfg, input_bits, 0);
    lut[1] = pcan_gain_lookup(cfg, input_bits, 1);
    lut -= 6;
    for (int interval = 2; interval <= kWideDynamicFunctionBits; ++interval) {
      const uint32_t x0 = (uint32_t)1 << (interval - 1);
      const uint32_t x1 = x0 + (x0 >> 1);
      const uint32_t x2 = (interval == kWideDynamicFunctionBits) ? x0 + (x0 - 1) : 2 * x0;
      const int16_t y0 = pcan_gain_lookup(cfg, input_bits, x0);
      const int16_t y1 = pcan_gain_lookup(cfg, input_bits, x1);
      const int16_t y2 = pcan_gain_lookup(cfg, input_bits, x2);
      const int32_t diff1 = (int32_t)y1 - y0;
      const int32_t diff2 = (int32_t)y2 - y0;
      const int32_t a1 = 4 * diff1 - diff2;
      const int32_t a2 = diff2 - a1;
      lut[4 * interval] = y0;
      lut[4 * interval + 1] = (int16_t)a1;
      lut[4 * interval + 2] = (int16_t)a2;
    }
  }
  /* log_lut.c kLogLut: round(2^16 * (log2(1 + k/128) - k/128)), k = 0..128, then a trailing 0 */
  for (int k = 0; k <= 128; ++k) {
    double v = 65536.0 * (log2(1.0 + k / 128.0) - k / 128.0);
    st->log_lut[k] = (uint16_t)floor(v + 0.5);
  }
  st->log_lut[129] = 0;
  return st;
}

/* ---- per-frame stages ---------------------------------------------------------------------- */
/* window.c WindowProcessSamples (the windowing loop); returns max_abs_output_value */
static int16_t window_apply(const mfo_state* st, const int16_t* in, int16_t* out) {
  int16_t max_abs = 0;
  for (int i = 0; i < st->window_size; ++i) {
    int16_t v = (int16_t)((((int32_t)in[i]) * st->window_coef[i]) >> kFrontendWindowBits);
    out[i] = v;
    if (v < 0) v = (int16_t)(-v);
    if (v > max_abs) max_abs = v;
  }
  return max_abs;
}

/* noise_reduction.c NoiseReductionApply */
static void noise_reduction_apply(const mfo_state* st, uint32_t* estimate, uint32_t* signal) {
  for (int i = 0; i < st->cfg.num_channels; ++i) {
    const uint32_t smoothing = ((i & 1) == 0) ? st->even_smoothing : st->odd_smoothing;
    const uint32_t one_minus_smoothing = (1 << kNoiseReductionBits) - smoothing;
    const uint32_t signal_scaled_up = signal[i] << st->cfg.smoothing_bits;
    uint32_t est = (uint32_t)((((uint64_t)signal_scaled_up * smoothing) + ((uint64_t)estimate[i] * one_minus_smoothing)) >> kNoiseReductionBits);
    estimate[i] = est;
    if (est > signal_scaled_up) est = signal_scaled_up;
    const uint32_t floor_ = (uint32_t)(((uint64_t)signal[i] * st->min_signal_remaining) >> kNoiseReductionBits);
    const uint32_t subtracted = (signal_scaled_up - est) >> st->cfg.smoothing_bits;
    signal[i] = subtracted > floor_ ? subtracted : floor_;
  }
}

/* pcan_gain_control.c WideDynamicFunction */
static int16_t wide_dynamic_function(uint32_t x, const int16_t* lut) {
  if (x <= 2) return lut[x];
  const int16_t interval = (int16_t)msb32(x);
  lut += 4 * interval - 6;
  const int16_t frac = (int16_t)(((interval < 11) ? (x << (11 - interval)) : (x >> (interval - 11))) & 0x3FF);
  int32_t result = ((int32_t)lut[2] * frac) >> 5;
  result += (int32_t)((uint32_t)lut[1] << 5);
  result *= frac;
  result = (result + (1 << 14)) >> 15;
  result += lut[0];
  return (int16_t)result;
}

/* pcan_gain_control.c PcanShrink */
static uint32_t pcan_shrink(uint32_t x) {
  if (x < (2 << kPcanSnrBits)) return (x * x) >> (2 + 2 * kPcanSnrBits - kPcanOutputBits);
  return (x >> (kPcanSnrBits - kPcanOutputBits)) - (1 << kPcanOutputBits);
}

/* log_scale.c Log2FractionPart + Log */
static uint32_t log_scale_log(const mfo_state* st, uint32_t x, uint32_t scale_shift) {
  const uint32_t integer = msb32(x) - 1;
  int32_t frac = (int32_t)(x - (1LL << integer));
  if (integer < kLogScaleLog2) frac <<= kLogScaleLog2 - integer;
  else frac >>= integer - kLogScaleLog2;
  const uint32_t base_seg = frac >> (kLogScaleLog2 - kLogSegmentsLog2);
  const uint32_t seg_unit = (((uint32_t)1) << kLogScaleLog2) >> kLogSegmentsLog2;
  const int32_t c0 = st->log_lut[base_seg];
  const int32_t c1 = st->log_lut[base_seg + 1];
  const int32_t seg_base = seg_unit * base_seg;
  const int32_t rel_pos = ((c1 - c0) * (frac - seg_base)) >> kLogScaleLog2;
  const uint32_t fraction = frac + c0 + rel_pos;
  const uint32_t log2v = (integer << kLogScaleLog2) + fraction;
  const uint32_t round = kLogScale / 2;
  const uint32_t loge = (uint32_t)((((uint64_t)kLogCoeff) * log2v + round) >> kLogScaleLog2);
  return ((loge << scale_shift) + round) >> kLogScaleLog2;
}

int mfo_num_frames(const mfo_state* st, int n) {
  if (n < st->window_size) return 0;
  return (n - st->window_size) / st->window_step + 1;
}

/*
 * frontend.c FrontendProcessSamples looped as audio_microfrontend_op.cc does for one op call
 * (fresh noise estimate per call, frame f = samples [step*f, step*f + size)).
 * out: uint16 [num_frames, num_channels].  Optional taps (may be NULL):
 *   tap_sig: uint32 [num_frames, num_channels] filterbank sqrt output (before noise reduction).
 */
int mfo_run_i16(const mfo_state* st, const int16_t* audio, int n, uint16_t* out, uint32_t* tap_sig) {
  const int F = mfo_num_frames(st, n);
  const int C = st->cfg.num_channels;
  const int nch1 = C + 1;
  int16_t* win = (int16_t*)malloc(sizeof(int16_t) * st->window_size);
  int16_t* fft_in = (int16_t*)malloc(sizeof(int16_t) * st->fft_size);
  cpx16* fft_out = (cpx16*)malloc(sizeof(cpx16) * (st->ncfft + 2));
  cpx16* tmpbuf = (cpx16*)malloc(sizeof(cpx16) * st->ncfft);
  uint64_t* work = (uint64_t*)malloc(sizeof(uint64_t) * nch1);
  uint32_t* estimate = (uint32_t*)calloc(C, sizeof(uint32_t));
  uint32_t* signal = (uint32_t*)malloc(sizeof(uint32_t) * C);
  for (int f = 0; f < F; ++f) {
    /* window.c */
    const int16_t max_abs = window_apply(st, audio + (size_t)f * st->window_step, win);
    /* frontend.c: input_shift; fft.c FftCompute */
    const int input_shift = 15 - msb32((uint32_t)max_abs);
    int i;
    for (i = 0; i < st->window_size; ++i) fft_in[i] = (int16_t)(((uint16_t)win[i]) << input_shift);
    for (; i < st->fft_size; ++i) fft_in[i] = 0;
    kiss_fftr(st, fft_in, fft_out, tmpbuf);
    /* filterbank.c FilterbankConvertFftComplexToEnergy: energy aliases the fft output buffer, so
       bins outside [start_index, end_index) keep the raw (real, imag) bit pattern; they only ever
       meet zero weights.  Kept as-is for fidelity. */
    int32_t* energy = (int32_t*)fft_out;
    for (i = st->start_index; i < st->end_index; ++i) {
      const int32_t real = fft_out[i].r, imag = fft_out[i].i;
      const uint32_t mag_squared = (uint32_t)(real * real) + (uint32_t)(imag * imag);
      energy[i] = (int32_t)mag_squared;
    }
    /* filterbank.c FilterbankAccumulateChannels */
    uint64_t wacc = 0, uacc = 0;
    for (int ch = 0; ch < nch1; ++ch) {
      const int32_t* mags = energy + st->chan_freq_starts[ch];
      const int16_t* w = st->weights + st->chan_weight_starts[ch];
      const int16_t* u = st->unweights + st->chan_weight_starts[ch];
      const int width = st->chan_widths[ch];
      for (int j = 0; j < width; ++j) {
        wacc += w[j] * ((uint64_t)mags[j]);
        uacc += u[j] * ((uint64_t)mags[j]);
      }
      work[ch] = wacc;
      wacc = uacc;
      uacc = 0;
    }
    /* filterbank.c FilterbankSqrt */
    for (int c = 0; c < C; ++c) signal[c] = sqrt64(work[c + 1]) >> input_shift;
    if (tap_sig) memcpy(tap_sig + (size_t)f * C, signal, sizeof(uint32_t) * C);
    noise_reduction_apply(st, estimate, signal);
    if (st->cfg.enable_pcan) {
      /* pcan_gain_control.c PcanGainControlApply (noise_estimate = the estimate just updated) */
      for (int c = 0; c < C; ++c) {
        const uint32_t gain = (uint32_t)wide_dynamic_function(estimate[c], st->gain_lut);
        const uint32_t snr = (uint32_t)(((uint64_t)signal[c] * gain) >> st->snr_shift);
        signal[c] = pcan_shrink(snr);
      }
    }
    /* log_scale.c LogScaleApply */
    for (int c = 0; c < C; ++c) {
      uint32_t value = signal[c];
      if (st->cfg.enable_log) {
        if (st->correction_bits < 0) value >>= -st->correction_bits;
        else value <<= st->correction_bits;
        value = (value > 1) ? log_scale_log(st, value, (uint32_t)st->cfg.scale_shift) : 0;
      }
      out[(size_t)f * C + c] = (uint16_t)((value < 0xFFFF) ? value : 0xFFFF);
    }
  }
  free(win); free(fft_in); free(fft_out); free(tmpbuf); free(work); free(estimate); free(signal);
  return F;
}

/* input_data.py:23 `tf.cast(tf.multiply(audio, 32768), tf.int16)`: float multiply, truncation
   toward zero.  Out-of-range values (only +1.0 after clip_by_value) are UB in TF; this build
   saturates (SURVEY.md risk R4). */
static inline int16_t f32_to_i16(float a) {
  float v = a * 32768.0f;
  if (v >= 32767.0f) return 32767;
  if (v <= -32768.0f) return -32768;
  return (int16_t)v;
}

/* to_micro_spectrogram for a batch: audio float [B, n] -> out float [B, F, C] (= uint16 * 10/256),
   optionally raw uint16 as well.  Clips are independent; OpenMP over clips for the CPU baseline. */
int mfo_run_batch_f32(const mfo_state* st, const float* audio, int B, int n, float* out_f32, uint16_t* out_u16, int nthreads) {
  const int F = mfo_num_frames(st, n);
  const int C = st->cfg.num_channels;
  const float scale = 10.0f / 256.0f;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
  for (int b = 0; b < B; ++b) {
    int16_t* pcm = (int16_t*)malloc(sizeof(int16_t) * n);
    uint16_t* raw = (uint16_t*)malloc(sizeof(uint16_t) * F * C + 2);
    for (int t = 0; t < n; ++t) pcm[t] = f32_to_i16(audio[(size_t)b * n + t]);
    mfo_run_i16(st, pcm, n, raw, NULL);
    for (int k = 0; k < F * C; ++k) {
      if (out_f32) out_f32[(size_t)b * F * C + k] = (float)raw[k] * scale;
      if (out_u16) out_u16[(size_t)b * F * C + k] = raw[k];
    }
    free(pcm); free(raw);
  }
  return F;
}

/* ---- introspection for per-stage tests ------------------------------------------------------ */
int mfo_window_size(const mfo_state* st) { return st->window_size; }
int mfo_window_step(const mfo_state* st) { return st->window_step; }
int mfo_fft_size(const mfo_state* st) { return st->fft_size; }
int mfo_start_index(const mfo_state* st) { return st->start_index; }
int mfo_end_index(const mfo_state* st) { return st->end_index; }
int mfo_num_weights(const mfo_state* st) { return st->num_weights; }
int mfo_snr_shift(const mfo_state* st) { return st->snr_shift; }
int mfo_correction_bits(const mfo_state* st) { return st->correction_bits; }
const int16_t* mfo_window_coef(const mfo_state* st) { return st->window_coef; }
const int16_t* mfo_weights(const mfo_state* st) { return st->weights; }
const int16_t* mfo_unweights(const mfo_state* st) { return st->unweights; }
const int16_t* mfo_chan_freq_starts(const mfo_state* st) { return st->chan_freq_starts; }
const int16_t* mfo_chan_weight_starts(const mfo_state* st) { return st->chan_weight_starts; }
const int16_t* mfo_chan_widths(const mfo_state* st) { return st->chan_widths; }
const int16_t* mfo_gain_lut(const mfo_state* st) { return st->gain_lut; }
const uint16_t* mfo_log_lut(const mfo_state* st) { return st->log_lut; }
const int16_t* mfo_twiddles(const mfo_state* st) { return (const int16_t*)st->twiddles; }
const int16_t* mfo_super_twiddles(const mfo_state* st) { return (const int16_t*)st->super_twiddles; }

int16_t mfo_window_frame(const mfo_state* st, const int16_t* in, int16_t* out) { return window_apply(st, in, out); }
void mfo_noise_reduction(const mfo_state* st, uint32_t* estimate, uint32_t* signal) { noise_reduction_apply(st, estimate, signal); }
uint32_t mfo_log(const mfo_state* st, uint32_t x) { return log_scale_log(st, x, (uint32_t)st->cfg.scale_shift); }
uint32_t mfo_sqrt64(uint64_t x) { return sqrt64(x); }
int16_t mfo_wide_dynamic_function(const mfo_state* st, uint32_t x) { return wide_dynamic_function(x, st->gain_lut); }
/* full real FFT of one already-scaled frame: in int16[fft_size] -> out int16[2*(ncfft+1)] */
void mfo_fft(const mfo_state* st, const int16_t* in, int16_t* out) {
  cpx16* tmp = (cpx16*)malloc(sizeof(cpx16) * st->ncfft);
  kiss_fftr(st, in, (cpx16*)out, tmp);
  free(tmp);
}
