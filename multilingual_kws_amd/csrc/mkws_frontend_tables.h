// Host-side construction of the micro-frontend's integer tables (window, FFT twiddles, mel
// filterbank, PCAN gain LUT, log LUT) for a given mkws_frontend_cfg.
//
// Restates what TensorFlow's microfrontend *_util.c "PopulateState" routines compute (the library
// behind multilingual_kws/embedding/input_data.py:25-33); spec: SURVEY.md Appendix A.1-A.8.  The C
// float/double evaluation order of the upstream table code is followed so every table is
// bit-identical (tests compare against oracle/ and the Appendix D checksums).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/mkws.h"

namespace mkws {

struct FrontendTables {
  // scalars
  int window_size = 0, window_step = 0, fft_size = 0, ncfft = 0;
  int start_index = 0, end_index = 0, num_weights = 0;
  int snr_shift = 0, correction_bits = 0;
  uint32_t even_smoothing = 0, odd_smoothing = 0, min_signal_remaining = 0;
  // upstream-layout tables
  std::vector<int16_t> window_coef;                       // [window_size]
  std::vector<int16_t> twiddles;                          // [2*ncfft] (r,i)
  std::vector<int16_t> super_twiddles;                    // [2*(ncfft/2)]
  std::vector<int16_t> fb_weights, fb_unweights;          // [num_weights] (aligned/padded layout)
  std::vector<int16_t> fb_freq_starts, fb_weight_starts, fb_widths;  // [C+1]
  std::vector<int16_t> pcan_lut;                          // [125]
  std::vector<uint16_t> log_lut;                          // [130]
  // GPU-friendly filterbank: output channel c = sum over bins [out_start[c], out_start[c]+out_len[c])
  // of out_coef[out_off[c]+j] * E[bin]  (the unweights of channel c's bins, then the weights of
  // channel c+1's bins -- identical sums to FilterbankAccumulateChannels' rolling accumulators).
  std::vector<int16_t> out_start, out_len, out_off;       // [C]
  std::vector<int16_t> out_coef;
};

// Returns MKWS_OK or a negative status (message via mkws::fail).
int build_frontend_tables(const mkws_frontend_cfg& cfg, FrontendTables* t);

}  // namespace mkws
