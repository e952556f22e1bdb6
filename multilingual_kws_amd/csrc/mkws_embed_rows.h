// Entry points of mkws_embed_rows.hip (the register-resident whole-block kernel of blocks 2b / 3b) for mkws_embed.hip's launcher.
#pragma once
#include <hip/hip_runtime.h>

namespace mkws {

struct MidArgs;

enum RowsVariant { kRows2b = 0, kRows3b = 1 };

const char* rows_kernel_name(int variant);                                 // the label the per-kernel tables print
int launch_rows_variant(hipStream_t s, int variant, const MidArgs& a);     // MKWS_OK or an error code (mkws::fail text set)

}  // namespace mkws
