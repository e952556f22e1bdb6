// Architecture table of the embedding network and the layout of its weight blob.
//
// The network is Keras' EfficientNetB0(include_top=False, weights=None, input_shape=(49,40,1))
// followed by GlobalAveragePooling2D and Dense 2048 relu / 2048 relu / 1024 selu, as built by
// multilingual_kws/train_multilingual_embedding.py:58-83 and cut at "dense_2" by
// multilingual_kws/embedding/transfer_learning.py:36-43.  Layer table: SURVEY.md Appendix B.
// Tensors appear in the blob in network order with Keras names, shapes and layouts (conv kernels
// HWIO, depthwise [kh,kw,C,1], dense [in,out], BN gamma/beta/moving_mean/moving_variance).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace mkws {

struct MBConvSpec {
  const char* name;   // "1a", "2a", ...
  int in_ch, out_ch, kernel, stride, expand;
};

// EfficientNet-B0 block list (width/depth multipliers 1.0): 16 MBConv blocks.
static const MBConvSpec kBlocks[] = {
    {"1a", 32, 16, 3, 1, 1},
    {"2a", 16, 24, 3, 2, 6},  {"2b", 24, 24, 3, 1, 6},
    {"3a", 24, 40, 5, 2, 6},  {"3b", 40, 40, 5, 1, 6},
    {"4a", 40, 80, 3, 2, 6},  {"4b", 80, 80, 3, 1, 6},  {"4c", 80, 80, 3, 1, 6},
    {"5a", 80, 112, 5, 1, 6}, {"5b", 112, 112, 5, 1, 6}, {"5c", 112, 112, 5, 1, 6},
    {"6a", 112, 192, 5, 2, 6}, {"6b", 192, 192, 5, 1, 6}, {"6c", 192, 192, 5, 1, 6}, {"6d", 192, 192, 5, 1, 6},
    {"7a", 192, 320, 3, 1, 6},
};
constexpr int kNumBlocks = sizeof(kBlocks) / sizeof(kBlocks[0]);
constexpr int kInH = 49, kInW = 40;
constexpr int kStemCh = 32, kTopCh = 1280;
constexpr int kDense0 = 2048, kDense1 = 2048, kEmbDim = 1024;
constexpr float kBnEps = 1e-3f;

inline int se_channels(const MBConvSpec& b) {
  int se = static_cast<int>(b.in_ch * 0.25);
  return se < 1 ? 1 : se;
}

struct TensorInfo {
  std::string name;
  std::vector<int> shape;
  size_t offset;   // in floats
  size_t count;
};

// Enumerates every tensor of the blob in order.
inline std::vector<TensorInfo> enumerate_tensors() {
  std::vector<TensorInfo> v;
  size_t off = 0;
  auto add = [&](const std::string& name, std::vector<int> shape) {
    size_t n = 1;
    for (int d : shape) n *= static_cast<size_t>(d);
    v.push_back({name, shape, off, n});
    off += n;
  };
  auto bn = [&](const std::string& p, int c) {
    add(p + "/gamma", {c}); add(p + "/beta", {c}); add(p + "/moving_mean", {c}); add(p + "/moving_variance", {c});
  };
  add("normalization/mean", {1});
  add("normalization/variance", {1});
  add("stem_conv/kernel", {3, 3, 1, kStemCh});
  bn("stem_bn", kStemCh);
  for (int i = 0; i < kNumBlocks; ++i) {
    const MBConvSpec& b = kBlocks[i];
    const std::string p = std::string("block") + b.name;
    const int ce = b.in_ch * b.expand, se = se_channels(b);
    if (b.expand != 1) { add(p + "_expand_conv/kernel", {1, 1, b.in_ch, ce}); bn(p + "_expand_bn", ce); }
    add(p + "_dwconv/depthwise_kernel", {b.kernel, b.kernel, ce, 1});
    bn(p + "_bn", ce);
    add(p + "_se_reduce/kernel", {1, 1, ce, se}); add(p + "_se_reduce/bias", {se});
    add(p + "_se_expand/kernel", {1, 1, se, ce}); add(p + "_se_expand/bias", {ce});
    add(p + "_project_conv/kernel", {1, 1, ce, b.out_ch}); bn(p + "_project_bn", b.out_ch);
  }
  add("top_conv/kernel", {1, 1, kBlocks[kNumBlocks - 1].out_ch, kTopCh});
  bn("top_bn", kTopCh);
  add("dense/kernel", {kTopCh, kDense0}); add("dense/bias", {kDense0});
  add("dense_1/kernel", {kDense0, kDense1}); add("dense_1/bias", {kDense1});
  add("dense_2/kernel", {kDense1, kEmbDim}); add("dense_2/bias", {kEmbDim});
  return v;
}

}  // namespace mkws
