"""CPU restatement (PyTorch, fp32 or fp64) of the embedding network the reference builds with Keras:

    EfficientNetB0(include_top=False, weights=None, input_shape=(49, 40, 1)) -> GlobalAveragePooling2D
    -> Dense(2048, relu) -> Dense(2048, relu) -> Dense(1024, selu)   [layer "dense_2" = the embedding]

(multilingual_kws/train_multilingual_embedding.py:58-83; cut at dense_2 by
multilingual_kws/embedding/transfer_learning.py:36-43).  The layer graph itself lives in Keras
(keras/applications/efficientnet.py, TF 2.7 -- an un-vendored third-party dependency); it is restated
here from SURVEY.md Appendix B with F.conv2d and explicit zero padding.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
PARITY UNPINNED BY THE REFERENCE: it ships neither tests nor the trained checkpoint, and
TensorFlow/Keras cannot be installed here.  What this restatement is checked against instead: an independent
float64 explicit-loop re-derivation of every block and Keras' documented parameter / MAC counts
(tests/test_oracle_embedding.py), and -- the one check not written by this project -- a THIRD-PARTY
PyTorch port of keras/applications/efficientnet.py (Hugging Face transformers' EfficientNetModel) on the
same weights: every block of the unmodified port where its padding equals Keras', and the whole trunk
(tests/golden/make_hf_efficientnet_golden.py -> tests/test_hf_efficientnet_golden.py).  Real TF output
(tests/golden/keras_golden.npz, tools/make_keras_golden.py) is still absent.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# (name, in, out, kernel, stride, expand) -- EfficientNet-B0, SURVEY.md Appendix B
BLOCKS = [
    ("1a", 32, 16, 3, 1, 1),
    ("2a", 16, 24, 3, 2, 6), ("2b", 24, 24, 3, 1, 6),
    ("3a", 24, 40, 5, 2, 6), ("3b", 40, 40, 5, 1, 6),
    ("4a", 40, 80, 3, 2, 6), ("4b", 80, 80, 3, 1, 6), ("4c", 80, 80, 3, 1, 6),
    ("5a", 80, 112, 5, 1, 6), ("5b", 112, 112, 5, 1, 6), ("5c", 112, 112, 5, 1, 6),
    ("6a", 112, 192, 5, 2, 6), ("6b", 192, 192, 5, 1, 6), ("6c", 192, 192, 5, 1, 6), ("6d", 192, 192, 5, 1, 6),
    ("7a", 192, 320, 3, 1, 6),
]
BN_EPS = 1e-3
SELU_SCALE = 1.0507009873554805
SELU_ALPHA = 1.6732632423543772


def tensor_list():
    """[(keras name, shape)] in blob order."""
    out = []

    def bn(p, c):
        out.extend([(p + "/gamma", (c,)), (p + "/beta", (c,)), (p + "/moving_mean", (c,)), (p + "/moving_variance", (c,))])

    out.append(("normalization/mean", (1,)))
    out.append(("normalization/variance", (1,)))
    out.append(("stem_conv/kernel", (3, 3, 1, 32)))
    bn("stem_bn", 32)
    for name, cin, cout, k, s, e in BLOCKS:
        p = "block" + name
        ce, se = cin * e, max(1, int(cin * 0.25))
        if e != 1:
            out.append((p + "_expand_conv/kernel", (1, 1, cin, ce)))
            bn(p + "_expand_bn", ce)
        out.append((p + "_dwconv/depthwise_kernel", (k, k, ce, 1)))
        bn(p + "_bn", ce)
        out.append((p + "_se_reduce/kernel", (1, 1, ce, se)))
        out.append((p + "_se_reduce/bias", (se,)))
        out.append((p + "_se_expand/kernel", (1, 1, se, ce)))
        out.append((p + "_se_expand/bias", (ce,)))
        out.append((p + "_project_conv/kernel", (1, 1, ce, cout)))
        bn(p + "_project_bn", cout)
    out.append(("top_conv/kernel", (1, 1, 320, 1280)))
    bn("top_bn", 1280)
    out.extend([("dense/kernel", (1280, 2048)), ("dense/bias", (2048,)),
                ("dense_1/kernel", (2048, 2048)), ("dense_1/bias", (2048,)),
                ("dense_2/kernel", (2048, 1024)), ("dense_2/bias", (1024,))])
    return out


def blob_size():
    return sum(int(np.prod(s)) for _, s in tensor_list())


def split_blob(blob):
    """flat float32 array -> {name: ndarray} following tensor_list()."""
    blob = np.asarray(blob)
    out, off = {}, 0
    for name, shape in tensor_list():
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
        off += n
    if off != blob.shape[0]:
        raise ValueError(f"blob has {blob.shape[0]} floats, architecture needs {off}")
    return out


def correct_pad(h, w, k):
    """keras.applications.imagenet_utils.correct_pad -> ((top, bottom), (left, right))."""
    adj = (1 - h % 2, 1 - w % 2)
    c = k // 2
    return (c - adj[0], c), (c - adj[1], c)


class EmbeddingOracle:
    def __init__(self, blob, dtype=torch.float32):
        self.dtype = dtype
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in split_blob(blob).items()}

    # -- layer helpers ---------------------------------------------------------------------------
    def _bn(self, x, p):
        w = self.w
        g, b, m, v = w[p + "/gamma"], w[p + "/beta"], w[p + "/moving_mean"], w[p + "/moving_variance"]
        inv = g / torch.sqrt(v + BN_EPS)
        return (x - m.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    def _conv(self, x, name, stride=1, bias=None):
        k = self.w[name].permute(3, 2, 0, 1).contiguous()        # HWIO -> OIHW
        return F.conv2d(x, k, bias=None if bias is None else self.w[bias], stride=stride)

    def _dwconv(self, x, name, stride):
        k = self.w[name]                                         # [kh, kw, C, 1]
        kk = k.permute(2, 3, 0, 1).contiguous()                   # -> [C, 1, kh, kw]
        return F.conv2d(x, kk, stride=stride, groups=x.shape[1])

    @staticmethod
    def _swish(x):
        return x * torch.sigmoid(x)

    # -- pieces (any spatial size: tests/test_hf_efficientnet_golden.py feeds even-sized maps) ---------
    def preprocess(self, x):
        """Rescaling(1/255) -> Normalization (NCHW in, NCHW out)."""
        x = x * (1.0 / 255.0)
        return (x - self.w["normalization/mean"].view(1, -1, 1, 1)) / torch.clamp(
            torch.sqrt(self.w["normalization/variance"]), min=1e-7).view(1, -1, 1, 1)

    def stem(self, x):
        """ZeroPadding2D(correct_pad(3)) -> Conv 3x3 s2 valid -> BN -> swish (NCHW)."""
        (pt, pb), (pl, pr) = correct_pad(x.shape[2], x.shape[3], 3)
        x = F.pad(x, (pl, pr, pt, pb))
        return self._swish(self._bn(self._conv(x, "stem_conv/kernel", stride=2), "stem_bn"))

    def mbconv(self, x, block, tap=None):
        """One MBConv block (keras efficientnet.block(), inference) on an NCHW tensor; block = a BLOCKS row."""
        name, cin, cout, k, s, e = block
        p = "block" + name
        tap = tap or (lambda n, t: None)
        inp = x
        if e != 1:
            x = self._swish(self._bn(self._conv(x, p + "_expand_conv/kernel"), p + "_expand_bn"))
            tap(p + "_expand", x)
        if s == 2:
            (pt, pb), (pl, pr) = correct_pad(x.shape[2], x.shape[3], k)
        else:
            pt = pb = pl = pr = k // 2
        x = F.pad(x, (pl, pr, pt, pb))
        x = self._swish(self._bn(self._dwconv(x, p + "_dwconv/depthwise_kernel", s), p + "_bn"))
        tap(p + "_dw", x)
        se = x.mean(dim=(2, 3), keepdim=True)
        se = self._swish(self._conv(se, p + "_se_reduce/kernel", bias=p + "_se_reduce/bias"))
        se = torch.sigmoid(self._conv(se, p + "_se_expand/kernel", bias=p + "_se_expand/bias"))
        tap(p + "_gate", se[:, :, 0, 0])
        x = x * se
        x = self._bn(self._conv(x, p + "_project_conv/kernel"), p + "_project_bn")
        if s == 1 and cin == cout:
            x = x + inp                                           # drop-connect is identity at inference
        tap(p, x)
        return x

    def top(self, x):
        return self._swish(self._bn(self._conv(x, "top_conv/kernel"), "top_bn"))

    # -- forward -----------------------------------------------------------------------------------
    def forward(self, spec, taps=None):
        """spec: [B,49,40] or [B,49,40,1] (numpy or torch) -> embedding [B,1024] (torch, self.dtype).
        taps: optional dict filled with NHWC numpy copies of named stage outputs."""
        x = torch.as_tensor(np.asarray(spec)).to(self.dtype)
        if x.dim() == 4:
            x = x[..., 0]
        x = x[:, None]                                           # NCHW, C = 1

        def tap(name, t):
            if taps is not None:
                taps[name] = (t.permute(0, 2, 3, 1) if t.dim() == 4 else t).contiguous().numpy().copy()

        x = self.stem(self.preprocess(x))
        tap("stem", x)
        for block in BLOCKS:
            x = self.mbconv(x, block, tap)
        x = self.top(x)
        tap("top", x)
        x = x.mean(dim=(2, 3))
        tap("gap", x)
        x = torch.relu(x @ self.w["dense/kernel"] + self.w["dense/bias"])
        tap("dense", x)
        x = torch.relu(x @ self.w["dense_1/kernel"] + self.w["dense_1/bias"])
        tap("dense_1", x)
        x = x @ self.w["dense_2/kernel"] + self.w["dense_2/bias"]
        x = SELU_SCALE * torch.where(x > 0, x, SELU_ALPHA * torch.expm1(x))
        tap("dense_2", x)
        return x


def mac_count():
    """Multiply-accumulates per clip (for the roofline arithmetic)."""
    h, w = 25, 20
    total = {"stem": 25 * 20 * 9 * 32, "pointwise": 0, "depthwise": 0, "se": 0, "dense": 0}
    for name, cin, cout, k, s, e in BLOCKS:
        ce, se = cin * e, max(1, int(cin * 0.25))
        if e != 1:
            total["pointwise"] += h * w * cin * ce
        if s == 2:
            (pt, pb), (pl, pr) = correct_pad(h, w, k)
            h, w = (h + pt + pb - k) // 2 + 1, (w + pl + pr - k) // 2 + 1
        total["depthwise"] += h * w * k * k * ce
        total["se"] += 2 * ce * se
        total["pointwise"] += h * w * ce * cout
    total["pointwise"] += h * w * 320 * 1280
    total["dense"] = 1280 * 2048 + 2048 * 2048 + 2048 * 1024
    return total
