"""Training-mode restatement of the embedding network (groundwork for SURVEY.md section 8f-4,
`backprop_into_embedding=True`, multilingual_kws/embedding/transfer_learning.py:94-112: the reference un-freezes the
whole nested base model, so Keras runs it with training=True -- BatchNormalization on batch statistics with
moving-average updates, and EfficientNet's per-block drop-connect (Dropout with noise_shape (None,1,1,1),
rate = drop_connect_rate * block_index / 16, keras/applications/efficientnet.py, TF 2.7)).

TEST INFRASTRUCTURE ONLY, like the rest of oracle/.  PARITY UNPINNED BY THE REFERENCE.
Gradients come from torch.autograd over the same F.conv2d formulation as efficientnet_oracle.EmbeddingOracle;
tests/test_oracle_train.py checks eval-mode equality with that oracle, batch-norm statistics against a hand
computation, and autograd against central finite differences in float64.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .efficientnet_oracle import BLOCKS, BN_EPS, SELU_ALPHA, SELU_SCALE, correct_pad, split_blob

BN_MOMENTUM = 0.99              # Keras BatchNormalization default (EfficientNet passes no momentum)
DROP_CONNECT_RATE = 0.2         # EfficientNetB0 default


class TrainableEmbeddingOracle:
    def __init__(self, blob, dtype=torch.float64):
        self.dtype = dtype
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in split_blob(blob).items()}
        # Keras trainable variables: everything except BN moving statistics and the Normalization layer's constants
        self.trainable = [k for k in self.w if not (k.endswith("moving_mean") or k.endswith("moving_variance") or k.startswith("normalization/"))]
        for k in self.trainable:
            self.w[k].requires_grad_(True)
        self.new_moving = {}        # filled by a training-mode forward: the moving statistics after one update

    def zero_grad(self):
        for k in self.trainable:
            self.w[k].grad = None

    def _bn(self, x, p, training):
        w = self.w
        g, b = w[p + "/gamma"], w[p + "/beta"]
        if training:
            mean = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)               # normalisation uses the biased batch variance
            with torch.no_grad():
                # Keras BatchNormalization on 4-D inputs takes the fused path, whose moving-variance update uses the
                # Bessel-corrected batch variance (N/(N-1), N = B*H*W) -- tf.compat.v1.nn.fused_batch_norm semantics
                n = x.shape[0] * x.shape[2] * x.shape[3]
                var_u = var * (n / max(n - 1, 1))
                self.new_moving[p + "/moving_mean"] = BN_MOMENTUM * w[p + "/moving_mean"] + (1 - BN_MOMENTUM) * mean
                self.new_moving[p + "/moving_variance"] = BN_MOMENTUM * w[p + "/moving_variance"] + (1 - BN_MOMENTUM) * var_u
        else:
            mean, var = w[p + "/moving_mean"], w[p + "/moving_variance"]
        inv = g / torch.sqrt(var + BN_EPS)
        return (x - mean.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    def _conv(self, x, name, stride=1, bias=None):
        k = self.w[name].permute(3, 2, 0, 1)
        return F.conv2d(x, k, bias=None if bias is None else self.w[bias], stride=stride)

    def _dwconv(self, x, name, stride):
        kk = self.w[name].permute(2, 3, 0, 1)
        return F.conv2d(x, kk, stride=stride, groups=x.shape[1])

    def forward(self, spec, training=False, drop_masks=None):
        """spec [B,49,40(,1)] -> embedding [B,1024] with autograd history.
        drop_masks: {block name: bool tensor [B]} of KEPT samples for the residual blocks (training only); None = keep all
        (the test default: Philox draws of the reference cannot be reproduced)."""
        sw = lambda t: t * torch.sigmoid(t)
        x = torch.as_tensor(np.asarray(spec)).to(self.dtype)
        if x.dim() == 4:
            x = x[..., 0]
        x = x[:, None] * (1.0 / 255.0)
        x = (x - self.w["normalization/mean"].view(1, -1, 1, 1)) / torch.clamp(torch.sqrt(self.w["normalization/variance"]), min=1e-7).view(1, -1, 1, 1)
        (pt, pb), (pl, pr) = correct_pad(x.shape[2], x.shape[3], 3)
        x = sw(self._bn(self._conv(F.pad(x, (pl, pr, pt, pb)), "stem_conv/kernel", 2), "stem_bn", training))
        for bi, (name, cin, cout, k, s, e) in enumerate(BLOCKS):
            p = "block" + name
            inp = x
            if e != 1:
                x = sw(self._bn(self._conv(x, p + "_expand_conv/kernel"), p + "_expand_bn", training))
            if s == 2:
                (pt, pb), (pl, pr) = correct_pad(x.shape[2], x.shape[3], k)
            else:
                pt = pb = pl = pr = k // 2
            x = sw(self._bn(self._dwconv(F.pad(x, (pl, pr, pt, pb)), p + "_dwconv/depthwise_kernel", s), p + "_bn", training))
            se = x.mean(dim=(2, 3), keepdim=True)
            se = sw(self._conv(se, p + "_se_reduce/kernel", bias=p + "_se_reduce/bias"))
            se = torch.sigmoid(self._conv(se, p + "_se_expand/kernel", bias=p + "_se_expand/bias"))
            x = self._bn(self._conv(x * se, p + "_project_conv/kernel"), p + "_project_bn", training)
            if s == 1 and cin == cout:
                if training and drop_masks is not None and name in drop_masks:
                    rate = DROP_CONNECT_RATE * bi / len(BLOCKS)
                    keep = torch.as_tensor(drop_masks[name]).to(self.dtype).view(-1, 1, 1, 1)
                    x = x * keep / (1.0 - rate)
                x = x + inp
        x = sw(self._bn(self._conv(x, "top_conv/kernel"), "top_bn", training))
        x = x.mean(dim=(2, 3))
        x = torch.relu(x @ self.w["dense/kernel"] + self.w["dense/bias"])
        x = torch.relu(x @ self.w["dense_1/kernel"] + self.w["dense_1/bias"])
        x = x @ self.w["dense_2/kernel"] + self.w["dense_2/bias"]
        return SELU_SCALE * torch.where(x > 0, x, SELU_ALPHA * torch.expm1(x))

    def grads(self):
        return {k: (self.w[k].grad.detach().numpy().copy() if self.w[k].grad is not None else None) for k in self.trainable}
