"""Policy/value network of the evaluator — PyTorch restatement of the reference's Keras models and the
weight export consumed by the HIP evaluator (csrc/resnet.hip).

Reference definitions (file:line under /root/reference):
  ResNet_Block            Net/ResNet/ResNet_Block.py:5-41  pre-activation: BN -> ReLU -> Conv3x3 -> BN -> ReLU -> Conv3x3 (+ residual,
                                                            1x1 projection iff C_in != filters)
  Connect4 build_model    Connect4/Build_Model.py:10-88    stem Conv3x3->128 + BN + GELU; N blocks; policy Conv3x3->8, flatten, BN, ReLU,
                                                            Dense128, BN, ReLU, Dense64, Dense7 (softmax / linear); value Conv3x3->8, flatten,
                                                            BN, ReLU, Dense128, BN, ReLU, Dense64, Dense1, tanh
  Stablemax               Net/Stablemax.py:3-11
Keras defaults restated explicitly: BatchNormalization eps = 1e-3, NHWC, "same" padding, exact (erf) GELU,
he_normal = truncated normal with stddev sqrt(2 / fan_in) / 0.87962566.

TensorFlow is not installable here and the reference ships no weights or ONNX files, so the evaluator's
numerics are "parity unpinned" against Keras; this module is the fp32 reference the HIP
kernels are checked against (bf16 tolerance written in tests/test_evaluator_gpu.py).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-3


def _he_normal_(w, fan_in):
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    return nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std)


class _BN(nn.Module):
    """Inference-mode BatchNormalization over the last (channel / feature) axis."""

    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n)); self.bias = nn.Parameter(torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n)); self.register_buffer("running_var", torch.ones(n))

    def affine(self):
        s = self.weight / torch.sqrt(self.running_var + BN_EPS)
        return s, self.bias - self.running_mean * s

    def forward(self, x):
        s, t = self.affine()
        return x * s + t

    def randomize(self, g):
        with torch.no_grad():
            n = self.weight.numel()
            self.weight.copy_(1.0 + 0.2 * torch.randn(n, generator=g)); self.bias.copy_(0.1 * torch.randn(n, generator=g))
            self.running_mean.copy_(0.1 * torch.randn(n, generator=g)); self.running_var.copy_(0.5 + torch.rand(n, generator=g))


class _ConvNHWC(nn.Module):
    """Conv2D on NHWC tensors, weight stored Keras-style [kh, kw, cin, cout]."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.k = k
        self.weight = nn.Parameter(_he_normal_(torch.empty(k, k, cin, cout), k * k * cin))
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x):
        y = F.conv2d(x.permute(0, 3, 1, 2), self.weight.permute(3, 2, 0, 1), self.bias, padding=self.k // 2)
        return y.permute(0, 2, 3, 1)


class ResNetBlock(nn.Module):
    def __init__(self, cin, filters):
        super().__init__()
        self.bn1 = _BN(cin); self.conv1 = _ConvNHWC(cin, filters, 3)
        self.bn2 = _BN(filters); self.conv2 = _ConvNHWC(filters, filters, 3)
        self.proj = _ConvNHWC(cin, filters, 1) if cin != filters else None

    def forward(self, x):
        res = x if self.proj is None else self.proj(x)
        y = self.conv1(F.relu(self.bn1(x)))
        y = self.conv2(F.relu(self.bn2(y)))
        return y + res


class _Dense(nn.Module):
    def __init__(self, cin, cout, zeros=False):
        super().__init__()
        w = torch.zeros(cin, cout) if zeros else _he_normal_(torch.empty(cin, cout), cin)
        self.weight = nn.Parameter(w); self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x):
        return x @ self.weight + self.bias


def stablemax(x):
    s = torch.where(x >= 0, x + 1.0, 1.0 / (1.0 - x))
    return s / s.sum(-1, keepdim=True)


class Connect4Net(nn.Module):
    """Connect4/Build_Model.py:10-88.  Input int8/float [B, 6, 7, 4]; outputs policy [B, 7], value [B, 1]."""
    H, W, C, A = 6, 7, 4, 7

    def __init__(self, num_resnet_layers=6, num_filters=128, policy_head="softmax", seed=0, final_std=0.05):
        super().__init__()
        torch.manual_seed(seed)
        self.policy_head = policy_head
        self.stem = _ConvNHWC(4, 128, 3); self.stem_bn = _BN(128)        # stem width is hard-coded to 128 (:22)
        blocks, cin = [], 128
        for _ in range(num_resnet_layers):
            blocks.append(ResNetBlock(cin, num_filters)); cin = num_filters
        self.blocks = nn.ModuleList(blocks)
        flat = self.H * self.W * 8
        self.p_conv = _ConvNHWC(cin, 8, 3); self.p_bn0 = _BN(flat); self.p_d1 = _Dense(flat, 128); self.p_bn1 = _BN(128)
        self.p_d2 = _Dense(128, 64); self.p_d3 = _Dense(64, self.A, zeros=True)
        self.v_conv = _ConvNHWC(cin, 8, 3); self.v_bn0 = _BN(flat); self.v_d1 = _Dense(flat, 128); self.v_bn1 = _BN(128)
        self.v_d2 = _Dense(128, 64); self.v_d3 = _Dense(64, 1, zeros=True)
        if final_std:   # the reference zero-inits the last Dense layers; synthetic runs need distinct priors
            g = torch.Generator().manual_seed(seed + 1)
            with torch.no_grad():
                self.p_d3.weight.copy_(final_std * torch.randn(64, self.A, generator=g))
                self.v_d3.weight.copy_(final_std * torch.randn(64, 1, generator=g))

    def randomize_bn(self, seed=123):
        g = torch.Generator().manual_seed(seed)
        for m in self.modules():
            if isinstance(m, _BN):
                m.randomize(g)
        return self

    def forward(self, x):
        x = x.float()
        x = F.gelu(self.stem_bn(self.stem(x)))
        for b in self.blocks:
            x = b(x)
        B = x.shape[0]
        p = F.relu(self.p_bn0(self.p_conv(x).reshape(B, -1)))
        p = self.p_d3(self.p_d2(F.relu(self.p_bn1(self.p_d1(p)))))
        if self.policy_head == "softmax":
            p = torch.softmax(p.double(), -1).float()          # Activation("softmax", dtype="float64") (:60)
        elif self.policy_head == "stablemax":
            p = stablemax(p)
        v = F.relu(self.v_bn0(self.v_conv(x).reshape(B, -1)))
        v = torch.tanh(self.v_d3(self.v_d2(F.relu(self.v_bn1(self.v_d1(v))))))
        return p, v

    # ------------------------------------------------------------------ export for csrc/resnet.hip
    @torch.no_grad()
    def export_engine_weights(self):
        """Folded inference tensors, float32, in the layouts the HIP kernels read:
        conv weights [tap = ky*3+kx][cout][cin]; BN as per-channel (scale, shift)."""
        out = {}

        def conv_w(c):   # [kh,kw,cin,cout] -> [9, cout, cin]
            return c.weight.permute(0, 1, 3, 2).reshape(9, c.weight.shape[3], c.weight.shape[2]).contiguous().numpy()

        s, t = self.stem_bn.affine()
        out["stem.w"] = conv_w(self.stem); out["stem.scale"] = s.numpy(); out["stem.shift"] = (self.stem.bias * s + t).numpy()
        for i, b in enumerate(self.blocks):
            assert b.proj is None, "projection blocks are not exported yet"
            s1, t1 = b.bn1.affine(); s2, t2 = b.bn2.affine()
            out[f"block{i}.bn1.scale"] = s1.numpy(); out[f"block{i}.bn1.shift"] = t1.numpy()
            out[f"block{i}.conv1.w"] = conv_w(b.conv1)
            out[f"block{i}.conv1.scale"] = s2.numpy(); out[f"block{i}.conv1.shift"] = (b.conv1.bias * s2 + t2).numpy()
            out[f"block{i}.conv2.w"] = conv_w(b.conv2); out[f"block{i}.conv2.bias"] = b.conv2.bias.numpy()
        # heads: one 3x3 conv with 16 real output channels (0-7 policy, 8-15 value), padded to 32 for the MFMA tile
        hw = np.zeros((9, 32, self.blocks[-1].conv2.weight.shape[3] if len(self.blocks) else 128), np.float32)
        hw[:, 0:8] = conv_w(self.p_conv); hw[:, 8:16] = conv_w(self.v_conv)
        hb = np.zeros(32, np.float32); hb[0:8] = self.p_conv.bias.numpy(); hb[8:16] = self.v_conv.bias.numpy()
        out["heads.conv.w"] = hw; out["heads.conv.bias"] = hb
        for pre in ("p", "v"):
            s0, t0 = getattr(self, pre + "_bn0").affine(); s1, t1 = getattr(self, pre + "_bn1").affine()
            d1, d2, d3 = getattr(self, pre + "_d1"), getattr(self, pre + "_d2"), getattr(self, pre + "_d3")
            out[f"{pre}.bn0.scale"] = s0.numpy(); out[f"{pre}.bn0.shift"] = t0.numpy()       # over the flat (cell*8 + c) index
            out[f"{pre}.d1.w"] = d1.weight.numpy(); out[f"{pre}.d1.scale"] = s1.numpy()      # relu(bn1(d1(x))) folded
            out[f"{pre}.d1.shift"] = (d1.bias * s1 + t1).numpy()
            out[f"{pre}.d2.w"] = d2.weight.numpy(); out[f"{pre}.d2.bias"] = d2.bias.numpy()
            out[f"{pre}.d3.w"] = d3.weight.numpy(); out[f"{pre}.d3.bias"] = d3.bias.numpy()
        return {k: np.ascontiguousarray(v, np.float32) for k, v in out.items()}


def flops_per_position(num_blocks=6, filters=128, H=6, W=7):
    """Algorithmic FLOPs of one Connect4 evaluation (2 x MACs), by layer class."""
    hw = H * W
    trunk = num_blocks * 2 * 2 * hw * 9 * filters * filters
    stem = 2 * hw * 9 * 4 * 128
    heads = 2 * hw * 9 * filters * 16 + 2 * 2 * (hw * 8 * 128 + 128 * 64) + 2 * (64 * 7 + 64)
    return dict(trunk=trunk, stem=stem, heads=heads, total=trunk + stem + heads)
