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

    # ------------------------------------------------------------------ the kernels' arithmetic, restated
    @torch.no_grad()
    def forward_engine_numerics(self, x):
        """The network as csrc/trunk.hpp + csrc/resnet.hip compute it — bf16 operands, fp32 accumulation — with a round-to-nearest-even
        to bf16 at exactly the points the kernels round (and nowhere else), so the HIP evaluator can be held to a TIGHT tolerance
        (tests/test_evaluator_gpu.py) instead of the loose bf16-vs-fp32 one:
          stem     int8 planes x (hi + lo) bf16 split of w * bn_scale, + shift, exact GELU              -> x  = bf16(.)
          block    a = bf16(relu(x s1 + t1)); h = bf16(relu(conv(a, bf16 w1) s2 + t2')); x = bf16((conv(h, bf16 w2) + b2) + x)
          heads    f = relu((conv(x, bf16 wh) + bh) fs + ft) in fp32; Dense / softmax / tanh in fp32
        Convolutions are summed in float64 and rounded to fp32 once (the MFMA's fp32 accumulation order differs from any host
        order by ~1e-6 relative; the test tolerance covers that and the rare bf16 roundings it flips).  Returns a dict of the head
        features, logits, pre-tanh value, policy and value."""
        bf = lambda t: t.float().to(torch.bfloat16).float()
        B = x.shape[0]

        def conv(a, w):                                  # a [B,H,W,Cin] float32 (bf16-representable), w [3,3,Cin,Cout] -> float32 accumulators
            y = F.conv2d(a.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=1)
            return y.permute(0, 2, 3, 1).float()
        s, t = self.stem_bn.affine()
        wv = (self.stem.weight * s).float()              # BN scale folded into the fp32 weights (resnet.hip stem_fragments)
        hi = bf(wv); lo = bf(wv - hi)
        xs = bf(F.gelu(conv(x.float(), hi.double() + lo.double()) + (self.stem.bias * s + t)))
        for b in self.blocks:
            s1, t1 = b.bn1.affine(); s2, t2 = b.bn2.affine()
            a = bf(F.relu(xs * s1 + t1))
            h = bf(F.relu(conv(a, bf(b.conv1.weight)) * s2 + (b.conv1.bias * s2 + t2)))
            xs = bf((conv(h, bf(b.conv2.weight)) + b.conv2.bias) + xs)
        out = {}
        z = {}
        for pre in ("p", "v"):
            c = getattr(self, pre + "_conv"); s0, t0 = getattr(self, pre + "_bn0").affine()
            f = F.relu((conv(xs, bf(c.weight)) + c.bias).reshape(B, -1) * s0 + t0)
            out[pre + "_feat"] = f
            d1, d2, d3 = getattr(self, pre + "_d1"), getattr(self, pre + "_d2"), getattr(self, pre + "_d3")
            s1, t1 = getattr(self, pre + "_bn1").affine()
            y = F.relu((f.double() @ d1.weight.double()).float() * s1 + (d1.bias * s1 + t1))
            y = (y.double() @ d2.weight.double() + d2.bias.double()).float()
            z[pre] = (y.double() @ d3.weight.double() + d3.bias.double()).float()
        out["logits"] = z["p"]; out["v_pre"] = z["v"].reshape(-1)
        if self.policy_head == "softmax":
            out["policy"] = torch.softmax(z["p"], -1)    # the kernel's softmax is fp32 (k_tail)
        elif self.policy_head == "stablemax":
            out["policy"] = stablemax(z["p"])
        else:
            out["policy"] = z["p"]
        out["value"] = torch.tanh(z["v"]).reshape(-1)
        return {k: v.numpy() for k, v in out.items()}

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


# =====================================================================================================================
# Gomoku and TicTacToe networks (same building blocks; exported as generic layer tensors for csrc/resnet.hip)
# =====================================================================================================================
def _conv_w(c):
    """[kh,kw,cin,cout] -> [kh*kw, cout, cin] float32 (tap-major, the layout every HIP conv kernel reads)."""
    k = c.weight.shape[0] * c.weight.shape[1]
    return c.weight.detach().permute(0, 1, 3, 2).reshape(k, c.weight.shape[3], c.weight.shape[2]).contiguous().numpy()


def _np(d):
    return {k: np.ascontiguousarray(v.detach().numpy() if isinstance(v, torch.Tensor) else v, np.float32) for k, v in d.items()}


class GomokuNet(nn.Module):
    """Gomoku/Build_Model.py:10-88.  Input [B,15,15,2]; stem Conv3x3 -> 256 (hard-coded, :21) + BN + ReLU; N pre-activation
    blocks of `num_filters` (the first one projects 256 -> num_filters with a 1x1 conv); policy: BN, ReLU, Conv3x3->32, BN,
    ReLU, Conv3x3->8, flatten(1800), BN, ReLU, Dense512, BN, ReLU, Dense225, softmax(f64); value: BN, ReLU, Conv3x3->32, BN,
    ReLU, Conv1x1->4, flatten(900), BN, ReLU, Dense256, BN, ReLU, Dense128, BN, ReLU, Dense1, tanh."""
    H, W, C, A = 15, 15, 2, 225

    def __init__(self, num_resnet_layers=10, num_filters=128, policy_head="softmax", seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.policy_head = policy_head
        self.stem = _ConvNHWC(2, 256, 3); self.stem_bn = _BN(256)
        blocks, cin = [], 256
        for _ in range(num_resnet_layers):
            blocks.append(ResNetBlock(cin, num_filters)); cin = num_filters
        self.blocks = nn.ModuleList(blocks)
        F = num_filters
        self.p_bn0 = _BN(F); self.p_c1 = _ConvNHWC(F, 32, 3); self.p_bn1 = _BN(32); self.p_c2 = _ConvNHWC(32, 8, 3)
        self.p_bn2 = _BN(1800); self.p_d1 = _Dense(1800, 512); self.p_bn3 = _BN(512); self.p_d2 = _Dense(512, 225)
        self.v_bn0 = _BN(F); self.v_c1 = _ConvNHWC(F, 32, 3); self.v_bn1 = _BN(32); self.v_c2 = _ConvNHWC(32, 4, 1)
        self.v_bn2 = _BN(900); self.v_d1 = _Dense(900, 256); self.v_bn3 = _BN(256); self.v_d2 = _Dense(256, 128)
        self.v_bn4 = _BN(128); self.v_d3 = _Dense(128, 1)

    randomize_bn = Connect4Net.randomize_bn

    def forward(self, x):
        x = F.relu(self.stem_bn(self.stem(x.float())))
        for b in self.blocks:
            x = b(x)
        B = x.shape[0]
        p = self.p_c2(F.relu(self.p_bn1(self.p_c1(F.relu(self.p_bn0(x)))))).reshape(B, -1)
        p = self.p_d2(F.relu(self.p_bn3(self.p_d1(F.relu(self.p_bn2(p))))))
        if self.policy_head == "softmax":
            p = torch.softmax(p.double(), -1).float()
        elif self.policy_head == "stablemax":
            p = stablemax(p)
        v = self.v_c2(F.relu(self.v_bn1(self.v_c1(F.relu(self.v_bn0(x)))))).reshape(B, -1)
        v = F.relu(self.v_bn3(self.v_d1(F.relu(self.v_bn2(v)))))
        v = torch.tanh(self.v_d3(F.relu(self.v_bn4(self.v_d2(v)))))
        return p, v

    @torch.no_grad()
    def forward_engine_numerics(self, x):
        """The Gomoku network as the HIP kernels compute it (bf16 roundings where they round, fp32 elsewhere, convolutions summed in
        float64) — see Connect4Net.forward_engine_numerics.  Rounding points: stem output x0 (k_stem_mfma, hi + lo weight split, ReLU);
        block 0 (k_block0): a0 = bf16(relu(x0 s1 + t1)) made in LDS, h = bf16(relu(conv1 s2 + t2')), x = bf16(conv2(h) + proj(x0) +
        (b2 + bp)) with conv2 and the 1x1 projection in ONE accumulator; blocks 1.. (k_trunk) as Connect4; heads (k_conv_head32):
        a = bf16(relu(x s0 + t0)), c1 = bf16(relu(conv s1 + t1')); then fp32: second head conv (k_conv_small) + flat BN + ReLU, Dense
        layers, softmax / tanh."""
        bf = lambda t: t.float().to(torch.bfloat16).float()
        B = x.shape[0]

        def conv(a, w, pad):
            y = F.conv2d(a.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=pad)
            return y.permute(0, 2, 3, 1)                 # float64: the caller rounds where the kernel does
        s, t = self.stem_bn.affine()
        wv = (self.stem.weight * s).float()
        hi = bf(wv); lo = bf(wv - hi)
        xs = bf(F.relu(conv(x.float(), hi.double() + lo.double(), 1).float() + (self.stem.bias * s + t)))
        for b in self.blocks:
            s1, t1 = b.bn1.affine(); s2, t2 = b.bn2.affine()
            a = bf(F.relu(xs * s1 + t1))
            h = bf(F.relu(conv(a, bf(b.conv1.weight), 1).float() * s2 + (b.conv1.bias * s2 + t2)))
            if b.proj is not None:                       # conv2 and the projection accumulate together; one fp32 bias (b2 + bp)
                acc = (conv(h, bf(b.conv2.weight), 1) + conv(xs, bf(b.proj.weight), 0)).float()
                xs = bf(acc + (b.conv2.bias + b.proj.bias))
            else:
                xs = bf((conv(h, bf(b.conv2.weight), 1).float() + b.conv2.bias) + xs)
        out = {}
        s0, t0 = self.p_bn0.affine(); s1, t1 = self.p_bn1.affine(); s2, t2 = self.p_bn2.affine()
        c1 = bf(F.relu(conv(bf(F.relu(xs * s0 + t0)), bf(self.p_c1.weight), 1).float() * s1 + (self.p_c1.bias * s1 + t1)))
        pf = F.relu((conv(c1, self.p_c2.weight, 1).float() + self.p_c2.bias).reshape(B, -1) * s2 + t2)
        s3, t3 = self.p_bn3.affine()
        y = F.relu((pf.double() @ self.p_d1.weight.double()).float() * s3 + (self.p_d1.bias * s3 + t3))
        logits = (y.double() @ self.p_d2.weight.double() + self.p_d2.bias.double()).float()
        s0, t0 = self.v_bn0.affine(); s1, t1 = self.v_bn1.affine(); s2, t2 = self.v_bn2.affine()
        c1 = bf(F.relu(conv(bf(F.relu(xs * s0 + t0)), bf(self.v_c1.weight), 1).float() * s1 + (self.v_c1.bias * s1 + t1)))
        vf = F.relu((conv(c1, self.v_c2.weight, 0).float() + self.v_c2.bias).reshape(B, -1) * s2 + t2)
        s3, t3 = self.v_bn3.affine(); s4, t4 = self.v_bn4.affine()
        y = F.relu((vf.double() @ self.v_d1.weight.double()).float() * s3 + (self.v_d1.bias * s3 + t3))
        y = F.relu((y.double() @ self.v_d2.weight.double()).float() * s4 + (self.v_d2.bias * s4 + t4))
        vpre = (y.double() @ self.v_d3.weight.double() + self.v_d3.bias.double()).float().reshape(-1)
        out["p_feat"] = pf; out["v_feat"] = vf; out["logits"] = logits; out["v_pre"] = vpre
        out["policy"] = torch.softmax(logits, -1) if self.policy_head == "softmax" else (stablemax(logits) if self.policy_head == "stablemax" else logits)
        out["value"] = torch.tanh(vpre)
        return {k: v.numpy() for k, v in out.items()}

    @torch.no_grad()
    def export_engine_weights(self):
        o = {}
        s, t = self.stem_bn.affine()
        o["stem.w"] = _conv_w(self.stem); o["stem.scale"] = s; o["stem.shift"] = self.stem.bias * s + t
        for i, b in enumerate(self.blocks):
            s1, t1 = b.bn1.affine(); s2, t2 = b.bn2.affine()
            o[f"block{i}.bn1.scale"] = s1; o[f"block{i}.bn1.shift"] = t1
            o[f"block{i}.conv1.w"] = _conv_w(b.conv1); o[f"block{i}.conv1.scale"] = s2; o[f"block{i}.conv1.shift"] = b.conv1.bias * s2 + t2
            o[f"block{i}.conv2.w"] = _conv_w(b.conv2); o[f"block{i}.conv2.bias"] = b.conv2.bias
            if b.proj is not None:
                o[f"block{i}.proj.w"] = _conv_w(b.proj); o[f"block{i}.proj.bias"] = b.proj.bias
        # heads: relu(bn0(x)) per head; first convs of both heads as ONE 3x3 conv with 64 outputs (0-31 policy, 32-63 value)
        sp, tp = self.p_bn0.affine(); sv, tv = self.v_bn0.affine()
        o["p.bn0.scale"] = sp; o["p.bn0.shift"] = tp; o["v.bn0.scale"] = sv; o["v.bn0.shift"] = tv
        s1p, t1p = self.p_bn1.affine(); s1v, t1v = self.v_bn1.affine()
        o["p.c1.w"] = _conv_w(self.p_c1); o["p.c1.scale"] = s1p; o["p.c1.shift"] = self.p_c1.bias * s1p + t1p
        o["v.c1.w"] = _conv_w(self.v_c1); o["v.c1.scale"] = s1v; o["v.c1.shift"] = self.v_c1.bias * s1v + t1v
        s2p, t2p = self.p_bn2.affine(); s2v, t2v = self.v_bn2.affine()
        o["p.c2.w"] = _conv_w(self.p_c2); o["p.c2.bias"] = self.p_c2.bias; o["p.bn2.scale"] = s2p; o["p.bn2.shift"] = t2p
        o["v.c2.w"] = _conv_w(self.v_c2); o["v.c2.bias"] = self.v_c2.bias; o["v.bn2.scale"] = s2v; o["v.bn2.shift"] = t2v
        s3p, t3p = self.p_bn3.affine()
        o["p.d1.w"] = self.p_d1.weight; o["p.d1.scale"] = s3p; o["p.d1.shift"] = self.p_d1.bias * s3p + t3p
        o["p.d2.w"] = self.p_d2.weight; o["p.d2.bias"] = self.p_d2.bias
        s3v, t3v = self.v_bn3.affine(); s4v, t4v = self.v_bn4.affine()
        o["v.d1.w"] = self.v_d1.weight; o["v.d1.scale"] = s3v; o["v.d1.shift"] = self.v_d1.bias * s3v + t3v
        o["v.d2.w"] = self.v_d2.weight; o["v.d2.scale"] = s4v; o["v.d2.shift"] = self.v_d2.bias * s4v + t4v
        o["v.d3.w"] = self.v_d3.weight; o["v.d3.bias"] = self.v_d3.bias
        return _np(o)


class TicTacToeNet(nn.Module):
    """TicTacToe/Build_Model.py:8-69.  Input [B,3,3,2]; stem Conv5x5 -> 128 + BN + GELU; N blocks of 64 filters (first one
    projects 128 -> 64); policy: Conv1x1->8, BN, flatten(72), Dense128, ReLU, Dense64, Dense9, softmax; value: Conv1x1->4, BN,
    flatten(36), Dense128, Dense64, ReLU, Dense1, tanh."""
    H, W, C, A = 3, 3, 2, 9

    def __init__(self, num_resnet_layers=2, policy_head="softmax", seed=0, final_std=0.05):
        super().__init__()
        torch.manual_seed(seed)
        self.policy_head = policy_head
        self.stem = _ConvNHWC(2, 128, 5); self.stem_bn = _BN(128)
        blocks, cin = [], 128
        for _ in range(num_resnet_layers):
            blocks.append(ResNetBlock(cin, 64)); cin = 64
        self.blocks = nn.ModuleList(blocks)
        self.p_conv = _ConvNHWC(cin, 8, 1); self.p_bn = _BN(8); self.p_d1 = _Dense(72, 128); self.p_d2 = _Dense(128, 64)
        self.p_d3 = _Dense(64, 9, zeros=True)
        self.v_conv = _ConvNHWC(cin, 4, 1); self.v_bn = _BN(4); self.v_d1 = _Dense(36, 128); self.v_d2 = _Dense(128, 64)
        self.v_d3 = _Dense(64, 1, zeros=True)
        if final_std:
            g = torch.Generator().manual_seed(seed + 1)
            with torch.no_grad():
                self.p_d3.weight.copy_(final_std * torch.randn(64, 9, generator=g)); self.v_d3.weight.copy_(final_std * torch.randn(64, 1, generator=g))

    randomize_bn = Connect4Net.randomize_bn

    def forward(self, x):
        x = F.gelu(self.stem_bn(self.stem(x.float())))
        for b in self.blocks:
            x = b(x)
        B = x.shape[0]
        p = self.p_bn(self.p_conv(x)).reshape(B, -1)
        p = self.p_d3(self.p_d2(F.relu(self.p_d1(p))))
        if self.policy_head == "softmax":
            p = torch.softmax(p, -1)
        elif self.policy_head == "stablemax":
            p = stablemax(p)
        v = self.v_bn(self.v_conv(x)).reshape(B, -1)
        v = torch.tanh(self.v_d3(F.relu(self.v_d2(self.v_d1(v)))))
        return p, v

    @torch.no_grad()
    def forward_engine_numerics(self, x):
        """The TicTacToe network as the HIP kernels compute it (netops.hpp k_stem_generic / k_conv_direct / k_dense: fp32 weights, bf16
        activations between layers, fp32 everywhere else; convolutions summed in float64 here) — see Connect4Net.forward_engine_numerics.
        Rounding points: stem output x0 = bf16(gelu(.)) and, from the UNROUNDED value, a0 = bf16(relu(bn1(.))); per block h =
        bf16(relu(bn2(conv1(a)))), block 0's skip path x = bf16(proj(x0) + bp), x = bf16(conv2(h) + b2 + x) and again from the unrounded sum
        the next block's operand a = bf16(relu(bn1'(.))); the heads' 1x1 convolutions read the bf16 x and stay in fp32 from there on."""
        bf = lambda t: t.float().to(torch.bfloat16).float()
        B = x.shape[0]

        def conv(a, w, pad):
            return F.conv2d(a.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=pad).permute(0, 2, 3, 1)
        s, t = self.stem_bn.affine()
        v = F.gelu(conv(x.float(), self.stem.weight, 2).float() * s + (self.stem.bias * s + t))
        xs = bf(v)
        s1, t1 = self.blocks[0].bn1.affine()
        a = bf(F.relu(v * s1 + t1))
        for i, b in enumerate(self.blocks):
            s2, t2 = b.bn2.affine()
            h = bf(F.relu(conv(a, b.conv1.weight, 1).float() * s2 + (b.conv1.bias * s2 + t2)))
            if b.proj is not None:
                xs = bf(conv(xs, b.proj.weight, 0).float() + b.proj.bias)
            v = (conv(h, b.conv2.weight, 1).float() + b.conv2.bias) + xs
            xs = bf(v)
            if i + 1 < len(self.blocks):
                s1, t1 = self.blocks[i + 1].bn1.affine()
                a = bf(F.relu(v * s1 + t1))
        sp, tp = self.p_bn.affine(); sv, tv = self.v_bn.affine()
        pf = (conv(xs, self.p_conv.weight, 0).float() * sp + (self.p_conv.bias * sp + tp)).reshape(B, -1)
        vf = (conv(xs, self.v_conv.weight, 0).float() * sv + (self.v_conv.bias * sv + tv)).reshape(B, -1)
        d = lambda y, layer: (y.double() @ layer.weight.double() + layer.bias.double()).float()
        logits = d(d(F.relu(d(pf, self.p_d1)), self.p_d2), self.p_d3)
        vpre = d(F.relu(d(d(vf, self.v_d1), self.v_d2)), self.v_d3).reshape(-1)
        out = dict(p_feat=pf, v_feat=vf, logits=logits, v_pre=vpre, value=torch.tanh(vpre))
        out["policy"] = torch.softmax(logits, -1) if self.policy_head == "softmax" else (stablemax(logits) if self.policy_head == "stablemax" else logits)
        return {k: v.numpy() for k, v in out.items()}

    @torch.no_grad()
    def export_engine_weights(self):
        o = {}
        s, t = self.stem_bn.affine()
        o["stem.w"] = _conv_w(self.stem); o["stem.scale"] = s; o["stem.shift"] = self.stem.bias * s + t
        for i, b in enumerate(self.blocks):
            s1, t1 = b.bn1.affine(); s2, t2 = b.bn2.affine()
            o[f"block{i}.bn1.scale"] = s1; o[f"block{i}.bn1.shift"] = t1
            o[f"block{i}.conv1.w"] = _conv_w(b.conv1); o[f"block{i}.conv1.scale"] = s2; o[f"block{i}.conv1.shift"] = b.conv1.bias * s2 + t2
            o[f"block{i}.conv2.w"] = _conv_w(b.conv2); o[f"block{i}.conv2.bias"] = b.conv2.bias
            if b.proj is not None:
                o[f"block{i}.proj.w"] = _conv_w(b.proj); o[f"block{i}.proj.bias"] = b.proj.bias
        sp, tp = self.p_bn.affine(); sv, tv = self.v_bn.affine()
        o["p.c.w"] = _conv_w(self.p_conv); o["p.c.scale"] = sp; o["p.c.shift"] = self.p_conv.bias * sp + tp
        o["v.c.w"] = _conv_w(self.v_conv); o["v.c.scale"] = sv; o["v.c.shift"] = self.v_conv.bias * sv + tv
        for pre in ("p", "v"):
            for k in (1, 2, 3):
                d = getattr(self, f"{pre}_d{k}")
                o[f"{pre}.d{k}.w"] = d.weight; o[f"{pre}.d{k}.bias"] = d.bias
        return _np(o)


NETS = {"Connect4": Connect4Net, "Gomoku": GomokuNet, "TicTacToe": TicTacToeNet}
