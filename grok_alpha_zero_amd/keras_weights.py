"""Import the reference's Keras `model.weights.h5` (model.save_weights, <Game>/main.py:119,135,159-160) into the fp32 PyTorch
restatement of the network (net.py), from where `export_engine_weights()` feeds the HIP evaluator.  Needs libhdf5 only (h5io.py).

File layout (Keras 3 `saving_lib`): every layer of the functional model is a group `layers/<snake_case(class name)>[_<k>]`, numbered
per class in `model.layers` order; its variables are `vars/0`, `vars/1`, ... in creation order (Conv2D / Dense: kernel, bias;
BatchNormalization: gamma, beta, moving_mean, moving_variance); a custom layer's sub-layers hang below it by ATTRIBUTE name
(`layers/res_net__block_3/conv1/vars/0`, Net/ResNet/ResNet_Block.py:12-20).  Kernels are already in the layout net.py uses
([kh, kw, cin, cout] and [in, out]).

`model.layers` of a functional model is sorted by decreasing depth (distance to the branch's output), ties in the order the graph
walk from outputs [policy, value] first met the layer — `_merge_heads` below restates that rule; for Connect4 and TicTacToe the two
heads have the same length and simply alternate policy, value; for Gomoku (14 vs 17 layers) they interleave from the outputs
backwards.  PARITY UNPINNED: the reference ships no weight file and TensorFlow
is not installed here, so this mapping is checked only against files written by `save_keras_style` (same layout rules).
"""
import re

import numpy as np
import torch

from . import h5io
from .net import Connect4Net, GomokuNet, TicTacToeNet

_VARS = {"conv": ("weight", "bias"), "dense": ("weight", "bias"), "bn": ("weight", "bias", "running_mean", "running_var")}


def _merge_heads(policy, value):
    """`model.layers` order of two heads that branch off the trunk (keras/src/ops/function.py map_graph): operations sorted by
    decreasing depth — depth counts layers to the OUTPUT of the branch, so heads of unequal length interleave from their ends — and,
    at equal depth, in the order the graph walk from outputs [policy, value] first met them (policy first).  Chains are lists of
    (keras class, kind, module) from the trunk to the output; weightless layers (Activation, Reshape) count with kind None."""
    ops = [(len(policy) - 1 - j, 0, j, e) for j, e in enumerate(policy)] + [(len(value) - 1 - j, 1, j, e) for j, e in enumerate(value)]
    ops.sort(key=lambda t: (-t[0], t[1]))
    return [e for _, _, _, e in ops if e[1] is not None]


_ACT = ("activation", None, None)
_RESH = ("reshape", None, None)


def _trunk(net):
    return [("conv2d", "conv", net.stem), ("batch_normalization", "bn", net.stem_bn)] + [("res_net__block", "block", b) for b in net.blocks]


def _connect4_order(net):
    """Connect4/Build_Model.py:19-80"""
    p = [("conv2d", "conv", net.p_conv), _RESH, ("batch_normalization", "bn", net.p_bn0), _ACT, ("dense", "dense", net.p_d1),
         ("batch_normalization", "bn", net.p_bn1), _ACT, ("dense", "dense", net.p_d2), ("dense", "dense", net.p_d3), _ACT]
    v = [("conv2d", "conv", net.v_conv), _RESH, ("batch_normalization", "bn", net.v_bn0), _ACT, ("dense", "dense", net.v_d1),
         ("batch_normalization", "bn", net.v_bn1), _ACT, ("dense", "dense", net.v_d2), ("dense", "dense", net.v_d3), _ACT]
    return _trunk(net) + _merge_heads(p, v)


def _gomoku_order(net):
    """Gomoku/Build_Model.py:21-86 (heads of 14 and 17 layers: they interleave from the outputs backwards)"""
    p = [("batch_normalization", "bn", net.p_bn0), _ACT, ("conv2d", "conv", net.p_c1), ("batch_normalization", "bn", net.p_bn1), _ACT,
         ("conv2d", "conv", net.p_c2), _RESH, ("batch_normalization", "bn", net.p_bn2), _ACT, ("dense", "dense", net.p_d1),
         ("batch_normalization", "bn", net.p_bn3), _ACT, ("dense", "dense", net.p_d2), _ACT]
    v = [("batch_normalization", "bn", net.v_bn0), _ACT, ("conv2d", "conv", net.v_c1), ("batch_normalization", "bn", net.v_bn1), _ACT,
         ("conv2d", "conv", net.v_c2), _RESH, ("batch_normalization", "bn", net.v_bn2), _ACT, ("dense", "dense", net.v_d1),
         ("batch_normalization", "bn", net.v_bn3), _ACT, ("dense", "dense", net.v_d2), ("batch_normalization", "bn", net.v_bn4), _ACT,
         ("dense", "dense", net.v_d3), _ACT]
    return _trunk(net) + _merge_heads(p, v)


def _tictactoe_order(net):
    """TicTacToe/Build_Model.py:17-51"""
    p = [("conv2d", "conv", net.p_conv), ("batch_normalization", "bn", net.p_bn), _RESH, ("dense", "dense", net.p_d1), _ACT,
         ("dense", "dense", net.p_d2), ("dense", "dense", net.p_d3), _ACT]
    v = [("conv2d", "conv", net.v_conv), ("batch_normalization", "bn", net.v_bn), _RESH, ("dense", "dense", net.v_d1),
         ("dense", "dense", net.v_d2), _ACT, ("dense", "dense", net.v_d3), _ACT]
    return _trunk(net) + _merge_heads(p, v)


ORDER = {Connect4Net: _connect4_order, GomokuNet: _gomoku_order, TicTacToeNet: _tictactoe_order}


def _groups(order):
    """-> [(group path under layers/, kind, module)] with Keras' per-class counters; blocks expand to their sub-layers"""
    seen, out = {}, []
    for cls, kind, mod in order:
        k = seen.get(cls, 0); seen[cls] = k + 1
        g = cls if k == 0 else f"{cls}_{k}"
        if kind == "block":
            out += [(f"{g}/bn1", "bn", mod.bn1), (f"{g}/conv1", "conv", mod.conv1), (f"{g}/bn2", "bn", mod.bn2), (f"{g}/conv2", "conv", mod.conv2)]
            if mod.proj is not None:
                out.append((f"{g}/residual_conv", "conv", mod.proj))
        else:
            out.append((g, kind, mod))
    return out


def save_keras_style(net, path):
    """Write `net` in the layout described above (used by the tests and to hand weights back to a Keras checkpoint)."""
    with h5io.H5File(path, "w") as f:
        for g, kind, mod in _groups(ORDER[type(net)](net)):
            for i, attr in enumerate(_VARS[kind]):
                f.create_dataset(f"layers/{g}/vars/{i}", getattr(mod, attr).detach().cpu().numpy().astype(np.float32))


def load_keras_weights(path, net):
    """Fill `net` (e.g. Connect4Net(num_resnet_layers, num_filters)) from a Keras weights file; returns `net`.
    Raises KeyError / ValueError naming the group when a layer is missing or a shape does not match."""
    with h5io.H5File(path, "r") as f:
        have = set(f.walk("layers"))
        groups = _groups(ORDER[type(net)](net))
        with torch.no_grad():
            for g, kind, mod in groups:
                for i, attr in enumerate(_VARS[kind]):
                    name = f"layers/{g}/vars/{i}"
                    if name not in have:
                        raise KeyError(f"{path}: {name} not found (layers present: {sorted({re.sub('/vars/.*', '', h) for h in have})[:8]} ...)")
                    arr = f.read(name)
                    dst = getattr(mod, attr)
                    if tuple(arr.shape) != tuple(dst.shape):
                        raise ValueError(f"{name}: shape {tuple(arr.shape)} does not match the network's {tuple(dst.shape)}")
                    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr, np.float32)))
    return net
