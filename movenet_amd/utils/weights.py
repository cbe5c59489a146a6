"""Deterministic weight recipe shared by tests, bench.py and smoke().

The golden fixtures under ``tests/golden`` do not store weights: they store the
recipe arguments (seed, gains) and a SHA-256 of the resulting tensors, and the
tests regenerate the weights on whatever box they run on.  The recipe therefore
uses only ``numpy.random.default_rng`` (PCG64, bit-reproducible across
platforms) and float64 -> float32 rounding.

Key names and shapes follow the reference checkpoint layout
(/root/reference/movenet/wavenet.py:94-123, movenet/modules.py:19-26, :36-43,
:52-65, :136-137).
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch

VIDEO_KERNEL_HW = (64, 64)
UPSAMPLE_KERNEL = 10
N_UPSAMPLE = 3


def parameter_shapes(
    layer_size: int,
    stack_size: int,
    input_channels: int,
    residual_channels: int = 16,
    skip_channels: int = 16,
    context_in_channels: int = 1,
) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {state_dict key: shape} exactly as the reference registers them."""
    Q, C, K = input_channels, residual_channels, skip_channels
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    shapes["video_conv.weight"] = (C, context_in_channels, 1) + VIDEO_KERNEL_HW
    shapes["video_conv.bias"] = (C,)
    for i in range(N_UPSAMPLE):
        shapes[f"video_transpose.{i}.weight"] = (C, C, UPSAMPLE_KERNEL)
        shapes[f"video_transpose.{i}.bias"] = (C,)
    shapes["causal_conv.conv.weight"] = (C, Q, 2)
    for l in range(layer_size * stack_size):
        p = f"residual_conv_stack.conv_layers.{l}."
        shapes[p + "conv_filter.conv.weight"] = (C, C, 2)
        shapes[p + "conv_gate.conv.weight"] = (C, C, 2)
        shapes[p + "context_conv_filter.weight"] = (C, C, 1)
        shapes[p + "context_conv_filter.bias"] = (C,)
        shapes[p + "context_conv_gate.weight"] = (C, C, 1)
        shapes[p + "context_conv_gate.bias"] = (C,)
        shapes[p + "conv_residual.weight"] = (C, C, 1)
        shapes[p + "conv_residual.bias"] = (C,)
        shapes[p + "conv_skip.weight"] = (K, C, 1)
        shapes[p + "conv_skip.bias"] = (K,)
    shapes["dense_conv.conv1.weight"] = (Q, K, 1)
    shapes["dense_conv.conv1.bias"] = (Q,)
    shapes["dense_conv.conv2.weight"] = (Q, Q, 1)
    shapes["dense_conv.conv2.bias"] = (Q,)
    return shapes


def _fan_in(key: str, shape: Tuple[int, ...]) -> int:
    if key.endswith(".bias"):
        return 0
    if key.startswith("video_transpose"):
        # ConvTranspose1d weight is (in, out, k): fan-in as torch counts it
        return shape[1] * shape[2]
    n = 1
    for s in shape[1:]:
        n *= s
    return n


def make_state_dict(
    layer_size: int,
    stack_size: int,
    input_channels: int,
    residual_channels: int = 16,
    skip_channels: int = 16,
    context_in_channels: int = 1,
    seed: int = 0,
    gain: float = 1.0,
    head_gain: float = 1.0,
) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 state_dict.  U(-b, b) with b = gain / sqrt(fan_in); the last
    head convolution is additionally scaled by ``head_gain`` ("sharpened"
    weights: wide logit range so greedy choices have large top-2 margins)."""
    rng = np.random.default_rng(seed)
    shapes = parameter_shapes(
        layer_size, stack_size, input_channels, residual_channels,
        skip_channels, context_in_channels,
    )
    fan_of_weight: Dict[str, int] = {}
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in shapes.items():
        if key.endswith(".weight"):
            fan = _fan_in(key, shape)
            fan_of_weight[key[: -len(".weight")]] = fan
        else:
            fan = fan_of_weight[key[: -len(".bias")]]
        bound = gain / np.sqrt(float(fan))
        if key.startswith("dense_conv.conv2"):
            bound *= head_gain
        arr = rng.uniform(-1.0, 1.0, size=shape) * bound
        sd[key] = torch.from_numpy(arr.astype(np.float32))
    return sd


def state_dict_sha256(sd: Dict[str, torch.Tensor]) -> str:
    h = hashlib.sha256()
    for key in sd:
        h.update(key.encode())
        h.update(np.ascontiguousarray(sd[key].detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def synthetic_indices(batch: int, length: int, classes: int, seed: int) -> torch.Tensor:
    """(B, T) int64 class indices, U{0..Q-1}; numpy-seeded so that the same
    indices come out on every box (seed 1234 + rank)."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.integers(0, classes, size=(batch, length), dtype=np.int64))


def one_hot(indices: torch.Tensor, classes: int, dtype=torch.float32) -> torch.Tensor:
    """(B, T) int64 -> (B, Q, T) one-hot, the construction of
    /root/reference/movenet/dataset.py:285-288."""
    B, T = indices.shape
    out = torch.zeros(B, classes, T, dtype=dtype, device=indices.device)
    return out.scatter_(1, indices.unsqueeze(1), 1)


def dilation_list(layer_size: int, stack_size: int) -> List[int]:
    return [2 ** x for _ in range(stack_size) for x in range(layer_size)]
