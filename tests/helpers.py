"""Shared helpers for the parity tests (CPU and GPU)."""
from __future__ import annotations

import numpy as np
import torch

from movenet_amd.utils.weights import (
    make_state_dict, one_hot, state_dict_sha256, synthetic_indices,
)
from oracle import wavenet_oracle as O

CFG_KEYS = ("layer_size", "stack_size", "input_channels", "residual_channels", "skip_channels")


def cfg_of(fx) -> dict:
    return dict(zip(CFG_KEYS, (int(v) for v in fx["cfg"])))


def weights_of(fx):
    """Regenerate the fixture's weights from its recipe and check the SHA."""
    kw = {}
    if "gain" in fx.files:
        kw = dict(gain=float(fx["gain"]), head_gain=float(fx["head_gain"]))
    cfg = cfg_of(fx)
    sd = make_state_dict(**cfg, seed=int(fx["weight_seed"]), **kw)
    assert state_dict_sha256(sd) == str(fx["weight_sha"]), "weight recipe drifted from the fixture"
    return cfg, O.Dims(**cfg), sd


def rel_err(a, b) -> float:
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
