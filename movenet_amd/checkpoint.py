"""Checkpoint I/O compatible with the reference's files:

* the legacy trainer writes ``torch.save(model.state_dict())`` to ``.../model.pth``
  (movenet/trainer.py:455-467), with a ``module.`` prefix when trained under DDP
  (:256-260);
* the Lightning trainer's checkpoints hold ``{"state_dict": {"model.<key>": ...}}``
  (movenet/pytorch_lightning_trainer.py:31 names the attribute ``model``).

Files are read with ``torch.load(..., weights_only=True)`` only: nothing in them is
executed.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch

_PREFIXES = ("model.", "module.")


def strip_prefixes(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        changed = True
        while changed:
            changed = False
            for p in _PREFIXES:
                if k.startswith(p):
                    k, changed = k[len(p):], True
        out[k] = v
    return out


def load_state_dict_file(path) -> "OrderedDict[str, torch.Tensor]":
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict) or not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError(f"{path}: neither a state_dict nor a Lightning checkpoint")
    return strip_prefixes(obj)


def load_into(model: torch.nn.Module, path, strict: bool = True):
    """model.load_state_dict from any of the reference's checkpoint flavours."""
    return model.load_state_dict(load_state_dict_file(path), strict=strict)
