"""ctypes loader of oracle/_build/libring_oracle.so (oracle/ring_oracle.c): the cached
ring-buffer generator restated in plain C.  TEST INFRASTRUCTURE ONLY -- imported by tests/
and by bench.py's CPU baselines, never by movenet_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libring_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "ring_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-s"] + (["-B"] if force else []), check=True)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        h = C.CDLL(LIB_PATH)
        h.ro_weight_floats.restype = C.c_long
        h.ro_weight_floats.argtypes = [C.c_int] * 5
        h.ro_create.restype = C.c_void_p
        h.ro_create.argtypes = [C.c_int] * 6 + [C.c_void_p]
        h.ro_destroy.argtypes = [C.c_void_p]
        h.ro_set_half_operands.argtypes = [C.c_void_p, C.c_int]
        h.ro_generate.restype = C.c_int
        h.ro_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int]
        _lib = h
    return _lib


def pack_weights(sd: Dict[str, torch.Tensor], dims, operand_dtype=None) -> np.ndarray:
    """state_dict -> the blob layout documented in ring_oracle.c (every matrix [in][out]).
    ``operand_dtype=np.float16``: weight MATRICES rounded to fp16 values (biases and the causal
    conv's gathered columns stay fp32), for the fp16-operand form."""
    g32 = lambda k: sd[k].detach().cpu().numpy().astype(np.float32)

    def g(k):
        v = g32(k)
        if operand_dtype is None or k.endswith("bias") or k.startswith("causal_conv"):
            return v
        return v.astype(operand_dtype).astype(np.float32)
    parts = []
    cw = g("causal_conv.conv.weight")              # (C, Q, 2)
    parts += [cw[:, :, 0].T, cw[:, :, 1].T]        # E0[Q][C], E1[Q][C]
    for l in range(dims.n_layers):
        key = lambda n: f"residual_conv_stack.conv_layers.{l}.{n}"
        wf, wg = g(key("conv_filter.conv.weight")), g(key("conv_gate.conv.weight"))  # (C, C, 2)
        parts += [wf[:, :, 0].T, wf[:, :, 1].T, wg[:, :, 0].T, wg[:, :, 1].T]
        parts += [g(key("conv_residual.weight"))[:, :, 0].T, g(key("conv_residual.bias"))]
        parts += [g(key("conv_skip.weight"))[:, :, 0].T, g(key("conv_skip.bias"))]
    parts += [g("dense_conv.conv1.weight")[:, :, 0].T, g("dense_conv.conv1.bias"),
              g("dense_conv.conv2.weight")[:, :, 0].T, g("dense_conv.conv2.bias")]
    blob = np.ascontiguousarray(np.concatenate([np.ascontiguousarray(p).reshape(-1) for p in parts]))
    want = lib().ro_weight_floats(dims.layer_size, dims.stack_size, dims.input_channels,
                                  dims.residual_channels, dims.skip_channels)
    assert blob.size == want, (blob.size, want)
    return blob


class RingC:
    def __init__(self, sd, dims, batch: int, operand_dtype=None):
        self.dims, self.batch = dims, batch
        self.blob = pack_weights(sd, dims, operand_dtype)
        self.h = lib().ro_create(dims.layer_size, dims.stack_size, dims.input_channels,
                                 dims.residual_channels, dims.skip_channels, batch,
                                 self.blob.ctypes.data)
        if not self.h:
            raise MemoryError("ro_create failed")
        if operand_dtype is not None:
            assert operand_dtype == np.float16, "only fp16 operands are restated in C"
            lib().ro_set_half_operands(self.h, 1)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ro_destroy(self.h)
            self.h = None

    def run(self, samples: np.ndarray, n_given: int, t_begin: int, t_end: int, threads: int = 1,
            choices: Optional[np.ndarray] = None, logits: Optional[np.ndarray] = None,
            logits_t0: int = 0) -> None:
        assert samples.dtype == np.int32 and samples.flags.c_contiguous and samples.shape[0] == self.batch
        rc = lib().ro_generate(self.h, samples.ctypes.data, samples.shape[1], n_given, t_begin, t_end,
                               None if choices is None else choices.ctypes.data,
                               None if logits is None else logits.ctypes.data, logits_t0, threads)
        if rc != 0:
            raise RuntimeError(f"ro_generate returned {rc}")


def generate_ring_c(sd, dims, prompt_idx: np.ndarray, n_samples: int, forced_idx=None, threads: int = 1,
                    operand_dtype=None):
    """Same contract as wavenet_oracle.generate_ring: (choices (B, n) int64, logits (B, n-RF, Q))."""
    rf, B = dims.receptive_fields, prompt_idx.shape[0]
    samples = np.zeros((B, n_samples), np.int32)
    n_given = rf
    if forced_idx is not None:
        samples[:] = forced_idx
        n_given = n_samples
    samples[:, :rf] = prompt_idx[:, :rf]
    choices = np.zeros((B, n_samples), np.int32)
    logits = np.zeros((B, max(n_samples - rf, 0), dims.input_channels), np.float32)
    RingC(sd, dims, B, operand_dtype).run(samples, n_given, 0, n_samples - 1, threads, choices, logits, rf)
    choices[:, :rf] = prompt_idx[:, :rf]
    return choices.astype(np.int64), logits
