"""Full-sequence forward/backward through the C ABI, wrapped as one
``torch.autograd.Function`` so that ``WaveNet.forward`` is trainable with the
stock torch optimizers (/root/reference/movenet/wavenet.py:158-191 and the
autograd graph PyTorch would build for it).

torch is used for device memory (activation buffers come from the caching
allocator) and for the autograd hook; every arithmetic kernel is HIP.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import _native as N
from .generation import _require_gpu, _stream_ptr, pack_params

_LAYER_PARAMS = ("conv_filter.conv.weight", "conv_gate.conv.weight", "conv_residual.weight",
                 "conv_residual.bias", "conv_skip.weight", "conv_skip.bias")


def decoder_param_names(n_layers: int) -> List[str]:
    """Parameters the audio path uses, in the order they are passed to autograd."""
    names = ["causal_conv.conv.weight"]
    for l in range(n_layers):
        names += [f"residual_conv_stack.conv_layers.{l}.{p}" for p in _LAYER_PARAMS]
    names += ["dense_conv.conv1.weight", "dense_conv.conv1.bias",
              "dense_conv.conv2.weight", "dense_conv.conv2.bias"]
    return names


class ForwardBuffers:
    """Device buffers of one forward pass (sizes: include/movenet_hip.h)."""

    def __init__(self, dims: N.Dims, batch: int, t_len: int, save: bool, device):
        lib = N.lib()
        L = dims.layer_size * dims.stack_size
        C, K, Q = dims.residual_channels, dims.skip_channels, dims.input_channels
        S = N.check(lib.mvn_output_size(dims, t_len), "mvn_output_size")
        self.S, self.Tp, self.Sp = S, lib.mvn_padded_len(t_len), lib.mvn_padded_len(S)
        f32 = dict(dtype=torch.float32, device=device)
        self.acts = torch.empty(((L + 1) if save else 2, batch, C, self.Tp), **f32)
        self.th = torch.empty((L, batch, C, self.Tp), **f32) if save else None
        self.sg = torch.empty((L, batch, C, self.Tp), **f32) if save else None
        self.z = torch.empty((batch, C, self.Tp), **f32)
        self.skip = torch.empty((batch, K, self.Sp), **f32)
        self.a1 = torch.empty((batch, Q, self.Sp), **f32)
        self.struct = N.FwdBuffers(
            self.acts.data_ptr(), self.th.data_ptr() if save else None,
            self.sg.data_ptr() if save else None, self.z.data_ptr(), self.skip.data_ptr(),
            self.a1.data_ptr())


def run_forward(dims: N.Dims, sd: Dict[str, torch.Tensor], idx: torch.Tensor, normalize: bool,
                remove_last: bool, save: bool) -> Tuple[torch.Tensor, ForwardBuffers]:
    lib = N.lib()
    _require_gpu(idx, "audio indices")
    B, T = idx.shape
    L = dims.layer_size * dims.stack_size
    dev = idx.device
    with torch.cuda.device(dev):
        buf = ForwardBuffers(dims, B, T, save, dev)
        s_out = buf.S - (1 if remove_last else 0)
        out = torch.empty((B, dims.input_channels, max(s_out, 0)), dtype=torch.float32, device=dev)
        params, keep = pack_params(dims, sd, L)
        N.check(lib.mvn_forward(dims, params, idx.data_ptr(), idx.stride(0), B, T, buf.struct,
                                out.data_ptr(), int(normalize), int(remove_last), int(save),
                                _stream_ptr(dev)), "mvn_forward")
    buf._keep = keep  # parameter tensors stay alive until the kernels have run
    return out, buf


class _WaveNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dims, names, idx, normalize, remove_last, *params):
        sd = dict(zip(names, params))
        save = any(ctx.needs_input_grad[5:])  # grad mode itself is off inside forward()
        out, buf = run_forward(dims, sd, idx, normalize, remove_last, save)
        ctx.dims, ctx.names, ctx.idx, ctx.buf = dims, names, idx, buf
        ctx.normalize, ctx.remove_last, ctx.saved_fwd = normalize, remove_last, save
        ctx.save_for_backward(out, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.saved_fwd:
            raise RuntimeError("movenet_amd: forward ran without saved activations")
        lib = N.lib()
        out, *params = ctx.saved_tensors
        dims, names, idx, buf = ctx.dims, ctx.names, ctx.idx, ctx.buf
        L = dims.layer_size * dims.stack_size
        C, K, Q = dims.residual_channels, dims.skip_channels, dims.input_channels
        B, T = idx.shape
        dev = idx.device
        sd = dict(zip(names, params))
        dout = dout.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            grads = {n: torch.zeros_like(p, dtype=torch.float32) for n, p in sd.items()}
            gp, gkeep = pack_params(dims, grads, L)
            g = N.ParamGrads(gp.causal_w, gp.filter_w, gp.gate_w, gp.residual_w, gp.residual_b,
                             gp.skip_w, gp.skip_b, gp.head1_w, gp.head1_b, gp.head2_w, gp.head2_b)
            f32 = dict(dtype=torch.float32, device=dev)
            dx_a = torch.empty((B, C, buf.Tp), **f32)
            dx_b = torch.empty((B, C, buf.Tp), **f32)
            dfg = torch.empty((B, 2 * C, buf.Tp), **f32)
            dskip = torch.empty((B, K, buf.Sp), **f32)
            da1 = torch.empty((B, Q, buf.Sp), **f32)
            dlogit = torch.empty((B, Q, buf.Sp), **f32)
            bw = N.BwdBuffers(dx_a.data_ptr(), dx_b.data_ptr(), dfg.data_ptr(), dskip.data_ptr(),
                              da1.data_ptr(), dlogit.data_ptr())
            params_c, pkeep = pack_params(dims, sd, L)
            N.check(lib.mvn_backward(dims, params_c, g, idx.data_ptr(), idx.stride(0), B, T,
                                     buf.struct, bw, out.data_ptr(), dout.data_ptr(),
                                     int(ctx.normalize), int(ctx.remove_last), _stream_ptr(dev)),
                    "mvn_backward")
        ctx.buf = None
        last = f"residual_conv_stack.conv_layers.{L - 1}.conv_residual."
        result = []
        for n, need in zip(names, ctx.needs_input_grad[5:]):
            # the last layer's residual conv never reaches the output: the
            # reference leaves its .grad None, so do we
            result.append(None if (n.startswith(last) or not need) else grads[n])
        return (None, None, None, None, None, *result)


def wavenet_forward(model, audio: torch.Tensor, output_unnormalized: bool = True,
                    remove_last: bool = True) -> torch.Tensor:
    """WaveNet.forward for the audio-only path.  NOTE the reference's inverted
    flag (wavenet.py:189-191): output_unnormalized=True returns PROBABILITIES."""
    idx = model._indices_of(audio)
    model.compute_output_size(audio)  # ValueError when T < RF, like the reference
    L = model.layer_size * model.stack_size
    names = decoder_param_names(L)
    lookup = dict(model.named_parameters())
    params = [lookup[n] for n in names]
    out = _WaveNetFunction.apply(model._dims, names, idx, bool(output_unnormalized),
                                 bool(remove_last), *params)
    return out if audio.dtype == torch.float32 else out.to(audio.dtype)
