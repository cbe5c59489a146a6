"""Full-sequence forward/backward through the C ABI, wrapped as one
``torch.autograd.Function`` so that ``WaveNet.forward`` is trainable with the
stock torch optimizers (/root/reference/movenet/wavenet.py:158-191 and the
autograd graph PyTorch would build for it).

torch is used for device memory (activation buffers come from the caching
allocator) and for the autograd hook; every arithmetic kernel is HIP.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import ctypes
import os

import torch

from . import _native as N
from .generation import _require_gpu, _stream_ptr, pack_params

C_void3 = ctypes.c_void_p * 3

_LAYER_PARAMS = ("conv_filter.conv.weight", "conv_gate.conv.weight", "conv_residual.weight",
                 "conv_residual.bias", "conv_skip.weight", "conv_skip.bias")
_CTX_PARAMS = ("context_conv_filter.weight", "context_conv_filter.bias",
               "context_conv_gate.weight", "context_conv_gate.bias")
VIDEO_PARAMS = ("video_conv.weight", "video_conv.bias",
                "video_transpose.0.weight", "video_transpose.0.bias",
                "video_transpose.1.weight", "video_transpose.1.bias",
                "video_transpose.2.weight", "video_transpose.2.bias")


def decoder_param_names(n_layers: int, with_context: bool = False) -> List[str]:
    """Parameters the decoder uses, in the order they are passed to autograd."""
    per_layer = _LAYER_PARAMS + (_CTX_PARAMS if with_context else ())
    names = ["causal_conv.conv.weight"]
    for l in range(n_layers):
        names += [f"residual_conv_stack.conv_layers.{l}.{p}" for p in per_layer]
    names += ["dense_conv.conv1.weight", "dense_conv.conv1.bias",
              "dense_conv.conv2.weight", "dense_conv.conv2.bias"]
    return names


class VideoGradSlot:
    """Hand-over between the two autograd nodes of a conditioned step.  The decoder's backward
    (which runs first: it produces the context gradient) allocates ONE zero-filled flat buffer
    for its own gradients AND the eight video-encoder gradients behind them, in the order
    ``optim.order_like_backward(model, with_context=True)`` lays the parameters out; the
    up-sampler's backward then writes into those views instead of eight buffers of its own.
    One storage => one all-reduce message (parallel.contiguous_grad_span) and one AdamW launch."""

    def __init__(self, shapes):
        self.shapes = [tuple(s) for s in shapes]
        self.sizes = [int(torch.Size(s).numel()) for s in self.shapes]
        self.total = sum(self.sizes)
        self.views = None

    def carve(self, tail: torch.Tensor) -> None:
        views, off = [], 0
        for shape, k in zip(self.shapes, self.sizes):
            views.append(tail[off:off + k].view(shape))
            off += k
        self.views = views

    def take(self):
        views, self.views = self.views, None
        return views


class ForwardBuffers:
    """Device buffers of one forward pass (sizes: include/movenet_hip.h)."""

    def __init__(self, dims: N.Dims, batch: int, t_len: int, save: bool, device, ctx=None,
                 dense=None):
        lib = N.lib()
        L = dims.layer_size * dims.stack_size
        C, K, Q = dims.residual_channels, dims.skip_channels, dims.input_channels
        S = N.check(lib.mvn_output_size(dims, t_len), "mvn_output_size")
        self.S, self.Tp, self.Sp = S, lib.mvn_padded_len(t_len), lib.mvn_padded_len(S + 31)
        f32 = dict(dtype=torch.float32, device=device)
        # (MOVENET_DEBUG_GUARD=1: a band of sentinels behind every buffer, checked after mvn_forward)
        self.guard = _GuardBands(device) if os.environ.get("MOVENET_DEBUG_GUARD") == "1" else None
        if self.guard is not None:
            lib.mvn_reload_switches()  # (debug / test mode: the MOVENET_HIP_* A/B switches may change inside the process)
        alloc = torch.empty if self.guard is None else self.guard.empty
        self.acts = alloc(((L + 1) if save else 2, batch, C, self.Tp), **f32)
        self.th = alloc((L, batch, C, self.Tp), **f32) if save else None
        self.sg = alloc((L, batch, C, self.Tp), **f32) if save else None
        self.z = alloc((batch, C, self.Tp), **f32)
        self.skip = alloc((batch, K, self.Sp), **f32)
        self.a1 = alloc((batch, Q, self.Sp), **f32)
        self.ctx = ctx
        self.struct = N.FwdBuffers(
            self.acts.data_ptr(), self.th.data_ptr() if save else None,
            self.sg.data_ptr() if save else None, self.z.data_ptr(), self.skip.data_ptr(),
            self.a1.data_ptr(), None if ctx is None else ctx.data_ptr(),
            0 if ctx is None else ctx.stride(1),
            None if dense is None else dense.data_ptr(), 0 if dense is None else dense.stride(1))
        self.dense = dense


def run_forward(dims: N.Dims, sd: Dict[str, torch.Tensor], idx: torch.Tensor, normalize: bool,
                remove_last: bool, save: bool, ctx=None, f16: bool = False) -> Tuple[torch.Tensor, ForwardBuffers]:
    """idx: (B,T) int32 class indices, or a (B,Q,T) fp32 tensor for inputs that are
    not one-hot (dense causal conv).  ``f16``: fp16 operands / fp32 accumulation in every
    product (mvn_forward_f16; inference and generator priming)."""
    lib = N.lib()
    _require_gpu(idx, "audio")
    dense = None
    if idx.dim() == 3:
        dense = idx.detach().to(torch.float32).contiguous()
        B, _, T = dense.shape
    else:
        B, T = idx.shape
    L = dims.layer_size * dims.stack_size
    dev = idx.device
    if ctx is not None:
        # same check as the reference's assert (wavenet.py:170-174)
        assert tuple(ctx.shape) == (B, dims.residual_channels, T), (
            "expected video and audio tensors to have equal sizes, found "
            f"{tuple(ctx.shape)}, {(B, dims.residual_channels, T)}")
        ctx = ctx.detach().to(torch.float32).contiguous()
    with torch.cuda.device(dev):
        buf = ForwardBuffers(dims, B, T, save, dev, ctx, dense)
        s_out = buf.S - (1 if remove_last else 0)
        out = (torch.empty if buf.guard is None else buf.guard.empty)(
            (B, dims.input_channels, max(s_out, 0)), dtype=torch.float32, device=dev)
        params, keep = pack_params(dims, sd, L)
        fwd = lib.mvn_forward_f16 if f16 else lib.mvn_forward
        N.check(fwd(dims, params, None if dense is not None else idx.data_ptr(),
                    0 if dense is not None else idx.stride(0), B, T, buf.struct,
                    out.data_ptr(), int(normalize), int(remove_last), int(save),
                    _stream_ptr(dev)), "mvn_forward_f16" if f16 else "mvn_forward")
        if buf.guard is not None:
            buf.guard.check("mvn_forward")
    buf._keep = keep  # parameter tensors stay alive until the kernels have run
    return out, buf


class _WaveNetFunction(torch.autograd.Function):
    """args: dims, names, idx, normalize, remove_last, ctx (Tensor or None), *params"""
    N_FIXED = 6

    @staticmethod
    def forward(ctx_, dims, names, idx, normalize, remove_last, context, *params):
        sd = dict(zip(names, params))
        # (grad mode itself is off inside forward(): the caller records whether it was on)
        save = any(ctx_.needs_input_grad[5:]) and bool(getattr(dims, "_grad_mode", True))
        f16 = bool(getattr(dims, "_f16", False))
        if f16 and save:
            raise RuntimeError("movenet_amd: forward_precision 'fp16' is inference-only "
                               "(mvn_backward differentiates the fp32 forward); use torch.no_grad()")
        out, buf = run_forward(dims, sd, idx, normalize, remove_last, save, context, f16=f16)
        ctx_.dims, ctx_.names, ctx_.idx, ctx_.buf = dims, names, idx, buf
        ctx_.normalize, ctx_.remove_last, ctx_.saved_fwd = normalize, remove_last, save
        ctx_.has_context = context is not None
        ctx_.video_slot = getattr(dims, "_video_slot", None) if context is not None else None
        ctx_.save_for_backward(out, *params)
        return out

    @staticmethod
    def backward(ctx_, dout):
        if not ctx_.saved_fwd:
            raise RuntimeError("movenet_amd: forward ran without saved activations")
        out, *params = ctx_.saved_tensors
        dout = dout.to(torch.float32).contiguous()
        grads, dctx = _run_backward(ctx_, params, out, dout, None)
        return _grads_for_autograd(ctx_, grads, dctx, _WaveNetFunction.N_FIXED, ctx_.needs_input_grad)


class _GuardBands:
    """Debug allocator (MOVENET_DEBUG_GUARD=1): tensors with a band of sentinel values behind them; ``check``
    raises if a kernel wrote past the end of one (r3: the fused backward halves' bias partial sums did, for short
    sequences in small batches -- into whichever tensor the allocator had placed next)."""
    BAND, SENTINEL = 1 << 18, -1234.5   # 1 MiB of floats behind each tensor

    def __init__(self, device):
        self.device, self.bands = device, []

    def empty(self, shape, dtype=torch.float32, device=None):
        n = 1
        for k in shape:
            n *= int(k)
        raw = torch.empty(n + self.BAND, dtype=dtype, device=self.device)
        raw[n:].fill_(self.SENTINEL)
        self.bands.append((tuple(shape), raw[n:]))
        return raw[:n].view(shape)

    def check(self, what: str) -> None:
        for shape, band in self.bands:
            bad = int((band != self.SENTINEL).sum().item())
            if bad:
                raise RuntimeError(f"movenet_amd: {what} wrote {bad} floats past the end of a {shape} scratch tensor")


def _run_backward(ctx_, params, out, dout, fill_dlogit):
    """mvn_backward for a saved forward.  ``dout``: gradient w.r.t. the model output, or None
    when ``fill_dlogit(dlogit, Sp, pad)`` writes the gradient w.r.t. the logits itself (the
    fused loss).  Returns ({name: grad view into ONE flat buffer}, dctx or None)."""
    lib = N.lib()
    dims, names, idx, buf = ctx_.dims, ctx_.names, ctx_.idx, ctx_.buf
    L = dims.layer_size * dims.stack_size
    C, K, Q = dims.residual_channels, dims.skip_channels, dims.input_channels
    dense_in = idx.dim() == 3
    B, T = (idx.shape[0], idx.shape[2]) if dense_in else idx.shape
    dev = idx.device
    sd = dict(zip(names, params))
    with torch.cuda.device(dev):
        # one zero-filled flat buffer, the per-parameter gradients are views into it (one
        # fill kernel instead of ~190; FlatGradSync all-reduces the buffer in place and
        # FlatAdamW steps over it with one kernel)
        sizes = [p.numel() for p in params]
        slot = getattr(ctx_, "video_slot", None)
        flat = torch.zeros(sum(sizes) + (slot.total if slot is not None else 0), dtype=torch.float32,
                           device=dev)
        if slot is not None:  # the video encoder's gradients live behind the decoder's
            slot.carve(flat[sum(sizes):])
        grads, off = {}, 0
        for n, p_, k in zip(names, params, sizes):
            grads[n] = flat[off:off + k].view(p_.shape)
            off += k
        gp, gkeep = pack_params(dims, grads, L)
        g = N.ParamGrads(gp.causal_w, gp.filter_w, gp.gate_w, gp.residual_w, gp.residual_b,
                         gp.skip_w, gp.skip_b, gp.head1_w, gp.head1_b, gp.head2_w, gp.head2_b,
                         gp.ctx_filter_w, gp.ctx_filter_b, gp.ctx_gate_w, gp.ctx_gate_b)
        f32 = dict(dtype=torch.float32, device=dev)
        # MOVENET_DEBUG_GUARD=1 (tests): every scratch tensor of the backward pass gets a guard band behind it,
        # checked after the call -- the library carves its slab / partial-sum scratch out of these tensors
        guard = _GuardBands(dev) if os.environ.get("MOVENET_DEBUG_GUARD") == "1" else None
        if guard is not None:
            lib.mvn_reload_switches()
        alloc = torch.empty if guard is None else guard.empty
        dx_a = alloc((B, C, buf.Tp), **f32)
        dx_b = alloc((B, C, buf.Tp), **f32)
        dfg = alloc((B, 2 * C, buf.Tp), **f32)
        dskip = alloc((B, K, buf.Sp), **f32)
        da1 = alloc((B, Q, buf.Sp), **f32)
        dlogit = alloc((B, Q, buf.Sp), **f32)
        dctx = alloc((B, C, buf.Tp), **f32) if ctx_.has_context else None
        bw = N.BwdBuffers(dx_a.data_ptr(), dx_b.data_ptr(), dfg.data_ptr(), dskip.data_ptr(),
                          da1.data_ptr(), dlogit.data_ptr(),
                          None if dctx is None else dctx.data_ptr())
        if fill_dlogit is not None:
            rf = N.check(lib.mvn_receptive_fields(dims), "mvn_receptive_fields")
            fill_dlogit(dlogit, buf.Sp, (rf - 1) & 31)
        params_c, pkeep = pack_params(dims, sd, L)
        N.check(lib.mvn_backward(dims, params_c, g, None if dense_in else idx.data_ptr(),
                                 0 if dense_in else idx.stride(0), B, T,
                                 buf.struct, bw, None if out is None else out.data_ptr(),
                                 None if dout is None else dout.data_ptr(),
                                 int(ctx_.normalize), int(ctx_.remove_last), _stream_ptr(dev)),
                "mvn_backward")
        if guard is not None:
            guard.check("mvn_backward")
    # (buffers are released with the autograd node; a retained graph may run backward again)
    return grads, (dctx[:, :, :T] if dctx is not None else None)


def _grads_for_autograd(ctx_, grads, dctx, n_fixed, need):
    dims, names = ctx_.dims, ctx_.names
    L = dims.layer_size * dims.stack_size
    last = f"residual_conv_stack.conv_layers.{L - 1}.conv_residual."
    result = []
    for n, want in zip(names, need[n_fixed:]):
        # the last layer's residual conv never reaches the output: the
        # reference leaves its .grad None, so do we
        result.append(None if (n.startswith(last) or not want) else grads[n])
    dcontext = dctx if (ctx_.has_context and need[n_fixed - 1]) else None
    return (*([None] * (n_fixed - 1)), dcontext, *result)


class _WaveNetLossFunction(torch.autograd.Function):
    """Forward + the trainer's loss in one autograd node (row F3): the head's logits become
    probabilities and the loss / accuracy partial sums in ONE pass (mvn_softmax_ce_forward);
    backward differentiates loss -> logits in ONE pass (mvn_softmax_ce_backward) straight into
    mvn_backward's dlogit buffer.  args: dims, names, idx, target (B,S) int64, ctx, *params;
    returns (loss, accuracy, probs)."""
    N_FIXED = 5

    @staticmethod
    def forward(ctx_, dims, names, idx, target, context, *params):
        lib = N.lib()
        sd = dict(zip(names, params))
        save = any(ctx_.needs_input_grad[4:]) and bool(getattr(dims, "_grad_mode", True))
        out, buf = run_forward(dims, sd, idx, False, True, save, context)  # logits, last column dropped
        B, Q, S = out.shape
        if target.shape != (B, S):
            raise ValueError(f"target must be (batch, {S}), got {tuple(target.shape)}")
        tg = target.detach().to(device=out.device, dtype=torch.int64).contiguous()
        with torch.cuda.device(out.device):
            parts = max(lib.mvn_ce_parts(B, S), 1)
            loss_part = torch.zeros(parts, dtype=torch.float32, device=out.device)
            ok_part = torch.zeros(parts, dtype=torch.int32, device=out.device)
            N.check(lib.mvn_softmax_ce_forward(out.data_ptr(), tg.data_ptr(), B, Q, S, loss_part.data_ptr(),
                                               ok_part.data_ptr(), _stream_ptr(out.device)),
                    "mvn_softmax_ce_forward")
        n = max(B * S, 1)
        loss = loss_part.sum() / n
        acc = ok_part.sum().to(torch.float32) / n
        ctx_.dims, ctx_.names, ctx_.idx, ctx_.buf = dims, names, idx, buf
        ctx_.normalize, ctx_.remove_last, ctx_.saved_fwd = True, True, save
        ctx_.has_context = context is not None
        ctx_.video_slot = getattr(dims, "_video_slot", None) if context is not None else None
        ctx_.save_for_backward(out, tg, *params)
        ctx_.mark_non_differentiable(acc, out)
        return loss, acc, out

    @staticmethod
    def backward(ctx_, dloss, _dacc, _dprobs):
        if not ctx_.saved_fwd:
            raise RuntimeError("movenet_amd: forward ran without saved activations")
        lib = N.lib()
        probs, tg, *params = ctx_.saved_tensors
        B, Q, S = probs.shape
        up = dloss.detach().to(device=probs.device, dtype=torch.float32).reshape(1).contiguous()

        def fill(dlogit, Sp, pad):
            N.check(lib.mvn_softmax_ce_backward(
                probs.data_ptr(), tg.data_ptr(), B, Q, S, 1.0 / max(B * S, 1), up.data_ptr(),
                dlogit.data_ptr(), Q * Sp, Sp, pad, S + 1, _stream_ptr(probs.device)),
                "mvn_softmax_ce_backward")

        grads, dctx = _run_backward(ctx_, params, None, None, fill)
        return _grads_for_autograd(ctx_, grads, dctx, _WaveNetLossFunction.N_FIXED, ctx_.needs_input_grad)


class _UpsampleVideoFunction(torch.autograd.Function):
    """video (B,F,64,64,Cin) -> context (B,C,1000F) through mvn_upsample_video."""

    @staticmethod
    def forward(ctx_, dims, slot, video, *params):
        lib = N.lib()
        _require_gpu(video, "video")
        if video.dim() != 5 or video.shape[2] != 64 or video.shape[3] != 64:
            raise ValueError(f"video must be (batch, frames, 64, 64, channels), got {tuple(video.shape)}")
        video = video.detach().to(torch.float32).contiguous()
        B, F, _, _, cin = video.shape
        C = dims.residual_channels
        dev = video.device
        pl = lib.mvn_padded_len
        with torch.cuda.device(dev):
            f32 = dict(dtype=torch.float32, device=dev)
            enc = torch.empty((B, C, pl(F)), **f32)
            u1 = torch.empty((B, C, pl(10 * F)), **f32)
            u2 = torch.empty((B, C, pl(100 * F)), **f32)
            out = torch.empty((B, C, 1000 * F), **f32)
            ps = [p.detach().to(torch.float32).contiguous() for p in params]
            vp = N.VideoParams(ps[0].data_ptr(), ps[1].data_ptr(),
                               (C_void3)(ps[2].data_ptr(), ps[4].data_ptr(), ps[6].data_ptr()),
                               (C_void3)(ps[3].data_ptr(), ps[5].data_ptr(), ps[7].data_ptr()))
            N.check(lib.mvn_upsample_video(dims, vp, video.data_ptr(), B, F, cin, enc.data_ptr(),
                                           u1.data_ptr(), u2.data_ptr(), out.data_ptr(),
                                           out.stride(1), _stream_ptr(dev)), "mvn_upsample_video")
        ctx_.dims, ctx_.shape, ctx_.slot = dims, (B, F, cin), slot
        ctx_.save_for_backward(video, enc, u1, u2, *ps)
        return out

    @staticmethod
    def backward(ctx_, dout):
        lib = N.lib()
        video, enc, u1, u2, *ps = ctx_.saved_tensors
        dims = ctx_.dims
        B, F, cin = ctx_.shape
        dev = video.device
        dout = dout.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            # zero-filled views behind the decoder's gradients when its backward has run (the
            # usual case: it produced dout), else eight buffers of this node's own
            grads = ctx_.slot.take() if ctx_.slot is not None else None
            if grads is None or grads[0].device != dev:
                grads = [torch.zeros_like(p) for p in ps]
            vp = N.VideoParams(ps[0].data_ptr(), ps[1].data_ptr(),
                               (C_void3)(ps[2].data_ptr(), ps[4].data_ptr(), ps[6].data_ptr()),
                               (C_void3)(ps[3].data_ptr(), ps[5].data_ptr(), ps[7].data_ptr()))
            vg = N.VideoParams(grads[0].data_ptr(), grads[1].data_ptr(),
                               (C_void3)(grads[2].data_ptr(), grads[4].data_ptr(), grads[6].data_ptr()),
                               (C_void3)(grads[3].data_ptr(), grads[5].data_ptr(), grads[7].data_ptr()))
            d_u2, d_u1, d_enc = torch.empty_like(u2), torch.empty_like(u1), torch.empty_like(enc)
            n_scratch = int(lib.mvn_upsample_video_scratch_floats(dims, B, F))  # per-workgroup weight-gradient slabs
            scratch = torch.empty(max(n_scratch, 1), dtype=torch.float32, device=dev)
            N.check(lib.mvn_upsample_video_backward(
                dims, vp, vg, video.data_ptr(), B, F, cin, enc.data_ptr(), u1.data_ptr(),
                u2.data_ptr(), dout.data_ptr(), dout.stride(1), d_u2.data_ptr(), d_u1.data_ptr(),
                d_enc.data_ptr(), scratch.data_ptr(), n_scratch, _stream_ptr(dev)), "mvn_upsample_video_backward")
        need = ctx_.needs_input_grad[3:]
        return (None, None, None, *[g if w else None for g, w in zip(grads, need)])


def upsample_video(model, video: torch.Tensor) -> torch.Tensor:
    lookup = dict(model.named_parameters())
    vparams = [lookup[n] for n in VIDEO_PARAMS]
    slot = VideoGradSlot([p.shape for p in vparams]) if all(p.requires_grad for p in vparams) else None
    out = _UpsampleVideoFunction.apply(model._dims, slot, video, *vparams)
    if slot is not None and out.requires_grad:
        out._mvn_video_slot = slot  # read back by wavenet_forward / wavenet_forward_loss
    return out


def _tagged_dims(dims, f16: bool = False, context=None):
    """A copy of the dims struct carrying what the autograd Functions cannot see from inside
    ``forward``: whether grad mode was on at the call, and the operand precision."""
    d = N.make_dims(dims.layer_size, dims.stack_size, dims.input_channels, dims.residual_channels,
                    dims.skip_channels)
    d._grad_mode = torch.is_grad_enabled()
    d._f16 = f16
    d._video_slot = getattr(context, "_mvn_video_slot", None) if context is not None else None
    return d


def _decoder_params(model, with_context: bool):
    L = model.layer_size * model.stack_size
    names = decoder_param_names(L, with_context=with_context)
    lookup = dict(model.named_parameters())
    return names, [lookup[n] for n in names]


def wavenet_forward(model, audio: torch.Tensor, context=None, output_unnormalized: bool = True,
                    remove_last: bool = True) -> torch.Tensor:
    """WaveNet.forward.  NOTE the reference's inverted flag (wavenet.py:189-191):
    output_unnormalized=True returns PROBABILITIES.  ``context``: upsampled video
    (B, C, T) or None.  One-hot input runs the causal conv as a gather; the check that the
    input IS one-hot is read after the kernels have been enqueued (no host wait on an idle
    GPU) and anything else is rerun through the dense causal conv."""
    model.compute_output_size(audio)  # ValueError when T < RF, like the reference
    idx, check = model._indices_async(audio)
    names, params = _decoder_params(model, context is not None)
    dims = _tagged_dims(model._dims, f16=model.forward_precision == "fp16", context=context)
    out = _WaveNetFunction.apply(dims, names, idx, bool(output_unnormalized),
                                 bool(remove_last), context, *params)
    if not model._all_one_hot(check):  # dense causal conv on the tensor itself
        # (a dense input pays the discarded index pass: one extra forward; its buffers --
        # saved activations included -- are released BEFORE the rerun allocates its own)
        del out
        dense = audio.detach().to(torch.float32).contiguous()
        out = _WaveNetFunction.apply(dims, names, dense, bool(output_unnormalized),
                                     bool(remove_last), context, *params)
    return out if audio.dtype == torch.float32 else out.to(audio.dtype)


def wavenet_forward_loss(model, audio: torch.Tensor, context=None, target=None):
    """(loss, accuracy, probabilities) of one trainer step in one autograd node:
    ``output = model(audio, video)`` (probabilities, Q1), ``target =
    audio[:, :, RF:].argmax(1)``, ``F.cross_entropy(output, target)`` (on probabilities, Q2) and
    the accuracy -- movenet/pytorch_lightning_trainer.py:62-66 -- with the softmax, the loss and
    the accuracy fused into one pass over the head's logits and their gradients into one pass
    back.  Same values as ``cross_entropy_on_probs(wavenet_forward(...), target)``."""
    model.compute_output_size(audio)
    rf = model.receptive_fields
    idx, check = model._indices_async(audio)
    names, params = _decoder_params(model, context is not None)
    tg = idx[:, rf:].to(torch.int64) if target is None else target
    dims = _tagged_dims(model._dims, context=context)
    res = _WaveNetLossFunction.apply(dims, names, idx, tg, context, *params)
    if not model._all_one_hot(check):  # (read after the enqueue: no idle GPU) dense causal conv
        del res  # release the discarded pass (saved activations) before the rerun allocates
        dense = audio.detach().to(torch.float32).contiguous()
        tg = audio[:, :, rf:].argmax(1) if target is None else target
        res = _WaveNetLossFunction.apply(dims, names, dense, tg, context, *params)
    return res


def mu_law_encode(x: torch.Tensor, quantization_channels: int) -> torch.Tensor:
    """float waveform in [-1, 1] (any shape, on the GPU) -> int32 class indices."""
    _require_gpu(x, "waveform")
    x = x.detach().to(torch.float32).contiguous()
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.lib().mvn_mu_law_encode(x.data_ptr(), out.data_ptr(), x.numel(),
                                          quantization_channels, _stream_ptr(x.device)),
                "mvn_mu_law_encode")
    return out


def mu_law_decode(index: torch.Tensor, quantization_channels: int) -> torch.Tensor:
    _require_gpu(index, "indices")
    index = index.detach().to(torch.int32).contiguous()
    out = torch.empty(index.shape, dtype=torch.float32, device=index.device)
    with torch.cuda.device(index.device):
        N.check(N.lib().mvn_mu_law_decode(index.data_ptr(), out.data_ptr(), index.numel(),
                                          quantization_channels, _stream_ptr(index.device)),
                "mvn_mu_law_decode")
    return out


class _CrossEntropyOnProbs(torch.autograd.Function):
    """loss, accuracy of the trainer (row F3): mvn_ce_on_probs_forward / _backward."""

    @staticmethod
    def forward(ctx_, probs, target):
        _require_gpu(probs, "probs")
        if probs.dim() != 3 or target.shape != (probs.shape[0], probs.shape[2]):
            raise ValueError(f"probs (B,Q,S) / target (B,S) expected, got {tuple(probs.shape)} / "
                             f"{tuple(target.shape)}")
        p = probs.detach().to(torch.float32).contiguous()
        tg = target.detach().to(device=p.device, dtype=torch.int64).contiguous()
        B, Q, S = p.shape
        lib = N.lib()
        with torch.cuda.device(p.device):
            parts = lib.mvn_ce_parts(B, S)
            loss_part = torch.zeros(max(parts, 1), dtype=torch.float32, device=p.device)
            ok_part = torch.zeros(max(parts, 1), dtype=torch.int32, device=p.device)
            N.check(lib.mvn_ce_on_probs_forward(p.data_ptr(), tg.data_ptr(), B, Q, S, loss_part.data_ptr(),
                                                ok_part.data_ptr(), _stream_ptr(p.device)),
                    "mvn_ce_on_probs_forward")
        n = max(B * S, 1)
        ctx_.save_for_backward(p, tg)
        ctx_.in_dtype = probs.dtype
        loss = loss_part.sum() / n
        acc = ok_part.sum().to(torch.float32) / n
        ctx_.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx_, dloss, _dacc):
        p, tg = ctx_.saved_tensors
        B, Q, S = p.shape
        dp = torch.empty_like(p)
        up = dloss.detach().to(device=p.device, dtype=torch.float32).reshape(1).contiguous()
        with torch.cuda.device(p.device):
            N.check(N.lib().mvn_ce_on_probs_backward(p.data_ptr(), tg.data_ptr(), B, Q, S,
                                                     1.0 / max(B * S, 1), up.data_ptr(), dp.data_ptr(),
                                                     _stream_ptr(p.device)), "mvn_ce_on_probs_backward")
        return dp.to(ctx_.in_dtype), None


def cross_entropy_on_probs(probs: torch.Tensor, target: torch.Tensor):
    """(loss, accuracy) of movenet/pytorch_lightning_trainer.py:64-66 in two kernels:
    ``F.cross_entropy(probs, target)`` -- a log-softmax applied to what already are
    probabilities -- and ``(probs.argmax(1) == target).float().mean()``."""
    return _CrossEntropyOnProbs.apply(probs, target)
