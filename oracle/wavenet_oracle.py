"""CPU oracle for the movenet WaveNet decoder path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the arithmetic of the
reference's hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``movenet_amd``) never does and fails loudly when its HIP library is missing.

Parity status: PINNED for the audio-only path -- ``tests/golden/make_golden.py``
imports the reference itself (``/root/reference/movenet/wavenet.py``) in the
build container, loads the same seeded weights into it and asserts that this
oracle reproduces its outputs (logits, probabilities, loss, greedy indices);
the resulting vectors are committed under ``tests/golden``.
Parity UNPINNED for the video-conditioned gated layer: the reference raises a
shape error there, so ``context`` alignment below
is this build's definition (right-aligned), not the reference's behaviour.
``upsample_video`` itself runs in the reference and is pinned.

Every function cites the reference lines it restates (paths are relative to
/root/reference).  Weights are passed as a flat state_dict with the
reference's key names, so no module hierarchy is mirrored here.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MAX_AUDIO_FRAMES = 160000  # movenet/wavenet.py:27
MAX_VIDEO_FRAMES = 160     # movenet/wavenet.py:28
UPSAMPLE_STRIDE = 10       # movenet/wavenet.py:31
LEAKY_SLOPE = 0.01         # F.leaky_relu default, movenet/modules.py:140-141


@dataclass(frozen=True)
class Dims:
    layer_size: int
    stack_size: int
    input_channels: int
    residual_channels: int = 16
    skip_channels: int = 16

    @property
    def n_layers(self) -> int:
        return self.layer_size * self.stack_size

    @property
    def dilations(self) -> List[int]:
        # movenet/modules.py:111-117
        return [2 ** x for _ in range(self.stack_size) for x in range(self.layer_size)]

    @property
    def receptive_fields(self) -> int:
        # movenet/wavenet.py:125-134
        return sum(self.dilations) + self.stack_size

    def output_size(self, t: int) -> int:
        # movenet/wavenet.py:136-147
        s = int(t) - self.receptive_fields + 1
        if s < 1:
            raise ValueError(
                "input time steps must be larger than the number of receptive "
                f"fields. Number of input timesteps = {t}, "
                f"receptive fields = {self.receptive_fields}"
            )
        return s


def _layer_key(l: int, name: str) -> str:
    return f"residual_conv_stack.conv_layers.{l}.{name}"


def causal_conv(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """movenet/modules.py:19-30: Conv1d(k=2, padding=1, no bias), drop last."""
    return F.conv1d(x, sd["causal_conv.conv.weight"], padding=1)[:, :, :-1]


def gated_layer(
    sd: Dict[str, torch.Tensor], l: int, d: int, x: torch.Tensor,
    context: Optional[torch.Tensor], skip_size: int,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """movenet/modules.py:67-93."""
    f = F.conv1d(x, sd[_layer_key(l, "conv_filter.conv.weight")], dilation=d)
    g = F.conv1d(x, sd[_layer_key(l, "conv_gate.conv.weight")], dilation=d)
    if context is not None:
        # BUILD DEFINITION (reference raises here, movenet/modules.py:75-77):
        # the context is right-aligned with f/g like the residual input is at
        # movenet/modules.py:84.
        ctx = context[:, :, -f.size(2):]
        f = f + F.conv1d(ctx, sd[_layer_key(l, "context_conv_filter.weight")],
                         sd[_layer_key(l, "context_conv_filter.bias")])
        g = g + F.conv1d(ctx, sd[_layer_key(l, "context_conv_gate.weight")],
                         sd[_layer_key(l, "context_conv_gate.bias")])
    gated = torch.tanh(f) * torch.sigmoid(g)
    residual = F.conv1d(gated, sd[_layer_key(l, "conv_residual.weight")],
                        sd[_layer_key(l, "conv_residual.bias")])
    residual = residual + x[:, :, -residual.size(2):]
    skip = F.conv1d(gated, sd[_layer_key(l, "conv_skip.weight")],
                    sd[_layer_key(l, "conv_skip.bias")])
    return residual, skip[:, :, -skip_size:]


def dense_head(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """movenet/modules.py:139-142."""
    x = F.conv1d(F.leaky_relu(x), sd["dense_conv.conv1.weight"], sd["dense_conv.conv1.bias"])
    return F.conv1d(F.leaky_relu(x), sd["dense_conv.conv2.weight"], sd["dense_conv.conv2.bias"])


def upsample_video(sd: Dict[str, torch.Tensor], video: torch.Tensor,
                   expect_frames: Optional[int] = MAX_AUDIO_FRAMES) -> torch.Tensor:
    """movenet/wavenet.py:149-156.  video (B,F,H,W,Cin) -> (B,C,1000*F)."""
    v = video.permute(0, 4, 1, 2, 3)
    enc = F.conv3d(v, sd["video_conv.weight"], sd["video_conv.bias"]).squeeze(-1).squeeze(-1)
    for i in range(3):
        enc = F.conv_transpose1d(enc, sd[f"video_transpose.{i}.weight"],
                                 sd[f"video_transpose.{i}.bias"], stride=UPSAMPLE_STRIDE)
    if expect_frames is not None:
        assert enc.shape[-1] == expect_frames
    return enc


def logits_full(sd: Dict[str, torch.Tensor], dims: Dims, audio: torch.Tensor,
                context: Optional[torch.Tensor] = None) -> torch.Tensor:
    """movenet/wavenet.py:166-181 up to the head: (B,Q,T) -> (B,Q,S) raw logits."""
    h = causal_conv(sd, audio)
    if context is not None:
        assert context.size() == h.size()  # movenet/wavenet.py:170-174
    skip_size = dims.output_size(h.size(2))
    skips = []
    for l, d in enumerate(dims.dilations):  # movenet/modules.py:125-130
        h, s = gated_layer(sd, l, d, h, context, skip_size)
        skips.append(s)
    return dense_head(sd, torch.sum(torch.stack(skips), dim=0))


def forward(sd: Dict[str, torch.Tensor], dims: Dims, audio: torch.Tensor,
            video: Optional[torch.Tensor] = None, output_unnormalized: bool = True,
            remove_last: bool = True, context: Optional[torch.Tensor] = None) -> torch.Tensor:
    """movenet/wavenet.py:158-191, including the inverted flag:
    ``output_unnormalized=True`` returns softmax probabilities."""
    if video is not None:
        context = upsample_video(sd, video)
    out = logits_full(sd, dims, audio, context)
    if remove_last:
        out = out[:, :, :-1]
    if not output_unnormalized:
        return out
    return F.softmax(out, dim=1)


def pre_sampling_probs(probs: torch.Tensor, temperature: float) -> torch.Tensor:
    """movenet/wavenet.py:227-233: the distribution generate() samples from /
    takes the argmax of.  ``probs`` is the (B,Q,1) output of forward() -- already
    softmaxed -- and is divided by T and softmaxed AGAIN."""
    if temperature > 0:
        probs = probs / temperature
    return F.softmax(probs, dim=1)


@torch.no_grad()
def generate_windowed(sd: Dict[str, torch.Tensor], dims: Dims, audio: torch.Tensor,
                      n_samples: Optional[int] = None, temperature: float = 1.0,
                      generator: Optional[torch.Generator] = None,
                      return_margins: bool = False, context: Optional[torch.Tensor] = None):
    """movenet/wavenet.py:193-239 -- the reference's NAIVE algorithm: one full
    forward over the last RF samples per generated sample.  This is what the
    CPU baseline times."""
    rf = dims.receptive_fields
    shape = tuple(audio.shape) if n_samples is None else (audio.shape[0], audio.shape[1], n_samples)
    gen = torch.zeros(shape, dtype=audio.dtype)
    gen[:, :, :rf] = audio[:, :, :rf]
    margins = []
    for i in range(rf, shape[-1]):
        # BUILD DEFINITION with context: the window sees the context columns of its own
        # times (the reference passes the whole context and fails its assert)
        ctx = None if context is None else context[:, :, i - rf:i]
        out = forward(sd, dims, gen[:, :, i - rf:i], output_unnormalized=True, remove_last=False,
                      context=ctx)
        assert out.shape[2] == 1
        p2 = pre_sampling_probs(out, temperature)
        if temperature > 0:
            choices = torch.multinomial(p2.squeeze(2), 1, generator=generator).unsqueeze(2)
        else:
            choices = p2.argmax(1, keepdim=True)
        if return_margins:
            top2 = torch.topk(torch.log(out.squeeze(2)), 2, dim=1).values
            margins.append((top2[:, 0] - top2[:, 1]))
        gen[:, :, [i]] = torch.zeros_like(out, dtype=audio.dtype).scatter_(1, choices, 1)
    if return_margins:
        return gen, (torch.stack(margins, 1) if margins else torch.zeros(shape[0], 0))
    return gen


# ---------------------------------------------------------------------------
# Ring-buffer ("fast WaveNet") restatement, numpy fp32, one sample at a time.
# Independent of torch's conv kernels: used to show on CPU that the cached
# formulation the HIP kernels implement is result-equivalent to the windowed
# reference algorithm, and as the checker for teacher-forced
# per-step logits at sizes where the windowed algorithm is too slow.
# ---------------------------------------------------------------------------
class RingState:
    """``operand_dtype=np.float16`` restates the fp16-OPERAND / fp32-ACCUMULATE form of
    BASELINE configs[4] (reference precedent: torch.autocast, movenet/trainer.py:124): every
    weight matrix and the vector operand of every product is rounded to fp16 (nearest even),
    products and sums stay fp32, and so do the residual stream, the skip sum, biases, the
    gate, the embedding rows (a gather: no product) and the queues.  It is the checker of the
    MVN_GEN_PIPE_F16 kernel; no reference output exists at this precision (parity against the
    fp32 fixtures is a TOLERANCE statement, tests/test_fp16_gpu.py)."""

    def __init__(self, sd: Dict[str, torch.Tensor], dims: Dims, batch: int, operand_dtype=None):
        self.dims = dims
        self.B = batch
        self.rnd = (lambda a: a) if operand_dtype is None else (
            lambda a: np.asarray(a, np.float32).astype(operand_dtype).astype(np.float32))
        g32 = lambda k: sd[k].detach().cpu().numpy().astype(np.float32)
        g = lambda k: g32(k) if (k.endswith("bias") or k.startswith("causal_conv")) else self.rnd(g32(k))
        self.E0 = g("causal_conv.conv.weight")[:, :, 0]  # multiplies x[t-1]
        self.E1 = g("causal_conv.conv.weight")[:, :, 1]  # multiplies x[t]
        self.layers = []
        for l, d in enumerate(dims.dilations):
            wf = g(_layer_key(l, "conv_filter.conv.weight"))
            wg = g(_layer_key(l, "conv_gate.conv.weight"))
            self.layers.append(dict(
                d=d, wf0=wf[:, :, 0], wf1=wf[:, :, 1], wg0=wg[:, :, 0], wg1=wg[:, :, 1],
                wcf=g(_layer_key(l, "context_conv_filter.weight"))[:, :, 0],
                bcf=g(_layer_key(l, "context_conv_filter.bias")),
                wcg=g(_layer_key(l, "context_conv_gate.weight"))[:, :, 0],
                bcg=g(_layer_key(l, "context_conv_gate.bias")),
                wr=g(_layer_key(l, "conv_residual.weight"))[:, :, 0],
                br=g(_layer_key(l, "conv_residual.bias")),
                ws=g(_layer_key(l, "conv_skip.weight"))[:, :, 0],
                bs=g(_layer_key(l, "conv_skip.bias")),
                ring=np.zeros((d, batch, dims.residual_channels), np.float32),
            ))
        self.w1 = g("dense_conv.conv1.weight")[:, :, 0]
        self.b1 = g("dense_conv.conv1.bias")
        self.w2 = g("dense_conv.conv2.weight")[:, :, 0]
        self.b2 = g("dense_conv.conv2.bias")
        self.t = 0
        self.prev = None  # index at t-1 (None => x[-1] = 0, the conv's zero pad)

    def step(self, idx: np.ndarray, ctx_t: Optional[np.ndarray] = None) -> np.ndarray:
        """Consume the sample at time t (class indices, shape (B,)), return the
        raw logits (B,Q) predicting time t+1.  Logits are only meaningful once
        t >= RF-1 (before that the valid-convolution stack has no output).
        ctx_t: optional (B,C) context column of time t (build-defined alignment)."""
        h = self.E1[:, idx].T.copy()
        if self.prev is not None:
            h += self.E0[:, self.prev].T
        skip = np.zeros((self.B, self.dims.skip_channels), np.float32)
        for L in self.layers:
            slot = self.t % L["d"]
            past = L["ring"][slot].copy()
            L["ring"][slot] = h
            ho, po = self.rnd(h), self.rnd(past)  # operands (identity for fp32)
            f = po @ L["wf0"].T + ho @ L["wf1"].T
            g = po @ L["wg0"].T + ho @ L["wg1"].T
            if ctx_t is not None:
                co = self.rnd(ctx_t)
                f = f + (co @ L["wcf"].T + L["bcf"])
                g = g + (co @ L["wcg"].T + L["bcg"])
            z = np.tanh(f) * (1.0 / (1.0 + np.exp(-g)))
            z = self.rnd(z.astype(np.float32))
            skip += z @ L["ws"].T + L["bs"]
            h = z @ L["wr"].T + L["br"] + h
        a = np.where(skip > 0, skip, LEAKY_SLOPE * skip).astype(np.float32)
        a = self.rnd(a) @ self.w1.T + self.b1
        a = np.where(a > 0, a, LEAKY_SLOPE * a).astype(np.float32)
        out = self.rnd(a) @ self.w2.T + self.b2
        self.prev = idx.copy()
        self.t += 1
        return out.astype(np.float32)


def generate_ring(sd: Dict[str, torch.Tensor], dims: Dims, prompt_idx: np.ndarray,
                  n_samples: int, forced_idx: Optional[np.ndarray] = None,
                  context: Optional[np.ndarray] = None, operand_dtype=None):
    """Greedy (temperature<=0) ring-buffer generation.  prompt_idx (B, >=RF).
    Returns (choices (B,n_samples) int64, logits (B, n_samples-RF, Q)); the
    first RF columns of ``choices`` are the prompt.  With ``forced_idx``
    (B,n_samples) the history fed back is teacher-forced while ``choices``
    still reports what the model would have picked at each step."""
    rf = dims.receptive_fields
    B = prompt_idx.shape[0]
    st = RingState(sd, dims, B, operand_dtype)
    choices = np.zeros((B, n_samples), np.int64)
    choices[:, :rf] = prompt_idx[:, :rf]
    logits = np.zeros((B, max(n_samples - rf, 0), dims.input_channels), np.float32)
    lg = None
    for t in range(n_samples):
        if t >= rf:
            logits[:, t - rf] = lg
            choices[:, t] = _double_softmax_argmax(lg)
        fed = choices[:, t] if (forced_idx is None or t < rf) else forced_idx[:, t]
        lg = st.step(fed, None if context is None else context[:, :, t])
    return choices, logits


def _double_softmax_argmax(logits: np.ndarray) -> np.ndarray:
    t = torch.from_numpy(logits)
    return F.softmax(F.softmax(t, dim=1), dim=1).argmax(1).numpy()


# ---------------------------------------------------------------------------
# Trainer arithmetic (movenet/pytorch_lightning_trainer.py:62-66)
# ---------------------------------------------------------------------------
def train_step_arithmetic(sd: Dict[str, torch.Tensor], dims: Dims, audio: torch.Tensor,
                          video: Optional[torch.Tensor] = None,
                          context: Optional[torch.Tensor] = None):
    """loss = cross_entropy(PROBABILITIES, target), accuracy, and the
    gradient of the loss w.r.t. every parameter that took part."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    out = forward(params, dims, audio, video=video, context=context)
    target = audio[:, :, dims.receptive_fields:].argmax(1)
    loss = F.cross_entropy(out, target)
    acc = (out.argmax(1) == target).float().mean()
    loss.backward()
    grads = {k: p.grad for k, p in params.items() if p.grad is not None}
    return loss.detach(), acc, out.detach(), grads
