"""MI355X-native WaveNet decoder behind the reference's model API.

Drop-in for ``movenet.wavenet.WaveNet`` (/root/reference/movenet/wavenet.py:50-239):
same constructor, same public attributes and module constants, same
``forward`` / ``generate`` / ``receptive_fields`` / ``compute_output_size`` /
``upsample_video`` signatures, same ``state_dict`` keys (so reference
checkpoints load with ``load_state_dict``), same exception types.  The
arithmetic runs in hand-written HIP kernels through the C ABI of
``include/movenet_hip.h``; tensors must live on an MI355X -- there is no CPU
or PyTorch-op fallback.

The sub-modules (``movenet_amd.modules``: the reference's five block classes by name)
hold the parameters under the reference's names and draw the same default
initialisation in the same order, so that the same ``torch.manual_seed`` yields the
same initial weights as the reference; their own ``forward`` is never used.
"""
from __future__ import annotations

import math
import sys
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _native as N
from .modules import CausalConv1d, DenseConv, ResidualConvStack
from .types import AudioTensor, VideoTensor  # noqa: F401  (re-exported like movenet/wavenet.py:15)
from .generation import (GroupedGenerator, HostWords, PipeHandoffTimeout, RingGenerator, _require_gpu,
                         _stream_ptr, auto_plan, max_pipe_batch)

# module constants other movenet files import (movenet/wavenet.py:27-31)
MAX_AUDIO_FRAMES = 160000
MAX_VIDEO_FRAMES = 160
VIDEO_KERNEL_SIZE = (1, 64, 64)
UPSAMPLE_STRIDE = 10


def upsample_kernel_size_solver(in_size, out_size, stride=1, padding=0, output_padding=0, dilation=1):
    """Kernel size k with (in-1)*stride - 2*padding + dilation*(k-1) + output_padding + 1 == out
    (the ConvTranspose1d length formula); same contract as movenet/wavenet.py:34-47."""
    span = out_size - 1 - output_padding - (in_size - 1) * stride + 2 * padding
    return (int(span / dilation + 1),)


class WaveNet(nn.Module):
    def __init__(self, layer_size: int, stack_size: int, input_channels: int,
                 residual_channels: int = 16, skip_channels: int = 16,
                 context_in_channels: int = 1):
        super().__init__()
        self.layer_size = layer_size
        self.stack_size = stack_size
        self.input_channels = input_channels
        self.residual_channels = residual_channels
        self.skip_channels = skip_channels
        C, K, Q = residual_channels, skip_channels, input_channels

        # registration order == the reference's (wavenet.py:94-123): same RNG draws
        self.video_conv = nn.Conv3d(context_in_channels, C, VIDEO_KERNEL_SIZE)
        n_up = math.ceil(np.log10(MAX_AUDIO_FRAMES / MAX_VIDEO_FRAMES) + 1)
        sizes = np.geomspace(MAX_VIDEO_FRAMES, MAX_AUDIO_FRAMES, num=n_up).astype(int)
        self.video_transpose = nn.Sequential(*[
            nn.ConvTranspose1d(C, C, upsample_kernel_size_solver(a, b, stride=UPSAMPLE_STRIDE),
                               stride=UPSAMPLE_STRIDE)
            for a, b in zip(sizes[:-1], sizes[1:])
        ])
        self.causal_conv = CausalConv1d(Q, C)
        self.residual_conv_stack = ResidualConvStack(layer_size, stack_size, C, K)
        self.dense_conv = DenseConv(K, Q)

        self._dims = N.make_dims(layer_size, stack_size, Q, C, K)
        self._gen_variant = N.GEN_AUTO
        self.last_generate_fallback = None  # variant a timed-out PIPE call was rerun on
        # "fp32" (default) or "fp16": fp16 operands / fp32 accumulation in every product of
        # forward() (mvn_forward_f16, any dims; inference only -- training stays fp32)
        self.forward_precision = "fp32"

    # ---- precision of generate() ----------------------------------------
    @property
    def generate_precision(self) -> str:
        """"fp32" (default: the reference's own precision) or "fp16": fp16 operands with fp32
        accumulation in every product of the autoregressive path (BASELINE configs[4]; the
        reference's precedent for reduced precision is torch.autocast, movenet/trainer.py:124).
        fp16 exists for C = K = 128, Q = 256 (kernel MVN_GEN_PIPE_F16)."""
        return "fp16" if self._gen_variant == N.GEN_PIPE_F16 else "fp32"

    @generate_precision.setter
    def generate_precision(self, value: str) -> None:
        if value == "fp32":
            self._gen_variant = N.GEN_AUTO
        elif value == "fp16":
            N.check(N.lib().mvn_gen_variant(self._dims, N.GEN_PIPE_F16, 1), "generate_precision = 'fp16'")
            self._gen_variant = N.GEN_PIPE_F16
        else:
            raise ValueError(f"generate_precision must be 'fp32' or 'fp16', got {value!r}")

    # ---- shape arithmetic (host only) ---------------------------------
    @property
    def receptive_fields(self) -> int:
        # sum of dilations + one more time point per stack (wavenet.py:125-134)
        return sum(self.residual_conv_stack.dilations) + self.stack_size

    def compute_output_size(self, x) -> int:
        out = int(x.size(2)) - self.receptive_fields + 1
        if out < 1:
            raise ValueError(
                "input time steps must be larger than the number of receptive "
                f"fields. Number of input timesteps = {x.size(2)}, "
                f"receptive fields = {self.receptive_fields}"
            )
        return out

    # ---- helpers --------------------------------------------------------
    def _indices_async(self, audio: torch.Tensor):
        """(B,Q,T) one-hot -> ((B,T) int32 indices on device, check token).  Nothing synchronises
        here: the indices' minimum is -1 where a column is not exactly one-hot, and the caller
        reads it (``_all_one_hot``) only AFTER it has enqueued the work that assumes one-hot
        input, so the host never waits on an idle GPU."""
        _require_gpu(audio, "audio")
        if audio.dim() != 3 or audio.size(1) != self.input_channels:
            raise ValueError(f"audio must be (batch, {self.input_channels}, frames), "
                             f"got {tuple(audio.shape)}")
        x = audio.detach().to(torch.float32).contiguous()
        B, Q, T = x.shape
        idx = torch.empty(B, T, dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            N.check(N.lib().mvn_onehot_to_index(x.data_ptr(), idx.data_ptr(), B, Q, T,
                                                _stream_ptr(x.device)), "mvn_onehot_to_index")
        low = idx.min() if B * T else torch.zeros((), dtype=torch.int32, device=x.device)
        # the minimum goes to pinned host memory from a one-wave kernel queued right behind the
        # kernels that produce it -- ahead of whatever the caller enqueues next (generation.HostWords;
        # nothing of it lives in the module: copy.deepcopy(model) / torch.save(model) keep working)
        return idx, HostWords.publish(low.reshape(1))

    def _all_one_hot(self, check) -> bool:
        """The result of ``_indices_async``'s check: polls the pinned word its publish kernel
        writes (no HIP call, no stream synchronisation: the host waits ONLY for the kernels that
        produced the minimum, not for whatever the caller has enqueued since)."""
        return HostWords.read(check)[0] >= 0

    def _indices_of(self, audio: torch.Tensor, strict: bool = True):
        """(B,Q,T) one-hot -> (B,T) int32 on device (mvn_onehot_to_index).  A column that
        is not one-hot: ValueError when ``strict`` (generation works on class indices),
        else None (forward then takes the dense causal-conv path)."""
        idx, check = self._indices_async(audio)
        if not self._all_one_hot(check):
            if not strict:
                return None
            raise ValueError(
                "movenet_amd.WaveNet expects one-hot audio (movenet/dataset.py:278-289); "
                "a column of the input is not one-hot")
        return idx

    def _one_hot_of(self, idx: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        B, T = idx.shape
        out = torch.empty(B, self.input_channels, T, dtype=torch.float32, device=idx.device)
        with torch.cuda.device(idx.device):
            N.check(N.lib().mvn_index_to_onehot(idx.data_ptr(), idx.stride(0), out.data_ptr(), B,
                                                self.input_channels, T, _stream_ptr(idx.device)),
                    "mvn_index_to_onehot")
        return out if dtype == torch.float32 else out.to(dtype)

    def _decoder_state(self):
        return {k: v for k, v in self.state_dict().items() if not k.startswith("video_")}

    # ---- model API ------------------------------------------------------
    def upsample_video(self, video):
        """(B, F, 64, 64, Cin) -> (B, C, 1000 F): wavenet.py:149-156 (pinned by fixture G7)."""
        from .ops import upsample_video
        out = upsample_video(self, video)
        assert out.shape[-1] == MAX_AUDIO_FRAMES  # same assert as the reference (:155)
        return out

    def forward(self, audio, video=None, global_features=None, output_unnormalized: bool = True,
                remove_last: bool = True, return_loss: bool = False, target=None):
        """BUILD DEFINITION for video != None: the reference raises a shape error at
        modules.py:75-77; here the upsampled video is added to the filter
        and gate pre-activations at the same absolute time (right-aligned, like the
        residual input at modules.py:84).

        ``return_loss=True`` (an extension; the default is the reference's signature and
        result): returns ``(loss, accuracy, probabilities)`` of the trainer's step -- the
        reference's ``output = self(audio, video)``, ``target = audio[:, :, RF:].argmax(1)``,
        ``F.cross_entropy(output, target)`` and accuracy (pytorch_lightning_trainer.py:62-66)
        -- as ONE autograd node (ops.wavenet_forward_loss); going through ``forward`` keeps
        module hooks firing on the fused path."""
        from .ops import wavenet_forward, wavenet_forward_loss  # HIP full-sequence kernels
        context = None if video is None else self.upsample_video(video)
        if return_loss:
            return wavenet_forward_loss(self, audio, context, target)
        return wavenet_forward(self, audio, context, output_unnormalized=output_unnormalized,
                               remove_last=remove_last)

    @torch.no_grad()
    def generate(self, audio, video=None, global_features=None, n_samples: Optional[int] = None,
                 temperature: float = 1.0):
        """wavenet.py:193-239: copy the first RF prompt samples, then generate
        autoregressively up to n_samples (default: the prompt's own length)."""
        self.eval()  # the reference leaves the module in eval mode
        # BUILD DEFINITION for video != None (the reference fails its size assert there,
        # SURVEY.md Q7): the context column of time t conditions the step that consumes x_t
        context = None if video is None else self.upsample_video(video)
        rf = self.receptive_fields
        n_total = int(audio.shape[2]) if n_samples is None else int(n_samples)
        idx = self._indices_of(audio)
        if idx.shape[1] < rf or n_total <= rf:
            # nothing to generate: the reference returns zeros with the prompt copied in
            out = torch.zeros(audio.shape[0], audio.shape[1], n_total, dtype=audio.dtype,
                              device=audio.device)
            n = min(rf, n_total, audio.shape[2])
            out[:, :, :n] = audio[:, :, :n]
            return out
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        state = self._decoder_state()
        if context is None:
            state = {k: v for k, v in state.items() if ".context_conv_" not in k}
        elif context.shape[2] < n_total:
            raise ValueError(f"the upsampled video covers {context.shape[2]} samples, "
                             f"n_samples={n_total} asked for")
        kw = dict(batch=idx.shape[0], n_total=n_total, device=audio.device,
                  temperature=float(temperature), seed=seed, context=context)
        def run(variant, group):
            if group:
                gen = GroupedGenerator(self.layer_size, self.stack_size, self.input_channels,
                                       self.residual_channels, self.skip_channels, state,
                                       group=group, variant=variant, **kw)
            else:
                gen = RingGenerator(self.layer_size, self.stack_size, self.input_channels,
                                    self.residual_channels, self.skip_channels, state,
                                    variant=variant, **kw)
            gen.prime(idx[:, :rf])
            gen.advance(n_total - rf)
            gen.check_errors()  # synchronises; never hand unchecked samples on
            return gen

        with torch.cuda.device(audio.device):
            if self._gen_variant == N.GEN_AUTO:
                kind, group, variant = auto_plan(self._dims, idx.shape[0], context is not None)
            else:
                kind, group, variant = "single", 0, self._gen_variant
                if variant == N.GEN_PIPE_F16 and idx.shape[0] > max_pipe_batch(self._dims, variant):
                    kind, group = "grouped", max_pipe_batch(self._dims, variant)  # groups take turns
        try:
            gen = run(variant, group if kind == "grouped" else 0)
        except PipeHandoffTimeout:
            # the pipelined kernel needs all its stages co-resident and something else held
            # CUs: rerun THIS call (same prompt, same seed) on a kernel without hand-offs
            with torch.cuda.device(audio.device):
                lib = N.lib()
                # (STREAM: C = K = 64, any batch in one launch, conditioned or not -- r4; GENERIC otherwise)
                fallback = N.GEN_STREAM if lib.mvn_gen_variant(
                    self._dims, N.GEN_STREAM, idx.shape[0]) == N.GEN_STREAM else N.GEN_GENERIC
            self.last_generate_fallback = fallback
            name = {N.GEN_STREAM: "STREAM", N.GEN_GENERIC: "GENERIC"}[fallback]
            print(f"[movenet_amd] generate: pipelined kernel (variant {variant}) timed out waiting for a "
                  f"co-resident stage; rerunning this call on {name}"
                  + (" -- fp32 arithmetic instead of the fp16 operands asked for" if variant == N.GEN_PIPE_F16
                     else ""), file=sys.stderr, flush=True)
            gen = run(fallback, 0)
        return self._one_hot_of(gen.samples, audio.dtype)
