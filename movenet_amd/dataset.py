"""Batches for the trainer entry point.

The reference's loader decodes Kinetics mp4 clips (movenet/dataset.py:59-364:
torchvision.io + torchaudio + pytorchvideo), none of which exist offline; that
storage/codec side is out of scope.  What the trainer
consumes is the ``Batch`` tuple ``(audio one-hot (B,Q,T), video|None, contexts,
filepaths, info)`` (dataset.py:186-203) and that contract is kept here, fed by
a synthetic source:

    --dataset synthetic://clips=64,frames=16000,seed=1234

Class indices are U{0..Q-1} (SURVEY.md section 8d "Synthetic inputs"); the
optional random contiguous crop ``batch_subsample_frac`` follows
dataset.py:232-242.
"""
from __future__ import annotations

import math
import random
from typing import Iterator, List, Optional

import numpy as np
import torch

from .utils.weights import one_hot


class Batch:
    def __init__(self, audio, video, contexts, filepaths, info):
        self.audio, self.video = audio, video
        self.contexts, self.filepaths, self.info = contexts, filepaths, info

    def pin_memory(self):
        self.audio = self.audio.pin_memory()
        if self.video is not None:
            self.video = self.video.pin_memory()
        return self

    def __iter__(self):
        yield from (self.audio, self.video, self.contexts, self.filepaths, self.info)


def parse_synthetic(spec: str) -> dict:
    if not spec.startswith("synthetic://"):
        raise ValueError(
            f"dataset {spec!r}: only synthetic://clips=N,frames=T[,seed=S] sources are built; "
            "the Kinetics mp4 decoder of the reference is out of scope (SURVEY.md section 2)")
    out = dict(clips=8, frames=16000, seed=1234)
    body = spec[len("synthetic://"):]
    for item in filter(None, body.split(",")):
        k, v = item.split("=")
        out[k.strip()] = int(v)
    return out


class SyntheticLoader:
    """Deterministic, shardable replacement for DataLoader(KineticsDataset)."""

    def __init__(self, spec: str, input_channels: int, batch_size: int, train: bool = True,
                 rank: int = 0, world_size: int = 1, shuffle: bool = False,
                 batch_subsample_frac: Optional[float] = None, use_video: bool = False,
                 device=None, **_ignored):
        cfg = parse_synthetic(spec)
        self.n_clips, self.frames = cfg["clips"], cfg["frames"]
        self.seed = cfg["seed"] + (0 if train else 10007)
        self.Q, self.batch_size = input_channels, batch_size
        self.rank, self.world = rank, max(world_size, 1)
        self.shuffle, self.frac = shuffle, batch_subsample_frac
        self.use_video = use_video
        # device (an MI355X) given: the class indices are shipped (B x T x 4 bytes) and the one-hot
        # (B,Q,T) tensor of the Batch contract is formed THERE (mvn_index_to_onehot, row F2) --
        # a 16 x 256 x 16000 fp32 one-hot is 262 MB of PCIe traffic per batch otherwise
        self.device = torch.device(device) if device is not None else None
        if use_video and (self.frames % 1000 or batch_subsample_frac is not None):
            raise ValueError("video batches need frames % 1000 == 0 and no batch_subsample_frac "
                             "(the reference crops audio and video independently, dataset.py:232-242, "
                             "which its own size assert then rejects)")
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _order(self) -> List[int]:
        order = list(range(self.n_clips))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order)
        # DistributedSampler semantics: pad to a multiple of world, stride by rank
        per = math.ceil(len(order) / self.world)
        order = (order + order[: per * self.world - len(order)])[self.rank::self.world]
        return order

    def __len__(self) -> int:
        return math.ceil(len(self._order()) / self.batch_size)

    def _clip(self, i: int) -> np.ndarray:
        rng = np.random.default_rng(self.seed * 1000003 + i)
        return rng.integers(0, self.Q, size=self.frames, dtype=np.int64)

    def __iter__(self) -> Iterator[Batch]:
        order = self._order()
        crop_rng = random.Random(self.seed * 31 + self.epoch * 7 + self.rank)
        for s in range(0, len(order), self.batch_size):
            ids = order[s:s + self.batch_size]
            idx = np.stack([self._clip(i) for i in ids])
            if self.device is not None and self.device.type == "cuda":
                audio = _one_hot_on_device(idx, self.Q, self.device)
            else:
                audio = one_hot(torch.from_numpy(idx), self.Q)
            if self.frac is not None:
                n = math.ceil(audio.shape[-1] * self.frac)
                start = crop_rng.randint(0, audio.shape[-1] - n)
                audio = audio[..., start:start + n]
            video = None
            if self.use_video:
                # U[0,1) frames, one 64x64 gray frame per 1000 samples
                # (movenet/wavenet.py:27-31: 160 frames <-> 160000 samples)
                vr = np.random.default_rng(self.seed * 7919 + 4321 + ids[0])
                frames = vr.random((len(ids), self.frames // 1000, 64, 64, 1), dtype=np.float32)
                if self.device is not None and self.device.type == "cuda":
                    video = _to_device_async(frames, self.device)
                else:
                    video = torch.from_numpy(frames)
            yield Batch(audio, video, ["synthetic"] * len(ids),
                        [f"synthetic://{i}" for i in ids],
                        [dict(video_fps=0.0, audio_fps=float(self.frames) / 10.0)] * len(ids))


class _PinnedRing:
    """A few persistent pinned host buffers the loader stages its batches in, taken in turn; an
    event behind each asynchronous copy says when its buffer may be overwritten (polled:
    generation.wait_event).  Pinned + asynchronous: a copy from pageable memory makes the host wait
    until everything queued on the stream -- the whole previous step -- has run."""

    def __init__(self, slots: int = 4):
        self.slots, self.next = [None] * slots, 0

    def stage(self, src: np.ndarray) -> tuple:
        """``src``: a numpy array.  It is copied into the slot with numpy (one thread): a torch CPU
        op of this size opens an OpenMP region on every visible core, whose threads then spin for
        their block time -- in a container with a CPU quota that alone can exhaust the quota of a
        scheduler period and freeze the whole process for the rest of it (r3: 60-90 ms stalls at
        arbitrary places of every second or third Trainer.fit step; none in loops without per-step
        CPU tensor ops)."""
        from .generation import wait_event
        i, self.next = self.next, (self.next + 1) % len(self.slots)
        slot = self.slots[i]
        dtype = torch.from_numpy(np.empty(0, dtype=src.dtype)).dtype
        if slot is None or tuple(slot[0].shape) != src.shape or slot[0].dtype != dtype:
            slot = self.slots[i] = [torch.empty(src.shape, dtype=dtype).pin_memory(), None]
        if slot[1] is not None:
            wait_event(slot[1])
        np.copyto(slot[0].numpy(), src)
        return slot

    @staticmethod
    def to_device(slot, device: torch.device) -> torch.Tensor:
        """Asynchronous copy on the CURRENT stream (pinned source: the host does not wait)."""
        out = slot[0].to(device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(device))
        return out


_RINGS: dict = {}


def _ring(kind: str, device: torch.device) -> _PinnedRing:
    key = (kind, device.index)
    if key not in _RINGS:
        _RINGS[key] = _PinnedRing()
    return _RINGS[key]


def _one_hot_on_device(idx: np.ndarray, Q: int, device: torch.device) -> torch.Tensor:
    """(B,T) integer host indices (numpy) -> (B,Q,T) fp32 one-hot on ``device`` through the C ABI (row F2).

    The indices (4 bytes per sample) cross PCIe from a persistent pinned staging buffer as an
    asynchronous copy on the training stream, which then expands them itself
    (mvn_index_to_onehot: ~70 us for 16 x 256 x 16000).  A copy from pageable memory would make
    the host wait until everything queued on that stream -- the whole previous step -- has run."""
    from . import _native as N
    B, T = idx.shape
    with torch.cuda.device(device):
        d_idx = _PinnedRing.to_device(_ring("idx", device).stage(np.ascontiguousarray(idx, dtype=np.int32)), device)
        out = torch.empty(B, Q, T, dtype=torch.float32, device=device)
        N.check(N.lib().mvn_index_to_onehot(d_idx.data_ptr(), d_idx.stride(0), out.data_ptr(), B, Q, T,
                                            torch.cuda.current_stream(device).cuda_stream), "mvn_index_to_onehot")
    return out


def _to_device_async(x: np.ndarray, device: torch.device) -> torch.Tensor:
    """Host array -> device through the pinned staging ring (see _one_hot_on_device)."""
    with torch.cuda.device(device):
        return _PinnedRing.to_device(_ring("video", device).stage(np.ascontiguousarray(x)), device)


def get_dataloader(filepath, input_channels: int, batch_size: int = 64, train: bool = True,
                   rank: int = 0, world_size: int = 0, use_video: bool = True,
                   normalize_audio: bool = True, batch_subsample_frac: Optional[float] = None,
                   **kwargs) -> SyntheticLoader:
    """Signature of movenet/dataset.py:59-98."""
    return SyntheticLoader(str(filepath), input_channels, batch_size, train=train, rank=rank,
                           world_size=world_size, shuffle=kwargs.get("shuffle", False),
                           batch_subsample_frac=batch_subsample_frac, use_video=use_video,
                           device=kwargs.get("device"))


# -- mu-law companding: the formula the project states (RESEARCH.md:156-163).
# The reference calls torchaudio.functional.mu_law_encoding/decoding, which is
# absent offline and pinned by no fixture in the reference => PARITY UNPINNED.
def mu_law_encoding(x: torch.Tensor, quantization_channels: int) -> torch.Tensor:
    mu = quantization_channels - 1.0
    x = x.to(torch.float32)
    y = torch.sign(x) * torch.log1p(mu * torch.abs(x)) / math.log1p(mu)
    return ((y + 1) / 2 * mu + 0.5).to(torch.int64)


def mu_law_decoding(q: torch.Tensor, quantization_channels: int) -> torch.Tensor:
    mu = quantization_channels - 1.0
    y = (q.to(torch.float32) / mu) * 2 - 1.0
    return torch.sign(y) * (torch.exp(torch.abs(y) * math.log1p(mu)) - 1.0) / mu
