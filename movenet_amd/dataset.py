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

    def _clip(self, i: int) -> torch.Tensor:
        rng = np.random.default_rng(self.seed * 1000003 + i)
        return torch.from_numpy(rng.integers(0, self.Q, size=self.frames, dtype=np.int64))

    def __iter__(self) -> Iterator[Batch]:
        order = self._order()
        crop_rng = random.Random(self.seed * 31 + self.epoch * 7 + self.rank)
        for s in range(0, len(order), self.batch_size):
            ids = order[s:s + self.batch_size]
            idx = torch.stack([self._clip(i) for i in ids])
            if self.device is not None and self.device.type == "cuda":
                audio = _one_hot_on_device(idx, self.Q, self.device)
            else:
                audio = one_hot(idx, self.Q)
            if self.frac is not None:
                n = math.ceil(audio.shape[-1] * self.frac)
                start = crop_rng.randint(0, audio.shape[-1] - n)
                audio = audio[..., start:start + n]
            video = None
            if self.use_video:
                # U[0,1) frames, one 64x64 gray frame per 1000 samples
                # (movenet/wavenet.py:27-31: 160 frames <-> 160000 samples)
                vr = np.random.default_rng(self.seed * 7919 + 4321 + ids[0])
                video = torch.from_numpy(
                    vr.random((len(ids), self.frames // 1000, 64, 64, 1), dtype=np.float32))
                if self.device is not None and self.device.type == "cuda":
                    video = _to_device_async(video, self.device)
            yield Batch(audio, video, ["synthetic"] * len(ids),
                        [f"synthetic://{i}" for i in ids],
                        [dict(video_fps=0.0, audio_fps=float(self.frames) / 10.0)] * len(ids))


_FEED_STREAMS: dict = {}


def _feed_stream(device: torch.device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _FEED_STREAMS.get(key)
    if st is None:
        st = _FEED_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def _one_hot_on_device(idx: torch.Tensor, Q: int, device: torch.device) -> torch.Tensor:
    """(B,T) int64 host indices -> (B,Q,T) fp32 one-hot on ``device`` through the C ABI.

    The indices (4 bytes per sample) cross PCIe from pinned memory on a FEED stream of their own;
    the consumer's stream waits for that copy's event and expands them itself (mvn_index_to_onehot:
    ~70 us for 16 x 256 x 16000).  A copy from pageable memory on the training stream would make the
    host wait until everything queued there -- the whole previous step -- has run, i.e. the loader
    would serialise host and GPU.  The large one-hot tensor is allocated on the TRAINING stream: a
    block handed from one stream's pool to another is not reusable until the other stream's work
    on it has been seen to finish, and the allocator then asks the driver for a new 262 MB block
    every step."""
    from . import _native as N
    B, T = idx.shape
    main = torch.cuda.current_stream(device)
    feed = _feed_stream(device)
    pinned = idx.to(torch.int32).pin_memory()
    with torch.cuda.device(device):
        with torch.cuda.stream(feed):
            d_idx = pinned.to(device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(feed)
        main.wait_event(ready)
        d_idx.record_stream(main)
        out = torch.empty(B, Q, T, dtype=torch.float32, device=device)
        N.check(N.lib().mvn_index_to_onehot(d_idx.data_ptr(), d_idx.stride(0), out.data_ptr(), B, Q, T,
                                            main.cuda_stream), "mvn_index_to_onehot")
    return out


def _to_device_async(x: torch.Tensor, device: torch.device) -> torch.Tensor:
    """Host tensor -> device through pinned memory on the feed stream (see _one_hot_on_device)."""
    main = torch.cuda.current_stream(device)
    feed = _feed_stream(device)
    pinned = x.pin_memory()
    with torch.cuda.device(device), torch.cuda.stream(feed):
        out = pinned.to(device, non_blocking=True)
        ready = torch.cuda.Event()
        ready.record(feed)
    main.wait_event(ready)
    out.record_stream(main)
    return out


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
