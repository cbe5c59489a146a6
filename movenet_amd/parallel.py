"""Data parallelism over clips: one process per GPU, full replica per rank,
ONE sum all-reduce of a flat gradient buffer per optimizer step (RCCL over xGMI
through torch.distributed's "nccl" backend; "gloo" on CPU for tests).

Precedent in the reference: DistributedDataParallel in the legacy trainer
(/root/reference/movenet/trainer.py:226-238, rendezvous :619-644).  This is not
a DDP translation: there are no buckets or autograd hooks -- the model's whole
gradient is 3.4 MB (30-layer, C=64), far below one xGMI ring step's
latency-bandwidth knee, so it travels as a single message after backward.
Only parameters that received a gradient take part (audio-only runs leave the
video/context parameters and the last layer's residual conv without one; the
reference needs find_unused_parameters=True for the same reason).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, port: str = "8888") -> tuple:
    """(rank, world, local_rank) from the torchrun environment; no-op when
    WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def contiguous_grad_span(used: List[torch.nn.Parameter]) -> Optional[torch.Tensor]:
    """One 1-D tensor covering every gradient of ``used`` when they are contiguous,
    non-overlapping views inside ONE storage -- in ANY order (the decoder's backward lays its
    gradients out in its own order, ops.decoder_param_names, not in registration order); gaps
    are allowed: parameters without a gradient lie there and the buffer is zero-filled, so a gap
    is zero on every rank.  Otherwise None."""
    if not used:
        return None
    g0 = used[0].grad
    store = g0.untyped_storage()
    spans = []
    for p in used:
        g = p.grad
        if (g.dtype != g0.dtype or g.device != g0.device or not g.is_contiguous() or
                g.untyped_storage().data_ptr() != store.data_ptr()):
            return None
        spans.append((g.storage_offset(), g.storage_offset() + g.numel()))
    spans.sort()
    for (_, end), (beg, _) in zip(spans, spans[1:]):
        if beg < end:  # two gradients share elements: not a partition of the buffer
            return None
    lo, hi = spans[0][0], spans[-1][1]
    return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(store, lo, (hi - lo,))


class FlatGradSync:
    """Broadcast parameters once, then average gradients with one all-reduce."""

    def __init__(self, params: Iterable[torch.nn.Parameter], world: Optional[int] = None,
                 single_rank_collective: bool = False):
        """``single_rank_collective``: run the all-reduce even in a world of one (a process
        group must exist) -- lets a one-GPU box execute the RCCL path itself."""
        self.single_rank_collective = bool(single_rank_collective)
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.world = world if world is not None else (
            dist.get_world_size() if dist.is_initialized() else 1)
        self._flat: Optional[torch.Tensor] = None
        self.last_path = "none"  # how the last sync_gradients() travelled
        self.last_floats = 0

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world <= 1:
            return
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self.params])
            dist.broadcast(flat, src=src)
            off = 0
            for p in self.params:
                n = p.numel()
                p.copy_(flat[off:off + n].view_as(p))
                off += n

    @staticmethod
    def _contiguous_span(used: List[torch.nn.Parameter]) -> Optional[torch.Tensor]:
        return contiguous_grad_span(used)

    def sync_gradients(self) -> int:
        """Average .grad over ranks; returns the number of floats sent.  The set
        of parameters with a gradient must be the same on every rank (it is: it
        depends only on the model and on use_video)."""
        used = [p for p in self.params if p.grad is not None]
        n = sum(p.grad.numel() for p in used)
        if n == 0 or (self.world <= 1 and not (self.single_rank_collective and dist.is_initialized())):
            self.last_path, self.last_floats = "none", 0
            return 0
        span = self._contiguous_span(used)
        if span is not None:  # the gradients already lie back to back in one buffer (ops.py)
            dist.all_reduce(span, op=dist.ReduceOp.SUM)
            span.div_(self.world)
            self.last_path, self.last_floats = "contiguous-span", span.numel()
            return span.numel()
        if self._flat is None or self._flat.numel() != n or self._flat.device != used[0].grad.device:
            self._flat = torch.empty(n, dtype=used[0].grad.dtype, device=used[0].grad.device)
        off = 0
        for p in used:
            k = p.grad.numel()
            self._flat[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
        self._flat.div_(self.world)
        off = 0
        for p in used:
            k = p.grad.numel()
            p.grad.copy_(self._flat[off:off + k].view_as(p.grad))
            off += k
        self.last_path, self.last_floats = "staged-copy", n
        return n
