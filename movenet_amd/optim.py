"""AdamW / Adam over ONE flat parameter buffer: one HIP kernel per optimizer step
(``mvn_adamw_step``) instead of torch's per-tensor / foreach kernels (row F3 of SURVEY.md
section 8; the optimizer the reference's ``configure_optimizers`` builds,
/root/reference/movenet/pytorch_lightning_trainer.py:186-189).

The parameters keep their identity (``nn.Parameter`` objects, ``state_dict`` names) but their
storage is re-pointed into one flat fp32 buffer, in the order the decoder's backward lays its
gradients out (``ops.decoder_param_names``).  ``ops._run_backward`` returns the gradients as
views of one flat buffer in the same order, so a step is ONE launch over
(parameters, gradients, exp_avg, exp_avg_sq); parameters without a gradient (the last layer's
residual conv, unused video / context parameters) are skip ranges of that launch, exactly as
torch leaves them alone.  Gradients that do not lie in one buffer (the eight video-encoder
tensors, or a caller's own) are stepped with one launch of the same kernel per tensor.

A ``torch.optim.Optimizer`` subclass: ``param_groups`` (so the torch LR schedulers drive
``lr``), ``zero_grad`` and ``state_dict`` behave as usual; one parameter group.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, List, Optional, Tuple

import torch

from . import _native as N


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, decoupled: bool = True):
        params = [p for p in params]
        if not params:
            raise ValueError("FlatAdamW: no parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("movenet_amd.FlatAdamW steps on an MI355X device; no CPU path exists")
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatAdamW: all parameters must be fp32 on one device")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      decoupled=decoupled))
        self._params: List[torch.nn.Parameter] = params
        self._offsets: List[int] = []
        off = 0
        for p in params:
            self._offsets.append(off)
            off += p.numel()
        self._n = off
        with torch.no_grad():
            self.flat = torch.empty(off, dtype=torch.float32, device=dev)
            for p, o in zip(params, self._offsets):
                self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)  # same values, flat storage
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        # torch counts steps PER PARAMETER (one that first receives a gradient at step 4 is
        # bias-corrected as step 1): kept here on the host, a run of parameters shares one count
        self._steps: List[int] = [0] * len(params)
        self.state["flat"] = {"steps": self._steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}
        self.last_launches = 0

    # ------------------------------------------------------------------
    def load_state_dict(self, state_dict) -> None:
        """torch's ``Optimizer.load_state_dict`` REPLACES ``self.state`` with new objects; the
        moments this optimizer steps over are the flat buffers made in the constructor, so the
        loaded moments and per-parameter step counts are copied INTO them (and ``self.state``
        is pointed back at them): a resumed run continues where the saved one stopped."""
        flat = state_dict["state"].get("flat") if "state" in state_dict else None
        if flat is None:
            raise ValueError("FlatAdamW.load_state_dict: no 'flat' entry (not a FlatAdamW state_dict)")
        steps, m, v = list(flat["steps"]), flat["exp_avg"], flat["exp_avg_sq"]
        if len(steps) != len(self._params) or m.numel() != self._n or v.numel() != self._n:
            raise ValueError(f"FlatAdamW.load_state_dict: saved state covers {len(steps)} parameters / "
                             f"{m.numel()} elements, this optimizer {len(self._params)} / {self._n}")
        super().load_state_dict(state_dict)  # param_groups (lr, betas, ...) as torch does it
        with torch.no_grad():
            self.exp_avg.copy_(m.reshape(-1))
            self.exp_avg_sq.copy_(v.reshape(-1))
        self._steps[:] = [int(x) for x in steps]
        self.state["flat"] = {"steps": self._steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    def _launch(self, p_ptr, g_ptr, m_ptr, v_ptr, n, step, skips, stream) -> None:
        grp = self.param_groups[0]
        arr = (ctypes.c_size_t * max(2 * len(skips), 1))(*[x for r in skips for x in r])
        N.check(N.lib().mvn_adamw_step(p_ptr, g_ptr, m_ptr, v_ptr, n, float(grp["lr"]),
                                       float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                                       float(grp["weight_decay"]), step, int(bool(grp["decoupled"])),
                                       arr, len(skips), stream), "mvn_adamw_step")
        self.last_launches += 1

    def _spans(self):
        """Greedy partition of the parameters (in flat order) into runs whose gradients mirror
        the flat parameter layout inside ONE gradient storage: (first, last_exclusive, skips)."""
        runs, i, P = [], 0, self._params
        while i < len(P):
            g = P[i].grad
            if g is None:
                i += 1
                continue
            ok = g.dtype == torch.float32 and g.is_contiguous() and g.device == self.flat.device
            if not ok:
                raise ValueError("FlatAdamW: gradients must be contiguous fp32 on the parameters' device")
            store, base = g.untyped_storage().data_ptr(), g.storage_offset() - self._offsets[i]
            j, skips, pending = i + 1, [], None
            last = i + 1
            while j < len(P):
                gj = P[j].grad
                if gj is None:
                    if pending is None:
                        pending = j
                    j += 1
                    continue
                same = (gj.dtype == torch.float32 and gj.is_contiguous()
                        and gj.untyped_storage().data_ptr() == store
                        and gj.storage_offset() - self._offsets[j] == base
                        and self._steps[j] == self._steps[i])
                if not same:
                    break
                if pending is not None:
                    if len(skips) == 4:
                        break
                    skips.append((self._offsets[pending] - self._offsets[i],
                                  self._offsets[j] - self._offsets[i]))
                    pending = None
                j += 1
                last = j
            runs.append((i, last, skips))
            i = last
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        dev = self.flat.device
        self.last_launches = 0
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for first, last, skips in self._spans():
                step = self._steps[first] + 1
                for k in range(first, last):
                    if self._params[k].grad is not None:
                        self._steps[k] = step
                o0 = self._offsets[first]
                end = self._offsets[last - 1] + self._params[last - 1].numel()
                g0 = self._params[first].grad
                self._launch(self.flat.data_ptr() + 4 * o0, g0.data_ptr(), self.exp_avg.data_ptr() + 4 * o0,
                             self.exp_avg_sq.data_ptr() + 4 * o0, end - o0, step, skips, stream)
        return loss


def order_like_backward(model, with_context: bool = False) -> List[torch.nn.Parameter]:
    """The model's parameters in the order ops._run_backward lays their gradients out for an
    audio-only (``with_context=False``) or a video-conditioned run, then everything else
    (context convs of an audio-only run, video encoder).  With the matching order a step is
    one launch (plus one per video tensor); with the other order it is still correct, in
    more launches."""
    from .ops import decoder_param_names
    lookup = dict(model.named_parameters())
    L = model.layer_size * model.stack_size
    names = decoder_param_names(L, with_context=with_context)
    seen = set(names)
    ordered = [lookup[n] for n in names] + [p for n, p in lookup.items() if n not in seen]
    return [p for p in ordered if p.requires_grad]
