"""Host side of the ring-buffer generator: owns the packed weights, the
dilation-queue state and the sample buffer as torch tensors (device memory and
stream plumbing only) and drives ``mvn_generate`` through the C ABI.

Replaces the per-sample window loop of /root/reference/movenet/wavenet.py:217-237.
"""
from __future__ import annotations

import threading

import os
from typing import Dict, Optional

import torch

from . import _native as N


class PipeHandoffTimeout(RuntimeError):
    """A stage of the PIPE generator waited for its predecessor beyond the spin bound: its
    workgroups were not co-resident (another process or stream held CUs).  The samples of
    that call are NOT valid."""


def wait_event(event, spin_s: float = 5.0) -> None:
    """Host wait for a recorded torch.cuda.Event by polling it (a core polling for the few
    milliseconds of a step costs nothing that matters: one process drives one GPU)."""
    import time
    t0 = time.perf_counter()
    spins = 0
    while not event.query():
        spins += 1
        if spins > 200:
            time.sleep(20e-6)
            if time.perf_counter() - t0 > spin_s:
                event.synchronize()
                return


class HostWords:
    """Device scalars read by the host WITHOUT a stream synchronisation, a second stream or a copy
    engine (``mvn_publish_words``): a ring of slots in pinned, device-visible host memory per
    device; a one-wave kernel queued on the producing stream writes up to 64 32-bit words and then
    a sequence number the host polls for.  ``publish`` returns a token, ``read`` spins on it.

    ``tensor.item()`` waits for everything enqueued on the stream so far -- read in the middle of a
    training step it would serialise the host's enqueueing with the GPU's execution.  Here the host
    waits exactly for the kernels in front of the publish kernel (the one-hot check of
    WaveNet.forward, published BEFORE the forward is enqueued; the trainer's log scalars, read one
    step late)."""
    RING, MAXW = 16, 64
    _rings: dict = {}
    _lock = threading.Lock()

    @classmethod
    def publish(cls, words: torch.Tensor):
        """``words``: 1-D device tensor of <= 64 int32 / float32 values (already computed or being
        computed on the current stream).  Returns the token ``read`` takes.  Thread-safe; when the
        ring wraps onto a slot whose token has not been read yet, that token is RESOLVED first (its
        values are copied out of the slot into the token), so a 17th publish never overwrites
        values somebody is still going to ask for."""
        n = int(words.numel())
        if not 1 <= n <= cls.MAXW or words.dtype not in (torch.int32, torch.float32) or not words.is_contiguous():
            raise ValueError("HostWords.publish: 1..64 contiguous int32 / float32 values")
        dev = words.device
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        with cls._lock:
            ring = cls._rings.get(key)
            if ring is None:
                ring = cls._rings[key] = [torch.zeros(cls.RING, cls.MAXW + 1, dtype=torch.int32).pin_memory(), 0,
                                          [None] * cls.RING]
            ring[1] += 1
            seq = (ring[1] & 0x3FFFFFFF) or 1
            at = ring[1] % cls.RING
            slot = ring[0][at]
            old = ring[2][at]
            if old is not None and old[4][0] is None:  # outstanding: wait for it, keep its values in the token
                old[4][0] = cls._wait(old)
            token = (slot, n, seq, words, [None])  # (words kept alive until read; [4]: values once resolved)
            ring[2][at] = token
            with torch.cuda.device(dev):
                N.check(N.lib().mvn_publish_words(words.data_ptr(), n, seq, slot.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream), "mvn_publish_words")
        return token

    @staticmethod
    def _wait(token, timeout_s: float = 30.0):
        import time
        slot, n, seq, words, _ = token
        t0 = time.perf_counter()
        spins = 0
        while int(slot[n]) != seq:
            spins += 1
            if spins > 200:
                # (a read that is not there after ~100 us is waiting for milliseconds of GPU work that
                # the caller has overlapped with its own: yield the core between polls -- eight ranks
                # of a node spinning flat out would eat eight cores of the container's quota)
                time.sleep(20e-6)
                if time.perf_counter() - t0 > timeout_s:  # (never seen: the blocking form as a last resort)
                    return words.tolist()
        vals = slot[:n]
        return vals.tolist() if words.dtype == torch.int32 else vals.view(torch.float32).tolist()

    @classmethod
    def read(cls, token, timeout_s: float = 30.0):
        """The published values as a list of Python ints / floats (by the dtype of ``words``)."""
        with cls._lock:  # (against a publish that wraps onto this token's slot at the same moment)
            if token[4][0] is None:
                token[4][0] = cls._wait(token, timeout_s)
        return token[4][0]


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"movenet_amd: {what} must live on an MI355X device (got {t.device}); "
            "there is no CPU path in this package"
        )


def pack_params(dims: N.Dims, sd: Dict[str, torch.Tensor], n_layers: int):
    """Build the mvn_params struct from a state_dict of fp32 device tensors.
    Returns (struct, keepalive) -- keepalive holds the pointer arrays and the
    contiguous tensors the struct points into."""
    keep = []

    def dev(key: str) -> int:
        t = sd[key]
        _require_gpu(t, key)
        if t.dtype != torch.float32:
            raise TypeError(f"{key}: expected float32, got {t.dtype}")
        t = t.detach().contiguous()
        keep.append(t)
        return t.data_ptr()

    def per_layer(name: str):
        ptrs = [dev(f"residual_conv_stack.conv_layers.{l}.{name}") for l in range(n_layers)]
        arr, cast = N.ptr_array(ptrs)
        keep.append(arr)
        return cast

    p = N.Params()
    p.causal_w = dev("causal_conv.conv.weight")
    p.filter_w = per_layer("conv_filter.conv.weight")
    p.gate_w = per_layer("conv_gate.conv.weight")
    p.residual_w = per_layer("conv_residual.weight")
    p.residual_b = per_layer("conv_residual.bias")
    p.skip_w = per_layer("conv_skip.weight")
    p.skip_b = per_layer("conv_skip.bias")
    has_ctx = "residual_conv_stack.conv_layers.0.context_conv_filter.weight" in sd
    if has_ctx:
        p.ctx_filter_w = per_layer("context_conv_filter.weight")
        p.ctx_filter_b = per_layer("context_conv_filter.bias")
        p.ctx_gate_w = per_layer("context_conv_gate.weight")
        p.ctx_gate_b = per_layer("context_conv_gate.bias")
    p.head1_w = dev("dense_conv.conv1.weight")
    p.head1_b = dev("dense_conv.conv1.bias")
    p.head2_w = dev("dense_conv.conv2.weight")
    p.head2_b = dev("dense_conv.conv2.bias")
    return p, keep


class RingGenerator:
    """One batch of sequences being generated on one GPU.

    >>> g = RingGenerator(cfg, state_dict, batch=16, n_total=19072, device="cuda:0")
    >>> g.prime(prompt_idx)            # (B, RF) int class indices
    >>> g.advance(16000)               # 16000 new samples per sequence
    >>> g.samples                      # (B, n_total) int32
    """

    def __init__(self, layer_size: int, stack_size: int, input_channels: int,
                 residual_channels: int, skip_channels: int, state_dict: Dict[str, torch.Tensor],
                 batch: int, n_total: int, device, variant: int = N.GEN_AUTO,
                 temperature: float = 0.0, seed: int = 0,
                 context: Optional[torch.Tensor] = None):
        self.lib = N.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("movenet_amd.RingGenerator needs an MI355X device; no CPU path exists")
        self.dims = N.make_dims(layer_size, stack_size, input_channels, residual_channels,
                                skip_channels)
        self.n_layers = layer_size * stack_size
        self.Q = input_channels
        self.rf = N.check(self.lib.mvn_receptive_fields(self.dims), "mvn_receptive_fields")
        self.batch, self.n_total = int(batch), int(n_total)
        with torch.cuda.device(self.device):
            self.variant = N.check(self.lib.mvn_gen_variant(self.dims, variant, self.batch),
                                   "mvn_gen_variant")
            # (r4: STREAM takes local conditioning too -- the context terms of all layers are formed at the top of a step)
        self.context = None      # (B, C, >= n_total) upsampled video as given
        self.context_tm = None   # (B, n_total, C) time-major copy the kernels read
        if context is not None:
            _require_gpu(context, "context")
            if context.shape[0] != batch or context.shape[1] != residual_channels or \
                    context.shape[2] < n_total:
                raise ValueError(f"context must be (batch, {residual_channels}, >= n_total), "
                                 f"got {tuple(context.shape)}")
            self.context = context.detach().to(torch.float32).contiguous()
            with torch.cuda.device(self.device):
                self.context_tm = torch.empty(self.batch, self.n_total, residual_channels,
                                              dtype=torch.float32, device=self.device)
                N.check(self.lib.mvn_transpose_context(
                    self.context.data_ptr(), self.context.stride(1), self.batch, residual_channels,
                    self.n_total, self.context_tm.data_ptr(), _stream_ptr(self.device)),
                    "mvn_transpose_context")
        self.temperature, self.seed = float(temperature), int(seed) & (2 ** 64 - 1)
        with torch.cuda.device(self.device):
            nw = self.lib.mvn_gen_weights_floats(self.dims, self.variant)
            ns = self.lib.mvn_gen_state_floats(self.dims, self.batch)
            self._queue_floats = self.batch * (self.rf - stack_size) * residual_channels
            # (MOVENET_DEBUG_GUARD=1, tests: a band of sentinels behind the packed weights and the state -- queues,
            # hand-off granules, placement words -- checked in check_errors())
            self._guard = None
            if os.environ.get("MOVENET_DEBUG_GUARD") == "1":
                band, sentinel = 1 << 16, -1234.5
                raw_p = torch.empty(nw + band, dtype=torch.float32, device=self.device)
                raw_s = torch.zeros(max(ns, 1) + band, dtype=torch.float32, device=self.device)
                raw_p[nw:].fill_(sentinel)
                raw_s[max(ns, 1):].fill_(sentinel)
                self.packed, self.state = raw_p[:nw], raw_s[:max(ns, 1)]
                self._guard = (sentinel, raw_p[nw:], raw_s[max(ns, 1):])
            else:
                self.packed = torch.empty(nw, dtype=torch.float32, device=self.device)
                self.state = torch.zeros(max(ns, 1), dtype=torch.float32, device=self.device)
            self.samples = torch.zeros(self.batch, self.n_total, dtype=torch.int32, device=self.device)
        self.t = 0          # number of time steps consumed so far
        self.n_given = 1
        # queues primed by one full-sequence forward (MFMA kernels) instead of stepping; the
        # fp16-operand variant primes with the fp16-operand forward: ONE arithmetic throughout
        self.prime_with_forward = True
        self.repack(state_dict)

    def repack(self, state_dict: Dict[str, torch.Tensor]) -> None:
        self._sd = state_dict
        params, keep = pack_params(self.dims, state_dict, self.n_layers)
        with torch.cuda.device(self.device):
            N.check(self.lib.mvn_gen_pack_weights(self.dims, self.variant, params,
                                                  self.packed.data_ptr(), _stream_ptr(self.device)),
                    "mvn_gen_pack_weights")
        # the source tensors must outlive the enqueued pack kernels
        torch.cuda.current_stream(self.device).synchronize()
        del keep

    def reset(self) -> None:
        self.state.zero_()
        self.t = 0

    def status_word(self) -> Optional[torch.Tensor]:
        """The PIPE variant's sticky status word (a 1-element int32 view into the state), or
        None.  Non-zero since the last ``reset()`` = a hand-off timed out; every later launch
        is then a no-op until the state is zeroed again."""
        if self.variant not in N.PIPE_VARIANTS:
            return None
        word = self.lib.mvn_gen_status_offset(self.dims, self.batch)
        return self.state[word:word + 1].view(torch.int32)

    def check_errors(self) -> None:
        """Synchronise and raise ``PipeHandoffTimeout`` if a PIPE hand-off timed out since
        the last ``reset()`` (workgroups of a pipeline not co-resident, e.g. another process
        occupying CUs).  Every product path calls this before handing samples on."""
        torch.cuda.current_stream(self.device).synchronize()
        if getattr(self, "_guard", None) is not None:
            sentinel, *bands = self._guard
            for band in bands:
                if bool((band != sentinel).any().item()):
                    raise RuntimeError("movenet_amd: a generator kernel wrote past the end of its packed weights / state")
        word = self.status_word()
        if word is not None and int(word[0].item()) != 0:
            raise PipeHandoffTimeout(
                "movenet_amd: PIPE generator hand-off timed out (pipeline stages not "
                "co-resident); the samples of this call are not valid")

    def _run(self, t_begin: int, t_end: int, n_given: int, logits_out=None, choices_out=None,
             logits_t0: int = 0) -> None:
        with torch.cuda.device(self.device):
            N.check(self.lib.mvn_generate(
                self.dims, self.variant, self.packed.data_ptr(), self.state.data_ptr(),
                self.samples.data_ptr(), self.batch, self.samples.stride(0), self.n_total, n_given,
                t_begin, t_end, self.temperature, self.seed,
                None if logits_out is None else logits_out.data_ptr(),
                None if choices_out is None else choices_out.data_ptr(),
                logits_t0, None if self.context_tm is None else self.context_tm.data_ptr(),
                _stream_ptr(self.device)), "mvn_generate")

    def prime(self, prompt_idx: torch.Tensor) -> None:
        """Load a (B, P) prompt (P >= 1 class indices per sequence) and run the
        network over all but its last sample so that the queues are primed."""
        _require_gpu(prompt_idx, "prompt indices")
        B, P = prompt_idx.shape
        if B != self.batch or P < 1 or P > self.n_total:
            raise ValueError(f"prompt shape {tuple(prompt_idx.shape)} does not fit "
                             f"(batch {self.batch}, n_total {self.n_total})")
        self.reset()
        self.samples.zero_()
        self.samples[:, :P] = prompt_idx.to(torch.int32)
        self.n_given = P
        if self.prime_with_forward and P >= self.rf:
            # one full-sequence forward over the prompt (MFMA kernels), then copy
            # each layer's most recent d_l inputs into its queue
            from .ops import run_forward
            idx = self.samples[:, :P].contiguous()
            ctx = None if self.context is None else self.context[:, :, :P]
            _, buf = run_forward(self.dims, self._sd, idx, False, False, save=True, ctx=ctx,
                                 f16=self.variant == N.GEN_PIPE_F16)
            with torch.cuda.device(self.device):
                N.check(self.lib.mvn_gen_prime_from_forward(
                    self.dims, buf.struct, self.batch, P, self.state.data_ptr(),
                    _stream_ptr(self.device)), "mvn_gen_prime_from_forward")
        else:
            self._run(0, P - 1, P)
        self.t = P - 1

    def advance(self, n_new: int) -> None:
        """Generate n_new further samples per sequence."""
        t_end = min(self.t + n_new, self.n_total - 1)
        self._run(self.t, t_end, self.n_given)
        self.t = t_end

    def teacher_forced(self, indices: torch.Tensor, logits_t0: int):
        """Feed a fully given (B, n_total) history; return (choices, logits) for
        times >= logits_t0 -- used by the parity tests."""
        _require_gpu(indices, "indices")
        assert indices.shape == (self.batch, self.n_total)
        self.reset()
        self.samples.copy_(indices.to(torch.int32))
        logits = torch.zeros(self.batch, self.n_total - logits_t0, self.Q, dtype=torch.float32,
                             device=self.device)
        choices = torch.full((self.batch, self.n_total), -1, dtype=torch.int32, device=self.device)
        self._run(0, self.n_total - 1, self.n_total, logits, choices, logits_t0)
        self.t = self.n_total - 1
        return choices, logits


# Measured cost model of a pipelined launch (DESIGN.md sections 4.1, 4.1c): its P pipelines serve
# ceil(n / P) sequences each in turn, and a step of ALL of them takes the pipeline's latency as long
# as the rounds fit under it, then rounds x the stages' service time per turn.  Keys: (C, variant);
# values: (us per step with one sequence per pipeline, us per step with several while the latency
# still bounds it, us per turn, most sequences per pipeline = GMAX of the kernel).
#   FOLD, config 2:  16 pipelines: 14.7 us for 16, 15.3 for 17 .. 80; beyond 80 sequences 23 pipelines (seven of
#                    them across XCDs: their slower hops set the latency): 16.4 us for 81 .. 128, 17.9 for 138,
#                    20.9 for 161, 23.2 for 184 (r2: 53 us for 64, 78 for 128)
#   PIPE, config 2:  24 pipelines: 17.4 us for 24 .. 96, 25.8 for 144, 34.4 for 192
#   PIPE, config 5:   4 pipelines: 72.4 us for 4, 73.2 for 5 .. 64 (the turn never bounds it)
_PIPELINED_US = {(64, N.GEN_FOLD): (14.7, 15.3, 2.64, 8), (64, N.GEN_PIPE): (17.4, 17.6, 4.3, 8),
                 (128, N.GEN_PIPE): (72.4, 73.2, 4.5, 16)}
_FOLD_CROSS_US = (16.4, 2.95)   # FOLD on all 23 pipelines: latency of a cross-XCD pipeline, us per turn
# ... and of the best kernel that takes EVERY sequence in one launch; keys: (C, conditioned):
# STREAM at C=64 without conditioning, GENERIC otherwise
# (r4: STREAM takes conditioning too -- (64, True) was GENERIC's 290 us)
_T_SINGLE_US = {(64, False): 78.0, (64, True): 100.0, (128, False): 490.0, (128, True): 490.0}

# The tables above were measured on ONE MI355X; boxes of a pool differ by up to 15 % and a cross-over between "k
# pipelined launches" and "one launch of the kernel that holds everything" moves with the RATIO of the two kernel
# families on the device at hand.  So the tables are only the SHAPE of the model: at first use per (device, C) the two
# families are timed once each on this device (`calibrate`) and the tables are scaled by measured / tabulated.
_CALIBRATION: dict = {}


def calibrate(dims, device=None, steps: int = 192) -> dict:
    """{"pipelined": measured / tabulated step time of one round of the pipelined kernel AUTO prefers, "single": the same
    for the one-launch kernel (STREAM or GENERIC)}, timed ONCE per (device, C, K, Q, L) with HIP events on synthetic weights
    (~0.1 s at config 2) and cached.  ``auto_plan`` multiplies its tables by these."""
    lib = N.lib()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (device.index, dims.layer_size, dims.stack_size, dims.input_channels, dims.residual_channels, dims.skip_channels)
    hit = _CALIBRATION.get(key)
    if hit is not None:
        return hit
    from .utils.weights import make_state_dict, synthetic_indices
    cfg = dict(layer_size=dims.layer_size, stack_size=dims.stack_size, input_channels=dims.input_channels,
               residual_channels=dims.residual_channels, skip_channels=dims.skip_channels)
    C = dims.residual_channels
    out = {"pipelined": 1.0, "single": 1.0}
    with torch.cuda.device(device):
        sd = {k: v.to(device) for k, v in make_state_dict(**cfg, seed=0).items()}
        rf = int(lib.mvn_receptive_fields(dims))
        one_launch = N.GEN_STREAM if lib.mvn_gen_variant(dims, N.GEN_STREAM, 1) == N.GEN_STREAM else N.GEN_GENERIC
        for kind, variant in (("pipelined", next((v for v in (N.GEN_FOLD, N.GEN_PIPE) if (C, v) in _PIPELINED_US and
                                                  lib.mvn_gen_variant(dims, v, 1) == v), None)),
                              ("single", one_launch)):
            if variant is None or (kind == "single" and (C, False) not in _T_SINGLE_US):
                continue
            n = max(1, lib.mvn_gen_launch_pipelines(dims, variant, 1 << 20)) if kind == "pipelined" else 4
            if kind == "pipelined":  # one round on the pipelines that sit inside one XCD (FOLD: 16 of the 23)
                n = max(1, lib.mvn_gen_launch_pipelines(dims, variant, min(n, 16)))
            g = RingGenerator(**cfg, state_dict=sd, batch=n, n_total=rf + 2 * steps + 2, device=device, variant=variant)
            g.prime(synthetic_indices(n, rf, dims.input_channels, 1234).to(device))
            g.advance(steps)  # warm-up (code objects, weights into place)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            g.advance(steps)
            ev[1].record()
            ev[1].synchronize()
            try:
                g.check_errors()
            except PipeHandoffTimeout:
                continue  # (a starved launch says nothing about the device: keep the table)
            us = ev[0].elapsed_time(ev[1]) * 1e3 / steps
            table = _PIPELINED_US[(C, variant)][0] if kind == "pipelined" else (
                _T_SINGLE_US[(C, False)] if variant == N.GEN_STREAM or C != 64 else 290.0)
            out[kind] = us / table
            out[kind + "_us_per_step"] = us
    _CALIBRATION[key] = out
    return out


def _launch_step_us(dims, variant: int, n: int):
    """Modelled step time (us) of ONE launch of ``variant`` holding ``n`` sequences, or None."""
    model = _PIPELINED_US.get((dims.residual_channels, variant))
    if model is None:
        return None
    t_one, t_multi, t_turn, gmax = model
    pipes = max(1, N.lib().mvn_gen_launch_pipelines(dims, variant, n))  # (FOLD: 16 up to 80 sequences, 23 beyond)
    rounds = -(-n // pipes)
    if variant == N.GEN_FOLD:
        lib = N.lib()
        every = lib.mvn_gen_launch_pipelines(dims, variant, 1 << 20)     # all the chip holds (config 2: 23)
        whole = lib.mvn_gen_launch_pipelines(dims, variant, 2 * every)   # those inside one XCD (16)
        if pipes > whole:                                                # the cross-XCD pipelines too
            t_multi, t_turn = _FOLD_CROSS_US
    return t_one if rounds <= 1 else max(t_multi, t_turn * rounds)


def auto_plan(dims, batch: int, has_context: bool, calibration=None):
    """``calibration``: ``{"pipelined": f, "single": f}`` scale factors of the two kernel families on the device at hand
    (default: ``calibrate(dims)``, measured once at first need -- only batches beyond ONE pipelined launch consult it;
    pass ``{"pipelined": 1.0, "single": 1.0}`` for the tables as measured on the reference box).

    What MVN_GEN_AUTO means at the Python level for ``batch`` sequences: returns
    ``("single", 0, variant)`` for one launch or ``("grouped", group, variant)`` for groups of
    ``group`` sequences taking turns on the pipelines of a pipelined variant.

    Chosen on measured per-step cost (the table above).  C=K=64: ONE FOLD launch up to 184 sequences
    (16 pipelines up to 80 sequences, 23 beyond, x 8 rounds), then balanced groups of FOLD launches as long
    as they beat the one-launch kernels (STREAM 78 us / conditioned GENERIC ~0.3 ms for any number); the
    PIPE kernel's 24 pipelines are the fallback of the C library's AUTO for 185 .. 192.  C=K=128: ONE PIPE launch up to 64 sequences (4
    pipelines x 16 rounds, 73 us whatever the number), groups of up to 64 until GENERIC's 490 us is
    cheaper (beyond 384)."""
    lib = N.lib()
    single = N.check(lib.mvn_gen_variant(dims, N.GEN_AUTO, batch), "mvn_gen_variant")
    C = dims.residual_channels
    best = None
    for variant in (N.GEN_FOLD, N.GEN_PIPE):
        cap = max_pipe_batch(dims, variant)
        if cap <= 0 or (C, variant) not in _PIPELINED_US:
            continue
        k = -(-batch // cap)          # launches per step
        group = -(-batch // k)        # balanced: the step time of a launch grows with its rounds
        cost = k * _launch_step_us(dims, variant, group)
        if best is None or cost < best[0]:
            best = (cost, k, group, variant)
    if best is None:
        return "single", 0, single
    cost, k, group, variant = best
    if k == 1:
        return "single", 0, variant
    cal = calibration if calibration is not None else calibrate(dims)
    if cost * cal.get("pipelined", 1.0) < _T_SINGLE_US.get((C, bool(has_context)), 0.0) * cal.get("single", 1.0):
        return "grouped", group, variant
    return "single", 0, single  # the kernel the C library's AUTO names: every sequence in one launch


def max_pipe_batch(dims, variant: int = N.GEN_PIPE) -> int:
    """Largest batch one PIPE launch holds co-resident for these dims (0: no PIPE kernel)."""
    lib = N.lib()
    if lib.mvn_gen_variant(dims, variant, 1) < 0:
        return 0
    lo, hi = 1, 512
    while lo < hi:  # mvn_gen_variant is monotone in the batch
        mid = (lo + hi + 1) // 2
        if lib.mvn_gen_variant(dims, variant, mid) >= 0:
            lo = mid
        else:
            hi = mid - 1
    return lo


class GroupedGenerator:
    """More sequences than one pipelined launch can hold (FOLD: 184 at config 2, PIPE: 192;
    C = 128: 64): groups of sequences take turns on the pipelines, one launch per group per
    ``advance``; ``auto_plan`` picks this as long as the groups' step times add up to less than
    one launch of a kernel that holds every sequence (STREAM: 78 us).
    Same interface as ``RingGenerator``; ``samples`` is one (B, n_total) tensor the groups
    write their row blocks of.  Each group draws from its own Philox key (seed + group)."""

    def __init__(self, layer_size, stack_size, input_channels, residual_channels, skip_channels,
                 state_dict, batch: int, n_total: int, device, group: int, temperature: float = 0.0,
                 seed: int = 0, context: Optional[torch.Tensor] = None, variant: int = N.GEN_PIPE):
        self.batch, self.n_total, self.device = int(batch), int(n_total), torch.device(device)
        self.variant = variant
        self.samples = torch.zeros(self.batch, self.n_total, dtype=torch.int32, device=self.device)
        self.groups, self.bounds = [], []
        for gi, b0 in enumerate(range(0, self.batch, group)):
            b1 = min(self.batch, b0 + group)
            g = RingGenerator(layer_size, stack_size, input_channels, residual_channels, skip_channels,
                              state_dict, batch=b1 - b0, n_total=n_total, device=device,
                              variant=variant, temperature=temperature,
                              seed=(int(seed) + 0x9E3779B97F4A7C15 * gi) & (2 ** 64 - 1),
                              context=None if context is None else context[b0:b1])
            g.samples = self.samples[b0:b1]  # a contiguous row block of the shared tensor
            self.groups.append(g)
            self.bounds.append((b0, b1))
        self.rf, self.dims = self.groups[0].rf, self.groups[0].dims

    def prime(self, prompt_idx: torch.Tensor) -> None:
        for g, (b0, b1) in zip(self.groups, self.bounds):
            g.prime(prompt_idx[b0:b1])

    def advance(self, n_new: int) -> None:
        for g in self.groups:
            g.advance(n_new)

    def check_errors(self) -> None:
        for g in self.groups:
            g.check_errors()
