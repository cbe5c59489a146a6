#!/usr/bin/env python3
"""Headline benchmark: autoregressive audio samples/sec on BASELINE config 2.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d "Config 2"): 30-layer
(10 x 3 dilation cycles) WaveNet, Q=256 mu-law classes, C=K=64, batch 16
sequences per GPU, fp32, synthetic random prompt of RF=3072 samples, seeded
random-init weights.  One STEP = one pass of the hot path over the batch:
16000 new samples (1 s of 16 kHz audio) for each of the 16 sequences, greedy
(temperature 0; the mode the reference's own test uses).  Queue priming over
the prompt happens before the timed region and is reported separately.
Multi-GPU = independent clips per rank (weak scaling), no data-path collective.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself
(one child process per device, spawned BEFORE anything touches the GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
BATCH = 16
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: "Peak FP32 (vector)" = "Peak FP32 (matrix)"
BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/F16 ~2.5 PF dense (v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD)
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PROFILE_TAG = "r04"              # the round whose rocprofv3 passes the profiled figures below are read from


def profiled(section: str, key: str = None):
    """A figure of THIS round's counter passes (profiles/<tag>_pmc_summary.json, written by
    scripts/profile_summary.py from scripts/profile_round.sh), or None when the file is not there: numbers that the
    run printing the line cannot measure itself are quoted with their source file, never as its own."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_summary.json")
    try:
        with open(path) as f:
            d = json.load(f).get(section)
        return d if key is None or d is None else d.get(key)
    except (OSError, ValueError):
        return None


def fold_stamps():
    """Per-stage / hop / head times of the FOLD generator from the stamped build (scripts/pipe_stamps.py --fold --json),
    tracked as profiles/<tag>_fold_stamps.json; None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_fold_stamps.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None

def headline_traffic(variant_used: int, n_new: int):
    """HBM-side bytes per launch of the generator kernel from THIS round's --pmc passes (profiles/), or None."""
    name = {3: "gen_pipe_kernel", 5: "gen_fold_kernel"}.get(variant_used)
    k = profiled("kernels") or {}
    hit = next((v for kk, v in k.items() if name and kk.startswith(name) and v.get("FETCH_SIZE_KiB_as_reported") is not None), None)
    if not hit:
        return None
    nbytes = (hit["FETCH_SIZE_KiB_as_reported"] + hit["WRITE_SIZE_KiB"]) * 1024.0
    return {"bytes_per_launch": nbytes, "bytes_per_sample_per_sequence": nbytes / (BATCH * n_new),
            "source": f"profiles/{PROFILE_TAG}_pmc_summary.json kernels.{name}* (FETCH_SIZE as reported + WRITE_SIZE, separate --pmc "
                      f"passes of this command with {n_new}-step launches of {BATCH} sequences); NOT measured by this run",
            "note": "far below the algorithmic 15,360 B per sample: the 12.6 MB of dilation queues stay in L2 / MALL"}


def latency_floor(variant_used: int, us_per_step: float) -> dict:
    """The step of the pipelined generator against ITS bound: the dependent chain of a sample -- stages x (chain + hop) +
    head -- from the in-kernel stamps of the diagnostic build (profiles/<tag>_fold_stamps.json; the stamps build runs
    ~8 % slower than the product, so the sum is scaled by product step / stamped step)."""
    st = fold_stamps() if variant_used == 5 else None
    if not st:
        return {"latency_floor_us": None, "frac_of_latency_floor": None}
    scale = st.get("product_over_stamped", 1.0)
    floor = (sum(st["stage_chain_us"]) + sum(st["hop_us"]) + st["head_us"]) * scale
    return {"latency_floor_us": floor, "frac_of_latency_floor": floor / us_per_step,
            "latency_floor_parts": {"stages": len(st["stage_chain_us"]), "stage_chain_us_mean": sum(st["stage_chain_us"]) / len(st["stage_chain_us"]),
                                    "hop_us_mean": sum(st["hop_us"]) / len(st["hop_us"]), "head_us": st["head_us"],
                                    "stamped_step_us": st["step_us"], "scale_product_over_stamped": scale,
                                    "source": f"profiles/{PROFILE_TAG}_fold_stamps.json (scripts/pipe_stamps.py --fold --json)"}}


def flop_per_sample(cfg) -> int:
    """MACs = L(5C^2 + CK) + KQ + Q^2 per generated sample
    per sequence (the causal conv on a one-hot input is a gather)."""
    L = cfg["layer_size"] * cfg["stack_size"]
    C, K, Q = cfg["residual_channels"], cfg["skip_channels"], cfg["input_channels"]
    return 2 * (L * (5 * C * C + C * K) + K * Q + Q * Q)


def log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """Threads for the CPU baseline: the process's CPU share (affinity mask, capped by the cgroup CPU
    quota), overridable with MOVENET_CPU_THREADS."""
    from movenet_amd.utils.host import cpu_share
    return min(cpu_share(), 64)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_cached_line(sd, n_timed: int = 4000, n_warm: int = 200):
    """The SAME ring-buffer (cached) algorithm the HIP kernels run, on the host cores: the C
    restatement oracle/ring_oracle.c (gcc -O3 -march=x86-64-v3, OpenMP over the 16 sequences;
    pinned to the reference's greedy output by tests/test_ring_c_oracle.py).  Reported beside
    the naive windowed baseline so that the algorithmic gain (windowed -> cached) and the
    hardware gain (CPU -> MI355X on the same algorithm) can be told apart."""
    import numpy as np
    from oracle import ring_c
    from oracle import wavenet_oracle as O
    dims = O.Dims(**CFG)
    threads = min(host_cores(), BATCH)
    ring = ring_c.RingC(sd, dims, BATCH)
    n = n_warm + n_timed + 1
    samples = np.random.default_rng(0).integers(0, CFG["input_channels"], size=(BATCH, n)).astype(np.int32)
    ring.run(samples, 1, 0, n_warm, threads)           # free-running greedy from one given sample
    t0 = time.perf_counter()
    ring.run(samples, 1, n_warm, n_warm + n_timed, threads)
    dt = time.perf_counter() - t0
    return {"value": BATCH * n_timed / dt, "unit": "samples/s", "cores": threads, "cpu_model": cpu_model(),
            "kind": "port (cached / ring-buffer algorithm, C: oracle/ring_oracle.c)",
            "sample": f"{n_timed} greedy steps at batch {BATCH} after {n_warm} warm-ups, {dt:.2f} s",
            "us_per_step": dt / n_timed * 1e6}


def cpu_baseline(sd, n_timed: int = 16, n_warm: int = 2):
    """The reference's NAIVE windowed generate (wavenet.py:217-237) on the host
    cores, via the CPU oracle (kind "port": pinned bit-exact to the reference in
    the build container by tests/golden/make_golden.py).  Per-step cost is
    constant (the window is always RF long), so a bounded sample suffices."""
    from oracle import wavenet_oracle as O
    from movenet_amd.utils.weights import one_hot, synthetic_indices
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads")
    dims = O.Dims(**CFG)
    rf = dims.receptive_fields
    prompt = one_hot(synthetic_indices(BATCH, rf, CFG["input_channels"], 1234), CFG["input_channels"])
    t0 = None
    with torch.no_grad():
        # generate_windowed runs exactly the reference loop; time n_timed steady steps
        gen = torch.zeros(BATCH, CFG["input_channels"], rf + n_warm + n_timed)
        gen[:, :, :rf] = prompt
        for i in range(rf, rf + n_warm + n_timed):
            if i == rf + n_warm:
                t0 = time.perf_counter()
            log(f"cpu_baseline: window forward {i - rf + 1}/{n_warm + n_timed}")
            out = O.forward(sd, dims, gen[:, :, i - rf:i], output_unnormalized=True, remove_last=False)
            choice = O.pre_sampling_probs(out, 0.0).argmax(1, keepdim=True)
            gen[:, :, [i]] = torch.zeros_like(out).scatter_(1, choice, 1)
        dt = time.perf_counter() - t0
    return {
        "value": BATCH * n_timed / dt,
        "unit": "samples/s",
        "cores": cores,
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": f"{n_timed} steady-state steps of the reference's naive windowed generate "
                  f"(RF={rf} window forward per sample) at batch {BATCH} after {n_warm} warm-ups, "
                  f"torch CPU fp32, {dt:.2f} s",
        "samples_per_s_per_sequence": n_timed / dt,
    }


def cpu_train_baseline(sd, batch: int = 2, t_len: int = 16000, n_timed: int = 3, n_warm: int = 1):
    """the reference's train step -- forward, cross_entropy on the probabilities, backward, AdamW -- through the
    CPU oracle (torch CPU fp32 autograd on oracle/wavenet_oracle.py) on the host cores, at B = 2 (memory), T = 16000."""
    from oracle import wavenet_oracle as O
    from movenet_amd.utils.weights import one_hot, synthetic_indices
    import torch.nn.functional as F
    cores = host_cores()
    torch.set_num_threads(cores)
    dims = O.Dims(**CFG)
    rf = dims.receptive_fields
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if not k.startswith("video_") and "context" not in k}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-4)
    audio = one_hot(synthetic_indices(batch, t_len, CFG["input_channels"], 1234), CFG["input_channels"])
    target = audio[:, :, rf:].argmax(1)
    t0 = None
    for i in range(n_warm + n_timed):
        if i == n_warm:
            t0 = time.perf_counter()
        log(f"cpu_train_baseline: step {i + 1}/{n_warm + n_timed}")
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(O.forward(params, dims, audio), target)
        loss.backward()
        opt.step()
    dt = (time.perf_counter() - t0) / n_timed
    tokens = batch * (t_len - rf)
    return {"value": tokens / dt, "unit": "tokens/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{n_timed} steps after {n_warm} warm-up of the reference's train step (oracle forward, cross_entropy on "
                      f"probabilities, autograd backward, AdamW) at B={batch}, T={t_len} ({tokens} tokens per step), torch CPU "
                      f"fp32, {dt:.2f} s per step"}


def dilations(cfg):
    return [1 << i for _ in range(cfg["stack_size"]) for i in range(cfg["layer_size"])]


def train_flop_per_step(cfg, batch: int, t_len: int, frames: int = 0) -> float:
    """forward MACs per sequence = sum_l 5C^2 T_l + L C K S + (KQ + Q^2) S
    (skip only where kept, causal conv as a gather), T_l = T - sum_{j<=l} d_j; with conditioning
    (frames > 0) + 2C^2 T_l per layer, the Conv3d encoder (F x 4096 x C) and the three stride-10
    transposed convs (C^2 per output sample: 10F + 100F + 1000F of them).  Step FLOP =
    2 x 3 x batch x that (forward + both gradients)."""
    C, K, Q = cfg["residual_channels"], cfg["skip_channels"], cfg["input_channels"]
    ds = dilations(cfg)
    rf = sum(ds) + cfg["stack_size"]
    S = t_len - rf + 1
    per_pos = (7 if frames else 5) * C * C
    macs, a = 0, 0
    for d in ds:
        a += d
        macs += per_pos * (t_len - a)
    macs += len(ds) * C * K * S + (K * Q + Q * Q) * S
    if frames:
        macs += frames * 4096 * C + C * C * 1110 * frames
    return 2.0 * 3.0 * batch * macs


def train_bytes_per_step(cfg, batch: int, t_len: int, frames: int = 0) -> dict:
    """HBM bytes one training step MUST move with this build's kernels ("design": every tensor a kernel reads or
    writes counted once per kernel, fp32) and with nothing saved or staged at all ("floor": a layer reads its input and
    writes its output, forward and backward, plus the skip sum).  Per layer and position, in floats (C = K = 64; DESIGN
    section 5 has the derivation):
      forward   x(t) 64 + x(t-d) 64 read; x' 64, tanh 64, sigmoid 64 written; skip sum 64 read + 64 written where t >= RF-1
                [conditioned: + context 64 read]                                                     -> 448 [512]
      backward  A' 64, P0 64 (the layer above's input gradient in scatter form), dskip 64 (t >= RF-1), tanh 64, sigmoid 64,
                x(t) 64, x(t-d) 64 read; A' 64, P0 64 written                                          -> 576
                [conditioned: + df|dg 128 written, and the context pass: df|dg 128 + context 64 + dctx 64 read, dctx 64 written]
      floor     forward x 64 read, x' 64 written, skip 128; backward dx' 64, x 64 read, dx 64 written, dskip 64
    Head per output position: 1344 forward (skip, a1 written and read, logits written, softmax + loss in place) + 2496
    backward; embedding 64 forward, 256 backward (combine + gradient).  r3's two-half backward moved 896 per layer
    position where this round's one-kernel form moves 576."""
    C, Q = cfg["residual_channels"], cfg["input_channels"]
    ds = dilations(cfg)
    L, rf = len(ds), sum(ds) + cfg["stack_size"]
    S = t_len - rf + 1
    design = floor = 0
    a = 0
    for li, d in enumerate(ds):
        a += d
        n, ns = t_len - a, min(t_len - a, S)          # positions of the layer, positions that also feed the skip sum
        last, first = li == L - 1, li == 0
        fwd = n * (2 * C + (0 if last else C) + 2 * C + (C if frames else 0)) + ns * (C + (0 if first else C))
        bwd = n * ((0 if last else 2 * C) + 4 * C + 2 * C) + ns * C
        if frames:
            bwd += n * (2 * C + 2 * C + C + 2 * C)
        design += fwd + bwd
        floor += n * (C + (0 if last else C)) + ns * 2 * C + n * ((0 if last else C) + C + C) + ns * C
    head = S * (1344 + 2496) * (Q / 256.0)
    embed = t_len * (C + 4 * C)
    video = frames * 4096 + 3 * 3 * C * int(1.11 * t_len) if frames else 0
    design += head + embed + video
    floor += S * (C + Q) + S * (Q + C) + t_len * 2 * C
    return {"design_bytes": 4.0 * batch * design, "floor_bytes": 4.0 * batch * floor}


class clip_frames:
    """the reference fixes the clip length through module constants (160 frames <->
    160000 samples, wavenet.py:27-31).  BASELINE configs[2]/[3] use 32-frame clips: the
    constants are set for the duration of the leg and restored."""

    def __init__(self, frames):
        self.frames = frames

    def __enter__(self):
        import movenet_amd.wavenet as W
        self.saved = (W.MAX_VIDEO_FRAMES, W.MAX_AUDIO_FRAMES)
        W.MAX_VIDEO_FRAMES, W.MAX_AUDIO_FRAMES = self.frames, 1000 * self.frames

    def __exit__(self, *exc):
        import movenet_amd.wavenet as W
        W.MAX_VIDEO_FRAMES, W.MAX_AUDIO_FRAMES = self.saved


TRAIN_WORKLOADS = {
    # BASELINE configs[1]: audio only, 16 clips of 1 s per GPU
    2: dict(batch=16, t_len=16000, frames=0,
            name="BASELINE configs[1]: 30-layer WaveNet, Q=256, C=K=64, audio only, 16 clips x 16000 samples per GPU"),
    # BASELINE configs[2] per GPU == configs[3] (batch 64 over 8 GPUs): video-conditioned, 8 clips of 32 frames
    3: dict(batch=8, t_len=32000, frames=32,
            name="BASELINE configs[2] (= configs[3] per GPU): config 2 + video conditioning, 32-frame clips "
                 "(T=32000), 8 clips per GPU"),
}


def train_leg(dev, world, rank, steps=6, warmup=5, config=2):
    """Secondary metric M2 (BASELINE.json): train-step tokens/sec -- the trainer's own step
    (movenet_amd/pytorch_lightning_trainer.py): forward + cross_entropy on the probabilities (Q2)
    as one fused node, backward through the HIP kernels, ONE flat gradient all-reduce when
    world > 1, FlatAdamW.  config 2 = BASELINE configs[1] (audio only); config 3 = configs[2] /
    configs[3] per GPU (video frames -> encoder -> up-sampler -> conditioned layers, their
    gradients included).  Token = one (sequence, time) position with a target: B * (T - RF)."""
    import numpy as np
    import torch.distributed as dist
    from movenet_amd.optim import FlatAdamW, order_like_backward
    from movenet_amd.parallel import FlatGradSync
    from movenet_amd.utils.weights import make_state_dict, one_hot, synthetic_indices
    from movenet_amd.wavenet import WaveNet
    wl = TRAIN_WORKLOADS[config]
    batch, t_len, frames = wl["batch"], wl["t_len"], wl["frames"]
    with clip_frames(frames or 160):
        model = WaveNet(**CFG)
        model.load_state_dict(make_state_dict(**CFG, seed=0), strict=not frames)
        model.to(dev).train()
        # torch.optim.AdamW's rule, one kernel over one flat buffer
        opt = FlatAdamW(order_like_backward(model, with_context=bool(frames)), lr=1e-4)
        sync = FlatGradSync(model.parameters(), world)
        sync.broadcast_parameters(0)
        Q, rf = CFG["input_channels"], model.receptive_fields
        audio = one_hot(synthetic_indices(batch, t_len, Q, 1234 + rank).to(dev), Q)
        target = audio[:, :, rf:].argmax(1)
        video = None
        if frames:  # U[0,1) frames, seed 4321 + rank
            video = torch.from_numpy(np.random.default_rng(4321 + rank).random(
                (batch, frames, 64, 64, 1), dtype=np.float32)).to(dev)

        def step():
            opt.zero_grad(set_to_none=True)
            loss, _, _ = model(audio, video, return_loss=True, target=target)  # the trainer's fused step
            loss.backward()
            sync.sync_gradients()
            opt.step()
            return loss

        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        marks[0].record()
        for i in range(steps):
            loss = step()
            marks[i + 1].record()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    log(f"rank {rank}: config-{config} train steps (ms) {[round(x, 1) for x in step_ms]}")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # replica consistency evidence (used by the two-process -m gpu test): after identical
    # optimizer steps on all-reduced gradients every rank must hold the same parameters
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    digest = hashlib.sha256(flat.tobytes()).hexdigest()
    digests = [digest]
    if world > 1:
        digests = [None] * world
        dist.all_gather_object(digests, digest)
    tokens_per_step = world * batch * (t_len - rf)
    flop_step = train_flop_per_step(CFG, batch, t_len, frames)  # per GPU
    # The leg's figure is the MEDIAN of the timed steps (rank 0's stream events around each step; the ranks run in step
    # through the all-reduce): one host hiccup -- a 133 ms step among five of 10.1 was seen once in this pool -- would
    # otherwise triple the mean of six.  The mean over the barrier-bracketed region (max over ranks) and every step's time
    # are printed beside it.
    med = sorted(step_ms)[len(step_ms) // 2] * 1e-3
    return {"metric": "train-step tokens/sec", "value": tokens_per_step / med, "unit": "tokens/s",
            "workload": wl["name"], "conditioned": bool(frames),
            "ms_per_step": med * 1e3, "timing": f"median of {steps} timed steps after {warmup} warm-ups",
            "ms_per_step_mean": dt / steps * 1e3, "tokens_per_s_mean": tokens_per_step * steps / dt,
            "step_ms_rank0": [round(x, 3) for x in step_ms],
            "global_batch": world * batch, "seq_len": t_len,
            "tokens_per_step": tokens_per_step, "optimizer": "AdamW (FlatAdamW, one launch)",
            "dtype": "f32 (tensors, accumulation and results; most products formed exactly from three bf16 planes per operand)",
            "loss": float(loss.detach()), "optimizer_launches_per_step": opt.last_launches,
            "param_sha256_per_rank": digests, "allreduce_path": sync.last_path,
            "allreduce_floats": sync.last_floats,
            "flop_per_step_per_gpu": flop_step, "flop_per_token": flop_step / (batch * (t_len - rf)),
            "roofline": train_roofline(flop_step, train_bytes_per_step(CFG, batch, t_len, frames), med,
                                       "train_config3" if frames else "train_config2")}


def train_roofline(flop_step: float, nbytes: dict, s_per_step: float, section: str) -> dict:
    """What bounds a training step, three ways ("report both"): the HBM roof over the bytes the step must
    move, the matrix cores priced against the unit the kernels actually issue, and the fp32-nominal figure of earlier
    rounds.  The bytes a profiler counted come from this round's --pmc passes (profiles/), labelled as such."""
    tf = flop_step / s_per_step / 1e12
    gbps = nbytes["design_bytes"] / s_per_step / 1e9
    bf3_peak = BF16_DENSE_PEAK_TFLOPS / 6.0
    prof = profiled("step_totals", section)
    out = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
           "algorithmic_bytes_per_step": nbytes["design_bytes"],
           "floor_bytes_per_step": nbytes["floor_bytes"],
           "floor_GBps": nbytes["floor_bytes"] / s_per_step / 1e9,
           "traffic": None,
           "traffic_profiled": ({"bytes_per_step": prof["bytes_per_step_fetch_x2_plus_write"],
                                 "GBps_at_this_runs_step_time": prof["bytes_per_step_fetch_x2_plus_write"] / s_per_step / 1e9,
                                 "frac_of_hbm_peak": prof["bytes_per_step_fetch_x2_plus_write"] / s_per_step / 1e9 / HBM_PEAK_GBPS,
                                 "source": f"profiles/{PROFILE_TAG}_pmc_summary.json step_totals.{section} (separate --pmc passes of "
                                           "scripts/train_steps.py: 2 x FETCH_SIZE + WRITE_SIZE summed over every kernel of a step); "
                                           "NOT measured by this run"} if prof else None),
           "mfma": {"unit_used": "bf16 x 3: six v_mfma_f32_32x32x16_bf16 per fp32 32x32x16 block (every product of the layers, "
                                 "the head and the weight gradients; the conditioned forward's x(t) / context blocks and the "
                                 "video up-sampler still issue v_mfma_f32_32x32x2_f32)",
                    "achieved": tf, "peak": bf3_peak, "unit": "TFLOP/s (fp32 model FLOP)", "frac": tf / bf3_peak,
                    "fp32_nominal": {"peak": FP32_PEAK_TFLOPS, "frac": tf / FP32_PEAK_TFLOPS,
                                     "note": "fp32 model FLOP over the fp32 MFMA peak: the figure of rounds 1-3, NOT a fraction of "
                                             "a ceiling these kernels run against (an fp32-equivalent rate)"}},
           "note": "the HBM roof binds on paper (bytes / 8 TB/s > FLOP x 6 / 2.5 PF); the layer kernels themselves are bound by "
                   "vector-instruction issue -- the exact split of every operand into three bf16 planes (DESIGN 4.3c)"}
    return out


def trainer_fit_line(dev, steps=16):
    """The SAME config-2 step driven by the trainer entry point's own loop (Trainer.fit: loader,
    training_step, gradient-norm tracking as the reference's track_grad_norm=2, optimizer,
    OneCycleLR, logging): two epochs of `steps` batches, the second one timed (the first holds
    code-object loading and allocator growth).  (16 steps: an epoch's fixed cost -- the loader's first batch
    with the GPU idle, the closing synchronise -- is ~5 ms, 0.6 ms per step over 8 steps of 11.5.)"""
    import contextlib
    from movenet_amd.config import ModelConfig, TrainingConfig
    from movenet_amd.pytorch_lightning_trainer import Dance2Music, Trainer
    wl = TRAIN_WORKLOADS[2]
    cfg = TrainingConfig(model_config=ModelConfig(**CFG), batch_size=wl["batch"], val_batch_size=wl["batch"],
                         use_video=False, n_epochs=2, optimizer="AdamW", scheduler="OneCycleLR")
    spec = f"synthetic://clips={wl['batch'] * steps},frames={wl['t_len']},seed=1234"
    with contextlib.redirect_stdout(sys.stderr):  # the loop prints its records; stdout is the JSON line's
        torch.manual_seed(0)
        module = Dance2Music(spec, cfg)
        trainer = Trainer(max_epochs=2, default_root_dir=None, gradient_clip_val=0.0, log_every_n_steps=1,
                          track_grad_norm=2, limit_train_batches=steps, device=str(dev),
                          enable_checkpointing=False, limit_val_batches=1)
        trainer.fit(module)
    ms = trainer.epoch_seconds[-1] / steps * 1e3
    return {"what": "Trainer.fit, config 2, second of two epochs", "steps": steps, "ms_per_step": ms,
            "tokens_per_s": wl["batch"] * (wl["t_len"] - 3072) / (ms / 1e3),
            "track_grad_norm": 2, "last_record": trainer.history[-1]}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start one child per
    device (same command line, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), relay rank 0's JSON
    line.  Runs before this process has made any GPU call; the children are ordinary child
    processes (no exec of an initialised process)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


CFG5 = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)
FP16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 matrix peak (no sparsity)


def timed_advance(gen, dev, steps, warm):
    """(seconds, HIP-event ms) of one ``advance(steps)`` launch after ``warm`` untimed steps."""
    if warm:
        gen.advance(warm)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    gen.advance(steps)
    e1.record()
    gen.check_errors()
    return time.perf_counter() - t0, e0.elapsed_time(e1)


def config3_generate_line(dev, rank, n_new=16000):
    """BASELINE configs[2]: the config-2 model + video conditioning, 8 clips of 32 frames
    (T = 32000), greedy generate of 16000 samples per clip through the generator (context
    convs folded into the past-tap precompute of the pipelined kernel)."""
    import numpy as np
    from movenet_amd.generation import RingGenerator
    from movenet_amd.utils.weights import make_state_dict, synthetic_indices
    from movenet_amd.wavenet import WaveNet
    B, F = 8, 32
    with clip_frames(F):
        model = WaveNet(**CFG)
        model.load_state_dict(make_state_dict(**CFG, seed=0), strict=False)
        model.to(dev).eval()
        video = torch.from_numpy(np.random.default_rng(4321 + rank).random((B, F, 64, 64, 1), dtype=np.float32)).to(dev)
        with torch.no_grad():
            context = model.upsample_video(video)
    rf = model.receptive_fields
    g = RingGenerator(**CFG, state_dict=model._decoder_state(), batch=B, n_total=rf + n_new + 1,
                      device=dev, temperature=0.0, seed=0, context=context)
    g.prime(synthetic_indices(B, rf, CFG["input_channels"], 1234 + rank).to(dev))
    # (no warm-up launch: the kernel is the headline's, already loaded, and every launch of it in this
    # process keeps the headline's step count -- see batch_sweep_lines; the clip's context covers
    # 32000 samples, one 16000-step launch behind the prompt fits it)
    dt, ms = timed_advance(g, dev, n_new, 0)
    C, K, Q, L = 64, 64, 256, 30
    flop = 2 * (L * (7 * C * C + C * K) + K * Q + Q * Q)  # 2,129,920 with conditioning
    tf = flop * B * n_new / (ms / 1e3) / 1e12
    return {"workload": "BASELINE configs[2]: config 2 + video conditioning, 32-frame clips, batch 8, greedy",
            "samples_per_s": B * n_new / dt, "us_per_sample_step": dt / n_new * 1e6, "launch_ms": ms,
            "kernel_variant": g.variant, "flop_per_sample": flop,
            "roofline": {"bound": "valu_fp32", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP32_PEAK_TFLOPS}}


def config5_lines(dev, rank, n_new=22050):
    """BASELINE configs[4]: 60-layer WaveNet, C = K = 128, 22.05 kHz, autoregressive generate of 1 s
    of audio (22050 samples), batch 1 -- in fp32 (the reference's precision) and with fp16
    operands / fp32 accumulation (the config as written)."""
    from movenet_amd import _native as N
    from movenet_amd.generation import RingGenerator
    from movenet_amd.utils.weights import make_state_dict, synthetic_indices
    sd5 = {k: v.to(dev) for k, v in make_state_dict(**CFG5, seed=0).items() if not k.startswith("video_")}
    rf = 6144
    flop = flop_per_sample(CFG5)  # 11,993,088
    out = {"workload": "BASELINE configs[4]: 60-layer (10x6) WaveNet, Q=256, C=K=128, 22.05 kHz, batch 1, "
                       "greedy generate of 22050 samples (1 s of audio)", "flop_per_sample": flop}
    for key, variant, peak, bound in (("fp32", N.GEN_PIPE, FP32_PEAK_TFLOPS, "valu_fp32"),
                                      ("fp16_operands_fp32_accumulate", N.GEN_PIPE_F16, FP16_DENSE_PEAK_TFLOPS,
                                       "mfma_fp16 (useful FLOP against the dense fp16 matrix peak; a mat-vec fills 1 of an MFMA's 16 columns)")):
        g = RingGenerator(**CFG5, state_dict=sd5, batch=1, n_total=rf + n_new + n_new // 10 + 1, device=dev,
                          variant=variant, temperature=0.0, seed=0)
        g.prime(synthetic_indices(1, rf, 256, 1234 + rank).to(dev))
        dt, ms = timed_advance(g, dev, n_new, n_new // 10)
        tf = flop * n_new / (ms / 1e3) / 1e12
        out[key] = {"us_per_sample_step": dt / n_new * 1e6, "seconds_per_second_of_audio": dt,
                    "real_time": dt <= 1.0, "samples_per_s": n_new / dt, "launch_ms": ms,
                    "kernel_variant": g.variant,
                    "roofline": {"bound": bound, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak}}
        del g
        # ... and 16 sequences for throughput: the pipelines serve them in turn within ONE launch
        B, n16 = 16, 4000
        g = RingGenerator(**CFG5, state_dict=sd5, batch=B, n_total=rf + n16 + n16 // 10 + 1, device=dev,
                          variant=variant, temperature=0.0, seed=0)
        g.prime(synthetic_indices(B, rf, 256, 4321 + rank).to(dev))
        dt, ms = timed_advance(g, dev, n16, n16 // 10)
        tf = flop * B * n16 / (ms / 1e3) / 1e12
        out[key]["batch_16"] = {"us_per_step_of_all_sequences": dt / n16 * 1e6, "samples_per_s": B * n16 / dt,
                                "roofline": {"bound": bound, "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                                             "frac": tf / peak}}
        del g
    return out


def batch_sweep_lines(dev, sd, rf, rank, n_new=4000):
    """Samples/s of the config-2 generator beyond the headline's 16 sequences per GPU (what
    MVN_GEN_AUTO picks for each batch; greedy).  (16 sequences IS the headline and is not run again
    here: every launch of its kernel in this process keeps the headline's step count, so that the
    kernel's average in a rocprofv3 --stats run of this command is the time the roofline quotes.)"""
    from movenet_amd import _native as N
    from movenet_amd.generation import GroupedGenerator, RingGenerator, auto_plan
    from movenet_amd.utils.weights import synthetic_indices
    out = {}
    dims = N.make_dims(*(CFG[k] for k in ("layer_size", "stack_size", "input_channels", "residual_channels",
                                          "skip_channels")))
    for B in (64, 128, 184):
        kind, group, variant = auto_plan(dims, B, False)
        kw = dict(state_dict=sd, batch=B, n_total=rf + n_new + n_new // 10 + 1, device=dev, variant=variant,
                  temperature=0.0, seed=0)
        g = GroupedGenerator(**CFG, group=group, **kw) if kind == "grouped" else RingGenerator(**CFG, **kw)
        g.prime(synthetic_indices(B, rf, CFG["input_channels"], 1234 + rank).to(dev))
        dt, ms = timed_advance(g, dev, n_new, n_new // 10)
        tf = flop_per_sample(CFG) * B * n_new / dt / 1e12
        out[str(B)] = {"samples_per_s": B * n_new / dt, "us_per_step_of_all_sequences": dt / n_new * 1e6,
                       "plan": [kind, group, variant],
                       "roofline": {"bound": "valu_fp32", "achieved": tf, "peak": FP32_PEAK_TFLOPS,
                                    "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS}}
        del g
    return out


def extra_lines(dev, sd, rf, n_new, rank):
    """Figures quoted beside the headline (outside the timed steps): sampling at the reference's
    default temperature 1.0, one END-TO-END WaveNet.generate call (one-hot prompt in HBM ->
    indices -> priming -> 16000 steps -> one-hot out), the generator beyond 16 sequences, and
    BASELINE configs[2] (conditioned) / configs[4] (60 layers, C = 128, fp32 and fp16) generate."""
    from movenet_amd.generation import RingGenerator
    from movenet_amd.utils.weights import make_state_dict, one_hot, synthetic_indices
    from movenet_amd.wavenet import WaveNet
    prompt = synthetic_indices(BATCH, rf, CFG["input_channels"], 1234 + rank).to(dev)
    out = {}
    g = RingGenerator(**CFG, state_dict=sd, batch=BATCH, n_total=rf + 2 * n_new + 1, device=dev,
                      temperature=1.0, seed=1)
    g.prime(prompt)
    g.advance(n_new)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    g.advance(n_new)
    g.check_errors()
    out["temperature_1.0_samples_per_s"] = BATCH * n_new / (time.perf_counter() - t0)
    del g
    model = WaveNet(**CFG)
    model.load_state_dict(make_state_dict(**CFG, seed=0))
    model.to(dev)
    audio = one_hot(prompt, CFG["input_channels"])
    # (warm-up at FULL length: every launch of the generator kernel in this process then has the
    # same step count, so the kernel's average in a rocprofv3 --stats run of this command is the
    # per-launch time the roofline line quotes)
    model.generate(audio, n_samples=rf + n_new, temperature=0.0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    y = model.generate(audio, n_samples=rf + n_new, temperature=0.0)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert y.shape == (BATCH, CFG["input_channels"], rf + n_new)
    out["end_to_end_generate_samples_per_s"] = BATCH * n_new / dt
    out["end_to_end_generate_ms"] = dt * 1e3
    out["end_to_end_fallback_variant"] = model.last_generate_fallback
    del model, audio, y
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return out  # (single-GPU figures: the other ranks of a multi-GPU run should not wait for them)
    for key, fn in (("batch_sweep", lambda: batch_sweep_lines(dev, sd, rf, rank)),
                    ("config3_generate", lambda: config3_generate_line(dev, rank)),
                    ("config5", lambda: config5_lines(dev, rank))):
        try:
            out[key] = fn()
        except Exception as e:  # one missing figure must not take the others with it
            out[key] = {"error": f"{type(e).__name__}: {e}"}
        log(f"rank {rank}: extras.{key} {out[key]}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--new-samples", type=int, default=16000, help="samples per sequence per step")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 generic, 2 stream, 3 pipe, 5 fold")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the T=1.0 and end-to-end lines")
    args = ap.parse_args()

    if os.environ.get("MOVENET_BENCH_DUMP_MAPS") == "1":
        # diagnostics: the executable mappings of the process, to resolve a native stack trace
        import atexit

        def _dump_maps():
            with open("/proc/self/maps") as f:
                sys.stderr.write("".join(l for l in f if " r-xp " in l))
        atexit.register(_dump_maps)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            raise SystemExit(spawn_ranks(args.gpus))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    from movenet_amd.utils.host import cap_torch_threads
    cap_torch_threads()  # (torch sizes its OpenMP pool by the visible cores, not by the container's quota)
    # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on cuda:0 and a
    # gloo process group, to exercise the N>1 control flow without a second GPU
    if os.environ.get("MOVENET_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MOVENET_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from movenet_amd.generation import RingGenerator
    from movenet_amd.utils.weights import make_state_dict, synthetic_indices

    sd_cpu = make_state_dict(**CFG, seed=0)
    sd = {k: v.to(dev) for k, v in sd_cpu.items() if not k.startswith("video_")}
    rf = 3072
    n_new, K, W = args.new_samples, args.steps, args.warmup
    n_total = rf + (K + W) * n_new + 1
    gen = RingGenerator(**CFG, state_dict=sd, batch=BATCH, n_total=n_total, device=dev,
                        variant=args.variant, temperature=0.0, seed=0)
    prompt = synthetic_indices(BATCH, rf, CFG["input_channels"], 1234 + rank).to(dev)

    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    gen.prime(prompt)  # first call in the process: code-object loading and allocator growth included
    torch.cuda.synchronize(dev)
    prime_cold_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    gen.prime(prompt)
    torch.cuda.synchronize(dev)
    prime_ms = (time.perf_counter() - t0) * 1e3
    log(f"rank {rank}: primed {rf} samples in {prime_ms:.1f} ms (first call {prime_cold_ms:.1f} ms)")

    for _ in range(W):
        gen.advance(n_new)
        torch.cuda.synchronize(dev)
        log(f"rank {rank}: warm-up step done")

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    fence()
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record()   # torch's current stream == the stream mvn_generate is launched on
        gen.advance(n_new)
        ev[k][1].record()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    log(f"rank {rank}: {K} timed steps in {elapsed:.3f} s; launch ms {kernel_ms}")

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    assert gen.t == rf - 1 + (K + W) * n_new
    gen.check_errors()
    variant_used = gen.variant
    total_samples = world * BATCH * n_new * K
    value = total_samples / elapsed
    avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    flops_per_launch = flop_per_sample(CFG) * BATCH * n_new
    achieved = flops_per_launch / avg_kernel_s / 1e12

    # (the train leg runs BEFORE the extra generator figures: after their 300 MB one-hot tensors
    # have come and gone, the same training step takes 21-24 ms instead of 17.5 on the same box --
    # every HBM-bound kernel slower, the slab reduces 3x -- with or without empty_cache(); device
    # memory handed back fragmented is our reading, not verified)
    train, train3, fit = None, None, None
    if not args.no_train_leg:
        del gen
        torch.cuda.empty_cache()
        # both M2 workloads at every N: config 2 (audio only, BASELINE configs[1]) and config 3
        # (video-conditioned, 8 clips of 32 frames per GPU = BASELINE configs[2]; at N = 8 that IS
        # configs[3], batch 64) -- so each has its own 1/2/4/8 curve on ONE workload
        for cfgn in (2, 3):
            try:
                line = train_leg(dev, world, rank, config=cfgn)
                log(f"rank {rank}: config-{cfgn} train leg {line['value']:.0f} tokens/s")
            except Exception as e:  # the headline metric must survive a failure here
                if world > 1:
                    raise  # ... but ranks must not diverge around collectives
                line = {"error": f"{type(e).__name__}: {e}"}
            if cfgn == 2:
                train = line
            else:
                train3 = line
            torch.cuda.empty_cache()
        if world == 1:
            try:
                fit = trainer_fit_line(dev)
                if train and "ms_per_step" in train:
                    fit["vs_train_step"] = fit["ms_per_step"] / train.get("ms_per_step_mean", train["ms_per_step"])  # mean against mean
                log(f"rank 0: Trainer.fit {fit['ms_per_step']:.2f} ms per step")
            except Exception as e:
                fit = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

    extras = None
    if rank == 0 and not args.no_extras:
        try:
            extras = extra_lines(dev, sd, rf, n_new, rank)
            log(f"rank 0: extras {extras}")
        except Exception as e:
            extras = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        out = {
            "metric": "autoregressive audio samples/sec (16 kHz, 30-layer WaveNet, whole job)",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: 30-layer (10x3) WaveNet, Q=256, C=K=64, 16 kHz, "
                            "batch 16 sequences per GPU, greedy autoregressive generate",
                "batch_per_gpu": BATCH,
                "new_samples_per_sequence_per_step": n_new,
                "prompt": rf,
                "kernel_variant": {1: "generic", 2: "stream64", 3: "pipe", 5: "fold"}[variant_used],
                "parallelism": f"independent clips x{world} (no collective)",
            },
            "samples_per_s_per_gpu": value / world,
            "samples_per_s_per_sequence": value / world / BATCH,
            "us_per_sample_step": elapsed / K / n_new * 1e6,
            "prime_ms": prime_ms,
            "prime_first_call_ms": prime_cold_ms,
            "roofline": {
                # the kernel issues v_pk_fma_f32 (fp32 VECTOR FMA), no MFMA: the roof it is
                # priced against is the fp32 vector peak, numerically the same 157.3 TFLOP/s
                # as the fp32-matrix peak SURVEY 8(d) names as the paper bound
                "bound": "valu_fp32",
                "achieved": achieved,
                "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP32_PEAK_TFLOPS,
                # HBM-side bytes: not measurable from inside the run (hardware counters need a
                # rocprofv3 --pmc pass) => null here; this round's profiled figure is quoted beside it,
                # labelled with the file it comes from (FETCH_SIZE as reported + WRITE_SIZE of separate passes)
                "traffic": None,
                "traffic_profiled": headline_traffic(variant_used, n_new),
                "algorithmic_bytes_per_launch": 2 * 30 * 64 * 4 * BATCH * n_new,  # 15,360 B per sample
                "kernel": {1: "gen_generic_kernel", 2: "gen_stream64_kernel", 3: "gen_pipe_kernel<64>",
                           5: "gen_fold_kernel"}[variant_used],
                "flop_per_launch": flops_per_launch,
                "avg_launch_ms": avg_kernel_s * 1e3,
                "note": "latency-bound: L-deep dependent chain per sample at batch 16 (DESIGN.md)",
                **latency_floor(variant_used, elapsed / K / n_new * 1e6),
            },
        }
        if extras and isinstance(extras.get("batch_sweep"), dict) and "error" not in extras["batch_sweep"]:
            # 16 sequences IS the headline (not launched a second time: see batch_sweep_lines)
            extras["batch_sweep"] = {"16": {"samples_per_s": value / world, "us_per_step_of_all_sequences":
                                            elapsed / K / n_new * 1e6, "plan": ["single", 0, variant_used],
                                            "roofline": {"bound": "valu_fp32", "achieved": achieved,
                                                         "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                         "frac": achieved / FP32_PEAK_TFLOPS},
                                            "note": "the headline's own timed steps"},
                                     **extras["batch_sweep"]}
        out["train_step"] = train
        out["train_step_config3"] = train3
        out["trainer_fit"] = fit
        out["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(sd_cpu)
            out["cpu_baseline"] = cb
            cached = cpu_cached_line(sd_cpu)
            out["cpu_cached_algorithm"] = cached
            # naive CPU -> cached CPU = the algorithmic change (Q4); cached CPU -> this = hardware
            out["speedup_vs_cpu_baseline"] = value / cb["value"]
            if isinstance(out.get("train_step"), dict):
                tb = cpu_train_baseline(sd_cpu)
                out["train_step"]["cpu_baseline"] = tb
                out["train_step"]["speedup_vs_cpu_baseline"] = out["train_step"]["value"] / tb["value"]
            out["speedup_split"] = {"algorithm_cached_over_windowed_cpu": cached["value"] / cb["value"],
                                    "hardware_gpu_over_cached_cpu": value / cached["value"]}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
