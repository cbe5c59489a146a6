"""BASELINE configs[4]: 60-layer WaveNet, C = K = 128, 22.05 kHz, autoregressive generate of 1 s
of audio on 1 x MI355X -- fp32 (the reference's precision, default) and fp16 operands / fp32
accumulation (as the config is written) side by side.  Usage: python scripts/bench_config5.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movenet_amd import _native as N  # noqa: E402
from movenet_amd.generation import RingGenerator  # noqa: E402
from movenet_amd.utils.weights import make_state_dict, synthetic_indices  # noqa: E402

DEV = "cuda:0"
cfg = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)
sd = {k: v.to(DEV) for k, v in make_state_dict(**cfg, seed=0).items() if not k.startswith("video_")}
rf, n_new = 6144, 22050
flop_per_sample = 2 * (60 * (5 * 128 * 128 + 128 * 128) + 128 * 256 + 256 * 256)  # 11,993,088
out = {}
FP16_ONLY = "--fp16-only" in sys.argv
for name, variant, batches in (("fp32 (gen_pipe_kernel<128>, 61 stages; beyond 4 sequences the 4 pipelines serve them in turn)", N.GEN_PIPE, (1, 4, 16, 64)),
                               ("fp16 operands / fp32 accumulate (gen_pipe_h16_kernel, 21 stages; 31 beyond 8 sequences)", N.GEN_PIPE_F16, (1, 4, 8, 16, 64)),
                               ("fp32 generic kernel", N.GEN_GENERIC, (1,))):
    for B in batches:
        if FP16_ONLY and (variant != N.GEN_PIPE_F16 or B != 1):
            continue
        g = RingGenerator(**cfg, state_dict=sd, batch=B, n_total=rf + 2 * n_new + 1, device=DEV, variant=variant)
        g.prime_with_forward = True  # queue priming is outside the timed region either way
        g.prime(synthetic_indices(B, rf, 256, 1234).to(DEV))
        steps = n_new if variant != N.GEN_GENERIC else 2000
        g.advance(steps // 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.advance(steps)
        g.check_errors()
        dt = time.perf_counter() - t0
        out[f"{name}, batch {B}"] = dict(us_per_step=round(dt / steps * 1e6, 2),
                                         seconds_per_1s_audio=round(dt / steps * n_new, 3),
                                         samples_per_s=round(B * steps / dt),
                                         tflops=round(B * steps * flop_per_sample / dt / 1e12, 3))
        del g
print(json.dumps(out, indent=1))
