"""Every BASELINE.json config that fits one GPU, measured once (context for DESIGN.md section 5;
bench.py remains the contract line for configs[1]).  Usage: python scripts/bench_configs.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import movenet_amd.wavenet as W  # noqa: E402
from movenet_amd import _native as N  # noqa: E402
from movenet_amd.optim import FlatAdamW, order_like_backward  # noqa: E402
from movenet_amd.utils.weights import make_state_dict, one_hot, synthetic_indices  # noqa: E402

DEV = "cuda:0"


def sync_time(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def model_of(cfg, seed=0, frames=None):
    if frames is not None:  # Q8: the module constants fix the clip length (F frames <-> 1000 F samples)
        W.MAX_VIDEO_FRAMES, W.MAX_AUDIO_FRAMES = frames, 1000 * frames
    m = W.WaveNet(**cfg)
    m.load_state_dict(make_state_dict(**cfg, seed=seed), strict=False)
    return m.to(DEV)


def generate_rate(m, B, n_new, video=None, temperature=0.0):
    Q, rf = m.input_channels, m.receptive_fields
    prompt = one_hot(synthetic_indices(B, rf, Q, 1234).to(DEV), Q)
    dt = sync_time(lambda: m.generate(prompt, video, n_samples=rf + n_new, temperature=temperature), reps=2)
    return dict(batch=B, new_samples=n_new, seconds_end_to_end=round(dt, 4),
                samples_per_s=round(B * n_new / dt), us_per_step=round(dt / n_new * 1e6, 2))


def train_rate(m, B, T, video=None):
    Q, rf = m.input_channels, m.receptive_fields
    audio = one_hot(synthetic_indices(B, T, Q, 1234).to(DEV), Q)
    target = audio[:, :, rf:].argmax(1)
    # the trainer's own step: fused forward + loss node, FlatAdamW (one launch)
    opt = FlatAdamW(order_like_backward(m, with_context=video is not None), lr=1e-4)
    m.train()

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = m(audio, video, return_loss=True, target=target)
        loss.backward()
        opt.step()
    dt = sync_time(step, reps=3, warm=2)
    return dict(batch=B, t_len=T, ms_per_step=round(dt * 1e3, 2), tokens_per_s=round(B * (T - rf) / dt))


out = {}
c1 = dict(layer_size=2, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)
c2 = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
c5 = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)

m = model_of(c1)
out["config1 (L=4, Q=64, C=16, B=2, 8 kHz)"] = dict(generate=generate_rate(m, 2, 8000), train=train_rate(m, 2, 8000))
m = model_of(c2)
out["config2 (L=30, Q=256, C=64, B=16, 16 kHz)"] = dict(
    generate_greedy=generate_rate(m, 16, 16000), generate_T1=generate_rate(m, 16, 16000, temperature=1.0),
    train=train_rate(m, 16, 16000))
m = model_of(c2, frames=32)
video = torch.from_numpy(np.random.default_rng(4321).random((8, 32, 64, 64, 1), dtype=np.float32)).to(DEV)
out["config3 (config 2 + video, F=32 -> T=32000, B=8)"] = dict(
    generate_greedy=generate_rate(m, 8, 16000, video=video), train=train_rate(m, 8, 32000, video=video))
m = model_of(c5)
out["config5 (L=60, C=128, 22.05 kHz, fp32)"] = dict(generate_b1=generate_rate(m, 1, 22050), generate_b4=generate_rate(m, 4, 22050))
# the reference's OWN experiment shapes (Q = 128: experiments/03_kinetics_scale_up.mk:7-10, :64-67,
# 04_kinetics_receptive_field.mk:8-11); r4: Q in {64, 128} takes the pipelined generators and the head's strip kernels
W.MAX_VIDEO_FRAMES, W.MAX_AUDIO_FRAMES = 160, 160000
for name, cfg, B, n_new, T in (
        ("reference shape (L=30, Q=128, C=64, B=16)", dict(layer_size=10, stack_size=3, input_channels=128, residual_channels=64, skip_channels=64), 16, 16000, 16000),
        ("reference shape (L=14, Q=128, C=16, layer_size=14: RF 16384, B=4)", dict(layer_size=14, stack_size=1, input_channels=128, residual_channels=16, skip_channels=16), 4, 2000, 32000),
        ("reference shape (L=4, Q=128, C=32, B=8)", dict(layer_size=2, stack_size=2, input_channels=128, residual_channels=32, skip_channels=32), 8, 8000, 16000)):
    m = model_of(cfg)
    variant = N.lib().mvn_gen_variant(m._dims, N.GEN_AUTO, B)
    out[name] = dict(generate_greedy=dict(generate_rate(m, B, n_new), kernel_variant=int(variant)), train=train_rate(m, B, T))
print(json.dumps(out, indent=1))
