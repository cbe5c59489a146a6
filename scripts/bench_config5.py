"""BASELINE config 5 (60-layer, C=K=128, 22.05 kHz): 1 s of audio, batch 1, through the
fp32 generators for this shape: the 61-stage PIPE kernel (batch <= 4) and the GENERIC one."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movenet_amd.generation import RingGenerator
from movenet_amd.utils.weights import make_state_dict, synthetic_indices

cfg = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)
dev = "cuda:0"
sd = {k: v.to(dev) for k, v in make_state_dict(**cfg, seed=0).items()}
rf, n_new = 6144, int(sys.argv[1]) if len(sys.argv) > 1 else 2205
from movenet_amd import _native as N
for B, variant in ((1, N.GEN_PIPE), (4, N.GEN_PIPE), (1, N.GEN_GENERIC), (16, N.GEN_GENERIC)):
    g = RingGenerator(**cfg, state_dict=sd, batch=B, n_total=rf + n_new + 1, device=dev, variant=variant)
    g.prime(synthetic_indices(B, rf, 256, 1).to(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.advance(n_new)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g.check_errors()
    print(f"config 5, variant {g.variant}, batch {B}: {n_new} samples/sequence in {dt:.3f} s = "
          f"{dt / n_new * 1e6:.1f} us/step, {B * n_new / dt:.0f} samples/s "
          f"(1 s of 22.05 kHz audio would take {dt / n_new * 22050:.2f} s)")
