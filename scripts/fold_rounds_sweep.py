"""Step time of the config-2 generator with several sequences per pipeline (gen_fold_kernel<true>):
    python scripts/fold_rounds_sweep.py [n_new]
Prints us per step of ALL sequences and samples/s for batch 16 .. 184 (FOLD: 16 pipelines up to 80 sequences, 23
beyond), and STREAM at 128."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from movenet_amd import _native as N  # noqa: E402
from movenet_amd.generation import RingGenerator  # noqa: E402
from movenet_amd.utils.weights import make_state_dict, synthetic_indices  # noqa: E402

dev = torch.device("cuda:0")
n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sd = {k: v.to(dev) for k, v in make_state_dict(**bench.CFG, seed=0).items() if not k.startswith("video_")}
rf = 3072
out = {}
for variant, batches in ((N.GEN_FOLD, (16, 17, 24, 32, 48, 64, 80, 81, 96, 112, 128, 138, 161, 184)), (N.GEN_PIPE, (24,)), (N.GEN_STREAM, (64, 128))):
    for B in batches:
        g = RingGenerator(**bench.CFG, state_dict=sd, batch=B, n_total=rf + n_new + n_new // 10 + 1, device=dev,
                          variant=variant, temperature=0.0, seed=0)
        g.prime(synthetic_indices(B, rf, 256, 1234).to(dev))
        dt, ms = bench.timed_advance(g, dev, n_new, n_new // 10)
        out[f"variant {g.variant} batch {B}"] = dict(us_per_step=round(dt / n_new * 1e6, 2), samples_per_s=round(B * n_new / dt))
        print(f"variant {g.variant} batch {B:4d}: {dt / n_new * 1e6:7.2f} us per step of all, {B * n_new / dt / 1e6:.3f} M samples/s",
              flush=True)
        del g
print(json.dumps(out))
