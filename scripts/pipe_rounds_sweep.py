"""Step time of the fp32 PIPE generator with several sequences per pipeline (gen_pipe_kernel<C, true>):
    python scripts/pipe_rounds_sweep.py [--c64] [n_new]
Default: BASELINE config 5's model (60 layers, C = K = 128: 4 pipelines of 61 stages, up to 16 sequences
each); --c64: config 2's (24 pipelines of 9 stages, up to 8 each).  Prints us per step of ALL sequences
and samples/s; the last line is the JSON of the table."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from movenet_amd import _native as N  # noqa: E402
from movenet_amd.generation import RingGenerator  # noqa: E402
from movenet_amd.utils.weights import make_state_dict, synthetic_indices  # noqa: E402

dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
c64 = "--c64" in sys.argv
n_new = int(args[0]) if args else (4000 if c64 else 1500)
cfg = dict(bench.CFG) if c64 else dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128,
                                       skip_channels=128)
batches = (24, 25, 48, 72, 96, 144, 192) if c64 else (1, 4, 5, 8, 12, 16, 24, 32, 48, 64)
sd = {k: v.to(dev) for k, v in make_state_dict(**cfg, seed=0).items() if not k.startswith("video_")}
rf = sum(2 ** (l % cfg["layer_size"]) for l in range(cfg["layer_size"] * cfg["stack_size"])) + 2
out = {}
for B in batches:
    g = RingGenerator(**cfg, state_dict=sd, batch=B, n_total=rf + n_new + n_new // 10 + 1, device=dev,
                      variant=N.GEN_PIPE, temperature=0.0, seed=0)
    g.prime(synthetic_indices(B, rf, 256, 1234).to(dev))
    dt, ms = bench.timed_advance(g, dev, n_new, n_new // 10)
    g.check_errors()
    out[f"batch {B}"] = dict(us_per_step=round(dt / n_new * 1e6, 2), samples_per_s=round(B * n_new / dt))
    print(f"PIPE C={cfg['residual_channels']} batch {B:4d}: {dt / n_new * 1e6:7.2f} us per step of all, "
          f"{B * n_new / dt / 1e3:.1f} k samples/s", flush=True)
    del g
print(json.dumps(out))
