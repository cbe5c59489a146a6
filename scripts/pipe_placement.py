"""Which XCD did each PIPE workgroup land on?  Diagnostic: reads the placement words the
kernel's handshake leaves in the hand-off area (XCC id + 1 per (sequence, stage))."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movenet_amd import _native as N  # noqa: E402
from movenet_amd.generation import RingGenerator  # noqa: E402
from movenet_amd.utils.weights import make_state_dict, synthetic_indices  # noqa: E402

dev = "cuda:0"
for cfg, B, rf in ((dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64), 24, 3072),
                   (dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128), 4, 6144)):
    sd = {k: v.to(dev) for k, v in make_state_dict(**cfg, seed=0).items()}
    g = RingGenerator(**cfg, state_dict=sd, batch=B, n_total=rf + 300, device=dev, variant=N.GEN_PIPE)
    g.prime(synthetic_indices(B, rf, 256, 1).to(dev))
    g.advance(200)
    g.check_errors()
    off = N.lib().mvn_gen_status_offset(g.dims, B)
    ns = (g.n_layers + 3) // 4 + 1 if cfg["residual_channels"] == 64 else g.n_layers + 1
    x = g.state[off + 16: off + 16 + B * ns].view(torch.int32).cpu().view(B, ns) - 1
    print(f"C={cfg['residual_channels']} batch {B}, {ns} stages: XCC id per (sequence, stage)")
    for b in range(B):
        row = x[b].tolist()
        edges = sum(row[s] == row[(s + 1) % ns] for s in range(ns))
        print(f"  seq {b:2d}: {''.join(str(v) for v in row)}   same-XCD edges {edges}/{ns}")
