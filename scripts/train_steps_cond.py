"""A few CONFIG-3 training steps (video-conditioned, F=32 -> T=32000, B=8) and nothing else
(profiling target, like scripts/train_steps.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import movenet_amd.wavenet as W  # noqa: E402
from movenet_amd.ops import cross_entropy_on_probs  # noqa: E402
from movenet_amd.utils.weights import make_state_dict, one_hot, synthetic_indices  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = "cuda:0"
cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
W.MAX_VIDEO_FRAMES, W.MAX_AUDIO_FRAMES = 32, 32000
m = W.WaveNet(**cfg)
m.load_state_dict(make_state_dict(**cfg, seed=0), strict=False)
m.to(dev).train()
B, T, rf = 8, 32000, m.receptive_fields
audio = one_hot(synthetic_indices(B, T, 256, 1234).to(dev), 256)
video = torch.from_numpy(np.random.default_rng(4321).random((B, 32, 64, 64, 1), dtype=np.float32)).to(dev)
target = audio[:, :, rf:].argmax(1)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
for i in range(steps + 1):
    opt.zero_grad(set_to_none=True)
    loss, _ = cross_entropy_on_probs(m(audio, video), target)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done", float(loss))
