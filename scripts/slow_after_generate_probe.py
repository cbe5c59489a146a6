"""'the same training step takes 21-24 ms instead of 17.5 after the bench's end-to-end
generate figures have come and gone in the process'.  Is it still so?  Config-2 train leg, then one
end-to-end WaveNet.generate of the bench's size (its 312 MB one-hot output and 50 MB prompt come and
go), then the train leg again -- same process, per-step times on stderr.
    python scripts/slow_after_generate_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from movenet_amd.utils.weights import make_state_dict, one_hot, synthetic_indices  # noqa: E402
from movenet_amd.wavenet import WaveNet  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
out = {}
out["before"] = bench.train_leg(dev, 1, 0, steps=6, warmup=5, config=2)["ms_per_step"]
model = WaveNet(**bench.CFG)
model.load_state_dict(make_state_dict(**bench.CFG, seed=0))
model.to(dev)
audio = one_hot(synthetic_indices(16, 3072, 256, 1234).to(dev), 256)
for _ in range(2):
    y = model.generate(audio, n_samples=3072 + 16000, temperature=0.0)
torch.cuda.synchronize()
del y, audio, model
out["after_generate"] = bench.train_leg(dev, 1, 0, steps=6, warmup=5, config=2)["ms_per_step"]
torch.cuda.empty_cache()
out["after_generate_and_empty_cache"] = bench.train_leg(dev, 1, 0, steps=6, warmup=5, config=2)["ms_per_step"]
print(json.dumps(out))
