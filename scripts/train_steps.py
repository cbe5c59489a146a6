"""A few training steps and nothing else (profiling target):
    rocprofv3 --kernel-trace --stats -d out -- python3 scripts/train_steps.py [steps] [--config 2|3]
config 2 = BASELINE configs[1] (audio only, 16 x 16000); config 3 = configs[2] (video-conditioned,
8 clips of 32 frames, T = 32000).  The step is bench.py's train leg = the trainer's step."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = [a for a in sys.argv[1:]]
config = 2
if "--config" in args:
    i = args.index("--config")
    config = int(args[i + 1])
    del args[i:i + 2]
steps = int(args[0]) if args else 2
print(json.dumps(bench.train_leg(torch.device("cuda:0"), 1, 0, steps=steps, warmup=1, config=config)))
