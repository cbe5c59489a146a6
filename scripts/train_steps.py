"""A few config-2 training steps and nothing else (profiling target):
    rocprofv3 --kernel-trace --stats -d out -- python3 scripts/train_steps.py [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
print(json.dumps(bench.train_leg(torch.device("cuda:0"), 1, 0, steps=steps, warmup=1)))
