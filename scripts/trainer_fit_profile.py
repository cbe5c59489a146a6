"""Where does a Trainer.fit step spend its host time?  cProfile of bench.trainer_fit_line (config 2,
two epochs of 8 steps) + per-step wall times.  Usage: python scripts/trainer_fit_profile.py"""
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pr = cProfile.Profile()
pr.enable()
line = bench.trainer_fit_line(dev)
pr.disable()
print(line, file=sys.stderr)
ms = torch.cuda.memory_stats(dev)
print({k: ms[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.peak",
                          "allocated_bytes.all.peak", "allocation.all.allocated")}, file=sys.stderr)
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue())
out = io.StringIO()
st = pstats.Stats(pr, stream=out).sort_stats("tottime")
st.print_stats(25)
st.print_callers("method 'to' of")
st.print_callers("pin_memory")
print(out.getvalue())
