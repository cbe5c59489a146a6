import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
for b in (16, 8, 4, 2):
    r = bench.train_leg(torch.device("cuda:0"), 1, 0, steps=3, warmup=1, batch=b)
    print(b, round(r["ms_per_step"],2), "ms/step", round(r["value"]/1e6,2), "Mtok/s", round(r["roofline"]["frac"],3), flush=True)
