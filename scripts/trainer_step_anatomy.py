"""Host and GPU time of each part of a Trainer.fit-style step (config 2), without cProfile:
    python scripts/trainer_step_anatomy.py
For every step: host ms spent in loader / training_step / backward / norm + optimizer, and the GPU
time between events recorded at the same points."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from movenet_amd.config import ModelConfig, TrainingConfig  # noqa: E402
from movenet_amd.parallel import contiguous_grad_span  # noqa: E402
from movenet_amd.pytorch_lightning_trainer import Dance2Music  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if "--no-thread-cap" not in sys.argv:  # (as Trainer.fit does; without it: the r3 stall hunt's setting)
    from movenet_amd.utils.host import cap_torch_threads
    print("torch intra-op threads:", cap_torch_threads())

# ---- fine timers inside training_step: wrap the callables it goes through
import movenet_amd.ops as ops_mod  # noqa: E402
import movenet_amd.wavenet as wn_mod  # noqa: E402
FINE = {}


def timed(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            FINE[label] = FINE.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
    setattr(owner, name, wrapper)


timed(wn_mod.WaveNet, "_indices_async")
timed(wn_mod.WaveNet, "_all_one_hot")
timed(ops_mod, "run_forward")
timed(ops_mod.ForwardBuffers, "__init__", "ForwardBuffers")
timed(ops_mod, "pack_params")
timed(ops_mod, "_decoder_params")
timed(torch, "empty", "torch.empty")
timed(torch, "zeros", "torch.zeros")
cfg = TrainingConfig(model_config=ModelConfig(**bench.CFG), batch_size=16, val_batch_size=16, use_video=False,
                     n_epochs=1, optimizer="AdamW", scheduler=None)
m = Dance2Music("synthetic://clips=192,frames=16000,seed=1234", cfg).to(dev)
opt = m.configure_optimizers()["optimizer"]
m.train()
names = ("loader", "training_step", "backward", "norm+opt")
rows = []
it = iter(m.train_dataloader())
for step in range(12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    t = [time.perf_counter()]
    ev[0].record()
    batch = next(it)
    t.append(time.perf_counter()); ev[1].record()
    FINE.clear()
    out = m.training_step(batch, step)
    fine = {k: round(v, 2) for k, v in FINE.items() if v >= 0.5}
    t.append(time.perf_counter()); ev[2].record()
    out["loss"].backward()
    t.append(time.perf_counter()); ev[3].record()
    used = [p for p in m.model.parameters() if p.grad is not None]
    gn = contiguous_grad_span(used).norm(2)
    opt.step()
    opt.zero_grad(set_to_none=True)
    t.append(time.perf_counter()); ev[4].record()
    rows.append((t, ev, (torch.cuda.memory_stats(dev)["num_device_alloc"], fine)))
torch.cuda.synchronize()
for i, (t, ev, nalloc) in enumerate(rows):
    host = [round((t[k + 1] - t[k]) * 1e3, 2) for k in range(4)]
    gpu = [round(ev[k].elapsed_time(ev[k + 1]), 2) for k in range(4)]
    print(f"step {i:2d} host ms {dict(zip(names, host))} total {sum(host):.2f} | gpu ms {dict(zip(names, gpu))} total {sum(gpu):.2f} | device allocs so far {nalloc}")
wall = (rows[-1][0][-1] - rows[4][0][0]) / 8 * 1e3
print(f"wall per step over the last 8: {wall:.2f} ms")
