"""Host-side facts the training loop and the benchmark need: the CPU share of this process."""
from __future__ import annotations

import os


def cpu_share() -> int:
    """Threads this process can actually keep running: its affinity mask capped by the cgroup CPU
    quota (cgroup v2 ``cpu.max``, v1 ``cpu.cfs_quota_us``); ``MOVENET_CPU_THREADS`` overrides."""
    if os.environ.get("MOVENET_CPU_THREADS"):
        return max(1, int(os.environ["MOVENET_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cap_torch_threads() -> int:
    """Limit torch's intra-op CPU threads to the CPU share (never raises them).

    torch sizes its OpenMP pool by the VISIBLE cores (128 on an MI355X box whose container is
    allowed 16).  Every CPU tensor op of a few hundred thousand elements then wakes 128 threads
    that spin for their block time; together they exhaust the container's CPU quota within a
    scheduler period and the kernel freezes the whole process -- the thread that enqueues GPU
    work included -- for the rest of it.  Measured (r3, scripts/trainer_step_anatomy.py): 60-90 ms
    stalls at arbitrary places of every second or third Trainer.fit step, the GPU idle meanwhile,
    28-50 ms per step on average; 13.2 ms with one OpenMP thread.  Returns the thread count."""
    import torch
    share = cpu_share()
    try:  # one process per GPU: the ranks of a node split the container's share
        share = max(1, share // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
    except ValueError:
        pass
    n = min(torch.get_num_threads(), share)
    if n != torch.get_num_threads():
        torch.set_num_threads(n)
    return n
