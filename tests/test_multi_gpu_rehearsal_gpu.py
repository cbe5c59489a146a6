"""The N > 1 control flow of bench.py on a ONE-GPU box: `python bench.py --gpus 2` starts its
two ranks itself (no torchrun), both on cuda:0 over gloo (the rehearsal knobs
MOVENET_BENCH_SINGLE_DEVICE / MOVENET_BENCH_BACKEND, never set by the driver).  Checks the
data-parallel train leg: identical parameters on both ranks after the optimizer steps, the
flat gradient travelling as ONE in-place all-reduce, and the whole-job aggregation of the line.
(The generate leg runs on the STREAM kernel here: two PIPE grids of 144 workgroups cannot be
co-resident on one GPU's 256 CUs.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_device():
    env = dict(os.environ, MOVENET_BENCH_SINGLE_DEVICE="1", MOVENET_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--new-samples", "128", "--variant", "2", "--no-cpu-baseline", "--no-extras"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["cpu_baseline"] is None
    assert out["value"] == pytest.approx(2 * out["samples_per_s_per_gpu"])
    tr = out["train_step"]
    assert "error" not in tr, tr
    assert tr["global_batch"] == 32 and tr["tokens_per_step"] == 2 * 16 * (16000 - 3072)
    shas = tr["param_sha256_per_rank"]
    assert len(shas) == 2 and shas[0] == shas[1]
    assert tr["allreduce_path"] == "contiguous-span"
    assert tr["allreduce_floats"] >= 856320 - 64 * 64 - 64  # every used decoder parameter, one message
    # BASELINE configs[2]/[3]: the video-conditioned workload, 8 clips of 32 frames per rank --
    # decoder, context-conv AND video-encoder gradients as ONE message, one AdamW launch
    t3 = out["train_step_config3"]
    assert "error" not in t3, t3
    assert t3["conditioned"] and t3["global_batch"] == 16 and t3["seq_len"] == 32000
    assert t3["tokens_per_step"] == 2 * 8 * (32000 - 3072)
    assert len(t3["param_sha256_per_rank"]) == 2 and t3["param_sha256_per_rank"][0] == t3["param_sha256_per_rank"][1]
    assert t3["allreduce_path"] == "contiguous-span"
    assert t3["allreduce_floats"] == 1491200  # every parameter of the model in one span
    assert t3["optimizer_launches_per_step"] == 1
