"""GPU: host-side paths added in round 3 -- the conditioned step's gradients (decoder, context
convs AND video encoder) in ONE flat buffer (one all-reduce message, one AdamW launch), the RCCL
backend executed at world size 1, FlatAdamW state round trip, model copies after a forward, the
trainer's one-reduction gradient norm."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import one_hot, synthetic_indices
from movenet_amd.utils.weights import make_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(layer_size=2, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)


def _conditioned_step(monkeypatch, cfg=SMALL, frames=2, B=2, seed=17, detach_slot=False):
    import movenet_amd.wavenet as W
    from movenet_amd.wavenet import WaveNet
    monkeypatch.setattr(W, "MAX_AUDIO_FRAMES", 1000 * frames)
    monkeypatch.setattr(W, "MAX_VIDEO_FRAMES", frames)
    m = WaveNet(**cfg)
    m.load_state_dict(make_state_dict(**cfg, seed=seed))
    m.to(DEV).train()
    Q = cfg["input_channels"]
    x = one_hot(synthetic_indices(B, 1000 * frames, Q, 1234), Q).to(DEV)
    video = torch.from_numpy(np.random.default_rng(4321).random((B, frames, 64, 64, 1), dtype=np.float32)).to(DEV)
    if detach_slot:  # the round-2 path: the up-sampler's backward allocates its own eight buffers
        from movenet_amd.ops import wavenet_forward_loss
        ctx = m.upsample_video(video)
        del ctx._mvn_video_slot
        loss, _, _ = wavenet_forward_loss(m, x, ctx)
    else:
        loss, _, _ = m(x, video, return_loss=True)
    loss.backward()
    return m, loss


def test_conditioned_gradients_are_one_buffer_and_one_adamw_launch(monkeypatch):
    from movenet_amd.optim import FlatAdamW, order_like_backward
    from movenet_amd.parallel import contiguous_grad_span
    m, loss = _conditioned_step(monkeypatch)
    used = [p for p in m.parameters() if p.grad is not None]
    n_all = sum(p.numel() for p in m.parameters())
    span = contiguous_grad_span(used)
    assert span is not None and span.numel() == n_all  # decoder + context + video encoder, one storage
    # same bits as the path with separate video-gradient buffers
    m2, loss2 = _conditioned_step(monkeypatch, detach_slot=True)
    assert contiguous_grad_span([p for p in m2.parameters() if p.grad is not None]) is None
    assert float(loss.detach()) == float(loss2.detach())
    for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert (p.grad is None) == (q.grad is None), k
        if p.grad is None:
            continue
        if k.startswith("video_"):
            # the up-sampler's weight gradients are summed with float atomics (wgrad_kernel): the same
            # kernels on the same inputs, equal up to the order of those additions
            assert (p.grad - q.grad).abs().max() <= 1e-4 * q.grad.abs().max().clamp_min(1e-30), k
        else:
            # same kernels, same inputs; the small model's embedding gradient adds with float atomics
            # when its table copies do not fit the scratch, so "equal" is up to the order of additions
            assert (p.grad - q.grad).abs().max() <= 1e-6 * q.grad.abs().max().clamp_min(1e-30), k
    # the one-reduction gradient norm of the trainer == the per-parameter form (gaps are zero)
    per_param = torch.stack([p.grad.norm(2) for p in used]).norm(2)
    assert abs(float(span.norm(2)) - float(per_param)) <= 1e-6 * float(per_param)
    # one optimizer launch over parameters, gradients and moments
    ref = copy.deepcopy(m)
    for p, q in zip(ref.parameters(), m.parameters()):
        p.grad = None if q.grad is None else q.grad.clone()
    opt = FlatAdamW(order_like_backward(m, with_context=True), lr=1e-3, weight_decay=0.01)
    opt.step()
    assert opt.last_launches == 1
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)
    topt.step()
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-8), k


def test_flat_adamw_state_dict_round_trip_resumes_where_it_stopped():
    """load_state_dict must restore the moments and step counts INTO the flat
    buffers; a resumed run then equals an uninterrupted torch.optim.AdamW run."""
    from movenet_amd.optim import FlatAdamW
    torch.manual_seed(0)
    shapes = [(7, 5), (5,), (3, 4, 2), (11,)]
    init = [torch.randn(s, device=DEV) for s in shapes]
    grads = [[torch.randn(s, device=DEV) for s in shapes] for _ in range(5)]

    def params():
        return [torch.nn.Parameter(t.clone()) for t in init]

    def run(opt, ps, steps):
        for g in steps:
            for i, (p, gi) in enumerate(zip(ps, g)):
                p.grad = None if i == 3 and g is grads[0] else gi.clone()  # a late first gradient
            opt.step()

    ref_p = params()
    ref = torch.optim.AdamW(ref_p, lr=3e-3, weight_decay=0.05)
    run(ref, ref_p, grads)
    a_p = params()
    a = FlatAdamW(a_p, lr=3e-3, weight_decay=0.05)
    run(a, a_p, grads[:3])
    saved = copy.deepcopy(a.state_dict())
    b_p = [torch.nn.Parameter(p.detach().clone()) for p in a_p]
    b = FlatAdamW(b_p, lr=1.0, weight_decay=0.0)  # different hyper-parameters: the load must win
    b.load_state_dict(saved)
    assert b.param_groups[0]["lr"] == 3e-3 and b._steps == a._steps
    assert torch.equal(b.exp_avg, a.exp_avg) and b.state["flat"]["exp_avg"] is b.exp_avg
    run(b, b_p, grads[3:])
    for p, q in zip(b_p, ref_p):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)
    with pytest.raises(ValueError):
        FlatAdamW([torch.nn.Parameter(torch.zeros(3, device=DEV))]).load_state_dict(saved)


def test_model_can_be_copied_and_pickled_after_a_forward(tmp_path):
    """no torch.Stream in the module's __dict__."""
    from movenet_amd.wavenet import WaveNet
    m = WaveNet(**SMALL)
    m.load_state_dict(make_state_dict(**SMALL, seed=3))
    m.to(DEV)
    x = one_hot(synthetic_indices(1, 64, 64, 5), 64).to(DEV)
    with torch.no_grad():
        y = m(x)
    m2 = copy.deepcopy(m)
    torch.save(m, tmp_path / "whole_model.pt")
    with torch.no_grad():
        assert torch.equal(m2(x), y)


def test_forward_hooks_fire_on_the_trainers_fused_step():
    """_shared_step goes through Dance2Music.forward and WaveNet.forward."""
    from movenet_amd.config import ModelConfig, TrainingConfig
    from movenet_amd.pytorch_lightning_trainer import Dance2Music
    cfg = TrainingConfig(model_config=ModelConfig(**SMALL), batch_size=2, use_video=False, scheduler=None)
    d = Dance2Music("synthetic://clips=2,frames=100,seed=1", cfg).to(DEV)
    fired = []
    d.register_forward_hook(lambda *a: fired.append("module"))
    d.model.register_forward_hook(lambda *a: fired.append("wavenet"))
    batch = next(iter(d.train_dataloader()))
    out = d.training_step(batch, 0)
    assert fired == ["wavenet", "module"] and out["loss"].requires_grad


_NCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import movenet_amd.wavenet as W
from movenet_amd.parallel import FlatGradSync
from movenet_amd.utils.weights import make_state_dict, one_hot, synthetic_indices
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT={port!r}, RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=dev)          # RCCL, as bench.py / the trainer do at N > 1
assert dist.get_backend() == "nccl"
cfg = dict(layer_size=2, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)
W.MAX_AUDIO_FRAMES, W.MAX_VIDEO_FRAMES = 2000, 2
m = W.WaveNet(**cfg); m.load_state_dict(make_state_dict(**cfg, seed=17)); m.to(dev).train()
sync = FlatGradSync(m.parameters(), 1, single_rank_collective=True)
sync.broadcast_parameters(0)
x = one_hot(synthetic_indices(2, 2000, 64, 1234), 64).to(dev)
video = torch.from_numpy(np.random.default_rng(4321).random((2, 2, 64, 64, 1), dtype=np.float32)).to(dev)
loss, _, _ = m(x, video, return_loss=True)
loss.backward()
before = [None if p.grad is None else p.grad.clone() for p in m.parameters()]
sent = sync.sync_gradients()
torch.cuda.synchronize()
n_all = sum(p.numel() for p in m.parameters())
assert sync.last_path == "contiguous-span" and sent == n_all, (sync.last_path, sent, n_all)
for p, g in zip(m.parameters(), before):
    assert (p.grad is None) == (g is None)
    if g is not None:
        assert torch.equal(p.grad, g)   # sum over one rank / 1
t = torch.ones(4, device=dev); dist.all_reduce(t); assert float(t.sum()) == 4.0
dist.barrier(); dist.destroy_process_group()
print("NCCL_WS1_OK", sent)
"""


def test_rccl_backend_at_world_size_one_reduces_the_conditioned_span():
    """RCCL ("nccl" backend, device_id=...) is loaded, initialised and used for the ONE message
    a conditioned step sends -- on the one GPU this box has, before any 8-GPU run."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, "-c", _NCCL_SCRIPT.format(root=ROOT, port=port)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "NCCL_WS1_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
