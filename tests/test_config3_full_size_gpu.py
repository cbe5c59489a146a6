"""BASELINE configs[2] at its STATED size through the public API: the 30-layer model with
video conditioning, F = 32 frames => T = 32000 samples, batch 8 (movenet/wavenet.py:149-156,
movenet/modules.py:75-77).  The oracle cannot run this size in seconds (and the conditioned
layer is a build definition, SURVEY Q6/Q7: parity UNPINNED), so the checks are the
size-independent properties of the path; the small-size comparison against the oracle is
tests/test_conditioning_gpu.py."""
import math

import numpy as np
import pytest
import torch

from helpers import one_hot, synthetic_indices
from movenet_amd import _native as N
from movenet_amd.utils.weights import make_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
FRAMES, B, T = 32, 8, 32000


@pytest.fixture()
def model(monkeypatch):
    import movenet_amd.wavenet as W
    monkeypatch.setattr(W, "MAX_AUDIO_FRAMES", T)   # Q8: module constants fix the clip length
    monkeypatch.setattr(W, "MAX_VIDEO_FRAMES", FRAMES)
    m = W.WaveNet(**CFG)
    m.load_state_dict(make_state_dict(**CFG, seed=11), strict=True)
    return m.to(DEV)


def _video(seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.random((B, FRAMES, 64, 64, 1), dtype=np.float32) * scale).to(DEV)


def test_config3_forward_backward_properties(model):
    from movenet_amd.ops import cross_entropy_on_probs
    rf = model.receptive_fields
    audio = one_hot(synthetic_indices(B, T, 256, 1234).to(DEV), 256)
    target = audio[:, :, rf:].argmax(1)
    video = _video(4321)
    model.train()
    out = model(audio, video)
    assert out.shape == (B, 256, T - rf)
    assert torch.isfinite(out).all() and (out >= 0).all()
    assert (out.sum(1) - 1).abs().max().item() < 1e-5          # probabilities (Q1)
    loss, acc = cross_entropy_on_probs(out, target)
    # cross_entropy applied to probabilities (Q2) sits at ln Q - O(1/Q) whatever the weights
    assert abs(loss.item() - math.log(256)) < 0.01
    assert 0.0 <= acc.item() <= 0.05
    loss.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    used = set(g1)
    assert any(k.startswith("video_conv") for k in used) and any(".context_conv_" in k for k in used)
    last = f"residual_conv_stack.conv_layers.{model.layer_size * model.stack_size - 1}.conv_residual."
    assert not any(k.startswith(last) for k in used)            # never reaches the output
    for k, g in g1.items():
        assert torch.isfinite(g).all(), k
    # linearity of the backward pass in the upstream gradient: d(3 * loss) = 3 * d(loss)
    model.zero_grad(set_to_none=True)
    loss3, _ = cross_entropy_on_probs(model(audio, video), target)
    (3.0 * loss3).backward()
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        scale = g1[k].abs().max().item()
        assert (p.grad - 3.0 * g1[k]).abs().max().item() <= 1e-4 * 3.0 * scale + 1e-12, k
    # the context reaches the output, and only through the conditioned path
    with torch.no_grad():
        other = model(audio, _video(99))
        plain = model(audio, None)
    assert (other - out.detach()).abs().max().item() > 1e-6
    assert (plain - out.detach()).abs().max().item() > 1e-6
    # batch rows are independent: the first two clips alone give the same rows
    with torch.no_grad():
        sub = model(audio[:2], video[:2])
    assert (sub - out.detach()[:2]).abs().max().item() < 1e-6


def test_config3_generate_properties(model):
    """Conditioned generation at B = 8 with the full 32000-column context: PIPE (what
    WaveNet.generate picks) equals GENERIC on a 64-step free run, the context changes the
    samples, and the teacher-forced replay reproduces the run; the FOLD variant agrees too."""
    from movenet_amd.generation import RingGenerator
    # sharpened weights: greedy margins far above fp32 rounding (as in fixture G3)
    model.load_state_dict(make_state_dict(**CFG, seed=1, gain=2.0, head_gain=6.0), strict=True)
    rf, n_new = model.receptive_fields, 64
    pidx = synthetic_indices(B, rf, 256, 77).to(DEV)
    prompt = one_hot(pidx, 256)
    video = _video(4321, scale=4.0)
    out = model.generate(prompt, video, n_samples=rf + n_new, temperature=0.0)
    assert model.last_generate_fallback is None
    assert out.shape == (B, 256, rf + n_new)
    assert torch.equal(out[:, :, :rf], prompt) and torch.equal(out.sum(1), torch.ones_like(out.sum(1)))
    idx = out.argmax(1).to(torch.int32)
    with torch.no_grad():
        ctx = model.upsample_video(video)
    assert ctx.shape == (B, 64, T)
    sd = {k: v for k, v in model.state_dict().items() if not k.startswith("video_")}
    runs = {}
    for variant in (N.GEN_GENERIC, N.GEN_PIPE, N.GEN_FOLD):
        g = RingGenerator(**CFG, state_dict=sd, batch=B, n_total=rf + n_new, device=DEV,
                          variant=variant, temperature=0.0, context=ctx)
        assert g.variant == variant
        g.prime(pidx)
        g.advance(n_new)
        g.check_errors()
        runs[variant] = g.samples.clone()
    assert torch.equal(runs[N.GEN_PIPE], runs[N.GEN_GENERIC])
    assert torch.equal(runs[N.GEN_FOLD], runs[N.GEN_GENERIC])
    assert torch.equal(runs[N.GEN_FOLD], idx)  # what WaveNet.generate ran
    g = RingGenerator(**CFG, state_dict=sd, batch=B, n_total=rf + n_new, device=DEV,
                      variant=N.GEN_GENERIC, temperature=0.0, context=ctx)
    choices, _ = g.teacher_forced(idx, logits_t0=rf)
    assert torch.equal(choices[:, rf:], idx[:, rf:])
    plain = model.generate(prompt, None, n_samples=rf + n_new, temperature=0.0)
    assert not torch.equal(plain, out)
    assert len(torch.unique(idx[:, rf:])) > 8
