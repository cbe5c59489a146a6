"""GPU: local conditioning (video encoder, learned upsampler, context convs).

* ``upsample_video`` is PINNED: fixture G7 was recorded from the reference itself.
* The conditioned gated layer is a BUILD DEFINITION (the reference raises there,
  SURVEY.md Q6/Q7): these tests compare against this repo's oracle, whose context
  alignment is the same definition -- parity with the reference is UNPINNED.
Tolerances as in test_forward_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import one_hot, rel_err, synthetic_indices, weights_of
from movenet_amd.utils.weights import make_state_dict
from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(cfg, sd):
    from movenet_amd.wavenet import WaveNet
    m = WaveNet(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


def test_g7_upsample_video_pinned(golden):
    fx = golden("g7_upsample_video.npz")
    cfg, dims, sd = weights_of(fx)
    rng = np.random.default_rng(int(fx["video_seed"]))
    video = torch.from_numpy(rng.random((1, 160, 64, 64, 1), dtype=np.float32)).to(DEV)
    with torch.no_grad():
        up = _model(cfg, sd).upsample_video(video)
    assert up.shape == (1, cfg["residual_channels"], 160000)
    assert rel_err(up[:, :, torch.from_numpy(fx["cols"]).to(DEV)].cpu(), fx["up_cols"]) < 1e-5
    got = up.double().abs().sum().item()
    assert abs(got - float(fx["up_abs_sum"])) / float(fx["up_abs_sum"]) < 1e-5


@pytest.mark.parametrize("cfg,frames,B", [
    (dict(layer_size=2, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16), 2, 2),
    (dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64), 4, 1),
])
def test_conditioned_forward_backward_vs_oracle(monkeypatch, cfg, frames, B):
    import movenet_amd.wavenet as W
    T = 1000 * frames
    # Q8: the module constants fix the clip length; F frames <-> 1000 F samples
    monkeypatch.setattr(W, "MAX_AUDIO_FRAMES", T)
    monkeypatch.setattr(W, "MAX_VIDEO_FRAMES", frames)
    sd = make_state_dict(**cfg, seed=17)
    dims = O.Dims(**cfg)
    Q = cfg["input_channels"]
    x = one_hot(synthetic_indices(B, T, Q, 1234), Q)
    rng = np.random.default_rng(4321)
    video = torch.from_numpy(rng.random((B, frames, 64, 64, 1), dtype=np.float32))

    # oracle (CPU autograd), trainer arithmetic
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ctx = O.upsample_video(params, video, expect_frames=T)
    out_o = O.forward(params, dims, x, context=ctx)
    target = x[:, :, dims.receptive_fields:].argmax(1)
    loss_o = F.cross_entropy(out_o, target)
    loss_o.backward()

    m = _model(cfg, sd).train()
    out = m(x.to(DEV), video.to(DEV))
    assert out.shape == out_o.shape
    assert np.abs(out.detach().cpu().numpy() - out_o.detach().numpy()).max() < 2e-6
    with torch.no_grad():
        logits = m(x.to(DEV), video.to(DEV), output_unnormalized=False, remove_last=False)
        want = O.forward(sd, dims, x, context=ctx.detach(), output_unnormalized=False, remove_last=False)
    assert rel_err(logits.cpu(), want) < 2e-5
    loss = F.cross_entropy(out, target.to(DEV))
    loss.backward()
    assert abs(loss.item() - loss_o.item()) < 2e-6
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    want_g = {k: p.grad for k, p in params.items() if p.grad is not None}
    assert sorted(got) == sorted(want_g)  # video, context and decoder parameters all take part
    for k in want_g:
        assert rel_err(got[k].cpu(), want_g[k]) < 3e-4, k


def test_conditioned_input_checks():
    cfg = dict(layer_size=2, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)
    m = _model(cfg, make_state_dict(**cfg, seed=1))
    x = one_hot(synthetic_indices(1, 3000, 64, 1), 64).to(DEV)
    video = torch.rand(1, 3, 64, 64, 1, device=DEV)
    with pytest.raises(AssertionError):   # 3 frames upsample to 3000 != MAX_AUDIO_FRAMES (wavenet.py:155)
        m(x, video)
    with pytest.raises(ValueError):
        m.upsample_video(torch.rand(1, 3, 32, 32, 1, device=DEV))


@pytest.mark.parametrize("cfg,frames,n_new", [
    (dict(layer_size=2, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16), 2, 80),
    (dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64), 4, 48),
])
def test_conditioned_generate_vs_oracle(monkeypatch, cfg, frames, n_new):
    """WaveNet.generate(video=...) (GENERIC kernel for the small model, PIPE for the
    30-layer one) == the oracle's windowed AND ring formulations with the same context."""
    import movenet_amd.wavenet as W
    T = 1000 * frames
    monkeypatch.setattr(W, "MAX_AUDIO_FRAMES", T)
    monkeypatch.setattr(W, "MAX_VIDEO_FRAMES", frames)
    sd = make_state_dict(**cfg, seed=3 if cfg["residual_channels"] == 16 else 1,
                         gain=3.0 if cfg["residual_channels"] == 16 else 2.0, head_gain=6.0)
    dims = O.Dims(**cfg)
    Q, rf, B = cfg["input_channels"], dims.receptive_fields, 2
    N_ = rf + n_new
    pidx = synthetic_indices(B, rf, Q, 77)
    prompt = one_hot(pidx, Q)
    rng = np.random.default_rng(4321)
    video = torch.from_numpy(rng.random((B, frames, 64, 64, 1), dtype=np.float32) * 4.0)
    with torch.no_grad():
        ctx = O.upsample_video(sd, video, expect_frames=T)
    ridx, rlog = O.generate_ring(sd, dims, pidx.numpy(), N_, context=ctx.numpy())
    if cfg["residual_channels"] == 16:  # the windowed algorithm is cheap enough only here
        want = O.generate_windowed(sd, dims, prompt, n_samples=N_, temperature=0.0, context=ctx)
        assert np.array_equal(want.argmax(1).numpy(), ridx)
    # the context must matter for this to be a test of the conditioning path
    plain, _ = O.generate_ring(sd, dims, pidx.numpy(), N_)
    assert not np.array_equal(plain, ridx)
    m = _model(cfg, sd)
    out = m.generate(prompt.to(DEV), video.to(DEV), n_samples=N_, temperature=0.0)
    assert np.array_equal(out.argmax(1).cpu().numpy(), ridx)


def test_conditioned_fused_backward_matches_generic_kernels_and_oracle(monkeypatch):
    """Conditioned layers at C = K = 64 take THREE fused passes per layer (csrc/fused_bwd.h): dz +
    residual/skip weight gradients, dx + the audio taps' weight gradients, and -- round 3 --
    dctx + the context-conv weight / bias gradients from the same dfg tile
    (bwd_dctx_wgctx64_kernel).  8 clips of 6 frames (T = 6000: ragged against the 64-step tiles,
    and long enough that the scratch holds the fused form's slabs) against the generic kernels
    (MOVENET_HIP_NO_FUSED_BACKWARD=1, same process) and torch autograd on the oracle; the video
    encoder's gradients are in the comparison, so the accumulated dctx is too.  The loss is a
    weighted sum of squared LOGITS (as in test_fused_backward_kernels_match_two_kernel_forms_and_oracle):
    the trainer's cross-entropy on probabilities has gradients of ~1e-8 that are sums of 48 000
    terms of mixed sign, whose fp32 rounding alone is 1e-3 of their size on CPU and GPU alike."""
    import movenet_amd.wavenet as W
    frames, B = 6, 8
    T = 1000 * frames
    # guard bands behind every scratch tensor of the backward pass (ops._GuardBands): r3 found the fused halves
    # writing their bias partial sums past the end of one at exactly this size
    monkeypatch.setenv("MOVENET_DEBUG_GUARD", "1")
    monkeypatch.setattr(W, "MAX_AUDIO_FRAMES", T)
    monkeypatch.setattr(W, "MAX_VIDEO_FRAMES", frames)
    cfg = dict(layer_size=3, stack_size=2, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=23, gain=1.5)
    dims = O.Dims(**cfg)
    x = one_hot(synthetic_indices(B, T, 256, 1234), 256)
    video = torch.from_numpy(np.random.default_rng(4321).random((B, frames, 64, 64, 1), dtype=np.float32))
    w = torch.linspace(0.5, 1.5, 256).view(1, 256, 1)

    def grads(no_fused):
        if no_fused:
            monkeypatch.setenv("MOVENET_HIP_NO_FUSED_BACKWARD", "1")
        else:
            monkeypatch.delenv("MOVENET_HIP_NO_FUSED_BACKWARD", raising=False)
        m = _model(cfg, sd).train()
        out = m(x.to(DEV), video.to(DEV), output_unnormalized=False)
        loss = (out * w.to(DEV)).square().mean()
        loss.backward()
        return loss.item(), {k: (None if p.grad is None else p.grad.cpu()) for k, p in m.named_parameters()}

    (loss_f, fused), (loss_p, plain) = grads(False), grads(True)
    assert loss_f == loss_p
    for k in fused:
        assert (fused[k] is None) == (plain[k] is None), k
        if fused[k] is not None:
            assert rel_err(fused[k], plain[k]) < 2e-5, k  # fp32 sums in another order
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ctx = O.upsample_video(params, video, expect_frames=T)
    loss_o = (O.forward(params, dims, x, context=ctx, output_unnormalized=False) * w).square().mean()
    loss_o.backward()
    assert abs(loss_f - loss_o.item()) < 2e-5 * abs(loss_o.item())
    for k, g in fused.items():
        if params[k].grad is None:
            assert g is None, k
        else:
            assert rel_err(g, params[k].grad) < 3e-4, k


@pytest.mark.parametrize("channels,which,B", [(64, "fold", 20), (64, "pipe", 30), (128, "pipe", 6)])
def test_conditioned_rounds_match_generic_kernel(channels, which, B):
    """Video-conditioned generation with several sequences per pipeline (r3): 20 sequences on FOLD's 16
    pipelines, 30 on PIPE's 24 (C = 64), 6 on the 4 sixty-one-stage pipelines of the C = K = 128 model -- some
    pipelines serve two sequences, and every sequence must read ITS context column: greedy indices equal to the
    GENERIC kernel's (one workgroup per sequence) on the same context."""
    from movenet_amd import _native as N
    from movenet_amd.generation import RingGenerator
    cfg = dict(layer_size=10, stack_size=3 if channels == 64 else 6, input_channels=256, residual_channels=channels,
               skip_channels=channels)
    pipelined = N.GEN_FOLD if which == "fold" else N.GEN_PIPE
    sd = {k: v.to(DEV) for k, v in make_state_dict(**cfg, seed=1, gain=2.0 if channels == 64 else 1.5,
                                                   head_gain=6.0).items() if not k.startswith("video_")}
    rf, n_new = O.Dims(**cfg).receptive_fields, 20
    pidx = synthetic_indices(B, rf, 256, 77).to(DEV)
    ctx = torch.from_numpy(np.random.default_rng(5).standard_normal((B, channels, rf + n_new)).astype(np.float32)).to(DEV)
    runs = {}
    for variant in (N.GEN_GENERIC, pipelined):
        g = RingGenerator(**cfg, state_dict=sd, batch=B, n_total=rf + n_new, device=DEV, variant=variant,
                          temperature=0.0, context=ctx)
        assert g.variant == variant
        g.prime(pidx)
        g.advance(n_new)
        g.check_errors()
        runs[variant] = g.samples.clone()
    assert torch.equal(runs[pipelined], runs[N.GEN_GENERIC])
    plain = RingGenerator(**cfg, state_dict=sd, batch=B, n_total=rf + n_new, device=DEV, variant=pipelined)
    plain.prime(pidx)
    plain.advance(n_new)
    assert not torch.equal(plain.samples, runs[pipelined])  # the context matters
