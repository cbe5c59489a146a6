"""GPU: BASELINE configs[4] as written -- fp16 OPERANDS, fp32 ACCUMULATION -- in the
autoregressive generator (kernel MVN_GEN_PIPE_F16, C = K = 128, Q = 256).

No reference output exists at this precision (the reference's reduced-precision precedent,
torch.autocast in movenet/trainer.py:124, needs a CUDA device), so parity is a TOLERANCE
statement against the pinned fp32 path plus agreement with an independent restatement of the
same fp16 arithmetic (oracle/ring_oracle.c with half operands):

  * |logits_fp16 - logits_fp32| <= FP16_TOL of the fp32 logit range (2^-11 relative rounding of
    every operand, accumulated over 60 layers; measured ~1e-3, bound 5e-3);
  * the kernel is as close to the fp32 logits as the restated fp16 arithmetic is (two fp16
    implementations differ from each other by as much as from fp32: a value that falls next to
    an fp16 rounding boundary flips with the last fp32 bit of its accumulation order);
  * greedy class indices equal the fp32 path's wherever the fp32 top-2 margin exceeds twice
    the tolerance (margin-checked), and on a free run of sharpened weights.
fp32 stays the default precision (WaveNet.generate_precision)."""
import numpy as np
import pytest
import torch

from helpers import one_hot, synthetic_indices
from movenet_amd import _native as N
from movenet_amd.utils.weights import make_state_dict
from oracle import ring_c
from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FP16_TOL = 5e-3
CFG5 = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)


def _gen(cfg, sd, batch, n_total, variant, **kw):
    from movenet_amd.generation import RingGenerator
    sd = {k: v.to(DEV) for k, v in sd.items()}
    return RingGenerator(**cfg, state_dict=sd, batch=batch, n_total=n_total, device=DEV, variant=variant, **kw)


@pytest.mark.parametrize("cfg,B,n_new", [
    (dict(layer_size=5, stack_size=2, input_channels=256, residual_channels=128, skip_channels=128), 3, 48),
    (CFG5, 1, 24),
    (CFG5, 4, 24),
])
def test_fp16_generator_logits_within_tolerance(cfg, B, n_new):
    sd = make_state_dict(**cfg, seed=2, gain=1.5, head_gain=6.0)
    dims = O.Dims(**cfg)
    rf = dims.receptive_fields
    hist = synthetic_indices(B, rf + n_new, 256, 5)
    # fp32 oracle (C restatement, pinned by G2/G3) and the fp16-operand restatement
    c32, l32 = ring_c.generate_ring_c(sd, dims, hist.numpy()[:, :rf], rf + n_new, forced_idx=hist.numpy(), threads=4)
    c16, l16 = ring_c.generate_ring_c(sd, dims, hist.numpy()[:, :rf], rf + n_new, forced_idx=hist.numpy(), threads=4,
                                      operand_dtype=np.float16)
    g = _gen(cfg, sd, B, rf + n_new, N.GEN_PIPE_F16)
    assert g.variant == N.GEN_PIPE_F16
    choices, logits = g.teacher_forced(hist.to(DEV), logits_t0=rf)
    g.check_errors()
    logits, choices = logits.cpu().numpy(), choices[:, rf:].cpu().numpy()
    scale = np.abs(l32).max()
    err_kernel = np.abs(logits - l32).max() / scale
    err_restated = np.abs(l16 - l32).max() / scale
    err_between = np.abs(logits - l16).max() / scale
    print(f"L={dims.n_layers} B={B}: |fp16 kernel - fp32| {err_kernel:.2e}, |fp16 restated - fp32| "
          f"{err_restated:.2e}, |kernel - restated| {err_between:.2e} of the logit range")
    assert err_kernel < FP16_TOL and err_between < FP16_TOL
    assert err_kernel < 3 * err_restated + 1e-4
    assert err_kernel > 1e-6  # it IS the reduced-precision arithmetic, not the fp32 kernel
    # margin-checked greedy choices: wherever the fp32 decision is clear, fp16 makes it too
    top2 = np.sort(l32, axis=2)[:, :, -2:]
    clear = (top2[:, :, 1] - top2[:, :, 0]) > 2 * FP16_TOL * scale
    assert clear.mean() > 0.5
    assert np.array_equal(choices[clear], c32[:, rf:][clear])


def test_fp16_free_run_and_model_api():
    """Sharpened config-5 weights: a 40-step greedy free run in fp16 equals the fp32 run, chunked
    launches carry the queues, and WaveNet.generate honours generate_precision."""
    from movenet_amd.wavenet import WaveNet
    sd = make_state_dict(**CFG5, seed=4, gain=1.5, head_gain=6.0)
    dims = O.Dims(**CFG5)
    rf, n_new, B = dims.receptive_fields, 40, 2
    pidx = synthetic_indices(B, rf, 256, 9)
    g = _gen(CFG5, sd, B, rf + n_new, N.GEN_PIPE_F16)
    g.prime(pidx.to(DEV))
    for _ in range(0, n_new, 8):
        g.advance(8)
    g.check_errors()
    want = g.samples.cpu().numpy()
    g1 = _gen(CFG5, sd, B, rf + n_new, N.GEN_PIPE_F16)
    g1.prime(pidx.to(DEV))
    g1.advance(n_new)
    assert np.array_equal(g1.samples.cpu().numpy(), want)  # chunked launches carry the queues
    # the fp32 path (C restatement) fed the fp16 run's own history: wherever its decision is
    # clear (top-2 margin above twice the tolerance) the fp16 run made the same one
    c32, l32 = ring_c.generate_ring_c(sd, dims, pidx.numpy(), rf + n_new, forced_idx=want, threads=2)
    top2 = np.sort(l32, axis=2)[:, :, -2:]
    clear = (top2[:, :, 1] - top2[:, :, 0]) > 2 * FP16_TOL * np.abs(l32).max()
    assert clear.mean() > 0.5 and np.array_equal(want[:, rf:][clear], c32[:, rf:][clear])
    assert len(np.unique(want[:, rf:])) > 4
    model = WaveNet(**CFG5)
    model.load_state_dict(sd, strict=False)
    model.to(DEV)
    assert model.generate_precision == "fp32"
    model.generate_precision = "fp16"
    out = model.generate(one_hot(pidx, 256).to(DEV), n_samples=rf + n_new, temperature=0.0)
    assert np.array_equal(out.argmax(1).cpu().numpy(), want)
    # sampling shares the Philox stream and the step-closing code with the fp32 kernels
    gs = _gen(CFG5, sd, B, rf + 8, N.GEN_PIPE_F16, temperature=1.0, seed=3)
    gs.prime(pidx.to(DEV))
    gs.advance(8)
    gs.check_errors()
    assert gs.samples[:, rf:].min().item() >= 0 and len(torch.unique(gs.samples[:, rf:])) > 4
    with pytest.raises(ValueError):
        WaveNet(10, 3, 256, 64, 64).generate_precision = "fp16"


def test_config5_full_length_one_second_of_audio():
    """BASELINE configs[4] at its STATED length: 22 050 samples (1 s of 22.05 kHz audio) generated
    by the fp16-operand kernel from an RF = 6144 prompt, batch 1, in ONE launch -- sampled at the
    reference's default temperature 1.0 (a greedy run of random weights settles into a cycle of
    three classes within a few hundred steps; sampling keeps the second of audio varied).
      * three chunked launches (7000 + 7000 + 8050 steps) produce the same 22 050 samples;
      * teacher-forced over its own history the kernel reproduces >= 99.9 % of its own draws (the
        free run's queues were primed by the fp16 FORWARD over the prompt, the teacher-forced pass
        steps through it: same rounding points, another summation order -- a draw can change only
        where the uniform falls within that rounding of a CDF edge);
      * against the fp32 ring oracle (C restatement, pinned by G2/G3) fed that same history, on a
        SAMPLED subset of the steps (every 89th: 248 of them, spread over the whole second):
        logits within FP16_TOL of the fp32 logit range, and the same arg-max wherever the fp32
        top-2 margin exceeds twice the tolerance."""
    sd = make_state_dict(**CFG5, seed=2, gain=1.5, head_gain=6.0)
    dims = O.Dims(**CFG5)
    rf, n_new, B = dims.receptive_fields, 22050, 1
    pidx = synthetic_indices(B, rf, 256, 5)
    g = _gen(CFG5, sd, B, rf + n_new, N.GEN_PIPE_F16, temperature=1.0, seed=7)
    g.prime(pidx.to(DEV))
    g.advance(n_new)               # one launch, 22 050 steps
    g.check_errors()
    run = g.samples.clone()
    assert len(torch.unique(run[:, rf:])) > 32
    g2 = _gen(CFG5, sd, B, rf + n_new, N.GEN_PIPE_F16, temperature=1.0, seed=7)
    g2.prime(pidx.to(DEV))
    for n in (7000, 7000, 8050):
        g2.advance(n)
    g2.check_errors()
    assert torch.equal(g2.samples, run)
    choices, logits = g.teacher_forced(run, logits_t0=rf)
    g.check_errors()
    assert (choices[:, rf:] == run[:, rf:]).float().mean().item() >= 0.999
    hist = run.cpu().numpy()
    _, l32 = ring_c.generate_ring_c(sd, dims, hist[:, :rf], rf + n_new, forced_idx=hist, threads=1)
    sub = np.arange(0, n_new, 89)
    lg16, lg32 = logits.cpu().numpy()[:, sub], l32[:, sub]
    scale = np.abs(l32).max()
    err = np.abs(lg16 - lg32).max() / scale
    print(f"config 5, 22050 steps: |fp16 kernel - fp32 oracle| {err:.2e} of the logit range on {sub.size} sampled steps")
    assert 1e-6 < err < FP16_TOL
    top2 = np.sort(lg32, axis=2)[:, :, -2:]
    clear = (top2[:, :, 1] - top2[:, :, 0]) > 2 * FP16_TOL * scale
    assert clear.mean() > 0.5
    assert np.array_equal(lg16.argmax(2)[clear], lg32.argmax(2)[clear])


def test_fp16_pipelines_serve_several_sequences_in_turn():
    """r3: the eight fp16 pipelines (one per XCD) serve up to eight sequences each in turn within
    one launch (gen_pipe_h16_kernel<true>).  20 sequences = rounds of 8 + 8 + 4: the same samples
    as the same sequences run eight at a time (one round: gen_pipe_h16_kernel<false>), also when
    the launch is chunked; sampled draws use the Philox counter of the SEQUENCE."""
    sd = make_state_dict(**CFG5, seed=4, gain=1.5, head_gain=6.0)
    rf, n_new, B = O.Dims(**CFG5).receptive_fields, 16, 20
    pidx = synthetic_indices(B, rf, 256, 9)
    for temperature in (0.0, 1.0):
        want = []
        for b0 in range(0, B, 8):
            # (a one-round launch numbers its sequences from 0: give it the counters of b0 .. by
            # running the greedy case only through it; the sampled case is compared below)
            g1 = _gen(CFG5, sd, min(8, B - b0), rf + n_new, N.GEN_PIPE_F16, temperature=0.0)
            g1.prime(pidx[b0:b0 + 8].to(DEV))
            g1.advance(n_new)
            g1.check_errors()
            want.append(g1.samples.clone())
        want = torch.cat(want)
        g = _gen(CFG5, sd, B, rf + n_new, N.GEN_PIPE_F16, temperature=temperature, seed=11)
        g.prime(pidx.to(DEV))
        g.advance(6)
        g.advance(n_new - 6)
        g.check_errors()
        if temperature == 0.0:
            assert torch.equal(g.samples, want)
        else:
            # same seed, one launch: identical; the draws differ between sequences and from greedy
            g2 = _gen(CFG5, sd, B, rf + n_new, N.GEN_PIPE_F16, temperature=temperature, seed=11)
            g2.prime(pidx.to(DEV))
            g2.advance(n_new)
            g2.check_errors()
            assert torch.equal(g2.samples, g.samples)
            assert not torch.equal(g.samples[:, rf:], want[:, rf:])
            assert len(torch.unique(g.samples[:, rf:])) > 8


def test_fp16_stage_forms_agree_over_a_queue_wrap():
    """The fp16 generator runs THREE layers per stage (21 stages) when a pipeline serves one sequence (batch <= 8:
    gen_pipe_h16_kernel<false, true>) and TWO (31 stages) when it serves several in turn (<true, true>), so batch 8 and
    batch 9 run different kernels.  Teacher-forced over the same histories for 520 steps behind the prompt -- past the
    wrap of the longest dilation queue (512) -- the two forms give the same logits (same per-layer arithmetic, another
    partition into stages) and the same arg-max wherever the margin is clear."""
    sd = make_state_dict(**CFG5, seed=6, gain=1.5, head_gain=6.0)
    rf = O.Dims(**CFG5).receptive_fields
    n_total = rf + 520
    hist = synthetic_indices(9, n_total, 256, 21)
    g8 = _gen(CFG5, sd, 8, n_total, N.GEN_PIPE_F16, temperature=0.0)
    c8, l8 = g8.teacher_forced(hist[:8].to(DEV), logits_t0=rf)
    g8.check_errors()
    g9 = _gen(CFG5, sd, 9, n_total, N.GEN_PIPE_F16, temperature=0.0)
    c9, l9 = g9.teacher_forced(hist.to(DEV), logits_t0=rf)
    g9.check_errors()
    l8, l9 = l8.cpu().numpy(), l9[:8].cpu().numpy()
    rng = float(l8.max() - l8.min())
    err = float(np.abs(l8 - l9).max()) / rng
    assert err <= 0.1 * FP16_TOL, err   # (measured: see the assertion message if it ever moves)
    top2 = np.sort(l8, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 2 * FP16_TOL * rng
    a8, a9 = l8.argmax(-1), l9.argmax(-1)
    assert clear.mean() > 0.5 and np.array_equal(a8[clear], a9[clear])


def test_fp16_capacity_one_xcd():
    """60 layers = 31 stages of two layers: one XCD per pipeline, eight pipelines per launch, up to
    eight sequences each (fp32: 61 stages over two XCDs, four sequences)."""
    from movenet_amd.generation import max_pipe_batch
    d5 = N.make_dims(10, 6, 256, 128, 128)
    assert max_pipe_batch(d5, N.GEN_PIPE) == 64 and max_pipe_batch(d5, N.GEN_PIPE_F16) == 64
    assert N.lib().mvn_gen_variant(d5, N.GEN_AUTO, 1) == N.GEN_PIPE  # fp32 stays the default


def test_fp16_forward_mfma_within_tolerance(golden):
    """mvn_forward_f16 (v_mfma_f32_32x32x16_f16 operands, fp32 accumulation) against the
    reference's own fp32 logits of the 60-layer, 128-channel model (fixture G6, recorded from
    movenet's WaveNet.forward as the yard-stick of this tolerance), against this build's fp32
    forward, and against the fp16 generator stepping over the same history (same rounding
    points, different summation order)."""
    from helpers import rel_err, weights_of
    from movenet_amd.wavenet import WaveNet
    fx = golden("g6_l60_forward.npz")
    cfg, dims, sd = weights_of(fx)
    T = int(fx["T"])
    hist = synthetic_indices(1, T, 256, int(fx["idx_seed"]))
    x = one_hot(hist, 256).to(DEV)
    m = WaveNet(**cfg)
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    with torch.no_grad():
        l32 = m(x, output_unnormalized=False, remove_last=False)
        m.forward_precision = "fp16"
        l16 = m(x, output_unnormalized=False, remove_last=False)
        p16 = m(x)
    scale = float(np.abs(fx["logits"]).max())
    e_ref = np.abs(l16.cpu().numpy() - fx["logits"]).max() / scale
    e_own = (l16 - l32).abs().max().item() / scale
    print(f"fp16 forward: {e_ref:.2e} of the logit range from the reference's fp32 logits (G6), {e_own:.2e} from "
          f"this build's fp32 forward")
    assert e_ref < FP16_TOL and e_own < FP16_TOL and e_own > 1e-6
    assert (p16.sum(1) - 1).abs().max().item() < 1e-5
    with pytest.raises(RuntimeError):   # inference only
        m(x).sum().backward()
    # the fp16 generator, primed by this forward and teacher-forced over the same history
    rf = dims.receptive_fields
    g = _gen(cfg, sd, 1, T, N.GEN_PIPE_F16)
    _, lg = g.teacher_forced(hist.to(DEV), logits_t0=rf)
    g.check_errors()
    want = l16[:, :, :-1].permute(0, 2, 1)   # (B, T - rf, Q): predicts times rf .. T-1
    assert (lg - want).abs().max().item() / scale < FP16_TOL
    # small model, every dims path of the kernel family (C = 16: padded rows / k)
    cfg1 = dict(layer_size=3, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)
    sd1 = make_state_dict(**cfg1, seed=7, gain=2.0, head_gain=4.0)
    m1 = WaveNet(**cfg1)
    m1.load_state_dict(sd1, strict=True)
    m1.to(DEV)
    x1 = one_hot(synthetic_indices(2, 300, 64, 3), 64).to(DEV)
    with torch.no_grad():
        a = m1(x1, output_unnormalized=False)
        m1.forward_precision = "fp16"
        b = m1(x1, output_unnormalized=False)
    assert 1e-7 < (a - b).abs().max().item() / a.abs().max().item() < FP16_TOL
