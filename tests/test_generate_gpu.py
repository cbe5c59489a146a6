"""GPU parity of the ring-buffer generator (through the C ABI) against the
golden vectors recorded from the reference and against the CPU oracle.

Tolerances: raw logits within 2e-5 of the logit range (fp32 accumulation in a
different order than ATen's CPU convolutions); greedy class indices bit-exact
(the fixtures' top-2 logit margins are >= 1e-2, three orders above that)."""
import numpy as np
import pytest
import torch

from helpers import cfg_of, one_hot, rel_err, synthetic_indices, weights_of
from movenet_amd import _native as N
from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu
LOGIT_TOL = 2e-5
DEV = "cuda:0"


def _gen(cfg, sd, batch, n_total, variant=0, temperature=0.0, seed=0):
    from movenet_amd.generation import RingGenerator
    sd = {k: v.to(DEV) for k, v in sd.items()}
    return RingGenerator(**cfg, state_dict=sd, batch=batch, n_total=n_total, device=DEV,
                         variant=variant, temperature=temperature, seed=seed)


def _variants(cfg):
    from movenet_amd import _native as N
    out = [N.GEN_GENERIC]
    if cfg["residual_channels"] == 64 and cfg["skip_channels"] == 64 and cfg["input_channels"] == 256:
        out += [N.GEN_STREAM, N.GEN_PIPE, N.GEN_FOLD]
    return out


def test_small_teacher_forced_logits(golden):
    fx = golden("g1_small_forward.npz")
    cfg, dims, sd = weights_of(fx)
    B, T, rf = int(fx["B"]), int(fx["T"]), dims.receptive_fields
    idx = synthetic_indices(B, T, cfg["input_channels"], int(fx["idx_seed"]))
    g = _gen(cfg, sd, B, T)
    choices, logits = g.teacher_forced(idx.to(DEV), logits_t0=rf)
    want = np.transpose(fx["logits"][:, :, :-1], (0, 2, 1))  # (B, T-rf, Q): predicts times rf..T-1
    assert rel_err(logits.cpu().numpy(), want) < LOGIT_TOL
    assert np.array_equal(choices[:, rf:].cpu().numpy(), want.argmax(2))


@pytest.mark.parametrize("which", ["generic", "stream", "pipe", "fold"])
def test_l30_teacher_forced_logits(golden, which):
    from movenet_amd import _native as N
    fx = golden("g2_l30_forward.npz")
    cfg, dims, sd = weights_of(fx)
    B, T, rf = int(fx["B"]), int(fx["T"]), dims.receptive_fields
    idx = synthetic_indices(B, T, 256, int(fx["idx_seed"]))
    g = _gen(cfg, sd, B, T, variant={"generic": N.GEN_GENERIC, "stream": N.GEN_STREAM,
                                     "pipe": N.GEN_PIPE, "fold": N.GEN_FOLD}[which])
    choices, logits = g.teacher_forced(idx.to(DEV), logits_t0=rf)
    g.check_errors()
    want = np.transpose(fx["logits"][:, :, :-1], (0, 2, 1))
    assert rel_err(logits.cpu().numpy(), want) < LOGIT_TOL


@pytest.mark.parametrize("name", ["g3_small_greedy.npz", "g3_l30_greedy.npz"])
def test_greedy_free_running_indices_bit_exact(golden, name):
    fx = golden(name)
    cfg, dims, sd = weights_of(fx)
    B, N_, rf, Q = int(fx["B"]), int(fx["N"]), dims.receptive_fields, cfg["input_channels"]
    pidx = synthetic_indices(B, rf, Q, int(fx["prompt_seed"]))
    for variant in _variants(cfg):
        g = _gen(cfg, sd, B, N_, variant=variant)
        g.prime(pidx.to(DEV))
        g.advance(N_ - rf)
        g.check_errors()
        assert np.array_equal(g.samples.cpu().numpy(), fx["indices"]), f"variant {variant}"
        # queues primed by stepping (not by the forward kernels) give the same run
        g1 = _gen(cfg, sd, B, N_, variant=variant)
        g1.prime_with_forward = False
        g1.prime(pidx.to(DEV))
        g1.advance(N_ - rf)
        g1.check_errors()
        assert np.array_equal(g1.samples.cpu().numpy(), fx["indices"]), f"variant {variant} (stepped)"
        # chunked launches carry the queues across calls
        g2 = _gen(cfg, sd, B, N_, variant=variant)
        g2.prime(pidx.to(DEV))
        for _ in range(0, N_ - rf, 7):
            g2.advance(7)
        assert np.array_equal(g2.samples.cpu().numpy(), fx["indices"])


@pytest.mark.parametrize("bias", ["all_equal", "two_tied", "clear_winner"])
def test_greedy_ties_take_the_first_maximum(bias):
    """Greedy decoding returns the FIRST arg-max of softmax(softmax(logits)) (movenet/wavenet.py:227-233,
    torch.argmax).  The pipelined heads pick by one (value, index, runner-up) reduction when the top logit
    leads by >= 1e-3 and run the full double softmax otherwise: with the head's last conv zeroed the logits
    are its bias, so exact ties (all classes equal; two classes sharing the maximum) exercise the
    second form and a clear winner the first, on every variant."""
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=3)
    sd["dense_conv.conv2.weight"] = torch.zeros_like(sd["dense_conv.conv2.weight"])
    b = torch.zeros(256)
    if bias == "two_tied":
        b[9] = b[200] = 1.25
        want = 9
    elif bias == "clear_winner":
        b[9], b[200] = 1.25, 1.5
        want = 200
    else:
        want = 0
    sd["dense_conv.conv2.bias"] = b
    rf = O.Dims(**cfg).receptive_fields
    pidx = synthetic_indices(2, rf, 256, 11)
    for variant in _variants(cfg):
        g = _gen(cfg, sd, 2, rf + 40, variant=variant)
        g.prime(pidx.to(DEV))
        g.advance(40)
        g.check_errors()
        got = g.samples[:, rf:].cpu().numpy()
        assert np.all(got == want), (variant, bias, np.unique(got))


def test_model_api_generate_matches_reference(golden):
    """WaveNet.generate on one-hot input == the reference's one-hot output."""
    from movenet_amd.wavenet import WaveNet
    fx = golden("g3_small_greedy.npz")
    cfg, dims, sd = weights_of(fx)
    Q, rf, N_ = cfg["input_channels"], dims.receptive_fields, int(fx["N"])
    model = WaveNet(**cfg)
    missing = model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    prompt = one_hot(synthetic_indices(int(fx["B"]), rf, Q, int(fx["prompt_seed"])), Q).to(DEV)
    out = model.generate(prompt, n_samples=N_, temperature=0.0)
    assert out.shape == (int(fx["B"]), Q, N_) and out.dtype == prompt.dtype and out.is_cuda
    assert not model.training  # Q10 side effect kept
    assert torch.equal(out.sum(1), torch.ones_like(out.sum(1)))
    assert np.array_equal(out.argmax(1).cpu().numpy(), fx["indices"])
    assert torch.equal(out[:, :, :rf], prompt)
    # default n_samples: the prompt's own length -> nothing to generate beyond it
    longer = one_hot(synthetic_indices(2, rf + 5, Q, 5), Q).to(DEV)
    out2 = model.generate(longer, temperature=0.0)
    assert out2.shape == longer.shape and torch.equal(out2[:, :, :rf], longer[:, :, :rf])


def test_presampling_distribution_and_determinism(golden):
    """temperature > 0: the sampler draws from softmax(softmax(x)/T) (G5)."""
    fx = golden("g5_small_presampling.npz")
    cfg, dims, sd = weights_of(fx)
    Q, rf = cfg["input_channels"], dims.receptive_fields
    p2 = fx["p2_T0_5"][0, :, 0].astype(np.float64)  # sequence 0's distribution at T=0.5
    pidx = synthetic_indices(int(fx["B"]), rf, Q, int(fx["prompt_seed"]))[0:1]
    B = 32768
    prompt = pidx.repeat(B, 1).to(DEV)
    draws = []
    for seed in (1, 1, 2):
        g = _gen(cfg, sd, B, rf + 1, temperature=0.5, seed=seed)
        g.prime(prompt)
        g.advance(1)
        draws.append(g.samples[:, rf].cpu().numpy())
    assert np.array_equal(draws[0], draws[1])       # same seed, same draws
    assert not np.array_equal(draws[0], draws[2])   # other seed, other draws
    freq = np.bincount(draws[0], minlength=Q) / B
    # per-class standard error ~ sqrt(p/B) ~ 7e-4; allow 6 sigma
    assert np.abs(freq - p2).max() < 6 * np.sqrt(p2.max() / B)
    assert abs(freq.sum() - 1) < 1e-12


def test_config2_full_size_properties():
    """BASELINE config 2 (L=30, Q=256, C=K=64, B=16): both kernel variants agree
    on a free run, and re-feeding the result teacher-forced reproduces it."""
    from movenet_amd import _native as N
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=1, gain=2.0, head_gain=6.0)
    rf, B, n_new = 3072, 16, 96
    pidx = synthetic_indices(B, rf, 256, 1234).to(DEV)
    runs = {}
    for variant in (N.GEN_GENERIC, N.GEN_STREAM, N.GEN_PIPE, N.GEN_FOLD):
        g = _gen(cfg, sd, B, rf + n_new, variant=variant)
        g.prime(pidx)
        g.advance(n_new)
        g.check_errors()
        runs[variant] = g.samples.clone()
    assert torch.equal(runs[N.GEN_GENERIC], runs[N.GEN_STREAM])
    assert torch.equal(runs[N.GEN_PIPE], runs[N.GEN_STREAM])
    assert torch.equal(runs[N.GEN_FOLD], runs[N.GEN_STREAM])
    assert _gen(cfg, sd, B, rf + 1).variant == N.GEN_FOLD  # what AUTO runs at config 2, batch 16
    g = _gen(cfg, sd, B, rf + n_new, variant=N.GEN_STREAM)
    choices, logits = g.teacher_forced(runs[N.GEN_STREAM], logits_t0=rf)
    assert torch.equal(choices[:, rf:], runs[N.GEN_STREAM][:, rf:])
    assert len(torch.unique(runs[N.GEN_STREAM][:, rf:])) > 8  # not a degenerate constant output
    # spot-check three sequences against the CPU ring oracle
    dims = O.Dims(**cfg)
    sub = [0, 7, 15]
    ridx, _ = O.generate_ring(sd, dims, pidx[sub].cpu().numpy(), rf + n_new)
    assert np.array_equal(ridx, runs[N.GEN_STREAM][sub].cpu().numpy())


def test_config2_pipe_at_full_occupancy():
    """24 sequences = 3 nine-stage pipelines in each of the 8 XCDs (the most that are
    co-resident): PIPE still equals STREAM, and one more sequence is refused for PIPE.  FOLD: 16
    eleven-stage pipelines of up to 8 sequences each (r3)."""
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=5, gain=2.0, head_gain=6.0)
    rf, B, n_new = 3072, 24, 40
    pidx = synthetic_indices(B, rf, 256, 77).to(DEV)
    runs = {}
    for variant in (N.GEN_STREAM, N.GEN_PIPE):
        g = _gen(cfg, sd, B, rf + n_new, variant=variant)
        g.prime(pidx)
        g.advance(n_new)
        g.check_errors()
        runs[variant] = g.samples.clone()
    assert torch.equal(runs[N.GEN_PIPE], runs[N.GEN_STREAM])
    assert _gen(cfg, sd, 185, rf + 1).variant == N.GEN_PIPE    # AUTO at the C level: 24 pipelines x 8 rounds
    assert _gen(cfg, sd, 193, rf + 1).variant == N.GEN_STREAM  # ... and falls back beyond
    assert _gen(cfg, sd, 24, rf + 1).variant == N.GEN_FOLD     # AUTO: FOLD wherever it holds the batch
    assert _gen(cfg, sd, 184, rf + 1, variant=N.GEN_FOLD).variant == N.GEN_FOLD  # 23 pipelines x 8 rounds
    with pytest.raises(Exception):
        _gen(cfg, sd, 185, rf + 1, variant=N.GEN_FOLD)
    with pytest.raises(Exception):
        _gen(cfg, sd, 193, rf + 1, variant=N.GEN_PIPE)  # 24 pipelines x 8 rounds


def test_grouped_pipelines_beyond_one_launch():
    """40 config-2 sequences = groups of 24 + 16 taking turns on the pipelines: greedy output
    equals one STREAM launch of all 40, and WaveNet.generate picks the grouped form."""
    from movenet_amd.generation import GroupedGenerator, max_pipe_batch
    from movenet_amd.utils.weights import make_state_dict
    from movenet_amd.wavenet import WaveNet
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=6, gain=2.0, head_gain=6.0)
    rf, B, n_new = 3072, 40, 24
    assert max_pipe_batch(N.make_dims(10, 3, 256, 64, 64)) == 192  # 24 pipelines x 8 rounds
    pidx = synthetic_indices(B, rf, 256, 99).to(DEV)
    ref = _gen(cfg, sd, B, rf + n_new, variant=N.GEN_STREAM)
    ref.prime(pidx)
    ref.advance(n_new)
    sdd = {k: v.to(DEV) for k, v in sd.items()}
    g = GroupedGenerator(**cfg, state_dict=sdd, batch=B, n_total=rf + n_new, device=DEV, group=24)
    assert [b - a for a, b in g.bounds] == [24, 16]
    g.prime(pidx)
    g.advance(n_new)
    g.check_errors()
    assert torch.equal(g.samples, ref.samples)
    model = WaveNet(**cfg)
    model.load_state_dict(sd, strict=False)
    out = model.to(DEV).generate(one_hot(pidx.cpu(), 256).to(DEV), n_samples=rf + n_new, temperature=0.0)
    assert torch.equal(out.argmax(1).to(torch.int32), ref.samples)


@pytest.mark.parametrize("B", [40, 128, 184])
def test_fold_pipelines_serve_several_sequences_in_turn(B):
    """r3: beyond 16 sequences a FOLD pipeline serves ceil(B / 16) sequences in turn within ONE
    launch (gen_fold_kernel<true>: weights shared, one inbox per sequence and stage).  B = 40 leaves
    the last round half empty; beyond 80 sequences the launch also uses the seven pipelines that the XCDs' left-over
    CUs form ACROSS XCDs (23 in all: B = 128 is six rounds with the last one partial, B = 184 the full eight on
    every pipeline).  Greedy indices bit-equal to the
    STREAM kernel, chunked launches equal to one launch, teacher-forced logits within tolerance of
    STREAM's, and sampled draws -- the Philox counter is (seed, step, SEQUENCE), whatever the
    round -- equal to STREAM's on >= 99.9 % of the draws."""
    from movenet_amd.utils.weights import make_state_dict
    sd = make_state_dict(**CFG2, seed=6, gain=2.0, head_gain=6.0)
    rf, n_new = 3072, (24 if B == 40 else 700 if B == 128 else 60)  # (the long run crosses every dilation's queue wrap: 512 steps)
    pidx = synthetic_indices(B, rf, 256, 99).to(DEV)
    ref = _gen(CFG2, sd, B, rf + n_new, variant=N.GEN_STREAM)
    ref.prime(pidx)
    ref.advance(n_new)
    g = _gen(CFG2, sd, B, rf + n_new, variant=N.GEN_FOLD)
    assert g.variant == N.GEN_FOLD
    g.prime(pidx)
    g.advance(n_new)
    g.check_errors()
    assert torch.equal(g.samples, ref.samples)
    assert len(torch.unique(g.samples[:, rf:])) > 8
    g2 = _gen(CFG2, sd, B, rf + n_new, variant=N.GEN_FOLD)
    g2.prime(pidx)
    g2.advance(10)
    g2.advance(n_new - 10)
    g2.check_errors()
    assert torch.equal(g2.samples, ref.samples)
    hist = synthetic_indices(B, rf + 40, 256, 4321).to(DEV)
    picks, logits = {}, {}
    for variant in (N.GEN_STREAM, N.GEN_FOLD):
        gt = _gen(CFG2, sd, B, rf + 40, variant=variant, temperature=1.0, seed=77)
        choices, lg = gt.teacher_forced(hist, logits_t0=rf)
        gt.check_errors()
        picks[variant], logits[variant] = choices[:, rf:].cpu().numpy(), lg.cpu().numpy()
    assert rel_err(logits[N.GEN_FOLD], logits[N.GEN_STREAM]) < LOGIT_TOL
    assert (picks[N.GEN_FOLD] == picks[N.GEN_STREAM]).mean() >= 0.999


@pytest.mark.parametrize("shape,B,n_new", [("c64", 40, 24), ("c128", 22, 600), ("c128", 64, 30)])
def test_pipe_pipelines_serve_several_sequences_in_turn(shape, B, n_new):
    """r3: the fp32 PIPE kernel serves several sequences per pipeline too (gen_pipe_kernel<C, true>: 24
    nine-stage pipelines of up to 8 sequences at C = 64, 4 sixty-one-stage pipelines of up to 16 at
    C = K = 128 -- config 5's batch of 16 in ONE launch).  A sequence's arithmetic is the same whatever
    its pipeline's round count, so teacher-forced logits are BIT-equal to launches that hold one sequence
    per pipeline; greedy runs equal the one-launch kernels' (STREAM / GENERIC), chunked launches equal
    one launch (the long run crosses the 512-step queue wrap), and sampled draws (Philox counter: seed,
    step, SEQUENCE) equal the one-launch kernel's on >= 99.5 % of the draws."""
    from movenet_amd.generation import max_pipe_batch
    from movenet_amd.utils.weights import make_state_dict
    if shape == "c64":
        cfg, other, per_launch = dict(CFG2), N.GEN_STREAM, 24
    else:
        cfg = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)
        other, per_launch = N.GEN_GENERIC, 4
    sd = make_state_dict(**cfg, seed=6, gain=2.0 if shape == "c64" else 1.5, head_gain=6.0)
    dims = O.Dims(**cfg)
    rf = dims.receptive_fields
    assert max_pipe_batch(N.make_dims(cfg["layer_size"], cfg["stack_size"], 256, cfg["residual_channels"],
                                      cfg["skip_channels"])) == (192 if shape == "c64" else 64)
    pidx = synthetic_indices(B, rf, 256, 99).to(DEV)
    ref = _gen(cfg, sd, B, rf + n_new, variant=other)
    ref.prime(pidx)
    ref.advance(n_new)
    g = _gen(cfg, sd, B, rf + n_new, variant=N.GEN_PIPE)
    assert g.variant == N.GEN_PIPE
    g.prime(pidx)
    g.advance(n_new)
    g.check_errors()
    assert torch.equal(g.samples, ref.samples)
    assert len(torch.unique(g.samples[:, rf:])) > 4
    g2 = _gen(cfg, sd, B, rf + n_new, variant=N.GEN_PIPE)
    g2.prime(pidx)
    g2.advance(10)
    g2.advance(n_new - 10)
    g2.check_errors()
    assert torch.equal(g2.samples, ref.samples)
    hist = synthetic_indices(B, rf + 24, 256, 4321).to(DEV)
    picks, logits = {}, {}
    for variant in (other, N.GEN_PIPE):
        gt = _gen(cfg, sd, B, rf + 24, variant=variant, temperature=1.0, seed=77)
        choices, lg = gt.teacher_forced(hist, logits_t0=rf)
        gt.check_errors()
        picks[variant], logits[variant] = choices[:, rf:].cpu().numpy(), lg.cpu()
    assert rel_err(logits[N.GEN_PIPE].numpy(), logits[other].numpy()) < LOGIT_TOL
    # (a draw differs only where the uniform falls within the two kernels' rounding distance of a step
    # of the CDF: ~1e-3 of the draws at C = 128; a wrong Philox counter would change nearly all of them)
    assert (picks[N.GEN_PIPE] == picks[other]).mean() >= 0.995
    for lo in range(0, B, per_launch):  # one sequence per pipeline: the same arithmetic, bit for bit
        hi = min(B, lo + per_launch)
        gs = _gen(cfg, sd, hi - lo, rf + 24, variant=N.GEN_PIPE)
        _, lg = gs.teacher_forced(hist[lo:hi].contiguous(), logits_t0=rf)
        gs.check_errors()
        assert torch.equal(lg.cpu(), logits[N.GEN_PIPE][lo:hi]), (lo, hi)


def test_bad_input_raises():
    from movenet_amd.wavenet import WaveNet
    model = WaveNet(2, 2, 64, 16, 16).to(DEV)
    bad = torch.rand(1, 64, 20, device=DEV)
    with pytest.raises(ValueError):
        model.generate(bad, n_samples=30, temperature=0.0)
    with pytest.raises(RuntimeError):
        model.generate(torch.zeros(1, 64, 20), n_samples=30)


@pytest.mark.parametrize("variant", [N.GEN_GENERIC, N.GEN_PIPE])
def test_config5_shape_generate_vs_oracle(variant):
    """BASELINE config 5's model (60 layers, C=K=128, RF=6144), fp32: greedy free run ==
    the CPU ring oracle, logits within tolerance of it -- through the GENERIC kernel and
    through the 61-stage PIPE kernel (one layer per CU, two XCDs per sequence)."""
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)
    sd = make_state_dict(**cfg, seed=2, gain=1.5, head_gain=6.0)
    dims = O.Dims(**cfg)
    rf, n_new, B = dims.receptive_fields, 12, 1
    pidx = synthetic_indices(B, rf, 256, 5)
    want, want_logits = O.generate_ring(sd, dims, pidx.numpy(), rf + n_new)
    g = _gen(cfg, sd, B, rf + n_new, variant=variant)
    assert g.variant == variant
    g.prime(pidx.to(DEV))
    g.advance(n_new)
    g.check_errors()
    assert np.array_equal(g.samples.cpu().numpy(), want)
    _, logits = g.teacher_forced(torch.from_numpy(want).to(DEV), logits_t0=rf)
    assert rel_err(logits.cpu().numpy(), want_logits) < LOGIT_TOL


def test_config5_pipe_matches_generic_over_a_long_run():
    """Four config-5 sequences (the most the 61-stage pipelines fit), ring priming without
    the forward kernels, 300 greedy steps: PIPE and GENERIC choose identical samples."""
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)
    sd = make_state_dict(**cfg, seed=4, gain=1.5, head_gain=6.0)
    rf, n_new, B = O.Dims(**cfg).receptive_fields, 300, 4
    pidx = synthetic_indices(B, rf, 256, 9).to(DEV)
    runs = {}
    for variant in (N.GEN_GENERIC, N.GEN_PIPE):
        g = _gen(cfg, sd, B, rf + n_new, variant=variant)
        g.prime(pidx)
        g.advance(n_new)
        g.check_errors()
        runs[variant] = g.samples.clone()
    assert torch.equal(runs[N.GEN_PIPE], runs[N.GEN_GENERIC])
    assert len(torch.unique(runs[N.GEN_PIPE][:, rf:])) > 4


def test_integration_stub_runs():
    """The ctypes stub INTEGRATION.md hands to a movenet maintainer, executed as written
    (against this build's WaveNet, which has the reference's attribute names): its greedy
    output equals WaveNet.generate's."""
    import os
    import re
    from movenet_amd.utils.weights import make_state_dict
    from movenet_amd.wavenet import WaveNet
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\nimport ctypes as C, torch\n(.*?)```", text, re.S).group(0)
    code = code[len("```python\n"):-3].replace(
        'C.CDLL("libmovenet_hip.so")', f'C.CDLL({os.path.join(root, "movenet_amd", "lib", "libmovenet_hip.so")!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    cfg = dict(layer_size=3, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)
    model = WaveNet(**cfg)
    model.load_state_dict(make_state_dict(**cfg, seed=3, gain=2.0, head_gain=6.0), strict=False)
    model.to(DEV)
    rf = model.receptive_fields
    prompt = one_hot(synthetic_indices(2, rf, 64, 11), 64).to(DEV)
    want = model.generate(prompt, n_samples=rf + 20, temperature=0.0)
    got = ns["fast_generate"](model, prompt, rf + 20, temperature=0.0)
    assert torch.equal(got, want)


# ---- temperature > 0 (the reference's DEFAULT, movenet/wavenet.py:200, :227-231) on the
# kernels config 2 actually runs, and the PIPE status word -------------------------------
CFG2 = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)


@pytest.mark.parametrize("temperature", [0.5, 1.0])
def test_sampler_same_draws_on_generic_stream_pipe(temperature):
    """All three kernels draw with the Philox counter (seed, u, b): on one teacher-forced
    history (so that a differing draw cannot change later inputs) they must pick the same
    class on >= 99.9 % of 11 200 draws -- a difference is only possible where the uniform
    falls within rounding of a CDF edge."""
    from movenet_amd.utils.weights import make_state_dict
    sd = make_state_dict(**CFG2, seed=3, gain=2.0, head_gain=6.0)
    rf, B, n_new = 3072, 16, 700
    hist = synthetic_indices(B, rf + n_new, 256, 4321).to(DEV)
    picks = {}
    for variant in (N.GEN_GENERIC, N.GEN_STREAM, N.GEN_PIPE, N.GEN_FOLD):
        g = _gen(CFG2, sd, B, rf + n_new, variant=variant, temperature=temperature, seed=77)
        choices, _ = g.teacher_forced(hist, logits_t0=rf)
        g.check_errors()
        picks[variant] = choices[:, rf:].cpu().numpy()
        assert picks[variant].min() >= 0 and picks[variant].max() < 256
    n = picks[N.GEN_GENERIC].size
    assert n >= 10000
    for a, b in ((N.GEN_GENERIC, N.GEN_STREAM), (N.GEN_GENERIC, N.GEN_PIPE), (N.GEN_STREAM, N.GEN_PIPE),
                 (N.GEN_GENERIC, N.GEN_FOLD), (N.GEN_PIPE, N.GEN_FOLD)):
        same = (picks[a] == picks[b]).mean()
        assert same >= 0.999, f"variants {a}/{b} agree on {same:.5f} of {n} draws"
    # the draws are samples, not the arg-max: many distinct classes, and another seed differs
    assert len(np.unique(picks[N.GEN_PIPE])) > 32
    g2 = _gen(CFG2, sd, B, rf + n_new, variant=N.GEN_PIPE, temperature=temperature, seed=78)
    other, _ = g2.teacher_forced(hist, logits_t0=rf)
    assert (other[:, rf:].cpu().numpy() != picks[N.GEN_PIPE]).mean() > 0.5


@pytest.mark.parametrize("variant", [N.GEN_PIPE, N.GEN_STREAM, N.GEN_FOLD])
def test_sampler_frequencies_match_oracle_distribution(variant):
    """>= 32 768 draws of ONE config-2 step (24 or 16 identical sequences x seeds) against the
    oracle's pre-sampling distribution softmax(softmax(logits) / T) for that step."""
    from movenet_amd.utils.weights import make_state_dict
    sd = make_state_dict(**CFG2, seed=3, gain=2.0, head_gain=6.0)
    dims = O.Dims(**CFG2)
    rf, T = 3072, 0.5
    B = 16 if variant == N.GEN_FOLD else 24   # the most sequences one launch of the variant holds
    n_seeds = -(-32768 // B)
    pidx = synthetic_indices(1, rf, 256, 555)
    with torch.no_grad():
        probs = O.forward(sd, dims, one_hot(pidx, 256), output_unnormalized=True, remove_last=False)
        p2 = O.pre_sampling_probs(probs, T)[0, :, 0].double().numpy()
    g = _gen(CFG2, sd, B, rf + 1, variant=variant, temperature=T, seed=0)
    g.prime(pidx.repeat(B, 1).to(DEV))
    state0, samples0, t0 = g.state.clone(), g.samples.clone(), g.t
    draws = []
    for seed in range(n_seeds):
        g.state.copy_(state0)
        g.samples.copy_(samples0)
        g.t, g.seed = t0, seed
        g.advance(1)
        draws.append(g.samples[:, rf].clone())
    g.check_errors()
    draws = torch.cat(draws).cpu().numpy()
    n = draws.size
    assert n == B * n_seeds >= 32768
    freq = np.bincount(draws, minlength=256) / n
    # per-class standard error sqrt(p (1 - p) / n); allow 6 sigma on every class
    assert (np.abs(freq - p2) < 6 * np.sqrt(p2 * (1 - p2) / n) + 1e-9).all()
    assert p2.max() > 2 * p2.min()  # not the near-uniform distribution of unsharpened weights


@pytest.mark.parametrize("variant", [N.GEN_PIPE, N.GEN_FOLD])
def test_sampler_pipe_chunked_launches_same_as_one_launch(variant):
    """T = 1.0 free run on PIPE: the draw for (b, u) does not depend on how the steps are
    partitioned into launches, nor on the queues being primed by the forward kernels."""
    from movenet_amd.utils.weights import make_state_dict
    sd = make_state_dict(**CFG2, seed=3, gain=2.0, head_gain=6.0)
    rf, B, n_new = 3072, 16, 60
    pidx = synthetic_indices(B, rf, 256, 99).to(DEV)
    runs = []
    for chunk in (n_new, 7, 1):
        g = _gen(CFG2, sd, B, rf + n_new, variant=variant, temperature=1.0, seed=5)
        g.prime(pidx)
        for _ in range(0, n_new, chunk):
            g.advance(chunk)
        g.check_errors()
        runs.append(g.samples.clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    s = _gen(CFG2, sd, B, rf + n_new, variant=N.GEN_STREAM, temperature=1.0, seed=5)
    s.prime(pidx)
    s.advance(n_new)
    # free-running: one differing draw changes the rest of that sequence, so compare the
    # first 8 draws of every sequence (the teacher-forced test above covers the long run)
    assert torch.equal(s.samples[:, :rf + 8], runs[0][:, :rf + 8])


@pytest.mark.parametrize("variant", [N.GEN_PIPE, N.GEN_FOLD])
def test_pipe_status_word_is_sticky_and_checked(variant):
    """A raised hand-off status word (here: raised by the test) makes every later launch a
    no-op until the state is reset, and check_errors() raises: a time-out in one advance()
    chunk is not erased by the next launch."""
    from movenet_amd.generation import PipeHandoffTimeout
    from movenet_amd.utils.weights import make_state_dict
    sd = make_state_dict(**CFG2, seed=1, gain=2.0, head_gain=6.0)
    rf, B, n_new = 3072, 4, 16
    pidx = synthetic_indices(B, rf, 256, 3).to(DEV)
    g = _gen(CFG2, sd, B, rf + n_new, variant=variant)
    g.prime(pidx)
    g.advance(4)
    g.check_errors()
    good = g.samples.clone()
    g.status_word().fill_(1)
    g.advance(4)
    with pytest.raises(PipeHandoffTimeout):
        g.check_errors()
    assert torch.equal(g.samples, good)  # the launch did nothing
    g.advance(4)                          # ... and the word survives the next launch
    with pytest.raises(PipeHandoffTimeout):
        g.check_errors()
    g.prime(pidx)                         # reset() clears it
    g.advance(n_new)
    g.check_errors()
    ref = _gen(CFG2, sd, B, rf + n_new, variant=N.GEN_STREAM)
    ref.prime(pidx)
    ref.advance(n_new)
    assert torch.equal(g.samples, ref.samples)


def test_conditioned_stream_matches_generic_and_fold():
    """r4: gen_stream64_kernel with local conditioning (the context terms of all layers formed at the top of a step):
    greedy samples bit-equal to GENERIC and FOLD on a conditioned run, teacher-forced logits within 2e-5 of GENERIC's."""
    from movenet_amd.utils.weights import make_state_dict
    sd = make_state_dict(**CFG2, seed=1, gain=2.0, head_gain=6.0)
    rf, B, n_new = 3072, 3, 40
    prompt = synthetic_indices(B, rf, 256, 5)
    ctx = torch.from_numpy(np.random.default_rng(9).standard_normal((B, 64, rf + n_new)).astype(np.float32)).to(DEV)
    runs = {}
    for variant in (N.GEN_GENERIC, N.GEN_STREAM, N.GEN_FOLD):
        from movenet_amd.generation import RingGenerator
        g = RingGenerator(**CFG2, state_dict={k: v.to(DEV) for k, v in sd.items()}, batch=B, n_total=rf + n_new, device=DEV,
                          variant=variant, temperature=0.0, context=ctx)
        assert g.variant == variant
        g.prime(prompt.to(DEV))
        g.advance(n_new)
        g.check_errors()
        samples = g.samples.clone()
        _, logits = g.teacher_forced(samples, logits_t0=rf)
        runs[variant] = (samples, logits)
    ref_s, ref_l = runs[N.GEN_GENERIC]
    for variant in (N.GEN_STREAM, N.GEN_FOLD):
        assert torch.equal(runs[variant][0], ref_s), variant
        err = float((runs[variant][1] - ref_l).abs().max() / ref_l.abs().max())
        assert err < 2e-5, (variant, err)
    # conditioning matters on this run: without it the samples differ
    g0 = RingGenerator(**CFG2, state_dict={k: v.to(DEV) for k, v in sd.items() if ".context_conv_" not in k}, batch=B,
                       n_total=rf + n_new, device=DEV, variant=N.GEN_STREAM, temperature=0.0)
    g0.prime(prompt.to(DEV))
    g0.advance(n_new)
    assert not torch.equal(g0.samples, ref_s)


@pytest.mark.parametrize("B,with_video", [(3, False), (20, False), (3, True)])
def test_model_generate_reruns_on_pipe_timeout(monkeypatch, B, with_video):
    """WaveNet.generate never returns unchecked samples: with the PIPE status word raised
    during the call it reruns the same call on STREAM in the same process.  B = 20 is a MULTI
    launch (16 pipelines, the first four serving two sequences in turn): the starved path with
    several sequences per pipeline."""
    from movenet_amd import generation as G
    from movenet_amd.utils.weights import make_state_dict
    from movenet_amd.wavenet import WaveNet
    sd = make_state_dict(**CFG2, seed=1, gain=2.0, head_gain=6.0)
    model = WaveNet(**CFG2)
    model.load_state_dict(sd, strict=False)
    model.to(DEV)
    rf, n_new = 3072, 20
    prompt = one_hot(synthetic_indices(B, rf, 256, 8), 256).to(DEV)
    video = None
    if with_video:  # (r4: a conditioned call is rerun on STREAM too, not on GENERIC; Q8: 4 frames <-> 4000 samples)
        import movenet_amd.wavenet as W
        monkeypatch.setattr(W, "MAX_AUDIO_FRAMES", 4000)
        monkeypatch.setattr(W, "MAX_VIDEO_FRAMES", 4)
        video = torch.from_numpy(np.random.default_rng(3).random((B, 4, 64, 64, 1), dtype=np.float32)).to(DEV)
    _generate = model.generate
    model.generate = lambda p, **kw: _generate(p, video, **kw)
    want = model.generate(prompt, n_samples=rf + n_new, temperature=0.0)
    assert model.last_generate_fallback is None
    assert N.lib().mvn_gen_launch_is_cooperative() == 1  # no profiler attached: the runtime guarantees co-residency
    real_advance, poked = G.RingGenerator.advance, []

    def advance(self, n):
        if self.variant in N.PIPE_VARIANTS:
            self.status_word().fill_(1)
            poked.append(self.variant)
        return real_advance(self, n)

    monkeypatch.setattr(G.RingGenerator, "advance", advance)
    got = model.generate(prompt, n_samples=rf + n_new, temperature=0.0)
    assert poked == [N.GEN_FOLD] and model.last_generate_fallback == N.GEN_STREAM
    assert torch.equal(got, want)
    # a variant forced by the caller that cannot be rerun differently still never returns
    # silently: the generator's own check raises
    g = _gen(CFG2, sd, B, rf + n_new, variant=N.GEN_PIPE)
    g.prime(synthetic_indices(B, rf, 256, 8).to(DEV))
    g.advance(n_new)
    with pytest.raises(G.PipeHandoffTimeout):
        g.check_errors()


def test_auto_plan_cost_based():
    from movenet_amd.generation import auto_plan as _plan, calibrate
    # the tables as measured on the reference box (the per-device calibration is checked at the end)
    TABLE = {"pipelined": 1.0, "single": 1.0}

    def auto_plan(d, n, c):
        return _plan(d, n, c, calibration=TABLE)

    d2, d5 = N.make_dims(10, 3, 256, 64, 64), N.make_dims(10, 6, 256, 128, 128)
    for n in (1, 16, 20, 32, 64, 128, 161, 184):                       # one FOLD launch: 16 pipelines, 23 beyond 80 sequences
        assert auto_plan(d2, n, False) == ("single", 0, N.GEN_FOLD)
    assert auto_plan(d2, 144, True) == ("single", 0, N.GEN_FOLD)
    assert auto_plan(d2, 190, False) == ("grouped", 95, N.GEN_FOLD)   # 2 x 16.4 us < 34.4 (PIPE, 8 rounds)
    assert auto_plan(d2, 256, False) == ("grouped", 128, N.GEN_FOLD)  # 2 x 17.7 us
    assert auto_plan(d2, 400, False) == ("grouped", 134, N.GEN_FOLD)  # 3 x 17.7 us
    assert auto_plan(d2, 500, False) == ("grouped", 167, N.GEN_FOLD)  # 3 x 23.6 us < 78 us
    assert auto_plan(d2, 600, False) == ("single", 0, N.GEN_STREAM)   # 4 x 20.7 us > 78 us
    assert auto_plan(d2, 600, True) == ("grouped", 150, N.GEN_FOLD)   # 4 x 20.7 us < 100 us (conditioned STREAM, r4)
    assert auto_plan(d2, 800, True) == ("single", 0, N.GEN_STREAM)    # 5 x 21 us > 100 us
    for n in (1, 4, 5, 16, 24, 64):                                    # one PIPE launch, 1 - 16 rounds: 73 us
        assert auto_plan(d5, n, False) == ("single", 0, N.GEN_PIPE)
    assert auto_plan(d5, 65, False) == ("grouped", 33, N.GEN_PIPE)    # 2 x 73 us < 490 us
    assert auto_plan(d5, 384, False) == ("grouped", 64, N.GEN_PIPE)   # 6 x 73 us
    assert auto_plan(d5, 385, False) == ("single", 0, N.GEN_GENERIC)  # 7 x 73 us > 490 us
    # per-device calibration: both kernel families timed once on THIS device, within a factor of two of the tables,
    # cached, and what auto_plan uses by default
    cal = calibrate(d2)
    assert 0.5 < cal["pipelined"] < 2.0 and 0.5 < cal["single"] < 2.0, cal
    assert calibrate(d2) is cal
    assert _plan(d2, 256, False)[0] == "grouped"


@pytest.mark.parametrize("layer_size,stack_size", [(1, 1), (2, 1), (4, 1), (5, 2), (10, 2), (7, 3)])
def test_pipelined_variants_with_partial_last_stage(layer_size, stack_size):
    """Layer counts that do not fill the last stage (FOLD holds 3 layers per stage, PIPE 4): the
    missing layers are packed as zeros and must act as the identity, and the head must add the
    skip 1x1 of the REAL last layer exactly once.  Greedy runs of the pipelined variants equal
    the generic kernel's; teacher-forced logits agree within the parity tolerance; what AUTO
    picks is one of them."""
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=layer_size, stack_size=stack_size, input_channels=256, residual_channels=64,
               skip_channels=64)
    sd = make_state_dict(**cfg, seed=3, gain=2.0, head_gain=6.0)
    rf = sum(2 ** (l % layer_size) for l in range(layer_size * stack_size)) + 2
    B, n_new = 3, 40
    pidx = synthetic_indices(B, rf, 256, 21).to(DEV)
    runs, logits, picks = {}, {}, {}
    for variant in (N.GEN_GENERIC, N.GEN_PIPE, N.GEN_FOLD):
        g = _gen(cfg, sd, B, rf + n_new, variant=variant)
        assert g.variant == variant
        g.prime(pidx)
        g.advance(n_new)
        g.check_errors()
        runs[variant] = g.samples.clone()
        g2 = _gen(cfg, sd, B, rf + n_new, variant=variant)
        ch, lg = g2.teacher_forced(runs[N.GEN_GENERIC], logits_t0=rf)
        g2.check_errors()
        logits[variant], picks[variant] = lg.cpu().numpy(), ch[:, rf:].cpu().numpy()
    ref = logits[N.GEN_GENERIC]
    top2 = np.sort(ref, axis=2)[:, :, -2:]
    clear = (top2[:, :, 1] - top2[:, :, 0]) > 4 * LOGIT_TOL * np.abs(ref).max()   # (B, n_new)
    assert clear.mean() > 0.5
    for variant in (N.GEN_PIPE, N.GEN_FOLD):
        assert rel_err(logits[variant], ref) < LOGIT_TOL, variant
        # same history => same pick wherever the winner is clear of the parity tolerance
        same = picks[variant] == picks[N.GEN_GENERIC]
        assert same[clear[:, :same.shape[1]]].all(), variant
        # ... and the free runs are equal up to their first unclear step
        eq = (runs[variant] == runs[N.GEN_GENERIC])[:, rf:].cpu().numpy()
        for b in range(B):
            bad = np.flatnonzero(~eq[b])
            assert bad.size == 0 or not clear[b, bad[0]], (variant, b)
    assert _gen(cfg, sd, B, rf + 1).variant == N.GEN_FOLD
