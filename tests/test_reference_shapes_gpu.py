"""GPU: the shapes of the reference's OWN experiments (Q = 128, C = 16 / 32 / 64, layer_size up to 14:
/root/reference/experiments/03_kinetics_scale_up.mk:7-10, :64-67, 04_kinetics_receptive_field.mk:8-11), which are not
BASELINE's Q = 256 / C = 64.  Forward logits, trainer loss and gradients against torch autograd on the oracle, greedy
generation bit-exact against the oracle's ring-buffer stepping, through whatever kernel `mvn_gen_variant(AUTO)` picks
-- a pipelined one for C = K = 64 at Q = 128 (r4: the pipelined / STREAM heads take Q in {64, 128, 256}), the generic
kernels elsewhere, including 14 layers per stack: dilation 8192, receptive field 16 384."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import one_hot, rel_err, synthetic_indices
from movenet_amd import _native as N
from movenet_amd.utils.weights import make_state_dict
from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = {
    "q128_c64_10x3": dict(layer_size=10, stack_size=3, input_channels=128, residual_channels=64, skip_channels=64),
    "q128_c16_14x1": dict(layer_size=14, stack_size=1, input_channels=128, residual_channels=16, skip_channels=16),
    "q128_c32_2x2": dict(layer_size=2, stack_size=2, input_channels=128, residual_channels=32, skip_channels=32),
    "q64_c64_10x3": dict(layer_size=10, stack_size=3, input_channels=64, residual_channels=64, skip_channels=64),
}


def _model(cfg, sd):
    from movenet_amd.wavenet import WaveNet
    m = WaveNet(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("name", ["q128_c64_10x3", "q128_c16_14x1", "q128_c32_2x2"])
def test_forward_loss_and_gradients_vs_oracle(name):
    cfg = SHAPES[name]
    dims = O.Dims(**cfg)
    rf, Q = dims.receptive_fields, cfg["input_channels"]
    B, T = 2, rf + 150
    sd = make_state_dict(**cfg, seed=11, gain=1.5)
    x = one_hot(synthetic_indices(B, T, Q, 1234), Q)
    # oracle: logits, then the trainer's arithmetic (cross_entropy ON the probabilities, Q2) and its gradients
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits_o = O.forward(params, dims, x, output_unnormalized=False)
    probs_o = O.forward(params, dims, x)
    target = x[:, :, rf:].argmax(1)
    loss_o = F.cross_entropy(probs_o, target)
    loss_o.backward()
    m = _model(cfg, sd).train()
    logits = m(x.to(DEV), output_unnormalized=False)
    assert logits.shape == logits_o.shape
    assert rel_err(logits.detach().cpu(), logits_o.detach()) < 2e-5
    probs = m(x.to(DEV))
    loss = F.cross_entropy(probs, target.to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 2e-6 * max(1.0, abs(float(loss_o.detach())))
    for k, p in m.named_parameters():
        go = params[k].grad
        if go is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad.cpu(), go) < 3e-4, k


@pytest.mark.parametrize("name,expect_pipelined", [("q128_c64_10x3", True), ("q64_c64_10x3", True),
                                                    ("q128_c16_14x1", False), ("q128_c32_2x2", False)])
def test_greedy_generation_bit_exact_vs_oracle(name, expect_pipelined):
    from movenet_amd.generation import RingGenerator
    cfg = SHAPES[name]
    dims = O.Dims(**cfg)
    rf, Q = dims.receptive_fields, cfg["input_channels"]
    B, n_new = 3, 40
    sd = make_state_dict(**cfg, seed=1, gain=2.0, head_gain=6.0)
    prompt = synthetic_indices(B, rf, Q, 4321)
    want, want_logits = O.generate_ring(sd, dims, prompt.numpy(), rf + n_new)
    picked = N.lib().mvn_gen_variant(N.make_dims(cfg["layer_size"], cfg["stack_size"], Q, cfg["residual_channels"],
                                                 cfg["skip_channels"]), N.GEN_AUTO, B)
    assert (picked in N.PIPE_VARIANTS) == expect_pipelined, picked
    variants = [N.GEN_AUTO, N.GEN_GENERIC] + ([N.GEN_STREAM, N.GEN_PIPE, N.GEN_FOLD] if expect_pipelined else [])
    for variant in variants:
        gen = RingGenerator(**cfg, state_dict={k: v.to(DEV) for k, v in sd.items()}, batch=B, n_total=rf + n_new, device=DEV,
                            variant=variant, temperature=0.0)
        gen.prime(prompt.to(DEV))
        gen.advance(n_new)
        gen.check_errors()
        assert np.array_equal(gen.samples.cpu().numpy(), want), variant
        _, logits = gen.teacher_forced(torch.from_numpy(want).to(DEV), logits_t0=rf)
        assert tuple(logits.shape) == (B, n_new, Q)
        err = np.abs(logits.cpu().numpy() - want_logits).max() / np.abs(want_logits).max()
        assert err < 2e-5, (variant, err)


def test_sampled_generation_small_q_matches_the_double_softmax():
    """temperature > 0 at Q = 128 on the pipelined kernel: the classes a 256-wide head pads with must carry no
    probability in either softmax (generate's quirk Q3: softmax(softmax(logits) / T))."""
    from movenet_amd.generation import RingGenerator
    cfg = SHAPES["q128_c64_10x3"]
    dims = O.Dims(**cfg)
    rf, Q = dims.receptive_fields, 128
    sd = make_state_dict(**cfg, seed=3, gain=2.0, head_gain=40.0)   # a head sharp enough for the double softmax to move
    B, n_new = 64, 33
    prompt = synthetic_indices(1, rf, Q, 7).repeat(B, 1)
    gen = RingGenerator(**cfg, state_dict={k: v.to(DEV) for k, v in sd.items()}, batch=B, n_total=rf + n_new, device=DEV,
                        variant=N.GEN_FOLD, temperature=1.0, seed=5)
    gen.prime(prompt.to(DEV))
    gen.advance(1)
    gen.check_errors()
    first = gen.samples[:, rf].cpu().numpy()
    assert first.min() >= 0 and first.max() < Q
    # every sequence shares the history: the first draw is B independent samples of ONE distribution
    _, lg = O.generate_ring(sd, dims, prompt[:1].numpy(), rf + 1)
    p = torch.softmax(torch.softmax(torch.from_numpy(lg[0, 0]), 0) / 1.0, 0).numpy()
    assert p.shape == (Q,)
    counts = np.bincount(first, minlength=Q).astype(np.float64)
    # a draw outside the support (a padded class leaking mass) or a grossly wrong distribution shows up here
    assert np.all(counts[p < 1e-9] == 0)
    exp = B * p
    chi = ((counts - exp) ** 2 / np.maximum(exp, 1e-9)).sum()
    assert chi < 3.0 * Q, chi
