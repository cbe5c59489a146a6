"""GPU parity of the full-sequence forward/backward (MFMA kernels through the
C ABI, wrapped in movenet_amd.WaveNet.forward) against the golden vectors from
the reference.

Tolerances: logits within 2e-5 of the logit range, probabilities within 2e-6
absolute, loss within 2e-6, gradients within 2e-4 relative to the tensor's
largest entry (fp32 sums over up to B*T = 6400 positions in another order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import cfg_of, one_hot, rel_err, synthetic_indices, weights_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOGIT_TOL = 2e-5


def _model(cfg, sd):
    from movenet_amd.wavenet import WaveNet
    m = WaveNet(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


def test_g1_small_forward_flags_and_shapes(golden):
    fx = golden("g1_small_forward.npz")
    cfg, dims, sd = weights_of(fx)
    Q, T = cfg["input_channels"], int(fx["T"])
    x = one_hot(synthetic_indices(int(fx["B"]), T, Q, int(fx["idx_seed"])), Q).to(DEV)
    m = _model(cfg, sd)
    with torch.no_grad():
        logits = m(x, output_unnormalized=False, remove_last=False)
        probs = m(x)  # default: softmax probabilities, last column dropped (Q1)
        logits_rl = m(x, output_unnormalized=False)
    assert logits.shape == fx["logits"].shape and probs.shape == fx["probs"].shape
    assert rel_err(logits.cpu(), fx["logits"]) < LOGIT_TOL
    assert np.abs(probs.cpu().numpy() - fx["probs"]).max() < 2e-6
    assert torch.equal(logits_rl, logits[:, :, :-1])
    assert torch.allclose(probs.sum(1), torch.ones_like(probs.sum(1)), atol=1e-5)
    with pytest.raises(ValueError, match="receptive"):
        m(x[:, :, :dims.receptive_fields - 1])
    # exactly RF samples: one output column, zero after remove_last
    assert m(x[:, :, :dims.receptive_fields], remove_last=False).shape[2] == 1
    assert m(x[:, :, :dims.receptive_fields]).shape[2] == 0


def test_g2_l30_forward(golden):
    fx = golden("g2_l30_forward.npz")
    cfg, dims, sd = weights_of(fx)
    x = one_hot(synthetic_indices(int(fx["B"]), int(fx["T"]), 256, int(fx["idx_seed"])), 256).to(DEV)
    m = _model(cfg, sd)
    with torch.no_grad():
        logits = m(x, output_unnormalized=False, remove_last=False)
    assert rel_err(logits.cpu(), fx["logits"]) < LOGIT_TOL
    ws = int(fx["window_start"])
    with torch.no_grad():
        win = m(x[:, :, ws:ws + dims.receptive_fields], output_unnormalized=False, remove_last=False)
    assert rel_err(win.cpu(), fx["window_logits"]) < LOGIT_TOL


def test_g6_l60_c128_forward(golden):
    fx = golden("g6_l60_forward.npz")
    cfg, dims, sd = weights_of(fx)
    x = one_hot(synthetic_indices(1, int(fx["T"]), 256, int(fx["idx_seed"])), 256).to(DEV)
    with torch.no_grad():
        logits = _model(cfg, sd)(x, output_unnormalized=False, remove_last=False)
    assert rel_err(logits.cpu(), fx["logits"]) < LOGIT_TOL


@pytest.mark.parametrize("name", ["g4_small_train.npz", "g4_l30_train.npz"])
def test_g4_trainer_arithmetic_and_gradients(golden, name):
    """loss = cross_entropy(PROBABILITIES, target) as in
    pytorch_lightning_trainer.py:62-66 (Q2), gradients via the HIP backward."""
    fx = golden(name)
    cfg, dims, sd = weights_of(fx)
    Q = cfg["input_channels"]
    x = one_hot(synthetic_indices(int(fx["B"]), int(fx["T"]), Q, int(fx["idx_seed"])), Q).to(DEV)
    m = _model(cfg, sd).train()
    from movenet_amd.ops import cross_entropy_on_probs
    out = m(x)
    target = x[:, :, m.receptive_fields:].argmax(1)
    loss, acc = cross_entropy_on_probs(out, target)  # the fused loss + accuracy kernels (row F3)
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) < 2e-6
    assert abs(acc.item() - float(fx["acc"])) < 1e-6
    # ... and they are what torch computes from the same probabilities
    with torch.no_grad():
        assert abs(F.cross_entropy(out, target).item() - loss.item()) < 1e-6
        assert (out.argmax(1) == target).float().mean().item() == acc.item()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in fx["grad_names"]]
    assert sorted(grads) == names  # same set of parameters receives a gradient
    got = np.array([grads[n].norm().item() for n in names])
    assert np.allclose(got, fx["grad_norms"], rtol=2e-4, atol=1e-12), \
        np.abs(got / fx["grad_norms"] - 1).max()
    L = dims.n_layers
    for key, pname in (("grad_causal", "causal_conv.conv.weight"),
                       ("grad_l0_filter", "residual_conv_stack.conv_layers.0.conv_filter.conv.weight"),
                       ("grad_last_skip_w", f"residual_conv_stack.conv_layers.{L - 1}.conv_skip.weight"),
                       ("grad_head2_b", "dense_conv.conv2.bias")):
        assert rel_err(grads[pname].cpu(), fx[key]) < 2e-4, key


def test_forward_priming_equals_step_priming(golden):
    """The dilation queues filled from one full-sequence forward equal those
    built by stepping the generator over the prompt."""
    from movenet_amd.generation import RingGenerator
    fx = golden("g3_l30_greedy.npz")
    cfg, dims, sd = weights_of(fx)
    rf, N_, B = dims.receptive_fields, int(fx["N"]), int(fx["B"])
    sdd = {k: v.to(DEV) for k, v in sd.items()}
    pidx = synthetic_indices(B, rf, 256, int(fx["prompt_seed"])).to(DEV)
    outs = []
    for use_fwd in (True, False):
        g = RingGenerator(**cfg, state_dict=sdd, batch=B, n_total=N_, device=DEV)
        g.prime_with_forward = use_fwd
        g.prime(pidx)
        g.advance(N_ - rf)
        outs.append(g.samples.cpu().numpy())
    assert np.array_equal(outs[0], fx["indices"]) and np.array_equal(outs[1], fx["indices"])


def test_config2_forward_vs_generator_logits_full_batch():
    """BASELINE config 2 at B=16: two independent HIP implementations (MFMA
    full-sequence forward, ring-buffer step kernel) agree on the logits."""
    from movenet_amd.generation import RingGenerator
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=0)
    rf, B, T = 3072, 16, 3072 + 200
    idx = synthetic_indices(B, T, 256, 99).to(DEV)
    m = _model(cfg, sd)
    with torch.no_grad():
        logits = m(one_hot(idx.cpu(), 256).to(DEV), output_unnormalized=False, remove_last=True)
    g = RingGenerator(**cfg, state_dict={k: v.to(DEV) for k, v in sd.items()}, batch=B, n_total=T,
                      device=DEV)
    _, glog = g.teacher_forced(idx, logits_t0=rf)  # (B, T-rf, Q)
    assert rel_err(logits.permute(0, 2, 1).cpu(), glog.cpu()) < LOGIT_TOL


def test_dense_non_one_hot_input_forward_backward():
    """The reference's forward accepts any (B,Q,T) float tensor (its causal conv is a
    dense Conv1d, modules.py:19-30); so does this one: dense MFMA path, checked against
    the oracle with autograd."""
    from oracle import wavenet_oracle as O
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=2, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)
    sd = make_state_dict(**cfg, seed=4)
    dims = O.Dims(**cfg)
    torch.manual_seed(0)
    x = torch.rand(2, 64, 300) - 0.3
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_o = O.forward(params, dims, x, output_unnormalized=False)
    out_o.square().mean().backward()
    m = _model(cfg, sd).train()
    out = m(x.to(DEV), output_unnormalized=False)
    assert rel_err(out.detach().cpu(), out_o.detach()) < LOGIT_TOL
    out.square().mean().backward()
    for k, p in m.named_parameters():
        if params[k].grad is None:
            assert p.grad is None, k
        else:
            assert rel_err(p.grad.cpu(), params[k].grad) < 3e-4, k


def test_mu_law_kernels():
    """Formula of RESEARCH.md:156-163 (torchaudio is absent: parity UNPINNED).  Property
    checks: every class index survives decode -> encode; encode is monotone; the GPU
    kernels agree with the host formulas used by the synthetic data path."""
    from movenet_amd.dataset import mu_law_decoding, mu_law_encoding
    from movenet_amd.ops import mu_law_decode, mu_law_encode
    for Q in (64, 256):
        q = torch.arange(Q, dtype=torch.int32, device=DEV)
        x = mu_law_decode(q, Q)
        assert torch.equal(mu_law_encode(x, Q), q)
        assert x.min().item() == -1.0 and abs(x.max().item() - 1.0) < 1e-6
        assert torch.allclose(x.cpu(), mu_law_decoding(q.cpu().long(), Q), atol=1e-6)
        w = torch.linspace(-1, 1, 100001, device=DEV)
        e = mu_law_encode(w, Q)
        assert e.min().item() == 0 and e.max().item() == Q - 1 and torch.all(e[1:] >= e[:-1])
        assert (e.cpu().long() != mu_law_encoding(w.cpu(), Q)).float().mean().item() < 1e-3


def test_config2_full_size_training_step_properties():
    """BASELINE config 2 at its full size (B=16, T=16000): outputs are distributions,
    the loss sits at ln 256 (Q2), the gradient is linear in the upstream gradient and
    zero upstream gradient gives zero parameter gradients."""
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    m = _model(cfg, make_state_dict(**cfg, seed=0)).train()
    x = one_hot(synthetic_indices(16, 16000, 256, 1234).to(DEV), 256)
    out = m(x)
    assert out.shape == (16, 256, 16000 - 3072)
    assert torch.allclose(out.sum(1), torch.ones_like(out.sum(1)), atol=1e-5)
    target = x[:, :, 3072:].argmax(1)
    loss = F.cross_entropy(out, target)
    assert abs(loss.item() - np.log(256)) < 1e-3
    w = m.dense_conv.conv2.weight
    g1, = torch.autograd.grad(loss, w, retain_graph=True)
    g2, = torch.autograd.grad(2.5 * loss, w, retain_graph=True)
    assert rel_err(g2.cpu(), (2.5 * g1).cpu()) < 1e-4        # atomics: last bits may differ
    g0, = torch.autograd.grad((out * 0).sum(), m.causal_conv.conv.weight)
    assert torch.count_nonzero(g0).item() == 0


def test_cross_entropy_on_probs_matches_torch_autograd():
    """Row F3 on its own: loss, accuracy and d loss / d probs against torch ops, with a
    non-unit upstream gradient, ragged column count and ties in the argmax."""
    from movenet_amd.ops import cross_entropy_on_probs
    torch.manual_seed(3)
    B, Q, S = 3, 37, 300  # S not a multiple of the 256-column workgroups
    probs = torch.softmax(torch.randn(B, Q, S, device=DEV) * 3, 1)
    probs[0, 5, 7] = probs[0, 9, 7] = probs[0, :, 7].max() + 0.1  # tie: first maximum wins
    target = torch.randint(0, Q, (B, S), device=DEV)
    a = probs.clone().requires_grad_(True)
    b = probs.clone().requires_grad_(True)
    loss, acc = cross_entropy_on_probs(a, target)
    (loss * 2.5).backward()
    ref = F.cross_entropy(b, target)
    (ref * 2.5).backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    assert acc.item() == (b.argmax(1) == target).float().mean().item()
    assert rel_err(a.grad.cpu().numpy(), b.grad.cpu().numpy()) < 1e-6
    assert not acc.requires_grad


@pytest.mark.parametrize("layers,t_len,batch", [((3, 2), 100, 3), ((10, 1), 1024 + 700 + 37, 2), ((6, 2), 64 * 9 + 1, 1)])
def test_fused_backward_kernels_match_two_kernel_forms_and_oracle(monkeypatch, layers, t_len, batch):
    """C = K = 64 takes the fused backward (r4: csrc/fused_bwd_l.h, ONE kernel per layer with the input
    gradient in scatter form; before: csrc/fused_bwd.h, dz + residual/skip weight gradients, then dx +
    filter/gate weight gradients).  Ragged lengths put t_lo, t_skip0 and
    T inside tiles and leave chunks with a single short tile; the last layer has no dxo.
    Checked against the two-kernel forms (MOVENET_HIP_NO_FUSED_BACKWARD=1, same process) and,
    where the oracle finishes in seconds, against torch autograd on the oracle."""
    from oracle import wavenet_oracle as O
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=layers[0], stack_size=layers[1], input_channels=256, residual_channels=64,
               skip_channels=64)
    sd = make_state_dict(**cfg, seed=5, gain=1.5)
    dims = O.Dims(**cfg)
    assert t_len > dims.receptive_fields
    x = one_hot(synthetic_indices(batch, t_len, 256, 77), 256)
    w = torch.linspace(0.5, 1.5, 256).view(1, 256, 1)
    monkeypatch.setenv("MOVENET_DEBUG_GUARD", "1")  # guard bands behind the backward pass's scratch tensors (ops.py)

    def grads(form):
        # "one": the layer's backward as ONE kernel, input gradients in scatter form (csrc/fused_bwd_l.h, the default);
        # "split": the two fused halves of r2 / r3 (csrc/fused_bwd.h); "plain": the two-kernel forms
        monkeypatch.delenv("MOVENET_HIP_NO_FUSED_BACKWARD", raising=False)
        monkeypatch.delenv("MOVENET_HIP_BWD_FORM", raising=False)
        if form == "plain":
            monkeypatch.setenv("MOVENET_HIP_NO_FUSED_BACKWARD", "1")
        elif form == "split":
            monkeypatch.setenv("MOVENET_HIP_BWD_FORM", "split")
        m = _model(cfg, sd).train()
        out = m(x.to(DEV), output_unnormalized=False)
        (out * w.to(DEV)).square().mean().backward()
        monkeypatch.delenv("MOVENET_HIP_NO_FUSED_BACKWARD", raising=False)
        monkeypatch.delenv("MOVENET_HIP_BWD_FORM", raising=False)
        from movenet_amd import _native as N
        # (a silent fall-back to a slower form must not pass for the form under test)
        # ("split" falls to the generic kernels where its per-chunk slabs do not fit these small tensors' scratch)
        assert N.lib().mvn_last_backward_form() in {"one": (N.BWD_FORM_ONE,), "split": (N.BWD_FORM_HALVES, N.BWD_FORM_GENERIC),
                                                    "plain": (N.BWD_FORM_GENERIC,)}[form]
        return {k: (None if p.grad is None else p.grad.cpu()) for k, p in m.named_parameters()}

    fused, split, plain = grads("one"), grads("split"), grads("plain")
    for other in (split, plain):
        for k in fused:
            assert (fused[k] is None) == (other[k] is None), k
            if fused[k] is not None:
                assert rel_err(fused[k], other[k]) < 2e-5, k  # fp32 sums in another order
    if t_len <= 700:
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out_o = O.forward(params, dims, x, output_unnormalized=False)
        (out_o * w).square().mean().backward()
        for k, g in fused.items():
            if params[k].grad is None:
                assert g is None, k
            else:
                assert rel_err(g, params[k].grad) < 3e-4, k


@pytest.mark.parametrize("pattern", ["constant", "runs", "pairs"])
def test_embedding_gradient_with_repeated_classes(pattern):
    """The C = 64 embedding gradient (embed_grad64_kernel) keeps four time steps in flight per wave and
    merges equal classes among them before its read-modify-write of the LDS table: inputs whose
    neighbouring samples repeat a class are the case random indices hardly ever produce.  The
    length crosses a 1024-step chunk and is no multiple of 4; checked against autograd on the
    oracle (movenet/modules.py CausalConv1d, the first op of WaveNet.forward)."""
    from oracle import wavenet_oracle as O
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=3, stack_size=1, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=9, gain=1.5)
    dims = O.Dims(**cfg)
    t_len, batch = 1024 + 131, 2
    t = torch.arange(t_len)
    if pattern == "constant":
        idx = torch.full((batch, t_len), 7, dtype=torch.int64)
    elif pattern == "runs":      # runs of 1..6 equal samples, a different class per run
        run = torch.cumsum((torch.arange(t_len) % 7 == 0).long(), 0)
        idx = torch.stack([(run * 37) % 256, (run * 91 + 5) % 256])
    else:                        # a b a b ...: equal classes two steps apart
        idx = torch.stack([torch.where(t % 2 == 0, 3, 200), torch.where(t % 2 == 0, 255, 0)])
    x = one_hot(idx, 256)
    w = torch.linspace(0.5, 1.5, 256).view(1, 256, 1)
    m = _model(cfg, sd).train()
    out = m(x.to(DEV), output_unnormalized=False)
    (out * w.to(DEV)).square().mean().backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_o = O.forward(params, dims, x, output_unnormalized=False)
    (out_o * w).square().mean().backward()
    got = dict(m.named_parameters())["causal_conv.conv.weight"].grad.cpu()
    want = params["causal_conv.conv.weight"].grad
    assert rel_err(got, want) < 3e-4
    # classes that never occur get exactly zero
    never = torch.ones(256, dtype=torch.bool)
    never[idx.unique()] = False
    assert torch.all(got[:, never, :] == 0)


@pytest.mark.parametrize("layers,t_len,batch", [((3, 2), 100, 3), ((10, 1), 1024 + 700 + 37, 2), ((6, 2), 64 * 9 + 1, 17)])
def test_persistent_forward_kernel_matches_per_tile_kernel_and_oracle(monkeypatch, layers, t_len, batch):
    """C = K = 64 takes the persistent forward layer kernel (csrc/fused_fwd.h); the per-tile
    kernel (MOVENET_HIP_NO_PERSISTENT_FORWARD=1) sums each f/g value in one 128-deep chain where
    this one adds two 64-deep halves: equal to fp32 rounding.  Ragged lengths put t_begin, RF - 1
    and T inside tiles; batch 17 makes the one-round chunking uneven.  Inference (no tanh/sigmoid
    saved) and training mode, logits and probabilities."""
    from oracle import wavenet_oracle as O
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=layers[0], stack_size=layers[1], input_channels=256, residual_channels=64,
               skip_channels=64)
    sd = make_state_dict(**cfg, seed=6, gain=1.5)
    dims = O.Dims(**cfg)
    x = one_hot(synthetic_indices(batch, t_len, 256, 78), 256)

    def outputs(per_tile, train):
        if per_tile:
            monkeypatch.setenv("MOVENET_HIP_NO_PERSISTENT_FORWARD", "1")
        else:
            monkeypatch.delenv("MOVENET_HIP_NO_PERSISTENT_FORWARD", raising=False)
        m = _model(cfg, sd)
        m.train(train)
        with torch.set_grad_enabled(train):
            return m(x.to(DEV), output_unnormalized=True).detach().cpu()

    for train in (False, True):
        a, b_ = outputs(False, train), outputs(True, train)
        assert rel_err(a, b_) < 2e-6, train
        # the strip kernel is the default; its tile-kernel siblings (32 / 64 columns) must agree too
        for tile in ("32", "64"):
            monkeypatch.setenv("MOVENET_HIP_FORWARD_TILE", tile)
            assert rel_err(outputs(False, train), b_) < 2e-6, (train, tile)
        monkeypatch.delenv("MOVENET_HIP_FORWARD_TILE")
        # ... and so must the strip kernel on fp32 MFMAs (the default forms each fp32 product from six bf16 MFMAs)
        monkeypatch.setenv("MOVENET_HIP_FORWARD_MFMA", "f32")
        assert rel_err(outputs(False, train), b_) < 2e-6, (train, "f32 strip")
        monkeypatch.delenv("MOVENET_HIP_FORWARD_MFMA")
    if t_len <= 700 and batch <= 4:
        want = O.forward(sd, dims, x, output_unnormalized=True)
        assert rel_err(outputs(False, False), want) < LOGIT_TOL


@pytest.mark.parametrize("gain", [1.5, 4.0])
def test_bf16x3_forward_is_fp32_class(monkeypatch, gain):
    """csrc/fused_fwd_bf3.h: the audio-only forward layer forms every fp32 product on the bf16 matrix cores --
    operands split EXACTLY into three bf16 planes, six MFMAs per block, fp32 accumulation.  Measured against the
    same layer on fp32 MFMAs in TWO summation orders (the strip kernel, MOVENET_HIP_FORWARD_MFMA=f32, and the
    per-tile kernel, MOVENET_HIP_NO_PERSISTENT_FORWARD=1), at a size that takes the packed weight images of the layers
    AND of the head's bf16 x 3 strip kernels (30 layers, 5 x 5000 samples, 1929 ragged output columns): logits, loss and every parameter gradient (the backward pass reads the tanh / sigmoid saved by
    the forward under test) differ from the fp32 strip by no more than the two fp32 forms differ from each other
    (x 3; floor: 2e-6 of range for logits, 1e-5 for gradients).  Gain 4 makes the 30-layer stack amplify last-bit
    differences ~300 x -- there only the comparison with the fp32 pair says anything."""
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=12, gain=gain, head_gain=2.0)
    idx = synthetic_indices(5, 5000, 256, 31)  # (5 x 64 x 5024 floats of z scratch: the layers' AND the head's images fit)
    x = one_hot(idx, 256).to(DEV)

    def run(form):
        monkeypatch.delenv("MOVENET_HIP_FORWARD_MFMA", raising=False)
        monkeypatch.delenv("MOVENET_HIP_NO_PERSISTENT_FORWARD", raising=False)
        if form == "f32 strip":
            monkeypatch.setenv("MOVENET_HIP_FORWARD_MFMA", "f32")
        elif form == "f32 tile":
            monkeypatch.setenv("MOVENET_HIP_NO_PERSISTENT_FORWARD", "1")
        m = _model(cfg, sd)
        m.train(True)
        logits = m(x, output_unnormalized=False)  # raw logits (Q1: the reference's flag is inverted)
        w = torch.linspace(-1.0, 1.0, logits.numel(), device=DEV).reshape(logits.shape)
        loss = (logits * logits * w).mean()   # a loss whose gradient is not tiny anywhere
        loss.backward()
        return logits.detach().cpu(), float(loss.detach()), {k: p.grad.detach().cpu() for k, p in m.named_parameters()
                                                     if p.grad is not None}

    la, loss_a, ga = run("bf16x3")
    lb, loss_b, gb = run("f32 strip")
    lc, loss_c, gc = run("f32 tile")
    monkeypatch.delenv("MOVENET_HIP_NO_PERSISTENT_FORWARD", raising=False)
    assert torch.isfinite(la).all() and la.abs().max() > 1.0
    assert rel_err(la, lb) < max(2e-6, 3 * rel_err(lc, lb)), (rel_err(la, lb), rel_err(lc, lb))
    rms = lambda u, v: float((u.double() - v.double()).pow(2).mean().sqrt())  # noqa: E731
    assert rms(la, lb) < max(2e-7 * float(lb.abs().max()), 2 * rms(lc, lb)), (rms(la, lb), rms(lc, lb))
    if gain == 1.5:  # (at gain 4 the loss, a sum with cancellation, moves by 1e-3 between any two forms)
        assert abs(loss_a - loss_b) < 1e-6 * max(1.0, abs(loss_b))
    assert set(ga) == set(gb) and len(ga) > 180
    worst = 0.0
    for k in ga:
        e, e0 = rel_err(ga[k], gb[k]), rel_err(gc[k], gb[k])
        assert e < max(1e-5, 3 * e0), (k, e, e0)
        worst = max(worst, e)
    if gain == 1.5:
        assert rel_err(la, lb) < 2e-6 and worst < 1e-5  # absolute statement where the stack does not amplify


@pytest.mark.parametrize("t_len", [64, 63, 1000])
def test_onehot_to_index_kernels(t_len):
    """mvn_onehot_to_index: class index per column, -1 where a column is not exactly one-hot
    (two ones, a value that is neither 0 nor 1, no one at all).  T % 4 == 0 takes the float4
    kernel, otherwise the scalar one."""
    from movenet_amd import _native as N
    B, Q = 3, 256
    idx = synthetic_indices(B, t_len, Q, 5)
    x = one_hot(idx, Q)
    want = idx.clone().to(torch.int32)
    x[0, 7, 3] = 1.0
    want[0, 3] = -1 if idx[0, 3] != 7 else want[0, 3]       # a second one
    x[1, int(idx[1, 5]), 5] = 0.5
    want[1, 5] = -1                                          # neither 0 nor 1
    x[2, int(idx[2, t_len - 1]), t_len - 1] = 0.0
    want[2, t_len - 1] = -1                                  # no one at all
    x[2, (int(idx[2, 0]) + 1) % Q, 0] = 1e-30
    want[2, 0] = -1                                          # a stray tiny value
    xd = x.to(DEV).contiguous()
    got = torch.empty((B, t_len), dtype=torch.int32, device=DEV)
    N.check(N.lib().mvn_onehot_to_index(xd.data_ptr(), got.data_ptr(), B, Q, t_len, None), "mvn_onehot_to_index")
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)
