"""GPU: the fused pieces of the trainer's tail (row F3) against their unfused forms.

* ``wavenet_forward_loss`` (softmax + cross_entropy-on-probabilities + accuracy in one pass,
  d loss / d logits in one pass) returns the same loss / accuracy / probabilities as
  ``cross_entropy_on_probs(model(audio), target)`` and the same gradients, and reproduces
  the reference's recorded trainer arithmetic (fixture G4).
* ``FlatAdamW`` (one HIP kernel over one flat buffer) follows torch.optim.AdamW / Adam to
  1e-6 relative, including parameters without a gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import one_hot, rel_err, synthetic_indices, weights_of
from movenet_amd.utils.weights import make_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(cfg, sd):
    from movenet_amd.wavenet import WaveNet
    m = WaveNet(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("name", ["g4_small_train.npz", "g4_l30_train.npz"])
def test_fused_loss_equals_unfused_and_g4(golden, name):
    from movenet_amd.ops import cross_entropy_on_probs, wavenet_forward_loss
    fx = golden(name)
    cfg, dims, sd = weights_of(fx)
    Q = cfg["input_channels"]
    x = one_hot(synthetic_indices(int(fx["B"]), int(fx["T"]), Q, int(fx["idx_seed"])), Q).to(DEV)
    m = _model(cfg, sd).train()
    out = m(x)
    target = x[:, :, m.receptive_fields:].argmax(1)
    loss, acc = cross_entropy_on_probs(out, target)
    loss.backward()
    want = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    loss2, acc2, probs = wavenet_forward_loss(m, x)
    assert torch.equal(probs, out.detach())           # same bits
    assert loss2.item() == loss.item() and acc2.item() == acc.item()
    assert abs(loss2.item() - float(fx["loss"])) < 2e-6 and abs(acc2.item() - float(fx["acc"])) < 1e-6
    (2.0 * loss2).backward()                           # the upstream gradient is honoured
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(want) == [str(n) for n in fx["grad_names"]]
    for k in want:
        assert rel_err(got[k].cpu(), 2.0 * want[k].cpu()) < 2e-6, k
    # a non-one-hot input takes the dense causal conv and an explicit target
    soft = torch.softmax(torch.randn(1, Q, int(fx["T"]), device=DEV), 1)
    tg = soft[:, :, m.receptive_fields:].argmax(1)
    l3, a3, p3 = wavenet_forward_loss(m, soft)
    with torch.no_grad():
        ref = m(soft)
    assert torch.equal(p3, ref) and abs(l3.item() - F.cross_entropy(ref, tg).item()) < 1e-6
    assert a3.item() == (ref.argmax(1) == tg).float().mean().item()


@pytest.mark.parametrize("decoupled", [True, False])
def test_flat_adamw_matches_torch(decoupled):
    from movenet_amd.optim import FlatAdamW
    torch.manual_seed(3)
    shapes = [(64, 64, 2), (64,), (7, 5), (1,), (256, 64, 1), (33,)]
    ref = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    topt = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ref, **kw)
    fopt = FlatAdamW(mine, decoupled=decoupled, **kw)
    sched_t = torch.optim.lr_scheduler.StepLR(topt, step_size=2, gamma=0.5)
    sched_f = torch.optim.lr_scheduler.StepLR(fopt, step_size=2, gamma=0.5)
    sizes = [p.numel() for p in ref]
    for step in range(5):
        no_grad = {2, 3} if step != 3 else {0}      # parameters 2, 3 (adjacent) get no gradient
        flat = torch.randn(sum(sizes), device=DEV)  # gradients as views of ONE buffer (ops.py)
        off = 0
        for i, (pr, pm, k) in enumerate(zip(ref, mine, sizes)):
            g = flat[off:off + k].view(pr.shape)
            off += k
            pr.grad = None if i in no_grad else g.clone()
            pm.grad = None if i in no_grad else g
        topt.step()
        fopt.step()
        sched_t.step()
        sched_f.step()
        if step < 3:
            assert fopt.last_launches == 1  # one launch, the gap is a skip range
        # (step 3 gives parameters 2, 3 their FIRST gradient: torch bias-corrects them as step 1,
        # so they travel in a launch of their own)
        for i, (pr, pm) in enumerate(zip(ref, mine)):
            assert rel_err(pm.detach().cpu(), pr.detach().cpu()) < 1e-6, (step, i)
    # separate gradient tensors (the video encoder's) step with one launch each
    for pr, pm in zip(ref, mine):
        g = torch.randn_like(pr)
        pr.grad, pm.grad = g.clone(), g.clone()
    topt.step()
    fopt.step()
    assert fopt.last_launches == len(ref)
    for pr, pm in zip(ref, mine):
        assert rel_err(pm.detach().cpu(), pr.detach().cpu()) < 1e-6
    # parameters are views of one flat buffer and keep their identity
    assert all(p.untyped_storage().data_ptr() == fopt.flat.untyped_storage().data_ptr() for p in mine)


def test_trainer_step_is_one_optimizer_launch():
    """Audio-only config-2 model through the trainer's own pieces: after backward the flat
    optimizer covers every parameter with ONE launch (the last layer's residual conv, without
    gradient, is a skip range) and leaves the gradient-less parameters untouched."""
    from movenet_amd.ops import wavenet_forward_loss
    from movenet_amd.optim import FlatAdamW, order_like_backward
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    sd = make_state_dict(**cfg, seed=0)
    m = _model(cfg, sd).train()
    opt = FlatAdamW(order_like_backward(m), lr=1e-3)
    x = one_hot(synthetic_indices(2, 3400, 256, 1), 256).to(DEV)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    loss, _, _ = wavenet_forward_loss(m, x)
    loss.backward()
    opt.step()
    assert opt.last_launches == 1
    after = m.state_dict()
    last = "residual_conv_stack.conv_layers.29.conv_residual."
    for k in before:
        untouched = k.startswith("video_") or ".context_conv_" in k or k.startswith(last)
        assert torch.equal(after[k], before[k]) == untouched, k
