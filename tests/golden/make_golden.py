"""Generate the golden vectors under tests/golden from the REFERENCE itself.

Run in the build container only (``python tests/golden/make_golden.py``): it
imports ``/root/reference/movenet/wavenet.py`` unmodified on CPU (torchtyping,
which the reference uses for annotations only, is absent offline and is
supplied as an in-memory stub), loads this build's seeded weights into it via
``load_state_dict`` and records its outputs.  While doing so it asserts that
``oracle/wavenet_oracle.py`` reproduces every recorded output, which is what
pins the oracle.  The reference never travels to the GPU
box: the tests read only the ``.npz`` files written here, which hold inputs'
seeds, expected outputs and a SHA-256 of the regenerated weights.

Fixtures:
  G1  config-1 shape (L=2x2, Q=64, C=K=16): logits + probs, B=2, T=64
  G2  30-layer config-2 model: logits for B=2, T=RF+40; window == full-forward
  G3  greedy generate (temperature=0.0), sharpened weights: free-running
      indices, top-2 margins, for the small and the 30-layer model
  G4  trainer arithmetic: CE-on-probs loss, accuracy, per-parameter grad norms
  G5  pre-sampling probabilities (double softmax) at temperature 0.5 and 1.0
  G6  60-layer C=K=128 fp32 logits (yard-stick for config 5's fp16 tolerance)
  G7  upsample_video (B,160,64,64,1) -> (B,C,160000) checksums + samples
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from movenet_amd.utils.weights import (  # noqa: E402
    make_state_dict, one_hot, state_dict_sha256, synthetic_indices,
)
from oracle import wavenet_oracle as O  # noqa: E402


def import_reference():
    stub = types.ModuleType("torchtyping")

    class TensorType:  # annotation-only stand-in
        def __class_getitem__(cls, item):
            return cls

    stub.TensorType = TensorType
    sys.modules["torchtyping"] = stub
    sys.path.insert(0, "/root/reference")
    from movenet.wavenet import WaveNet  # type: ignore
    return WaveNet


def build_ref(WaveNet, cfg, sd):
    m = WaveNet(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.eval()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


def eq(a, b, what, tol=0.0):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    d = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    print(f"  oracle vs reference [{what}]: max abs diff {d:.3e}")
    assert d <= tol, (what, d)


CFG1 = dict(layer_size=2, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)
CFG2 = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
CFG5 = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)


def main():
    torch.set_num_threads(8)
    WaveNet = import_reference()

    # ---- G1 ------------------------------------------------------------
    seed, B, T = 11, 2, 64
    sd = make_state_dict(**CFG1, seed=seed)
    dims = O.Dims(**CFG1)
    ref = build_ref(WaveNet, CFG1, sd)
    assert ref.receptive_fields == dims.receptive_fields == 8
    idx = synthetic_indices(B, T, 64, seed=1234)
    x = one_hot(idx, 64)
    with torch.no_grad():
        logits = ref(x, output_unnormalized=False, remove_last=False)
        probs = ref(x)
    eq(O.forward(sd, dims, x, output_unnormalized=False, remove_last=False), logits, "G1 logits")
    eq(O.forward(sd, dims, x), probs, "G1 probs")
    save("g1_small_forward.npz", cfg=np.array(list(CFG1.values())), weight_seed=seed,
         weight_sha=state_dict_sha256(sd), idx_seed=1234, B=B, T=T,
         logits=logits.numpy(), probs=probs.numpy())

    # ---- G2 ------------------------------------------------------------
    seed, B = 0, 2
    sd2 = make_state_dict(**CFG2, seed=seed)
    dims2 = O.Dims(**CFG2)
    ref2 = build_ref(WaveNet, CFG2, sd2)
    rf = ref2.receptive_fields
    assert rf == dims2.receptive_fields == 3072
    T = rf + 40
    idx = synthetic_indices(B, T, 256, seed=1234)
    x = one_hot(idx, 256)
    with torch.no_grad():
        logits = ref2(x, output_unnormalized=False, remove_last=False)  # (B,256,41)
        win = ref2(x[:, :, 7:7 + rf], output_unnormalized=False, remove_last=False)
    eq(O.forward(sd2, dims2, x, output_unnormalized=False, remove_last=False), logits, "G2 logits")
    print("  window-vs-full max abs diff",
          (win[:, :, 0] - logits[:, :, 7]).abs().max().item())
    # ring-buffer restatement against the reference, teacher-forced
    _, ring_logits = O.generate_ring(sd2, dims2, idx.numpy(), T, forced_idx=idx.numpy())
    d = np.abs(ring_logits - logits[:, :, :-1].permute(0, 2, 1).numpy()).max()
    print(f"  ring-buffer restatement vs reference logits: max abs diff {d:.3e}")
    assert d < 5e-6
    save("g2_l30_forward.npz", cfg=np.array(list(CFG2.values())), weight_seed=seed,
         weight_sha=state_dict_sha256(sd2), idx_seed=1234, B=B, T=T, logits=logits.numpy(),
         window_start=7, window_logits=win.numpy())

    # ---- G3 greedy generate -------------------------------------------
    for tag, cfg, n_new, wseed, gn_, hg in (("small", CFG1, 96, 3, 3.0, 6.0),
                                            ("l30", CFG2, 64, 1, 2.0, 6.0)):
        sdg = make_state_dict(**cfg, seed=wseed, gain=gn_, head_gain=hg)
        dg = O.Dims(**cfg)
        refg = build_ref(WaveNet, cfg, sdg)
        rf = refg.receptive_fields
        Q = cfg["input_channels"]
        B = 2
        N = rf + n_new
        pidx = synthetic_indices(B, rf, Q, seed=77)
        prompt = one_hot(pidx, Q)
        gen = refg.generate(prompt, n_samples=N, temperature=0.0)
        assert gen.shape == (B, Q, N) and torch.all(gen.sum(1) == 1)
        gidx = gen.argmax(1)
        ogen, margins = O.generate_windowed(sdg, dg, prompt, n_samples=N, temperature=0.0,
                                            return_margins=True)
        eq(ogen, gen, f"G3 {tag} greedy one-hot")
        # teacher-forced logits from the reference (one full forward over the result)
        with torch.no_grad():
            tf_logits = refg(gen, output_unnormalized=False, remove_last=True)  # (B,Q,N-rf)
        top2 = torch.topk(tf_logits, 2, dim=1).values
        lmargin = (top2[:, 0] - top2[:, 1])
        print(f"  G3 {tag}: min top-2 logit margin {lmargin.min().item():.4e}, "
              f"logit range {tf_logits.max().item() - tf_logits.min().item():.3f}, "
              f"distinct classes {len(torch.unique(gidx[:, rf:]))}")
        ridx, _ = O.generate_ring(sdg, dg, pidx.numpy(), N)
        assert np.array_equal(ridx, gidx.numpy()), "ring-buffer greedy != reference greedy"
        save(f"g3_{tag}_greedy.npz", cfg=np.array(list(cfg.values())), weight_seed=wseed,
             gain=gn_, head_gain=hg, weight_sha=state_dict_sha256(sdg), prompt_seed=77, B=B, N=N,
             indices=gidx.numpy(), logit_margin=lmargin.numpy(), tf_logits=tf_logits.numpy())

    # ---- G4 trainer arithmetic ----------------------------------------
    for tag, cfg, T, wseed in (("small", CFG1, 200, 5), ("l30", CFG2, 3072 + 128, 0)):
        sdt = make_state_dict(**cfg, seed=wseed)
        dt = O.Dims(**cfg)
        reft = build_ref(WaveNet, cfg, sdt).train()
        Q = cfg["input_channels"]
        B = 2
        idx = synthetic_indices(B, T, Q, seed=1234)
        x = one_hot(idx, Q)
        out = reft(x)
        target = x[:, :, reft.receptive_fields:].argmax(1)
        loss = F.cross_entropy(out, target)
        acc = (out.argmax(1) == target).float().mean()
        loss.backward()
        gn = {k: p.grad.norm().item() for k, p in reft.named_parameters() if p.grad is not None}
        oloss, oacc, oout, ograds = O.train_step_arithmetic(sdt, dt, x)
        eq(oloss, loss.detach(), f"G4 {tag} loss")
        eq(oacc, acc, f"G4 {tag} acc")
        for k in gn:
            eq(ograds[k], dict(reft.named_parameters())[k].grad, f"G4 {tag} grad {k}", tol=1e-9)
        assert set(gn) == set(ograds)
        names = sorted(gn)
        # a few full gradients for element-wise checks
        full = {
            "grad_causal": dict(reft.named_parameters())["causal_conv.conv.weight"].grad.numpy(),
            "grad_l0_filter": dict(reft.named_parameters())[
                "residual_conv_stack.conv_layers.0.conv_filter.conv.weight"].grad.numpy(),
            "grad_last_skip_w": dict(reft.named_parameters())[
                f"residual_conv_stack.conv_layers.{dt.n_layers - 1}.conv_skip.weight"].grad.numpy(),
            "grad_head2_b": dict(reft.named_parameters())["dense_conv.conv2.bias"].grad.numpy(),
        }
        save(f"g4_{tag}_train.npz", cfg=np.array(list(cfg.values())), weight_seed=wseed,
             weight_sha=state_dict_sha256(sdt), idx_seed=1234, B=B, T=T,
             loss=loss.item(), acc=acc.item(), grad_names=np.array(names),
             grad_norms=np.array([gn[k] for k in names]), **full)

    # ---- G5 pre-sampling probabilities --------------------------------
    sds = make_state_dict(**CFG1, seed=3, gain=3.0, head_gain=6.0)
    ds = O.Dims(**CFG1)
    refs = build_ref(WaveNet, CFG1, sds)
    pidx = synthetic_indices(3, ds.receptive_fields, 64, seed=78)
    win = one_hot(pidx, 64)
    with torch.no_grad():
        out = refs(win, output_unnormalized=True, remove_last=False)  # probs (3,64,1)
    arrays = {}
    for Tmp in (0.5, 1.0):
        o = out.clone()
        o /= Tmp
        p2 = F.softmax(o, dim=1)
        eq(O.pre_sampling_probs(out, Tmp), p2, f"G5 T={Tmp}")
        arrays[f"p2_T{str(Tmp).replace('.', '_')}"] = p2.numpy()
    save("g5_small_presampling.npz", cfg=np.array(list(CFG1.values())), weight_seed=3,
         gain=3.0, head_gain=6.0, weight_sha=state_dict_sha256(sds), prompt_seed=78, B=3,
         probs=out.numpy(), **arrays)

    # ---- G6 60-layer C=128 --------------------------------------------
    sd5 = make_state_dict(**CFG5, seed=0)
    d5 = O.Dims(**CFG5)
    ref5 = build_ref(WaveNet, CFG5, sd5)
    assert ref5.receptive_fields == 6144
    T = 6144 + 8
    idx = synthetic_indices(1, T, 256, seed=1234)
    x = one_hot(idx, 256)
    with torch.no_grad():
        logits = ref5(x, output_unnormalized=False, remove_last=False)
    eq(O.forward(sd5, d5, x, output_unnormalized=False, remove_last=False), logits, "G6 logits")
    save("g6_l60_forward.npz", cfg=np.array(list(CFG5.values())), weight_seed=0,
         weight_sha=state_dict_sha256(sd5), idx_seed=1234, B=1, T=T, logits=logits.numpy())

    # ---- G7 upsample_video --------------------------------------------
    sdv = make_state_dict(**CFG1, seed=9)
    refv = build_ref(WaveNet, CFG1, sdv)
    rng = np.random.default_rng(4321)
    video = torch.from_numpy(rng.random((1, 160, 64, 64, 1), dtype=np.float32))
    with torch.no_grad():
        up = refv.upsample_video(video)
    eq(O.upsample_video(sdv, video), up, "G7 upsample_video")
    cols = np.array([0, 1, 9, 10, 999, 1000, 12345, 80000, 159990, 159999])
    save("g7_upsample_video.npz", cfg=np.array(list(CFG1.values())), weight_seed=9,
         weight_sha=state_dict_sha256(sdv), video_seed=4321, cols=cols,
         up_cols=up[:, :, cols].numpy(), up_sum=up.double().sum().item(),
         up_abs_sum=up.double().abs().sum().item())


if __name__ == "__main__":
    main()
