"""CPU: the oracle against the golden vectors recorded from the reference
(tests/golden/make_golden.py).  On the box that generated them the match is
bit-exact; across hosts ATen may pick other conv kernels, hence a 2e-6-of-range
tolerance on floats and exact equality on indices."""
import numpy as np
import torch

from helpers import cfg_of, one_hot, rel_err, synthetic_indices, weights_of
from oracle import wavenet_oracle as O

TOL = 2e-6


def test_g1_small_forward(golden):
    fx = golden("g1_small_forward.npz")
    cfg, dims, sd = weights_of(fx)
    assert dims.receptive_fields == 8
    x = one_hot(synthetic_indices(int(fx["B"]), int(fx["T"]), cfg["input_channels"], int(fx["idx_seed"])),
                cfg["input_channels"])
    logits = O.forward(sd, dims, x, output_unnormalized=False, remove_last=False)
    probs = O.forward(sd, dims, x)
    assert logits.shape == fx["logits"].shape and probs.shape == fx["probs"].shape
    assert probs.shape[2] == int(fx["T"]) - 8  # remove_last drops one column
    assert rel_err(logits, fx["logits"]) < TOL
    assert np.abs(probs.numpy() - fx["probs"]).max() < TOL
    assert np.allclose(probs.sum(1).numpy(), 1.0, atol=1e-5)  # Q1: default returns probabilities


def test_g2_l30_forward_and_ring_equivalence(golden):
    fx = golden("g2_l30_forward.npz")
    cfg, dims, sd = weights_of(fx)
    assert dims.receptive_fields == 3072
    idx = synthetic_indices(int(fx["B"]), int(fx["T"]), 256, int(fx["idx_seed"]))
    logits = O.forward(sd, dims, one_hot(idx, 256), output_unnormalized=False, remove_last=False)
    assert rel_err(logits, fx["logits"]) < TOL
    # the cached formulation equals the reference's full forward
    _, ring = O.generate_ring(sd, dims, idx.numpy(), int(fx["T"]), forced_idx=idx.numpy())
    want = np.transpose(fx["logits"][:, :, :-1], (0, 2, 1))
    assert np.abs(ring - want).max() < 5e-6
    # and the RF-long window equals the matching column
    ws = int(fx["window_start"])
    assert np.abs(fx["window_logits"][:, :, 0] - fx["logits"][:, :, ws]).max() < 1e-6


def test_g3_small_greedy(golden):
    fx = golden("g3_small_greedy.npz")
    cfg, dims, sd = weights_of(fx)
    Q, rf, N = cfg["input_channels"], dims.receptive_fields, int(fx["N"])
    pidx = synthetic_indices(int(fx["B"]), rf, Q, int(fx["prompt_seed"]))
    gen = O.generate_windowed(sd, dims, one_hot(pidx, Q), n_samples=N, temperature=0.0)
    assert np.array_equal(gen.argmax(1).numpy(), fx["indices"])
    ridx, rlog = O.generate_ring(sd, dims, pidx.numpy(), N)
    assert np.array_equal(ridx, fx["indices"])
    assert rel_err(np.transpose(rlog, (0, 2, 1)), fx["tf_logits"]) < 1e-5
    assert fx["logit_margin"].min() > 1e-2


def test_g3_l30_greedy_ring(golden):
    fx = golden("g3_l30_greedy.npz")
    cfg, dims, sd = weights_of(fx)
    pidx = synthetic_indices(int(fx["B"]), dims.receptive_fields, 256, int(fx["prompt_seed"]))
    ridx, rlog = O.generate_ring(sd, dims, pidx.numpy(), int(fx["N"]))
    assert np.array_equal(ridx, fx["indices"])
    assert rel_err(np.transpose(rlog, (0, 2, 1)), fx["tf_logits"]) < 1e-5
    assert fx["logit_margin"].min() > 1e-2


def test_g4_small_train_arithmetic(golden):
    fx = golden("g4_small_train.npz")
    cfg, dims, sd = weights_of(fx)
    x = one_hot(synthetic_indices(int(fx["B"]), int(fx["T"]), cfg["input_channels"], int(fx["idx_seed"])),
                cfg["input_channels"])
    loss, acc, out, grads = O.train_step_arithmetic(sd, dims, x)
    assert abs(loss.item() - float(fx["loss"])) < 1e-6
    assert abs(acc.item() - float(fx["acc"])) < 1e-7
    # Q2: cross_entropy over probabilities sits at ~ln(Q)
    assert abs(loss.item() - np.log(cfg["input_channels"])) < 0.05
    names = [str(n) for n in fx["grad_names"]]
    assert sorted(grads) == names  # video/context parameters receive no gradient
    got = np.array([grads[n].norm().item() for n in names])
    assert np.allclose(got, fx["grad_norms"], rtol=1e-4, atol=1e-10)
    assert np.allclose(grads["causal_conv.conv.weight"].numpy(), fx["grad_causal"], rtol=1e-4, atol=1e-9)


def test_g5_presampling(golden):
    fx = golden("g5_small_presampling.npz")
    cfg, dims, sd = weights_of(fx)
    Q = cfg["input_channels"]
    win = one_hot(synthetic_indices(int(fx["B"]), dims.receptive_fields, Q, int(fx["prompt_seed"])), Q)
    probs = O.forward(sd, dims, win, output_unnormalized=True, remove_last=False)
    assert np.abs(probs.numpy() - fx["probs"]).max() < TOL
    for T, key in ((0.5, "p2_T0_5"), (1.0, "p2_T1_0")):
        p2 = O.pre_sampling_probs(probs, T)
        assert np.abs(p2.numpy() - fx[key]).max() < TOL
        # Q3: the double softmax is nearly uniform
        assert p2.max().item() < 3.0 / Q * np.e ** (1 / T)


def test_g6_l60_forward(golden):
    fx = golden("g6_l60_forward.npz")
    cfg, dims, sd = weights_of(fx)
    assert dims.receptive_fields == 6144
    x = one_hot(synthetic_indices(1, int(fx["T"]), 256, int(fx["idx_seed"])), 256)
    logits = O.forward(sd, dims, x, output_unnormalized=False, remove_last=False)
    assert rel_err(logits, fx["logits"]) < TOL


def test_g7_upsample_video(golden):
    fx = golden("g7_upsample_video.npz")
    cfg, dims, sd = weights_of(fx)
    rng = np.random.default_rng(int(fx["video_seed"]))
    video = torch.from_numpy(rng.random((1, 160, 64, 64, 1), dtype=np.float32))
    up = O.upsample_video(sd, video)
    assert up.shape == (1, cfg["residual_channels"], 160000)
    assert rel_err(up[:, :, fx["cols"]], fx["up_cols"]) < 1e-5
    assert abs(up.double().abs().sum().item() - float(fx["up_abs_sum"])) / float(fx["up_abs_sum"]) < 1e-5


def test_errors_and_edges():
    dims = O.Dims(2, 2, 64, 16, 16)
    try:
        dims.output_size(dims.receptive_fields - 1)
        assert False
    except ValueError as e:
        assert "receptive" in str(e)
    assert dims.output_size(dims.receptive_fields) == 1
    assert O.Dims(10, 3, 256, 64, 64).dilations[:11] == [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1]
