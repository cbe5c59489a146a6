"""Pins oracle/ring_oracle.c (the cached generator restated in C, used by bench.py's
cached-CPU line) to the reference's own outputs: fixture G3 (greedy indices recorded from
movenet's WaveNet.generate) and G2 (logits of its forward), and to the numpy twin."""
import numpy as np

from helpers import cfg_of, synthetic_indices, weights_of
from oracle import ring_c
from oracle import wavenet_oracle as O


def test_c_ring_matches_reference_greedy_fixtures(golden):
    for name in ("g3_small_greedy.npz", "g3_l30_greedy.npz"):
        fx = golden(name)
        cfg, dims, sd = weights_of(fx)
        B, N_, rf = int(fx["B"]), int(fx["N"]), dims.receptive_fields
        pidx = synthetic_indices(B, rf, cfg["input_channels"], int(fx["prompt_seed"])).numpy()
        got, _ = ring_c.generate_ring_c(sd, dims, pidx, N_, threads=2)
        assert np.array_equal(got, fx["indices"]), name


def test_c_ring_teacher_forced_logits_match_reference(golden):
    fx = golden("g2_l30_forward.npz")
    cfg, dims, sd = weights_of(fx)
    B, T, rf = int(fx["B"]), int(fx["T"]), dims.receptive_fields
    idx = synthetic_indices(B, T, 256, int(fx["idx_seed"])).numpy()
    _, logits = ring_c.generate_ring_c(sd, dims, idx[:, :rf], T, forced_idx=idx, threads=2)
    want = np.transpose(fx["logits"][:, :, :-1], (0, 2, 1))  # (B, T-rf, Q): predicts times rf..T-1
    assert np.abs(logits - want).max() / np.abs(want).max() < 2e-5


def test_c_ring_equals_numpy_ring():
    from movenet_amd.utils.weights import make_state_dict
    cfg = dict(layer_size=3, stack_size=2, input_channels=64, residual_channels=16, skip_channels=16)
    sd = make_state_dict(**cfg, seed=3, gain=2.0, head_gain=6.0)
    dims = O.Dims(**cfg)
    rf = dims.receptive_fields
    p = synthetic_indices(3, rf, 64, 11).numpy()
    a, la = O.generate_ring(sd, dims, p, rf + 60)
    b, lb = ring_c.generate_ring_c(sd, dims, p, rf + 60, threads=3)
    assert np.array_equal(a, b)
    assert np.abs(la - lb).max() / np.abs(la).max() < 2e-6
