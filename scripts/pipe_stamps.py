"""Where does a PIPE generator step spend its time?  Diagnostic only.

    python -m movenet_amd.csrc.build --stamps
    MOVENET_HIP_LIB=movenet_amd/lib/libmovenet_hip_stamps.so python scripts/pipe_stamps.py

Prints, averaged over steps 8..63 of one launch and over the 16 sequences:
per-stage compute time (inbox complete -> outbox sent) and per-hop time
(outbox sent by stage s -> inbox complete at stage s+1), in microseconds
(s_memrealtime ticks are 10 ns)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movenet_amd import _native as N  # noqa: E402
from movenet_amd.generation import RingGenerator  # noqa: E402
from movenet_amd.utils.weights import make_state_dict, synthetic_indices  # noqa: E402

H16 = "--h16" in sys.argv      # config 5 with fp16 operands: 31 stages, the stamps cover the first 16
WIDE = "--c128" in sys.argv or H16   # BASELINE config 5: 61 stages, the stamps cover the first 16
FOLD = "--fold" in sys.argv    # the FOLD variant (11 stages of three folded layers)
if WIDE:
    CFG = dict(layer_size=10, stack_size=6, input_channels=256, residual_channels=128, skip_channels=128)
    B, rf = 4, 6144
else:
    CFG = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    B, rf = 16, 3072
dev = "cuda:0"
sd = {k: v.to(dev) for k, v in make_state_dict(**CFG, seed=0).items()}
g = RingGenerator(**CFG, state_dict=sd, batch=B, n_total=rf + 4000, device=dev,
                  variant=N.GEN_PIPE_F16 if H16 else N.GEN_FOLD if FOLD else N.GEN_PIPE)
g.prime(synthetic_indices(B, rf, 256, 1234).to(dev))
g.advance(1000)
g.advance(1000)
g.check_errors()
lib = N.lib()
buf = np.zeros((16, 16, 64, 4), dtype=np.uint64)
read_stamps = lib.mvn_debug_read_stamps_h16 if H16 else lib.mvn_debug_read_stamps_fold if FOLD else lib.mvn_debug_read_stamps
read_fine = lib.mvn_debug_read_fine_h16 if H16 else lib.mvn_debug_read_fine_fold if FOLD else lib.mvn_debug_read_fine
read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert read_stamps(buf.ctypes.data, buf.size) == 0
NS = 16 if WIDE else (11 if FOLD else 9)          # stages looked at
st = buf[:B, :NS, 8:, :2].astype(np.int64)         # (B, NS, steps, {in, out}) wall clock
ck = buf[:B, :NS, 8:, 2:].astype(np.int64)         # same stamps on the shader clock
mhz = ((ck[:, :NS - 1, :, 1] - ck[:, :NS - 1, :, 0]) /
       np.maximum(st[:, :NS - 1, :, 1] - st[:, :NS - 1, :, 0], 1) * 100.0)
print("shader clock during layer-stage compute: median %.0f MHz (min %.0f, max %.0f)"
      % (np.median(mhz), mhz.min(), mhz.max()))
tick = 0.01                                         # us
compute = (st[:, :NS - 1, :, 1] - st[:, :NS - 1, :, 0]) * tick        # layer stages
head = (st[:, NS - 1, 1:, 1] - st[:, NS - 1, :-1, 0]) * tick           # head: in(step k) -> out(step k+1)
hops = []
for s in range(NS):
    nxt = (s + 1) % NS
    if WIDE and s == NS - 1:
        hops.append(np.zeros(1))
    elif s == NS - 1:    # head's send at step k is received by stage 0 at step k
        d = (st[:, 0, :, 0] - st[:, NS - 1, :, 1]) * tick
    else:
        d = (st[:, nxt, :, 0] - st[:, s, :, 1]) * tick
    hops.append(d)
step = (st[:, 0, 1:, 0] - st[:, 0, :-1, 0]) * tick
print("step period (us): mean %.2f  min %.2f  max %.2f" % (step.mean(), step.min(), step.max()))
for s in range(NS - 1):
    print(f"stage {s}: compute {compute[:, s].mean():6.2f} us   hop to next {hops[s].mean():6.2f} us"
          f" (min {hops[s].min():.2f}, max {hops[s].max():.2f})")
if not WIDE:
  print(f"head   : compute {head.mean():6.2f} us   hop to stage 0 {hops[NS-1].mean():6.2f} us"
      f" (min {hops[NS-1].min():.2f}, max {hops[NS-1].max():.2f})")
print("sum compute %.2f us, sum hops %.2f us" % (compute.mean(0).mean(1).sum() + head.mean(),
                                               sum(h.mean() for h in hops)))
if "--json" in sys.argv and not WIDE:
    import json
    path = sys.argv[sys.argv.index("--json") + 1]
    json.dump({"variant": "fold" if FOLD else "pipe", "batch": B, "stages": NS - 1,
               "note": "in-kernel s_memrealtime stamps of the diagnostic build (libmovenet_hip_stamps.so), averaged over steps 8..63 "
                       "of one launch and the sequences: stage chain = inbox complete -> outbox sent, hop = outbox sent -> the next "
                       "stage's inbox complete (the last hop: head -> stage 0), head = its inbox complete -> its send",
               "step_us": float(step.mean()), "stage_chain_us": [float(compute[:, s].mean()) for s in range(NS - 1)],
               "hop_us": [float(h.mean()) for h in hops], "head_us": float(head.mean()),
               "shader_clock_MHz_median": float(np.median(mhz))}, open(path, "w"), indent=1)

fine = np.zeros((16, 16, 64, 8), dtype=np.uint64)
read_fine.argtypes = [C.c_void_p, C.c_size_t]
if H16 and read_fine(fine.ctypes.data, fine.size) == 0:
    f = fine[:B, :NS - 1, 8:, :6].astype(np.int64)
    print("a stage of gen_pipe_h16_kernel (MFMA form), shader cycles (median over stages, steps, sequences; lane 0 of wave 0):")
    for nm, i0, i1 in (("layer 0 filter/gate phase (vector reads, 8 MFMAs, gate, z write)", 0, 1), ("barrier", 1, 2),
                       ("layer 0 residual phase (vector reads, 4 MFMAs, stream update)", 2, 3), ("barrier", 3, 4),
                       ("layer 1: both phases and their barrier, up to the hand-on", 4, 5), ("stage: first phase start -> hand-on", 0, 5)):
        print(f"  {nm:52s} {np.median(f[..., i1] - f[..., i0]):7.0f}")
elif FOLD and read_fine(fine.ctypes.data, fine.size) == 0:
    f = fine[:B, :NS - 1, 8:, :].astype(np.int64)
    seg = [("phase 0 chain work (xp, zl dots, gate, z0 write)", 0, 1), ("phase 0 helper work (thread 256)", 0, 6),
           ("phase 0 incl. barrier", 0, 2), ("phase 1 chain work", 2, 3), ("phase 1 helper work", 2, 7),
           ("phase 1 incl. barrier", 2, 4), ("phase 2 chain work incl. send", 4, 5), ("whole stage", 0, 5)]
    print("a FOLD stage, shader cycles (median over stages, steps, sequences):")
    for nm, i0, i1 in seg:
        print(f"  {nm:52s} {np.median(f[..., i1] - f[..., i0]):7.0f}")
elif read_fine(fine.ctypes.data, fine.size) == 0:
    f = fine[:B, :NS - 1, 8:, :6].astype(np.int64)
    names = ["FG: x loads, dots, lane sums, gate, z write", "barrier FG->RS (two threads' clocks)",
             "RS: z loads, dots, lane sums", "RS: residual/skip update (+ hand-off stores)",
             "barrier RS->FG"]
    print("first layer of a stage, shader cycles (median):")
    for k, nm in enumerate(names):
        d = f[..., k + 1] - f[..., k]
        print(f"  {nm:46s} {np.median(d):7.0f}")
    print(f"  {'whole layer':46s} {np.median(f[..., 5] - f[..., 0]):7.0f}")
