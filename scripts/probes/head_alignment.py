"""How much do the head and loss kernels lose to the 4-byte-aligned rows of the (B, Q, S) output?
    rocprofv3 --kernel-trace --stats -d out -o t --output-format csv -- python3 scripts/probes/head_alignment.py 16031
Runs bench.py's config-2 train step with T given on the command line: T = 16000 gives S = T - 3071 = 12929 columns
(rows start at any 4-byte offset); T = 16031 gives S = 12960 = 405 x 32 (every row 128-byte aligned).  Compare the
per-column times of dense_strip_kernel<256, ...>, softmax_ce_*_cols_kernel, gemm_wx_staged<DenseOp<0, 2, true>>
and wgrad2_kernel<WgDenseOp<0>, 2> between the two runs."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

bench.TRAIN_WORKLOADS[2]["t_len"] = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
print(json.dumps(bench.train_leg(torch.device("cuda:0"), 1, 0, steps=3, warmup=1, config=2)))
