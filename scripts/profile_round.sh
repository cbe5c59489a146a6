#!/bin/bash
# Profiles of one round, written under gpurun_out/<tag>/ on the GPU box (copy the summaries
# into profiles/ afterwards: scripts/profile_summary.py).  Usage: bash scripts/profile_round.sh r02
# Each rocprofv3 invocation has the program itself after "--" (python3 <script>), counters are
# collected in their own passes with --kernel-trace only (never with sys/hip/hsa traces).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
# 1. the bench command itself, plain and under the kernel trace
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o b --output-format csv -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# 2. HBM-side traffic of the generator kernel: FETCH_SIZE and WRITE_SIZE in separate passes
GEN="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/gen_fetch -o p --output-format csv -- $GEN > /dev/null 2> $OUT/gen_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/gen_write -o p --output-format csv -- $GEN > /dev/null 2> $OUT/gen_write.err
# 3. the MFMA kernels of the training step: matrix-core busy cycles, then HBM traffic
TR="python3 $R/scripts/train_steps.py 2"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/tr_mfma -o p --output-format csv -- $TR > /dev/null 2> $OUT/tr_mfma.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/tr_fetch -o p --output-format csv -- $TR > /dev/null 2> $OUT/tr_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/tr_write -o p --output-format csv -- $TR > /dev/null 2> $OUT/tr_write.err
rocprofv3 --kernel-trace --stats -d $OUT/tr_trace -o t --output-format csv -- $TR > /dev/null 2> $OUT/tr_trace.err
ls -R $OUT | head -60
