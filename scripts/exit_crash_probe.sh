#!/bin/bash
# Which launch form leaves the process unable to exit cleanly under rocprofv3?  
# Runs the generator-only bench pass under the kernel trace three ways and records exit codes.
# (r3 result: only the cooperative form crashes; profiles/r03_exit_crash.md.  Diagnostic -- not part
# of a profile round: the first pass is EXPECTED to end in SIGSEGV at process exit.)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_exit
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MOVENET_BENCH_DUMP_MAPS=1
GEN="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --no-extras"
MOVENET_PIPE_COOPERATIVE_LAUNCH=1 rocprofv3 --kernel-trace --stats -d $OUT/coop -o p --output-format csv -- $GEN > $OUT/coop.json 2> $OUT/coop.err
echo "cooperative rc=$?" | tee -a $OUT/rc.txt
rocprofv3 --kernel-trace --stats -d $OUT/plain -o p --output-format csv -- $GEN > $OUT/plain.json 2> $OUT/plain.err
echo "plain rc=$?" | tee -a $OUT/rc.txt
rocprofv3 --kernel-trace --stats -d $OUT/stream -o p --output-format csv -- $GEN --variant 2 > $OUT/stream.json 2> $OUT/stream.err
echo "stream(variant 2) rc=$?" | tee -a $OUT/rc.txt
