#!/bin/bash
# Profiles of one round, written under gpurun_out/<tag>/ on the GPU box (copy the summaries
# into profiles/ afterwards: scripts/profile_summary.py).  Usage: bash scripts/profile_round.sh r03
# Each rocprofv3 invocation has the program itself after "--" (python3 <script>), counters are
# collected in their own passes with --kernel-trace only (never with sys/hip/hsa traces).
# Every pass must EXIT 0: its code goes to <tag>/passes.txt and the script stops at the first
# pass that does not (r2: nine passes ended in SIGSEGV at process exit and were walked past;
# profile_summary.py refuses a directory whose passes.txt is missing or holds a failure).
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/passes.txt
cd /tmp && export TMPDIR=/tmp

pass() {  # pass <name> <stdout file> <command ...>
  local name=$1 out=$2
  shift 2
  "$@" > "$out" 2> "$OUT/$name.err"
  local rc=$?
  echo "$name rc=$rc" | tee -a $OUT/passes.txt
  if [ $rc -ne 0 ]; then
    echo "profile_round: pass '$name' exited $rc -- stopping (see $OUT/$name.err)" | tee -a $OUT/passes.txt
    tail -5 "$OUT/$name.err"
    exit $rc
  fi
}

BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
# 1. the bench command itself, plain and under the kernel trace
pass bench $OUT/bench.json $BENCH
pass bench_under_rocprof $OUT/bench_under_rocprof.json \
  rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o b --output-format csv -- $BENCH
# 2. HBM-side traffic of the generator kernel: FETCH_SIZE and WRITE_SIZE in separate passes
GEN="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --no-extras"
pass gen_fetch /dev/null rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/gen_fetch -o p --output-format csv -- $GEN
pass gen_write /dev/null rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/gen_write -o p --output-format csv -- $GEN
# 3. the MFMA kernels of the training steps (config 2, then the conditioned config 3):
#    matrix-core busy cycles, then HBM traffic, then the kernel trace
for CFGN in 2 3; do
  TR="python3 $R/scripts/train_steps.py 2 --config $CFGN"
  P=tr$CFGN
  pass ${P}_mfma /dev/null rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${P}_mfma -o p --output-format csv -- $TR
  pass ${P}_fetch /dev/null rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${P}_fetch -o p --output-format csv -- $TR
  pass ${P}_write /dev/null rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${P}_write -o p --output-format csv -- $TR
  pass ${P}_trace /dev/null rocprofv3 --kernel-trace --stats -d $OUT/${P}_trace -o t --output-format csv -- $TR
done
# 4. the generator's step against its dependent chain: in-kernel stamps of the diagnostic build
if [ -f $R/movenet_amd/lib/libmovenet_hip_stamps.so ]; then
  export MOVENET_HIP_LIB=$R/movenet_amd/lib/libmovenet_hip_stamps.so
  pass fold_stamps $OUT/fold_stamps.txt python3 $R/scripts/pipe_stamps.py --fold --json $OUT/fold_stamps.json
  unset MOVENET_HIP_LIB
fi
echo "all passes ok" | tee -a $OUT/passes.txt
