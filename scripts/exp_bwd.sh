# Timing builds of the fused layer backward (python -m movenet_amd.csrc.build --stamps --exp=71 ... --exp=74;
# EXPS="0 71 72 73 74" bash scripts/exp_bwd.sh): per-kernel averages of three training steps under rocprofv3,
# one library after the other.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in ${EXPS:-0 71 72 73 74}; do
  if [ $e = 0 ]; then unset MOVENET_HIP_LIB; else export MOVENET_HIP_LIB=$R/movenet_amd/lib/libmovenet_hip_exp$e.so; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/expb$e -o e --output-format csv -- python3 $R/scripts/train_steps.py 3 ${CFG:+--config $CFG} > $R/gpurun_out/expb$e.log 2>&1
  f=$(find $R/gpurun_out/expb$e -name '*kernel_stats.csv' | head -1)
  echo "== exp $e"
  test -n "$f" && grep -E "fused_layer64s|bwd_layer64|bwd_dx_wgfg64|bwd_dz_wgrs64|bwd_dctx" "$f" | cut -d, -f1-4 | cut -c1-140
done
