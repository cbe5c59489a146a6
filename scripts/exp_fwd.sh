# Timing builds of the strip forward (python -m movenet_amd.csrc.build --stamps --exp=11 ... --exp=14; the bf16 x 3
# form: --exp=21 ... --exp=25, EXPS="0 21 22 23 24 25" bash scripts/exp_fwd.sh):
# per-kernel averages of three training steps under rocprofv3, one library after the other.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in ${EXPS:-0 11 12 13 14}; do
  if [ $e = 0 ]; then unset MOVENET_HIP_LIB; else export MOVENET_HIP_LIB=$R/movenet_amd/lib/libmovenet_hip_exp$e.so; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/expf$e -o e --output-format csv -- python3 $R/scripts/train_steps.py 3 > $R/gpurun_out/expf$e.log 2>&1
  f=$(find $R/gpurun_out/expf$e -name '*kernel_stats.csv' | head -1)
  echo "== exp $e"
  test -n "$f" && grep -E "fused_layer64s|bwd_dx_wgfg64|bwd_dz_wgrs64|dense_strip" "$f" | cut -d, -f1-4 | cut -c1-120
done
