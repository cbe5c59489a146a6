#!/bin/bash
# Same-box A/B against a reference build of the library (movenet_amd/lib/libmovenet_hip_ref.so, built
# by hand from an earlier commit; not tracked): headline generator and config-5 fp16 generator.
R=${GRAFT_REPO_ROOT:-/root/repo}
REF=$R/movenet_amd/lib/libmovenet_hip_ref.so
one() { python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   headline us/step', round(d['us_per_sample_step'],3), 'launch ms', round(d['roofline']['avg_launch_ms'],2))"; }
for i in 1 2; do
  echo "== reference"; MOVENET_HIP_LIB=$REF one; MOVENET_HIP_LIB=$REF python3 $R/scripts/bench_config5.py --fp16-only | grep us_per_step
  echo "== current"; one; python3 $R/scripts/bench_config5.py --fp16-only | grep us_per_step
done
