# Counter passes over three training steps for the layer kernels (separate --pmc passes, kernel trace only).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcb$i -o p --output-format csv -- python3 $R/scripts/train_steps.py 2 ${CFG:+--config $CFG} > $R/gpurun_out/pmcb$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for i in (1,2,3):
    f=glob.glob(f'{R}/gpurun_out/pmcb{i}/*counter_collection.csv')
    if not f: print('no counters for pass', i); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n=r['Kernel_Name']
        if 'bwd_layer64' in n or 'fused_layer64s_bf3' in n:
            acc[n[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for n,c in acc.items():
        print(n, {k: round(sum(v)/len(v)) for k,v in c.items()})
PY
