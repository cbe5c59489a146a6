"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files: python scripts/pmc_summary.py DIR..."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[(k, f)] += 1
names = sorted({c for v in acc.values() for c in v})
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_LDS_IDX_ACTIVE", 0))):
    n = max(c for (kk, _), c in calls.items() if kk == k)
    print(f"{k}  dispatches={n}")
    for c in names:
        if c in v:
            print(f"    {c:32s} {v[c] / n:16.0f} per dispatch")
