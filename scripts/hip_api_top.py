"""Longest HIP API calls of a rocprofv3 --hip-trace run: python scripts/hip_api_top.py <dir> [n]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"])))
rows.sort(reverse=True)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = min(r[2] for r in rows) if rows else 0
for d, fn, st in rows[:n]:
    print(f"{d / 1e6:9.3f} ms  {fn:40s} at +{(st - t0) / 1e6:10.3f} ms")
import collections
tot = collections.Counter()
cnt = collections.Counter()
for d, fn, _ in rows:
    tot[fn] += d
    cnt[fn] += 1
print("--- totals")
for fn, d in tot.most_common(15):
    print(f"{d / 1e6:9.3f} ms  {cnt[fn]:7d} calls  {fn}")
