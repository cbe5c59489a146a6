"""Condense gpurun_out/<tag>/ (scripts/profile_round.sh) into the tracked files under profiles/:

    python scripts/profile_summary.py r02

  profiles/<tag>_bench.json, <tag>_bench_under_rocprof.json   the two JSON lines
  profiles/<tag>_bench_kernel_stats.csv                        rocprofv3 --kernel-trace --stats of the bench command
  profiles/<tag>_train_kernel_stats.csv                        ... of scripts/train_steps.py
  profiles/<tag>_pmc_summary.json                              per kernel: calls, mean duration, counters per
                                                               dispatch, MFMA utilisation, HBM GB/s
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / n_se * 4 SIMD ...) is not used: the
portable form here is busy cycles per SIMD over the kernel's shader cycles,
    util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock),  clock = GRBM_GUI_ACTIVE / 8 / duration
(MI355X_MICROARCH.md, 'DVFS give-back').  HBM GB/s = (2 x FETCH_SIZE + WRITE_SIZE) KiB / duration for the
wide (16-byte) streaming kernels -- the guide's x2 correction of FETCH_SIZE on gfx950 -- and
(FETCH_SIZE + WRITE_SIZE) as reported for the generator kernel (4/8-byte accesses: uncalibrated)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
# every pass of scripts/profile_round.sh must have exited 0 (r2: passes that ended in SIGSEGV at
# process exit fed this summary unnoticed)
passes = os.path.join(src, "passes.txt")
if not os.path.exists(passes):
    sys.exit(f"{passes} is missing: run scripts/profile_round.sh {tag} first")
lines = open(passes).read().strip().splitlines()
bad = [ln for ln in lines if " rc=" in ln and not ln.endswith("rc=0")]
if bad or not lines or lines[-1] != "all passes ok":
    sys.exit(f"profile round {tag} is incomplete or has failed passes: {bad or lines[-1:]}")
shutil_passes = os.path.join(dst, f"{tag}_passes.txt")


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


for name, pat in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json")):
    if one(pat):
        shutil.copy(one(pat), os.path.join(dst, f"{tag}_{name}"))
shutil.copy(passes, shutil_passes)
for name, pat in (("bench_kernel_stats.csv", "bench_trace/**/*kernel_stats.csv"),
                  ("train_kernel_stats.csv", "tr2_trace/**/*kernel_stats.csv"),
                  ("train_config3_kernel_stats.csv", "tr3_trace/**/*kernel_stats.csv")):
    if one(pat):
        shutil.copy(one(pat), os.path.join(dst, f"{tag}_{name}"))


def short(n):
    return n.split("(")[0].replace("void ", "").replace("mvn::", "")[:70]


def counters(pattern):
    """kernel -> counter -> mean per dispatch; kernel -> (calls, mean duration ns)"""
    cf, tf = one(pattern + "/**/*counter_collection.csv"), one(pattern + "/**/*kernel_trace.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    if cf:
        for r in csv.DictReader(open(cf)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(list)
    if tf:
        for r in csv.DictReader(open(tf)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k in acc:
        calls = max(len(n[k]), 1)
        out[k] = {c: v / calls for c, v in acc[k].items()}
        if dur[k]:
            out[k]["calls"] = len(dur[k])
            out[k]["duration_ns"] = sum(dur[k]) / len(dur[k])
    return out


summary = {"note": __doc__.split("MFMA utilisation")[0].strip().splitlines()[0], "kernels": {}, "kernels_config3": {}}
for section, pre in (("kernels", "tr2"), ("kernels_config3", "tr3")):
    mf, fe, wr = counters(pre + "_mfma"), counters(pre + "_fetch"), counters(pre + "_write")
    for k in sorted(mf, key=lambda k: -mf[k].get("duration_ns", 0) * mf[k].get("calls", 0)):
        m = mf[k]
        d = m.get("duration_ns")
        if not d or m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) == 0:
            continue
        clock_ghz = m.get("GRBM_GUI_ACTIVE", 0) / 8 / d
        e = {"calls": m["calls"], "duration_us_under_pmc": d / 1e3, "clock_GHz": round(clock_ghz, 3),
             "SQ_VALU_MFMA_BUSY_CYCLES": m["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_BUSY_CYCLES": m.get("SQ_BUSY_CYCLES"),
             "mfma_util": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d * clock_ghz), 4) if clock_ghz else None}
        f, w = fe.get(k, {}), wr.get(k, {})
        if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
            df, dw = f.get("duration_ns", d), w.get("duration_ns", d)
            e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = f["FETCH_SIZE"], w["WRITE_SIZE"]
            e["hbm_GBps_fetch_x2_corrected"] = round((2 * f["FETCH_SIZE"] * 1024 / df + w["WRITE_SIZE"] * 1024 / dw), 1)
        summary[section][k] = e


def step_total(pre):
    """HBM bytes of ONE training step, summed over EVERY kernel of the fetch / write passes (2 x FETCH_SIZE: the guide's
    gfx950 correction for 16-byte streams; WRITE_SIZE as reported), steps = launches of the loss kernel."""
    tot = {}
    for kind in ("fetch", "write"):
        cf = one(f"{pre}_{kind}/**/*counter_collection.csv")
        if not cf:
            return None
        s, steps = 0.0, set()
        for r in csv.DictReader(open(cf)):
            if r["Counter_Name"] == ("FETCH_SIZE" if kind == "fetch" else "WRITE_SIZE"):
                s += float(r["Counter_Value"])
            if "softmax_ce_fwd" in r["Kernel_Name"]:
                steps.add(r["Dispatch_Id"])
        tot[kind] = (s * 1024.0, max(len(steps), 1))
    f, w = tot["fetch"][0] / tot["fetch"][1], tot["write"][0] / tot["write"][1]
    return {"steps_profiled": tot["fetch"][1], "FETCH_SIZE_bytes_per_step_as_reported": f, "WRITE_SIZE_bytes_per_step": w,
            "bytes_per_step_fetch_x2_plus_write": 2 * f + w}


summary["step_totals"] = {"train_config2": step_total("tr2"), "train_config3": step_total("tr3")}
# the generator's step against its dependent chain (scripts/pipe_stamps.py --fold --json): scaled to the product's step
fs = one("fold_stamps.json")
if fs:
    st = json.load(open(fs))
    b = one("bench.json")
    if b:
        try:
            st["product_step_us"] = json.load(open(b))["us_per_sample_step"]
            st["product_over_stamped"] = st["product_step_us"] / st["step_us"]
        except (ValueError, KeyError):
            pass
    json.dump(st, open(os.path.join(dst, f"{tag}_fold_stamps.json"), "w"), indent=1)
gf, gw = counters("gen_fetch"), counters("gen_write")
for k in gf:
    if "gen_" in k and "kernel" in k and k in gw:
        summary["kernels"][k] = {"calls": gf[k].get("calls"), "duration_us_under_pmc": gf[k].get("duration_ns", 0) / 1e3,
                                 "FETCH_SIZE_KiB_as_reported": gf[k].get("FETCH_SIZE"),
                                 "WRITE_SIZE_KiB": gw[k].get("WRITE_SIZE")}
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
