"""Copy the summaries scripts/profile_round3.sh left under gpurun_out/<tag>/ into profiles/<round>_* (tracked) and merge the
measured HBM traffic rows into profiles/traffic.json (one row per configuration; a newer measurement replaces the older one).
    python scripts/collect_profiles.py r03 r03"""
import glob
import json
import os
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for f in ("gpu_tests.log", "bench_default.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{rnd}_{f}"))
for f in glob.glob(os.path.join(src, "bench_*.json")):
    name = os.path.basename(f)
    if name != "bench_default.json":
        shutil.copy(f, os.path.join(dst, f"{rnd}_{name}"))
rows = json.load(open(os.path.join(dst, "traffic.json"))) if os.path.exists(os.path.join(dst, "traffic.json")) else []
for wdir in sorted(glob.glob(os.path.join(src, "*", ""))):
    w = os.path.basename(os.path.dirname(wdir))
    for f in ("kernel_stats.csv", "kernel_trace_timed_region.json", "pmc_fetch_summary.csv", "pmc_write_summary.csv", "bench_under_rocprof.json"):
        if os.path.exists(os.path.join(wdir, f)):
            shutil.copy(os.path.join(wdir, f), os.path.join(dst, f"{rnd}_{w}_{f}"))
    tr = os.path.join(wdir, "traffic_row.json")
    if os.path.exists(tr):
        row = json.load(open(tr))
        row["source"] = row["source"].replace("<round>", rnd)
        rows = [r for r in rows if r["config"] != row["config"]] + [row]
json.dump(rows, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("traffic rows:", len(rows))
