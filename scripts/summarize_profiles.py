"""Condense the rocprofv3 outputs of scripts/profile_round.sh:  summarize_profiles.py <dir> <timed_steps> <pmc_steps>
  kernel_trace_timed_region.json : k_block_step average duration over the timed region's launches only (sweeps are
                                   delimited by their k_prepare launch; the --stats average also covers the warm-up)
  pmc_{fetch,write}_summary.csv  : counter sums per kernel over the whole run, plus per-launch averages of k_block_step
                                   over the last <pmc_steps> sweeps (the steady state the timed region runs in)
"""
import csv
import glob
import json
import os
import sys

root, timed_steps, pmc_steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
wname = sys.argv[4] if len(sys.argv) > 4 else "config2"
traffic = {}


def find(sub, pat):
    hits = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def is_step(name):
    """The step kernel: k_block_step (one block per launch) or k_group_step (grouped launches)."""
    return "k_block_step" in name or "k_group_step" in name


def short(name):
    return name.split("(")[0].replace("void ", "")


kt = find("ktrace", "*kernel_trace.csv")
if kt:
    rows = []
    with open(kt) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    sweeps, cur = [], None
    for s, e, name in rows:
        if "k_prepare" in name:
            cur = []
            sweeps.append(cur)
        elif is_step(name) and cur is not None:
            cur.append(e - s)
    allk = [d for sw in sweeps for d in sw]
    timed = [d for sw in sweeps[-timed_steps:] for d in sw]
    out = {"kernel": "k_block_step", "sweeps_in_trace": len(sweeps), "launches_all": len(allk),
           "avg_ns_all": sum(allk) / max(1, len(allk)), "timed_sweeps": timed_steps, "launches_timed_region": len(timed),
           "avg_ns_timed_region": sum(timed) / max(1, len(timed))}
    try:
        b = json.loads(open(os.path.join(root, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
        out["kernel"] = b["roofline"].get("kernel", "k_block_step")
        out["blocks_per_launch"] = b["config"].get("blocks_per_launch", 1)
        out["bench_avg_launch_us_same_run"] = b["roofline"]["avg_launch_us"]
        out["bench_value_same_run"] = b["value"]
    except Exception as ex:                                     # noqa: BLE001
        out["bench_json_error"] = str(ex)
    json.dump(out, open(os.path.join(root, "kernel_trace_timed_region.json"), "w"), indent=1)
    print(out)
    st = find("ktrace", "*kernel_stats.csv")
    if st:
        open(os.path.join(root, "kernel_stats.csv"), "w").write(open(st).read())

for tag, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    cc = find("pmc_" + tag, "*counter_collection.csv")
    if not cc:
        continue
    rows = []
    with open(cc) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    tot = {}
    sweeps, cur = [], None
    for _, name, v in rows:
        k = short(name)
        t = tot.setdefault(k, [0, 0.0])
        t[0] += 1
        t[1] += v
        if "k_prepare" in name:
            cur = []
            sweeps.append(cur)
        elif is_step(name) and cur is not None:
            cur.append(v)
    last = [v for sw in sweeps[-pmc_steps:] for v in sw]
    with open(os.path.join(root, f"pmc_{tag}_summary.csv"), "w") as fh:
        fh.write(f"kernel,dispatches,{counter}_sum_KB,{counter}_per_dispatch_KB\n")
        for k, (n, v) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            fh.write(f"\"{k}\",{n},{v:.1f},{v / n:.2f}\n")
        fh.write(f"\"step kernel: last {pmc_steps} sweeps (steady state)\",{len(last)},{sum(last):.1f},{sum(last) / max(1, len(last)):.2f}\n")
    print(tag, "steady-state per dispatch KB:", sum(last) / max(1, len(last)), "dispatches", len(last))
    traffic[tag] = sum(last) / max(1, len(last))

# HBM bytes per k_block_step launch for bench.py's roofline.traffic: FETCH_SIZE x 2 (the guide's gfx950 correction,
# calibrated on k_xpx which reads X exactly once) + WRITE_SIZE, in KB -> bytes; keyed by the configuration it was measured on
if "fetch" in traffic and "write" in traffic:
    try:
        b = json.loads(open(os.path.join(root, "bench_under_pmc_fetch.json")).read().strip().splitlines()[-1])
        cfg = b["config"]
        row = {"config": {"workload": cfg["name"], "n": cfg["n"], "p": cfg["p"], "block_size": cfg["block_size"], "storage": cfg["storage"],
                          "n_gpus": b["n_gpus"], "pi_fixed": (0.95 if "pifixed" in wname else None), "variant": cfg.get("variant"),
                          **({"blocks_per_launch": cfg["blocks_per_launch"]} if cfg.get("blocks_per_launch", 1) >= 2 else {})},
               "bytes_per_launch": (traffic["fetch"] * 2 + traffic["write"]) * 1024.0,
               "fetch_KB_per_launch": traffic["fetch"], "write_KB_per_launch": traffic["write"],
               "algorithmic_bytes_per_launch": b["roofline"]["bytes_per_launch"],
               "source": f"profiles/<round>_{wname}_pmc_fetch_summary.csv + _pmc_write_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, last {pmc_steps} sweeps)"}
        json.dump(row, open(os.path.join(root, "traffic_row.json"), "w"), indent=1)
        print(row)
    except Exception as ex:                                     # noqa: BLE001
        print("traffic row failed:", ex)
