#!/usr/bin/env python3
"""Copy the judged parts of a tools/profile_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
    python tools/collect_profile.py <tag> <name>      e.g.  r01c r01_final
writes profiles/<name>_summary.json, profiles/<name>_kernel_stats.csv, profiles/<name>_kernel_trace_step.csv and the
matching entry of profiles/traffic.json (HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "profile_" + tag)
dst = os.path.join(ROOT, "profiles")
summary = json.load(open(os.path.join(src, "summary.json")))
json.dump(summary, open(os.path.join(dst, name + "_summary.json"), "w"), indent=1)
stats = sorted(glob.glob(os.path.join(src, "kt", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join(dst, name + "_kernel_stats.csv"))
trace = sorted(glob.glob(os.path.join(src, "kt", "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(trace)))
keep = [r for r in rows if "orlg_rmsa" in r["Kernel_Name"]]
with open(os.path.join(dst, name + "_kernel_trace_step.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)
b = summary["bench"]
cfg = b["config"]
key = f"rmsa_nsfnet320_B{cfg['batch_per_gpu']}_chunk{cfg['chunk']}_{cfg['stats_level']}_{cfg['policy']}"
tpath = os.path.join(dst, "traffic.json")
tj = json.load(open(tpath)) if os.path.exists(tpath) else {"entries": {}}
h = summary["hbm_traffic_bytes_per_launch"]
tj["entries"][key] = {"traffic_bytes_per_launch": h["total_x2"], "fetch_raw": h["fetch_raw"], "write": h["write"],
                      "source": f"profiles/{name}_summary.json"}
json.dump(tj, open(tpath, "w"), indent=1)
print("wrote", name, key, h)
