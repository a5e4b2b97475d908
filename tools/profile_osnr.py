#!/usr/bin/env python3
"""rocprofv3 summary of the stand-alone GN-model kernel (orlg_gn_osnr_kernel) on tools/bench_osnr.py's workload (GPU box):
kernel-trace + one SQ counter pass -> gpurun_out/profiles_out/<tag>_osnr_{summary.json,kernel_stats.csv}.
The kernel is fp64 / transcendental bound: reported are the kernel time per batch, the VALU wave-instructions per check and
per evaluated asinh (2 per link x interferer; the reference evaluates 2 per span x interferer), the VALU-issue fraction
(256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles, as bench.py's valu_issue block) and an fp64 rate estimate from the instruction
count (one fp64 operation per lane per VALU instruction is an upper bound on useful work, not a flop count)."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "profiles_out")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    os.makedirs(OUT, exist_ok=True)
    scratch = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_osnr")
    shutil.rmtree(scratch, ignore_errors=True)
    os.makedirs(scratch)
    os.environ["TMPDIR"] = "/tmp"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_osnr.py")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(scratch, "kt"), "--"] + cmd,
                   stdout=open(os.path.join(scratch, "kt.log"), "w"), stderr=subprocess.STDOUT, check=True)
    subprocess.run(["rocprofv3", "--pmc", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                    "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD", "SQ_BUSY_CYCLES", "--output-format", "csv", "-d", os.path.join(scratch, "pmc"), "--"] + cmd,
                   stdout=open(os.path.join(scratch, "pmc.log"), "w"), stderr=subprocess.STDOUT, check=True)
    stats_csv = sorted(glob.glob(os.path.join(scratch, "kt", "**", "*kernel_stats.csv"), recursive=True))[-1]
    trace_csv = sorted(glob.glob(os.path.join(scratch, "kt", "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [x for x in csv.DictReader(open(trace_csv)) if "orlg_gn_osnr_kernel" in x["Kernel_Name"]]
    d = [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in rows]
    pmc = {}
    f = sorted(glob.glob(os.path.join(scratch, "pmc", "**", "*counter_collection.csv"), recursive=True))[-1]
    n = 0
    for x in csv.DictReader(open(f)):
        if "orlg_gn_osnr_kernel" in x["Kernel_Name"]:
            pmc[x["Counter_Name"]] = pmc.get(x["Counter_Name"], 0.0) + float(x["Counter_Value"])
    n = len(rows)
    per_launch = {k: v / n for k, v in pmc.items()}
    M, links, services = line["checks"], 5, 268
    asinh = M * links * (services - 1) * 2
    avg_s = sum(d) / len(d) * 1e-9
    valu = per_launch.get("SQ_INSTS_VALU", 0.0)
    summary = {"bench": line, "kernel": rows[0]["Kernel_Name"], "launches": n, "avg_kernel_ns": sum(d) / len(d),
               "vgpr": rows[0].get("VGPR_Count"), "pmc_per_launch": per_launch,
               "valu_insts_per_check": valu / M, "asinh_evaluated_per_launch": asinh, "valu_insts_per_asinh": valu / asinh,
               "asinh_per_s": asinh / avg_s,
               "valu_issue": {"achieved_G_wave_insts_per_s": valu / avg_s / 1e9, "peak": 256 * 4 * 2.4 / 4.0,
                              "frac": valu / avg_s / 1e9 / (256 * 4 * 2.4 / 4.0)},
               "fp64_upper_bound_TFLOPs": valu * 64 / avg_s / 1e12,
               "note": "fp64_upper_bound = one fp64 operation per lane per VALU wave-instruction (fma counted once); MI355X fp64 vector peak 78.6 TFLOP/s counts an fma as two"}
    json.dump(summary, open(os.path.join(OUT, f"{tag}_osnr_summary.json"), "w"), indent=1)
    shutil.copy(stats_csv, os.path.join(OUT, f"{tag}_osnr_kernel_stats.csv"))
    print(json.dumps({k: summary[k] for k in ("avg_kernel_ns", "valu_insts_per_check", "valu_insts_per_asinh", "asinh_per_s", "valu_issue", "fp64_upper_bound_TFLOPs")}, indent=1))


if __name__ == "__main__":
    main()
