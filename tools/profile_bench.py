#!/usr/bin/env python3
"""rocprofv3 passes over ONE bench.py workload (GPU box) -> gpurun_out/profiles_out/<tag>_<name>_{summary.json,kernel_stats.csv}
and a pmc.json entry; `python tools/profile_bench.py --collect` (build container) then merges gpurun_out/profiles_out into
profiles/ (tracked).

    python tools/profile_bench.py <name> [--tag r02] [--steps 3] [--quick]
    name: headline | rmsa_b4096 | phy | phy_metrics | phy_defrag | phy_gn | deeprmsa      (bench.py --only)

Passes (separate rocprofv3 invocations, as MI355X_MICROARCH.md prescribes: kernel trace and counters never together,
FETCH_SIZE and WRITE_SIZE in passes of their own):
    --kernel-trace --stats                                  kernel durations (must agree with bench.py's HIP events)
    --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_{VALU,SALU,LDS,SMEM,VMEM_RD,VMEM_WR}
    --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_{ANY,VALU,SCA,LDS} SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES
    --pmc FETCH_SIZE            --pmc WRITE_SIZE            (--quick skips these two)
Counters are averaged over the LAST `steps` dispatches of the workload's step kernel (the timed launches: steady state) and
divided by the env-steps of one launch.  HBM bytes: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 counts
64 B per 128-B request for wide coalesced reads -- the step kernels move their state in 16-B-per-lane rows).
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "profiles_out")
# bench.py --only name -> (kernel-name pattern of the step kernel, pmc.json key, batch, env-steps per launch per env)
WORK = {
    "headline": ("orlg_rmsa_group_kernel", "rmsa_b65536", 65536, 1000),
    "rmsa_b4096": ("orlg_rmsa_kernel_ff", "rmsa_b4096", 4096, 1000),
    "phy": ("orlg_phy_kernel", "phy", 4096, 1000),
    "phy_metrics": ("orlg_phy_kernel", "phy_metrics", 4096, 1000),
    "phy_defrag": ("orlg_phy_kernel", "phy_defrag", 4096, 1000),
    "phy_gn": ("orlg_phy_kernel", "phy_gn", 4096, 1000),
    "deeprmsa": ("orlg_rmsa_", "deeprmsa", 32768, 1),
}
PMC_PASSES = {
    "inst": "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR",
    "busy": "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
}


def collect():
    dst = os.path.join(ROOT, "profiles")
    pj = os.path.join(dst, "pmc.json")
    cur = json.load(open(pj)) if os.path.exists(pj) else {
        "comment": "per-env-step rocprofv3 counters of bench.py's workloads (tools/profile_bench.py): instruction counts are "
                   "wave-instructions, SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* quad-cycles, hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE "
                   "(KiB -> bytes; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md). bench.py reads the "
                   "entry of its workload for roofline.traffic and roofline.valu_issue.", "entries": {}}
    n = 0
    for f in sorted(glob.glob(os.path.join(OUT, "*"))):
        base = os.path.basename(f)
        if base.endswith("_pmc_entry.json"):
            e = json.load(open(f))
            cur["entries"][e["key"]] = e["entry"]
        else:
            shutil.copy(f, os.path.join(dst, base))
        n += 1
    json.dump(cur, open(pj, "w"), indent=1)
    print("collected", n, "files into profiles/; pmc.json keys:", sorted(cur["entries"]))


def run(cmd, log):
    with open(log, "w") as f:
        r = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        print("FAILED:", " ".join(cmd), "\n", open(log).read()[-3000:])
        sys.exit(1)


def main():
    if "--collect" in sys.argv:
        return collect()
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=sorted(WORK))
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--quick", action="store_true", help="skip the FETCH_SIZE / WRITE_SIZE passes")
    ap.add_argument("--extra", default="", help="extra bench.py arguments, e.g. '--step-kernel wave'")
    ap.add_argument("--suffix", default="", help="appended to the output names and the pmc.json key")
    args = ap.parse_args()
    kpat, key, B, chunk = WORK[args.name]  # (kernel pattern: substring of the kernel name)
    key += args.suffix
    os.makedirs(OUT, exist_ok=True)
    scratch = os.path.join(ROOT, "gpurun_out", f"prof_{args.tag}_{args.name}{args.suffix}")
    shutil.rmtree(scratch, ignore_errors=True)
    os.makedirs(scratch)
    os.environ["TMPDIR"] = "/tmp"
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--only", args.name, "--steps", str(args.steps), "--warmup", "1",
             "--no-cpu-baseline"] + (["--no-pcie-loop"] if args.name == "deeprmsa" else []) + args.extra.split()
    # the un-profiled line first (never compare a profiled arm with an un-profiled one: both are recorded)
    run(bench, os.path.join(scratch, "bench.json"))
    bench_line = json.loads([ln for ln in open(os.path.join(scratch, "bench.json")) if ln.startswith("{")][-1])
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(scratch, "kt"), "--"] + bench,
        os.path.join(scratch, "kt.log"))
    passes = {k: v for k, v in PMC_PASSES.items() if not (args.quick and k in ("fetch", "write"))}
    for pname, ctrs in passes.items():
        run(["rocprofv3", "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", os.path.join(scratch, "pmc_" + pname), "--"] + bench,
            os.path.join(scratch, f"pmc_{pname}.log"))
    # ---- kernel trace
    stats_csv = sorted(glob.glob(os.path.join(scratch, "kt", "**", "*kernel_stats.csv"), recursive=True))[-1]
    trace_csv = sorted(glob.glob(os.path.join(scratch, "kt", "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(trace_csv)) if kpat in r["Kernel_Name"] and "reset" not in r["Kernel_Name"]]
    names = collections.Counter(r["Kernel_Name"] for r in rows)
    summary = {"workload": args.name, "bench_args": bench[2:], "bench_unprofiled": bench_line, "kernel_pattern": kpat,
               "kernels_matched": dict(names), "kernel_stats": list(csv.DictReader(open(stats_csv)))[:8]}
    per_kernel = {}
    for kn in names:
        rr = [r for r in rows if r["Kernel_Name"] == kn]
        timed = rr[-args.steps:] if args.name != "deeprmsa" else rr[len(rr) // 2:]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
        last = timed[-1]
        per_kernel[kn] = {"dispatches": len(rr), "timed_dispatches": len(d), "avg_ns_timed": sum(d) / len(d),
                          "vgpr": last.get("VGPR_Count"), "accum_vgpr": last.get("Accum_VGPR_Count"), "sgpr": last.get("SGPR_Count"),
                          "lds": last.get("LDS_Block_Size"), "scratch": last.get("Scratch_Size"),
                          "workgroup": last.get("Workgroup_Size_X"), "grid": last.get("Grid_Size_X")}
    summary["step_kernels"] = per_kernel
    # ---- counters: last `steps` dispatches of every matched kernel, summed over the kernels (deeprmsa: step + observation)
    pmc = collections.defaultdict(float)
    for pname in passes:
        f = sorted(glob.glob(os.path.join(scratch, "pmc_" + pname, "**", "*counter_collection.csv"), recursive=True))[-1]
        per = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> per-dispatch values
        disp = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if (kpat in kn or (args.name == "deeprmsa" and "orlg_deeprmsa_obs" in kn)) and "reset" not in kn:
                disp[(kn, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (kn, _), cs in disp.items():
            for c, v in cs.items():
                per[kn][c].append(v)
        for kn, cs in per.items():
            for c, v in cs.items():
                timed = v[-args.steps:] if args.name != "deeprmsa" else v[len(v) // 2:]
                pmc[c] += sum(timed) / len(timed)
    env_steps = B * chunk
    per_env = {c: v / env_steps for c, v in pmc.items()}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        per_env["hbm_bytes"] = (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024 / env_steps
        per_env["hbm_fetch_bytes_x2"] = 2 * pmc["FETCH_SIZE"] * 1024 / env_steps
        per_env["hbm_write_bytes"] = pmc["WRITE_SIZE"] * 1024 / env_steps
    summary["pmc_per_launch"] = dict(pmc)
    summary["per_env_step"] = per_env
    entry = {"batch": B, "env_steps_per_launch_per_env": chunk, "per_env_step": per_env,
             "source": f"profiles/{args.tag}_{key}_summary.json"}
    if "SQ_ACTIVE_INST_VALU" in pmc and pmc.get("SQ_WAVE_CYCLES"):
        # waves resident per SIMD from the launch shape the library reports (bench.py "launch": "... block=704 lds=158336")
        import re
        rec = bench_line if "launch" in bench_line else next(iter(bench_line.get("sub_records", {}).values()), {})
        m = re.search(r"block=(\d+) lds=(\d+)", rec.get("launch", ""))
        wg, lds = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
        waves_wg = wg // 64 if wg else 0
        wg_per_cu = min(160 * 1024 // lds if lds else 32, 32 // max(1, waves_wg)) if waves_wg else 0
        k0 = next(iter(per_kernel.values()))
        vg = int(k0.get("vgpr") or 0) + int(k0.get("accum_vgpr") or 0)
        reg_waves = min(8, 512 // (((vg + 7) // 8) * 8)) if vg else 8   # MI355X_MICROARCH.md: 512 registers per lane and SIMD
        entry["waves_per_simd"] = min(waves_wg * wg_per_cu / 4.0, float(reg_waves))
        entry["valu_busy"] = pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"] * entry["waves_per_simd"]
        entry["valu_cycles_per_inst"] = 4.0 * pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_INSTS_VALU"]
        summary["valu"] = {k: entry[k] for k in ("waves_per_simd", "valu_busy", "valu_cycles_per_inst")}
    base = os.path.join(OUT, f"{args.tag}_{key}")
    json.dump(summary, open(base + "_summary.json", "w"), indent=1)
    shutil.copy(stats_csv, base + "_kernel_stats.csv")
    json.dump({"key": key, "entry": entry}, open(base + "_pmc_entry.json", "w"), indent=1)
    print(json.dumps({"key": key, "kernels": per_kernel, "per_env_step": {k: round(v, 3) for k, v in per_env.items()},
                      "valu": summary.get("valu"), "bench_value": bench_line.get("value") or
                      next(iter(bench_line.get("sub_records", {}).values()), {}).get("value")}, indent=1))


if __name__ == "__main__":
    main()
