#!/usr/bin/env python3
"""Throughput of the RMSA step (NSFNET-320, load 50, SAP-FF, full statistics, 1000 steps per launch: the headline workload) against
the batch size -- `bench.py --only headline --batch B` for each B, the kernel "auto" picks (GPU box).  Prints one JSON line.
usage: python tools/exp_batch_sweep.py [--batches 1024,4096,...]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1024,2048,4096,8192,16384,32768,65536,131072,262144")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    rows = []
    for b in [int(x) for x in args.batches.split(",")]:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only", "headline", "--batch", str(b), "--steps", str(args.steps),
                            "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        if not line:
            rows.append({"batch": b, "error": p.stderr[-300:]})
            continue
        d = json.loads(line[-1])
        rows.append({"batch": b, "M_env_steps_per_s": round(d["value"] / 1e6, 1), "ms_per_launch": round(d["ms_per_step"], 3),
                     "kernel": d["roofline"].get("kernel"), "service_blocking_rate": round(d["blocking"]["service_blocking_rate"], 4)})
        print(rows[-1], file=sys.stderr, flush=True)
    print(json.dumps({"workload": "RMSA NSFNET-320 load 50 SAP-FF stats=full, 1000 env-steps per launch", "rows": rows}))


if __name__ == "__main__":
    main()
