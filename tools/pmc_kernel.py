#!/usr/bin/env python3
"""Per-launch averages of rocprofv3 PMC counters for kernels whose name contains a pattern.
usage: python tools/pmc_kernel.py <dir with *counter_collection.csv> <name pattern> [env-steps per launch]"""
import collections, csv, glob, json, sys
d, pat = sys.argv[1], sys.argv[2]
per = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    v = sorted(v)
    big = [x for x in v if x > 0.75 * v[-1]] or v
    out[k] = round(sum(big) / len(big) / per, 2)
print(json.dumps(out))
