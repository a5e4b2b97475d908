set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_obs; rm -rf gpurun_out/pmc_obs/*
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_obs/$tag -- python bench.py --only deeprmsa > gpurun_out/pmc_obs/$tag.log 2>&1 || { tail -5 gpurun_out/pmc_obs/$tag.log; }
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_obs/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        kn = "obs" if "obs_kernel" in r["Kernel_Name"] else "step" if "group_kernel" in r["Kernel_Name"] else None
        if kn: agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn in agg:
        for k, v in sorted(agg[kn].items()):
            v = sorted(v)
            print(kn, k, "n=%d" % len(v), "per-env=%.2f" % (v[len(v)//2] / 32768))
PY
