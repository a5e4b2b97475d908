set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_mix
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  tag=mix
  rm -rf gpurun_out/pmc_mix/$tag
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_mix/$tag -- python bench.py --only ${1:-phy} > gpurun_out/pmc_mix/$tag.log 2>&1 || { tail -5 gpurun_out/pmc_mix/$tag.log; }
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_mix/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "orlg_phy_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        v = sorted(v)
        print(k, "n=%d" % len(v), "per-env-step=%.2f" % (v[len(v)//2] / (4096 * 250)))
PY
