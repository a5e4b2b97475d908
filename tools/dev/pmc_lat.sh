# average memory latencies of the phy kernel: in-flight levels / instruction counts
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_lat
rocprofv3 -L > gpurun_out/pmc_lat/counters.txt 2>&1 || true
grep -o "SQ_INST_LEVEL_[A-Z_]*\|SQ_WAIT_INST_[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_WAIT_[A-Z_]*\|SQ_LEVEL_WAVES\|SQ_IFETCH[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_THREAD_CYCLES_VALU\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INSTS_FLAT[A-Z_]*" gpurun_out/pmc_lat/counters.txt | sort -u | tr '\n' ' ' > gpurun_out/pmc_lat/sq_names.txt
cat gpurun_out/pmc_lat/sq_names.txt
for pass in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_lat/$tag -- python bench.py --only phy > gpurun_out/pmc_lat/$tag.log 2>&1 || { tail -5 gpurun_out/pmc_lat/$tag.log; }
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_lat/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "orlg_phy_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        v = sorted(v)
        print(k, "n=%d" % len(v), "median=%.5g" % v[len(v)//2], "per-env-step=%.2f" % (v[len(v)//2] / (4096 * 250)))
PY
