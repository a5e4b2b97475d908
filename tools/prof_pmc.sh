# PMC passes (separate from kernel-trace runs, as the MI355X guide prescribes); usage: bash tools/prof_pmc.sh [bench args]
set -e
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 500 --warmup 500 $@"
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -- python bench.py $ARGS > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc/*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "orlg_rmsa_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = sorted(v)
            print(d.split("/")[-1], k, "n=%d" % len(v), "median=%.4g" % v[len(v)//2], "max=%.4g" % v[-1])
PY
