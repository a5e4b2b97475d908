# instruction-mix PMC pass; usage: bash tools/prof_pmc_quick.sh <tag> [bench args]
set -e
tag=$1; shift
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 500 --warmup 500 $@"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc/$tag -- python bench.py $ARGS > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc/${tag}_b -- python bench.py $ARGS > gpurun_out/pmc_${tag}_b.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_b.log; exit 1; }
python - $tag <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in (f"gpurun_out/pmc/{tag}", f"gpurun_out/pmc/{tag}_b"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "orlg_rmsa_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = sorted(v)
            print(tag, k, "n=%d" % len(v), "median=%.4g" % v[len(v)//2], "per-env-step=%.1f" % (v[len(v)//2] / (4096 * 250)))
PY
