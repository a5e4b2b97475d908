set -e
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench2.log 2>&1
tail -2 gpurun_out/bench2.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python bench.py --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1
find gpurun_out/prof -name "*stats*" | head
