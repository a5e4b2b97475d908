set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pcs
rocprofv3 --help 2>&1 | grep -i -A3 "pc-sampling" | head -40 > gpurun_out/pcs/help.txt || true
cat gpurun_out/pcs/help.txt
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1000 --output-format csv -d gpurun_out/pcs/run -- python bench.py --only phy > gpurun_out/pcs/run.log 2>&1 || { tail -20 gpurun_out/pcs/run.log; exit 0; }
tail -3 gpurun_out/pcs/run.log
ls -la gpurun_out/pcs/run/*/ | head
