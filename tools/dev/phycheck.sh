timeout -k 10 600 python -m pytest tests/test_gpu_phy.py tests/test_gpu_phy_env.py tests/test_gpu_monitor.py tests/test_gpu_errors.py tests/test_gpu_checkpoint.py -m gpu -x -q > gpurun_out/phy_tests.log 2>&1; tail -15 gpurun_out/phy_tests.log; rm -f gpurun_out/phy_bench.log; for v in phy phy_metrics phy_defrag; do timeout -k 10 120 python bench.py --only $v 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        for k,v in d['sub_records'].items(): print(k, v['value']/1e6, v['launch'])
" >> gpurun_out/phy_bench.log; done; cat gpurun_out/phy_bench.log; timeout -k 10 300 python tools/section_profile.py --phy bmfa --steps 1000 2>/dev/null
