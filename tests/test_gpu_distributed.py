"""The RCCL leg of the multi-GPU path, executed with the hardware a test box has: ONE rank.

`bench.py` launched the way the driver launches its ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) takes
its distributed branch also when WORLD_SIZE is 1: `init_process_group(backend="nccl")` on cuda:0 (RCCL load, communicator
setup), the all-reduce of the int64 statistics vector as a device tensor, `ReduceOp.MAX` on the float64 clock, the barriers
around the timed region, `destroy_process_group`.  With one rank every collective is the identity, which is what the test
asserts; the world-size-2 arithmetic is covered on CPU through gloo (tests/test_distributed_gloo.py, test_bench_launcher.py).
The rank runs in a fresh child process (a process that has touched the GPU is never re-exec'ed)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_nccl_bench_path():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    B, chunk, steps, warm = 4096, 100, 2, 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", str(B), "--chunk", str(chunk),
                        "--steps", str(steps), "--warmup", str(warm), "--no-sub-records", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    c = d["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and d["n_gpus"] == 1
    # SUM over one rank of the device-side int64 vector = orlg_reduce_counters' vector; MAX of the float64 clock = the clock
    assert c["reduced_stats"] == c["local_stats"]
    assert c["clock_max_s"] == c["clock_local_s"] and c["clock_max_s"] > 0
    assert c["local_stats"][9] == B                                      # num_envs
    assert c["local_stats"][0] == B * (chunk * (steps + warm) + 1)       # services_processed: every env, every launch
    assert d["blocking"]["num_envs"] == B and d["blocking"]["services_processed"] == c["local_stats"][0]
    assert 0.0 < d["blocking"]["service_blocking_rate"] < 0.5
    assert d["value"] > 0 and d["scaling"] == "weak"
