"""bench.py --gpus N starts its own ranks (fresh child processes, torch.distributed.run) when no launcher did: checked on CPU
with --dry-run, where the ranks report their shards through gloo instead of stepping environments."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout)


def test_self_launch_two_ranks_dry_run():
    r = _run(["--gpus", "2", "--dry-run", "--batch", "1024", "--mixed"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout     # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["num_envs"] == 2048
    shards = sorted(d["shards"], key=lambda s: s["rank"])
    assert [s["rank"] for s in shards] == [0, 1] and all(s["world"] == 2 for s in shards)
    # contiguous seed ranges: the environments a single process with batch 2048 would run
    assert shards[0]["first_seed"] == 10 and shards[0]["last_seed"] == 10 + 1023
    assert shards[1]["first_seed"] == 10 + 1024 and shards[1]["last_seed"] == 10 + 2047
    # --mixed: one topology group per rank (rank mod 3)
    assert shards[0]["topology"].startswith("nsfnet") and shards[1]["topology"].startswith("jpn12")


def test_single_rank_dry_run_and_world_mismatch():
    r = _run(["--dry-run", "--batch", "64"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["num_envs"] == 64 and len(d["shards"]) == 1
    # a launcher that set WORLD_SIZE to something else than --gpus is an error, not a silent single-GPU run
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
