#!/usr/bin/env python3
"""bench.py -- env steps/s of the batched RMSA step() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--chunk T] [--stats full|network|counters]

One "step" = every environment of the batch advances by one RMSAEnv.step (policy -> provision ->
release -> next arrival, all statistics) -- K steps are executed as ceil(K / chunk) launches of the
persistent step kernel (chunk env-steps per launch).  Workload at N=1: BASELINE.json configs[1]
(RMSA-v0, NSFNET, 320 slots, load 50, B = 4096 envs per GPU, shortest-available-path first-fit run on
the device, seeds 10 + i).  Inputs are synthetic (the reference's own Poisson traffic generator run on
the device) and all state is resident in HBM before the timed region starts.

For N > 1 the driver launches one process per GPU with torch.distributed.run; envs shard across ranks
with no data-path communication ("weak" scaling: B per GPU fixed); the only collective is the RCCL
all-reduce of the episode statistics vector at the end (orlg_reduce_counters).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ENV_KW = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000)
TOPOLOGY = "nsfnet_chen_5-paths_6-modulations"
# --mixed (BASELINE configs[4]): one topology group per rank, rank r takes MIXED[r % 3]; "JPN48" is not shipped with the
# reference, jpn12 stands in
MIXED = ("nsfnet_chen_5-paths_6-modulations", "jpn12_5-paths_6-modulations", "us14_3-paths_6-modulations")


def algorithmic_bytes_per_env_step(topo, W):
    """read the env's occupancy once + RMW on provision and release over the mean
    hop count + service record + request record + action/reward/done."""
    import numpy as np
    E = topo.num_links
    hbar = float(np.mean(topo.path_hops))
    return E * W * 8 + 2 * 2 * hbar * W * 8 + 48 + 40 + 16


def cpu_baseline(topo, seconds=10.0):
    """The CPU oracle (plain C restatement of the reference algorithm) on the host cores of the same box: one
    environment per thread (ctypes releases the GIL inside the C call), same workload, bounded sample.  Reported:
    the aggregate over all threads and the single-thread rate."""
    import threading
    from conftest import oracle_env_from_kwargs

    def worker(seed, secs, res, idx):
        o = oracle_env_from_kwargs(topo, dict(ENV_KW, seed=seed))
        o.run("sap_ff", 500, fields=[])  # warm-up to steady state
        n, done, t0 = 20000, 0, time.perf_counter()
        while time.perf_counter() - t0 < secs:
            o.run("sap_ff", n, fields=[])
            done += n
        res[idx] = (done, time.perf_counter() - t0)
        o.close()

    one = [None]
    worker(10, seconds / 2, one, 0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    res = [None] * cores
    th = [threading.Thread(target=worker, args=(10 + i, seconds / 2, res, i)) for i in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    return {"value": total / wall, "unit": "env steps/s", "cores": cores, "kind": "port",
            "value_1core": one[0][0] / one[0][1],
            "sample": f"{cores} threads x 1 env each, {total} steps in {wall:.1f} s (+ {one[0][0]} steps on one thread), SAP-FF, "
                      "same NSFNET-320 load-50 workload, oracle/orlg_oracle.c"}


def north_star_measurement(topo, args, stream, dev):
    """BASELINE.json north_star quotes its target (>= 10 M env-steps/s) at batch 65 536 on one MI355X: the same workload at
    that batch, timed the same way (1 launch of warm-up, 2 timed launches), reported next to the headline.  At this batch
    the library's AUTO rule runs the four-environments-per-wave step kernel; results are bit-identical."""
    import torch
    from optical_rl_gym_amd import BatchedRMSAEnv
    B2 = 65536
    env = BatchedRMSAEnv(topo, B2, **ENV_KW, seed=10, stats_level=args.stats, device=dev.index or 0)
    env.set_stream(stream.cuda_stream)
    env.run(args.policy, args.chunk, auto_reset=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(2):
        env.run(args.policy, args.chunk, auto_reset=True)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    red, _ = env.reduce_counters()
    env.close()
    return {"value": B2 * 2 * args.chunk / dt, "unit": "env steps/s", "batch": B2, "steps": 2 * args.chunk,
            "ms_per_launch": dt * 1e3 / 2, "step_kernel": "auto -> orlg_rmsa_group_kernel (4 envs per wave)" if args.policy in ("sap_ff", "sp_ff") else "auto",
            "service_blocking_rate": (red["services_processed"] - red["services_accepted"]) / red["services_processed"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--chunk", type=int, default=1000, help="env steps per kernel launch")
    ap.add_argument("--stats", default="full", choices=["full", "network", "counters"])
    ap.add_argument("--policy", default="sap_ff")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true", help="skip the extra B=65536 measurement (north_star batch)")
    ap.add_argument("--mixed", action="store_true", help="configs[4]: NSFNET / JPN12 / US14 topology groups, one per rank (r %% 3)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    from conftest import load_topology
    from optical_rl_gym_amd import BatchedRMSAEnv

    topo_name = MIXED[rank % 3] if args.mixed else TOPOLOGY
    topo = load_topology(topo_name)
    B = args.batch
    from optical_rl_gym_amd.distributed import allreduce_stats, shard_base_seed
    env = BatchedRMSAEnv(topo, B, **ENV_KW, seed=shard_base_seed(10, B, rank), stats_level=args.stats, device=local_rank)
    # a dedicated (non-default) stream: the step kernels AND the timing events live on it
    stream = torch.cuda.Stream(device=dev)
    env.set_stream(stream.cuda_stream)

    def run_steps(k, events=None):
        left = k
        while left > 0:
            n = min(left, args.chunk)
            if events is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
            env.run(args.policy, n, auto_reset=True)
            if events is not None:
                b.record(stream)
                events.append((a, b, n))
            left -= n

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    env.launch_info()  # kernel attributes / occupancy queried once, outside any timing
    run_steps(args.warmup)
    events = []
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, events)
    barrier()
    elapsed = time.perf_counter() - t0

    # statistics all-reduce (the path's only collective)
    red, vec = env.reduce_counters()
    stats = allreduce_stats(vec, dist, dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        W = env.words_per_link
        total_steps = B * world * args.steps
        kernel_ms = [a.elapsed_time(b) for a, b, _ in events]
        full = [(ms, n) for (ms, (_, _, n)) in zip(kernel_ms, events) if n == args.chunk] or list(zip(kernel_ms, [e[2] for e in events]))
        avg_ms = float(np.mean([ms for ms, _ in full]))
        n_per_launch = full[0][1]
        A = algorithmic_bytes_per_env_step(topo, W)
        achieved = A * B * n_per_launch / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch measured with rocprofv3 PMC passes on this workload (profiles/traffic.json), if recorded
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["entries"]
            ent = tj.get(f"rmsa_nsfnet320_B{B}_chunk{n_per_launch}_{args.stats}_{args.policy}")
            if ent:
                traffic, traffic_src = ent["traffic_bytes_per_launch"], ent["source"]
        except Exception:
            pass
        out = {
            "metric": "env steps/sec (whole node) + blocking-prob parity, NSFNET RMSA 320 slots",
            "value": total_steps / elapsed,
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 bitmap + f64 statistics",
            "data": "synthetic (reference's Poisson traffic generator run on the device, seeds 10+i)",
            "config": {"workload": (f"RMSA-v0 {'NSFNET/JPN12/US14 (rank mod 3)' if args.mixed else 'NSFNET'} 320 slots load 50, "
                                    f"batch {B} envs per GPU, {args.policy} on device, "
                                    f"stats={args.stats}, {args.chunk} env-steps per launch"),
                       "batch_per_gpu": B, "global_batch": B * world, "policy": args.policy, "stats_level": args.stats,
                       "chunk": args.chunk, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src,
                         "kernel": "orlg_rmsa_kernel_ff<5,%d>" % {"counters": 0, "network": 1, "full": 2}[args.stats],
                         "kernel_ms_per_launch": avg_ms, "algorithmic_bytes_per_env_step": A,
                         "env_steps_per_launch": B * n_per_launch},
            "blocking": {"services_processed": int(stats[0]), "services_accepted": int(stats[1]),
                         "service_blocking_rate": float((stats[0] - stats[1]) / max(1, stats[0])),
                         "bit_rate_blocking_rate": float((stats[4] - stats[5]) / max(1, stats[4])),
                         "episodes_done": int(stats[8]), "num_envs": int(stats[9])},
        }
        if world == 1 and not args.no_north_star and B != 65536:
            out["north_star_batch_65536"] = north_star_measurement(topo, args, stream, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(topo)
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
