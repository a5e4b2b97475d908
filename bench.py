#!/usr/bin/env python3
"""bench.py -- env steps/s of the batched RMSA step() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--chunk T] [--mixed] [--only NAME] [--dry-run]

One bench "step" = ONE LAUNCH of the persistent step kernel = every environment of the batch advances by `chunk`
(default 1000) RMSAEnv.step calls (policy -> provision -> release -> next arrival, all statistics).  K timed launches
follow W >= 1 untimed launches of the same shape (so every environment has made >= 300 steps -- SURVEY 8(d) config 2 --
and the network is in its steady state before the clock starts); ms_per_step x K = the timed region.

Headline workload at N=1: the batch BASELINE.json's north_star quotes its target on -- RMSA-v0, NSFNET, 320 slots,
load 50, B = 65 536 environments per GPU, shortest-available-path first fit run on the device, seeds 10 + i, full
statistics.  Sub-records (N=1 only), each timed the same way with its own roofline block and kernel name:
    rmsa_b4096        BASELINE configs[1] (the same env x 4096)
    phy_us14_b4096    BASELINE configs[2]: QoT-aware RMSA, US14, load 1400, bmfa -- with number_cuts_total / rss_total_metric
                      written every step as the reference's step() does (lead), without them, with the periodic
                      defragmentation (defrag_period 10, number_moves 10), and with the GN-model OSNR gate of the chosen
                      channels inside the step (configs[2] as BASELINE words it; parity unpinned by the reference)
    deeprmsa_b32768   BASELINE configs[3]: one launch + one observation build per step; also the PCIe-inclusive rate of
                      the agent loop (actions from host memory, observations copied back)
Inputs are synthetic (the reference's own Poisson traffic generator run on the device) and all state is resident in HBM
before the timed region starts.

N > 1: one process per GPU (the driver launches them with torch.distributed.run; `python bench.py --gpus N` without
WORLD_SIZE in the environment starts them itself, as fresh child processes, before anything touches a GPU).  Environments
shard across ranks with no data-path communication ("weak" scaling: B per GPU fixed, seeds base + rank*B + i); the only
collective is the RCCL all-reduce of the episode statistics vector.  --mixed: BASELINE configs[4], one topology group per
rank (NSFNET / JPN12 / US14 by rank mod 3).
"""
import argparse
import json
import os
import subprocess
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
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# VALU issue ceiling: 256 CUs x 4 SIMDs, one wave-instruction per 4 cycles per SIMD (measured: SQ_ACTIVE_INST_VALU = 4.2
# cycles per VALU instruction on these kernels), 2.4 GHz
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed launches (one launch = chunk env-steps per environment)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed launches of the same shape (at least 1)")
    ap.add_argument("--batch", type=int, default=65536, help="environments per GPU")
    ap.add_argument("--chunk", type=int, default=1000, help="env steps per kernel launch")
    ap.add_argument("--stats", default="full", choices=["full", "network", "counters"])
    ap.add_argument("--policy", default="sap_ff")
    ap.add_argument("--step-kernel", default="auto", choices=["auto", "wave", "group"])
    ap.add_argument("--queue-capacity", type=int, default=0, help="release-queue slots per environment (0 = the library's choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true", help="headline only")
    ap.add_argument("--no-pcie-loop", action="store_true", help="DeepRMSA record without the PCIe-inclusive agent loops (profiling: "
                    "their quarter-batch launches would be mistaken for the record's own)")
    ap.add_argument("--only", default=None, choices=["headline", "rmsa_b4096", "phy", "phy_metrics", "phy_defrag", "phy_gn", "deeprmsa"],
                    help="run one workload only (profiling)")
    ap.add_argument("--phy-chunk", type=int, default=1000, help="PhyRMSA sub-records: env-steps per launch per environment (a launch "
                    "ends with its slowest environment: 250-step launches run 3-6 %% below 1000-step ones)")
    ap.add_argument("--mixed", action="store_true", help="configs[4]: NSFNET / JPN12 / US14 topology groups, one per rank (r %% 3)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: ranks report their shard through gloo (launcher test)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes with torch.distributed.run as a CHILD process
    (this parent never touches a GPU) and pass rank 0's JSON line through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    return r.returncode


def dry_run(args, rank, world):
    """The launcher and the sharding without a GPU: every rank reports its shard, rank 0 prints one line."""
    from optical_rl_gym_amd.distributed import shard_base_seed
    import numpy as np
    B = args.batch
    mine = {"rank": rank, "world": world, "topology": MIXED[rank % 3] if args.mixed else TOPOLOGY,
            "first_seed": shard_base_seed(10, B, rank), "last_seed": shard_base_seed(10, B, rank) + B - 1}
    vec = np.zeros(16, np.int64)
    vec[9] = B
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        from optical_rl_gym_amd.distributed import allreduce_stats
        vec = allreduce_stats(vec, dist, None)
        shards = [None] * world
        dist.all_gather_object(shards, mine)
        dist.barrier()
        dist.destroy_process_group()
    else:
        shards = [mine]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "num_envs": int(vec[9]), "shards": shards}), flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ accounting
def algorithmic_bytes_per_env_step(topo, W, extra=0.0):
    """read the env's occupancy once + RMW on provision and release over the mean
    hop count + service record + request record + action/reward/done (+ `extra`: table rows / observation)."""
    import numpy as np
    E = topo.num_links
    hbar = float(np.mean(topo.path_hops))
    return E * W * 8 + 2 * 2 * hbar * W * 8 + 48 + 40 + 16 + extra


def load_pmc(key):
    """Per-env-step counters of this workload from the committed rocprofv3 PMC summary (profiles/pmc.json, written by
    tools/collect_profile.py from the passes of tools/profile_round.sh)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc.json")))["entries"].get(key)
    except Exception:
        return None


ROOFLINE_NOTE = ("every roofline block: bound/achieved/frac = SURVEY 8(d) algorithmic bytes per env-step x env-steps per launch / mean "
                 "launch time (HIP events on the kernel's stream), peak 8 TB/s; traffic = HBM bytes per launch from the FETCH_SIZE "
                 "(x2, gfx950) + WRITE_SIZE passes in profiles/pmc.json; the state lives in LDS for a whole launch, so what binds is "
                 "in valu_issue (wave-instructions/s against 256 CU x 4 SIMD x 2.4 GHz / 4 cycles = 614.4 G/s; valu_busy from the same "
                 "counters); sub-record blocks share bound / peak / unit with this one and come from profiles/r03_<workload>_summary.json")


def roofline_block(key, kernel, kernel_ms, A, env_steps_per_launch, batch):
    """HBM roofline by the SURVEY 8(d) accounting (what the contract asks for) + what the counters say binds the kernel."""
    achieved = A * env_steps_per_launch / (kernel_ms * 1e-3) / 1e9
    rate = env_steps_per_launch / (kernel_ms * 1e-3)
    rl = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
          "traffic": None, "kernel": kernel, "kernel_ms_per_launch": kernel_ms,
          "algorithmic_bytes_per_env_step": A, "env_steps_per_launch": env_steps_per_launch}
    pmc = load_pmc(key)
    if pmc and pmc.get("batch") == batch:
        per = pmc["per_env_step"]
        if "hbm_bytes" in per:
            rl["traffic"] = per["hbm_bytes"] * env_steps_per_launch
            rl["hbm_measured_GBps"] = per["hbm_bytes"] * rate / 1e9
        if "SQ_INSTS_VALU" in per:
            ginst = per["SQ_INSTS_VALU"] * rate / 64.0 / 1e9 if pmc.get("insts_are_per_lane") else per["SQ_INSTS_VALU"] * rate / 1e9
            rl["valu_issue"] = {"bound": "valu_issue", "valu_insts_per_env_step": per["SQ_INSTS_VALU"],
                                "salu_insts_per_env_step": per.get("SQ_INSTS_SALU"), "lds_insts_per_env_step": per.get("SQ_INSTS_LDS"),
                                "achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                                "frac": ginst / VALU_PEAK_GINST, "valu_busy": pmc.get("valu_busy"),
                                "waves_per_simd": pmc.get("waves_per_simd"), "source": pmc.get("source")}
    return rl


# ------------------------------------------------------------------------------------------------ measurements
class Clock:
    """barrier + synchronize on both sides of the timed region, HIP events per launch on the kernels' own stream."""

    def __init__(self, torch, dev, stream, dist):
        self.torch, self.dev, self.stream, self.dist = torch, dev, stream, dist

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def timed(self, launch, steps):
        ev = []
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            a.record(self.stream)
            launch()
            b.record(self.stream)
            ev.append((a, b))
        self.barrier()
        elapsed = time.perf_counter() - t0
        return elapsed, [a.elapsed_time(b) for a, b in ev]


def rmsa_record(clock, topo_name, B, chunk, warm, steps, args, rank, local_rank, key):
    import numpy as np
    from conftest import load_topology
    from optical_rl_gym_amd import BatchedRMSAEnv
    from optical_rl_gym_amd.distributed import shard_base_seed
    topo = load_topology(topo_name)
    env = BatchedRMSAEnv(topo, B, **ENV_KW, seed=shard_base_seed(10, B, rank), stats_level=args.stats, device=local_rank,
                         step_kernel=args.step_kernel, queue_capacity=args.queue_capacity)
    env.set_stream(clock.stream.cuda_stream)
    launch = lambda: env.run(args.policy, chunk, auto_reset=True)
    for _ in range(max(1, warm)):
        launch()
    elapsed, kms = clock.timed(launch, steps)
    red, vec = env.reduce_counters()
    kernel = env.last_kernel()
    W = env.words_per_link
    env.close()
    A = algorithmic_bytes_per_env_step(topo, W)
    kernel_ms = float(np.mean(kms))
    return {"value": B * chunk * steps / elapsed, "unit": "env steps/s", "batch": B, "env_steps_per_launch_per_env": chunk,
            "launches_timed": steps, "launches_warmup": max(1, warm), "timed_region_s": elapsed,
            "ms_per_launch": elapsed * 1e3 / steps, "step_kernel": kernel.split(" ")[0], "launch": kernel,
            "service_blocking_rate": (red["services_processed"] - red["services_accepted"]) / max(1, red["services_processed"]),
            "roofline": roofline_block(key, kernel.split(" ")[0], kernel_ms, A, B * chunk, B)}, vec, elapsed, topo


def phy_record(clock, args, variant):
    """BASELINE configs[2]: PhyRMSA US14 load 1400, B = 4096, bmfa (cut metric) on the device."""
    import numpy as np
    import torch
    from conftest import load_phy_tables, load_topology
    from optical_rl_gym_amd import BatchedPhyRMSAEnv
    B, chunk = 4096, args.phy_chunk
    topo = load_topology("us14_3-paths_6-modulations")
    pairs, mod, gsnr = load_phy_tables("us14_k3")
    defrag = variant == "phy_defrag"
    gn = variant == "phy_gn"
    metrics = variant in ("phy_metrics", "phy_defrag", "phy_gn")
    gate = None
    if gn:   # configs[2] as BASELINE words it: the GN-model OSNR gate of the chosen channels inside the step
        from optical_rl_gym_amd import gn_gate_parameters
        gate = gn_gate_parameters(topo)
    env = BatchedPhyRMSAEnv(topo, B, modulation_level=mod, connections_detail=pairs, gsnr=gsnr, load=1400,
                            mean_service_holding_time=25, episode_length=200, seed=10, grooming=False,
                            defrag_period=10 if defrag else None, number_moves=10 if defrag else None, metric="cut", gn_gate=gate)
    env.set_stream(clock.stream.cuda_stream)
    out = None
    if metrics:   # the reference's step() computes both every step (phy_rmsa_env.py:319-348): written to device buffers
        out = {"number_cuts_total": torch.empty((chunk, B), dtype=torch.float64, device=clock.dev),
               "rss_total_metric": torch.empty((chunk, B), dtype=torch.float64, device=clock.dev)}
    launch = lambda: env.run("bmfa", chunk, auto_reset=True, out=out)
    for _ in range(max(3, 3000 // chunk)):   # 3000 steps: load 1400 needs a few thousand arrivals to fill the network
        launch()
    # as many launches as make about one second
    clock.barrier()
    t0 = time.perf_counter()
    launch()
    clock.barrier()
    one = time.perf_counter() - t0
    steps = int(min(200, max(4, round(1.0 / max(one, 1e-4)))))
    elapsed, kms = clock.timed(launch, steps)
    red, _ = env.reduce_counters()
    kernel = env.last_kernel()
    gn_info = None
    if gn:
        # what the gate costs per env-step, from one more launch with outputs: a check = one chosen channel against the live
        # occupancy (2 asinh per other channel, evaluated for the W x 64 channel lanes, + 1; 2 exp per hop of the path)
        tr = env.run("bmfa", chunk, auto_reset=True, outputs=("act_path", "n_channels", "accepted", "gn_gsnr_db"))
        checked = ~np.isnan(tr["gn_gsnr_db"])
        nch = tr["n_channels"].astype(np.int64)
        checks = float(np.where(checked, np.where(tr["accepted"] != 0, nch, 1), 0).sum()) / checked.size
        hbar = float(np.mean(topo.path_hops))
        gn_info = {"steps_with_a_check": float(checked.mean()), "gate_rejections_per_env_step": float((checked & (tr["accepted"] == 0)).mean()),
                   "checks_per_env_step_at_least": checks, "asinh_per_check": 2 * 267 + 1, "exp_per_check_mean": 2 * hbar,
                   "asinh_per_env_step_at_least": checks * (2 * 267 + 1), "exp_per_env_step_at_least": checks * 2 * hbar,
                   "bound": "fp64 transcendental / VALU issue (see roofline.valu_issue)", "parity": "unpinned by the reference "
                   "(it gates by table only: phy_rmsa_env.py:596); pinned to the oracle's restatement of examples/calculate_osnr.py:9-56"}
    st = env.episode_stats()
    running = float(env.num_running().mean())
    env.close()
    A = algorithmic_bytes_per_env_step(topo, env.words_per_link, extra=268 * topo.k_paths)
    kernel_ms = float(np.mean(kms))
    rec = {"value": B * chunk * steps / elapsed, "unit": "env steps/s", "batch": B, "env_steps_per_launch_per_env": chunk,
           "launches_timed": steps, "timed_region_s": elapsed, "ms_per_launch": elapsed * 1e3 / steps,
           "policy": "bmfa (cut metric)", "per_step_metrics": metrics, "defragmentation": "period 10, 10 moves, cut" if defrag else None,
           "gn_gate": None,
           "step_kernel": kernel.split(" ")[0], "launch": kernel, "mean_running_services": running,
           "queue_overflow": int(st["queue_overflow"].max()),
           "service_blocking_rate": (red["services_processed"] - red["services_accepted"]) / max(1, red["services_processed"]),
           "roofline": roofline_block(variant, kernel.split(" ")[0], kernel_ms, A, B * chunk, B)}
    if gn_info:
        rec["gn_gate"] = gn_info
    return rec


def deeprmsa_record(clock, args):
    """BASELINE configs[3]: DeepRMSA-v0 NSFNET S=320 j=1, holding 7.5, inter-arrival 1/12, B = 32768: every step is one
    step launch (SAP-FF on the device standing in for the agent) + one observation build into a device buffer."""
    import numpy as np
    import torch
    from conftest import DEEPRMSA_NODE_PROBS, load_topology
    from optical_rl_gym_amd import BatchedDeepRMSAEnv
    B = 32768
    topo = load_topology(TOPOLOGY)
    env = BatchedDeepRMSAEnv(topo, B, num_spectrum_resources=320, j=1, mean_service_holding_time=7.5,
                             mean_service_inter_arrival_time=1 / 12.0, node_request_probabilities=DEEPRMSA_NODE_PROBS,
                             episode_length=50, seed=10)
    env.set_stream(clock.stream.cuda_stream)
    obs = torch.empty((B, env.obs_dim), dtype=torch.float64, device=clock.dev)

    def step():
        env.run("deeprmsa_sap_ff", 1, auto_reset=True)
        env.observation(out=obs)
    for _ in range(300):
        step()
    steps = 3000
    clock.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    clock.barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel durations on a sample (events between the two launches of a step)
    ks, ko = [], []
    evs = []
    for _ in range(100):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(clock.stream)
        env.run("deeprmsa_sap_ff", 1, auto_reset=True)
        b.record(clock.stream)
        env.observation(out=obs)
        c.record(clock.stream)
        evs.append((a, b, c))
    clock.barrier()
    for a, b, c in evs:
        ks.append(a.elapsed_time(b)); ko.append(b.elapsed_time(c))
    step_kernel = env.last_kernel()
    el_pcie = el_pipe = 1.0
    n_pcie = n_pipe = 1
    if not args.no_pcie_loop:
        # the agent loop as an SB3 agent pays it (PCIe-inclusive; never `value`).  (a) the plain form: actions from pageable host
        # memory, float64 observations copied back into pageable memory, one blocking round trip per step.
        acts = np.zeros(B, np.int32)
        obs_h = np.zeros((B, env.obs_dim), np.float64)
        for _ in range(20):
            env.run("deeprmsa_external", 1, actions=acts, auto_reset=True)
            env.observation(out=obs_h)
        n_pcie = 200
        clock.barrier()
        t0 = time.perf_counter()
        for _ in range(n_pcie):
            env.run("deeprmsa_external", 1, actions=acts, auto_reset=True)
            env.observation(out=obs_h)
        clock.barrier()
        el_pcie = time.perf_counter() - t0
        # (b) the same loop as an asynchronous vector environment runs it: float32 observations (what the agent's network takes),
        # pinned host buffers, the batch in four parts on four streams -- the agent works on one part's observations while the
        # others step, so the D2H of one part overlaps the H2D / step / observation build of the others (tools/exp_pcie_loop.py:
        # whole batch 122 M, halves 129 M, quarters 206 M env-steps/s; the host's issue cost, ~40 us per part, bounds it from there)
        PARTS = 4
        halves = []
        for h in range(PARTS):
            e2 = BatchedDeepRMSAEnv(topo, B // PARTS, num_spectrum_resources=320, j=1, mean_service_holding_time=7.5,
                                    mean_service_inter_arrival_time=1 / 12.0, node_request_probabilities=DEEPRMSA_NODE_PROBS,
                                    episode_length=50, seed=10 + h * (B // PARTS))
            st = torch.cuda.Stream(device=clock.dev)
            e2.set_stream(st.cuda_stream)
            halves.append(dict(env=e2, st=st, ev=torch.cuda.Event(),
                               acts_h=torch.zeros(B // PARTS, dtype=torch.int32).pin_memory(),
                               obs_h=torch.empty((B // PARTS, env.obs_dim), dtype=torch.float32).pin_memory()))

        def issue(hv):
            # pinned host buffers are used in place (include/orlg.h): the step kernel reads the actions, the observation kernel
            # writes its rows over the bus -- no staging copies, three host calls per part and step
            hv["env"].run("deeprmsa_external", 1, actions=hv["acts_h"], auto_reset=True)
            hv["env"].observation(out=hv["obs_h"])
            hv["ev"].record(hv["st"])
        for _ in range(300):
            for hv in halves:
                issue(hv)
        torch.cuda.synchronize(clock.dev)
        n_pipe = 1000
        clock.barrier()
        t0 = time.perf_counter()
        for hv in halves:
            issue(hv)
        for _ in range(n_pipe - 1):
            for hv in halves:
                hv["ev"].synchronize()     # this half's observations are in host memory: the agent would write its actions now
                issue(hv)
        torch.cuda.synchronize(clock.dev)
        el_pipe = time.perf_counter() - t0
        for hv in halves:
            hv["env"].close()
    red, _ = env.reduce_counters()
    W = env.words_per_link
    obs_dim = env.obs_dim
    env.close()
    A = algorithmic_bytes_per_env_step(topo, W, extra=8 * obs_dim)
    kernel_ms = float(np.mean(ks)) + float(np.mean(ko))
    rec = {"value": B * steps / elapsed, "unit": "env steps/s", "batch": B, "env_steps_per_launch_per_env": 1,
            "steps_timed": steps, "timed_region_s": elapsed, "ms_per_step": elapsed * 1e3 / steps,
            "step_kernel": step_kernel.split(" ")[0], "launch": step_kernel, "observation_kernel": "orlg_deeprmsa_obs_kernel<%d>" % W,
            "kernel_ms_step": float(np.mean(ks)), "kernel_ms_observation": float(np.mean(ko)), "obs_dim": obs_dim,
            "pcie_inclusive": {"value": B * n_pipe / el_pipe, "unit": "env steps/s", "ms_per_step": el_pipe * 1e3 / n_pipe,
                               "what": "agent loop, four quarter-batches pipelined on four streams, pinned host buffers read / written in "
                                       "place by the kernels: 4 B/env actions, %d B/env float32 observation" % (4 * obs_dim),
                               "serial_f64_pageable": {"value": B * n_pcie / el_pcie, "ms_per_step": el_pcie * 1e3 / n_pcie,
                                                       "what": "one blocking round trip per step, %d B/env float64 D2H" % (8 * obs_dim)}},
            "service_blocking_rate": (red["services_processed"] - red["services_accepted"]) / max(1, red["services_processed"]),
            "roofline": roofline_block("deeprmsa", step_kernel.split(" ")[0] + " + orlg_deeprmsa_obs_kernel<%d>" % W, kernel_ms, A, B, B)}
    if args.no_pcie_loop:
        rec.pop("pcie_inclusive")
    return rec


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
                      "same NSFNET-320 load-50 workload, oracle/orlg_oracle.c",
            "note": "C oracle only; the reference (Python + NumPy) runs 242 env-steps/s on one core (BASELINE.md 2) and cannot travel"}


# keys of a SUB-record that repeat what the line already says once (the unit, the 8 TB/s peak, the kernel's name inside "launch",
# the VALU peak of ROOFLINE_NOTE, the profile a counter block came from = profiles/r03_<workload>_summary.json)
SUB_DROP = {"unit", "step_kernel", "timed_region_s", "launches_timed", "launches_warmup", "steps_timed", "peak", "kernel",
            "env_steps_per_launch", "bound", "source", "queue_overflow", "observation_kernel", "obs_dim", "what"}


def compact(x, keep=("value", "ms_per_step", "timed_region_s")):
    """Floats to 6 significant digits (the headline's value / ms_per_step / timed_region_s stay as measured), no null entries
    below the top level and no repeated constants inside the sub-records: the whole line, every sub-record included, has to
    fit the 8 KB tail the driver keeps."""
    def walk(v, top, sub=False):
        if isinstance(v, dict):
            if top:
                return {k: (vv if k in keep else walk(vv, False, k == "sub_records")) for k, vv in v.items()}
            return {k: walk(vv, False, sub) for k, vv in v.items() if vv is not None and not (sub and k in SUB_DROP)}
        if isinstance(v, (list, tuple)):
            return [walk(e, False, sub) for e in v]
        if isinstance(v, float):
            return float("%.6g" % v)
        return v
    return walk(x, True)


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if args.dry_run:
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    if world_env is not None:   # launched as a rank (also WORLD_SIZE=1: the RCCL leg then runs with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        local_rank = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)
    # a dedicated (non-default) stream: the step kernels AND the timing events live on it
    stream = torch.cuda.Stream(device=dev)
    clock = Clock(torch, dev, stream, dist)
    from optical_rl_gym_amd.distributed import allreduce_stats

    sub = {}
    only = args.only
    out = None
    if only in (None, "headline", "rmsa_b4096"):
        B = 4096 if only == "rmsa_b4096" else args.batch
        topo_name = MIXED[rank % 3] if args.mixed else TOPOLOGY
        key = "rmsa_b%d" % B
        rec, vec, elapsed, topo = rmsa_record(clock, topo_name, B, args.chunk, args.warmup, args.steps, args, rank, local_rank, key)
        # statistics all-reduce (the path's only collective) and the slowest rank's clock
        stats = allreduce_stats(vec, dist, dev)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        collective = None
        if dist is not None:   # what the collective did, for the one-rank test of the RCCL leg (tests/test_gpu_distributed.py)
            collective = {"backend": dist.get_backend(), "world": world, "local_stats": [int(x) for x in vec],
                          "reduced_stats": [int(x) for x in stats], "clock_local_s": elapsed, "clock_max_s": float(tmax.item())}
        elapsed = float(tmax.item())
        total_steps = B * world * args.chunk * args.steps
        out = {
            "metric": "env steps/sec (whole node) + blocking-prob parity, NSFNET RMSA 320 slots",
            "value": total_steps / elapsed,
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": max(1, args.warmup),
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 bitmap + f64 statistics",
            "data": "synthetic (reference's Poisson traffic generator run on the device, seeds 10+i)",
            "config": {"workload": (f"RMSA-v0 {'NSFNET/JPN12/US14 (rank mod 3)' if args.mixed else 'NSFNET'} 320 slots load 50, "
                                    f"batch {B} envs per GPU, {args.policy} on device, stats={args.stats}; one bench step = one "
                                    f"launch of {args.chunk} env-steps per env, after {max(1, args.warmup)} warm-up launch(es) of the same shape"),
                       "batch_per_gpu": B, "global_batch": B * world, "policy": args.policy, "stats_level": args.stats,
                       "env_steps_per_launch_per_env": args.chunk, "parallelism": f"env-shard x{world}"},
            "timed_region_s": elapsed,
            "step_kernel": rec["step_kernel"], "launch": rec["launch"],
            "roofline": rec["roofline"], "roofline_note": ROOFLINE_NOTE,
            "blocking": {"services_processed": int(stats[0]), "services_accepted": int(stats[1]),
                         "service_blocking_rate": float((stats[0] - stats[1]) / max(1, stats[0])),
                         "bit_rate_blocking_rate": float((stats[4] - stats[5]) / max(1, stats[4])),
                         "episodes_done": int(stats[8]), "num_envs": int(stats[9])},
        }
        if collective is not None and args.no_sub_records:
            out["collective"] = collective
        if elapsed < 1.0:
            out["timed_region_short"] = "timed region below 1 s: raise --steps"
    if rank == 0 and world == 1 and not args.no_sub_records:
        jobs = {"rmsa_b4096": lambda: rmsa_record(clock, TOPOLOGY, 4096, args.chunk, 1, 100, args, 0, 0, "rmsa_b4096")[0],
                "phy_metrics": lambda: phy_record(clock, args, "phy_metrics"),
                "phy": lambda: phy_record(clock, args, "phy"),
                "phy_defrag": lambda: phy_record(clock, args, "phy_defrag"),
                "phy_gn": lambda: phy_record(clock, args, "phy_gn"),
                "deeprmsa": lambda: deeprmsa_record(clock, args)}
        for name, job in jobs.items():
            if only is None and name == "rmsa_b4096" and args.batch == 4096:
                continue
            if only is None or (only == name and name != "rmsa_b4096"):
                sub[name] = job()
    if rank == 0:
        if out is None:   # --only <sub-record>
            out = {"only": only}
        if sub:
            names = {"rmsa_b4096": "rmsa_b4096 (BASELINE configs[1])", "phy_metrics": "phy_us14_b4096 (BASELINE configs[2], with per-step metrics)",
                     "phy": "phy_us14_b4096_lazy_metrics", "phy_defrag": "phy_us14_b4096_defragmentation",
                     "phy_gn": "phy_us14_b4096_gn_gate (BASELINE configs[2] as worded)",
                     "deeprmsa": "deeprmsa_b32768 (BASELINE configs[3])"}
            out["sub_records"] = {names[k]: v for k, v in sub.items()}
        if world == 1 and not args.no_cpu_baseline and only is None:
            from conftest import load_topology
            out["cpu_baseline"] = cpu_baseline(load_topology(TOPOLOGY))
        print(json.dumps(compact(out)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
