#!/usr/bin/env python3
"""Where the PCIe-inclusive agent loop of bench.py's DeepRMSA record spends its time (GPU box): host cost of issuing one
half-step, GPU-side period when the host never waits, and the loop variants (one stream / two streams, halves / whole batch,
float32 / float64, pinned / pageable).  Prints one JSON line.
usage: python tools/exp_pcie_loop.py [--batch 32768] [--steps 500]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=500)
    args = ap.parse_args()
    import numpy as np
    import torch
    torch.zeros(1, device="cuda")
    from conftest import DEEPRMSA_NODE_PROBS, load_topology
    from optical_rl_gym_amd import BatchedDeepRMSAEnv
    dev = torch.device("cuda", 0)
    topo = load_topology("nsfnet_chen_5-paths_6-modulations")
    B, n = args.batch, args.steps

    def make(parts, dtype):
        hv = []
        for h in range(parts):
            b = B // parts
            e = BatchedDeepRMSAEnv(topo, b, num_spectrum_resources=320, j=1, mean_service_holding_time=7.5,
                                   mean_service_inter_arrival_time=1 / 12.0, node_request_probabilities=DEEPRMSA_NODE_PROBS,
                                   episode_length=50, seed=10 + h * b)
            st = torch.cuda.Stream(device=dev)
            e.set_stream(st.cuda_stream)
            hv.append(dict(env=e, st=st, ev=torch.cuda.Event(), acts_h=torch.zeros(b, dtype=torch.int32).pin_memory(),
                           acts_d=torch.zeros(b, dtype=torch.int32, device=dev),
                           obs_d=torch.empty((b, e.obs_dim), dtype=dtype, device=dev),
                           obs_h=torch.empty((b, e.obs_dim), dtype=dtype).pin_memory()))
        return hv

    def issue_direct(hv):
        hv["env"].run("deeprmsa_external", 1, actions=hv["acts_h"], auto_reset=True)
        hv["env"].observation(out=hv["obs_h"])
        hv["ev"].record(hv["st"])

    def issue(hv, copy=True):
        with torch.cuda.stream(hv["st"]):
            if copy:
                hv["acts_d"].copy_(hv["acts_h"], non_blocking=True)
            hv["env"].run("deeprmsa_external", 1, actions=hv["acts_d"], auto_reset=True)
            hv["env"].observation(out=hv["obs_d"])
            if copy:
                hv["obs_h"].copy_(hv["obs_d"], non_blocking=True)
            hv["ev"].record(hv["st"])

    out = {}
    for parts, dtype, name in ((2, torch.float32, "halves_f32"), (1, torch.float32, "whole_f32"), (4, torch.float32, "quarters_f32"),
                               (2, torch.float64, "halves_f64")):
        hvs = make(parts, dtype)
        for _ in range(100):
            for hv in hvs:
                issue(hv)
        torch.cuda.synchronize()
        # host cost of issuing (no waits), then the drain
        t0 = time.perf_counter()
        for _ in range(n):
            for hv in hvs:
                issue(hv)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        # no copies: kernels only
        t0 = time.perf_counter()
        for _ in range(n):
            for hv in hvs:
                issue(hv, copy=False)
        torch.cuda.synchronize()
        t_nocopy = time.perf_counter() - t0
        # the ping-pong loop of bench.py
        t0 = time.perf_counter()
        for hv in hvs:
            issue(hv)
        for _ in range(n - 1):
            for hv in hvs:
                hv["ev"].synchronize()
                issue(hv)
        torch.cuda.synchronize()
        t_pp = time.perf_counter() - t0
        # pinned buffers used in place by the kernels (no staging copies)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for hv in hvs:
            issue_direct(hv)
        for _ in range(n - 1):
            for hv in hvs:
                hv["ev"].synchronize()
                issue_direct(hv)
        torch.cuda.synchronize()
        t_dir = time.perf_counter() - t0
        out[name] = {"direct_us_per_step": t_dir / n * 1e6, "direct_M_env_steps_per_s": B * n / t_dir / 1e6,
                     "host_issue_us_per_step": t_issue / n * 1e6, "free_running_us_per_step": t_all / n * 1e6,
                     "kernels_only_us_per_step": t_nocopy / n * 1e6, "ping_pong_us_per_step": t_pp / n * 1e6,
                     "ping_pong_M_env_steps_per_s": B * n / t_pp / 1e6}
        # D2H alone
        hv = hvs[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            with torch.cuda.stream(hv["st"]):
                hv["obs_h"].copy_(hv["obs_d"], non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        out[name]["d2h_us_per_part"] = dt * 1e6
        out[name]["d2h_GBps"] = hv["obs_h"].numel() * hv["obs_h"].element_size() / dt / 1e9
        for hv in hvs:
            hv["env"].close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
