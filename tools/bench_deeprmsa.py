#!/usr/bin/env python3
"""Secondary benchmark: BASELINE.json configs[3] -- DeepRMSA-v0 on NSFNET, 320 slots, j = 1, batch 32768: every step is
the SAP-FF policy + DeepRMSAEnv.step on the device followed by the observation build (deeprmsa_env.py:60-121) for all
environments into a device buffer.  Prints one JSON line.  usage: python tools/bench_deeprmsa.py [--batch B] [--steps K]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=300)
    args = ap.parse_args()
    import torch
    from conftest import load_topology
    from optical_rl_gym_amd import BatchedDeepRMSAEnv
    topo = load_topology("nsfnet_chen_5-paths_6-modulations")
    env = BatchedDeepRMSAEnv(topo, args.batch, num_spectrum_resources=320, j=1, mean_service_holding_time=7.5,
                             mean_service_inter_arrival_time=1 / 12.0, episode_length=50, seed=10)
    obs = torch.empty((args.batch, env.obs_dim), dtype=torch.float64, device="cuda")

    def run(k):
        for _ in range(k):
            env.run("deeprmsa_sap_ff", 1, auto_reset=True)
            env.observation(out=obs)
    run(args.warmup)
    env.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    env.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    red, _ = env.reduce_counters()
    obs_bytes = args.batch * env.obs_dim * 8
    print(json.dumps({"metric": "env steps/s, DeepRMSA-v0 NSFNET S=320 j=1, step + observation per step", "value": args.batch * args.steps / dt,
                      "batch": args.batch, "steps": args.steps, "ms_per_step": dt * 1e3 / args.steps, "obs_dim": env.obs_dim,
                      "obs_GBps_written": obs_bytes * args.steps / dt / 1e9,
                      "service_blocking_rate": (red["services_processed"] - red["services_accepted"]) / red["services_processed"]}))
    env.close()


if __name__ == "__main__":
    main()
