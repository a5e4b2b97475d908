#!/usr/bin/env python3
"""What the arrivals' refills cost a one-step launch (GPU box).  A launch of one step ends with its slowest wave, and with the
environments' refills staggered (62 - env % 56) ~6.5 % of the quads refill in EVERY launch.  This builds the library with
-DORLG_EXP_NO_STAGGER (every environment refills in the same launch, the other 61 of 62 none), steps the DeepRMSA shape of
bench.py one launch per step and prints the step kernel's time per launch: the launches without a refill are what a launch
would cost if refills were taken out of the step launches (topped up by a kernel of their own).  Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "optical-rl-gym-qot-aware_amd")
for p in (ROOT, os.path.join(ROOT, "tests"), PKG):
    sys.path.insert(0, p)


def main():
    import build as orlg_build
    lib = os.path.join(PKG, "liborlg_nostagger.so")
    orlg_build.build_unity(lib, ["-DORLG_EXP_NO_STAGGER", "-DORLG_PHY_FEW_POLICIES"], w=5, verbose=False)
    os.environ["ORLG_LIB_PATH"] = lib
    import numpy as np
    import torch
    torch.zeros(1, device="cuda")
    from conftest import DEEPRMSA_NODE_PROBS, load_topology
    from optical_rl_gym_amd import BatchedDeepRMSAEnv
    B = 32768
    env = BatchedDeepRMSAEnv(load_topology("nsfnet_chen_5-paths_6-modulations"), B, num_spectrum_resources=320, j=1,
                             mean_service_holding_time=7.5, mean_service_inter_arrival_time=1 / 12.0,
                             node_request_probabilities=DEEPRMSA_NODE_PROBS, episode_length=50, seed=10)
    st = torch.cuda.Stream()
    env.set_stream(st.cuda_stream)
    for _ in range(300):
        env.run("deeprmsa_sap_ff", 1, auto_reset=True)
    n = 186
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    with torch.cuda.stream(st):
        ev[0].record(st)
        for i in range(n):
            env.run("deeprmsa_sap_ff", 1, auto_reset=True)
            ev[i + 1].record(st)
    torch.cuda.synchronize()
    us = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)])
    srt = np.sort(us)
    print(json.dumps({"launch": env.last_kernel(), "launches": n, "median_us": float(np.median(us)), "p10_us": float(srt[n // 10]),
                      "p90_us": float(srt[9 * n // 10]), "max_us": float(srt[-1]), "slowest_3_us": srt[-3:].round(1).tolist(),
                      "mean_us": float(us.mean())}))


if __name__ == "__main__":
    main()
