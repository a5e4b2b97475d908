#!/usr/bin/env python3
"""Two handles stepping at the same time on their own streams, long launches with tickets in chunks of steps (GPU box): the waves
of one kernel wait for chunks only other RUNNING waves hold (every ticket is drawn, none is assigned to a workgroup that may not
be resident), so two kernels sharing the device must both drain.  Checks the saved states against the same launches run one
handle at a time.  Run it under `timeout`: a hang here would be a kernel that does not drain.
usage: timeout -k 10 300 python tools/exp_concurrent_handles.py [--batch 40000] [--steps 500] [--launches 3]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=40000)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--launches", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    from conftest import load_topology
    from optical_rl_gym_amd import BatchedRMSAEnv
    topo = load_topology("nsfnet_chen_5-paths_6-modulations")
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=400)

    def make(seed):
        return BatchedRMSAEnv(topo, args.batch, seed=seed, step_kernel="group", **kw)

    ref = []
    for seed in (11, 500011):
        e = make(seed)
        for _ in range(args.launches):
            e.run("sap_ff", args.steps, auto_reset=True)
        ref.append(e.save_state().copy())
        name = e.last_kernel()
        e.close()
    a, b = make(11), make(500011)
    t0 = time.perf_counter()
    for _ in range(args.launches):
        a.run("sap_ff", args.steps, auto_reset=True)   # (no outputs: the call returns when the launch is queued)
        b.run("sap_ff", args.steps, auto_reset=True)
    a.synchronize(); b.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(a.save_state(), ref[0]) and np.array_equal(b.save_state(), ref[1]))
    print(json.dumps({"kernel": name, "batch_per_handle": args.batch, "steps": args.steps, "launches": args.launches,
                      "states_equal_to_serial_runs": ok, "seconds_concurrent": round(dt, 4),
                      "M_env_steps_per_s_both": round(2 * args.batch * args.steps * args.launches / dt / 1e6, 1)}))
    a.close(); b.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
