#!/usr/bin/env python3
"""BASELINE configs[4] shapes on one GPU: each topology group of `bench.py --mixed` (NSFNET / JPN12 / US14) at the per-GPU batch of
the scaling run (65 536 environments), one after the other -- env-steps/s, kernel picked, blocking rate.  (The 8-GPU run itself is
the driver's; this checks that every rank's shape fits and what it delivers.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from conftest import load_topology
from optical_rl_gym_amd import BatchedRMSAEnv
sys.path.insert(0, ROOT)
import bench
out = {}
for name in bench.MIXED:
    topo = load_topology(name)
    env = BatchedRMSAEnv(topo, 65536, **bench.ENV_KW, seed=10)
    for _ in range(2):
        env.run("sap_ff", 1000, auto_reset=True)
    env.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        env.run("sap_ff", 1000, auto_reset=True)
    env.synchronize(); dt = time.perf_counter() - t0
    red, _ = env.reduce_counters()
    out[name] = {"env_steps_per_s": 65536 * 5000 / dt, "launch": env.last_kernel(),
                 "service_blocking_rate": (red["services_processed"] - red["services_accepted"]) / red["services_processed"]}
    env.close()
print(json.dumps(out))
