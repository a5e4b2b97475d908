#!/usr/bin/env python3
"""Randomised cross-check of the two RMSA step kernels (wave per environment / four environments per wave): random grid
topologies, slot counts, k, loads, policies, batch sizes and launch lengths; both kernels are driven through the same launches
and their complete states (save_state) must be equal byte for byte after every launch.  Prints one JSON line.
usage (GPU box): python tools/cross_check_kernels.py [--configs N] [--seed S]"""
import argparse, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def grid_edges(rows, cols, rng):
    node = lambda r, c: r * cols + c + 1
    edges = []
    for r in range(rows):
        for c in range(cols):
            if c + 1 < cols:
                edges.append((node(r, c), node(r, c + 1), int(rng.integers(60, 400))))
            if r + 1 < rows:
                edges.append((node(r, c), node(r + 1, c), int(rng.integers(60, 400))))
            if r + 1 < rows and c + 1 < cols and (r + c) % 2 == 0:
                edges.append((node(r, c), node(r + 1, c + 1), int(rng.integers(80, 500))))
    return edges


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=24)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import numpy as np
    from optical_rl_gym_amd import BatchedRMSAEnv, OrlgError
    from optical_rl_gym_amd.topology_io import topology_from_txt
    rng = np.random.default_rng(args.seed)
    done, skipped, launches = 0, 0, 0
    tmp = tempfile.mkdtemp()
    for c in range(args.configs):
        rows, cols = int(rng.integers(2, 5)), int(rng.integers(2, 5))
        edges = grid_edges(rows, cols, rng)
        path = os.path.join(tmp, f"g{c}.txt")
        with open(path, "w") as f:
            f.write(f"{rows * cols}\n{len(edges)}\n" + "".join(f"{a} {b} {l}\n" for a, b, l in edges))
        k = int(rng.integers(1, 7))
        topo = None
        while topo is None:
            try:
                topo = topology_from_txt(path, f"g{c}", k_paths=k)
            except ValueError:   # a small grid does not have k simple paths for every pair
                k -= 1
        S = int(rng.choice([64, 80, 100, 128, 200, 256, 320, 400, 512]))
        load = float(rng.uniform(0.1, 0.6)) * S * len(edges) / 40.0
        kw = dict(num_spectrum_resources=S, load=max(2.0, load), mean_service_holding_time=float(rng.uniform(5, 30)),
                  episode_length=int(rng.integers(20, 200)), seed=int(rng.integers(1, 10000)),
                  stats_level=str(rng.choice(["full", "network", "counters"])))
        B = int(rng.choice([1, 3, 4, 5, 17, 64]))
        policy = str(rng.choice(["sap_ff", "sp_ff", "external", "path_ff_external"]))
        try:
            a = BatchedRMSAEnv(topo, B, step_kernel="wave", **kw)
            b = BatchedRMSAEnv(topo, B, step_kernel="group", **kw)
        except OrlgError as e:
            skipped += 1
            continue
        for n in [int(x) for x in rng.choice([1, 2, 7, 40, 150, 1000], size=5)]:
            for t in range(1 if policy in ("sap_ff", "sp_ff") else n):
                if policy == "external":
                    act = np.stack([rng.integers(0, k + 1, B), rng.integers(0, S + 1, B)], axis=-1).astype(np.int32)
                    a.run(policy, 1, actions=act, auto_reset=True); b.run(policy, 1, actions=act, auto_reset=True)
                elif policy == "path_ff_external":
                    act = rng.integers(0, k + 1, B).astype(np.int32)
                    a.run(policy, 1, actions=act, auto_reset=True); b.run(policy, 1, actions=act, auto_reset=True)
                else:
                    a.run(policy, n, auto_reset=True); b.run(policy, n, auto_reset=True)
                launches += 1
            sa, sb = a.save_state(), b.save_state()
            if not np.array_equal(sa, sb):
                print(json.dumps({"mismatch": {"config": c, "kw": kw, "k": k, "B": B, "policy": policy, "grid": [rows, cols],
                                               "first_byte": int(np.nonzero(sa != sb)[0][0])}}))
                sys.exit(1)
        try:
            a.reduce_counters(); b.reduce_counters()
        except OrlgError:
            pass   # a full queue at this random load is reported by both
        a.close(); b.close()
        done += 1
    print(json.dumps({"configs_checked": done, "skipped_shape_too_large_for_group_kernel": skipped, "launches_per_kernel": launches,
                      "result": "states equal byte for byte after every launch"}))


if __name__ == "__main__":
    main()
