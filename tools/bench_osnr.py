#!/usr/bin/env python3
"""Secondary benchmark: the GN-model admission check (calculate_osnr.py:9-56) at BASELINE configs[2] scale: M checks, each a
US14-like path of 5 links x 12 spans x (267 interferers + the service itself), all inputs resident in HBM.  Prints one
JSON line: checks/s and the (span x interferer) terms of the reference's double loop those checks cover per second (the
kernel factors the sum per link, so it does not evaluate them one by one).
usage: python tools/bench_osnr.py [--checks M]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checks", type=int, default=4096)
    ap.add_argument("--links", type=int, default=5)
    ap.add_argument("--spans", type=int, default=12)
    ap.add_argument("--services", type=int, default=268)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch
    from optical_rl_gym_amd import _lib
    from optical_rl_gym_amd.osnr import FIELDS, OsnrBatch
    M, Lk, Sp, V = args.checks, args.links, args.spans, args.services
    rng = np.random.default_rng(0)
    nl = M * Lk
    b = {"check_link_off": np.arange(M + 1, dtype=np.int32) * Lk,
         "link_span_off": np.arange(nl + 1, dtype=np.int32) * Sp,
         "link_svc_off": np.arange(nl + 1, dtype=np.int32) * V,
         "bandwidth": np.full(M, 50e9), "center_frequency": 193.1e12 + rng.integers(-130, 130, M) * 50e9,
         "launch_power": np.full(M, 1e-3),
         "span_length_km": rng.uniform(40, 80, nl * Sp), "span_attenuation": np.full(nl * Sp, 0.2 / (2 * 10 * np.log10(np.e) * 1e3)),
         "span_noise_figure": np.full(nl * Sp, 10 ** 0.55),
         "svc_bandwidth": np.full(nl * V, 50e9), "svc_se": rng.integers(1, 7, nl * V).astype(np.int32)}
    grid = 193.1e12 + (np.arange(V) - V // 2) * 50e9
    b["svc_center_frequency"] = np.tile(grid, nl)
    self_flag = np.zeros(nl * V, np.uint8)
    # the checked service sits on its own centre frequency in every link's list
    for m in range(M):
        idx = int(round((b["center_frequency"][m] - 193.1e12) / 50e9)) + V // 2
        for l in range(Lk):
            self_flag[(m * Lk + l) * V + idx] = 1
    b["svc_is_self"] = self_flag
    dev = {n: torch.from_numpy(np.ascontiguousarray(b[n], dtype=dt)).cuda() for n, dt in FIELDS}
    out = torch.empty(M, dtype=torch.float64, device="cuda")
    L = _lib.load()
    L.orlg_gn_osnr.argtypes = [C.POINTER(OsnrBatch), C.c_void_p, C.c_int32, C.c_void_p]
    q = OsnrBatch()
    for n, _ in FIELDS:
        setattr(q, n, C.c_void_p(dev[n].data_ptr()))
    q.num_checks, q.num_links, q.num_spans, q.num_services = M, nl, nl * Sp, nl * V
    _lib.check(L.orlg_gn_osnr(C.byref(q), C.c_void_p(out.data_ptr()), 0, None))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        _lib.check(L.orlg_gn_osnr(C.byref(q), C.c_void_p(out.data_ptr()), 0, None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    evals = M * Lk * Sp * (V - 1)
    print(json.dumps({"metric": "GN-model admission checks/s", "value": M / dt, "checks": M, "ms_per_batch": dt * 1e3,
                      "interferer_span_terms_covered_per_s": evals / dt,
                      "gsnr_db_min_max": [float(out.min()), float(out.max())]}))


if __name__ == "__main__":
    main()
