#!/usr/bin/env python3
"""Where the step kernel's time goes: builds liborlg_sections.so (-DORLG_SECTIONS: s_memtime around every section of
RMSAEnv.step, summed over all waves), runs the bench workload and prints the share of wave-cycles per section.
usage (GPU box): python tools/section_profile.py [--stats full|network|counters] [--batch B]"""
import argparse, ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "optical-rl-gym-qot-aware_amd")
LIB = os.path.join(PKG, "liborlg_sections.so")
NAMES = ["idle/ticket", "state load", "policy", "validate+provision", "stats@provision", "queue insert", "outputs",
         "next arrival", "refill", "release scan", "release apply", "stats@release", "done/reset", "state store", "link replay", ""]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", default="full")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--chunk", type=int, default=0, help="env steps per launch (0 = all in one launch)")
    ap.add_argument("--phy", default=None, help="profile the PhyRMSA kernel with this policy (bmfa, sapff, ...) instead")
    ap.add_argument("--defrag", action="store_true")
    ap.add_argument("--metrics", action="store_true", help="PhyRMSA: per-step number_cuts_total / rss_total_metric outputs")
    ap.add_argument("--gn", action="store_true", help="PhyRMSA: the GN-model gate of the chosen channels inside the step")
    args = ap.parse_args()
    sys.path.insert(0, PKG)
    import build as orlg_build   # single-translation-unit build (csrc/orlg_unity.hip, W = 5: NSFNET-320 / US14-268)
    orlg_build.build_unity(LIB, ["-DORLG_SECTIONS", "-DORLG_PHY_FEW_POLICIES"], w=5, verbose=False)
    os.environ["ORLG_LIB_PATH"] = LIB
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    from conftest import load_topology
    from optical_rl_gym_amd import BatchedRMSAEnv, _lib
    if args.phy:
        if args.metrics:
            import torch
            torch.zeros(1, device="cuda")   # torch's HIP context first
        from conftest import load_phy_tables
        from optical_rl_gym_amd import BatchedPhyRMSAEnv
        pairs, mod, gsnr = load_phy_tables("us14_k3")
        env = BatchedPhyRMSAEnv(load_topology("us14_3-paths_6-modulations"), args.batch, modulation_level=mod, connections_detail=pairs,
                                gsnr=gsnr, load=1400, mean_service_holding_time=25, episode_length=200, seed=10,
                                defrag_period=10 if args.defrag else None, number_moves=10 if args.defrag else None,
                                **({"gn_gate": __import__("optical_rl_gym_amd").gn_gate_parameters(load_topology("us14_3-paths_6-modulations"))}
                                   if args.gn else {}))
        names = ["idle/ticket", "state load", "policy: virtual layer", "policy: row metrics", "policy: channel selection", "provision",
                 "outputs", "next arrival + RNG", "defrag: grooming walk", "release: buffer / rebuild", "release apply (+ next scan)",
                 "defrag: grooming scan", "defrag: candidate scan", "state store", "defrag: candidate rounds", "defrag: candidate ranks"]
        env.run(args.phy, 3000, auto_reset=True)
        L = _lib.load()
        out = (C.c_ulonglong * 16)()
        L.orlg_debug_sections(out, 1)
        if args.metrics:
            import torch
            ob = {"number_cuts_total": torch.empty((250, args.batch), dtype=torch.float64, device="cuda"),
                  "rss_total_metric": torch.empty((250, args.batch), dtype=torch.float64, device="cuda")}
            for _ in range(args.steps // 250):
                env.run(args.phy, 250, auto_reset=True, out=ob)
        else:
            env.run(args.phy, args.steps, auto_reset=True)
        env.synchronize()
        L.orlg_debug_sections(out, 1)
        tot = sum(out)
        res = {names[i]: round(100.0 * out[i] / tot, 2) for i in range(16) if names[i]}
        res["cycles_per_env_step"] = tot / (args.batch * args.steps)
        print(json.dumps({"kernel": "phy", "policy": args.phy, "defrag": args.defrag, "metrics": args.metrics, "percent_of_wave_cycles": res}))
        return
    env = BatchedRMSAEnv(load_topology("nsfnet_chen_5-paths_6-modulations"), args.batch, num_spectrum_resources=320, load=50,
                         mean_service_holding_time=25, episode_length=1000, seed=10, stats_level=args.stats)
    env.run("sap_ff", 500, auto_reset=True)
    L = _lib.load()
    out = (C.c_ulonglong * 16)()
    L.orlg_debug_sections(out, 1)
    chunk = args.chunk or args.steps
    for _ in range(args.steps // chunk):
        env.run("sap_ff", chunk, auto_reset=True)
    env.synchronize()
    L.orlg_debug_sections(out, 1)
    tot = sum(out)
    res = {NAMES[i]: round(100.0 * out[i] / tot, 2) for i in range(15)}
    res["cycles_per_env_step"] = tot / (args.batch * args.steps)
    print(json.dumps({"stats": args.stats, "batch": args.batch, "percent_of_wave_cycles": res}))


if __name__ == "__main__":
    main()
