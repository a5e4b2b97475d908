#!/usr/bin/env python3
"""Experiment: what would the step kernel gain if the shape / LDS-layout fields of OrlgParams were compile-time constants?
Dumps the fields of the bench workload (orlg_debug_layout), builds liborlg_shape.so with them as __builtin_assume
equalities (-DORLG_SHAPE_ASSUME=...), and runs bench.py against both libraries.
usage (GPU box): python tools/shape_experiment.py"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "optical-rl-gym-qot-aware_amd")
FIELDS = ("N E S K NBR Q NW lint_stride tab_bytes t_pair t_recs t_nslots t_bitrates t_brcum t_srccum t_dstcum t_divs t_inv l_occ "
          "l_qtime l_qdesc l_mt l_lstat l_hist l_lint l_scratch l_wsc l_ring l_wave_bytes l_shared_bytes l_outs").split()


def main():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    from conftest import load_topology
    from optical_rl_gym_amd import BatchedRMSAEnv, _lib
    env = BatchedRMSAEnv(load_topology("nsfnet_chen_5-paths_6-modulations"), 64, num_spectrum_resources=320, load=50,
                         mean_service_holding_time=25, episode_length=1000, seed=10, stats_level="full")
    L = _lib.load()
    out = (C.c_int32 * 64)()
    L.orlg_debug_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    n = L.orlg_debug_layout(env.h, out, 64)
    assert n == len(FIELDS), (n, len(FIELDS))
    vals = dict(zip(FIELDS, out[:n]))
    print(json.dumps(vals))
    skip = set(sys.argv[1:])
    assume = " ".join("__builtin_assume(p.%s==%d);" % (f, v) for f, v in vals.items() if f not in skip)
    lib = os.path.join(PKG, "liborlg_shape.so")
    sys.path.insert(0, PKG)
    import build as orlg_build
    orlg_build.build_unity(lib, ["-DORLG_SHAPE_ASSUME=" + assume], w=5, verbose=False)
    for tag, envv in (("generic", {}), ("shape", {"ORLG_LIB_PATH": lib})):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-north-star"],
                           env={**os.environ, **envv}, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
        try:
            print(tag, json.loads(line)["value"])
        except Exception:
            print(tag, line)


if __name__ == "__main__":
    main()
