"""Build liborlg.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python optical-rl-gym-qot-aware_amd/build.py [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the fp64 statistics and the
arrival process must perform exactly the reference's IEEE operations (no fused multiply-add).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liborlg.so")
SOURCES = ["orlg_api.hip"]
DEPS = ["orlg_api.hip", "orlg_kernels.hip", "orlg_phy_api.hip", "orlg_phy_kernels.hip", "orlg_device.h", "orlg_math.h", os.path.join("..", "..", "include", "orlg.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-I", CSRC] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
