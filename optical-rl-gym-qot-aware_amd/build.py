"""Build liborlg.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python optical-rl-gym-qot-aware_amd/build.py [--force] [-j N]

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the fp64 statistics and the
arrival process must perform exactly the reference's IEEE operations (no fused multiply-add).

The library is a set of translation units compiled in parallel into csrc/build/*.o and linked: the host API
(orlg_api.hip, orlg_phy_api.hip, orlg_osnr.hip) and one object per (kernel family, words per link W) from
orlg_inst_{wave,group,phy}.hip.  An object is rebuilt when any file its depfile names (or the flags) changed,
so an edit of one kernel family recompiles that family only.
"""
import concurrent.futures
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "liborlg.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]
WAVE_W = (1, 2, 3, 4, 5, 6, 8)
PHY_W = (1, 2, 3, 4, 5)


def units():
    """(object name, source, extra defines)"""
    u = [("orlg_api", "orlg_api.hip", []), ("orlg_phy_api", "orlg_phy_api.hip", []), ("orlg_osnr", "orlg_osnr.hip", [])]
    # the largest objects first: the pool then finishes with the small ones
    for w in sorted(WAVE_W, reverse=True):
        u.append((f"orlg_inst_wave_w{w}", "orlg_inst_wave.hip", [f"-DORLG_INST_W={w}"]))
    for w in sorted(PHY_W, reverse=True):
        u.append((f"orlg_inst_phy_w{w}", "orlg_inst_phy.hip", [f"-DORLG_INST_W={w}"]))
    for w in sorted(WAVE_W, reverse=True):
        u.append((f"orlg_inst_group_w{w}", "orlg_inst_group.hip", [f"-DORLG_INST_W={w}"]))
    return u


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _flags_tag(extra):
    return hashlib.sha1(" ".join([_hipcc()] + FLAGS + extra).encode()).hexdigest()


def _deps(dfile):
    try:
        txt = open(dfile).read()
    except OSError:
        return None
    txt = txt.replace("\\\n", " ")
    return [t for t in txt.split(":", 1)[1].split() if t] if ":" in txt else None


def _stale(name, extra):
    obj, dfile, tag = (os.path.join(OBJ, name + ext) for ext in (".o", ".d", ".flags"))
    if not os.path.exists(obj):
        return True
    try:
        if open(tag).read() != _flags_tag(extra):
            return True
    except OSError:
        return True
    deps = _deps(dfile)
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    for d in deps:
        try:
            if os.path.getmtime(d) > t:
                return True
        except OSError:
            return True
    return False


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    # every source the library is made of (cheap check; the per-object depfiles decide what is actually recompiled)
    srcs = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(HERE, "..", "include", "orlg.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def _compile(name, src, extra, verbose):
    obj, dfile, tag = (os.path.join(OBJ, name + ext) for ext in (".o", ".d", ".flags"))
    cmd = [_hipcc()] + FLAGS + extra + ["-I", CSRC, "-MD", "-MF", dfile, "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    open(tag, "w").write(_flags_tag(extra))
    return obj


def build(force=False, verbose=True, jobs=None):
    """Compile what is stale and link liborlg.so.  One builder at a time: the ranks of a multi-GPU launch all import the package
    at once, and a stale tree must not be rebuilt by eight of them into the same object directory (the others wait for the
    lock, then find the library fresh); the library appears under its name only when it is complete."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, jobs):
    todo = [(n, s, x) for n, s, x in units() if force or _stale(n, x)]
    if not todo and os.path.exists(LIB) and not force and not needs_build():
        return LIB
    jobs = jobs or min(len(todo) or 1, os.cpu_count() or 1, 16)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        futs = [ex.submit(_compile, n, s, x, verbose) for n, s, x in todo]
        for f in futs:
            f.result()
    objs = [os.path.join(OBJ, n + ".o") for n, _, _ in units()]
    tmp = LIB + ".tmp%d" % os.getpid()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd).replace(tmp, LIB), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    return LIB


def build_unity(out, defines=(), w=5, verbose=True):
    """The instrumented single-translation-unit builds of tools/ (one word count)."""
    cmd = [_hipcc()] + FLAGS + ["-shared", f"-DORLG_INST_W={w}"] + list(defines) + ["-I", CSRC, os.path.join(CSRC, "orlg_unity.hip"), "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    j = None
    if "-j" in sys.argv:
        j = int(sys.argv[sys.argv.index("-j") + 1])
    build(force="--force" in sys.argv, jobs=j)
