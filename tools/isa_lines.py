#!/usr/bin/env python3
"""Static instruction counts of one kernel by source line (from `hipcc -S --cuda-device-only -gline-tables-only`):
    python tools/isa_lines.py <file.s> <mangled kernel name substring> [--top N] [--ranges a-b,c-d,...]
Prints VALU / SALU / LDS / VMEM counts per source file:line (the innermost .loc in force), and totals per given line ranges
of the kernel's main source file.  Static counts are not dynamic counts, but they show which source blocks are heavy."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_", "V_")):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    ranges = []
    if "--ranges" in sys.argv:
        for r in sys.argv[sys.argv.index("--ranges") + 1].split(","):
            f, rr = r.split(":") if ":" in r else ("", r)
            a, b = rr.split("-")
            ranges.append((f, int(a), int(b)))
    files = {}
    inside = False
    cur = ("?", 0)
    per = collections.defaultdict(lambda: collections.Counter())
    tot = collections.Counter()
    for ln in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        if re.match(r"^_Z\w*:", ln) or re.match(r"^\w+:\s*; @", ln):
            inside = pat in ln
            continue
        if not inside:
            continue
        if ln.startswith(".Lfunc_end"):
            inside = False
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"\s+([a-z_0-9]+)\s", ln)
        if m and not ln.strip().startswith((".", ";")):
            c = classify(m.group(1))
            per[cur][c] += 1
            tot[c] += 1
    print("total", dict(tot))
    rows = sorted(per.items(), key=lambda kv: -kv[1]["valu"])[:top]
    for (f, l), c in rows:
        print(f"{f}:{l}\tvalu {c['valu']}\tsalu {c['salu']}\tlds {c['lds']}\tvmem {c['vmem']}")
    for f, a, b in ranges:
        c = collections.Counter()
        for (ff, l), cc in per.items():
            if (not f or ff == f) and a <= l <= b:
                c.update(cc)
        print(f"range {f}:{a}-{b}\t{dict(c)}")


if __name__ == "__main__":
    main()
