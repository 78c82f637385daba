#!/usr/bin/env python3
"""Static instruction budget of the tracking kernel by section.

Compiles visual_odometry_amd/csrc/vo_klt.hip to assembly with -DVO_KLT_MARKS (comment lines at the borders of the
kernel's sections, nothing else changes but the memory ordering around them), counts the instructions of every class
between the marks of the 41x41 kernel (Cfg<5,6,9,7,52>, W41, no score) in both directions, and multiplies by the trip
counts per keypoint -- set-ups and iterations from the executed counters of a 50k-keypoint launch, cells from a
-DVO_KLT_COUNT_CELLS build (defaults below: the bench workload).  The sum is compared with SQ_INSTS_VALU per wave of
the PMC passes.  The compiler moves instructions across the marks (they pin memory operations only), so a section is
good to ~10 %.

    python3 scripts/klt_budget.py [--iters 36.95] [--cells 10.4] [--setups 8] [--restage 2.0] [--pmc profiles/round3_k_lk_pmc.json]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "visual_odometry_amd", "csrc", "vo_klt.hip")
KERNEL = "k_lkINS_3CfgILi5ELi6ELi9ELi7ELi52ELb0ELi4EEELb1ELb0EEEvNS_6LKArgsE"


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=float, default=36.95)
    ap.add_argument("--cells", type=float, default=10.4)
    ap.add_argument("--setups", type=float, default=8.0)
    ap.add_argument("--restage", type=float, default=2.0, help="search patches staged again inside a level, per keypoint")
    ap.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "round3_k_lk_pmc.json"))
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "klt.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-ffp-contract=fast", "-fno-slp-vectorize", "-DVO_KLT_MARKS", "-S", "--cuda-device-only", "-o", out, SRC],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, s in enumerate(lines) if s.startswith("_ZN") and KERNEL in s and s.rstrip().split(":")[0].endswith(KERNEL))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    # sections: the text between two marks belongs to the first one's name; the two directions are the two inlined
    # copies of lk_dir (the marks repeat)
    sect = collections.OrderedDict()
    cur = "prologue"
    dirs = 0
    for s in lines[start:end]:
        t = s.strip()
        m = re.match(r"; LKMARK (\w+)", t)
        if m:
            cur = m.group(1)
            if cur == "level":
                dirs += 1
            if cur == "dirend":
                cur = "epilogue%d" % dirs
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        c = classify(t.split()[0])
        if c:
            sect.setdefault(cur, collections.Counter())[c] += 1
    # trips per keypoint of each section (both directions together; a section's static count is the sum over the two
    # copies, so the trips are per copy = half)
    half = 0.5
    trips = {"prologue": 1, "level": args.setups * half, "stageA": args.setups * half, "rows": args.setups * half,
             "stageB": args.setups * half, "sums": args.setups * half, "solve": args.setups * half,
             "iter": args.iters * half, "cell": args.cells * half, "cellsums": args.cells * half, "update": args.iters * half,
             "levelend": args.setups * half, "epilogue1": 1, "epilogue2": 1}
    names = {"prologue": "kernel entry (point, masks, lane geometry)", "level": "level entry (dims, position as integer + fraction)",
             "stageA": "template patch: addresses, 12 LDS-DMA, wait", "rows": "template rows: lerps, Sobel halves, gradients, masks, 5 sums",
             "stageB": "first search patch sent ahead", "sums": "(wave sums of the 5: moved behind the next mark)",
             "solve": "5 wave sums, flat test, 2x2 inverse, bounds", "iter": "iteration head: position, lost test, cell test",
             "cell": "new cell: (re-staging test,) 42 LDS reads, 240 FMA", "cellsums": "8 wave sums, cell coefficients",
             "update": "residual from the cell, 2x2 update, exit test", "levelend": "level exit (displacement to the next level)",
             "epilogue1": "between the directions (forward result, backward start)", "epilogue2": "kernel exit (fwd-bwd test, stores, counters, head count)"}
    tot = collections.Counter()
    print("%-62s %6s | %-22s | %s" % ("section", "trips", "static v/s/lds/vmem", "per keypoint v/s/lds/vmem"))
    for k, c in sect.items():
        t = trips.get(k, 0)
        dyn = {x: c[x] * t for x in ("valu", "salu", "lds", "vmem")}
        for x in dyn:
            tot[x] += dyn[x]
        print("%-62s %6.2f | %5d %5d %4d %4d | %7.0f %6.0f %5.0f %5.0f" % (names.get(k, k), t, c["valu"], c["salu"], c["lds"], c["vmem"],
                                                                          dyn["valu"], dyn["salu"], dyn["lds"], dyn["vmem"]))
    # a patch staged again inside a level runs the staging code in the cell section's front: counted there statically
    # only once per cell; add its share (stage = address arithmetic + 12 transfers)
    print("%-62s %6s | %22s | %7.0f %6.0f %5.0f %5.0f" % ("sum", "", "", tot["valu"], tot["salu"], tot["lds"], tot["vmem"]))
    try:
        pmc = json.load(open(args.pmc))["per_wave"]
        print("%-62s %6s | %22s | %7.0f %6.0f %5.0f %5.0f" % ("measured (PMC, per wave = per keypoint)", "", "", pmc["SQ_INSTS_VALU"],
                                                              pmc["SQ_INSTS_SALU"], pmc["SQ_INSTS_LDS"], pmc["SQ_INSTS_VMEM_RD"]))
    except (OSError, KeyError, ValueError):
        pass


if __name__ == "__main__":
    sys.exit(main())
