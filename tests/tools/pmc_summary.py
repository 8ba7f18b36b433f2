#!/usr/bin/env python3
"""Per-kernel, per-launch averages of the rocprofv3 counter passes of tests/tools/profile.sh -> <tag>_pmc_hbm.csv, <tag>_pmc_sq.csv, and a copy
of the kernel stats as <tag>_kernel_stats.csv.  Usage: python tests/tools/pmc_summary.py gpurun_out/<tag> <tag>"""
import collections
import csv
import glob
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]


def per_launch(sub):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    acc, disp = collections.defaultdict(float), collections.defaultdict(set)
    if not f:
        return acc, disp
    for r in csv.DictReader(open(f[0])):
        acc[(r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return acc, disp


fa, fd = per_launch("fetch")
wa, wd = per_launch("write")
names = sorted(set(k for k, _ in fa) | set(k for k, _ in wa), key=lambda k: -(fa.get((k, "FETCH_SIZE"), 0) + wa.get((k, "WRITE_SIZE"), 0)))
with open(os.path.join(out, tag + "_pmc_hbm.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch"])
    for k in names:
        n = max(len(fd.get(k, ())), len(wd.get(k, ())), 1)
        w.writerow([k, n, round(fa.get((k, "FETCH_SIZE"), 0) / max(len(fd.get(k, ())), 1), 1), round(wa.get((k, "WRITE_SIZE"), 0) / max(len(wd.get(k, ())), 1), 1)])
sa, sd = per_launch("sq")
ctrs = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"]
with open(os.path.join(out, tag + "_pmc_sq.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches"] + [c + "_per_launch" for c in ctrs])
    for k in sorted(sd, key=lambda k: -sa.get((k, "SQ_WAVE_CYCLES"), 0)):
        n = max(len(sd[k]), 1)
        w.writerow([k, n] + [int(sa.get((k, c), 0) / n) for c in ctrs])
st = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, tag + "_kernel_stats.csv"))
