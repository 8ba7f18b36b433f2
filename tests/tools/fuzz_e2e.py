#!/usr/bin/env python3
"""Randomised whole-run comparison on the GPU box: the real reference binary (README build, oracle/_ref/AlignGraph_ref) vs AlignGraph_amd on
seeded random small data sets and random command lines (--kMer, --coverage, --insertVariation, --ratioCheck, --uniqueExtension, --fastMap,
--misassemblyRemoval), aligner stubs of tests/e2e_stubs/ on PATH.  Compares stdout (seconds masked), the final files and the per-unit tmp/
files byte for byte.  Usage: python tests/tools/fuzz_e2e.py [--n 20] [--seed 1]"""
import argparse
import os
import random
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import harness as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
STUBS = os.path.join(ROOT, "tests", "e2e_stubs")
CLI = os.path.join(ROOT, "aligngraph_amd", "AlignGraph_amd")
FINALS = ("e.fa", "r.fa", "in.fa", "ex.fa", "corrected_e.fa", "corrected_r.fa")


def strip_time(out):
    return re.sub(rb"for \d+ seconds \(\d+ seconds for alignment\)", b"for N seconds (N seconds for alignment)", out)


rng = random.Random(a.seed)
for it in range(a.n):
    n_units = rng.choice([1, 2, 2, 3])
    chroms = ",".join(str(rng.randrange(4000, 20000)) for _ in range(n_units))
    total = sum(int(c) for c in chroms.split(","))
    k = rng.choice([5, 5, 7, 9])
    cov = rng.choice([2, 3, 4, 6])
    iv = rng.choice([20, 50, 100])
    fast = rng.random() < 0.3
    masb = (not fast) and rng.random() < 0.4          # --fastMap --misassemblyRemoval stops with CANNOT OPEN FILE! in the reference (AG:3833)
    cfg = dict(seed=rng.randrange(1, 10**6), chroms=chroms, part=1, pairs=max(400, total * rng.choice([10, 20, 40]) // 200), coverage=cov, k=k, e2e=1, sam_seq=0,
               multi=rng.choice([0, 0.1, 0.3]), contig_min=rng.choice([400, 600, 1500]), contig_max=rng.choice([2500, 4000]), contig_minus=rng.random() * 0.6,
               contig_overlap=rng.random() * 0.4, contig_dup=rng.random() * 0.2, read_indel=rng.choice([0, 0.1, 0.3]), read_clip=rng.choice([0, 0.05]),
               chimeric=0.5 if masb else 0.0)
    args = ["--read1", "reads_1.fa", "--read2", "reads_2.fa", "--contig", "contigs.fa", "--genome", "genome.fa", "--distanceLow", "100", "--distanceHigh", "1500",
            "--extendedContig", "e.fa", "--remainingContig", "r.fa", "--coverage", str(cov), "--kMer", str(k), "--insertVariation", str(iv)]
    if rng.random() < 0.3:
        args.append("--ratioCheck")
    if rng.random() < 0.3:
        args.append("--uniqueExtension")
    if fast:
        args.append("--fastMap")
    if masb:
        args.append("--misassemblyRemoval")
    src = H.synth("/tmp/agx_fz_src", **cfg)
    outs = {}
    for name, exe in (("ref", H.REF_O0), ("amd", CLI)):
        work = "/tmp/agx_fz_" + name
        shutil.rmtree(work, ignore_errors=True)
        shutil.copytree(src, work); shutil.rmtree(os.path.join(work, "tmp"))
        env = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"))
        p = subprocess.run([exe] + args, cwd=work, env=env, stdout=subprocess.PIPE)
        outs[name] = (p.returncode, strip_time(p.stdout))
    bad = None
    if outs["ref"][0] != outs["amd"][0] and not (outs["ref"][0] < 0):          # (a reference crash is the reference's business)
        bad = "exit status %d vs %d" % (outs["ref"][0], outs["amd"][0])
    elif outs["ref"][0] == 0:
        if outs["ref"][1] != outs["amd"][1]:
            bad = "stdout"
        for fn in FINALS:
            ra, rb = os.path.join("/tmp/agx_fz_ref", fn), os.path.join("/tmp/agx_fz_amd", fn)
            if os.path.exists(ra) != os.path.exists(rb) or (os.path.exists(ra) and open(ra, "rb").read() != open(rb, "rb").read()):
                bad = bad or fn
        for fn in sorted(os.listdir("/tmp/agx_fz_ref/tmp")):
            if fn.startswith(("_initial", "_pre_extended", "_extended_contigs.", "_short", "_contigs.fa", "_genome", "_chaff")) and not fn.endswith((".bt2",)):
                rb = os.path.join("/tmp/agx_fz_amd/tmp", fn)
                if not os.path.exists(rb) or open(os.path.join("/tmp/agx_fz_ref/tmp", fn), "rb").read() != open(rb, "rb").read():
                    bad = bad or ("tmp/" + fn)
    if bad:
        keep = "/tmp/agx_fz_FAILED_%d" % it
        shutil.rmtree(keep, ignore_errors=True); shutil.copytree(src, keep)
        print("MISMATCH iteration %d: %s\nargs=%r\ncfg=%r\nkept in %s" % (it, bad, args, cfg, keep), flush=True)
        print(outs["ref"][1][-300:], outs["amd"][1][-300:])
        sys.exit(1)
    print("iteration %d ok: units=%d pairs=%d rc=%d %s" % (it, n_units, cfg["pairs"], outs["ref"][0], " ".join(x for x in args if x.startswith("--") and x not in ("--read1", "--read2", "--contig", "--genome", "--distanceLow", "--distanceHigh", "--extendedContig", "--remainingContig"))), flush=True)
print("all %d whole runs identical" % a.n)
