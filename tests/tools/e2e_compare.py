#!/usr/bin/env python3
"""Whole-run comparison on the GPU box: the real reference binary (README build, oracle/_ref/AlignGraph_ref) vs AlignGraph_amd on the same
synthetic inputs with the deterministic aligner stubs of tests/e2e_stubs/.  Prints both wall times (stages after "(0) Alignment finished"
for the reference) and checks that every output file is identical.  Usage: python tests/tools/e2e_compare.py [--chroms a,b,..] [--pairs N]"""
import argparse
import filecmp
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import harness as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chroms", default="1000000,800000,600000")
ap.add_argument("--pairs", type=int, default=500000)
ap.add_argument("--coverage", type=int, default=5)
a = ap.parse_args()
STUBS = os.path.join(ROOT, "tests", "e2e_stubs")
src = H.synth("/tmp/agx_e2e_src", seed=5, chroms=a.chroms, pairs=a.pairs, coverage=a.coverage, e2e=1, contig_min=2000, contig_max=30000)
args = ["--read1", "reads_1.fa", "--read2", "reads_2.fa", "--contig", "contigs.fa", "--genome", "genome.fa", "--distanceLow", "100", "--distanceHigh", "1500",
        "--extendedContig", "e.fa", "--remainingContig", "r.fa", "--coverage", str(a.coverage)]
res = {}
for name, exe in (("reference", H.REF_O0), ("AlignGraph_amd", os.path.join(ROOT, "aligngraph_amd", "AlignGraph_amd"))):
    work = "/tmp/agx_e2e_" + name
    if os.path.exists(work):
        shutil.rmtree(work)
    shutil.copytree(src, work); shutil.rmtree(os.path.join(work, "tmp"))
    env = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"))
    p = subprocess.Popen([exe] + args, cwd=work, env=env, stdout=subprocess.PIPE, bufsize=0)
    t0 = time.perf_counter(); t_align = t_units = None; buf = b""
    while True:
        chunk = p.stdout.read(4096)
        now = time.perf_counter()
        if not chunk:
            break
        buf += chunk
        if t_align is None and b"(0) Alignment finished" in buf:
            t_align = now
        if b"(5) Contigs scaffolded" in chunk:
            t_units = now
    rc = p.wait(); t_end = time.perf_counter()
    assert rc == 0, buf[-500:]
    res[name] = dict(total=t_end - t0, front=t_align - t0, units=t_units - t_align, refinement=t_end - t_units)
    print(name, {k: round(v, 2) for k, v in res[name].items()}, flush=True)
same = True
for fn in ("e.fa", "r.fa", "in.fa", "ex.fa"):
    same &= filecmp.cmp("/tmp/agx_e2e_reference/" + fn, "/tmp/agx_e2e_AlignGraph_amd/" + fn, shallow=False)
for fn in os.listdir("/tmp/agx_e2e_reference/tmp"):
    if fn.startswith(("_initial", "_pre_extended", "_extended", "_short")):
        same &= filecmp.cmp("/tmp/agx_e2e_reference/tmp/" + fn, "/tmp/agx_e2e_AlignGraph_amd/tmp/" + fn, shallow=False)
print("all outputs identical:", same)
n_reads = 2 * a.pairs
print("unit loop: reference %.0f reads/s, AlignGraph_amd %.0f reads/s (text parsing, upload, kernels, walk, file writes) -> %.1fx" %
      (n_reads / res["reference"]["units"], n_reads / res["AlignGraph_amd"]["units"], res["reference"]["units"] / res["AlignGraph_amd"]["units"]))
assert same
