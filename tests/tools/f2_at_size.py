#!/usr/bin/env python3
"""Rows f1 + f2 at the size of one BASELINE configs[3] slice (62 Mb, 15 M pairs of 2x100, --misassemblyRemoval): the second half of a whole run — refinement
(AG:2864-3195) and misassembly removal (AG:3821-4297) — by AlignGraph_amd and by the REAL reference binary on the same tmp/, final files compared byte for
byte, wall time per stage.  Needs no GPU: the first half of the run (formalize, aligners, distribute) is AlignGraph_amd's own and stops at the unit loop
where there is no device; the unit's three files come from the oracle (what the engine's are compared with everywhere else); both binaries then go on with
--resume.  The aligners are the indexed stand-ins of tests/e2e_stubs/fast/ (same answers as the Python ones, tests/test_stub_fast.py).

    python tests/tools/f2_at_size.py [--mb 62] [--pairs 15000000] [--work /tmp/agx_f2] [--skip-reference]"""
import argparse
import hashlib
import os
import resource
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import harness as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=float, default=62.0)
ap.add_argument("--pairs", type=int, default=15000000)
ap.add_argument("--work", default="/tmp/agx_f2")
ap.add_argument("--seed", type=int, default=404)
ap.add_argument("--chimeric", type=float, default=0.2)
ap.add_argument("--skip-reference", action="store_true")
ap.add_argument("--keep", action="store_true")
a = ap.parse_args()
FAST = os.path.join(ROOT, "tests", "e2e_stubs", "fast")
EXE = os.environ.get("AGX_CLI_PATH", os.path.join(ROOT, "aligngraph_amd", "AlignGraph_amd"))
BIN = os.path.join(ROOT, "build", "agx_stub_align")
FINALS = ("e.fa", "r.fa", "in.fa", "ex.fa", "corrected_e.fa", "corrected_r.fa")
COVERAGE = 5


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def run_timed(cmd, cwd, env):
    """(returncode, stdout, stderr, wall seconds) of cmd; prints the CPU seconds of it and its children (the aligner stand-ins included: the stage lines have the split)"""
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    print("  [%s: %.1f s wall, %.1f CPU-s with its children; largest process so far %.1f GB resident]" % (os.path.basename(cmd[0]), wall, r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime, r1.ru_maxrss / 1048576.0))
    return p.returncode, p.stdout, p.stderr, wall


H.build()
from aligngraph_amd import build as B  # noqa: E402
B.build()
os.makedirs(os.path.dirname(BIN), exist_ok=True)
src = os.path.join(FAST, "agx_stub_align.cpp")
if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(src):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", BIN, src])

t0 = time.perf_counter()
work = H.synth(a.work, seed=a.seed, chroms=str(int(a.mb * 1e6)), pairs=a.pairs, L=100, k=5, coverage=COVERAGE, e2e=1, sam_seq=1, chimeric=a.chimeric, contig_overlap=0.3)
shutil.rmtree(os.path.join(work, "tmp"))
sizes = {f: os.path.getsize(os.path.join(work, f)) for f in ("reads_1.fa", "genome.fa", "contigs.fa", "stub/reads_genome.sam")}
print("generated in %.1f s: %s" % (time.perf_counter() - t0, sizes), flush=True)
args = ["--read1", "reads_1.fa", "--read2", "reads_2.fa", "--contig", "contigs.fa", "--genome", "genome.fa", "--distanceLow", "100", "--distanceHigh", "1500",
        "--extendedContig", "e.fa", "--remainingContig", "r.fa", "--coverage", str(COVERAGE), "--misassemblyRemoval"]
env = dict(os.environ, PATH=FAST + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"), AGX_STUB_BIN=BIN, AGX_CLI_TIMING="1")

# ---- first half: AlignGraph_amd up to the unit loop ----
rc, out, err, wall = run_timed([EXE] + args, work, env)
print("first half (AlignGraph_amd, rc %d): %.1f s" % (rc, wall))
print(err.decode(errors="replace"), flush=True)
assert b"(0) Alignment finished" in out, out[-600:]
whole_run = rc == 0            # (a GPU box: the run went through — its finals are compared with the oracle-driven ones below)
if not whole_run:
    assert rc == 255 and b"NO HIP DEVICE" in out, out[-600:]
got_whole = {f: md5(os.path.join(work, f)) for f in FINALS} if whole_run else None

# ---- the unit loop: the oracle's files ----
t0 = time.perf_counter()
want = H.run_oracle(os.path.join(work, "tmp"), 0, 5, 50, COVERAGE)
print("oracle (unit 0): %.1f s" % (time.perf_counter() - t0), flush=True)
for key, fn in (("initial", "_initial_contigs.0.fa"), ("pre", "_pre_extended_contigs.0.fa"), ("extended", "_extended_contigs.0.fa")):
    path = os.path.join(work, "tmp", fn)
    if whole_run:
        assert open(path, "rb").read() == want[key], fn
    with open(path, "wb") as f:
        f.write(want[key])
with open(os.path.join(work, "tmp", "_checkpoint.txt"), "w") as f:
    f.write("0\n1\n")
for f in FINALS:
    if os.path.exists(os.path.join(work, f)):
        os.remove(os.path.join(work, f))

# ---- a second directory for the reference: the large read-only inputs as hard links, everything else copied ----
ref = work.rstrip("/") + ".ref"
if os.path.exists(ref):
    shutil.rmtree(ref)
if not a.skip_reference:
    for d, _, files in os.walk(work):
        rd = os.path.join(ref, os.path.relpath(d, work))
        os.makedirs(rd, exist_ok=True)
        for fn in files:
            s = os.path.join(d, fn)
            if os.path.getsize(s) > (64 << 20) and ("reads" in fn):
                os.link(s, os.path.join(rd, fn))
            else:
                shutil.copy(s, os.path.join(rd, fn))

# ---- second half: ours ----
rc, out, err, wall_ours = run_timed([EXE, "--resume"], work, env)
assert rc == 0 and b"FINISHED SUCCESSFULLY" in out, out[-600:] + err[-600:]
print("AlignGraph_amd --resume (refinement + misassembly removal): %.1f s" % wall_ours)
print(err.decode(errors="replace"), flush=True)
ours = {f: md5(os.path.join(work, f)) for f in FINALS}
print("final files:", {f: (ours[f][:8], os.path.getsize(os.path.join(work, f))) for f in FINALS})
if whole_run:
    assert ours == got_whole, "the whole run's final files differ from the oracle-driven ones"

# ---- second half: the reference ----
if not a.skip_reference:
    rc, out, err, wall_ref = run_timed([H.REF_O2, "--resume"], ref, env)
    assert rc == 0 and b"FINISHED SUCCESSFULLY" in out, out[-600:] + err[-600:]
    print("reference --resume (the same steps, the same stand-in aligners): %.1f s" % wall_ref)
    theirs = {f: md5(os.path.join(ref, f)) for f in FINALS}
    same = ours == theirs
    print("identical to the reference's final files:", same)
    for fn in sorted(os.listdir(os.path.join(ref, "tmp"))):
        if fn.endswith((".psl", ".bowtie")) and "_reads_genome" not in fn:
            eq = md5(os.path.join(ref, "tmp", fn)) == md5(os.path.join(work, "tmp", fn))
            print("  tmp/%s: %s" % (fn, "same" if eq else "DIFFERS"))
    assert same, {f: (ours[f], theirs[f]) for f in FINALS if ours[f] != theirs[f]}
if not a.keep:
    shutil.rmtree(work, ignore_errors=True)
    shutil.rmtree(ref, ignore_errors=True)
