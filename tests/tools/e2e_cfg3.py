#!/usr/bin/env python3
"""One whole AlignGraph_amd run at the size of BASELINE configs[2] (A. thaliana shape: 5 chromosomes, 119 Mb, 20 M pairs of 2x100) on the GPU box, the aligners
replaced by the deterministic stubs of tests/e2e_stubs/ (they replay the SAM / PSL the generator derived): wall time per stage (AGX_CLI_TIMING), then the
ORACLE-driven pipeline — the same tmp/ with every unit's three output files replaced by the oracle's, refinement run again through --resume — must give the
same final files.  (The real reference needs ~40 minutes for the unit loop at this size; whole runs against it: tests/tools/e2e_compare.py at 2.4 Mb,
tests/test_cli.py on the golden runs.)   Usage: python tests/tools/e2e_cfg3.py [--scale 1.0] [--sam-seq 1]"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import harness as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--sam-seq", type=int, default=1)
ap.add_argument("--work", default="/tmp/agx_e2e_cfg3")
a = ap.parse_args()
CFG3 = [30427671, 19698289, 23459830, 18585056, 26975502]
chroms = [int(c * a.scale) for c in CFG3]
pairs = int(20000000 * a.scale)
STUBS = os.path.join(ROOT, "tests", "e2e_stubs")
EXE = os.path.join(ROOT, "aligngraph_amd", "AlignGraph_amd")


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


t0 = time.perf_counter()
work = H.synth(a.work, seed=1000, chroms=",".join(map(str, chroms)), pairs=pairs, L=100, k=5, coverage=5, e2e=1, sam_seq=a.sam_seq)
shutil.rmtree(os.path.join(work, "tmp"))                      # the generator's own tmp/: the run makes its own from the user-level files
print("generated in %.1f s: %s" % (time.perf_counter() - t0, {f: os.path.getsize(os.path.join(work, f)) for f in ("reads_1.fa", "genome.fa", "contigs.fa", "stub/reads_genome.sam")}), flush=True)
args = ["--read1", "reads_1.fa", "--read2", "reads_2.fa", "--contig", "contigs.fa", "--genome", "genome.fa", "--distanceLow", "100", "--distanceHigh", "1500",
        "--extendedContig", "e.fa", "--remainingContig", "r.fa", "--coverage", "5"]
env = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"), AGX_CLI_TIMING="1")
t0 = time.perf_counter()
p = subprocess.run([EXE] + args, cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
wall = time.perf_counter() - t0
assert p.returncode == 0, p.stdout[-800:] + p.stderr[-800:]
print("AlignGraph_amd: %.1f s wall" % wall)
print(p.stderr.decode(errors="replace"))
print(p.stdout.decode(errors="replace")[-300:])
got = {f: md5(os.path.join(work, f)) for f in ("e.fa", "r.fa", "in.fa", "ex.fa")}
units = len(chroms)
# ---- the oracle-driven pipeline: every unit's three files from the oracle, refinement again ----
want, errs = {}, []


def oracle(u):
    try:
        want[u] = H.run_oracle(os.path.join(work, "tmp"), u, 5, 50, 5)
    except BaseException as e:
        errs.append(e)


t0 = time.perf_counter()
th = [threading.Thread(target=oracle, args=(u,)) for u in range(units)]
for t in th:
    t.start()
for t in th:
    t.join()
assert not errs, errs
print("oracle: %d units in %.1f s" % (units, time.perf_counter() - t0), flush=True)
for u in range(units):
    for key, fn in (("initial", "_initial_contigs.%d.fa"), ("pre", "_pre_extended_contigs.%d.fa"), ("extended", "_extended_contigs.%d.fa")):
        path = os.path.join(work, "tmp", fn % u)
        same = open(path, "rb").read() == want[u][key]
        if not same:
            print("unit %d: %s differs from the oracle" % (u, fn % u))
        with open(path + ".new", "wb") as f:
            f.write(want[u][key])
        os.replace(path + ".new", path)
for f in ("e.fa", "r.fa", "in.fa", "ex.fa"):
    os.remove(os.path.join(work, f))
t0 = time.perf_counter()
p2 = subprocess.run([EXE, "--resume"], cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
assert p2.returncode == 0, p2.stdout[-800:] + p2.stderr[-800:]
print("oracle-driven refinement (--resume behind the unit loop): %.1f s" % (time.perf_counter() - t0))
again = {f: md5(os.path.join(work, f)) for f in ("e.fa", "r.fa", "in.fa", "ex.fa")}
print("final files:", got)
print("identical to the oracle-driven pipeline:", got == again)
assert got == again
