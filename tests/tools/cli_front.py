#!/usr/bin/env python3
"""AlignGraph_amd's front end at (a fraction of) the size of BASELINE configs[2] on the GPU box: wall time of every stage up to the unit loop (AGX_CLI_TIMING), threaded (the default) and
line by line (AGX_CLI_SERIAL=1), with the md5 of the three read files both ways.  The run is stopped when the unit loop is over (the refinement's stand-in aligner takes minutes at this size
and is not what is measured).   Usage: python tests/tools/cli_front.py [--scale 0.25]"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import agx_data as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=0.25)
ap.add_argument("--work", default="/tmp/agx_cli_front")
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
work = D.synth(a.work, seed=1000, chroms=",".join(map(str, chroms)), pairs=pairs, L=100, k=5, coverage=5, e2e=1, sam_seq=1)
print("generated %d pairs in %.1f s: reads_1.fa %.2f GB" % (pairs, time.perf_counter() - t0, os.path.getsize(os.path.join(work, "reads_1.fa")) / 1e9), flush=True)
args = ["--read1", "reads_1.fa", "--read2", "reads_2.fa", "--contig", "contigs.fa", "--genome", "genome.fa", "--distanceLow", "100", "--distanceHigh", "1500",
        "--extendedContig", "e.fa", "--remainingContig", "r.fa", "--coverage", "5"]
sums = {}
for mode, extra in (("threaded", {}), ("line by line", {"AGX_CLI_SERIAL": "1"})):
    shutil.rmtree(os.path.join(work, "tmp"), ignore_errors=True)
    env = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"), AGX_CLI_TIMING="1", **extra)
    p = subprocess.Popen([EXE] + args, cwd=work, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    print("---- %s" % mode, flush=True)
    for line in p.stderr:
        line = line.decode(errors="replace").rstrip()
        if line.startswith("[agx cli]"):
            print(line, flush=True)
        if "unit loop" in line:
            break
    p.kill(); p.wait()
    sums[mode] = {f: md5(os.path.join(work, "tmp", f)) for f in ("_reads.fa", "_reads_1.fa", "_reads_2.fa")}
print("read files identical both ways:", sums["threaded"] == sums["line by line"], sums["threaded"])
shutil.rmtree(work, ignore_errors=True)
