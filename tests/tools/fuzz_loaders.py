#!/usr/bin/env python3
"""Randomised sweep of the fast loaders (aligngraph_amd/csrc/agx_load.cpp) against the general loaders (agx_host.cpp + staging): seeded random
generator settings — read length, k, indel / clip / multi-hit rates, contig layout, units, BATCH sizes that put batch boundaries inside the
SAM — every unit loaded both ways on 1..8 threads and compared array by array (tests/hostsim: agx_hostsim_compare_loaders).  No GPU needed.
Usage: python tests/tools/fuzz_loaders.py [--n 40] [--seed 1]"""
import argparse
import os
import random
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from hostsim import sim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--workdir", default="/tmp/agx_fuzz_load")
a = ap.parse_args()
sim.build()
rng = random.Random(a.seed)
t0 = time.time()
declined = {0: 0, 1: 0, 2: 0, 3: 0}
n_units = 0
for it in range(a.n):
    L = rng.choice([36, 50, 75, 100, 150, 250])
    k = rng.choice([3, 5, 7, 11, 15, 21, 31]); k = min(k, L - 5)
    nu = rng.choice([1, 1, 1, 2, 3])
    chroms = ",".join(str(rng.randrange(3000, 40000)) for _ in range(nu))
    total = sum(int(c) for c in chroms.split(","))
    depth = rng.choice([4, 10, 25, 60])
    pairs = max(200, total * depth // (2 * L))
    cfg = dict(seed=rng.randrange(1, 10**6), chroms=chroms, part=rng.choice([1, 1, 2]), pairs=pairs, L=L, k=k,
               snp=rng.choice([0, 0.005, 0.02, 0.05]), indel=rng.choice([0, 0.001, 0.01]),
               contig_min=rng.choice([250, 600, 2000]), contig_max=rng.choice([800, 3000, 20000]),
               contig_minus=rng.random() * 0.6, contig_split=rng.random() * 0.5, contig_dup=rng.random() * 0.4, contig_overlap=rng.random() * 0.6,
               contig_lowid=rng.random() * 0.2, read_err=rng.choice([0, 0.005, 0.03]), read_indel=rng.choice([0, 0.1, 0.5]),
               read_clip=rng.choice([0, 0.05, 0.4]), read_n=rng.choice([0, 0.01, 0.2]), multi=rng.choice([0, 0.1, 0.5, 0.9]), multi_near=rng.choice([0, 0.3]),
               unaligned=rng.choice([0, 0.1]), frag_mean=rng.choice([300, 500, 900]), frag_sd=rng.choice([10, 30, 150, 400]), sam_seq=rng.choice([0, 1]))
    if cfg["contig_max"] < cfg["contig_min"]:
        cfg["contig_max"] = cfg["contig_min"] * 2
    run = os.path.join(a.workdir, "run")
    shutil.rmtree(run, ignore_errors=True)
    H.synth(run, **cfg)
    tmp = os.path.join(run, "tmp")
    units = sorted(int(f.split(".")[1]) for f in os.listdir(tmp) if f.startswith("_genome.") and f.endswith(".fa") and f.count(".") == 2)
    for u in units:
        for batch in (1000000, rng.choice([pairs // 3 + 1, pairs // 2, pairs, 7, max(1, pairs - 1), pairs + 1])):
            th = rng.choice([1, 2, 3, 5, 8])
            os.environ["AGX_LOAD_THREADS"] = str(th)
            try:
                rc = sim.compare_loaders(tmp, u, k, max(1, batch), th)
            except sim.SimError as e:
                print("MISMATCH iteration %d unit %d batch %d threads %d: %s\n  cfg = %r" % (it, u, batch, th, e, cfg))
                sys.exit(1)
            declined[rc] += 1
            n_units += 1
print("%d configurations, %d unit loads compared in %.0f s; fast loaders declined: contigs only %d, reads only %d, both %d" %
      (a.n, n_units, time.time() - t0, declined[1], declined[2], declined[3]))
