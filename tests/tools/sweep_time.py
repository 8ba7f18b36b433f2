#!/usr/bin/env python3
"""Kernel timing without the host walk: builds the bench unit N times and prints the mean per-kernel times from agx_unit_stats.
Usage (GPU box): [AGX_LIB_PATH=...] python tests/tools/sweep_time.py [--n 20]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import agx_data as D
import aligngraph_amd as A
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=20); ap.add_argument("--genome", default="4600000"); ap.add_argument("--pairs", type=int, default=1000000); ap.add_argument("--idle-ms", type=float, default=0.0, help="sleep between builds (does an idle GPU clock down?)")
a = ap.parse_args()
run = "/tmp/agx_sweep_time_%s_%d" % (a.genome, a.pairs)
if not os.path.exists(os.path.join(run, "tmp")):
    D.synth(run, seed=1000, chroms=a.genome, pairs=a.pairs, L=100, k=5, coverage=5)
with A.Unit(k=5, insert_variation=50, coverage=5, flags=A.AGX_FLAG_TIME_SECTIONS) as u:
    u.load_files(os.path.join(run, "tmp"), 0); u.upload()
    acc = {}
    for i in range(a.n + 2):
        if a.idle_ms: time.sleep(a.idle_ms * 1e-3)
        u.build()
        st = u.stats()
        if i >= 2:
            for k, v in st.items():
                if k.startswith("ms_"): acc[k] = acc.get(k, 0.0) + v / a.n
    print(os.environ.get("AGX_LIB_PATH", "base"), " ".join("%s=%.3f" % (k[3:], v) for k, v in acc.items() if v), "nodes", st["n_nodes"], "entries", st["n_tile_entries"], "mid", st["n_mid_tiles"], "big", st["n_big_tiles"])
