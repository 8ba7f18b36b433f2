#!/usr/bin/env python3
"""Randomised parity sweep: seeded random generator settings (read length, k, indel/clip/multi-hit rates, contig layout, insert spread,
coverage threshold, units), each compared byte for byte (three output files + node/edge tables) between the oracle and
  --engine hostsim : the kernels' per-lane functions on the CPU serial executor (tests/hostsim; no GPU needed), or
  --engine gpu     : the HIP engine through the C-ABI.
Usage: python tests/tools/fuzz_parity.py [--n 40] [--seed 1] [--engine hostsim|gpu]"""
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
from conftest import graph_mismatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--engine", default="hostsim", choices=["hostsim", "gpu"])
ap.add_argument("--workdir", default="/tmp/agx_fuzz")
a = ap.parse_args()

if a.engine == "hostsim":
    from hostsim import sim
    sim.build()

    sim_rng = random.Random(a.seed * 104729 + 7)

    def run(tmp, u, k, iv, cov):
        # r06: every unit with its own draw of how the walk graph arrives (all at once, or window by window into arrays full of junk) and of who walks it
        for name in ("AGX_SIM_STREAM", "AGX_SIM_ASSISTANT", "AGX_WALK_SPLIT_MIN", "AGX_WALK_SPLIT_WALKERS", "AGX_WALK_SPLIT_WARMUP", "AGX_WALK_POISON", "AGX_WALK_EVEN_CUTS"):
            os.environ.pop(name, None)
        r = sim_rng
        if r.random() < 0.6:
            os.environ["AGX_SIM_STREAM"] = r.choice(["1", "3,30", "7,0", "16,10"])
        if r.random() < 0.7:
            os.environ["AGX_SIM_ASSISTANT"] = "1"; os.environ["AGX_WALK_SPLIT_MIN"] = "0"; os.environ["AGX_WALK_SPLIT_WALKERS"] = str(r.choice([2, 3, 5, 8, 16])); os.environ["AGX_WALK_SPLIT_WARMUP"] = str(r.choice([200, 2000, 20000])); os.environ["AGX_WALK_POISON"] = "1"
            if r.random() < 0.3:
                os.environ["AGX_WALK_EVEN_CUTS"] = "1"
        return sim.run(tmp, u, k, iv, cov, graph=True)
else:
    import aligngraph_amd as A
    knob_rng = random.Random(a.seed * 7919 + 13)
    KNOBS = ("AGX_UPLOAD_WINDOWS", "AGX_STREAM_PIECES", "AGX_WALK_SPLIT_MIN", "AGX_WALK_SPLIT_WALKERS", "AGX_WALK_SPLIT_WARMUP", "AGX_ROW_DIFF", "AGX_NO_TILED_UPLOAD", "AGX_WALK_POISON", "AGX_WALK_EVEN_CUTS")

    def run(tmp, u, k, iv, cov):
        # r06: every unit with its own draw of the round's paths — the sweep by 1-8 windows of a tile-ordered upload, the download streamed in 1-16 windows into 2-16 walkers
        # (or taken whole), the rows as differences, r05's upload forms, one-shot units (whose landing memory is their dead staged inputs)
        for name in KNOBS:
            os.environ.pop(name, None)
        r = knob_rng
        if r.random() < 0.7:
            os.environ["AGX_UPLOAD_WINDOWS"] = str(r.choice([1, 2, 3, 5, 8]))
        stream = r.random() < 0.6
        if stream:
            os.environ["AGX_STREAM_PIECES"] = str(r.choice([1, 2, 3, 7, 16]))
        if r.random() < 0.7:
            os.environ["AGX_WALK_SPLIT_MIN"] = "0"; os.environ["AGX_WALK_SPLIT_WALKERS"] = str(r.choice([2, 3, 4, 8, 16])); os.environ["AGX_WALK_SPLIT_WARMUP"] = str(r.choice([200, 2000, 20000])); os.environ["AGX_WALK_POISON"] = "1"
            if r.random() < 0.3:
                os.environ["AGX_WALK_EVEN_CUTS"] = "1"
        if r.random() < 0.3:
            os.environ["AGX_ROW_DIFF"] = "1"
        if r.random() < 0.15:
            os.environ["AGX_NO_TILED_UPLOAD"] = "1"
        one_shot = r.random() < 0.4
        with A.Unit(k=k, insert_variation=iv, coverage=cov, keep_counts=True, flags=A.AGX_FLAG_ONE_SHOT if one_shot else 0) as un:
            un.load_files(tmp, u); un.upload(); un.build()
            graph = un.graph()
            if not stream and r.random() < 0.5:
                un.download()
            out = un.finish(); out["graph"] = graph
        return out

rng = random.Random(a.seed)
t0 = time.time()
for it in range(a.n):
    L = rng.choice([36, 50, 75, 100, 150, 250])
    k = rng.choice([3, 5, 7, 11, 15, 21, 31]); k = min(k, L - 5)
    n_units = rng.choice([1, 1, 1, 2, 3])
    chroms = ",".join(str(rng.randrange(3000, 40000)) for _ in range(n_units))
    total = sum(int(c) for c in chroms.split(","))
    depth = rng.choice([4, 10, 25, 60])
    cfg = dict(seed=rng.randrange(1, 10**6), chroms=chroms, part=rng.choice([1, 1, 2]), pairs=max(200, total * depth // (2 * L)), L=L, k=k,
               coverage=rng.choice([1, 2, 3, 5, 8, 20]), insert_variation=rng.choice([0, 10, 50, 200]),
               snp=rng.choice([0, 0.005, 0.02, 0.05]), indel=rng.choice([0, 0.001, 0.01]),
               contig_min=rng.choice([250, 600, 2000]), contig_max=rng.choice([800, 3000, 20000]),
               contig_minus=rng.random() * 0.6, contig_split=rng.random() * 0.5, contig_dup=rng.random() * 0.4, contig_overlap=rng.random() * 0.6,
               contig_lowid=rng.random() * 0.2, read_err=rng.choice([0, 0.005, 0.03]), read_indel=rng.choice([0, 0.1, 0.5]),
               read_clip=rng.choice([0, 0.05, 0.4]), read_n=rng.choice([0, 0.01]), multi=rng.choice([0, 0.1, 0.5]), multi_near=rng.choice([0, 0.3]),
               unaligned=rng.choice([0, 0.1]), frag_mean=rng.choice([300, 500, 900]), frag_sd=rng.choice([10, 30, 150, 400]), sam_seq=0)
    if cfg["contig_max"] < cfg["contig_min"]:
        cfg["contig_max"] = cfg["contig_min"] * 2
    runp = H.synth(os.path.join(a.workdir, "run"), **cfg)
    meta = H.read_meta(runp)
    tmp = os.path.join(runp, "tmp")
    for u in range(meta["units"]):
        o = H.run_oracle(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
        try:
            g = run(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"])
        except Exception as e:                                     # the engine rejects what the reference would misread; the oracle must agree it is odd
            print("iteration %d unit %d: engine refused (%s) cfg=%r" % (it, u, e, cfg), flush=True)
            continue
        bad = graph_mismatch(o["graph"], g["graph"])
        if bad is None:
            bad = next((key for key in ("initial", "pre", "extended") if o[key] != g[key]), None)
        if bad is not None:
            keep = os.path.join(a.workdir, "FAILED_%d" % it)
            shutil.rmtree(keep, ignore_errors=True); shutil.copytree(runp, keep)
            print("MISMATCH iteration %d unit %d: %s\ncfg=%r\nenv=%r\nkept in %s" % (it, u, bad, cfg, {n: os.environ[n] for n in os.environ if n.startswith("AGX_")}, keep), flush=True)
            sys.exit(1)
    print("iteration %d ok: L=%d k=%d units=%d pairs=%d nodes=%d edges=%d (%.0fs)" % (it, L, k, meta["units"], cfg["pairs"], g["graph"]["n_nodes"], g["graph"]["n_edges"], time.time() - t0), flush=True)
print("all %d configurations identical" % a.n)
