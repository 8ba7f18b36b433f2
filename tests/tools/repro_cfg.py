#!/usr/bin/env python3
"""One generator configuration (a fuzz_parity.py failure line's cfg=...) through the oracle and the engine under several settings of the environment.
    python tests/tools/repro_cfg.py "<cfg dict>" [unit] [--hostsim]"""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from conftest import graph_mismatch  # noqa: E402

cfg = ast.literal_eval(sys.argv[1])
only = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None
run = H.synth("/tmp/agx_repro/run", **cfg)
meta = H.read_meta(run)
tmp = os.path.join(run, "tmp")
if "--hostsim" in sys.argv:
    from hostsim import sim
    sim.build()
    settings = [{}]
else:
    import aligngraph_amd as A
    settings = [{}, {"AGX_UPLOAD_WINDOWS": "1"}, {"AGX_UPLOAD_WINDOWS": "2"}, {"AGX_UPLOAD_WINDOWS": "8"}, {"AGX_NO_TILED_UPLOAD": "1"}, {"AGX_UPLOAD_WINDOWS": "8", "AGX_DEBUG_SYNC": "1"}]
for u in range(meta["units"]):
    if only is not None and u != only:
        continue
    o = H.run_oracle(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
    for env in settings:
        for n in ("AGX_UPLOAD_WINDOWS", "AGX_NO_TILED_UPLOAD", "AGX_DEBUG_SYNC"):
            os.environ.pop(n, None)
        os.environ.update(env)
        if "--hostsim" in sys.argv:
            g = sim.run(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
        else:
            with A.Unit(k=meta["k"], insert_variation=meta["insert_variation"], coverage=meta["coverage"], keep_counts=True) as un:
                un.load_files(tmp, u); un.upload(); un.build()
                g = {"graph": un.graph()}; g.update(un.finish()); st = un.stats()
        bad = graph_mismatch(o["graph"], g["graph"]) or next((k for k in ("initial", "pre", "extended") if o[k] != g[k]), None)
        print("unit %d %r: %s (%d nodes)" % (u, env, "ok" if bad is None else "MISMATCH " + str(bad), g["graph"]["n_nodes"]), flush=True)
