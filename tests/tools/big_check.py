#!/usr/bin/env python3
"""One-off scale check on the GPU box: a cfg4-sized unit (default 62 M positions, one --part slice of human chr1) with a thin read set,
engine vs oracle byte for byte, plus timings and device memory.  Usage: python tests/tools/big_check.py [--genome N] [--pairs N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import aligngraph_amd as A  # noqa: E402
import harness as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=int, default=62000000)
ap.add_argument("--pairs", type=int, default=2000000)
ap.add_argument("--coverage", type=int, default=2)
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()
run = "/tmp/agx_big"
t = time.time(); H.synth(run, seed=77, chroms=str(a.genome), pairs=a.pairs, coverage=a.coverage, sam_seq=0); print("generate %.1fs" % (time.time() - t), flush=True)
tmp = os.path.join(run, "tmp")
with A.Unit(k=5, insert_variation=50, coverage=a.coverage) as u:
    t = time.time(); u.load_files(tmp, 0); t_load = time.time() - t
    t = time.time(); u.upload(); t_up = time.time() - t
    t = time.time(); u.build(); t_b1 = time.time() - t
    t = time.time(); u.build(); t_b2 = time.time() - t
    t = time.time(); got = u.finish(); t_f = time.time() - t
    st = u.stats()
    import ctypes
    fr, tot = ctypes.c_uint64(), ctypes.c_uint64()
    A.lib().agx_device_memory(0, ctypes.byref(fr), ctypes.byref(tot))
    print("device memory in use with the unit resident: %.1f GB of %.0f GB (%.0f B per position)" % ((tot.value - fr.value) / 2**30, tot.value / 2**30, (tot.value - fr.value) / st["n_pos"]))
print("load %.2fs upload %.2fs first build %.3fs rebuild %.3fs download+walk %.3fs" % (t_load, t_up, t_b1, t_b2, t_f))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
if not a.no_oracle:
    t = time.time(); want = H.run_oracle(tmp, 0, 5, 50, a.coverage); print("oracle %.1fs" % (time.time() - t))
    for k in ("initial", "pre", "extended"):
        print(k, "identical" if want[k] == got[k] else "DIFFERENT", len(got[k]))
    assert all(want[k] == got[k] for k in ("initial", "pre", "extended"))
print("extended records", got["extended"].count(b">"))
