#!/usr/bin/env python3
"""Does a build slow down while another unit's upload runs?  Unit A is rebuilt N times alone, then N times while a second thread keeps
uploading unit B (copies + upload-time kernels).  Usage (GPU box): python tests/tools/build_vs_upload.py [--genome 30000000 --pairs 5000000]"""
import argparse, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import agx_data as D
import aligngraph_amd as A
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=12); ap.add_argument("--genome", default="30000000"); ap.add_argument("--pairs", type=int, default=5000000); ap.add_argument("--torch", action="store_true"); ap.add_argument("--uploaders", type=int, default=1)
a = ap.parse_args()
if a.torch:
    import torch
    torch.cuda.init(); torch.cuda.synchronize(); print('HSA_ENABLE_SDMA', os.environ.get('HSA_ENABLE_SDMA'))
run = "/tmp/agx_sweep_time_%s_%d" % (a.genome, a.pairs)
if not os.path.exists(os.path.join(run, "tmp")):
    D.synth(run, seed=1000, chroms=a.genome, pairs=a.pairs, L=100, k=5, coverage=5, threads=16)
tmp = os.path.join(run, "tmp")
ua = A.Unit(k=5, insert_variation=50, coverage=5); ua.load_files(tmp, 0)
ubs = [A.Unit(k=5, insert_variation=50, coverage=5) for _ in range(a.uploaders)]
for ub in ubs:
    ub.load_files(tmp, 0)
ua.upload(); ua.build()


def spans(n):
    out = []
    for _ in range(n):
        t = time.perf_counter(); ua.build(); out.append((time.perf_counter() - t) * 1e3)
    return out


print("alone:            build wall ms", " ".join("%.1f" % x for x in spans(a.n)), "device span", round(ua.stats()["ms_build_span"], 2))
stop = False


def uploader(ub):
    while not stop:
        ub.upload(); ub.release()


ths = [threading.Thread(target=uploader, args=(ub,)) for ub in ubs]
for th in ths:
    th.start()
time.sleep(0.05)
print("beside uploads:   build wall ms", " ".join("%.1f" % x for x in spans(a.n)), "device span", round(ua.stats()["ms_build_span"], 2))
stop = True
for th in ths:
    th.join()
print("alone again:      build wall ms", " ".join("%.1f" % x for x in spans(a.n)), "device span", round(ua.stats()["ms_build_span"], 2))
