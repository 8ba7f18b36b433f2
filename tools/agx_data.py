"""Seeded synthetic inputs: builds and drives tools/agx_synth (writes a reference-style run directory: genome.fa, contigs.fa, tmp/_genome.N.fa,
tmp/_contigs.fa, tmp/_contigs_genome.N.psl, tmp/_reads.fa, tmp/_reads_genome.N.bowtie, ...).  Plain data generation — nothing here knows
the algorithm, the oracle or the reference; bench.py, the tests and the tools all get their inputs through it."""
import os
import shutil
import subprocess
import threading

_build_lock = threading.Lock()      # (tests generate several units on several threads: one of them compiles, into a scratch name that is then renamed)


def _compile(cmd, out):
    tmp = "%s.%d.tmp" % (out, os.getpid())
    subprocess.check_call(cmd + ["-o", tmp])
    os.replace(tmp, out)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SYNTH = os.path.join(ROOT, "build", "agx_synth")


def build():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    src = os.path.join(HERE, "agx_synth.cpp")
    with _build_lock:
        if not os.path.exists(SYNTH) or os.path.getmtime(SYNTH) < os.path.getmtime(src):
            _compile(["g++", "-O2", "-std=c++17", "-pthread", src], SYNTH)


SYNTH_BIN = os.path.join(ROOT, "build", "agx_synth_bin")
CSRC = os.path.join(ROOT, "aligngraph_amd", "csrc")


def build_bin():
    """build/agx_synth_bin: the generator with the staged-pairs mode (--pairs-bin), compiled together with the engine's host-side loader sources (no HIP)."""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    src = [os.path.join(HERE, "agx_synth.cpp")] + [os.path.join(CSRC, f) for f in ("agx_host.cpp", "agx_walk.cpp", "agx_load.cpp")]
    deps = src + [os.path.join(CSRC, f) for f in ("agx_host.h", "agx_parse.h", "agx_core.h")]
    with _build_lock:
        if not os.path.exists(SYNTH_BIN) or any(os.path.getmtime(SYNTH_BIN) < os.path.getmtime(d) for d in deps):
            _compile(["g++", "-O2", "-std=c++17", "-pthread", "-DAGX_SYNTH_WITH_ENGINE"] + src, SYNTH_BIN)


def synth(out, **kw):
    """kw: seed=1, chroms="50000", pairs=10000, L=100, ... (see tools/agx_synth.cpp Params)."""
    staged = int(kw.get("pairs_bin", 0)) != 0
    if staged:
        build_bin()
    else:
        build()
    if os.path.exists(out):
        shutil.rmtree(out)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    cmd = [SYNTH_BIN if staged else SYNTH, "--out", out]
    for k, v in kw.items():
        cmd += ["--" + k.replace("_", "-"), str(v)]
    subprocess.check_call(cmd)
    return out


def read_meta(run_dir):
    meta = {"unit_len": []}
    with open(os.path.join(run_dir, "synth_meta.txt")) as f:
        for line in f:
            t = line.split()
            if t[0] == "unit":
                meta["unit_len"].append(int(t[3]))
            else:
                meta[t[0]] = int(t[1])
    return meta
