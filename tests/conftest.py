import io
import json
import os
import sys
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = sorted(f[:-7] for f in os.listdir(GOLDEN) if f.endswith(".tar.gz") and f != "e2e.tar.gz")   # e2e: whole-run fixture of tests/test_cli.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class GoldenCase:
    """A tests/golden/<name>.tar.gz unpacked into a scratch directory."""

    def __init__(self, name, dest):
        self.name = name
        self.dir = str(dest)
        with tarfile.open(os.path.join(GOLDEN, name + ".tar.gz")) as tar:
            tar.extractall(self.dir)
        with open(os.path.join(self.dir, "params.json")) as f:
            self.params = json.load(f)
        self.tmp = os.path.join(self.dir, "tmp")

    def expected(self, coverage, unit):
        out = {}
        for key, fn in (("initial", "_initial_contigs"), ("pre", "_pre_extended_contigs"), ("extended", "_extended_contigs")):
            with open(os.path.join(self.dir, "expected", str(coverage), "%s.%d.fa" % (fn, unit)), "rb") as f:
                out[key] = f.read()
        return out


@pytest.fixture(params=GOLDEN_CASES)
def golden(request, tmp_path):
    return GoldenCase(request.param, tmp_path)


@pytest.fixture(scope="session")
def built():
    """Builds the oracle, the generator and the CPU test executor once per session."""
    import harness
    from hostsim import sim
    harness.build()
    sim.build()
    return True


def graph_mismatch(go, gs, counts=True):
    """First difference between two canonical graph dumps (oracle order-of-insertion edges are sorted here), or None."""
    import numpy as np
    if go["n_pos"] != gs["n_pos"] or go["n_nodes"] != gs["n_nodes"]:
        return "sizes: n_pos %d/%d n_nodes %d/%d" % (go["n_pos"], gs["n_pos"], go["n_nodes"], gs["n_nodes"])
    keys = ["node_start", "node_key", "node_slen"] + (["node_cnt"] if counts else [])
    for k in keys:
        if not np.array_equal(go[k], gs[k]):
            return k
    if go["n_edges"] != gs["n_edges"]:
        return "n_edges %d/%d" % (go["n_edges"], gs["n_edges"])

    def canon(g):
        es = g["edge_start"].astype(np.int64)
        owner = np.repeat(np.arange(g["n_nodes"]), np.diff(es))
        return g["edge_dst"][np.lexsort((g["edge_dst"], owner))]
    if not np.array_equal(go["edge_start"], gs["edge_start"]) or not np.array_equal(canon(go), canon(gs)):
        return "edges"
    return None


def write_pileup_unit(run, n_pairs, spacing=300, genome_len=60000, seed=5):
    """A hand-made unit (no contigs) in which `n_pairs` pairs share ONE left-mate alignment while their right mates lie `spacing` apart:
    at each of the ~96 positions of the left mate every pair adds a node variant of its own (the mate positions differ by more than
    2 * insertVariation + 25, AG:1296-1307) — what deep repeats under a wide --distanceHigh do.  Returns the tmp/ directory."""
    import random
    rnd = random.Random(seed)
    tmp = os.path.join(run, "tmp")
    os.makedirs(tmp, exist_ok=True)
    g = "".join(rnd.choice("ACGT") for _ in range(genome_len))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    with open(os.path.join(tmp, "_genome.0.fa"), "w") as f:
        f.write(">0\n" + "".join(g[i:i + 60] + "\n" for i in range(0, genome_len, 60)))
    open(os.path.join(tmp, "_contigs.fa"), "w").close()
    open(os.path.join(tmp, "_contigs_genome.0.psl"), "w").close()
    left = 1000
    with open(os.path.join(tmp, "_reads.fa"), "w") as rf, open(os.path.join(tmp, "_reads_genome.0.bowtie"), "w") as sf:
        for i in range(n_pairs):
            right = 2000 + spacing * i
            assert right + 100 <= genome_len
            m1 = g[left:left + 100]
            m2 = "".join(comp[c] for c in reversed(g[right:right + 100]))
            rf.write(">%d\n%s\n>%d\n%s\n" % (i, m1, i, m2))
            sf.write("%d\t99\t0\t%d\t42\t100M\t=\t%d\t%d\t*\t*\n" % (i, left + 1, right + 1, right + 100 - left))
            sf.write("%d\t147\t0\t%d\t42\t100M\t=\t%d\t%d\t*\t*\n" % (i, right + 1, left + 1, -(right + 100 - left)))
    return tmp


def write_long_deletion_unit(run, n_pairs, deletion=60, genome_len=200000, seed=7):
    """A hand-made unit (no contigs) whose left mates all carry ONE long deletion (40M<deletion>D60M: the longest the identity filter of loadReadAli lets through is
    two thirds of the read, AG:1261): their arrivals span 160 positions, i.e. up to four tiles — more than a tile list's window looks back over at 2x100 bp (r05:
    agx_tile_lookback), so they go through the list of long hits, or, beyond AGX_LONG_MAX of them, send the whole unit through the scatter fallback.  Returns tmp/."""
    import random
    rnd = random.Random(seed)
    tmp = os.path.join(run, "tmp")
    os.makedirs(tmp, exist_ok=True)
    g = "".join(rnd.choice("ACGT") for _ in range(genome_len))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    with open(os.path.join(tmp, "_genome.0.fa"), "w") as f:
        f.write(">0\n" + "".join(g[i:i + 60] + "\n" for i in range(0, genome_len, 60)))
    open(os.path.join(tmp, "_contigs.fa"), "w").close()
    open(os.path.join(tmp, "_contigs_genome.0.psl"), "w").close()
    with open(os.path.join(tmp, "_reads.fa"), "w") as rf, open(os.path.join(tmp, "_reads_genome.0.bowtie"), "w") as sf:
        for i in range(n_pairs):
            left = rnd.randrange(100, genome_len - 1000)
            right = left + 400 + rnd.randrange(0, 60)
            plain = rnd.random() < 0.3                      # some ordinary pairs between them
            m1 = g[left:left + 100] if plain else g[left:left + 40] + g[left + 40 + deletion:left + 100 + deletion]
            m2 = "".join(comp[c] for c in reversed(g[right:right + 100]))
            rf.write(">%d\n%s\n>%d\n%s\n" % (i, m1, i, m2))
            sf.write("%d\t99\t0\t%d\t42\t%s\t=\t%d\t%d\t*\t*\n" % (i, left + 1, "100M" if plain else "40M%dD60M" % deletion, right + 1, right + 100 - left))
            sf.write("%d\t147\t0\t%d\t42\t100M\t=\t%d\t%d\t*\t*\n" % (i, right + 1, left + 1, -(right + 100 - left)))
    return tmp
