"""tmp/_agx_pairs.<u>.bin — a unit's read alignments handed over STAGED (aligngraph_amd/csrc/agx_host.h: pairsfile) instead of as SAM text + tmp/_reads.fa.
tools/agx_synth.cpp --pairs-bin writes it (through the engine's own line parser, its rules across line pairs and its staging); `--pairs-bin 2` also writes the
text files of the same stream.  Here (CPU): what the general loader + staging make of that text must be the staged file, byte for byte — every array, every count,
and every base of every row must come back out of the 2-bit rows the way the walk reads its k-mer tails.  The GPU side (the engine taking the file, the unit cache made
from it, outputs against the oracle) is in tests/test_gpu_parity.py."""
import os

import pytest

import harness as H
from hostsim import sim

# generator settings of the golden fixtures' kinds (clean, noisy CIGARs + multi-hits, both strands, several units, long reads with k=21) and the batch rule
SETTINGS = [
    dict(seed=1, chroms="60000", pairs=12000, coverage=5),
    dict(seed=2, chroms="50000", pairs=12000, coverage=5, read_indel=0.3, read_clip=0.2, read_badclip=0.05, multi=0.5, multi_near=0.4, read_n=0.01),
    dict(seed=3, chroms="40000,30000,20000", part=2, pairs=20000, coverage=5, mate1_left=0.3, unaligned=0.1),
    dict(seed=4, chroms="50000", pairs=8000, L=150, k=21, coverage=3, read_indel=0.2),
]


@pytest.mark.parametrize("cfg", SETTINGS, ids=lambda c: "seed%d" % c["seed"])
def test_staged_pairs_equal_what_the_loaders_make_of_the_text(cfg, built, tmp_path):
    run = H.synth(str(tmp_path / "run"), sam_seq=0, threads=3, pairs_bin=2, **cfg)
    meta = H.read_meta(run)
    for u in range(meta["units"]):
        sim.compare_staged(os.path.join(run, "tmp"), u, meta["k"], 1000000, 4)


@pytest.mark.parametrize("batch", [7, 1000, 1999, 2000, 2001, 6000])
def test_batch_boundaries(built, tmp_path, batch):
    """BATCH (AG:37) below the pair count: the line pair that loadReadAli has read when it notices the boundary is lost (AG:1258-1259); 6000 pairs in batches of 2000 and
    1000 are followed by the empty batch.  The staged file is made for one BATCH and says which."""
    run = H.synth(str(tmp_path / "run"), seed=11, chroms="30000,20000", pairs=6000, coverage=5, multi=0.6, multi_near=0.3, read_indel=0.3, read_clip=0.2, sam_seq=0, threads=2, pairs_bin=2, batch=batch)
    for u in range(2):
        sim.compare_staged(os.path.join(run, "tmp"), u, 5, batch, 3)
    with pytest.raises(sim.SimError):                                  # another BATCH than the file was made for is refused, not silently different
        sim.compare_staged(os.path.join(run, "tmp"), 0, 5, batch + 1, 3)


def test_same_stream_with_and_without_text(built, tmp_path):
    """--pairs-bin 1 (no text at all: the mode the whole-human configuration is generated in) writes the very staged files of --pairs-bin 2, for any thread count."""
    kw = dict(seed=5, chroms="40000,25000", pairs=9000, coverage=5, read_indel=0.2, multi=0.3, sam_seq=0)
    a = H.synth(str(tmp_path / "a"), threads=1, pairs_bin=2, **kw)
    b = H.synth(str(tmp_path / "b"), threads=5, pairs_bin=1, lean=1, **kw)
    for u in range(2):
        pa, pb = (os.path.join(r, "tmp", "_agx_pairs.%d.bin" % u) for r in (a, b))
        assert open(pa, "rb").read() == open(pb, "rb").read()
    assert not os.path.exists(os.path.join(b, "tmp", "_reads.fa")) and not os.path.exists(os.path.join(b, "tmp", "_reads_genome.0.bowtie"))
    assert os.path.getsize(os.path.join(b, "tmp", "_genome.fa")) == 0 and os.path.getsize(os.path.join(b, "tmp", "_genome.0.fa")) > 40000      # --lean: only what the unit loop reads


def test_checker_text_with_placeholder_reads(built, tmp_path):
    """--oracle-units: the text a checker needs for single units of a job too large to exist as text — the unit's SAM lines and a reads file with the unit's reads in their
    places and a one-base placeholder for everybody else's.  The oracle must get the same bytes out of it as out of the complete text, batch boundaries included."""
    run = H.synth(str(tmp_path / "run"), seed=9, chroms="40000,30000,20000", pairs=9000, coverage=4, read_indel=0.2, multi=0.3, sam_seq=0, threads=3, pairs_bin=2, batch=2000, oracle_units="0,2")
    for u in (0, 2):
        full = H.run_oracle(os.path.join(run, "tmp"), u, 5, 50, 4, batch=2000)
        thin = H.run_oracle(os.path.join(run, "oracle_%d" % u, "tmp"), u, 5, 50, 4, batch=2000)
        assert full == thin and full["pre"].count(b">") > 3
        sim.compare_staged(os.path.join(run, "tmp"), u, 5, 2000, 2)


def test_a_unit_alone_has_the_whole_jobs_alignments(built, tmp_path):
    """--only-units (r05: how configs[4]'s chromosomes are checked one at a time at full size): a unit made alone has the sequence, the staged read alignments (ids, batch
    boundaries) and the checker's text that it has in the whole job; only its contigs come from a generator of their own.  Its outputs still equal the oracle's on the serial executor."""
    kw = dict(seed=13, chroms="40000,30000,20000", pairs=9000, coverage=4, read_indel=0.2, multi=0.3, sam_seq=0, threads=3, lean=1, batch=2000)
    whole = H.synth(str(tmp_path / "whole"), pairs_bin=1, oracle_units="1", **kw)
    alone = H.synth(str(tmp_path / "alone"), pairs_bin=1, oracle_units="1", only_units="1", **kw)
    for f in ("tmp/_agx_pairs.1.bin", "tmp/_genome.1.fa", "oracle_1/tmp/_reads.fa", "oracle_1/tmp/_reads_genome.1.bowtie"):
        assert open(os.path.join(whole, f), "rb").read() == open(os.path.join(alone, f), "rb").read(), f
    assert not os.path.exists(os.path.join(alone, "tmp", "_agx_pairs.0.bin")) and not os.path.exists(os.path.join(alone, "tmp", "_genome.2.fa"))
    assert H.read_meta(alone)["unit_len"] == [40000, 30000, 20000]
    want = H.run_oracle(os.path.join(alone, "oracle_1", "tmp"), 1, 5, 50, 4, batch=2000)
    assert want["pre"].count(b">") > 3
    got = sim.run(os.path.join(alone, "oracle_1", "tmp"), 1, 5, 50, 4, batch=2000)          # (the kernels' lane functions, serially, on the same text)
    for key in ("initial", "pre", "extended"):
        assert got[key] == want[key], key
