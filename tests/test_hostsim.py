"""CPU checks of the engine's algorithm: the per-lane kernel functions of aligngraph_amd/csrc/agx_core.h, run serially by
tests/hostsim, plus the product's host loaders and host walk, against the oracle and the reference's golden vectors.
(The kernels themselves run in tests/test_gpu_parity.py, -m gpu.)"""
import os

import pytest

import harness as H
from conftest import graph_mismatch
from hostsim import sim


def test_golden_vectors(golden, built):
    p = golden.params
    for cov in p["coverages"]:
        for u in range(p["units"]):
            got = sim.run(golden.tmp, u, p["k"], p["insert_variation"], cov)
            exp = golden.expected(cov, u)
            for key in ("initial", "pre", "extended"):
                assert got[key] == exp[key], "%s cov=%d unit=%d %s" % (golden.name, cov, u, key)


CONFIGS = [
    dict(seed=101, chroms="40000", pairs=12000, coverage=5, contig_min=1500, contig_max=3000),
    dict(seed=102, chroms="30000", pairs=9000, coverage=3, L=50, k=8, read_indel=0.3, read_clip=0.3, indel=0.005, multi=0.3),
    dict(seed=103, chroms="30000", pairs=9000, coverage=5, frag_sd=300, insert_variation=10),          # buckets outgrow the LDS window
    dict(seed=104, chroms="20000,20000", part=2, pairs=8000, coverage=4, L=150, k=21, contig_min=300, contig_max=2000, contig_overlap=0.5,
         contig_dup=0.3, contig_split=0.5, contig_minus=1.0),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "seed%d" % c["seed"])
@pytest.mark.parametrize("maxv", [0, 1])
def test_node_and_edge_tables_match_oracle(cfg, maxv, built, tmp_path):
    kw = dict(cfg)
    run = H.synth(str(tmp_path / "run"), sam_seq=0, **kw)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    for u in range(meta["units"]):
        o = H.run_oracle(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
        s = sim.run(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], maxv_first=maxv, graph=True)
        assert graph_mismatch(o["graph"], s["graph"]) is None
        for key in ("initial", "pre", "extended"):
            assert o[key] == s[key], key
        if maxv == 1 or cfg.get("frag_sd") == 300:
            assert s["n_big_tiles"] > 0          # the global-scratch fallback really ran


def test_sparse_record_table_and_fetch_hook(built, tmp_path, monkeypatch):
    # The walk reads node records from the sparse table of "special" ids (agx_core.h, walk preparation); only the +1000 position skip
    # inside records longer than 100 kb (AG:2194-2202) can put it on another id, which then comes through the fetch hook.  A 300 kb
    # unit with 120-200 kb contigs exercises both; with the table cut down to the side ids nearly every record takes the fetch path.
    run = H.synth(str(tmp_path / "run"), seed=105, chroms="300000", pairs=60000, coverage=5, contig_min=120000, contig_max=200000, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    o = H.run_oracle(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    assert max(len(r) for r in o["pre"].split(b">")) > 100000
    s = sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    monkeypatch.setenv("AGX_SIM_ASSISTANT", "1")             # the outputs written by a second thread while the walk goes on, as in the engine
    a = sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    monkeypatch.delenv("AGX_SIM_ASSISTANT")
    monkeypatch.setenv("AGX_SIM_SPARSE_MIN", "1")
    m = sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    for key in ("initial", "pre", "extended"):
        assert o[key] == s[key] == m[key] == a[key], key
    assert s["n_special"] * 4 < s["n_walk_ids"] and s["n_fetched"] <= 8          # the skip positions come in one strided copy per long record
    assert m["n_special"] < s["n_special"] and m["n_fetched"] > 100


@pytest.mark.parametrize("warm", ["1000000", "30000", "2000", "0"])
def test_walk_by_several_walkers_is_the_sequential_walk(built, tmp_path, monkeypatch, warm):
    # agx_walk.cpp: walk_split — further walkers start a warm-up stretch in front of their share of the unit, each on its own window of the visited
    # bytes; where the walker in front arrives the two states are compared, and the stretch either stands or is walked again.  The
    # oracle's bytes whatever the warm-up is worth (long enough, marginal, far too short, none).
    run = H.synth(str(tmp_path / "run"), seed=109, chroms="260000", pairs=52000, coverage=4, read_indel=0.2, multi=0.2, contig_overlap=0.4, contig_minus=0.5, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    o = H.run_oracle(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    for key, val in (("AGX_SIM_ASSISTANT", "1"), ("AGX_WALK_SPLIT_MIN", "0"), ("AGX_WALK_SPLIT_WARMUP", warm), ("AGX_WALK_POISON", "1")):      # (poison: the walkers' windows lie in memory full of junk, as in a process that has walked other units)
        monkeypatch.setenv(key, val)
    # look: the walkers see 20 kb of their stretch only: they give up.  More than four walkers share the three copies of the visited bytes, each behind
    # its own window of three stretches (short warm-ups only: the stretches must be longer than the warm-up for that)
    for walkers, look in (("2", None), ("3", None), ("4", None), ("4", "20000"), ("5", None), ("8", None), ("8", "20000")):
        monkeypatch.setenv("AGX_WALK_SPLIT_WALKERS", walkers)
        if look:
            monkeypatch.setenv("AGX_WALK_SPLIT_LOOK", look)
        else:
            monkeypatch.delenv("AGX_WALK_SPLIT_LOOK", raising=False)
        s = sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
        for key in ("initial", "pre", "extended"):
            assert o[key] == s[key], (walkers, look, key)


def test_shared_reads_index_loads_the_same_pairs(built, tmp_path, monkeypatch):
    # agx_reads (one map + record index of tmp/_reads.fa for all units of a run) against the per-unit scan of the file, including the
    # batch rule with a shrunk BATCH and reads files that end in an empty line
    run = H.synth(str(tmp_path / "run"), seed=108, chroms="9000,7000,5000", pairs=3000, coverage=3, multi=0.3, unaligned=0.2, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    for batch in (0, 700):
        plain = [sim.run(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], batch=batch or 1000000, graph=True) for u in range(meta["units"])]
        monkeypatch.setenv("AGX_SIM_READS_INDEX", "1")
        shared = [sim.run(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], batch=batch or 1000000, graph=True) for u in range(meta["units"])]
        monkeypatch.delenv("AGX_SIM_READS_INDEX")
        for a, b in zip(plain, shared):
            assert graph_mismatch(a["graph"], b["graph"]) is None
            for key in ("initial", "pre", "extended"):
                assert a[key] == b[key], key
        for u, s in enumerate(plain):
            o = H.run_oracle(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], batch=batch or 1000000)
            assert o["extended"] == s["extended"]
    with open(os.path.join(tmp, "_reads.fa"), "a") as f:          # an empty line ends the file for both loaders; records behind it do not exist
        f.write("\n>9999999\nACGT\n")
    for shared_loader in (False, True):
        if shared_loader:
            monkeypatch.setenv("AGX_SIM_READS_INDEX", "1")
        again = sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"], batch=700)
        assert again["extended"] == plain[0]["extended"] and again["pre"] == plain[0]["pre"]


@pytest.mark.parametrize("threads", [2, 3, 7])
def test_multi_thread_sam_parsing_is_the_one_thread_loader(built, tmp_path, monkeypatch, threads):
    # large SAM files are cut into line-pair-aligned byte ranges and parsed on several threads (agx_host.cpp); forced here on small files,
    # with multi-hit pairs, unaligned pairs, CIGARs with several runs and a shrunk BATCH, against the one-thread path
    run = H.synth(str(tmp_path / "run"), seed=109, chroms="15000,9000", pairs=5000, coverage=3, multi=0.3, unaligned=0.2, read_indel=0.4, read_clip=0.2, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    for u in range(meta["units"]):
        for batch in (1000000, 900):
            monkeypatch.setenv("AGX_LOAD_THREADS", "1")
            one = sim.run(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], batch=batch, graph=True)
            monkeypatch.setenv("AGX_LOAD_THREADS", str(threads))
            many = sim.run(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], batch=batch, graph=True)
            assert graph_mismatch(one["graph"], many["graph"]) is None
            for key in ("initial", "pre", "extended"):
                assert one[key] == many[key], key
    # an odd number of lines is reported when its place in the file is reached, whichever thread found it
    sam = os.path.join(tmp, "_reads_genome.0.bowtie")
    lines = open(sam).read().split("\n")
    open(sam, "w").write("\n".join(lines[:len(lines) // 2 | 1]) + "\n")
    for t in ("1", str(threads)):
        monkeypatch.setenv("AGX_LOAD_THREADS", t)
        with pytest.raises(sim.SimError) as e:
            sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
        assert "BROKEN BOWTIE FILE" in e.value.msg
    # the reads file is indexed and copied on the same threads: reads that are not as long as their CIGAR says are the same error either way
    reads = os.path.join(tmp, "_reads.fa")
    rl = open(reads).read().split("\n")
    open(reads, "w").write("\n".join(x[:-1] if i > len(rl) // 2 and i % 2 else x for i, x in enumerate(rl)))
    msgs = []
    for t in ("1", str(threads)):
        monkeypatch.setenv("AGX_LOAD_THREADS", t)
        with pytest.raises(sim.SimError) as e:
            sim.run(tmp, 1, meta["k"], meta["insert_variation"], meta["coverage"])
        msgs.append(e.value.msg)
    assert msgs[0] == msgs[1] and "CIGAR length differs from the read length" in msgs[0]


def test_batch_boundary_drops_first_pair_of_next_batch(built, tmp_path):
    # AG:1258-1259 with BATCH shrunk to 500 pairs: oracle and engine loaders must lose the same line pairs
    run = H.synth(str(tmp_path / "run"), seed=7, chroms="8000", pairs=2300, coverage=3, multi=0.3, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    a = H.run_oracle(tmp, 0, 5, 50, 3, batch=500, graph=True)
    b = sim.run(tmp, 0, 5, 50, 3, batch=500, graph=True)
    c = H.run_oracle(tmp, 0, 5, 50, 3, batch=1000000, graph=True)
    assert graph_mismatch(a["graph"], b["graph"]) is None and a["pre"] == b["pre"] and a["extended"] == b["extended"]
    assert graph_mismatch(a["graph"], c["graph"]) is not None     # the dropped pairs are visible in the counts
    # a reads file of exactly m*BATCH pairs is followed by one empty batch
    d = H.run_oracle(tmp, 0, 5, 50, 3, batch=2300, graph=True)
    e = sim.run(tmp, 0, 5, 50, 3, batch=2300, graph=True)
    assert graph_mismatch(d["graph"], e["graph"]) is None and d["pre"] == e["pre"]
    f = H.run_oracle(tmp, 0, 5, 50, 3, batch=1150, graph=True)
    g = sim.run(tmp, 0, 5, 50, 3, batch=1150, graph=True)
    assert graph_mismatch(f["graph"], g["graph"]) is None and f["pre"] == g["pre"]


def test_empty_and_ragged_inputs(built, tmp_path):
    run = H.synth(str(tmp_path / "run"), seed=9, chroms="6000", pairs=400, coverage=2, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    # no contig alignments at all
    open(os.path.join(tmp, "_contigs_genome.0.psl"), "w").close()
    a = H.run_oracle(tmp, 0, 5, 50, 2, graph=True); b = sim.run(tmp, 0, 5, 50, 2, graph=True)
    assert graph_mismatch(a["graph"], b["graph"]) is None and a["pre"] == b["pre"] and a["extended"] == b["extended"] and a["initial"] == b["initial"]
    # no read alignments at all: every position stays empty, outputs are empty files
    open(os.path.join(tmp, "_reads_genome.0.bowtie"), "w").close()
    a = H.run_oracle(tmp, 0, 5, 50, 2, graph=True); b = sim.run(tmp, 0, 5, 50, 2, graph=True)
    assert a["graph"]["n_nodes"] == 0 and b["graph"]["n_nodes"] == 0 and a["pre"] == b["pre"] == b"" and a["extended"] == b["extended"] == b""


def test_loader_rejects_what_the_reference_would_misread(built, tmp_path):
    run = H.synth(str(tmp_path / "run"), seed=5, chroms="5000", pairs=200, coverage=2, sam_seq=0, multi=0, read_indel=0, read_clip=0, read_badclip=0)
    tmp = os.path.join(run, "tmp")
    sam = os.path.join(tmp, "_reads_genome.0.bowtie")
    lines = open(sam).read().split("\n")
    open(sam, "w").write("\n".join(lines[2:4] + lines[0:2] + lines[4:]))          # not sorted by read id
    with pytest.raises(sim.SimError, match="not sorted"):
        sim.run(tmp, 0, 5, 50, 2)
    f = lines[0].split("\t"); f[5] = "90M"
    open(sam, "w").write("\n".join(["\t".join(f)] + lines[1:]))                  # CIGAR shorter than the read
    with pytest.raises(sim.SimError):
        sim.run(tmp, 0, 5, 50, 2)
    open(sam, "w").write(lines[0] + "\n")
    with pytest.raises(sim.SimError, match="BROKEN BOWTIE FILE"):
        sim.run(tmp, 0, 5, 50, 2)
    os.remove(sam)
    with pytest.raises(sim.SimError, match="CANNOT OPEN FILE"):
        sim.run(tmp, 0, 5, 50, 2)


@pytest.mark.parametrize("n_pairs,ok", [(100, True), (180, True), (300, True), (1100, False)])
def test_positions_with_more_variants_than_the_sweeps_first_buckets(built, tmp_path, n_pairs, ok):
    """The reference keeps an unbounded vector<KMer> per position (AG:1375-1390).  The engine sweeps a tile with 2, then 4, then 64, then 1024
    variants per position (node_cnt is 16 bits wide since r04; 300 used to be refused); beyond 1024 it refuses with AGX_E_OVERFLOW instead of diverging.  Here: the serial executor."""
    from conftest import write_pileup_unit
    tmp = write_pileup_unit(str(tmp_path / "run"), n_pairs, spacing=300 if n_pairs <= 180 else 190 if n_pairs <= 300 else 130, genome_len=60000 if n_pairs <= 300 else 150000)
    o = H.run_oracle(tmp, 0, 5, 50, 1, graph=True)
    import numpy as np
    assert int(np.diff(o["graph"]["node_start"]).max()) == n_pairs
    if not ok:
        with pytest.raises(sim.SimError, match="more than 1024 node variants"):
            sim.run(tmp, 0, k=5, insert_variation=50, coverage=1)
        return
    s = sim.run(tmp, 0, k=5, insert_variation=50, coverage=1, graph=True)
    assert graph_mismatch(o["graph"], s["graph"]) is None
    for key in ("initial", "pre", "extended"):
        assert o[key] == s[key], key


def test_walkers_behind_windows_of_the_visited_bytes(built, tmp_path, monkeypatch, capfd):
    # walk_split with five to sixteen walkers: walker i copies the window [c(i-1), c(i+2)) of the visited bytes for itself and may look nowhere else.  The
    # oracle's bytes, with stretches that stand (so the merged windows are what the appended positions are walked on) and with walkers that give up at the
    # edge of a narrow window.
    run = H.synth(str(tmp_path / "run"), seed=131, chroms="1000000", pairs=200000, coverage=4, read_indel=0.2, multi=0.2, contig_overlap=0.4, contig_minus=0.5, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    o = H.run_oracle(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    for key, val in (("AGX_SIM_ASSISTANT", "1"), ("AGX_WALK_SPLIT_MIN", "0"), ("AGX_WALK_TIMING", "1"), ("AGX_WALK_POISON", "1")):
        monkeypatch.setenv(key, val)
    stood = {}
    # (even cuts: whether a stretch stands behind a warm-up this short depends on where it begins; r06 cuts the stretches by when the walkers can begin — the first one's is
    # longer — which is tried at the end for the bytes alone)
    monkeypatch.setenv("AGX_WALK_EVEN_CUTS", "1")
    for walkers, warm, look in (("5", "40000", None), ("6", "40000", None), ("7", "40000", None), ("8", "40000", None), ("8", "5000", None), ("8", "40000", "60000"),
                                ("12", "20000", None), ("16", "20000", None), ("8", "40000", "model"), ("16", "20000", "model")):
        model = look == "model"
        if model:
            monkeypatch.delenv("AGX_WALK_EVEN_CUTS", raising=False)
            look = None
        monkeypatch.setenv("AGX_WALK_SPLIT_WALKERS", walkers)
        monkeypatch.setenv("AGX_WALK_SPLIT_WARMUP", warm)
        if look:
            monkeypatch.setenv("AGX_WALK_SPLIT_LOOK", look)
        else:
            monkeypatch.delenv("AGX_WALK_SPLIT_LOOK", raising=False)
        capfd.readouterr()
        s = sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
        err = capfd.readouterr().err
        for key in ("initial", "pre", "extended"):
            assert o[key] == s[key], (walkers, warm, look, key)
        line = [ln for ln in err.splitlines() if "stretches stood" in ln]
        assert line and ("%s walkers" % walkers) in line[0], err[-2000:]
        stood[(walkers, warm, look, model)] = int(line[0].split(" walkers, ")[1].split()[0])
    assert max(stood.values()) >= 7, stood                      # most stretches of many walkers stood somewhere


@pytest.mark.parametrize("stream", ["1", "4,300", "16,50"])
def test_walk_on_a_graph_that_is_still_arriving(built, tmp_path, monkeypatch, capfd, stream):
    # r06, the streamed download (agx_engine.cpp: begin_streamed_download; GraphView::wait_landed): the walk begins while the walk graph is still coming in, a position window at
    # a time from the front, the bases last.  The serial executor delivers it the same way — the arrays the walk is given are full of junk and a thread copies the real ones in
    # (pieces, microseconds per piece) — so a walker that looks at a byte it has not waited for reads junk and the outputs differ.  The further walkers wait for their windows and
    # their stretches are cut by when those land; the first walker waits for everything; one walker alone (no helper threads) waits for everything too.
    run = H.synth(str(tmp_path / "run"), seed=137, chroms="900000", pairs=180000, coverage=4, read_indel=0.2, multi=0.2, contig_overlap=0.4, contig_minus=0.5, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    o = H.run_oracle(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    monkeypatch.setenv("AGX_SIM_STREAM", stream)
    s1 = sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])            # one walker, nobody to format beside it
    for key in ("initial", "pre", "extended"):
        assert o[key] == s1[key], ("one walker", key)
    for key, val in (("AGX_SIM_ASSISTANT", "1"), ("AGX_WALK_SPLIT_MIN", "0"), ("AGX_WALK_TIMING", "1"), ("AGX_WALK_POISON", "1"), ("AGX_WALK_SPLIT_WARMUP", "30000")):
        monkeypatch.setenv(key, val)
    stood = []
    for walkers, look in (("2", None), ("5", None), ("8", None), ("8", "40000"), ("16", None)):
        monkeypatch.setenv("AGX_WALK_SPLIT_WALKERS", walkers)
        if look:
            monkeypatch.setenv("AGX_WALK_SPLIT_LOOK", look)
        else:
            monkeypatch.delenv("AGX_WALK_SPLIT_LOOK", raising=False)
        capfd.readouterr()
        s = sim.run(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
        err = capfd.readouterr().err
        for key in ("initial", "pre", "extended"):
            assert o[key] == s[key], (walkers, look, key)
        line = [ln for ln in err.splitlines() if "stretches stood" in ln]
        assert line and "streamed download" in err, err[-2000:]
        stood.append(int(line[0].split(" walkers, ")[1].split()[0]))
    assert max(stood) >= 1, stood               # (some stretches stood behind their windows: what stands depends on the warm-up, as in the tests above)
