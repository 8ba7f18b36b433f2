"""Parity tests proper (-m gpu): the HIP engine through the C-ABI against the oracle and the reference's golden vectors."""
import os
import shutil

import pytest

import harness as H
from conftest import graph_mismatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def agx():
    import aligngraph_amd as A
    if not os.path.exists(A.LIB_PATH):
        from aligngraph_amd import build as B
        B.build()
    assert A.device_count() > 0, "no HIP device: the gpu tests must run on the MI355X box"
    return A


def run_engine(agx, tmp, unit, k, iv, cov, batch=0, graph=False, flags=0):
    with agx.Unit(k=k, insert_variation=iv, coverage=cov, batch=batch, keep_counts=graph, flags=flags) as u:
        u.load_files(tmp, unit)
        u.upload()
        u.build()
        out = u.finish()
        out["stats"] = u.stats()
        if graph:
            out["graph"] = u.graph()
    return out


def test_golden_vectors(agx, golden, built):
    p = golden.params
    for cov in p["coverages"]:
        for u in range(p["units"]):
            got = run_engine(agx, golden.tmp, u, p["k"], p["insert_variation"], cov)
            exp = golden.expected(cov, u)
            for key in ("initial", "pre", "extended"):
                assert got[key] == exp[key], "%s cov=%d unit=%d %s" % (golden.name, cov, u, key)


CONFIGS = [
    dict(seed=201, chroms="60000", pairs=20000, coverage=5, contig_min=1500, contig_max=3000),
    dict(seed=202, chroms="40000", pairs=12000, coverage=3, L=50, k=8, read_indel=0.3, read_clip=0.3, indel=0.005, multi=0.3),
    dict(seed=203, chroms="30000", pairs=9000, coverage=5, frag_sd=300, insert_variation=10),
    dict(seed=204, chroms="30000,20000", part=2, pairs=10000, coverage=4, L=150, k=21, contig_min=300, contig_max=2000, contig_overlap=0.5,
         contig_dup=0.3, contig_split=0.5, contig_minus=1.0),
    dict(seed=205, chroms="300000", pairs=60000, coverage=5),
    dict(seed=207, chroms="50000", pairs=4000, coverage=3, L=250, k=11, read_indel=0.3),      # spans of five tiles: the binning's second round of atomics
    # found by tests/tools/fuzz_parity.py: narrow windows (insert_variation 0) on short deep units -> most tiles outgrow LDS and the node pool's
    # first guess (2 x positions) is too small, so the first build gives up half-way and is repeated with a larger pool
    dict(seed=860528, chroms="5551,21059", part=2, pairs=9239, L=36, k=3, coverage=3, insert_variation=0, snp=0.02, indel=0.001, contig_min=2000,
         contig_max=3000, contig_minus=0.54, contig_split=0.23, contig_dup=0.09, contig_overlap=0.09, contig_lowid=0.12, read_indel=0.5, read_clip=0.05,
         multi=0.1),
    # found by tests/tools/fuzz_parity.py --engine gpu (seed 606, iteration 10) on r06's lean tile records: reads of 36 bases with k = 31 and an indel in every second read — many list
    # entries that fit no lean record (kind GENERAL: pass 0 decodes the hit's own derived record) on reverse-strand reads; the first form lost the strand of such a record and
    # counted its votes from the wrong end of the read: node keys and edges equal, vote counters different
    dict(seed=449022, chroms="34734,33365,11871", part=1, pairs=66641, L=36, k=31, coverage=5, insert_variation=10, snp=0.02, indel=0.01, contig_min=250, contig_max=20000,
         contig_minus=0.09160248584068127, contig_split=0.46331666789922493, contig_dup=0.30874604567655217, contig_overlap=0.18161683683418484, contig_lowid=0.07560065568958672,
         read_err=0, read_indel=0.5, read_clip=0.05),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "seed%d" % c["seed"])
def test_node_edge_tables_and_outputs_match_oracle(agx, cfg, built, tmp_path):
    run = H.synth(str(tmp_path / "run"), sam_seq=0, **cfg)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    for u in range(meta["units"]):
        o = H.run_oracle(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
        g = run_engine(agx, tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
        assert graph_mismatch(o["graph"], g["graph"]) is None
        for key in ("initial", "pre", "extended"):
            assert o[key] == g[key], key
        if cfg.get("frag_sd") == 300:
            assert g["stats"]["n_big_tiles"] > 0 and g["stats"]["n_mid_tiles"] > g["stats"]["n_big_tiles"]      # all three sweep passes had tiles


def test_sparse_record_table_and_device_fetch(agx, built, tmp_path):
    # see tests/test_hostsim.py::test_sparse_record_table_and_fetch_hook; here the records come from the kernels and the fetch hook
    # reads the full table that stays in HBM
    run = H.synth(str(tmp_path / "run"), seed=105, chroms="300000", pairs=60000, coverage=5, contig_min=120000, contig_max=200000, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    o = H.run_oracle(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    g = run_engine(agx, tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"])
    m = run_engine(agx, tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"], flags=agx.AGX_FLAG_SPARSE_MIN)
    for key in ("initial", "pre", "extended"):
        assert o[key] == g[key] == m[key], key
    gs, ms = g["stats"], m["stats"]
    assert gs["n_special"] * 4 < gs["n_walk_ids"] and gs["n_fetched"] <= 8       # the skip positions come in one strided copy per long record
    assert ms["n_special"] < gs["n_special"] and ms["n_fetched"] > 100
    assert gs["download_bytes"] < 8 * gs["n_walk_ids"]                 # was 40 bytes per id with the dense record table


@pytest.mark.parametrize("general_loader", [False, True])
def test_hop_entries_outside_the_sparse_table_come_from_the_runs(agx, built, tmp_path, monkeypatch, general_loader):
    """a unit without a per-position hop table (every unit since r03: the fast loader keeps none, nor does the cache file) must still find the hop
    entry of a main id that is not in the sparse table — AGX_FLAG_SPARSE_MIN puts every main id there, long contigs make the walk hop on and off them.  The
    walk takes it from the conti-mer runs by bisection (Walker::hop_of), as the device does for the special ids.  From the text (both loaders) and from the cache file."""
    if general_loader:
        monkeypatch.setenv("AGX_NO_FAST_LOAD", "1")
    run = H.synth(str(tmp_path / "run"), seed=106, chroms="300000", pairs=50000, coverage=5, contig_min=60000, contig_max=150000, contig_overlap=0.3, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    want = H.run_oracle(tmp, 0, 5, 50, 5)
    for cached in (False, True):
        if cached:
            agx.cache_build(tmp, 0)
        with agx.Unit(k=5, insert_variation=50, coverage=5, flags=agx.AGX_FLAG_SPARSE_MIN) as u:
            u.load_files(tmp, 0)
            assert u.stats()["from_cache"] == (1 if cached else 0)
            u.upload(); u.build()
            got = u.finish()
            assert u.stats()["n_fetched"] > 100
        for key in ("initial", "pre", "extended"):
            assert got[key] == want[key], (key, cached)


def test_every_capacity_regrows_from_the_device_counters(agx, built, tmp_path, monkeypatch):
    # A build queues all its kernels against first-guess capacities (tile lists, node pool, edge overflow list, sparse record table) and
    # only then reads the counters; AGX_TEST_SMALL_CAPS starts every one of them far too small, so the build has to be repeated once per
    # capacity — and the kernels behind a sweep that gave up must not touch the half-written node table.
    run = H.synth(str(tmp_path / "run"), seed=208, chroms="60000", pairs=20000, coverage=4, insert_variation=10, frag_sd=150, contig_overlap=0.4, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    o = H.run_oracle(tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
    monkeypatch.setenv("AGX_TEST_SMALL_CAPS", "1")
    g = run_engine(agx, tmp, 0, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
    assert graph_mismatch(o["graph"], g["graph"]) is None
    for key in ("initial", "pre", "extended"):
        assert o[key] == g[key], key
    assert g["stats"]["n_edge_overflow"] > 4            # the overflow list really outgrew its first guess
    assert g["stats"]["build_attempts"] >= 4            # tile lists, node pool, overflow list, sparse record table


def test_run_unit_writes_the_three_files(agx, built, tmp_path):
    run = H.synth(str(tmp_path / "run"), seed=31, chroms="20000", pairs=5000, coverage=5, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    o = H.run_oracle(tmp, 0, 5, 50, 5)
    got = agx.run_unit(tmp, 0, k=5, insert_variation=50, coverage=5, write_files=True)
    for key, fn in (("initial", "_initial_contigs.0.fa"), ("pre", "_pre_extended_contigs.0.fa"), ("extended", "_extended_contigs.0.fa")):
        assert got[key] == o[key]
        assert open(os.path.join(tmp, fn), "rb").read() == o[key]


def test_rebuild_is_idempotent_and_batch_rule(agx, built, tmp_path):
    run = H.synth(str(tmp_path / "run"), seed=7, chroms="8000", pairs=2300, coverage=3, multi=0.3, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    o = H.run_oracle(tmp, 0, 5, 50, 3, batch=500)
    with agx.Unit(k=5, insert_variation=50, coverage=3, batch=500) as u:
        u.load_files(tmp, 0); u.upload()
        for _ in range(3):
            u.build()
            got = u.finish()
            assert got["pre"] == o["pre"] and got["extended"] == o["extended"]


def test_units_built_from_several_threads_share_the_device(agx, built, tmp_path):
    # the way bench.py and AlignGraph_amd drive a device: every unit has a host thread that keeps rebuilding / walking it.  libagx queues the
    # builds of a device on two shared streams (the front of one build beside the back of the previous one, sweeps one after the other);
    # whatever the interleaving, every build must give its own unit's bytes.  One of the units times its sections (exclusive builds).
    import threading
    run = H.synth(str(tmp_path / "run"), seed=411, chroms="30000,22000,16000,9000", pairs=24000, coverage=3, multi=0.2, read_indel=0.2, sam_seq=0)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    want = [H.run_oracle(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"]) for u in range(meta["units"])]
    errors = []

    def worker(i):
        try:
            with agx.Unit(k=meta["k"], insert_variation=meta["insert_variation"], coverage=meta["coverage"],
                          flags=agx.AGX_FLAG_TIME_SECTIONS if i == 1 else 0) as u:
                unit = i % meta["units"]
                u.load_files(tmp, unit); u.upload()
                for _ in range(12):
                    u.build(); u.download()
                    got = u.finish()
                    for key in ("initial", "pre", "extended"):
                        assert got[key] == want[unit][key], (i, key)
        except BaseException as e:
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2 * meta["units"])]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


def test_one_launch_scan_against_a_host_scan(agx):
    # the decoupled look-back scan of the build (tile histogram, side ids, special ids): one block, block edges, more than one look-back
    # window of 64 predecessors (a unit needs 16 M positions for that), a few hundred windows
    for n, seed in ((0, 1), (1, 2), (4094, 3), (4095, 4), (4096, 5), (4097, 6), (72000, 7), (300000, 8), (1200000, 9), (20000000, 10), (20000001, 11)):
        assert agx.lib().agx_selftest_scan(0, n, seed) == 0, n


def test_errors_come_back_as_codes(agx, built, tmp_path):
    run = H.synth(str(tmp_path / "run"), seed=5, chroms="5000", pairs=200, coverage=2, sam_seq=0, multi=0, read_indel=0, read_clip=0, read_badclip=0)
    tmp = os.path.join(run, "tmp")
    sam = os.path.join(tmp, "_reads_genome.0.bowtie")
    lines = open(sam).read().split("\n")
    f = lines[0].split("\t"); g = lines[1].split("\t")
    g[1] = str((int(g[1]) & ~0x10) | (int(f[1]) & 0x10))
    open(sam, "w").write("\n".join([lines[0], "\t".join(g)] + lines[2:]))
    with pytest.raises(agx.AgxError) as e:
        run_engine(agx, tmp, 0, 5, 50, 2)
    assert e.value.code == agx.AGX_E_ALIGNMENT and "BOWTIE ALIGNMENT ERROR" in e.value.msg
    os.remove(sam)
    with pytest.raises(agx.AgxError) as e:
        run_engine(agx, tmp, 0, 5, 50, 2)
    assert e.value.code == agx.AGX_E_IO


def test_packed_array_boundary_matches_text_path(agx, built, tmp_path):
    """agx_unit_set_reference / _set_contig_threads / _push_pairs (the packed boundary of SURVEY §8b) fed by hand must give the bytes the
    text loaders + oracle give for the equivalent files: gap-free FR pairs on a contig-free unit, two batches."""
    import ctypes
    import random
    rnd = random.Random(5)
    G, L, N = 3000, 60, 400
    ref = "".join(rnd.choice("ACGT") for _ in range(G))
    comp = str.maketrans("ACGT", "TGCA")
    pairs = []
    for i in range(N):
        s = rnd.randrange(0, G - 400)
        f = rnd.randrange(200, 380)
        left, right = ref[s:s + L], ref[s + f - L:s + f]
        m1_left = rnd.random() < 0.5
        pairs.append((s, s + f - L, left, right.translate(comp)[::-1], m1_left))
    # the same data as the reference's tmp/ files, for the oracle
    tmp = tmp_path / "tmp"
    tmp.mkdir()
    (tmp / "_genome.0.fa").write_text(">0\n" + "\n".join(ref[i:i + 60] for i in range(0, G, 60)) + "\n")
    (tmp / "_contigs.fa").write_text("")
    (tmp / "_contigs_genome.0.psl").write_text("")
    with open(tmp / "_reads.fa", "w") as rf, open(tmp / "_reads_genome.0.bowtie", "w") as sf:
        for i, (pl, pr, fl, fr_, m1_left) in enumerate(pairs):
            m1, m2 = (fl, fr_) if m1_left else (fr_, fl)
            rf.write(">%d\n%s\n>%d\n%s\n" % (i, m1, i, m2))
            p1, p2 = (pl, pr) if m1_left else (pr, pl)
            f1, f2 = (99, 147) if m1_left else (83, 163)
            sf.write("%d\t%d\t0\t%d\t42\t%dM\t=\t%d\t0\t*\t*\n%d\t%d\t0\t%d\t42\t%dM\t=\t%d\t0\t*\t*\n" % (i, f1, p1 + 1, L, p2 + 1, i, f2, p2 + 1, L, p1 + 1))
    want = H.run_oracle(str(tmp), 0, 5, 50, 2)
    # the packed boundary, two batches
    with agx.Unit(k=5, insert_variation=50, coverage=2) as u:
        L_ = agx.lib()
        u._check(L_.agx_unit_set_reference(u._h, ref.encode(), G))
        cm_start = (ctypes.c_uint32 * (G + 1))()
        u._check(L_.agx_unit_set_contig_threads(u._h, None, 0, cm_start, None, 0, b"", 0))
        stride = 64
        for lo, hi in ((0, 150), (150, N)):
            n = hi - lo
            hits = (agx.Hit * n)()
            bases = bytearray(b"N" * (2 * n * stride))
            for j, (pl, pr, fl, fr_, m1_left) in enumerate(pairs[lo:hi]):
                m1, m2 = (fl, fr_) if m1_left else (fr_, fl)
                bases[(2 * j) * stride:(2 * j) * stride + L] = m1.encode()
                bases[(2 * j + 1) * stride:(2 * j + 1) * stride + L] = m2.encode()
                h = hits[j]
                h.slot1 = 2 * j
                h.pos1, h.pos2 = (pl, pr) if m1_left else (pr, pl)
                h.len = L
                h.rev1, h.rev2 = (0, 1) if m1_left else (1, 0)
            b = agx.PairBatch(hits, n, None, 0, bytes(bases), stride, 2 * n)
            u._check(L_.agx_unit_push_pairs(u._h, ctypes.byref(b)))
        u.upload(); u.build()
        got = u.finish()
    assert got["pre"] == want["pre"] and got["extended"] == want["extended"] and got["initial"] == want["initial"] == b""
    assert want["pre"].count(b">") > 0


def test_empty_and_ragged_inputs(agx, built, tmp_path):
    run = H.synth(str(tmp_path / "run"), seed=9, chroms="6000", pairs=400, coverage=2, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    open(os.path.join(tmp, "_contigs_genome.0.psl"), "w").close()               # no contig alignments
    a = H.run_oracle(tmp, 0, 5, 50, 2, graph=True)
    b = run_engine(agx, tmp, 0, 5, 50, 2, graph=True)
    assert graph_mismatch(a["graph"], b["graph"]) is None and all(a[k] == b[k] for k in ("initial", "pre", "extended"))
    open(os.path.join(tmp, "_reads_genome.0.bowtie"), "w").close()              # no read alignments: empty graph, empty outputs
    b = run_engine(agx, tmp, 0, 5, 50, 2, graph=True)
    assert b["graph"]["n_nodes"] == 0 and b["pre"] == b"" and b["extended"] == b""
    # k as long as the reads: every hit is skipped (the reference's loop bound underflows there; here nothing is emitted)
    run2 = H.synth(str(tmp_path / "run2"), seed=10, chroms="6000", pairs=300, L=40, coverage=2, sam_seq=0)
    b = run_engine(agx, os.path.join(run2, "tmp"), 0, 40, 50, 2, graph=True)
    assert b["graph"]["n_nodes"] == 0


def test_walk_by_several_walkers_gives_the_same_bytes(agx, built, tmp_path, monkeypatch):
    """Large units are walked by two to sixteen walkers (agx_walk.cpp: walk_split; more than four share three copies of the visited bytes): forced here on
    a small unit, with warm-up stretches that let the other walkers' stretches stand and with some so short that the first walker has to walk on — the
    oracle's bytes either way."""
    run = H.synth(str(tmp_path / "run"), seed=77, chroms="400000", pairs=80000, coverage=4, read_indel=0.2, multi=0.2, contig_overlap=0.3, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    want = H.run_oracle(tmp, 0, 5, 50, 4)
    monkeypatch.setenv("AGX_WALK_SPLIT_MIN", "0")
    monkeypatch.setenv("AGX_WALK_POISON", "1")          # the walkers' windows lie in junk (agx_walk.cpp): what a window does not hold must never be looked at
    for walkers, warm in (("2", "400000"), ("2", "50000"), ("2", "20"), ("3", "50000"), ("4", "400000"), ("4", "30000"), ("6", "20000"), ("8", "20000"), ("8", "2000"), ("16", "10000")):
        monkeypatch.setenv("AGX_WALK_SPLIT_WALKERS", walkers)
        monkeypatch.setenv("AGX_WALK_SPLIT_WARMUP", warm)
        for flags in (0, agx.AGX_FLAG_ONE_SHOT):
            got = run_engine(agx, tmp, 0, 5, 50, 4, flags=flags)
            for key in ("initial", "pre", "extended"):
                assert got[key] == want[key], (walkers, warm, flags, key)


@pytest.mark.parametrize("L", [100, 57])
def test_tile_ordered_upload_and_a_sweep_by_windows(agx, built, tmp_path, monkeypatch, L):
    """r06 (agx_engine.cpp: stage_tiled): the wire records and the read rows cross in the order of the hits' first tiles — the record carries the hit's number instead of a row, the
    rule of AG:1650-1655 is decided where the arrays are staged, every hit's left-mate row travels at the hit's place — and a unit's first build expands the rows and sweeps the tiles
    window by window as the pieces of the upload land (forced here on a small unit: AGX_UPLOAD_WINDOWS).  Reads with many second hits (rows sent twice, later hits dropped), many
    bases that are not A, C, G, T (the re-indexed list of them, patched window by window) and a read length whose rows are not a multiple of 16 bytes; the node and edge tables and
    the three outputs against the oracle for every number of windows, for r05's forms (AGX_NO_TILED_UPLOAD) and with every capacity regrown (the repeated builds sweep in one piece)."""
    run = H.synth(str(tmp_path / "run"), seed=83, chroms="300000", pairs=70000, L=L, coverage=4, read_indel=0.2, multi=0.3, multi_near=0.5, read_n=0.01, contig_overlap=0.3, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    want = H.run_oracle(tmp, 0, 5, 50, 4, graph=True)
    for windows, flags, env in (("1", 0, {}), ("2", 0, {}), ("3", agx.AGX_FLAG_ONE_SHOT, {}), ("8", 0, {}), ("5", agx.AGX_FLAG_ONE_SHOT, {"AGX_NO_CACHE": "1"}), ("4", 0, {"AGX_TEST_SMALL_CAPS": "1"}),
                                ("3", 0, {"AGX_NO_TILED_UPLOAD": "1"}), ("1", 0, {"AGX_ROW_DIFF": "1"}), ("3", 0, {"AGX_ROW_DIFF": "1"}), ("7", agx.AGX_FLAG_ONE_SHOT, {"AGX_ROW_DIFF": "1"}),
                                ("2", 0, {"AGX_ROW_DIFF": "1", "AGX_NO_TILED_UPLOAD": "1"})):      # (rows as differences: in tile-ordered form — a window's piece of the stream — and in r05's)
        monkeypatch.setenv("AGX_UPLOAD_WINDOWS", windows)
        for k2, v2 in env.items():
            monkeypatch.setenv(k2, v2)
        got = run_engine(agx, tmp, 0, 5, 50, 4, graph=(flags == 0), flags=flags)
        for k2 in env:
            monkeypatch.delenv(k2)
        for key in ("initial", "pre", "extended"):
            assert got[key] == want[key], (windows, flags, env, key)
        if flags == 0:
            assert graph_mismatch(want["graph"], got["graph"]) is None, (windows, env)
        assert got["stats"]["ms_node_sweep"] > 0
        assert (got["stats"]["rows_by_reference"] > 0.4 * got["stats"]["n_hits"]) == ("AGX_ROW_DIFF" in env), (windows, env, got["stats"]["rows_by_reference"])


def test_windows_without_rows(agx, built, tmp_path, monkeypatch):
    """Found by the randomised sweep (tests/tools/fuzz_parity.py --engine gpu, seed 606, iteration 42): a unit with so few hits that an early window's piece of the upload already
    holds the LAST read row (the windows behind it have no rows of their own).  The last row's piece is the one that is padded to whole 16-base groups; r06's first form padded the
    last WINDOW's piece instead and left the last row's last bases unexpanded — node keys differed.  Every unit of that configuration, eight windows, against node and edge tables."""
    cfg = {'seed': 92938, 'chroms': '25305,7448,22345', 'part': 2, 'pairs': 440, 'L': 250, 'k': 7, 'coverage': 1, 'insert_variation': 50, 'snp': 0, 'indel': 0.001, 'contig_min': 250, 'contig_max': 3000,
           'contig_minus': 0.21803743656655059, 'contig_split': 0.15109879794405667, 'contig_dup': 0.008181960270310018, 'contig_overlap': 0.5785686444445618, 'contig_lowid': 0.09784268627746147, 'read_err': 0,
           'read_indel': 0, 'read_clip': 0.05, 'read_n': 0.01, 'multi': 0.5, 'multi_near': 0.3, 'unaligned': 0.1, 'frag_mean': 300, 'frag_sd': 10, 'sam_seq': 0}
    run = H.synth(str(tmp_path / "run"), **cfg)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    for windows in ("8", "3"):
        monkeypatch.setenv("AGX_UPLOAD_WINDOWS", windows)
        for u in range(meta["units"]):
            o = H.run_oracle(tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
            for rowdiff in (False, True):
                if rowdiff:
                    monkeypatch.setenv("AGX_ROW_DIFF", "1")
                g = run_engine(agx, tmp, u, meta["k"], meta["insert_variation"], meta["coverage"], graph=True)
                monkeypatch.delenv("AGX_ROW_DIFF", raising=False)
                assert graph_mismatch(o["graph"], g["graph"]) is None, (windows, u, rowdiff)
                for key in ("initial", "pre", "extended"):
                    assert o[key] == g[key], (windows, u, rowdiff, key)


def test_walk_begins_while_the_download_is_still_arriving(agx, built, tmp_path, monkeypatch):
    """r06, the streamed download (agx_engine.cpp: begin_streamed_download): agx_unit_finish on a unit that has not been downloaded sends the walk graph down in position windows
    from the front (forced here on a small unit: AGX_STREAM_PIECES), the walkers wait for their windows, the first one for all of them, the bases come last.  The walkers' own bytes
    are poisoned, the landing buffers of a one-shot unit are its dead staged inputs (junk until the copies arrive): a byte looked at before it landed shows in the outputs.
    agx_unit_download + agx_unit_finish (the whole download first: what a caller does who trims the unit's HBM in between) must give the same bytes, and so must a unit
    streamed twice."""
    run = H.synth(str(tmp_path / "run"), seed=79, chroms="500000", pairs=100000, coverage=4, read_indel=0.2, multi=0.2, contig_overlap=0.3, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    want = H.run_oracle(tmp, 0, 5, 50, 4)
    monkeypatch.setenv("AGX_WALK_SPLIT_MIN", "0")
    monkeypatch.setenv("AGX_WALK_POISON", "1")
    monkeypatch.setenv("AGX_WALK_SPLIT_WARMUP", "30000")
    for pieces, walkers in (("1", "2"), ("2", "4"), ("5", "3"), ("5", "8"), ("16", "8"), ("16", "16"), ("7", "1")):
        monkeypatch.setenv("AGX_STREAM_PIECES", pieces)
        if walkers == "1":
            monkeypatch.setenv("AGX_WALK_NO_SPLIT", "1")      # one walker on a streamed download: it waits for everything
        else:
            monkeypatch.setenv("AGX_WALK_SPLIT_WALKERS", walkers)
        for flags in (0, agx.AGX_FLAG_ONE_SHOT):
            got = run_engine(agx, tmp, 0, 5, 50, 4, flags=flags)
            for key in ("initial", "pre", "extended"):
                assert got[key] == want[key], (pieces, walkers, flags, key)
            assert got["stats"]["ms_download"] > 0
    monkeypatch.delenv("AGX_WALK_NO_SPLIT")
    monkeypatch.setenv("AGX_STREAM_PIECES", "6")
    monkeypatch.setenv("AGX_WALK_SPLIT_WALKERS", "6")
    with agx.Unit(k=5, insert_variation=50, coverage=4) as u:
        u.load_files(tmp, 0)
        u.upload(); u.build()
        a = u.finish()                    # streamed
        b = u.finish()                    # the walk consumed the first download: streamed again
        u.download()
        c = u.finish()                    # the whole download, then the walk
        u.download(); u.trim()            # (nothing to give back on a device without a region for a unit this small: the call must still leave the unit walkable)
        d = u.finish()
    for got in (a, b, c, d):
        for key in ("initial", "pre", "extended"):
            assert got[key] == want[key], key


def test_unit_cache_file_replaces_the_text(agx, built, tmp_path, monkeypatch):
    """tmp/_agx_unit.<u>.bin holds a unit's staged arrays; load_files takes it instead of the five text files as long as it
    is current, and falls back to the text when a source file changed, when the batch size differs or when the file is damaged."""
    run = H.synth(str(tmp_path / "run"), seed=41, chroms="50000,30000", pairs=16000, coverage=4, read_indel=0.2, multi=0.2, contig_overlap=0.3, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    want = [H.run_oracle(tmp, uu, 5, 50, 4) for uu in range(2)]

    def load_and_run(uu, batch=0):
        with agx.Unit(k=5, insert_variation=50, coverage=4, batch=batch) as u:
            u.load_files(tmp, uu)
            st0 = u.stats()
            u.upload(); u.build()
            got = u.finish()
            exp = want[uu] if batch == 0 else H.run_oracle(tmp, uu, 5, 50, 4, batch=batch)
            for key in ("initial", "pre", "extended"):
                assert got[key] == exp[key], key
            return st0
    assert load_and_run(0)["from_cache"] == 0
    agx.cache_build(tmp, 0)                                  # from the text, as AlignGraph_amd does after the aligners
    with agx.Unit(k=5, insert_variation=50, coverage=4) as u:   # ... or from a unit that has just parsed the text
        u.load_files(tmp, 1)
        u.cache_save(tmp, 1)
    for uu in range(2):
        st = load_and_run(uu)
        assert st["from_cache"] == 1 and st["ms_parse"] == 0 and st["n_hits"] > 0 and st["sam_line_pairs"] > 0
    assert load_and_run(0, batch=5000)["from_cache"] == 0    # another BATCH keeps other pairs (AG:1258-1259): not this cache
    with agx.Unit(k=7, insert_variation=50, coverage=4) as u:  # another k names other left mates: not this cache
        u.load_files(tmp, 0)
        assert u.stats()["from_cache"] == 0
    monkeypatch.setenv("AGX_NO_CACHE", "1")
    assert load_and_run(0)["from_cache"] == 0
    monkeypatch.delenv("AGX_NO_CACHE")
    sam = os.path.join(tmp, "_reads_genome.0.bowtie")
    os.utime(sam, ns=(os.stat(sam).st_atime_ns, os.stat(sam).st_mtime_ns + 10**9))      # a source file touched: stale
    assert load_and_run(0)["from_cache"] == 0 and load_and_run(1)["from_cache"] == 1
    path = os.path.join(tmp, "_agx_unit.1.bin")
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) // 2)                # damaged
    assert load_and_run(1)["from_cache"] == 0


@pytest.mark.parametrize("n_pairs,ok", [(100, True), (180, True), (300, True), (1100, False)])
def test_positions_beyond_64_variants_take_the_last_pass(agx, built, tmp_path, n_pairs, ok):
    """ADVICE r01 / VERDICT r01 item 8: a position with more than 64 node variants used to abort the unit (AGX_E_OVERFLOW) where the reference,
    whose vector<KMer> is unbounded (AG:1375-1390), carries on.  Now a build that meets one queues a fourth sweep pass with 1024 variants per
    position (r04: node_cnt is 16 bits wide; the 300-variant pile-up that r03 refused must equal the oracle) and repeats; only beyond 1024 the unit is refused — loudly, never with a different graph."""
    from conftest import write_pileup_unit
    tmp = write_pileup_unit(str(tmp_path / "run"), n_pairs, spacing=300 if n_pairs <= 180 else 190 if n_pairs <= 300 else 130, genome_len=60000 if n_pairs <= 300 else 150000)
    if not ok:
        with pytest.raises(agx.AgxError) as e:
            run_engine(agx, tmp, 0, 5, 50, 1)
        assert e.value.code == agx.AGX_E_OVERFLOW and "1024" in e.value.msg
        return
    o = H.run_oracle(tmp, 0, 5, 50, 1, graph=True)
    g = run_engine(agx, tmp, 0, 5, 50, 1, graph=True)
    assert graph_mismatch(o["graph"], g["graph"]) is None
    for key in ("initial", "pre", "extended"):
        assert o[key] == g[key], key
    assert g["stats"]["build_attempts"] == 2 and g["stats"]["n_big_tiles"] >= 1


def test_a_list_longer_than_the_packed_counters_hold(agx, built, tmp_path):
    """r06: pass 0 of the node sweep keeps a variant's six counters as 16-bit halves (agx_bucket::packed), so a tile whose list holds more than 65 535 entries must be left to
    pass 1, whose buckets are not packed.  70 000 pairs with ONE alignment: every position of the left mate counts 70 000 arrivals of the same variant; counters, node and
    edge tables and the three outputs against the oracle."""
    from conftest import write_pileup_unit
    tmp = write_pileup_unit(str(tmp_path / "run"), 70000, spacing=0, genome_len=60000)
    o = H.run_oracle(tmp, 0, 5, 50, 1, graph=True)
    g = run_engine(agx, tmp, 0, 5, 50, 1, graph=True)
    assert int(o["graph"]["node_cnt"].max()) > 65535
    assert graph_mismatch(o["graph"], g["graph"]) is None
    for key in ("initial", "pre", "extended"):
        assert o[key] == g[key], key
    assert g["stats"]["n_mid_tiles"] >= 1


@pytest.mark.parametrize("n_pairs,mode", [(400, 1), (6000, 2)])
def test_hits_that_span_more_tiles_than_the_window(agx, built, tmp_path, n_pairs, mode):
    """r05: a tile's list is a filter over a window of the hits' tile order (agx_k_tile_fill); hits that reach further than the window looks back (here: 60-base deletions in
    2x100 bp reads, four tiles) come through the list of long hits (mode 1), and a unit with more than 1024 of them makes all its lists by scatter (mode 2: agx_k_bin_fill +
    agx_k_tile_sort).  Node and edge tables and the three outputs must be the oracle's either way."""
    from conftest import write_long_deletion_unit
    tmp = write_long_deletion_unit(str(tmp_path / "run"), n_pairs)
    want = H.run_oracle(tmp, 0, 5, 50, 2, graph=True)
    got = run_engine(agx, tmp, 0, 5, 50, 2, graph=True)
    assert got["stats"]["dense_lists"] == mode, got["stats"]["dense_lists"]
    assert graph_mismatch(want["graph"], got["graph"]) is None, graph_mismatch(want["graph"], got["graph"])
    for key in ("initial", "pre", "extended"):
        assert got[key] == want[key], key


@pytest.mark.parametrize("masked", [False, True])
def test_reference_bytes_that_are_not_acgt_survive_the_packed_upload(agx, built, tmp_path, masked):
    """The unit sequence crosses PCIe as 2 bits per base + the stretches of other bytes (agx_core.h wire formats); the reference emits a position's
    own byte wherever a walk falls back on it (AG:1997-2001) and fills gaps with it (AG:2426-2435).  An N run, IUPAC codes and — masked — lower case
    over half the unit (too many stretches: the bytes then cross as they are) must come out as the oracle writes them."""
    run = H.synth(str(tmp_path / "run"), seed=31, chroms="40000", pairs=6000, coverage=3, contig_min=1500, contig_max=3000, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    path = os.path.join(tmp, "_genome.0.fa")
    head, body = open(path).read().split("\n", 1)
    seq = list(body.replace("\n", ""))
    seq[5000:5800] = "N" * 800
    for i in (100, 101, 9000, 20001, 39999):
        seq[i] = "RYKMSW"[i % 6]
    if masked:
        for i in range(10000, 30000):
            if i % 7 < 3:
                seq[i] = seq[i].lower()
    seq = "".join(seq)
    open(path, "w").write(head + "\n" + "".join(seq[i:i + 60] + "\n" for i in range(0, len(seq), 60)))
    want = H.run_oracle(tmp, 0, 5, 50, 3)
    got = run_engine(agx, tmp, 0, 5, 50, 3)
    for key in ("initial", "pre", "extended"):
        assert got[key] == want[key], key
    assert b"N" * 100 in got["extended"] or b"N" * 100 in got["pre"] or True      # (whether a walk crosses the N run depends on the reads; the bytes were compared above)


def test_read_alignments_handed_over_staged(agx, built, tmp_path):
    """tmp/_agx_pairs.<u>.bin (agx_host.h: pairsfile) stands in for tmp/_reads.fa + the unit's SAM text: the engine takes the staged arrays as they are, the walk
    reads its k-mer tails out of the 2-bit rows, and the unit cache written from such a unit carries them along.  Same stream with and without the text
    (tools/agx_synth.cpp --pairs-bin 2 / 1): the oracle runs on the text, the engine on a directory that has none.  (tests/test_staged_pairs.py: the file itself
    against the loaders, on the CPU.)"""
    kw = dict(seed=77, chroms="60000,35000", pairs=26000, coverage=3, read_indel=0.25, read_clip=0.1, multi=0.3, read_n=0.02, contig_overlap=0.3, sam_seq=0)
    text = H.synth(str(tmp_path / "text"), threads=2, pairs_bin=2, **kw)
    run = H.synth(str(tmp_path / "run"), threads=3, pairs_bin=1, lean=1, **kw)
    tmp = os.path.join(run, "tmp")
    assert not os.path.exists(os.path.join(tmp, "_reads.fa"))
    want = [H.run_oracle(os.path.join(text, "tmp"), uu, 5, 50, 3) for uu in range(2)]
    assert sum(w["pre"].count(b">") for w in want) > 20

    def load_and_run(uu, flags=0, **params):
        with agx.Unit(k=params.get("k", 5), insert_variation=50, coverage=3, flags=flags, batch=params.get("batch", 0)) as u:
            u.load_files(tmp, uu)
            st0 = u.stats()
            u.upload(); u.build()
            got = u.finish()
            for key in ("initial", "pre", "extended"):
                assert got[key] == want[uu][key], key
            return st0
    for uu in range(2):
        st = load_and_run(uu)
        assert st["from_cache"] == 0 and st["n_hits"] > 0 and st["sam_line_pairs"] > st["n_hits"] and st["pairs_in_file"] == 26000
        load_and_run(uu, flags=agx.AGX_FLAG_ONE_SHOT)              # the download lands in the staged arrays' pinned memory: the tails come from the mapped file
    for bad in (dict(k=7), dict(batch=5000)):                       # staged for another k / another BATCH: refused, never re-interpreted
        with pytest.raises(agx.AgxError) as e:
            load_and_run(0, **bad)
        assert e.value.code == agx.AGX_E_ARG
    agx.cache_build(tmp, 0)
    with agx.Unit(k=5, insert_variation=50, coverage=3) as u:
        u.load_files(tmp, 1)
        u.cache_save(tmp, 1)
    for uu in range(2):
        st = load_and_run(uu)
        assert st["from_cache"] == 1 and st["n_hits"] > 0
        load_and_run(uu, flags=agx.AGX_FLAG_ONE_SHOT)              # (tails out of the mapped cache file's 2-bit rows)
    for uu in range(2):                                             # a cache made from a staged file is current only while that file is as it was
        os.remove(os.path.join(tmp, "_agx_pairs.%d.bin" % uu))
        with agx.Unit(k=5, insert_variation=50, coverage=3) as u:
            with pytest.raises(agx.AgxError):
                u.load_files(tmp, uu)                                # (no current cache, no staged pairs, no text)


@pytest.mark.gpu
def test_read_rows_cross_as_differences_from_the_reference(agx, built, tmp_path, monkeypatch):
    """AGX_ROW_DIFF=1: the upload sends a unit's read rows as their differences from the reference under each row's first hit (agx_core.h "read rows relative to the
    reference"; agx_k_expand_rows turns them back into vote codes) instead of as 2-bit rows (agx_k_expand_codes, the default).  Both ways give the oracle's three files
    and the same node and edge tables — on reads with indels, clips, N bases, both strands, several hits per pair — and the first way sends fewer bytes.  A soft-masked
    unit sequence (the reference crosses as bytes, nothing to predict from) keeps the 2-bit rows by itself.  (tests/test_row_diffs.py: the codec on the CPU.)"""
    kw = dict(seed=91, chroms="70000,30000", pairs=30000, coverage=3, L=150, k=21, read_indel=0.25, read_clip=0.1, multi=0.3, read_n=0.02, indel=0.003, contig_overlap=0.3, sam_seq=0)
    run = H.synth(str(tmp_path / "run"), **kw)
    tmp = os.path.join(run, "tmp")

    def both_ways(tmp_dir, uu, expect_diffed=True, flags=0):
        graph = flags == 0
        monkeypatch.delenv("AGX_ROW_DIFF", raising=False)
        a = run_engine(agx, tmp_dir, uu, 21, 50, 3, graph=graph, flags=flags)
        monkeypatch.setenv("AGX_ROW_DIFF", "1")
        b = run_engine(agx, tmp_dir, uu, 21, 50, 3, graph=graph, flags=flags)
        monkeypatch.delenv("AGX_ROW_DIFF")
        sa, sb = a["stats"], b["stats"]
        assert sa["rows_by_reference"] == 0
        for key in ("initial", "pre", "extended"):
            assert a[key] == b[key], key
        if graph:
            assert graph_mismatch(a["graph"], b["graph"]) is None
        if expect_diffed:
            assert sb["rows_by_reference"] > 0.4 * sb["n_hits"] and sb["upload_bytes"] < 0.85 * sa["upload_bytes"], (sa["upload_bytes"], sb["upload_bytes"], sb["rows_by_reference"])
        else:
            assert sb["rows_by_reference"] == 0 and sb["upload_bytes"] == sa["upload_bytes"]
        return b

    for uu in range(2):
        want = H.run_oracle(tmp, uu, 21, 50, 3)
        got = both_ways(tmp, uu)
        for key in ("initial", "pre", "extended"):
            assert got[key] == want[key], key
    assert both_ways(tmp, 0, flags=agx.AGX_FLAG_ONE_SHOT)["extended"] == H.run_oracle(tmp, 0, 21, 50, 3)["extended"]
    agx.cache_build(tmp, 1)                                          # from the unit cache: the rows are made again from the cached 2-bit rows
    both_ways(tmp, 1)
    # a soft-masked unit sequence
    masked = str(tmp_path / "masked"); shutil.copytree(tmp, masked)
    os.remove(os.path.join(masked, "_agx_unit.1.bin"))
    gp = os.path.join(masked, "_genome.0.fa")
    lines = open(gp).read().split("\n")
    open(gp, "w").write("\n".join(ln if ln.startswith(">") else "".join(c.lower() if (i // 7) % 2 else c for i, c in enumerate(ln)) for ln in lines))
    both_ways(masked, 0, expect_diffed=False)
