"""Parity at BASELINE.json's configuration sizes (-m gpu): the HIP engine through the C-ABI on

  * configs[1] (E. coli shape, 4.6 Mb, exactly 1 000 000 pairs) and a 1 000 100-pair unit, against md5s of the REAL reference's output
    files (tests/golden/big_md5.json, made by tests/golden/make_golden_big.py in the build container; the inputs are regenerated here
    from the seed and their md5s are checked first);
  * configs[2] (A. thaliana shape: 5 units, 119 Mb, 20 M pairs) at full size, every unit byte for byte against the oracle (five oracle
    runs on five host threads beside the GPU work) — driven through the very job loop bench.py times (shard.run_job, pipelined units);
  * configs[3] (human chr1 shape, --part 4: four units of 62 Mb, 60 M pairs) at full size, all four slices through the job loop, every
    unit byte for byte against the oracle (four oracle runs on four host threads), plus a rebuild with every capacity started too small;
  * configs[4] ITSELF (r04): whole human, GRCh38 lengths — 24 units of 249 .. 47 Mb, 3.1 Gb, 400 M pairs of 2x150 bp, 400 read batches — as one job through
    shard.run_job on one GPU with admission by HBM: every unit built once with its first-guess capacities and within its admission estimate, the three
    smallest chromosomes (chr21, chr22, chrY at their full lengths) byte for byte against the ORACLE and against rebuilds from tiny capacities.  The read
    alignments are handed over staged (tmp/_agx_pairs.<u>.bin: as text the job is 170 GB), the checker's text exists for the three units only;

  * configs[4]'s chromosomes ONE AT A TIME at full size (r05; what fits any box: 6 GB of disk and 15 GB of memory per unit): chr21, chr22 and chrY — and chr1, 249 M positions,
    behind AGX_BIG_CHR1=1 — each generated alone at its GRCh38 length with exactly its share of the whole job's 400 M pairs of 2x150 (the whole job's read ids, so the 400 read
    batches and their lost line pairs, AG:1258-1259, fall where they fall in the job), handed over staged, one build, byte for byte against the ORACLE;
  * configs[4]'s 24-unit shard SHAPE (GRCh38 chromosome lengths / 256, 2x150 bp reads), one-shot units from their cache files through the
    job loop, every unit against the oracle.

AGX_SKIP_BIG=1 skips the two multi-minute cases.
"""
import hashlib
import json
import os
import threading
import time

import pytest

import harness as H

pytestmark = pytest.mark.gpu
BIG = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_md5.json")))
THREADS = min(32, os.cpu_count() or 1)
slow = pytest.mark.skipif(os.environ.get("AGX_SKIP_BIG") == "1", reason="AGX_SKIP_BIG=1")


@pytest.fixture(scope="module")
def agx():
    import aligngraph_amd as A
    if not os.path.exists(A.LIB_PATH):
        from aligngraph_amd import build as B
        B.build()
    assert A.device_count() > 0, "no HIP device: the gpu tests must run on the MI355X box"
    return A


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


@pytest.mark.parametrize("case", sorted(BIG))
def test_full_size_unit_matches_the_reference_md5(agx, case, built, tmp_path):
    from golden import big_cases
    e = BIG[case]
    run = big_cases.generate(case, str(tmp_path / "run"))
    tmp = os.path.join(run, "tmp")
    for fn, want in e["inputs_md5"].items():
        assert md5_file(os.path.join(tmp, fn)) == want, "the generator no longer reproduces %s of %s: regenerate tests/golden/big_md5.json" % (fn, case)
    with agx.Unit(k=e["k"], insert_variation=e["insert_variation"], coverage=e["coverage"]) as u:
        u.load_files(tmp, 0)
        u.upload(); u.build()
        got = u.finish()
        st = u.stats()
    for key, want in e["expected"].items():
        assert len(got[key]) == want["bytes"] and hashlib.md5(got[key]).hexdigest() == want["md5"], "%s: %s differs from the reference binary's output" % (case, key)
    assert st["build_attempts"] == 1, "a unit of this shape must not need a second build"
    if case == "batch2":
        assert st["pairs_in_file"] == 1000100 and st["n_hits"] < st["sam_line_pairs"]          # the second batch exists, and line pairs were dropped


CFG3 = [30427671, 19698289, 23459830, 18585056, 26975502]          # = bench.py CONFIGS["cfg3"]


@slow
def test_cfg3_full_size_every_unit_matches_the_oracle(agx, built, tmp_path):
    from aligngraph_amd import shard
    run = H.synth(str(tmp_path / "run"), seed=1000, chroms=",".join(map(str, CFG3)), pairs=20000000, L=100, k=5, coverage=5, sam_seq=1, threads=THREADS)      # full-width SAM lines (SEQ, QUAL, tags: what bowtie2 writes, AG:3609): 12 GB of text for the loaders
    tmp = os.path.join(run, "tmp")
    meta = H.read_meta(run)
    assert meta["unit_len"] == CFG3
    want, errs = {}, []

    def oracle(uu):
        try:
            want[uu] = H.run_oracle(tmp, uu, 5, 50, 5)
        except BaseException as e:
            errs.append(e)
    checkers = [threading.Thread(target=oracle, args=(uu,)) for uu in range(5)]
    for t in checkers:
        t.start()
    units, got, stats = {}, {}, {}
    with agx.Reads(os.path.join(tmp, "_reads.fa")) as reads:
        for uu in range(5):
            units[uu] = agx.Unit(k=5, insert_variation=50, coverage=5)
            units[uu].load_files(tmp, uu, reads=reads)

    def start_unit(uu):                                # bench.py's start_unit / run_unit
        units[uu].upload()

    def run_unit(uu):
        un = units[uu]
        un.build(); un.download()
        got[uu] = un.finish()
        stats[uu] = un.stats()
        un.release()
        return got[uu]["extended"]
    for job in range(2):                               # the second job runs on recycled memory blocks
        out = shard.run_job(meta["unit_len"], 0, 1, run_unit, None, None, inflight=5, start_unit=start_unit)
        assert sorted(out) == list(range(5))
        if job == 0:
            first = {uu: dict(got[uu]) for uu in range(5)}
    for un in units.values():
        un.close()
    for t in checkers:
        t.join()
    assert not errs, errs
    for uu in range(5):
        for key in ("initial", "pre", "extended"):
            assert first[uu][key] == want[uu][key], "unit %d: %s differs from the oracle" % (uu, key)
            assert got[uu][key] == want[uu][key], "unit %d: %s differs from the oracle in the second job" % (uu, key)
        assert stats[uu]["build_attempts"] == 1
    assert sum(stats[uu]["sam_line_pairs"] for uu in range(5)) > 20000000          # 20 M pairs, 5 % of them with a second hit, 2 % unaligned


HUMAN = (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309, 114364328,
         107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415)      # GRCh38, = bench.py CONFIGS["cfg5s"] x 16


def test_cfg5_shard_shape_at_1_256_every_unit_matches_the_oracle(agx, built, tmp_path):
    """configs[4] (whole human: 24 units, 2x150 bp reads) cannot run here at full size; its SHAPE can: the 24 chromosome lengths / 256 (12 Mb),
    1.5 M pairs of 2x150, every unit a one-shot unit from its cache file through the job loop (what `bench.py --config cfg5s` times at 1/16),
    every unit byte for byte against the oracle."""
    from aligngraph_amd import shard
    lens = [c // 256 for c in HUMAN]
    run = H.synth(str(tmp_path / "run"), seed=1005, chroms=",".join(map(str, lens)), pairs=1500000, L=150, k=5, coverage=5, sam_seq=0, threads=THREADS)
    tmp = os.path.join(run, "tmp")
    meta = H.read_meta(run)
    assert meta["unit_len"] == lens
    want, errs, nxt, lock = {}, [], iter(range(24)), threading.Lock()

    def oracle():
        try:
            while True:
                with lock:
                    uu = next(nxt, None)
                if uu is None:
                    return
                want[uu] = H.run_oracle(tmp, uu, 5, 50, 5)
        except BaseException as e:
            errs.append(e)
    checkers = [threading.Thread(target=oracle) for _ in range(min(8, THREADS))]
    for t in checkers:
        t.start()
    with agx.Reads(os.path.join(tmp, "_reads.fa")) as reads:
        for uu in range(24):
            agx.cache_build(tmp, uu, reads=reads, k=5)
    units, got, stats = {}, {}, {}
    for uu in range(24):
        units[uu] = agx.Unit(k=5, insert_variation=50, coverage=5, flags=agx.AGX_FLAG_ONE_SHOT)
        units[uu].load_files(tmp, uu)
        assert units[uu].stats()["from_cache"] == 1

    def run_unit(uu):
        un = units[uu]
        un.build(); un.download()
        got[uu] = un.finish()
        stats[uu] = un.stats()
        un.release()
        return got[uu]["extended"]
    out = shard.run_job(lens, 0, 1, run_unit, None, None, inflight=8, start_unit=lambda uu: units[uu].upload())
    assert sorted(out) == list(range(24))
    for un in units.values():
        un.close()
    for t in checkers:
        t.join()
    assert not errs, errs
    for uu in range(24):
        for key in ("initial", "pre", "extended"):
            assert got[uu][key] == want[uu][key], "unit %d: %s differs from the oracle" % (uu, key)
        assert stats[uu]["build_attempts"] == 1 and stats[uu]["n_pos"] >= lens[uu]


def _cfg5_unit_alone(agx, base, uu, threads, stream=False):
    """One chromosome of configs[4] alone: generator (--only-units), oracle on a thread beside the engine, the engine's three outputs and the oracle's.
    stream (r06): agx_unit_finish on a unit that has not been downloaded — the walk graph comes down in position windows and the walk begins on what has landed — instead of
    download, trim, finish."""
    import shutil
    run = H.synth(os.path.join(base, "u%d" % uu), seed=1000, chroms=",".join(map(str, HUMAN)), pairs=400000000, L=150, k=5, coverage=5, sam_seq=0, threads=threads,
                  pairs_bin=1, lean=1, oracle_units=uu, only_units=uu)
    tmp = os.path.join(run, "tmp")
    assert H.read_meta(run)["unit_len"] == list(HUMAN)
    want, errs = {}, []

    def oracle():
        try:
            want.update(H.run_oracle(os.path.join(run, "oracle_%d" % uu, "tmp"), uu, 5, 50, 5))
        except BaseException as e:
            errs.append(e)
    checker = threading.Thread(target=oracle)
    checker.start()
    with agx.Unit(k=5, insert_variation=50, coverage=5, flags=agx.AGX_FLAG_ONE_SHOT) as un:
        un.load_files(tmp, uu)
        need = un.hbm_needed()
        t0 = time.perf_counter()
        un.upload(); un.build()
        if stream:
            freed = need                                   # (nothing is given back early: nobody waits for this unit's HBM)
        else:
            un.download()
            freed = un.trim()                              # (r05) three quarters of the unit's HBM go back to the device before the walk; the walk's record fetches still find theirs
        got = un.finish()
        chain_ms = 1e3 * (time.perf_counter() - t0)
        st = un.stats()
        st["chain_ms"], st["trimmed"] = chain_ms, freed
    assert freed > 0.6 * need, "agx_unit_trim gave %.1f GB of %.1f GB back" % (freed / 1e9, need / 1e9)
    checker.join()
    shutil.rmtree(run, ignore_errors=True)
    assert not errs, errs
    return got, want, st, need


def _check_cfg5_unit(uu, got, want, st, need):
    assert st["build_attempts"] == 1 and st["n_pos"] >= HUMAN[uu] and st["pairs_in_file"] == 400000000, st
    assert st["device_bytes"] <= need
    assert abs(st["sam_line_pairs"] / (400e6 * 1.03 * HUMAN[uu] / sum(HUMAN)) - 1) < 0.02          # its share of the whole job's pairs (5 % with a second hit, 2 % unaligned)
    assert 300 <= st["sam_line_pairs"] - st["n_hits"] and st["n_hits"] > 0.9 * st["sam_line_pairs"]      # 399 batch boundaries, a line pair lost at (nearly) each, + the identity filter's
    for key in ("initial", "pre", "extended"):
        assert got[key] == want[key], "configs[4] unit %d (%d positions): %s differs from the oracle" % (uu, HUMAN[uu], key)
    assert got["extended"].count(b">") > HUMAN[uu] / 1e6 and len(got["extended"]) > 0.9 * HUMAN[uu]


@slow
def test_cfg5_chr21_chr22_chrY_alone_at_full_size_match_the_oracle(agx, built, tmp_path):
    """BASELINE configs[4]'s three smallest chromosomes, each as the unit it is in the whole-human job: full GRCh38 length (47, 51, 57 Mb), its share of the 400 M pairs of
    2x150 bp with the whole job's read numbering (400 read batches: every batch boundary loses a line pair, AG:1258-1259), staged hand-over, one-shot unit, ONE build — byte for
    byte against the oracle.  Sized for any box (6 GB of disk, 15 GB of memory per unit; the whole job, test_cfg5_whole_human_at_full_size, needs 70 GB / 200 GB)."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    free_disk = shutil.disk_usage(str(tmp_path)).free
    assert free_disk > 8e9, "this test needs 8 GB of scratch disk (here: %.1f GB)" % (free_disk / 1e9)
    try:
        mem_limit = int(open("/sys/fs/cgroup/memory.max").read())
    except (OSError, ValueError):
        mem_limit = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    side_by_side = 3 if free_disk > 25e9 and mem_limit > 80e9 else 1
    units = (20, 21, 23)
    with ThreadPoolExecutor(max_workers=side_by_side) as ex:
        res = list(ex.map(lambda uu: _cfg5_unit_alone(agx, str(tmp_path), uu, max(2, THREADS // side_by_side)), units))
    for uu, (got, want, st, need) in zip(units, res):
        _check_cfg5_unit(uu, got, want, st, need)
    print("configs[4] alone: " + "; ".join("unit %d: %d hits, %.1f GB of HBM (%.1f given back after the download), node sweep %.1f ms, walk %.0f ms" % (uu, r[2]["n_hits"], r[2]["device_bytes"] / 1e9, r[2]["trimmed"] / 1e9, r[2]["ms_node_sweep"], r[2]["ms_walk"])
                                          for uu, r in zip(units, res)))


def _box():
    import shutil
    import tempfile
    try:
        mem_limit = int(open("/sys/fs/cgroup/memory.max").read())
    except (OSError, ValueError):
        mem_limit = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    return shutil.disk_usage(tempfile.gettempdir()).free, mem_limit


@slow
def test_cfg5_chrX_alone_at_full_size_matches_the_oracle(agx, built, tmp_path):
    """r06 (a LARGE unit in the default suite): chrX of configs[4] — 156 040 895 positions, 20 M pairs of 2x150 with the whole job's read numbering, 35 GB of
    HBM — alone, one-shot, one build, its download STREAMED into its walk (eight position windows, sixteen walkers gated on them), byte for byte against the oracle.  The unit is
    two and a half times the largest one the suite held so far (62 Mb); 32-bit offsets, the memory region, the tile-ordered upload in eight windows and the streamed download all run
    at a size where they matter.  Needs 25 GB of scratch disk and 90 GB of host memory (the oracle holds 50): skipped where the box has less.  chr1 (249 Mb) the same way behind AGX_BIG_CHR1=1."""
    free_disk, mem_limit = _box()
    if free_disk < 25e9 or mem_limit < 90e9:
        pytest.skip("chrX alone against the oracle needs 25 GB of scratch disk and 90 GB of host memory (here: %.0f GB, %.0f GB)" % (free_disk / 1e9, mem_limit / 1e9))
    got, want, st, need = _cfg5_unit_alone(agx, str(tmp_path), 22, THREADS, stream=True)
    _check_cfg5_unit(22, got, want, st, need)
    print("configs[4] chrX alone, download streamed into the walk: %d positions, %d hits, %.1f GB of HBM, upload -> output bytes %.1f ms (node sweep %.1f ms, download %.1f ms, walk incl. the download %.0f ms)" %
          (st["n_pos"], st["n_hits"], st["device_bytes"] / 1e9, st["chain_ms"], st["ms_node_sweep"], st["ms_download"], st["ms_walk"]))


@pytest.mark.skipif(os.environ.get("AGX_BIG_CHR1") != "1", reason="chr1 alone against the oracle takes the oracle ~15 minutes and 80 GB of host memory: AGX_BIG_CHR1=1 (a passing run's log: profiles/r05_chr1_oracle.txt)")
def test_cfg5_chr1_alone_at_full_size_matches_the_oracle(agx, built, tmp_path):
    """configs[4]'s LARGEST unit — chr1, 248 956 422 positions, 33 M pairs of 2x150, 57 GB of HBM — against the oracle (r03 checked it against the serial executor only)."""
    got, want, st, need = _cfg5_unit_alone(agx, str(tmp_path), 0, THREADS, stream=os.environ.get("AGX_BIG_CHR1_NO_STREAM") != "1")
    _check_cfg5_unit(0, got, want, st, need)
    print("configs[4] chr1 alone: %d positions, %d hits, %.1f GB of HBM (%.1f given back after the download), upload -> output bytes %.1f ms (node sweep %.1f ms, download %.1f ms, walk %.0f ms); outputs %d / %d / %d bytes identical to the oracle" %
          (st["n_pos"], st["n_hits"], st["device_bytes"] / 1e9, st["trimmed"] / 1e9, st["chain_ms"], st["ms_node_sweep"], st["ms_download"], st["ms_walk"], len(got["initial"]), len(got["pre"]), len(got["extended"])))


@slow
def test_human_sized_units_cfg4_slices(agx, built, tmp_path, monkeypatch):
    """configs[3] at full size: human chr1 (GRCh38 length) cut by --part 4 (formalizeGenome's rule, AG:3382-3413) into four units of 62 Mb, 60 M
    pairs of 2x100 — all four through the job loop bench.py times, every unit byte for byte against the ORACLE (four oracle runs on four host
    threads), then one slice again with every capacity started too small.  (chr1 WHOLE as one unit — r03 checked it here against the serial executor —
    is unit 0 of the whole-human job below.)"""
    import shutil
    from aligngraph_amd import shard
    run4 = H.synth(str(tmp_path / "cfg4"), seed=1004, chroms="248956422", part=4, pairs=60000000, L=100, k=5, coverage=5, sam_seq=0, threads=THREADS)
    tmp4 = os.path.join(run4, "tmp")
    lens = H.read_meta(run4)["unit_len"]
    assert len(lens) == 4 and sum(lens) == 248956422
    want4, errs = {}, []

    def oracle(uu):
        try:
            want4[uu] = H.run_oracle(tmp4, uu, 5, 50, 5)
        except BaseException as e:
            errs.append(e)
    checkers = [threading.Thread(target=oracle, args=(uu,)) for uu in range(4)]
    for t in checkers:
        t.start()
    units, got4, stats4 = {}, {}, {}
    with agx.Reads(os.path.join(tmp4, "_reads.fa")) as reads:
        for uu in range(4):
            units[uu] = agx.Unit(k=5, insert_variation=50, coverage=5, flags=agx.AGX_FLAG_ONE_SHOT)
            units[uu].load_files(tmp4, uu, reads=reads)
            assert units[uu].stats()["from_cache"] == 0

    def run_unit(uu):
        un = units[uu]
        un.build(); un.download()
        got4[uu] = un.finish()
        stats4[uu] = un.stats()
        un.release()
        return got4[uu]["extended"]
    out = shard.run_job(lens, 0, 1, run_unit, None, None, inflight=4, start_unit=lambda uu: units[uu].upload())
    assert sorted(out) == list(range(4))
    for un in units.values():
        un.close()
    monkeypatch.setenv("AGX_TEST_SMALL_CAPS", "1")          # every capacity far too small: tile lists, node pool, sparse table all regrow
    with agx.Unit(k=5, insert_variation=50, coverage=5) as u2:
        u2.load_files(tmp4, 3)
        u2.upload(); u2.build()
        again4 = u2.finish()
        assert u2.stats()["build_attempts"] > 1
    monkeypatch.delenv("AGX_TEST_SMALL_CAPS")
    for t in checkers:
        t.join()
    shutil.rmtree(run4, ignore_errors=True)
    assert not errs, errs
    for uu in range(4):
        for key in ("initial", "pre", "extended"):
            assert got4[uu][key] == want4[uu][key], "cfg4 slice %d: %s differs from the oracle" % (uu, key)
        assert stats4[uu]["build_attempts"] == 1 and stats4[uu]["n_pos"] >= lens[uu]
    assert sum(stats4[uu]["sam_line_pairs"] for uu in range(4)) > 60000000
    for key in ("initial", "pre", "extended"):
        assert again4[key] == want4[3][key], "cfg4: %s of slice 3 differs from the oracle after its capacities regrew" % key


@slow
def test_cfg5_whole_human_at_full_size(agx, built, tmp_path, monkeypatch):
    """BASELINE configs[4] as a job: GRCh38's 24 chromosome lengths (3.1 Gb), 400 M pairs of 2x150 bp (400 read batches: BATCH, AG:37), --coverage 5, every unit a
    one-shot unit, all of them through shard.run_job on ONE GPU, admitted by agx_unit_hbm_needed (eight units in flight at most; chr1 alone takes 57 GB of HBM).
    The read alignments cross staged (tools/agx_synth.cpp --pairs-bin 1: the engine's own line parser, batch rules and staging make them; tests/test_staged_pairs.py
    compares that hand-over with the text path byte for byte) — as SAM + reads text this job is 170 GB.  For chr21, chr22 and chrY (47 - 57 Mb at full length) the
    generator also writes the checker's text (the unit's SAM lines + a reads file with placeholders for everybody else's reads): their bytes must be the ORACLE's,
    and a rebuild from absurdly small capacities must give them again.  Every unit: one build, HBM within its admission estimate, the batch boundaries' lost line
    pairs, records and N50 of the right order."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    from aligngraph_amd import shard
    # what the job needs of the box: 66 GB of scratch disk (staged alignments 23 GB, contig text, the checkers' SAM) and ~150 GB of host memory (24 staged units)
    free_disk = shutil.disk_usage(str(tmp_path)).free
    try:
        mem_limit = int(open("/sys/fs/cgroup/memory.max").read())
    except (OSError, ValueError):
        mem_limit = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    if free_disk < 70e9 or mem_limit < 200e9:
        pytest.skip("configs[4] at full size needs 70 GB of scratch disk and 200 GB of host memory (here: %.0f GB, %.0f GB)" % (free_disk / 1e9, mem_limit / 1e9))
    CHECK = (20, 21, 23)                                       # chr21, chr22, chrY
    run = H.synth(str(tmp_path / "run"), seed=1000, chroms=",".join(map(str, HUMAN)), pairs=400000000, L=150, k=5, coverage=5, sam_seq=0, threads=THREADS,
                  pairs_bin=1, lean=1, oracle_units=",".join(map(str, CHECK)))
    tmp = os.path.join(run, "tmp")
    lens = H.read_meta(run)["unit_len"]
    assert lens == list(HUMAN)
    want, errs = {}, []

    def oracle(uu):
        try:
            want[uu] = H.run_oracle(os.path.join(run, "oracle_%d" % uu, "tmp"), uu, 5, 50, 5)
        except BaseException as e:
            errs.append(e)
    checkers = [threading.Thread(target=oracle, args=(uu,)) for uu in CHECK]
    for t in checkers:
        t.start()

    def load(uu):
        un = agx.Unit(k=5, insert_variation=50, coverage=5, flags=agx.AGX_FLAG_ONE_SHOT)
        un.load_files(tmp, uu)
        return un
    with ThreadPoolExecutor(max_workers=8) as ex:
        units = dict(zip(range(24), ex.map(load, range(24))))
    need = {uu: units[uu].hbm_needed() for uu in range(24)}
    total_hbm = agx.device_memory(0)[1]
    assert need[0] == max(need.values()) and 40e9 < need[0] < 70e9 and sum(need.values()) > 2 * total_hbm      # the job does not fit the device at once: admission matters
    got, stats = {}, {}

    def run_unit(uu):
        un = units[uu]
        un.build(); un.download()
        got[uu] = un.finish()
        stats[uu] = un.stats()
        un.release()
        return got[uu]["extended"]
    out = shard.run_job(lens, 0, 1, run_unit, None, None, inflight=8, start_unit=lambda uu: units[uu].upload(), hbm_need=need, hbm_budget=int(0.85 * total_hbm))
    assert sorted(out) == list(range(24))
    for un in units.values():
        un.close()
    agx.pool_trim(0, host=True)
    again = {}
    monkeypatch.setenv("AGX_TEST_SMALL_CAPS", "1")              # every capacity far too small: tile lists, node pool, sparse table all regrow
    for uu in CHECK:
        with agx.Unit(k=5, insert_variation=50, coverage=5) as u2:
            u2.load_files(tmp, uu)
            u2.upload(); u2.build()
            again[uu] = u2.finish()
            assert u2.stats()["build_attempts"] > 1
    monkeypatch.delenv("AGX_TEST_SMALL_CAPS")
    for t in checkers:
        t.join()
    shutil.rmtree(run, ignore_errors=True)
    assert not errs, errs
    for uu in range(24):
        st = stats[uu]
        assert st["build_attempts"] == 1 and st["n_pos"] >= lens[uu] and st["pairs_in_file"] == 400000000, (uu, st)
        assert st["device_bytes"] <= need[uu], "unit %d took %.1f GB of HBM, it was admitted with %.1f GB" % (uu, st["device_bytes"] / 1e9, need[uu] / 1e9)
        assert abs(st["sam_line_pairs"] / (400e6 * 1.03 * lens[uu] / sum(lens)) - 1) < 0.02          # its share of the pairs (5 % with a second hit, 2 % unaligned)
        assert 300 <= st["sam_line_pairs"] - st["n_hits"] - 0 and st["n_hits"] > 0.9 * st["sam_line_pairs"]      # 399 batch boundaries, a line pair lost at (nearly) each, + the identity filter's
        ext = got[uu]["extended"]
        assert ext.count(b">") > lens[uu] / 1e6 and len(ext) > 0.9 * lens[uu], "unit %d: %d extended contigs, %d bytes" % (uu, ext.count(b">"), len(ext))
    for uu in CHECK:
        for key in ("initial", "pre", "extended"):
            assert got[uu][key] == want[uu][key], "unit %d (%d positions): %s differs from the oracle" % (uu, lens[uu], key)
            assert again[uu][key] == want[uu][key], "unit %d: %s differs from the oracle after its capacities regrew" % (uu, key)
    print("whole human: 24 units, %d hits, largest unit %.1f GB of HBM, node sweeps %.1f ms in all, walks %.0f ms in all" %
          (sum(s["n_hits"] for s in stats.values()), stats[0]["device_bytes"] / 1e9, sum(s["ms_node_sweep"] for s in stats.values()), sum(s["ms_walk"] for s in stats.values())))
