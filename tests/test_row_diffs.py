"""The read rows' upload form (agx_core.h "read rows relative to the reference", agx_load.cpp build_row_diffs, agx_k_expand_rows).

No reference counterpart: the reference keeps every read as a string (AG:361-404).  The engine sends a unit's left-mate rows over PCIe as the bases
that differ from the reference under the row's first hit; the device turns them back into the very vote codes the 2-bit rows expand to.  The CPU tests
run the host encoder against the decoder function the kernel calls (one AGX_HD function, agx_row_chunk16) — on made-up geometries and on the golden
units' real alignments; the -m gpu test (test_gpu_parity.py) runs whole units both ways on the device.
"""
import os

import pytest

import harness as H
from hostsim import sim


@pytest.mark.parametrize("seed,n_pos,n_rows,stride,maxlen,mut,dirty", [
    (4, 100000, 5000, 152, 150, 20, False),      # 2x150 reads, 2 % of the bases differ
    (8, 100000, 5000, 152, 150, 20, True),       # noise behind the reads' ends: the codec stays exact (and keeps those rows as they are)
    (12, 5000, 3000, 104, 100, 5, False),        # many rows per position
    (16, 300, 2000, 16, 13, 100, False),         # one chunk per row
    (20, 70000, 3000, 4, 4, 0, False),           # the shortest stride: a row is one unit, a single difference already costs as much
    (24, 70000, 4000, 256, 253, 10, False),      # the longest rows the form takes: dozens of differences -> the limit of a row as it is
    (28, 40, 500, 152, 150, 10, False),          # reads longer than the unit: anchors hanging over both ends
    (32, 100000, 70000, 152, 150, 1000, False),  # reads that have nothing to do with the reference: every row as it is
    (36, 100000, 130, 100, 100, 10, False),      # a last block of two rows
])
def test_round_trip_on_made_up_rows(built, seed, n_pos, n_rows, stride, maxlen, mut, dirty):
    units, explicit = sim.rowdiff_roundtrip(seed, n_pos, n_rows, stride, maxlen, mut, dirty, threads=3)
    expl_units = (stride // 4 + 1) // 2
    assert units <= n_rows * expl_units                                   # never more than the rows as they are
    if mut == 1000:
        assert explicit >= n_rows * 0.8                                  # (short reads differ in fewer bases than a row as it is costs)
    if (seed, mut) == (4, 20):
        assert units * 2 < n_rows * stride // 4 // 2                      # and less than half of them on aligner-like rows


@pytest.mark.parametrize("seed,stride", [(1, 152), (2, 152), (4, 260)], ids=["rows not in hit order", "rows no hit names", "rows too long"])
def test_what_the_device_could_not_decode_is_declined(built, seed, stride):
    # the device finds a row's anchor by counting anchor bits from its block's first anchor on, and holds 64 rows in LDS: the encoder says no (the unit then
    # sends its 2-bit rows) when the rows are not numbered in the order of their first hits, when a row has no hit, when rows are longer than 256 bases
    with pytest.raises(sim.SimError) as e:
        sim.rowdiff_roundtrip(seed, 50000, 3000, stride, 150, 10)
    assert e.value.code == 3 and "declined" in e.value.msg


def test_round_trip_on_the_golden_units(golden, built):
    p = golden.params
    for u in range(p["units"]):
        r = sim.rowdiff_unit(golden.tmp, u, p["k"])
        assert r is not None, "unit sequence did not pack"
        assert r["explicit"] <= r["rows"]


@pytest.mark.parametrize("kw", [
    dict(seed=201, chroms="60000", pairs=20000, coverage=5, L=150, k=21),
    dict(seed=202, chroms="30000", pairs=9000, coverage=3, L=50, k=8, read_indel=0.3, read_clip=0.3, indel=0.005, multi=0.3),      # most left mates have runs: rows as they are
    dict(seed=203, chroms="20000,20000", part=2, pairs=8000, coverage=4, L=100, mixed_len=1),
], ids=lambda k: "seed%d" % k["seed"])
def test_round_trip_on_generated_units(built, tmp_path, kw):
    run = H.synth(str(tmp_path / "run"), sam_seq=0, **kw)
    meta = H.read_meta(run)
    tmp = os.path.join(run, "tmp")
    for u in range(meta["units"]):
        r = sim.rowdiff_unit(tmp, u, meta["k"])
        assert r is not None
        if kw["seed"] == 201:
            assert r["bytes_upload"] < 0.5 * r["bytes_2bit"], r           # 2x150 at 1 % SNP + sequencing errors: the form pays
