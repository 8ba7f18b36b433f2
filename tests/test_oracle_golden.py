"""The CPU oracle against the golden vectors captured from the real reference binary (tests/golden/make_golden.py)."""
import os

import pytest

import harness as H


def test_oracle_reproduces_reference_outputs(golden, built):
    p = golden.params
    for cov in p["coverages"]:
        for u in range(p["units"]):
            got = H.run_oracle(golden.tmp, u, p["k"], p["insert_variation"], cov)
            exp = golden.expected(cov, u)
            for key in ("initial", "pre", "extended"):
                assert got[key] == exp[key], "%s cov=%d unit=%d %s differs from the reference" % (golden.name, cov, u, key)


@pytest.mark.skipif(not H.have_reference(), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_oracle_matches_live_reference_on_fresh_seed(built, tmp_path):
    run = H.synth(str(tmp_path / "run"), seed=4242, chroms="30000,20000", part=2, pairs=8000, coverage=5, contig_min=800, contig_max=4000,
                  read_indel=0.2, read_clip=0.2, multi=0.2, sam_seq=0)
    ref, _ = H.run_reference(run)
    assert len(ref) == 4
    for u, exp in enumerate(ref):
        got = H.run_oracle(os.path.join(run, "tmp"), u, 5, 50, 5)
        assert got["initial"] == exp["initial"] and got["pre"] == exp["pre"] and got["extended"] == exp["extended"]


def test_oracle_error_messages(built, tmp_path):
    # same-strand mates -> "BOWTIE ALIGNMENT ERROR" (AG:1667-1671); unknown CIGAR op -> "unknown character" (AG:265-269)
    run = H.synth(str(tmp_path / "run"), seed=5, chroms="5000", pairs=200, coverage=2, sam_seq=0, multi=0, read_indel=0, read_clip=0, read_badclip=0)
    sam = os.path.join(run, "tmp", "_reads_genome.0.bowtie")
    lines = open(sam).read().split("\n")
    f = lines[0].split("\t"); g = lines[1].split("\t")
    g[1] = str((int(g[1]) & ~0x10) | (int(f[1]) & 0x10))
    bad = lines[:]; bad[1] = "\t".join(g)
    open(sam, "w").write("\n".join(bad))
    with pytest.raises(H.OracleError, match="BOWTIE ALIGNMENT ERROR"):
        H.run_oracle(os.path.join(run, "tmp"), 0, 5, 50, 2)
    f[5] = "50M50H"
    bad = lines[:]; bad[0] = "\t".join(f)
    open(sam, "w").write("\n".join(bad))
    with pytest.raises(H.OracleError, match="unknown character: H"):
        H.run_oracle(os.path.join(run, "tmp"), 0, 5, 50, 2)
    open(sam, "w").write("\n".join(lines[:1]) + "\n")
    with pytest.raises(H.OracleError, match="BROKEN BOWTIE FILE"):
        H.run_oracle(os.path.join(run, "tmp"), 0, 5, 50, 2)
