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


def test_batch_boundary_unit_matches_the_reference_md5(built, tmp_path):
    """1 000 100 pairs on one unit cross a batch boundary (BATCH, AG:37); the line pair that loadReadAli has read when it
    notices (AG:1258-1259) is lost — and tests/golden/big_cases.py edits that very pair so that keeping it changes the output.
    tests/golden/big_md5.json holds the md5s of what the REAL reference wrote for these inputs (regenerated here from the seed): the
    oracle and the serial executor of the kernels' lane functions must both reproduce them."""
    import hashlib
    import json
    from hostsim import sim
    from golden import big_cases
    e = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_md5.json")))["batch2"]
    run = big_cases.generate("batch2", str(tmp_path / "run"))
    tmp = os.path.join(run, "tmp")
    for fn, want in e["inputs_md5"].items():
        with open(os.path.join(tmp, fn), "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == want, "the generator no longer reproduces %s: regenerate tests/golden/big_md5.json" % fn
    got = H.run_oracle(tmp, 0, e["k"], e["insert_variation"], e["coverage"])
    ser = sim.run(tmp, 0, k=e["k"], insert_variation=e["insert_variation"], coverage=e["coverage"])
    for key, want in e["expected"].items():
        assert hashlib.md5(got[key]).hexdigest() == want["md5"], "oracle: %s differs from the reference" % key
        assert hashlib.md5(ser[key]).hexdigest() == want["md5"], "serial executor: %s differs from the reference" % key
    # (make_golden_big.py checked that the rule is really exercised: run as ONE batch the same inputs give a different _pre_extended_contigs)


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
