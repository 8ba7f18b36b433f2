"""The large checksum-only golden cases (tests/golden/big_md5.json): generator settings and the deterministic edits made to the generated
inputs.  Shared by make_golden_big.py (build container: runs the real reference) and the tests (regenerate the inputs, compare md5s)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import agx_data  # noqa: E402

CASES = {
    # BASELINE.json configs[1] at full size: one 4.6 Mb unit, exactly 1 000 000 pairs = one full batch followed by the EMPTY batch of AG:397 / 1258
    "cfg2": dict(synth=dict(seed=1000, chroms="4600000", pairs=1000000, L=100, k=5, coverage=5, sam_seq=0, threads=4), edit=None),
    # 1 000 100 pairs on one unit: crosses a batch boundary (BATCH, AG:37).  The SAM line pair that loadReadAli has read when it notices
    # (AG:1258-1259) is lost.  `edit` makes that very pair matter: it is moved into a stretch of the reference that no contig covers and
    # every other alignment there is removed, so — with --coverage 1 — keeping the pair adds a record of its own to
    # _pre_extended_contigs (make_golden_big.py checks that with the oracle run as one batch).
    "batch2": dict(synth=dict(seed=2001, chroms="2000000", pairs=1000100, L=100, k=5, coverage=1, sam_seq=0, threads=4), edit="lonely_pair_at_batch_boundary"),
}


def lonely_pair_at_batch_boundary(run):
    tmp = os.path.join(run, "tmp")
    # a stretch of >= 1400 reference positions without any contig alignment (PSL columns 16, 17 = tStart, tEnd)
    iv = sorted((int(t[15]), int(t[16])) for t in (ln.split("\t") for ln in open(os.path.join(tmp, "_contigs_genome.0.psl")) if ln.strip()))
    gap, end = None, 0
    for s0, e0 in iv:
        if s0 - end >= 1400 and end > 5000:
            gap = (end, s0)
            break
        end = max(end, e0)
    assert gap is not None
    left = gap[0] + 350                                # 1-based POS of the left mate; the right mate 400 further on
    path = os.path.join(tmp, "_reads_genome.0.bowtie")
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    n = (len(lines) - 1) // 2 * 2
    at = next(i for i in range(0, n, 2) if int(lines[i].split(b"\t", 1)[0]) >= 1000000)
    a, b = lines[at].split(b"\t"), lines[at + 1].split(b"\t")
    l, r = (a, b) if int(a[3]) <= int(b[3]) else (b, a)
    l[3], l[5], l[7] = b"%d" % left, b"100M", b"%d" % (left + 400)
    r[3], r[5], r[7] = b"%d" % (left + 400), b"100M", b"%d" % left
    lo, hi = gap[0] - 99, gap[1]
    keep = []
    for i in range(0, n, 2):
        if i == at:
            keep += [b"\t".join(a), b"\t".join(b)]
            continue
        p1, p2 = int(lines[i].split(b"\t", 4)[3]), int(lines[i + 1].split(b"\t", 4)[3])
        if lo <= p1 <= hi or lo <= p2 <= hi:
            continue                                   # an alignment that touches the uncovered stretch: removed
        keep += [lines[i], lines[i + 1]]
    with open(path, "wb") as f:
        f.write(b"\n".join(keep) + b"\n")
    return int(a[0])


def generate(name, out):
    """Regenerates the inputs of a case under `out`; returns the run directory."""
    case = CASES[name]
    run = agx_data.synth(out, **case["synth"])
    if case["edit"]:
        globals()[case["edit"]](run)
    return run
