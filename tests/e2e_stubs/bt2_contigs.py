#!/usr/bin/env python3
"""TEST STUB: stands in for `bowtie2 -k 1 --no-mixed --no-discordant` of reads against output contigs.  A pair is reported when both
mates occur verbatim on one contig in FR orientation; everything else is reported unaligned.  Deterministic (first contig, first place)."""
import sys

COMP = str.maketrans("ACGT", "TGCA")


def fasta(path):
    recs, name, seq = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if not line:
            break
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:].split()[0], []
        else:
            seq.append(line)
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


contigs, r1, r2 = fasta(sys.argv[1]), fasta(sys.argv[2]), fasta(sys.argv[3])
out = sys.stdout
out.write("@HD\tVN:1.0\tSO:unsorted\n")
for n, s in contigs:
    out.write("@SQ\tSN:%s\tLN:%d\n" % (n, len(s)))
for (q, a), (_, b) in zip(r1, r2):
    hit = None
    for n, s in contigs:
        for fwd1 in (True, False):
            x = a if fwd1 else a.translate(COMP)[::-1]
            y = b.translate(COMP)[::-1] if fwd1 else b
            p, o = s.find(x), s.find(y)
            if p >= 0 and o >= 0 and ((fwd1 and p <= o) or (not fwd1 and o <= p)):
                hit = (n, fwd1, p, o, x, y)
                break
        if hit:
            break
    if not hit:
        out.write("%s\t77\t*\t0\t0\t*\t*\t0\t0\t%s\t*\n%s\t141\t*\t0\t0\t*\t*\t0\t0\t%s\t*\n" % (q, a, q, b))
        continue
    n, fwd1, p, o, x, y = hit
    f1, f2 = (99, 147) if fwd1 else (83, 163)
    out.write("%s\t%d\t%s\t%d\t42\t%dM\t=\t%d\t0\t%s\t*\n" % (q, f1, n, p + 1, len(x), o + 1, x))
    out.write("%s\t%d\t%s\t%d\t42\t%dM\t=\t%d\t0\t%s\t*\n" % (q, f2, n, o + 1, len(y), p + 1, y))
