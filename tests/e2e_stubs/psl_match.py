#!/usr/bin/env python3
"""TEST STUB: stands in for BLAT in the refinement step.  Every query record (first 20 kb of an initial contig) that occurs verbatim,
forward or reverse-complemented, in a database record (an extended contig) yields one single-block 21-column PSL line."""
import sys


def fasta(path):
    recs, name, seq = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:].split()[0], []
        elif line:
            seq.append(line)
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


COMP = str.maketrans("ACGT", "TGCA")
db, qs, out = fasta(sys.argv[1]), fasta(sys.argv[2]), sys.argv[3]
with open(out, "w") as f:
    for qn, q in qs:
        if not q:
            continue
        for strand, s in (("+", q), ("-", q.translate(COMP)[::-1])):
            for tn, t in db:
                at = t.find(s)
                if at >= 0:
                    n = len(q)
                    f.write("\t".join(map(str, [n, 0, 0, 0, 0, 0, 0, 0, strand, qn, n, 0, n, tn, len(t), at, at + n, 1, "%d," % n, "0,", "%d," % at])) + "\n")
