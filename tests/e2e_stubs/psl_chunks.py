#!/usr/bin/env python3
"""TEST STUB: stands in for BLAT of output contigs against the whole genome.  Each query is cut into 40-base chunks; chunks that occur
verbatim in a genome record vote for (record, strand); chunks of the winning pair that lie on one diagonal are merged into blocks and one
21-column PSL line is written per query (several when the blocks are far apart).  Deterministic."""
import sys

COMP = str.maketrans("ACGT", "TGCA")
CH = 40


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


db, qs, out = fasta(sys.argv[1]), fasta(sys.argv[2]), sys.argv[3]
with open(out, "w") as f:
    for qn, q in qs:
        best = None
        for strand, s in (("+", q), ("-", q.translate(COMP)[::-1])):
            for tn, t in db:
                hits = []
                for i in range(0, len(s) - CH + 1, CH):
                    p = t.find(s[i:i + CH])
                    if p >= 0:
                        hits.append((i, p))
                if hits and (best is None or len(hits) > len(best[3])):
                    best = (strand, tn, t, hits)
        if not best or len(best[3]) < 3:
            continue
        strand, tn, t, hits = best
        blocks = []
        for i, p in hits:                                  # merge chunks on one diagonal; drop chunks that go backwards on the target
            if blocks and i == blocks[-1][0] + blocks[-1][2] and p == blocks[-1][1] + blocks[-1][2]:
                blocks[-1][2] += CH
            elif not blocks or (p >= blocks[-1][1] + blocks[-1][2] and i >= blocks[-1][0] + blocks[-1][2]):
                blocks.append([i, p, CH])
        groups, cur = [], [blocks[0]]
        for b in blocks[1:]:                               # a target jump of more than 2 kb starts a new alignment line
            if b[1] - (cur[-1][1] + cur[-1][2]) > 2000:
                groups.append(cur); cur = [b]
            else:
                cur.append(b)
        groups.append(cur)
        for g in groups:
            m = sum(b[2] for b in g)
            qni = sum(1 for a, b in zip(g, g[1:]) if b[0] > a[0] + a[2]); qbi = sum(b[0] - a[0] - a[2] for a, b in zip(g, g[1:]))
            tni = sum(1 for a, b in zip(g, g[1:]) if b[1] > a[1] + a[2]); tbi = sum(b[1] - a[1] - a[2] for a, b in zip(g, g[1:]))
            qs_, qe_ = g[0][0], g[-1][0] + g[-1][2]
            if strand == "-":
                qs_, qe_ = len(q) - qe_, len(q) - qs_
            f.write("\t".join(map(str, [m, 0, 0, 0, qni, qbi, tni, tbi, strand, qn, len(q), qs_, qe_, tn, len(t), g[0][1], g[-1][1] + g[-1][2], len(g),
                                        "".join("%d," % b[2] for b in g), "".join("%d," % b[0] for b in g), "".join("%d," % b[1] for b in g)])) + "\n")
