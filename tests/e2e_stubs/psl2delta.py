#!/usr/bin/env python3
"""TEST STUB helper: rewrites 21-column PSL lines as a MUMmer .delta file (what `nucmer -p prefix` leaves behind), so that the
--fastMap path (NUCMER + delta2psl, AG:524-729) can be driven by the same deterministic alignments as the BLAT path.
Usage: psl2delta.py ref.fa qry.fa in.psl out.delta"""
import sys

ref, qry, psl, out = sys.argv[1:5]
with open(out, "w") as f:
    f.write("%s %s\nNUCMER\n" % (ref, qry))
    for line in open(psl):
        c = line.rstrip("\n").split("\t")
        if len(c) < 21:
            continue
        strand, qname, qsize, qstart, qend = c[8], c[9], int(c[10]), int(c[11]), int(c[12])
        tname, tsize, tstart, tend = c[13], int(c[14]), int(c[15]), int(c[16])
        sizes = [int(x) for x in c[18].split(",") if x]
        qs = [int(x) for x in c[19].split(",") if x]
        ts = [int(x) for x in c[20].split(",") if x]
        f.write(">%s %s %d %d\n" % (tname, qname, tsize, qsize))
        if strand == "+":
            f.write("%d %d %d %d 0 0 0\n" % (tstart + 1, tend, qstart + 1, qend))
        else:
            f.write("%d %d %d %d 0 0 0\n" % (tstart + 1, tend, qend, qstart + 1))
        run = 0                                              # aligned columns since the last indel
        for i in range(len(sizes)):
            run += sizes[i]
            if i + 1 == len(sizes):
                break
            tgap = ts[i + 1] - (ts[i] + sizes[i])
            qgap = qs[i + 1] - (qs[i] + sizes[i])
            for _ in range(tgap):                            # bases only the reference has: positive distances
                f.write("%d\n" % (run + 1))
                run = 0
            for _ in range(qgap):                            # bases only the query has: negative distances
                f.write("%d\n" % -(run + 1))
                run = 0
        f.write("0\n")
