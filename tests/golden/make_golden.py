"""Generates the golden fixtures under tests/golden/ — run in the build container only (needs /root/reference).

For every case: seeded synthetic inputs (tools/agx_synth), replayed through the REAL reference binary built from
/root/reference/AlignGraph/AlignGraph.cpp (oracle/Makefile -> oracle/_ref/AlignGraph_ref, README build line, and the
-O2 build; both must agree), and the reference's three per-unit output files captured as the expected bytes.

A fixture is data only: <case>.tar.gz holds tmp/ INPUT files (_genome.u.fa, _contigs.fa, _contigs_genome.u.psl,
_reads.fa, _reads_genome.u.bowtie), params.json, and expected/<coverage>/{_initial_contigs,_pre_extended_contigs,
_extended_contigs}.u.fa as written by the reference.  Nothing of the reference's source is stored.
"""
import io
import json
import os
import shutil
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import harness as H  # noqa: E402

CASES = {
    # name: (synth kwargs, list of --coverage values replayed on the same inputs)
    "clean": (dict(seed=11, chroms="20000", pairs=4000, L=100, k=5, contig_min=1500, contig_max=3000, read_indel=0, read_clip=0,
                   read_badclip=0, multi=0, indel=0, contig_split=0, contig_dup=0, contig_overlap=0, contig_lowid=0, sam_seq=0), [5, 20]),
    "noisy": (dict(seed=12, chroms="20000", pairs=5000, L=100, k=5, contig_min=800, contig_max=4000, read_indel=0.3, read_clip=0.3,
                   read_badclip=0.03, multi=0.3, indel=0.005, snp=0.02, frag_sd=60, sam_seq=0), [5]),
    "contigs": (dict(seed=13, chroms="25000", pairs=5000, L=100, k=5, contig_min=300, contig_max=2500, contig_minus=0.5, contig_split=0.5,
                     contig_dup=0.3, contig_overlap=0.5, contig_lowid=0.1, indel=0.004, sam_seq=0), [5]),
    "multiunit": (dict(seed=14, chroms="16000,10000", part=2, pairs=6000, L=100, k=5, contig_min=500, contig_max=3000, sam_seq=0), [4]),
    "k21_L150": (dict(seed=15, chroms="20000", pairs=3000, L=150, k=21, insert_variation=30, contig_min=1000, contig_max=5000, read_indel=0.2,
                      read_clip=0.2, sam_seq=1), [5]),
    # the SAME pairs as a forward-order twin, with the read ids reversed (pair i becomes pair N-1-i in tmp/_reads.fa and in
    # the SAM): pins the first-come rule of a bucket's variants (AG:1381 vs 1386) — the stored keys of a node are those of the first
    # read to arrive, and they reach the output headers.  main() checks that the reference's output really differs from the twin's.
    "reversed": (dict(seed=16, chroms="20000", pairs=5000, L=100, k=5, contig_min=800, contig_max=4000, read_indel=0.2, read_clip=0.2,
                      multi=0.2, frag_sd=60, sam_seq=0), [5]),
    # trimmed reads: 40 % of the pairs cut to 60 / 75 / 90 bp (the mates of a pair share a length, AG:3454; the event loop bound is per pair,
    # AG:1672/1681), with indels, clips and multi-hits on both kinds — pairs of different lengths in one unit, one row stride for all of them
    "mixedlen": (dict(seed=17, chroms="30000", pairs=6000, L=100, k=5, mixed_len=0.4, contig_min=800, contig_max=4000, read_indel=0.3, read_clip=0.3,
                      multi=0.2, frag_sd=60, sam_seq=0), [5]),
}
REVERSED = {"reversed"}


def reverse_pair_order(run, units):
    """Renumbers the pairs of a generated run N-1-i and rewrites tmp/_reads.fa and every unit's SAM in the new id order (the hits of one
    pair keep their order: bowtie2 -k prints them best first)."""
    tmp = os.path.join(run, "tmp")
    recs = open(os.path.join(tmp, "_reads.fa")).read().split(">")[1:]
    n = len(recs) // 2
    with open(os.path.join(tmp, "_reads.fa"), "w") as f:
        for new in range(n):
            old = n - 1 - new
            for m in (0, 1):
                f.write(">%d\n%s\n" % (new, recs[2 * old + m].split("\n")[1]))
    for u in range(units):
        path = os.path.join(tmp, "_reads_genome.%d.bowtie" % u)
        lines = open(path).read().split("\n")
        lines = [ln for ln in lines if ln]
        pairs = [(n - 1 - int(lines[i].split("\t")[0]), i) for i in range(0, len(lines), 2)]
        pairs.sort()                                   # stable: the hits of one pair stay in their order
        with open(path, "w") as f:
            for new, i in pairs:
                for ln in (lines[i], lines[i + 1]):
                    t = ln.split("\t"); t[0] = str(new)
                    f.write("\t".join(t) + "\n")


INPUTS = ("_genome.%d.fa", "_contigs_genome.%d.psl", "_reads_genome.%d.bowtie")


def main():
    H.build()
    if not (H.have_reference(True) and H.have_reference(False)):
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    for name, (kw, coverages) in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:          # regenerate only the named cases (a .tar.gz carries its creation time)
            continue
        run = H.synth("/tmp/golden_" + name, coverage=coverages[0], **kw)
        meta = H.read_meta(run)
        forward = None
        if name in REVERSED:
            forward, _ = H.run_reference(run, opt=True)
            reverse_pair_order(run, meta["units"])
        expected = {}
        for cov in coverages:
            # coverage is read from tmp/_command.txt by --resume
            with open(os.path.join(run, "tmp", "_command.txt")) as f:
                lines = f.read().split("\n")
            lines[lines.index("--coverage") + 1] = str(cov)
            with open(os.path.join(run, "tmp", "_command.txt"), "w") as f:
                f.write("\n".join(lines))
            o2, _ = H.run_reference(run, opt=True)
            o0, _ = H.run_reference(run, opt=False)
            assert o2 == o0, "README build and -O2 build of the reference disagree on " + name
            expected[cov] = o2
            if forward is not None:
                assert o2 != forward, "reversing the read order does not change the reference's output: the fixture pins nothing"
                print("   forward / reversed pre-extended records:", [o["pre"].count(b">") for o in forward], [o["pre"].count(b">") for o in o2])
            for u, o in enumerate(o2):
                mine = H.run_oracle(os.path.join(run, "tmp"), u, meta["k"], meta["insert_variation"], cov)
                assert all(mine[k] == o[k] for k in ("initial", "pre", "extended")), "oracle differs from the reference on %s unit %d" % (name, u)
        out = os.path.join(HERE, name + ".tar.gz")
        with tarfile.open(out, "w:gz", compresslevel=9) as tar:
            def add(arc, data):
                ti = tarfile.TarInfo(arc); ti.size = len(data); ti.mtime = 0
                tar.addfile(ti, io.BytesIO(data))
            params = dict(k=meta["k"], insert_variation=meta["insert_variation"], units=meta["units"], coverages=coverages, synth=kw)
            add("params.json", json.dumps(params, indent=1, sort_keys=True).encode())
            for fn in ("_contigs.fa", "_reads.fa"):
                add("tmp/" + fn, open(os.path.join(run, "tmp", fn), "rb").read())
            for u in range(meta["units"]):
                for pat in INPUTS:
                    add("tmp/" + pat % u, open(os.path.join(run, "tmp", pat % u), "rb").read())
            for cov, outs in expected.items():
                for u, o in enumerate(outs):
                    add("expected/%d/_initial_contigs.%d.fa" % (cov, u), o["initial"])
                    add("expected/%d/_pre_extended_contigs.%d.fa" % (cov, u), o["pre"])
                    add("expected/%d/_extended_contigs.%d.fa" % (cov, u), o["extended"])
        print(name, os.path.getsize(out), "bytes;", meta["units"], "units; extended records:",
              [o["extended"].count(b">") for o in expected[coverages[0]]])
        shutil.rmtree(run)


if __name__ == "__main__":
    main()
