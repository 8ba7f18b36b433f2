"""Generates tests/golden/e2e.tar.gz — run in the build container only (needs /root/reference).

A whole AlignGraph run, front to back, on seeded synthetic inputs with the aligners replaced by the deterministic test stubs of
tests/e2e_stubs/ (bowtie2 replays a pre-generated SAM, pblat replays the contig PSL and runs an exact matcher for the refinement
step, nucmer writes the same alignments as .delta files for the --fastMap variant).  The REAL reference binary (README build; the -O2 build cannot run the aligner threads) is run for every command variant and
its stdout, its final FASTA files and its per-unit tmp/ outputs are stored as the expected bytes.  Data only — no reference source.
"""
import io
import os
import shutil
import subprocess
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import harness as H  # noqa: E402

STUBS = os.path.join(ROOT, "tests", "e2e_stubs")
BASE = ["--read1", "reads_1.fa", "--read2", "reads_2.fa", "--contig", "contigs.fa", "--genome", "genome.fa", "--distanceLow", "100",
        "--distanceHigh", "1500", "--extendedContig", "e.fa", "--remainingContig", "r.fa"]
VARIANTS = {
    "default": BASE + ["--coverage", "4"],
    "flags": BASE + ["--coverage", "3", "--kMer", "7", "--insertVariation", "40", "--ratioCheck", "--uniqueExtension"],
    "masb": BASE + ["--coverage", "3", "--misassemblyRemoval"],
    "fastmap": BASE + ["--coverage", "4", "--fastMap"],              # NUCMER (stub) + the reference's own delta2psl
    # (--part > 1 is not in the end-to-end set: the reference indexes genomeIds by unit in refinement, AG:3102, and crashes on it)
}
INPUTS = ["reads_1.fa", "reads_2.fa", "contigs.fa", "genome.fa", "stub/reads_genome.sam"]


def run_reference(src, args):
    work = src + ".refrun"
    if os.path.exists(work):
        shutil.rmtree(work)
    shutil.copytree(src, work)
    shutil.rmtree(os.path.join(work, "tmp"))
    env = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"))
    out = subprocess.run([H.REF_O0] + args, cwd=work, env=env, stdout=subprocess.PIPE, check=True).stdout
    return work, out


def main():
    H.build()
    tar_path = os.path.join(HERE, "e2e.tar.gz")
    with tarfile.open(tar_path, "w:gz", compresslevel=9) as tar:
        def add(arc, data):
            ti = tarfile.TarInfo(arc); ti.size = len(data); ti.mtime = 0
            tar.addfile(ti, io.BytesIO(data))
        for name, args in VARIANTS.items():
            part = int(args[args.index("--part") + 1]) if "--part" in args else 1
            # the aligner stubs replay per-unit data, so the data set is generated with the same --part the command uses
            extra = dict(seed=17, chimeric=0.5, contig_overlap=0.3, contig_min=500, contig_max=3500) if name == "masb" else dict(seed=21, contig_min=600, contig_max=3000)
            if name == "fastmap":
                extra = dict(seed=23, contig_min=600, contig_max=3000, contig_minus=0.5)       # both strands through delta2psl
            src = H.synth("/tmp/golden_e2e_" + name, chroms="12000,9000", part=part, pairs=4000, coverage=4, e2e=1, sam_seq=0, multi=0.1, **extra)
            work, out = run_reference(src, args)
            for fn in INPUTS + ["stub/" + f for f in sorted(os.listdir(os.path.join(src, "stub"))) if f.endswith(".psl")]:
                add("%s/in/%s" % (name, fn), open(os.path.join(src, fn), "rb").read())
            add("%s/args.txt" % name, "\n".join(args).encode())
            add("%s/expected/stdout.txt" % name, out)
            for fn in ("e.fa", "r.fa", "in.fa", "ex.fa", "corrected_e.fa", "corrected_r.fa"):
                if os.path.exists(os.path.join(work, fn)):
                    add("%s/expected/%s" % (name, fn), open(os.path.join(work, fn), "rb").read())
            for fn in sorted(os.listdir(os.path.join(work, "tmp"))):
                keep = ("_initial_contigs", "_pre_extended_contigs", "_extended_contigs", "_short_initial", "_checkpoint", "_contigs.fa", "_chaff", "_genome")
                if "--fastMap" in args:
                    keep += ("_contigs_genome",)                      # NUCMER's .delta and what delta2psl made of it
                if fn.startswith(keep):
                    add("%s/expected/tmp/%s" % (name, fn), open(os.path.join(work, "tmp", fn), "rb").read())
            print(name, "extended records:", open(os.path.join(work, "e.fa")).read().count(">"), "remaining:", open(os.path.join(work, "r.fa")).read().count(">"))
            print(out.decode()[-300:])
            shutil.rmtree(work); shutil.rmtree(src)
    print(tar_path, os.path.getsize(tar_path), "bytes")


if __name__ == "__main__":
    main()
