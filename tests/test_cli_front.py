"""AlignGraph_amd's front end on several threads (r05: maxReadLength AG:3197-3226 and formalizeInput for the reads AG:3420-3518; aligngraph_amd/csrc/agx_cli.cpp): the
three read files it writes must be, byte for byte, what its line-by-line form writes (AGX_CLI_SERIAL=1: the form that r04 compared with the reference's own files and that
tests/test_cli.py compares through whole runs) — for the inputs the threaded form takes (a header and one sequence line per record) and for the ones it hands back (records
over several lines, a missing last newline, mates of different lengths cut to the shorter).  No GPU: the run stops where the unit loop would begin."""
import os
import subprocess

import pytest

import harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "e2e_stubs")
CLI = os.environ.get("AGX_CLI_PATH", os.path.join(ROOT, "aligngraph_amd", "AlignGraph_amd"))      # (AGX_CLI_PATH: a sanitizer build of the same source)
ARGS = ["--read1", "reads_1.fa", "--read2", "reads_2.fa", "--contig", "contigs.fa", "--genome", "genome.fa", "--distanceLow", "100", "--distanceHigh", "1500",
        "--extendedContig", "e.fa", "--remainingContig", "r.fa", "--coverage", "5"]


def front(work, **env):
    import shutil
    shutil.rmtree(os.path.join(work, "tmp"), ignore_errors=True)
    e = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"), **env)
    p = subprocess.run([CLI] + ARGS, cwd=work, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p, {f: open(os.path.join(work, "tmp", f), "rb").read() for f in ("_reads.fa", "_reads_1.fa", "_reads_2.fa") if os.path.exists(os.path.join(work, "tmp", f))}


@pytest.fixture(scope="module")
def run_dir(tmp_path_factory):
    from aligngraph_amd import build as B
    B.build()
    return H.synth(str(tmp_path_factory.mktemp("front") / "run"), seed=77, chroms="60000,40000", pairs=30000, L=100, k=5, coverage=5, e2e=1, sam_seq=0)


def test_threaded_front_end_writes_the_serial_files(built, run_dir):
    ps, serial = front(run_dir, AGX_CLI_SERIAL="1")
    assert b"(0) Alignment finished" in ps.stdout and len(serial) == 3 and serial["_reads.fa"].count(b">") == 60000
    for threads in ("2", "5", "16"):
        pt, got = front(run_dir, AGX_CLI_THREADS=threads, AGX_CLI_TIMING="1", AGX_CLI_FAST_MIN="0")
        assert pt.stdout == ps.stdout
        assert b"reads on %s threads" % threads.encode() in pt.stderr, pt.stderr[-300:]
        assert got == serial, "threads=%s" % threads


@pytest.mark.parametrize("shape", ["multiline", "no_final_newline", "unequal_mates", "empty_line"])
def test_inputs_the_threaded_form_hands_back(built, run_dir, tmp_path, shape):
    import shutil
    work = str(tmp_path / "w")
    shutil.copytree(run_dir, work)
    r1 = open(os.path.join(work, "reads_1.fa"), "rb").read().split(b"\n")
    r2 = open(os.path.join(work, "reads_2.fa"), "rb").read().split(b"\n")
    if shape == "multiline":                          # one record of each file over two lines (at the same line numbers: the files stay consistent)
        for r in (r1, r2):
            s = r[20001]; r[20001:20002] = [s[:40], s[40:]]
    elif shape == "no_final_newline":
        r1, r2 = r1[:-1], r2[:-1]
        r1[-1] = r1[-1]; r2[-1] = r2[-1]
    elif shape == "unequal_mates":                    # taken by the threaded form: the longer mate is cut
        r1[101] = r1[101][:77]; r2[4001] = r2[4001][:50]
    else:                                              # the scan stops at the empty line, in both forms
        r1[30000:30000] = [b""]; r2[30000:30000] = [b""]
    open(os.path.join(work, "reads_1.fa"), "wb").write(b"\n".join(r1) + (b"" if shape == "no_final_newline" else b""))
    open(os.path.join(work, "reads_2.fa"), "wb").write(b"\n".join(r2))
    ps, serial = front(work, AGX_CLI_SERIAL="1")
    pt, got = front(work, AGX_CLI_THREADS="6", AGX_CLI_FAST_MIN="0")
    assert pt.returncode == ps.returncode and pt.stdout == ps.stdout
    assert got == serial and len(serial) == 3
    if shape == "unequal_mates":
        assert b">50\n" + r1[101] + b"\n>50\n" + r2[101][:77] + b"\n" in serial["_reads.fa"]


def _formalize_genome_model(chroms, p):
    """formalizeGenome, AG:3347-3418, base by base: files of the units and tmp/_genome.fa."""
    units, whole, unit = {}, [], 0
    for s in chroms:
        out = [">0\n"]; whole.append(">%d\n" % unit)
        step, q = len(s) // p, 1
        for c, ch in enumerate(s):
            out.append(ch); whole.append(ch)
            cut = step != 0 and (c + 1) % step == 0 and q < p
            if (c + 1) % 60 == 0 or c == len(s) - 1 or cut:
                out.append("\n"); whole.append("\n")
            if c != len(s) - 1 and cut:
                units[unit] = "".join(out); unit += 1; q += 1
                out = [">0\n"]; whole.append(">%d\n" % unit)
        units[unit] = "".join(out); unit += 1
    return units, "".join(whole)


@pytest.mark.parametrize("part", [1, 2, 3, 7])
def test_formalize_genome_parts(built, run_dir, tmp_path, part):
    """r05 writes the unit sequences a line at a time; the rule (a newline after every 60th base of the chromosome, at a part's end and at the chromosome's end; a cut on the last
    base opens no unit) against a base-by-base model of AG:3382-3413, for lengths around the multiples of 60 and of the part count."""
    import random
    import shutil
    work = str(tmp_path / "w")
    shutil.copytree(run_dir, work)
    rnd = random.Random(part)
    chroms = ["".join(rnd.choice("ACGT") for _ in range(n)) for n in (1, 59, 60, 61, 119, 120, 121, 360, 361, 420 * part, 420 * part + 1, 1237)]
    with open(os.path.join(work, "genome.fa"), "w") as f:
        for i, s in enumerate(chroms):
            f.write(">c%d\n" % i + "".join(s[j:j + 70] + "\n" for j in range(0, len(s), 70)))
    args = list(ARGS) + ["--part", str(part)]
    shutil.rmtree(os.path.join(work, "tmp"), ignore_errors=True)
    e = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"))
    subprocess.run([CLI] + args, cwd=work, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    units, whole = _formalize_genome_model(chroms, part)
    assert open(os.path.join(work, "tmp", "_genome.fa")).read() == whole
    for u, text in units.items():
        assert open(os.path.join(work, "tmp", "_genome.%d.fa" % u)).read() == text, u
    assert not os.path.exists(os.path.join(work, "tmp", "_genome.%d.fa" % len(units)))


def _distribute_model(sam, units):
    """distributeAlignments + parseBT, AG:3520-3579, line by line: '@' lines dropped, the scan ends at the first empty line, a '*' anywhere in RNAME = unplaced,
    the unit = atoi of RNAME's first nine bytes"""
    import re
    out = {u: [] for u in range(units)}
    for ln in sam.split(b"\n"):
        if ln.startswith(b"@"):
            continue
        if not ln or ln[:1] == b"\0":
            break
        f = ln.split(b"\t")
        rname = f[2] if len(f) > 2 else b""
        if b"*" in rname:
            continue
        m = re.match(rb"\s*[+-]?\d+", rname[:9])
        u = int(m.group(0)) if m else 0
        if 0 <= u < units:
            out[u].append(ln + b"\n")
    return {u: b"".join(v) for u, v in out.items()}


@pytest.mark.parametrize("shape", ["plain", "ragged", "empty_line", "no_final_newline"])
def test_alignments_distributed_on_threads(built, run_dir, tmp_path, shape):
    """r05: distributeAlignments (AG:3545-3579) on several threads (two passes over the mapped SAM, every unit's file written in place): the units' files against a line-by-line
    model of the reference's rule and against the one-thread form."""
    import shutil
    work = str(tmp_path / "w")
    shutil.copytree(run_dir, work)
    path = os.path.join(work, "stub", "reads_genome.sam")
    lines = open(path, "rb").read().split(b"\n")
    assert lines[-1] == b""
    lines = lines[:-1]
    if shape == "ragged":                                  # comment lines in the body, unplaced lines, units that do not exist, a line without an RNAME column, a signed and a padded unit
        lines[1000:1000] = [b"@CO\tin the middle", b"x\t4\t*\t0\t0\t*", b"y\t0\t7.1\t5\t42\t10M", b"z\t0", b"w\t0\t+1.9\t3\t42\t5M", b"v\t0\t 1.2\t3\t42\t5M", b"q\t0\t1*\t3\t42\t5M", b"p\t0\t-1.0\t1\t1\t1M"]
    elif shape == "empty_line":
        lines[40000:40000] = [b""]
    data = b"\n".join(lines) + (b"" if shape == "no_final_newline" else b"\n")
    open(path, "wb").write(data)
    want = _distribute_model(data, 2)
    assert all(len(v) > 1000 for v in want.values())

    def units_files(**env):
        shutil.rmtree(os.path.join(work, "tmp"), ignore_errors=True)
        e = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"), AGX_CLI_TIMING="1", **env)
        p = subprocess.run([CLI] + ARGS, cwd=work, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert b"(0) Alignment finished" in p.stdout
        return p, {u: open(os.path.join(work, "tmp", "_reads_genome.%d.bowtie" % u), "rb").read() for u in range(2)}
    ps, serial = units_files(AGX_CLI_SERIAL="1")
    assert b"units on" not in ps.stderr and serial == want
    for threads in ("2", "5", "16"):
        pt, got = units_files(AGX_CLI_THREADS=threads, AGX_CLI_FAST_MIN="0")
        assert b"to 2 units on %s threads" % threads.encode() in pt.stderr, pt.stderr[-300:]
        assert got == want, "threads=%s" % threads
