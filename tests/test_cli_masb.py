"""Misassembly removal's read coverage (row f2; loadReadAlignment AG:3940-3984 -> masb_read_coverage in aligngraph_amd/csrc/agx_cli.cpp) on several threads: the mapped SAM
file cut into pieces at pair boundaries, +1/-1 at the two ends of every span, one running sum per record.  `corrected_*.fa` must be what the REAL reference binary writes from
the same tmp/ and the same SAM — with soft clips, insertions, deletions, POS 0 (an empty span in the reference's unsigned comparison), unplaced pairs and comment lines between
pairs (which shift the pairing: the one-thread form takes those files) — and the same for every thread count."""
import os
import re
import shutil
import subprocess

import pytest

import harness as H
from test_cli import CLI, STUBS, Case, cli  # noqa: F401


def _second_half(cli_path, work, stubs, **env):
    """corrected_*.fa of a --resume behind the unit loop, and (AlignGraph_amd only: AGX_CLI_MASB_COV) the per-base counts of the two contig sets"""
    e = dict(os.environ, PATH=stubs + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"), AGX_CLI_MASB_COV=os.path.join(work, "cov"), **env)
    for which in ("extended", "remaining"):
        if os.path.exists(os.path.join(work, "cov.%s.bin" % which)):
            os.remove(os.path.join(work, "cov.%s.bin" % which))
    p = subprocess.run([cli_path, "--resume"], cwd=work, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"FINISHED SUCCESSFULLY" in p.stdout, p.stdout[-400:] + p.stderr[-400:]
    out = {f: open(os.path.join(work, f), "rb").read() for f in ("corrected_e.fa", "corrected_r.fa")}
    out["stderr"] = p.stderr
    for which in ("extended", "remaining"):
        if os.path.exists(os.path.join(work, "cov.%s.bin" % which)):
            out[which] = open(os.path.join(work, "cov.%s.bin" % which), "rb").read()
    return out


def _coverage_model(sam, fasta):
    """the counts by the reference's rules (AG:181-285, 3940-3984), one base at a time: records in the order of the FASTA file"""
    import numpy as np
    names, sizes = [], []
    for ln in fasta.split(b"\n"):
        if ln.startswith(b">"):
            names.append(ln[1:]); sizes.append(0)
        elif names:
            sizes[-1] += len(ln)
    cov = [np.zeros(n, np.int32) for n in sizes]
    lines = [ln for ln in sam.split(b"\n")]
    i = 0

    def span(ln):
        f = (ln.split(b"\t") + [b""] * 6)[:6]
        if b"*" in f[2]:
            return None
        tid = int(f[2].split(b".")[0]) if b"." in f[2] else 0
        total = ins = dele = 0
        for n, op in re.findall(rb"(\d*)([A-Z*])", f[5]):
            n = int(n or 0)
            if op == b"I":
                ins += n; total += n
            elif op == b"D":
                dele += n
            elif op in (b"M", b"S"):
                total += n
        ts = (int(f[3]) - 1) & 0xFFFFFFFF
        return tid, ts, (ts + total + dele - ins) & 0xFFFFFFFF
    while i < len(lines) and lines[i]:
        if lines[i].startswith(b"@"):
            i += 1
            continue
        x, y = span(lines[i]), span(lines[i + 1])
        i += 2
        if x is None or y is None:
            continue
        for tid, ts, te in (x, y):
            if ts < te:
                cov[tid][ts:te] += 1
    return b"".join(c.tobytes() for c in cov)


def _ragged(sam, comments):
    """every aligned pair's CIGARs and POS rewritten, in rotation, to forms whose spans stay inside their record"""
    lines = sam.split(b"\n")
    length = {}
    out, turn, i = [], 0, 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith(b"@"):
            m = re.match(rb"@SQ\tSN:(\S+)\tLN:(\d+)", ln)
            if m:
                length[m.group(1)] = int(m.group(2))
            out.append(ln); i += 1
            continue
        if not ln:
            i += 1
            continue
        a, b = ln.split(b"\t"), lines[i + 1].split(b"\t")
        i += 2
        if a[2] != b"*":
            room = min(length[a[2]] - int(a[3]), length[b[2]] - int(b[3]))
            form = turn % 6
            turn += 1
            if form == 1 and room > 120:
                a[5], b[5] = b"40M5D60M", b"97M3D3M"
            elif form == 2:
                a[5], b[5] = b"10S80M10S", b"50M3I47M"
            elif form == 3:
                a[3] = b"0"                                   # start = 0 - 1, unsigned: nothing is counted for this mate
            elif form == 4:
                a[5], b[5] = b"*", b"30M2I10M2D58M"
            elif form == 5 and room > 120:
                a[5] = b"1M1I1M1D97M"
        if comments and len(out) % 7 == 0:
            out.append(b"@CO\tbetween two pairs")
        out.append(b"\t".join(a)); out.append(b"\t".join(b))
    return b"\n".join(out) + b"\n"


@pytest.mark.parametrize("comments", [False, True])
def test_read_coverage_on_threads_is_the_references(cli, tmp_path, comments):  # noqa: F811
    c = Case("masb", tmp_path)
    p = c.run(cli, c.args)
    assert b"(0) Alignment finished" in p.stdout
    for fn in os.listdir(os.path.join(c.exp, "tmp")):
        if re.fullmatch(r"_(initial|pre_extended|extended)_contigs\.\d+\.fa", fn):
            shutil.copy(os.path.join(c.exp, "tmp", fn), os.path.join(c.work, "tmp", fn))
    with open(os.path.join(c.work, "tmp", "_checkpoint.txt"), "w") as f:
        f.write("0\n%d\n" % c.units)
    plain = _second_half(cli, c.work, STUBS)                  # the stand-in aligner's own SAM files: what the golden run saw
    assert plain["corrected_e.fa"] == c.expected("corrected_e.fa") and plain["corrected_r.fa"] == c.expected("corrected_r.fa")
    model = {}
    # a stand-in bowtie2 that replays the rewritten SAM files for the two contig sets
    stubs = str(tmp_path / "stubs")
    shutil.copytree(STUBS, stubs, ignore=shutil.ignore_patterns("fast"))
    samdir = str(tmp_path / "sam")
    os.makedirs(samdir)
    for which in ("extended", "remaining"):
        sam = open(os.path.join(c.work, "tmp", "_reads_%s_contigs.bowtie" % which), "rb").read()
        ragged = _ragged(sam, comments)
        open(os.path.join(samdir, "_%s_contigs.sam" % which), "wb").write(ragged)
        model[which] = _coverage_model(ragged, open(os.path.join(c.work, "tmp", "_%s_contigs.fa" % which), "rb").read())
        assert model[which] != plain[which] and len(model[which]) == len(plain[which])
    with open(os.path.join(stubs, "bowtie2"), "w") as f:
        f.write('#!/bin/sh\nx=""\nwhile [ $# -gt 0 ]; do case "$1" in -h) exit 0 ;; -x) x="$2"; shift ;; esac; shift; done\n'
                'case "$x" in *_contigs) exec cat "%s/$(basename "$x").sam" ;; esac\nexec cat "$AGX_STUB_DIR/reads_genome.sam"\n' % samdir)
    os.chmod(os.path.join(stubs, "bowtie2"), 0o755)
    serial = _second_half(cli, c.work, stubs, AGX_CLI_SERIAL="1")
    assert serial["extended"] == model["extended"] and serial["remaining"] == model["remaining"]
    for threads in ("2", "5", "16"):
        got = _second_half(cli, c.work, stubs, AGX_CLI_THREADS=threads, AGX_CLI_FAST_MIN="0", AGX_CLI_TIMING="1")
        assert (b"of SAM on %s threads" % threads.encode() in got["stderr"]) == (not comments), got["stderr"][-300:]      # (comment lines between pairs: one thread)
        for key in ("corrected_e.fa", "corrected_r.fa", "extended", "remaining"):
            assert got[key] == serial[key], "threads=%s: %s" % (threads, key)
    if H.have_reference():
        ref = c.work + ".ref"
        shutil.copytree(c.work, ref)
        theirs = _second_half(H.REF_O2, ref, stubs)
        assert theirs["corrected_e.fa"] == serial["corrected_e.fa"] and theirs["corrected_r.fa"] == serial["corrected_r.fa"]


@pytest.mark.parametrize("fault", ["bad_cigar", "odd_lines"])
def test_read_coverage_stops_where_the_reference_stops(cli, tmp_path, fault):  # noqa: F811
    """A CIGAR character parseBOWTIE does not know ("unknown character: X", AG:263-267) and a pair without its second line ("BROKEN BOWTIE FILE!", AG:3961-3965) end the run —
    with the message of the FIRST such line in file order, whatever thread met one first."""
    c = Case("masb", tmp_path)
    p = c.run(cli, c.args)
    assert b"(0) Alignment finished" in p.stdout
    for fn in os.listdir(os.path.join(c.exp, "tmp")):
        if re.fullmatch(r"_(initial|pre_extended|extended)_contigs\.\d+\.fa", fn):
            shutil.copy(os.path.join(c.exp, "tmp", fn), os.path.join(c.work, "tmp", fn))
    with open(os.path.join(c.work, "tmp", "_checkpoint.txt"), "w") as f:
        f.write("0\n%d\n" % c.units)
    _second_half(cli, c.work, STUBS)
    stubs = str(tmp_path / "stubs")
    shutil.copytree(STUBS, stubs, ignore=shutil.ignore_patterns("fast"))
    samdir = str(tmp_path / "sam")
    os.makedirs(samdir)
    for which in ("extended", "remaining"):
        lines = open(os.path.join(c.work, "tmp", "_reads_%s_contigs.bowtie" % which), "rb").read().split(b"\n")[:-1]
        placed = [i for i, ln in enumerate(lines) if not ln.startswith(b"@") and ln.split(b"\t")[2] != b"*"]
        if fault == "bad_cigar":                           # two of them, far apart: the message names the first one's character
            for at, ch in ((placed[len(placed) // 3], b"Q"), (placed[2 * len(placed) // 3], b"Z")):
                f = lines[at].split(b"\t"); f[5] = b"50M1" + ch + b"49M"; lines[at] = b"\t".join(f)
        else:
            lines = lines[:-1]
        open(os.path.join(samdir, "_%s_contigs.sam" % which), "wb").write(b"\n".join(lines) + b"\n")
    with open(os.path.join(stubs, "bowtie2"), "w") as f:
        f.write('#!/bin/sh\nx=""\nwhile [ $# -gt 0 ]; do case "$1" in -h) exit 0 ;; -x) x="$2"; shift ;; esac; shift; done\n'
                'case "$x" in *_contigs) exec cat "%s/$(basename "$x").sam" ;; esac\nexec cat "$AGX_STUB_DIR/reads_genome.sam"\n' % samdir)
    os.chmod(os.path.join(stubs, "bowtie2"), 0o755)

    def ending(exe, work, **env):
        e = dict(os.environ, PATH=stubs + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(work, "stub"), **env)
        p = subprocess.run([exe, "--resume"], cwd=work, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return p.returncode, p.stdout.split(b"\n")[-2]
    want = (255, b"unknown character: Q" if fault == "bad_cigar" else b"BROKEN BOWTIE FILE!")
    assert ending(cli, c.work, AGX_CLI_SERIAL="1") == want
    for threads in ("2", "7"):
        assert ending(cli, c.work, AGX_CLI_THREADS=threads, AGX_CLI_FAST_MIN="0") == want, threads
    if H.have_reference():
        ref = c.work + ".ref"
        shutil.copytree(c.work, ref)
        assert ending(H.REF_O2, ref) == want
