"""AlignGraph_amd (aligngraph_amd/csrc/agx_cli.cpp): the reference's command line around the engine, against the end-to-end golden
fixture tests/golden/e2e.tar.gz captured from the real reference binary (tests/golden/make_golden_e2e.py).

Without a GPU the unit loop cannot run, so the CPU tests check the two halves around it: (1) a fresh run must produce the reference's
tmp/ inputs byte for byte and then stop with the no-GPU message; (2) given the reference's per-unit outputs, `--resume` must produce the
reference's final files.  The -m gpu test runs the whole thing."""
import os
import re
import shutil
import subprocess
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "e2e_stubs")
CLI = os.environ.get("AGX_CLI_PATH", os.path.join(ROOT, "aligngraph_amd", "AlignGraph_amd"))      # (AGX_CLI_PATH: a sanitizer build of the same source)
FINALS = ("e.fa", "r.fa", "in.fa", "ex.fa", "corrected_e.fa", "corrected_r.fa")   # corrected_*: --misassemblyRemoval (AG:4224)


@pytest.fixture(scope="module")
def cli():
    from aligngraph_amd import build as B
    B.build()
    assert os.path.exists(CLI)
    return CLI


class Case:
    def __init__(self, name, dest):
        with tarfile.open(os.path.join(ROOT, "tests", "golden", "e2e.tar.gz")) as tar:
            tar.extractall(dest, members=[m for m in tar.getmembers() if m.name.startswith(name + "/")])
        self.dir = os.path.join(str(dest), name)
        self.work = os.path.join(self.dir, "in")
        self.exp = os.path.join(self.dir, "expected")
        self.args = open(os.path.join(self.dir, "args.txt")).read().split("\n")
        self.units = len([f for f in os.listdir(os.path.join(self.exp, "tmp")) if re.fullmatch(r"_extended_contigs\.\d+\.fa", f)])

    def run(self, cli, args):
        env = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], AGX_STUB_DIR=os.path.join(self.work, "stub"))
        return subprocess.run([cli] + args, cwd=self.work, env=env, stdout=subprocess.PIPE)

    def expected(self, rel):
        return open(os.path.join(self.exp, rel), "rb").read()

    def got(self, rel):
        return open(os.path.join(self.work, rel), "rb").read()


def strip_time(out):
    return re.sub(rb"for \d+ seconds \(\d+ seconds for alignment\)", b"for N seconds (N seconds for alignment)", out)


@pytest.mark.parametrize("name", ["default", "flags", "masb", "fastmap"])
def test_front_half_and_refinement_match_reference(cli, name, tmp_path):
    import aligngraph_amd as A
    if A.device_count() > 0:
        pytest.skip("a GPU is present: the full run is covered by test_full_run_matches_reference")
    c = Case(name, tmp_path)
    p = c.run(cli, c.args)
    assert p.returncode == 255 and b"(0) Alignment finished" in p.stdout and b"NO HIP DEVICE" in p.stdout     # loud, not a fallback
    for fn in os.listdir(os.path.join(c.exp, "tmp")):
        if fn.startswith(("_contigs.fa", "_chaff", "_genome", "_contigs_genome")):     # _contigs_genome.*: --fastMap's .delta and delta2psl's PSL
            assert c.got("tmp/" + fn) == c.expected("tmp/" + fn), fn
    assert c.got("tmp/_checkpoint.txt") == b"0\n"
    # hand over the reference's unit outputs and resume at the end of the unit loop
    for fn in os.listdir(os.path.join(c.exp, "tmp")):
        if re.fullmatch(r"_(initial|pre_extended|extended)_contigs\.\d+\.fa", fn):
            shutil.copy(os.path.join(c.exp, "tmp", fn), os.path.join(c.work, "tmp", fn))
    with open(os.path.join(c.work, "tmp", "_checkpoint.txt"), "a") as f:
        f.write("%d\n" % c.units)
    p = c.run(cli, ["--resume"])
    assert p.returncode == 0 and b"RESUMED SUCCESSFULLY :-)" in p.stdout and b"FINISHED SUCCESSFULLY" in p.stdout
    for fn in FINALS:
        if os.path.exists(os.path.join(c.exp, fn)):
            assert c.got(fn) == c.expected(fn), fn
    for fn in os.listdir(os.path.join(c.exp, "tmp")):
        if fn.startswith("_short_initial_contigs"):                                       # incl. the refinement alignment of --fastMap (.delta, .psl)
            if fn.endswith(".psl") and "--fastMap" not in c.args:
                continue
            assert c.got("tmp/" + fn) == c.expected("tmp/" + fn), fn


def test_usage_and_parameter_errors(cli, tmp_path):
    c = Case("default", tmp_path)
    p = c.run(cli, ["--read1", "reads_1.fa"])                      # required flags missing: usage, exit status 0 (AG:4726-4730)
    assert p.returncode == 0 and b"AlignGraph --read1 reads_1.fa --read2" in p.stdout
    p = c.run(cli, c.args + ["--kMer", "5", "--kMer", "6"])        # duplicate flag: usage, exit(-1)
    assert p.returncode == 255 and b"Options:" in p.stdout
    p = c.run(cli, c.args[:-1] + ["4x"])                            # integer that does not round-trip
    assert p.returncode == 255
    p = c.run(cli, ["--read1", "missing.fa"])
    assert p.returncode == 255 and b"CANNOT OPEN FILE!" in p.stdout
    p = c.run(cli, ["--resume", "--kMer", "5"])                     # --resume must be the only argument (AG:4627)
    assert p.returncode == 255
    p = c.run(cli, ["--bogus"])
    assert p.returncode == 255


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "flags", "masb", "fastmap"])
def test_full_run_matches_reference(cli, name, tmp_path):
    c = Case(name, tmp_path)
    p = c.run(cli, c.args)
    assert p.returncode == 0, p.stdout[-400:]
    assert strip_time(p.stdout) == strip_time(c.expected("stdout.txt"))
    for fn in FINALS:
        if os.path.exists(os.path.join(c.exp, fn)):
            assert c.got(fn) == c.expected(fn), fn
    for fn in os.listdir(os.path.join(c.exp, "tmp")):
        assert c.got("tmp/" + fn) == c.expected("tmp/" + fn), fn
    # --resume from the middle: drop the last unit's outputs, rewind the checkpoint, finish the run again
    last = c.units - 1
    for stem in ("_initial_contigs", "_pre_extended_contigs", "_extended_contigs"):
        os.remove(os.path.join(c.work, "tmp", "%s.%d.fa" % (stem, last)))
    with open(os.path.join(c.work, "tmp", "_checkpoint.txt"), "w") as f:
        f.write("0\n%d\n" % last)
    p = c.run(cli, ["--resume"])
    assert p.returncode == 0 and b"RESUMED SUCCESSFULLY :-)" in p.stdout
    for fn in ("e.fa", "r.fa"):
        assert c.got(fn) == c.expected(fn), fn
