"""The indexed stand-in aligners of tests/e2e_stubs/fast/ (what whole runs at configuration size use, tests/tools/f2_at_size.py) give the bytes of the
Python stand-ins the golden runs were captured with: the second half of the `masb` and `default` runs with either set, file by file."""
import os
import re
import shutil
import subprocess

import pytest

from test_cli import CLI, STUBS, Case, FINALS, cli  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = os.path.join(STUBS, "fast")
BIN = os.path.join(ROOT, "build", "agx_stub_align")


@pytest.fixture(scope="module")
def stub_bin():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(FAST, "agx_stub_align.cpp")
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", BIN + ".tmp%d" % os.getpid(), src])
        os.replace(BIN + ".tmp%d" % os.getpid(), BIN)
    return BIN


@pytest.mark.parametrize("name", ["masb", "default"])
def test_indexed_stand_ins_answer_like_the_python_ones(cli, stub_bin, name, tmp_path):  # noqa: F811
    c = Case(name, tmp_path)
    p = c.run(cli, c.args)                                                  # first half: tmp/ up to the unit loop (either set of stand-ins replays the same files there)
    assert b"(0) Alignment finished" in p.stdout
    for fn in os.listdir(os.path.join(c.exp, "tmp")):                       # the reference's unit outputs
        if re.fullmatch(r"_(initial|pre_extended|extended)_contigs\.\d+\.fa", fn):
            shutil.copy(os.path.join(c.exp, "tmp", fn), os.path.join(c.work, "tmp", fn))
    with open(os.path.join(c.work, "tmp", "_checkpoint.txt"), "w") as f:
        f.write("0\n%d\n" % c.units)
    slow = c.work + ".python"
    shutil.copytree(c.work, slow)
    env = dict(os.environ, AGX_STUB_DIR=os.path.join(c.work, "stub"), AGX_STUB_BIN=stub_bin, AGX_STUB_THREADS="3")
    for work, stubs in ((c.work, FAST), (slow, STUBS)):
        p = subprocess.run([cli, "--resume"], cwd=work, env=dict(env, PATH=stubs + os.pathsep + os.environ["PATH"]), stdout=subprocess.PIPE)
        assert p.returncode == 0 and b"FINISHED SUCCESSFULLY" in p.stdout, p.stdout[-400:]
    for fn in FINALS:
        if os.path.exists(os.path.join(c.exp, fn)):
            assert c.got(fn) == c.expected(fn), fn
    seen = 0
    for fn in sorted(os.listdir(os.path.join(slow, "tmp"))):                # the aligners' own files, one by one
        if re.search(r"_short_initial_contigs_extended_contigs.*\.psl$|_contigs_genome\.psl$|_contigs\.bowtie$", fn):
            assert c.got("tmp/" + fn) == open(os.path.join(slow, "tmp", fn), "rb").read(), fn
            seen += 1
    assert seen >= (5 if name == "masb" else 1), seen
