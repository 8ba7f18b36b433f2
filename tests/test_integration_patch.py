"""INTEGRATION.md §1 as a test: the patch a maintainer would apply to the reference's unit loop (AG:4765-4783) — include agx.h, replace the
five calls by agx_run_unit — applied to a scratch copy of the reference's own source, compiled and LINKED against aligngraph_amd/libagx.so.
Runs where /root/reference exists (the build container); the copy lives under the test's tmp_path and is never committed.  Without a GPU
the patched binary must then stop at its first unit with the library's no-device error (the boundary reports, it never falls back)."""
import os
import subprocess

import pytest

import harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/AlignGraph/AlignGraph.cpp"

@pytest.mark.skipif(not os.path.exists(REF_SRC), reason="needs the reference's source (build container only)")
def test_the_integration_patch_compiles_and_links(built, tmp_path):
    import aligngraph_amd as A
    if not os.path.exists(A.LIB_PATH):
        from aligngraph_amd import build as B
        B.build()
    patched = H.patch_reference_source(open(REF_SRC, encoding="latin-1").read())
    cpp = tmp_path / "AlignGraph_patched.cpp"
    cpp.write_text(patched, encoding="latin-1")
    exe = str(tmp_path / "AlignGraph_patched")
    libdir = os.path.dirname(A.LIB_PATH)
    subprocess.check_call(["g++", "-w", "-o", exe, str(cpp), "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lagx", "-lpthread", "-Wl,-rpath," + libdir])
    if A.device_count() > 0:
        return                                         # (on a GPU box the patched binary runs a unit: tests/test_gpu_patched_reference.py)
    run = H.synth(str(tmp_path / "run"), seed=3, chroms="6000", pairs=500, coverage=2, sam_seq=0)
    env = dict(os.environ, PATH=H.STUBS + os.pathsep + os.environ.get("PATH", ""))
    p = subprocess.run([exe, "--resume"], cwd=run, env=env, stdout=subprocess.PIPE, timeout=120)
    assert p.returncode == 255 and b"CHROMOSOME 0:" in p.stdout and b"no HIP device" in p.stdout and b"(5) Contigs scaffolded" not in p.stdout
