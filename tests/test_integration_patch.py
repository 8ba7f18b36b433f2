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

PATCH = '''		agx_params p = { (uint32_t) k, (uint32_t) insertVariation, (uint32_t) coverage, 0 /* BATCH = 1000000 */, 0 /* device */, 0 };
		agx_result r; char err[512];
		int rc = agx_run_unit(&p, "tmp", chromosomeID, /*write_files=*/1, &r, err, sizeof err);
		if(rc != AGX_OK) { string m = err; cout << m.substr(0, m.find(" (")) << endl; exit(-1); }
		agx_result_free(&r);
'''


@pytest.mark.skipif(not os.path.exists(REF_SRC), reason="needs the reference's source (build container only)")
def test_the_integration_patch_compiles_and_links(built, tmp_path):
    import aligngraph_amd as A
    if not os.path.exists(A.LIB_PATH):
        from aligngraph_amd import build as B
        B.build()
    src = open(REF_SRC, encoding="latin-1").read()
    first, last = "\t\tloadGenome(genome, chromosomeID);\n", '\t\tcout << "(5) Contigs scaffolded" << endl;\n'
    a = src.rindex(first)
    b = src.index(last, a)
    assert src.count(first) == 1 or a > src.index("int main(")
    patched = '#include "agx.h"\n' + src[:a] + PATCH + src[b:]
    cpp = tmp_path / "AlignGraph_patched.cpp"
    cpp.write_text(patched, encoding="latin-1")
    exe = str(tmp_path / "AlignGraph_patched")
    libdir = os.path.dirname(A.LIB_PATH)
    subprocess.check_call(["g++", "-w", "-o", exe, str(cpp), "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lagx", "-lpthread", "-Wl,-rpath," + libdir])
    if A.device_count() > 0:
        return                                         # (on a GPU box the patched binary would simply run: tests/test_cli.py covers that path with AlignGraph_amd)
    run = H.synth(str(tmp_path / "run"), seed=3, chroms="6000", pairs=500, coverage=2, sam_seq=0)
    env = dict(os.environ, PATH=H.STUBS + os.pathsep + os.environ.get("PATH", ""))
    p = subprocess.run([exe, "--resume"], cwd=run, env=env, stdout=subprocess.PIPE, timeout=120)
    assert p.returncode == 255 and b"CHROMOSOME 0:" in p.stdout and b"no HIP device" in p.stdout and b"(5) Contigs scaffolded" not in p.stdout
