"""Test-side drivers for the CPU oracle and (when built) the real reference binary.

TEST INFRASTRUCTURE ONLY — imported by tests/, tests/golden/make_golden.py, bench.py's cpu_baseline
leg and __graft_entry__.smoke(); never by aligngraph_amd/.

* ``synth(out, **kw)``           tools/agx_data.synth (seeded synthetic tmp/ directory; re-exported)
* ``run_oracle(tmp, unit, ...)`` call oracle/liboracle.so (this repo's restatement) through ctypes
* ``run_reference(run_dir)``     replay a prepared run directory through oracle/_ref/AlignGraph_ref*
                                 with ``--resume`` (AG:4748-4760) and aligner stubs on PATH
"""
import ctypes
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SYNTH = os.path.join(ROOT, "build", "agx_synth")
LIBORACLE = os.path.join(HERE, "liboracle.so")
REF_O0 = os.path.join(HERE, "_ref", "AlignGraph_ref")
REF_O2 = os.path.join(HERE, "_ref", "AlignGraph_ref_O2")
STUBS = os.path.join(HERE, "ref_stubs")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import agx_data  # noqa: E402


REF_SRC = "/root/reference/AlignGraph/AlignGraph.cpp"
REF_PATCHED = os.path.join(HERE, "_ref", "AlignGraph_patched")
# what replaces the five calls of the reference's unit loop (AG:4768-4776)
INTEGRATION_PATCH = '''		agx_params p = { (uint32_t) k, (uint32_t) insertVariation, (uint32_t) coverage, 0 /* BATCH = 1000000 */, 0 /* device */, 0 };
		agx_result r; char err[512];
		int rc = agx_run_unit(&p, "tmp", chromosomeID, /*write_files=*/1, &r, err, sizeof err);
		if(rc != AGX_OK) { string m = err; cout << m.substr(0, m.find(" (")) << endl; exit(-1); }
		agx_result_free(&r);
'''


def patch_reference_source(src):
    """The reference's source text with INTEGRATION.md §1 applied: include agx.h, the five calls replaced by agx_run_unit."""
    first, last = "\t\tloadGenome(genome, chromosomeID);\n", '\t\tcout << "(5) Contigs scaffolded" << endl;\n'
    a = src.rindex(first)
    b = src.index(last, a)
    assert src.count(first) == 1 or a > src.index("int main(")
    return '#include "agx.h"\n' + src[:a] + INTEGRATION_PATCH + src[b:]


def build_patched_reference(out=REF_PATCHED, workdir=None):
    """Where the reference's source is present (the build container) and libagx.so is built: the reference WITH the INTEGRATION.md patch, linked against
    libagx.so, as oracle/_ref/AlignGraph_patched — a git-ignored artefact like the reference binaries beside it (it rides to the GPU box with them; the
    patched source text only ever exists in a scratch directory).  tests/test_gpu_patched_reference.py runs a unit through it on the GPU."""
    import tempfile
    lib = os.path.join(ROOT, "aligngraph_amd", "libagx.so")
    if not os.path.exists(REF_SRC) or not os.path.exists(lib):
        return None
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(REF_SRC), os.path.getmtime(os.path.join(ROOT, "include", "agx.h")), os.path.getmtime(__file__)):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with tempfile.TemporaryDirectory(dir=workdir) as d:
        cpp = os.path.join(d, "AlignGraph_patched.cpp")
        with open(cpp, "w", encoding="latin-1") as f:
            f.write(patch_reference_source(open(REF_SRC, encoding="latin-1").read()))
        subprocess.check_call(["g++", "-w", "-O2", "-o", out, cpp, "-I" + os.path.join(ROOT, "include"), "-L" + os.path.dirname(lib), "-lagx", "-lpthread",
                               "-Wl,-rpath,$ORIGIN/../../aligngraph_amd"])
    return out


def build():
    """Compile the oracle, the generator and (if /root/reference exists) oracle/_ref — the reference as it is, and with the INTEGRATION.md patch."""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    agx_data.build()
    agx_data.build_bin()          # (the generator with the staged-pairs mode: the big GPU tests use it from several threads — built here, once)
    build_patched_reference()


synth = agx_data.synth            # the generator lives in tools/ (it is not part of the checker); kept here for the tests' convenience
read_meta = agx_data.read_meta


class _Result(ctypes.Structure):
    _fields_ = [
        ("initial_contigs", ctypes.c_void_p), ("initial_len", ctypes.c_size_t),
        ("pre_extended", ctypes.c_void_p), ("pre_len", ctypes.c_size_t),
        ("extended", ctypes.c_void_p), ("extended_len", ctypes.c_size_t),
        ("error", ctypes.c_char * 256),
        ("n_pos", ctypes.c_uint32), ("n_nodes", ctypes.c_uint32), ("n_edges", ctypes.c_uint32),
        ("node_start", ctypes.POINTER(ctypes.c_uint32)), ("node_key", ctypes.POINTER(ctypes.c_uint32)),
        ("node_cnt", ctypes.POINTER(ctypes.c_int32)), ("node_slen", ctypes.POINTER(ctypes.c_uint32)),
        ("edge_start", ctypes.POINTER(ctypes.c_uint32)), ("edge_dst", ctypes.POINTER(ctypes.c_uint32)),
    ]


_lib = None


def _oracle():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBORACLE):
            build()
        _lib = ctypes.CDLL(LIBORACLE)
        _lib.agx_oracle_run_unit.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_long, ctypes.c_int, ctypes.POINTER(_Result)]
        _lib.agx_oracle_run_unit.restype = ctypes.c_int
        _lib.agx_oracle_free.argtypes = [ctypes.POINTER(_Result)]
    return _lib


class OracleError(RuntimeError):
    pass


def run_oracle(tmp_dir, unit, k=5, insert_variation=50, coverage=20, batch=1000000, graph=False):
    """Returns dict(initial=bytes, pre=bytes, extended=bytes[, graph=dict of numpy arrays])."""
    lib = _oracle()
    r = _Result()
    rc = lib.agx_oracle_run_unit(tmp_dir.encode(), unit, k, insert_variation, coverage, batch, 1 if graph else 0, ctypes.byref(r))
    if rc != 0:
        msg = r.error.decode()
        lib.agx_oracle_free(ctypes.byref(r))
        raise OracleError(msg)
    out = {
        "initial": ctypes.string_at(r.initial_contigs, r.initial_len),
        "pre": ctypes.string_at(r.pre_extended, r.pre_len),
        "extended": ctypes.string_at(r.extended, r.extended_len),
    }
    if graph:
        import numpy as np

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        out["graph"] = {
            "n_pos": r.n_pos, "n_nodes": r.n_nodes, "n_edges": r.n_edges,
            "node_start": arr(r.node_start, r.n_pos + 1, "uint32"),
            "node_key": arr(r.node_key, r.n_nodes * 6, "uint32").reshape(-1, 6),
            "node_cnt": arr(r.node_cnt, r.n_nodes * 6, "int32").reshape(-1, 6),
            "node_slen": arr(r.node_slen, r.n_nodes, "uint32"),
            "edge_start": arr(r.edge_start, r.n_nodes + 1, "uint32"),
            "edge_dst": arr(r.edge_dst, r.n_edges, "uint32"),
        }
    lib.agx_oracle_free(ctypes.byref(r))
    return out


def have_reference(opt=True):
    return os.path.exists(REF_O2 if opt else REF_O0)


def run_reference(run_dir, opt=True, keep=False, exe=None, suffix=".ref"):
    """Copy run_dir to a scratch sibling, replay it with the real reference through --resume, and return
    (outputs, stage_seconds): outputs[u] = dict(initial=, pre=, extended=) for every unit; stage_seconds is the
    wall time between the reference's own "(0)"/"RESUMED" and last "(5) Contigs scaffolded" stdout markers."""
    exe = exe or (REF_O2 if opt else REF_O0)
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    work = run_dir.rstrip("/") + suffix
    if os.path.exists(work):
        shutil.rmtree(work)
    shutil.copytree(run_dir, work)
    env = dict(os.environ)
    env["PATH"] = STUBS + os.pathsep + env.get("PATH", "")
    p = subprocess.Popen([exe, "--resume"], cwd=work, env=env, stdout=subprocess.PIPE, bufsize=0)
    t0 = t1 = None
    buf = b""
    while True:
        chunk = p.stdout.read(4096)
        now = time.perf_counter()
        if not chunk:
            break
        buf += chunk
        if t0 is None and b"RESUMED SUCCESSFULLY" in buf:
            t0 = now
        if b"(5) Contigs scaffolded" in chunk:
            t1 = now
    rc = p.wait()
    if rc != 0 or b"FINISHED SUCCESSFULLY" not in buf:
        raise RuntimeError("reference failed (rc=%d): %s" % (rc, buf[-400:].decode(errors="replace")))
    outs = []
    u = 0
    while os.path.exists(os.path.join(work, "tmp", "_extended_contigs.%d.fa" % u)):
        o = {}
        for key, name in (("initial", "_initial_contigs"), ("pre", "_pre_extended_contigs"), ("extended", "_extended_contigs")):
            with open(os.path.join(work, "tmp", "%s.%d.fa" % (name, u)), "rb") as f:
                o[key] = f.read()
        outs.append(o)
        u += 1
    if not keep:
        shutil.rmtree(work)
    return outs, (t1 - t0 if t0 is not None and t1 is not None else None)
