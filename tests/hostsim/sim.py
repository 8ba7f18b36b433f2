"""ctypes driver of tests/hostsim/libagx_hostsim.so (TEST-ONLY serial executor of the engine kernels)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.environ.get("AGX_HOSTSIM_LIB", os.path.join(HERE, "libagx_hostsim.so"))      # (override: a variant build, e.g. -DAGX_WALK_CHECK, made by hand)
SRC = [os.path.join(HERE, "agx_hostsim.cpp"), os.path.join(ROOT, "aligngraph_amd", "csrc", "agx_host.cpp"),
       os.path.join(ROOT, "aligngraph_amd", "csrc", "agx_walk.cpp"), os.path.join(ROOT, "aligngraph_amd", "csrc", "agx_load.cpp")]
DEPS = SRC + [os.path.join(ROOT, "aligngraph_amd", "csrc", h) for h in ("agx_core.h", "agx_host.h", "agx_parse.h")]


def build():
    if "AGX_HOSTSIM_LIB" in os.environ:
        return
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in DEPS):
        return
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-fPIC", "-shared", "-pthread", "-o", LIB] + SRC)


class _Result(ctypes.Structure):
    _fields_ = [
        ("initial_contigs", ctypes.c_void_p), ("initial_len", ctypes.c_size_t),
        ("pre_extended", ctypes.c_void_p), ("pre_len", ctypes.c_size_t),
        ("extended", ctypes.c_void_p), ("extended_len", ctypes.c_size_t),
        ("error", ctypes.c_char * 256),
        ("n_pos", ctypes.c_uint32), ("n_nodes", ctypes.c_uint32), ("n_edges", ctypes.c_uint32), ("n_big_tiles", ctypes.c_int32),
        ("node_start", ctypes.POINTER(ctypes.c_uint32)), ("node_key", ctypes.POINTER(ctypes.c_uint32)),
        ("node_cnt", ctypes.POINTER(ctypes.c_int32)), ("node_slen", ctypes.POINTER(ctypes.c_uint32)),
        ("edge_start", ctypes.POINTER(ctypes.c_uint32)), ("edge_dst", ctypes.POINTER(ctypes.c_uint32)),
        ("n_walk_ids", ctypes.c_uint64), ("n_special", ctypes.c_uint64), ("n_fetched", ctypes.c_uint64),
    ]


_lib = None


class SimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%d: %s" % (code, msg))
        self.code = code
        self.msg = msg


def run(tmp_dir, unit, k=5, insert_variation=50, coverage=20, batch=1000000, maxv_first=0, graph=False):
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.agx_hostsim_run_unit.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_Result)]
        _lib.agx_hostsim_free.argtypes = [ctypes.POINTER(_Result)]
    r = _Result()
    rc = _lib.agx_hostsim_run_unit(tmp_dir.encode(), unit, k, insert_variation, coverage, batch, maxv_first, 1 if graph else 0, ctypes.byref(r))
    if rc != 0:
        msg = r.error.decode()
        _lib.agx_hostsim_free(ctypes.byref(r))
        raise SimError(rc, msg)
    out = {"initial": ctypes.string_at(r.initial_contigs, r.initial_len), "pre": ctypes.string_at(r.pre_extended, r.pre_len),
           "extended": ctypes.string_at(r.extended, r.extended_len), "n_big_tiles": r.n_big_tiles,
           "n_walk_ids": r.n_walk_ids, "n_special": r.n_special, "n_fetched": r.n_fetched}
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
    _lib.agx_hostsim_free(ctypes.byref(r))
    return out


def compare_loaders(tmp_dir, unit, k=5, batch=1000000, threads=4):
    """Fast loaders (agx_load.cpp) against the general loaders on one unit's text files: returns the bit mask of fast loaders that declined
    (1 contigs, 2 read alignments); raises SimError when the two disagree."""
    build()
    lib = ctypes.CDLL(LIB)
    lib.agx_hostsim_compare_loaders.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    msg = ctypes.create_string_buffer(1024)
    rc = lib.agx_hostsim_compare_loaders(tmp_dir.encode(), unit, k, batch, threads, msg, 1024)
    if rc < 0 or rc >= 64:
        raise SimError(rc, msg.value.decode())
    return rc


def compare_staged(tmp_dir, unit, k=5, batch=1000000, threads=4):
    """tmp/_agx_pairs.<unit>.bin (staged read alignments) against the general loader + staging over the unit's text files; raises SimError when they differ."""
    build()
    lib = ctypes.CDLL(LIB)
    lib.agx_hostsim_compare_staged.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    msg = ctypes.create_string_buffer(1024)
    rc = lib.agx_hostsim_compare_staged(tmp_dir.encode(), unit, k, batch, threads, msg, 1024)
    if rc != 0:
        raise SimError(rc, msg.value.decode())
    return rc


def _lib_now():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def cgroup_quota(proc_cgroup, sys_root):
    """agx::cgroup_cpu_quota on a made-up control-group tree: CPUs allowed (rounded up), 0 = no quota."""
    L = _lib_now()
    L.agx_hostsim_cgroup_quota.restype = ctypes.c_uint
    L.agx_hostsim_cgroup_quota.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    return int(L.agx_hostsim_cgroup_quota(str(proc_cgroup).encode(), str(sys_root).encode()))


def usable_cpus():
    L = _lib_now()
    L.agx_hostsim_usable_cpus.restype = ctypes.c_uint
    return int(L.agx_hostsim_usable_cpus())


def rowdiff_roundtrip(seed, n_pos, n_rows, stride, maxlen, mut_permille=20, dirty_tail=False, threads=4):
    """build_row_diffs + the device's decoder on made-up rows; returns (units, explicit rows); raises SimError on a mismatch."""
    L = _lib_now()
    f = L.agx_hostsim_rowdiff_roundtrip
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_uint] * 6 + [ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_char_p, ctypes.c_size_t]
    nu, ne = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    msg = ctypes.create_string_buffer(512)
    rc = f(seed, n_pos, n_rows, stride, maxlen, mut_permille, int(bool(dirty_tail)), threads, ctypes.byref(nu), ctypes.byref(ne), msg, 512)
    if rc != 0:
        raise SimError(rc, msg.value.decode())
    return nu.value, ne.value


def rowdiff_unit(tmp_dir, unit, k=5, batch=1000000, threads=4):
    """The rows of a unit's real files through build_row_diffs and the device's decoder: {rows, explicit, bytes_2bit, bytes_upload}; None if the unit sequence does not pack."""
    L = _lib_now()
    f = L.agx_hostsim_rowdiff_unit
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_char_p, ctypes.c_size_t]
    out = (ctypes.c_ulonglong * 4)()
    msg = ctypes.create_string_buffer(512)
    rc = f(str(tmp_dir).encode(), unit, k, batch, threads, out, msg, 512)
    if rc == -2:
        return None
    if rc != 0:
        raise SimError(rc, msg.value.decode())
    return {"rows": out[0], "explicit": out[1], "bytes_2bit": out[2], "bytes_upload": out[3]}
