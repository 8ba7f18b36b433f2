"""aligngraph_amd — ctypes binding of libagx.so (include/agx.h), the MI355X engine for AlignGraph's per-unit
graph build + extend path (reference seam: AlignGraph/AlignGraph.cpp:4765-4783).

The library is the product; this module only loads it, mirrors the C structs and raises on error codes.
There is no CPU fallback: without a HIP device `Unit(...)` raises AgxError(AGX_E_NOGPU).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AGX_LIB_PATH", os.path.join(_HERE, "libagx.so"))   # override only for kernel A/B experiments

AGX_OK, AGX_E_IO, AGX_E_FORMAT, AGX_E_UNSUPPORTED, AGX_E_ALIGNMENT, AGX_E_DEVICE, AGX_E_ARG, AGX_E_OVERFLOW, AGX_E_NOGPU = 0, -1, -2, -3, -4, -5, -6, -7, -8
AGX_FLAG_KEEP_COUNTS = 1
AGX_FLAG_SPARSE_MIN = 2
AGX_FLAG_TIME_SECTIONS = 4
AGX_FLAG_ONE_SHOT = 8

# every symbol include/agx.h declares (tests check that the built library exports all of them)
EXPORTS = [
    "agx_version", "agx_device_count", "agx_device_memory", "agx_selftest_scan", "agx_unit_create", "agx_unit_destroy", "agx_unit_error", "agx_unit_set_reference",
    "agx_unit_set_contig_threads", "agx_unit_push_pairs", "agx_unit_load_files", "agx_unit_upload", "agx_unit_build", "agx_unit_download",
    "agx_unit_finish", "agx_result_free", "agx_unit_stats", "agx_unit_graph", "agx_graph_free", "agx_run_unit",
    "agx_reads_open", "agx_reads_close", "agx_unit_load_files_shared", "agx_run_unit_shared",
    "agx_unit_stage", "agx_unit_release", "agx_pool_trim", "agx_unit_cache_build", "agx_unit_cache_save", "agx_unit_hbm_needed",
    "agx_unit_trim",
]


class Params(ctypes.Structure):
    _fields_ = [("k", ctypes.c_uint32), ("insert_variation", ctypes.c_uint32), ("coverage", ctypes.c_uint32),
                ("batch", ctypes.c_uint32), ("device", ctypes.c_int32), ("flags", ctypes.c_uint32)]


class Run(ctypes.Structure):
    _fields_ = [("q", ctypes.c_uint32), ("t", ctypes.c_uint32), ("n", ctypes.c_uint32)]


class Hit(ctypes.Structure):
    _fields_ = [("slot1", ctypes.c_uint32), ("pos1", ctypes.c_uint32), ("pos2", ctypes.c_uint32), ("runs1", ctypes.c_uint32),
                ("runs2", ctypes.c_uint32), ("nruns1", ctypes.c_uint16), ("nruns2", ctypes.c_uint16), ("len", ctypes.c_uint16),
                ("rev1", ctypes.c_uint8), ("rev2", ctypes.c_uint8), ("back", ctypes.c_uint8), ("pad", ctypes.c_uint8 * 3)]


class ContiMer(ctypes.Structure):
    _fields_ = [("cid", ctypes.c_uint32), ("coff", ctypes.c_uint32), ("next_off", ctypes.c_uint32), ("next_item", ctypes.c_uint32),
                ("nuc", ctypes.c_char), ("pad", ctypes.c_char * 3)]


class PairBatch(ctypes.Structure):
    _fields_ = [("hits", ctypes.POINTER(Hit)), ("n_hits", ctypes.c_uint64), ("runs", ctypes.POINTER(Run)), ("n_runs", ctypes.c_uint64),
                ("bases", ctypes.c_char_p), ("stride", ctypes.c_uint32), ("n_slots", ctypes.c_uint32)]


class Result(ctypes.Structure):
    _fields_ = [("initial_contigs", ctypes.c_void_p), ("initial_len", ctypes.c_size_t), ("pre_extended", ctypes.c_void_p),
                ("pre_len", ctypes.c_size_t), ("extended", ctypes.c_void_p), ("extended_len", ctypes.c_size_t)]


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("n_pos", "n_ref", "n_hits", "n_runs", "n_nodes", "n_tiles", "n_tile_entries", "n_big_tiles",
                                                "n_edge_overflow", "pairs_in_file", "sam_line_pairs")] + \
               [(n, ctypes.c_double) for n in ("ms_parse", "ms_thread", "ms_upload", "ms_prep", "ms_bin", "ms_node_sweep", "ms_node_big",
                                                "ms_edge_sweep", "ms_compact", "ms_download", "ms_walk")] + \
               [("node_sweep_launches", ctypes.c_uint32), ("edge_sweep_launches", ctypes.c_uint32)] + \
               [(n, ctypes.c_uint64) for n in ("n_walk_ids", "n_special", "n_fetched", "download_bytes")] + \
               [(n, ctypes.c_double) for n in ("ms_edge_fast", "ms_edge_slow")] + [("n_mid_tiles", ctypes.c_uint64), ("ms_build_span", ctypes.c_double)] + \
               [(n, ctypes.c_double) for n in ("ms_stage", "ms_upload_dev")] + \
               [(n, ctypes.c_uint64) for n in ("upload_bytes", "device_bytes", "pinned_bytes_cached", "device_bytes_cached", "n_spilled")] + \
               [("build_attempts", ctypes.c_uint32), ("from_cache", ctypes.c_uint32), ("dense_lists", ctypes.c_uint32), ("rows_by_reference", ctypes.c_uint32)]


class Graph(ctypes.Structure):
    _fields_ = [("n_pos", ctypes.c_uint32), ("n_nodes", ctypes.c_uint32), ("n_edges", ctypes.c_uint32),
                ("node_start", ctypes.POINTER(ctypes.c_uint32)), ("node_key", ctypes.POINTER(ctypes.c_uint32)),
                ("node_cnt", ctypes.POINTER(ctypes.c_int32)), ("node_slen", ctypes.POINTER(ctypes.c_uint32)),
                ("edge_start", ctypes.POINTER(ctypes.c_uint32)), ("edge_dst", ctypes.POINTER(ctypes.c_uint32))]


class AgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("agx error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


_lib = None


def lib():
    """Loads libagx.so (build it first with `python -m aligngraph_amd.build` or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s is missing: run `python aligngraph_amd/build.py` (needs hipcc)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.agx_version.restype = ctypes.c_char_p
        L.agx_device_count.restype = ctypes.c_int
        L.agx_unit_create.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(ctypes.c_void_p)]
        L.agx_unit_destroy.argtypes = [ctypes.c_void_p]
        L.agx_unit_destroy.restype = None
        L.agx_unit_error.argtypes = [ctypes.c_void_p]
        L.agx_unit_error.restype = ctypes.c_char_p
        L.agx_unit_set_reference.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32]
        L.agx_unit_set_contig_threads.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
                                                  ctypes.POINTER(ContiMer), ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
        L.agx_unit_push_pairs.argtypes = [ctypes.c_void_p, ctypes.POINTER(PairBatch)]
        L.agx_device_memory.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        L.agx_selftest_scan.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
        L.agx_unit_load_files.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.agx_reads_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_size_t]
        L.agx_reads_close.argtypes = [ctypes.c_void_p]
        L.agx_reads_close.restype = None
        L.agx_unit_load_files_shared.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
        L.agx_unit_cache_build.argtypes = [ctypes.POINTER(Params), ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        L.agx_unit_cache_save.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.agx_unit_hbm_needed.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.agx_unit_trim.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.agx_pool_trim.argtypes = [ctypes.c_int]
        L.agx_pool_trim.restype = None
        for f in ("agx_unit_upload", "agx_unit_build", "agx_unit_download", "agx_unit_stage", "agx_unit_release"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.agx_unit_finish.argtypes = [ctypes.c_void_p, ctypes.POINTER(Result)]
        L.agx_result_free.argtypes = [ctypes.POINTER(Result)]
        L.agx_result_free.restype = None
        L.agx_unit_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(Stats)]
        L.agx_unit_graph.argtypes = [ctypes.c_void_p, ctypes.POINTER(Graph)]
        L.agx_graph_free.argtypes = [ctypes.POINTER(Graph)]
        L.agx_graph_free.restype = None
        L.agx_run_unit.argtypes = [ctypes.POINTER(Params), ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Result), ctypes.c_char_p, ctypes.c_size_t]
        _lib = L
    return _lib


def device_memory(device=0):
    """(free, total) bytes of HBM on one device (hipMemGetInfo)."""
    f, t = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = lib().agx_device_memory(device, ctypes.byref(f), ctypes.byref(t))
    if rc != AGX_OK:
        raise AgxError(rc, "agx_device_memory failed")
    return f.value, t.value


def device_count():
    return lib().agx_device_count()


def pool_trim(device=0, host=True, retire_host=False):
    """Frees what the library's memory caches hold: HBM blocks of `device` (>= 0) and (host=True) the pinned host blocks.  retire_host: the
    cached pinned blocks are only taken out of circulation (later allocations map and register fresh memory); a pool_trim(host=True) unmaps them."""
    if device >= 0:
        lib().agx_pool_trim(device)
    if retire_host:
        lib().agx_pool_trim(-2)
    elif host:
        lib().agx_pool_trim(-1)


def _take(res):
    out = {"initial": ctypes.string_at(res.initial_contigs, res.initial_len) if res.initial_contigs else b"",
           "pre": ctypes.string_at(res.pre_extended, res.pre_len) if res.pre_extended else b"",
           "extended": ctypes.string_at(res.extended, res.extended_len) if res.extended else b""}
    lib().agx_result_free(ctypes.byref(res))
    return out


class ResultViews:
    """The three output buffers of agx_unit_finish without a Python copy: view(name) is a numpy uint8 array over the C buffer (valid until
    free()), bytes(name) a copy.  A loop that runs many units from several threads should not hold the interpreter lock for megabytes of
    byte-string copies per unit (bench.py)."""
    _FIELDS = {"initial": ("initial_contigs", "initial_len"), "pre": ("pre_extended", "pre_len"), "extended": ("extended", "extended_len")}

    def __init__(self, res):
        self._res = res

    def view(self, name):
        import numpy as np
        p, n = (getattr(self._res, f) for f in self._FIELDS[name])
        if not p or not n:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n,))

    def bytes(self, name):
        p, n = (getattr(self._res, f) for f in self._FIELDS[name])
        return ctypes.string_at(p, n) if p else b""

    def free(self):
        if self._res is not None:
            lib().agx_result_free(ctypes.byref(self._res))
            self._res = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Reads:
    """tmp/_reads.fa mapped and indexed once (agx_reads); pass it to Unit.load_files of every unit of the run."""

    def __init__(self, path):
        self._h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = lib().agx_reads_open(path.encode(), ctypes.byref(self._h), err, 512)
        if rc != AGX_OK:
            self._h = ctypes.c_void_p()
            raise AgxError(rc, err.value.decode(errors="replace"))

    def close(self):
        if self._h:
            lib().agx_reads_close(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Unit:
    """One reference unit (chromosome or --part slice): the body of the reference's unit loop, AG:4765-4783."""

    def __init__(self, k=5, insert_variation=50, coverage=20, batch=0, device=0, keep_counts=False, flags=0):
        self._h = ctypes.c_void_p()
        self.params = Params(k, insert_variation, coverage, batch, device, (AGX_FLAG_KEEP_COUNTS if keep_counts else 0) | flags)
        rc = lib().agx_unit_create(ctypes.byref(self.params), ctypes.byref(self._h))
        if rc != AGX_OK:
            self._h = ctypes.c_void_p()
            raise AgxError(rc, "no HIP device (libagx has no CPU path)" if rc == AGX_E_NOGPU else "agx_unit_create failed")

    def _check(self, rc):
        if rc != AGX_OK:
            raise AgxError(rc, lib().agx_unit_error(self._h).decode(errors="replace"))

    def close(self):
        if self._h:
            lib().agx_unit_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_files(self, tmp_dir, unit, reads=None):
        """reads: an optional Reads (tmp/_reads.fa opened once for all units of a run)."""
        self._check(lib().agx_unit_load_files_shared(self._h, tmp_dir.encode(), unit, reads._h if reads is not None else None))

    def cache_save(self, tmp_dir, unit):
        """Writes tmp_dir/_agx_unit.<unit>.bin from this unit (just loaded from the text files of tmp_dir, unit)."""
        self._check(lib().agx_unit_cache_save(self._h, tmp_dir.encode(), unit))

    def stage(self):
        self._check(lib().agx_unit_stage(self._h))

    def hbm_needed(self):
        """HBM the upload of this (loaded) unit will take at its first-guess capacities: what AlignGraph_amd admits units to a device by."""
        v = ctypes.c_uint64(0)
        self._check(lib().agx_unit_hbm_needed(self._h, ctypes.byref(v)))
        return v.value

    def upload(self):
        self._check(lib().agx_unit_upload(self._h))

    def trim(self):
        """After download(): the part of the unit's HBM that the host walk cannot ask for goes back to the device's memory region (agx_unit_trim); returns the bytes given back."""
        v = ctypes.c_uint64(0)
        self._check(lib().agx_unit_trim(self._h, ctypes.byref(v)))
        return v.value

    def release(self):
        """HBM and download buffers back to the library's caches; the staged inputs stay (upload again = a new unit)."""
        self._check(lib().agx_unit_release(self._h))

    def build(self):
        self._check(lib().agx_unit_build(self._h))

    def download(self):
        self._check(lib().agx_unit_download(self._h))

    def finish(self):
        r = Result()
        self._check(lib().agx_unit_finish(self._h, ctypes.byref(r)))
        return _take(r)

    def finish_views(self):
        r = Result()
        self._check(lib().agx_unit_finish(self._h, ctypes.byref(r)))
        return ResultViews(r)

    def stats(self):
        s = Stats()
        self._check(lib().agx_unit_stats(self._h, ctypes.byref(s)))
        return {n: getattr(s, n) for n, _ in Stats._fields_}

    def graph(self):
        import numpy as np
        g = Graph()
        self._check(lib().agx_unit_graph(self._h, ctypes.byref(g)))

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        out = {"n_pos": g.n_pos, "n_nodes": g.n_nodes, "n_edges": g.n_edges,
               "node_start": arr(g.node_start, g.n_pos + 1, "uint32"), "node_key": arr(g.node_key, g.n_nodes * 6, "uint32").reshape(-1, 6),
               "node_cnt": arr(g.node_cnt, g.n_nodes * 6, "int32").reshape(-1, 6), "node_slen": arr(g.node_slen, g.n_nodes, "uint32"),
               "edge_start": arr(g.edge_start, g.n_nodes + 1, "uint32"), "edge_dst": arr(g.edge_dst, g.n_edges, "uint32")}
        lib().agx_graph_free(ctypes.byref(g))
        return out


def cache_build(tmp_dir, unit, batch=0, device=0, reads=None, k=5):
    """Writes tmp_dir/_agx_unit.<unit>.bin (the unit's staged arrays) from the five text files; load_files then takes it instead of the text
    (units of the same k and batch size only)."""
    p = Params(k, 50, 20, batch, device, 0)
    err = ctypes.create_string_buffer(512)
    rc = lib().agx_unit_cache_build(ctypes.byref(p), tmp_dir.encode(), unit, reads._h if reads is not None else None, err, 512)
    if rc != AGX_OK:
        raise AgxError(rc, err.value.decode(errors="replace"))


def run_unit(tmp_dir, unit, k=5, insert_variation=50, coverage=20, batch=0, device=0, write_files=False):
    """The five-call seam in one call (agx_run_unit)."""
    p = Params(k, insert_variation, coverage, batch, device, 0)
    r = Result()
    err = ctypes.create_string_buffer(512)
    rc = lib().agx_run_unit(ctypes.byref(p), tmp_dir.encode(), unit, 1 if write_files else 0, ctypes.byref(r), err, 512)
    if rc != AGX_OK:
        raise AgxError(rc, err.value.decode(errors="replace"))
    return _take(r)


def usable_cpus():
    """CPUs this process can keep busy: its affinity mask capped by the CPU quota of its control group (cgroup v2 cpu.max) — what the loaders size
    their thread teams by (agx_host.cpp: usable_cpus)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = max(1, min(n, -(-int(quota) // int(period))))
    except Exception:
        pass
    return n
