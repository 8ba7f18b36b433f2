#!/usr/bin/env python3
"""bench.py — aligned reads/sec through graph build + extend (BASELINE.json metric) on N MI355X GPUs of one node.

What is timed is SURVEY §8(d)'s T_core of a whole job, the way the application runs it (AG:4765-4783: one unit after another, every unit
new data): a "step" takes every unit of the configuration from its packed arrays in host memory to its three output byte buffers in
host memory — upload (one HBM block per unit, PCIe copies), the unit's FIRST build (all kernels, capacities sized on the spot, a repeat
if one proves too small), download of the walk graph, the sequential host walk/join/scaffold — with the units pipelined against each other
on the device exactly as AlignGraph_amd pipelines them (a worker thread per unit in flight, largest unit first).  Every step starts from
nothing on the device and gets its OWN set of units, loaded from the unit caches before the clock starts and used once, as the application
uses a unit (AGX_FLAG_ONE_SHOT: the download lands in the pinned memory of the unit's dead staged inputs); --reupload times the same unit
objects in every step instead, whose downloads then need freshly pinned buffers (the library's pinned-host cache is emptied before every
step: --pool host-cold, the default; --pool cold also gives the HBM blocks back to the driver; see --help for why that is not the default).

    value = 2 * pairs of the configuration / seconds per step            (whole job, all GPUs)

Text parsing of the five per-unit files (T_unit = parse + T_core: the scope of the reference's stages (1)-(5), the like-for-like comparison) is measured
once while the inputs are loaded (a rank's units side by side, each on its share of the CPUs the process can keep busy) and reported beside it
(`t_unit_s`, `value_t_unit`, `load_ms_per_unit`), and so is loading the units from their binary caches.  `roofline` carries `frac` (bytes the dominant
kernel cannot avoid / its time / HBM peak), `frac_hbm` (counter traffic), `job_frac` (SURVEY 8(d)'s algorithmic bytes / T_core / peak)
and a `pcie` block; `cpu_baseline` the reference binary on one CPU and, under `parallel`, one process per usable CPU.

Configurations (--config; BASELINE.json `configs`, synthetic data of that shape from tools/agx_synth, seeded):
    cfg3 (default)  A. thaliana shape: 5 units of 30.4 / 19.7 / 23.5 / 18.6 / 27.0 Mb, 20 M 2x100 bp pairs, k=5      <- the north-star 1-GPU target
    cfg2            E. coli shape: one 4.6 Mb unit, 1 M pairs
    cfg4            human chr1 shape: 249 Mb --part 4 (4 units of 62 Mb), 60 M pairs (needs ~40 GB of scratch disk)
    cfg5s           whole-human shape SCALED 1/16: 24 units of 15.6 .. 3.6 Mb, 25 M 2x150 bp pairs (the shard shape of configs[4]; the default with --gpus N > 1)
    cfg5q           the same at 1/4: 24 units of 62 .. 12 Mb, 100 M 2x150 bp pairs (48 GB of text: what fits the GPU box's scratch disk; minutes to generate)
    custom          --chroms / --pairs / --part

Multi-GPU (driver: torch.distributed.run, one rank per GPU): units are the shard — assigned longest-first to the least
loaded rank (shard.assign_units), each rank runs its own list, and ONE gather of the extended-contig bytes to rank 0 ends the step
(RCCL: sizes by all_gather, then exact-size point-to-point sends to the root).  Total work is fixed: "scaling": "strong".  After the timed N-rank steps rank 0 runs the
same job alone on its GPU (`single_gpu_ms_same_config`, `speedup_vs_1gpu`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import datetime
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

HUMAN = (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
         133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415)      # GRCh38: chr1..22, X, Y
STAGED = {"cfg5"}      # configurations whose read alignments are generated staged (tmp/_agx_pairs.<u>.bin, tools/agx_synth.cpp --pairs-bin): as text they would not fit the box

CONFIGS = {
    # name: (chromosome lengths, --part, pairs, L, label)
    "cfg2": ([4600000], 1, 1000000, 100, "configs[1]: E. coli K-12 shape, one 4.6 Mb unit, 1M 2x100 bp pairs"),
    "cfg3": ([30427671, 19698289, 23459830, 18585056, 26975502], 1, 20000000, 100,
             "configs[2]: A. thaliana shape (TAIR10 chromosome lengths), 5 units / 119.1 Mb, 20M 2x100 bp pairs"),
    "cfg4": ([248956422], 4, 60000000, 100, "configs[3]: human chr1 shape (GRCh38 length), --part 4 = 4 units of 62 Mb, 60M 2x100 bp pairs"),
    # configs[4] (whole human, 24 units, 400M 2x150 bp pairs) does not fit a bench run: the same 24-unit shape at 1/16 of the lengths and pairs
    "cfg5s": ([c // 16 for c in (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
                                133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415)],
              1, 25000000, 150, "configs[4] SCALED 1/16: GRCh38 chromosome lengths / 16 (24 units, 193 Mb), 25M 2x150 bp pairs — the 8-GPU shard shape, not the configuration itself"),
    # configs[4] itself: 24 units of 249 .. 47 Mb, 3.1 Gb, 400 M pairs.  No SAM / reads text exists at this size (170 GB): the generator hands the alignments over staged,
    # 23 GB, made by the engine's own line parser, batch rules and staging (tests/test_staged_pairs.py); 64 GB of disk with the unit caches, ~4 minutes to generate.
    "cfg5": (list(HUMAN), 1, 400000000, 150, "configs[4]: whole human GRCh38 shape at FULL size (24 units / 3.1 Gb, 400M 2x150 bp pairs), read alignments handed over staged "
                                              "(tmp/_agx_pairs.<u>.bin) instead of as 170 GB of SAM + reads text"),
    # the same at 1/4: the largest scale whose text files (48 GB) fit the GPU box's scratch disk; units of 62 .. 12 Mb
    "cfg5q": ([c // 4 for c in (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
                               133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415)],
              1, 100000000, 150, "configs[4] SCALED 1/4: GRCh38 chromosome lengths / 4 (24 units, 772 Mb), 100M 2x150 bp pairs — the 8-GPU shard shape, not the configuration itself"),
}


def algorithmic_bytes(n_pairs, L, k, n_pos):
    """per pair (L-k)*40 + L + 32 bytes, plus 64 bytes per position for the extend pass — the bytes of the WHOLE path
    (node state, votes, edge reads, the extend pass).  Used for `job_frac` (against the job's time).  r01/r02 divided them by the node
    sweep's time — a quotient that passes 1, because one kernel is charged the whole path's bytes; no longer printed."""
    return n_pairs * ((L - k) * 40 + L + 32) + 64 * n_pos


def sweep_compulsory_bytes(st, L, k):
    """HBM bytes agx_k_node_sweep<0> cannot avoid for one unit, every array once: the tile records it walks (32 B per (tile, hit) list
    entry) and the tile offsets, one vote-code byte per arrival (an arrival = one read index of a left mate landing on a position: hits x (L-k+1)),
    the 16-byte conti-mer head of every position (read for the mate side of the arrivals there), and the node table it writes: 9 B per position
    (node_start, node_cnt, pos_succ, side ids) + 37 B per node (5-word key, base, flags, k-mer reference, 4 inline edge slots; r02-r05: 41 with a position word that r06 no longer writes).  The buckets
    themselves (the 32 B of node state per arrival that SURVEY §8(d) counts) live in LDS and never touch HBM."""
    return (32 * st["n_tile_entries"] + 4 * st["n_tiles"] + st["n_hits"] * max(1, L - k + 1) + 16 * (st["n_pos"] + 1) + 9 * st["n_pos"] + 37 * st["n_nodes"])


def pin_to_gpu_numa_node(torch, index):
    """Run this process (and the worker threads it starts later) on the CPU cores of the NUMA node the GPU hangs off: the host walk is
    bound by memory latency, and pinned buffers plus threads on the far socket cost it 30-50 %.  Returns a note for the JSON line."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return "gpu %s: no NUMA node reported" % bdf
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return "gpu %s: NUMA node %d has no allowed cpu" % (bdf, node)
        os.sched_setaffinity(0, cpus)
        return "gpu %s: pinned to NUMA node %d (%d cpus)" % (bdf, node, len(cpus))
    except Exception as e:                               # no sysfs, no such attribute: run unpinned
        return "not pinned (%s)" % e


def shutil_rm(path):
    import shutil
    shutil.rmtree(path, ignore_errors=True)


def in_use(run_dir):
    """Another bench.py is reading this input directory: it holds <dir>/.in_use.<its process id> (a marker of a process that is gone does not count, and is removed)."""
    busy = False
    try:
        names = [n for n in os.listdir(run_dir) if n.startswith(".in_use")]
    except OSError:
        return False
    for n in names:
        try:
            pid = int(n.rpartition(".")[2]) if n != ".in_use" else int(open(os.path.join(run_dir, n)).read().strip() or "0")
        except (OSError, ValueError):
            continue
        if pid == os.getpid():
            continue
        try:
            os.kill(pid, 0)
            busy = True
        except ProcessLookupError:
            try:
                os.remove(os.path.join(run_dir, n))
            except OSError:
                pass
        except OSError:
            busy = True
    return busy


def n50_of(fasta_bytes):
    """Eval-AlignGraph's rule, EV:372-380."""
    lens = sorted((len("".join(rec.split("\n")[1:])) for rec in fasta_bytes.decode().split(">")[1:]), reverse=True)
    tot, acc = sum(lens), 0
    for ln in lens:
        acc += ln
        if acc > tot / 2:
            return ln
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS) + ["custom"],
                    help="default: cfg3 on one GPU (the north-star 1-GPU configuration); with --gpus N > 1: cfg5s, the 24-unit whole-human shard shape (cfg3 has five units: "
                         "three of eight ranks would idle)")
    ap.add_argument("--same-config-steps", type=int, default=4, help="N > 1: after the timed N-rank steps rank 0 runs the SAME job alone on its GPU for this many steps "
                                                                      "(single_gpu_ms_same_config, speedup_vs_1gpu in the JSON line); 0 = skip")
    ap.add_argument("--host-gb", type=float, default=0.0, help="host memory a rank may hold in staged unit sets; more steps than fit re-upload the same unit objects (as --reupload).  "
                                                               "0 (default): 160 GB or 40 %% of the control group's memory limit, whichever is less, divided by the ranks on this host")
    ap.add_argument("--clean", action="store_true", help="delete the generated inputs at the end even for cfg5 (whose 64 GB are otherwise kept for the next run: N = 1, 2, 4, 8 back to back)")
    ap.add_argument("--chroms", default="4600000", help="custom: comma-separated chromosome lengths")
    ap.add_argument("--part", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1000000)
    ap.add_argument("--L", type=int, default=100)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--sam-seq", type=int, default=None, choices=[0, 1],
                    help="1: the generated SAM carries SEQ, QUAL and tag columns like bowtie2's output (AG:3609) — ~290 bytes per 2x100 bp line; 0: those columns are '*' "
                         "(~50 bytes per line).  Only T_unit (text parsing) depends on it.  Default: 1 for cfg2 / cfg3 / custom, 0 for the larger shapes (cfg4: 49 GB of text "
                         "with the columns, more than the box's scratch disk is known to hold); config.workload says which")
    ap.add_argument("--coverage", type=int, default=5, help="--coverage of the run (the reference's default 20 is above the graph depth of these read sets: SURVEY §8d)")
    ap.add_argument("--pool", default="host-cold", choices=["cold", "host-cold", "warm"],
                    help="what the library's memory caches hold when a step starts.  host-cold (default): the pinned-host cache is emptied before every "
                         "step, so a step that needs pinned memory (--reupload: its download buffers; one-shot units need none) maps, faults and registers "
                         "it like the first units of a fresh process; HBM blocks are "
                         "kept (a fresh hipMalloc costs 0.2-0.7 ms per unit, profiles/r02_membench.txt — but HBM that was just given back to the "
                         "driver can stall the next hipMalloc for seconds, profiles/r02_recycle.txt: an artefact of the loop, not of the application).  "
                         "cold: both caches emptied.  warm: both keep what earlier steps left (units 6, 7, .. of a long run)")
    ap.add_argument("--emulate-rank", default=None, metavar="R/N",
                    help="one GPU, one process: run only the units rank R of an N-rank job would get (shard.plan: longest-first onto the least loaded rank) on this process's "
                         "share of the host's CPUs (usable_cpus / N, by affinity; --emulate-cpus overrides) — the makespan of that rank of the N-GPU job without an N-GPU node.  "
                         "The line's value counts that rank's reads only; `emulated_rank` says what was run.  cfg5: only that rank's units are generated")
    ap.add_argument("--emulate-cpus", type=int, default=0, help="--emulate-rank: CPUs the emulated rank may use (default: usable_cpus // N, at least 1)")
    ap.add_argument("--no-stream", action="store_true", help="A/B: every unit's download completes before its walk begins (r05), also where nobody waits for its HBM")
    ap.add_argument("--reupload", action="store_true",
                    help="time the SAME unit objects in every step (their staged inputs are uploaded again and again, the downloads land in freshly pinned "
                         "buffers).  Default: every step gets its own set of units, loaded from the unit caches before the clock starts and used once "
                         "(AGX_FLAG_ONE_SHOT, what agx_run_unit and AlignGraph_amd do: a unit's download lands in the pinned memory of its dead inputs)")
    ap.add_argument("--inflight", type=int, default=0, help="units in flight per GPU = worker threads (0: all of the rank's units, at most 8)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=200000, help="pairs in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--workdir", default=os.environ.get("AGX_BENCH_DIR", "/tmp/agx_bench"))
    ap.add_argument("--keep", action="store_true", help="keep the generated inputs (a later run with the same configuration re-uses them)")
    args = ap.parse_args()

    import torch
    import aligngraph_amd as A
    import agx_data as D                        # synthetic inputs (tools/); nothing under oracle/ is imported outside the cpu_baseline leg
    from aligngraph_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    emu = None
    if args.emulate_rank:
        assert world == 1 and args.gpus == 1, "--emulate-rank runs one process on one GPU"
        er, _, en = args.emulate_rank.partition("/")
        emu = (int(er), int(en))
        assert 0 <= emu[0] < emu[1], "--emulate-rank R/N with 0 <= R < N"
        if args.config is None:
            args.config = "cfg5"                # the configuration the 8-GPU target is quoted on
    dist = None
    # AGX_BENCH_SHARE_GPU=1 is a plumbing test for boxes with fewer GPUs than ranks: ranks share devices and the gather runs over
    # gloo with CPU tensors (RCCL refuses two ranks on one device).  Never set by the driver; numbers from it are not bench results.
    share = os.environ.get("AGX_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=90))
        else:      # (the whole-human inputs take rank 0 five to seven minutes to generate while the others wait at a barrier)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=90))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    if not os.path.exists(A.LIB_PATH):
        if rank == 0:
            from aligngraph_amd import build as B
            B.build()
        if dist:
            dist.barrier()
    assert A.device_count() > local_rank, "bench.py needs a HIP device (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    gdev = torch.device("cpu") if share else torch.device("cuda", local_rank)
    numa_note = pin_to_gpu_numa_node(torch, local_rank) if os.environ.get("AGX_BENCH_NO_PIN") != "1" else "not pinned (AGX_BENCH_NO_PIN=1)"

    def bytes_limit(path, default):
        try:
            v = open(path).read().strip()
            return int(v) if v.isdigit() else default
        except OSError:
            return default
    mem_limit = bytes_limit("/sys/fs/cgroup/memory.max", 1 << 50)
    try:
        mem_limit = min(mem_limit, os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE"))
    except (ValueError, OSError):
        pass
    if args.config is None and os.environ.get("AGX_BENCH_CONFIG") in CONFIGS:
        args.config = os.environ["AGX_BENCH_CONFIG"]      # (a driver that wants ONE workload at every N sets it: the defaults differ between N = 1 and N > 1)
    if args.config is None:
        if world == 1:
            args.config = "cfg3"
        else:      # the whole-human configuration where the box can hold it (64 GB of disk, ~60 GB of host memory while it is generated), else its shape at 1/4.
            # Rank 0 decides for everybody: a rank that looked at the disk after rank 0 had begun to write the inputs could decide otherwise.
            import shutil
            choice = [None]
            if rank == 0:
                os.makedirs(args.workdir, exist_ok=True)
                have = os.path.exists(os.path.join(args.workdir, "cfg5_full", "synth_meta.txt"))
                choice[0] = "cfg5" if (have or shutil.disk_usage(args.workdir).free > 70e9) and mem_limit > 150e9 else "cfg5q"
            dist.broadcast_object_list(choice, src=0)
            args.config = choice[0]
            if args.config == "cfg5q":      # staged like cfg5 (17 GB instead of 49 GB of text, a minute to generate) and kept for the runs at the other rank counts
                os.environ["AGX_BENCH_STAGED"] = "1"
                args.keep = True
    if not args.host_gb:
        args.host_gb = min(160.0, 0.4 * mem_limit / 1e9) / max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if args.config == "custom":
        chroms, part, pairs, L = [int(x) for x in args.chroms.split(",")], args.part, args.pairs, args.L
        label = "custom: chromosomes %s, --part %d, %d 2x%d bp pairs" % (args.chroms, part, pairs, L)
    else:
        chroms, part, pairs, L, label = CONFIGS[args.config]
    k = args.k
    sam_seq = args.sam_seq if args.sam_seq is not None else (1 if args.config in ("cfg2", "cfg3", "custom") else 0)
    label += ", SAM lines %s" % ("with SEQ/QUAL/tag columns as bowtie2 writes them" if sam_seq else "WITHOUT SEQ/QUAL/tag columns ('*': a sixth of an aligner's bytes per line; T_unit is optimistic)")

    # ---- synthetic inputs (text files like the reference's tmp/): rank 0 generates, every rank parses its own units ----
    extra = {}
    for kv in filter(None, os.environ.get("AGX_BENCH_SYNTH", "").split(",")):      # experiments only (e.g. contig_overlap=0): changes the workload, the
        key, _, val = kv.partition("=")                                             # JSON line then says so in config.workload
        extra[key] = val
    staged = args.config in STAGED or os.environ.get("AGX_BENCH_STAGED") == "1"      # (the variable: any configuration through the staged hand-over — plumbing tests)
    # (the staged files depend on k — which mate is the left one — and on BATCH: both are part of a staged directory's name)
    only = {}
    if args.config in STAGED:
        run_name = "cfg5_full" if k == 5 else "cfg5_full_k%d" % k
        if emu and part == 1 and not os.path.exists(os.path.join(args.workdir, run_name, "synth_meta.txt")):
            # the emulated rank's units alone (tools/agx_synth --only-units: each exactly as long as in the whole job, with its share of the whole job's pairs and their numbering)
            only = {"only_units": ",".join(str(uu) for uu in sorted(shard.plan(chroms, emu[0], emu[1])))}
            run_name += "_r%dof%d" % emu
    elif staged:
        run_name = "staged_%s_p%d_k%d" % ("_".join(str(c) for c in chroms), pairs, k) + ("_x" if extra else "")
    else:
        run_name = "%s_p%d_k%d" % ("_".join(str(c) for c in chroms), pairs, k) + ("_x" if extra else "") + ("" if sam_seq else "_noseq")
    run = os.path.join(args.workdir, run_name)
    if args.config in STAGED:
        args.keep = not args.clean
    if staged:
        label = label[:label.index(", SAM lines ")] + ("" if args.config in STAGED else ", read alignments handed over staged (tmp/_agx_pairs.<u>.bin)")
    t0 = time.perf_counter()
    stamp = os.path.join(run, "synth_meta.txt")
    if rank == 0 and not os.path.exists(stamp):
        import shutil
        os.makedirs(args.workdir, exist_ok=True)
        wanted = (68e9 * (1.3 * sum(chroms[int(uu)] for uu in only["only_units"].split(",")) / sum(chroms) if only else 1.0)) if args.config in STAGED else pairs * (2.0 * (L + 12) + 2.0 * ((2 * L + 110) if sam_seq else 55) + 60) + 3.2 * sum(chroms)      # bytes about to be written (inputs + unit caches)
        if shutil.disk_usage(args.workdir).free < 1.15 * wanted:      # what OTHER configurations of this script left here (the whole-human inputs are kept between runs) goes first:
            for other in os.listdir(args.workdir):                    # only directories that carry the generator's stamp (tools/agx_synth.cpp writes synth_meta.txt last), are not
                d = os.path.join(args.workdir, other)                  # in use (a run holds <dir>/.in_use while it reads its inputs) — never anything else a --workdir may hold
                if other != os.path.basename(run) and os.path.isfile(os.path.join(d, "synth_meta.txt")) and not in_use(d):
                    shutil.rmtree(d, ignore_errors=True)
            if shutil.disk_usage(args.workdir).free < 1.05 * wanted:
                raise SystemExit("bench.py: %s has %.0f GB free, the inputs of %s need %.0f GB (nothing that is not this script's own was deleted)" %
                                 (args.workdir, shutil.disk_usage(args.workdir).free / 1e9, args.config, 1.05 * wanted / 1e9))
        if staged:
            D.synth(run, seed=1000, chroms=",".join(str(c) for c in chroms), part=part, pairs=pairs, L=L, k=k, coverage=args.coverage, sam_seq=0,
                    threads=min(32, max(4, A.usable_cpus())), pairs_bin=1, lean=1, **only, **extra)
        else:
            D.synth(run, seed=1000, chroms=",".join(str(c) for c in chroms), part=part, pairs=pairs, L=L, k=k, coverage=args.coverage, sam_seq=sam_seq,
                    threads=min(32, os.cpu_count() or 1), **extra)
    if dist:
        dist.barrier()
    t_gen = time.perf_counter() - t0
    if rank == 0:
        with open(os.path.join(run, ".in_use.%d" % os.getpid()), "w") as f:      # (a later run that needs the disk must not delete these inputs under this one; one marker per process: ADVICE r05)
            f.write(str(os.getpid()))
    tmp = os.path.join(run, "tmp")
    unit_len = D.read_meta(run)["unit_len"]
    n_units = len(unit_len)
    mine = shard.plan(unit_len, rank, world)                               # longest-first onto the least loaded rank, longest first within the rank
    emu_note = None
    if emu:
        mine = shard.plan(unit_len, emu[0], emu[1])
        have = sorted(os.sched_getaffinity(0))
        n_cpu = args.emulate_cpus or max(1, A.usable_cpus() // emu[1])
        os.sched_setaffinity(0, set(have[:max(1, min(n_cpu, len(have)))]))
        emu_note = {"rank": emu[0], "of": emu[1], "units": mine, "unit_positions": [unit_len[uu] for uu in mine], "cpus": len(os.sched_getaffinity(0)),
                    "note": "this rank's share of the N-rank job (shard.plan) alone on one GPU, on usable_cpus // N CPUs by affinity: ms_per_step is its makespan without the gather; value counts its reads only"}
    t1 = time.perf_counter()
    reads = A.Reads(os.path.join(tmp, "_reads.fa")) if mine and not staged else None      # tmp/_reads.fa mapped and indexed once for all units (agx_reads)
    t_index = time.perf_counter() - t1
    units, t_parse, t_stage, t_cached = {}, 0.0, 0.0, 0.0
    parse_threads = max(1, min(len(mine), 8))                              # units loaded side by side (each on its share of the cores: agx_host.cpp loader_threads)
    stage_s, load_ms, un_bytes, hbm_need = {}, {}, {}, {}

    def parse_unit(uu):
        un = A.Unit(k=k, insert_variation=50, coverage=args.coverage, device=local_rank)
        un.load_files(tmp, uu, reads=reads)                                # text -> packed arrays, staged in pinned memory (T_unit - T_core)
        st = un.stats()
        stage_s[uu] = st["ms_stage"] * 1e-3
        load_ms[uu] = {"contigs": round(st["ms_thread"], 1), "read_alignments": round(st["ms_parse"], 1), "rest": round(st["ms_stage"], 1)}
        hbm_need[uu] = un.hbm_needed()                                  # what the upload will take of the device: units are admitted to it by this (shard.run_job)
        un_bytes[uu] = hbm_need[uu] // 24 + 2 * st["n_pos"]             # what a staged one-shot unit holds in (pinned) host memory, generously: wire arrays + the download's landing area (cfg3's largest: 0.38 GB staged, 0.48 by this)
        return un

    os.environ["AGX_NO_CACHE"] = "1"
    cpus = A.usable_cpus()
    if "AGX_LOAD_THREADS" not in os.environ and mine:                      # the rank's units are loaded side by side: the CPUs this process can keep busy (affinity, cgroup quota) shared between them
        os.environ["AGX_LOAD_THREADS"] = str(max(2, min(32, cpus // min(len(mine), parse_threads))))
    t1 = time.perf_counter()
    if mine:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=parse_threads) as ex:
            parsed = list(ex.map(parse_unit, mine))
    else:
        parsed = []
    t_parse = time.perf_counter() - t1
    del os.environ["AGX_NO_CACHE"]
    t_stage = sum(stage_s.values())
    for uu, un in zip(mine, parsed):
        un.cache_save(tmp, uu)                                             # the unit's binary cache (what AlignGraph_amd writes when it distributes the alignments)
        un.close()
    if reads is not None:
        reads.close()
    # the units the timed steps run, loaded from their cache files: one set per step (every unit of the application is new data and is
    # uploaded once), or one set for all steps with --reupload
    per_set = sum(un_bytes.values()) if un_bytes else 0
    fit = max(1, int(args.host_gb * 1e9 // max(1, per_set)))
    reupload = args.reupload or fit < args.steps + args.warmup      # (a one-shot unit is used once: every step needs a set of its own in host memory)
    if reupload and not args.reupload and args.pool == "host-cold":
        args.pool = "warm"      # the sets do not fit the host: the same units are uploaded again, and their downloads land in the pinned buffers the step before left in the library's
                                # cache — what a long run's later units find.  (Emptying that cache first would bill every step for pinning 5 bytes per position again.)
    n_sets = 1 if reupload else args.steps + args.warmup
    unit_sets = []

    def load_unit(uu):
        un = A.Unit(k=k, insert_variation=50, coverage=args.coverage, device=local_rank, flags=0 if reupload else A.AGX_FLAG_ONE_SHOT)
        un.load_files(tmp, uu)
        assert un.stats()["from_cache"] == 1
        return un

    for si in range(n_sets):
        t1 = time.perf_counter()
        if mine:
            with ThreadPoolExecutor(max_workers=parse_threads) as ex:      # (as the application's units in flight load theirs)
                one = dict(zip(mine, ex.map(load_unit, mine)))
        else:
            one = {}
        if si == 0:
            t_cached = time.perf_counter() - t1
        unit_sets.append(one)
    units = dict(unit_sets[0])
    my_pairs = sum(units[uu].stats()["sam_line_pairs"] for uu in mine)

    inflight = args.inflight or min(8, max(1, len(mine)))
    hbm_budget = int(0.85 * min(A.device_memory(local_rank)))      # (free, total) before this process holds any of it: blocks recycled between units may be a sixteenth larger than the unit asked for, a build that has to grow a capacity takes more
    unit_stats, held, t_start = {}, {}, {}
    job = {"hbm_pressure": sum(hbm_need[uu] for uu in mine) > hbm_budget}      # the units of this rank do not fit its device side by side: each gives HBM back after its download (agx_unit_trim) instead of streaming it into its walk

    def run_unit(uu, release=None):
        """One unit from its staged packed arrays to its output bytes: (upload: start_unit) -> first build -> download -> walk -> release.  Runs on one of
        shard.run_job's worker threads (ctypes releases the interpreter lock, so the walks of several units run on several cores while
        libagx queues their kernel chains on the device's build streams)."""
        un = units[uu]
        t_a = t_start[uu]
        un.build()                         # hit prep, binning, node sweep (+ edges), edge passes, walk preparation: the unit's first build
        t_b = time.perf_counter()
        if job["hbm_pressure"] or args.no_stream:
            un.download()                  # walk graph -> pinned host memory, all of it before the walk ...
            if release is not None and not reupload:
                release(un.trim())         # ... because three quarters of the unit's HBM go back to the device now: the next unit is admitted while this one is walked (one-shot units only: a trimmed unit is uploaded again before another build)
        # else (r06): the rank's units all fit its device at once, nobody waits for this one's HBM: agx_unit_finish STREAMS the download — position windows from the front — and the
        # walk begins on what has landed
        res = un.finish_views()            # (download,) host walk/join/scaffold; the outputs stay in C memory until they are packed for the gather
        st = un.stats()
        st["s_upload_build"], st["s_total"] = t_b - t_a, time.perf_counter() - t_a
        un.release()                       # HBM and download buffers back to the library
        unit_stats[uu] = st
        old, held[uu] = held.get(uu), res  # the outputs stay where agx_unit_finish left them (C memory): a view, no interpreter-lock-held copy
        if old is not None:
            old.free()
        return res.view("extended")

    run_unit.takes_release = True

    def start_unit(uu):
        """Called by shard.run_job when a worker takes the unit, in plan order: queues the upload (one HBM block, asynchronous PCIe copies,
        conti-mer heads, vote codes) and returns — the uploads of a device run one after the other in this order."""
        t_start[uu] = time.perf_counter()
        units[uu].upload()

    step_no = [0]

    def run_job():
        """One step: this rank's units (longest first) through run_unit on `inflight` worker threads, then the path's only exchange — one
        gather of the extended contigs to rank 0 (aligngraph_amd/shard.py: the function the world_size-2 gloo test drives)."""
        if args.pool == "cold":
            A.pool_trim(local_rank, host=True)
        elif args.pool == "host-cold":
            A.pool_trim(-1, retire_host=True)          # (the previous steps' buffers are unmapped after the timed region)
        units.clear(); units.update(unit_sets[step_no[0] % n_sets]); step_no[0] += 1
        return shard.run_job(unit_len, rank, world, run_unit, dist, gdev, inflight=inflight, start_unit=start_unit, hbm_need=hbm_need, hbm_budget=hbm_budget, plan_as=emu)

    for _ in range(args.warmup):
        run_job()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    def thread_cpu():                                   # AGX_BENCH_THREAD_CPU=1 (development aid): CPU seconds of every thread of this process, by thread name
        out = {}
        for tid in os.listdir("/proc/self/task"):
            try:
                with open("/proc/self/task/%s/stat" % tid) as f:
                    st = f.read()
                name = st[st.index("(") + 1:st.rindex(")")]
                fld = st[st.rindex(")") + 2:].split()
                out[int(tid)] = (name, (int(fld[11]) + int(fld[12])) / os.sysconf("SC_CLK_TCK"))
            except (OSError, ValueError):
                pass
        return out
    thr0 = thread_cpu() if os.environ.get("AGX_BENCH_THREAD_CPU") else None
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    gathered = None
    for _ in range(args.steps):
        gathered = run_job()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    cpu_s_per_step = (time.process_time() - cpu0) / max(1, args.steps)      # every thread of this process (walkers, helpers, the runtime's), user + system
    if thr0 is not None and rank == 0:
        by_name = {}
        for tid, (name, sec) in thread_cpu().items():
            d = sec - thr0.get(tid, (name, 0.0))[1]
            n, tot = by_name.get(name, (0, 0.0))
            by_name[name] = (n + 1, tot + d)
        by_name["(threads that ended: the job's unit threads)"] = (0, cpu_s_per_step * max(1, args.steps) - sum(t for _, t in by_name.values()))
        for name, (n, tot) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
            print("[bench] threads %-16s x%-3d %8.1f ms of CPU per step" % (name, n, 1e3 * tot / max(1, args.steps)), file=sys.stderr)
    my_sam_bytes = sum(os.path.getsize(os.path.join(tmp, ("_agx_pairs.%d.bin" if staged else "_reads_genome.%d.bowtie") % uu)) for uu in mine)
    tot = torch.tensor([elapsed, t_parse, t_stage, float(my_pairs), t_cached, float(my_sam_bytes)], dtype=torch.float64, device=gdev)
    if dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed, t_parse_max, t_cached_max = float(mx[0].item()), float(mx[1].item()), float(mx[4].item())
    else:
        t_parse_max, t_cached_max = t_parse, t_cached
    sam_pairs_total = int(tot[3].item())
    sam_bytes_total = int(tot[5].item())

    A.pool_trim(-1, host=True)
    # ---- N > 1: the SAME job on ONE GPU (rank 0 alone, all units), so that the speed-up does not rest on comparing different driver runs ----
    single_ms = None
    if dist:
        dist.barrier()
        if rank == 0 and args.same_config_steps > 0:
            everything = shard.plan(unit_len, 0, 1)
            ms = []
            for it in range(args.same_config_steps + 1):                 # (the first one is a warm-up)
                one = {}
                with ThreadPoolExecutor(max_workers=parse_threads) as ex:
                    one = dict(zip(everything, ex.map(load_unit, everything)))
                units.clear(); units.update(one)
                hbm_all = {uu: un.hbm_needed() for uu, un in one.items()}
                job["hbm_pressure"] = sum(hbm_all.values()) > hbm_budget
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                shard.run_job(unit_len, 0, 1, run_unit, None, gdev, inflight=min(8, len(everything)), start_unit=start_unit, hbm_need=hbm_all, hbm_budget=hbm_budget)
                torch.cuda.synchronize()
                if it:
                    ms.append(1e3 * (time.perf_counter() - t1))
                for un in one.values():
                    un.close()
            single_ms = sum(ms) / len(ms)
        dist.barrier()
    # ---- after the timed region, rank 0: kernel sections of the largest unit (exclusive builds with section events) ----
    kern, big_stats = {}, None
    if rank == 0 and mine:
        big = max(mine, key=lambda uu: (unit_len[uu], -uu))
        with A.Unit(k=k, insert_variation=50, coverage=args.coverage, device=local_rank, flags=A.AGX_FLAG_TIME_SECTIONS) as ub:
            ub.load_files(tmp, big)
            keys = ("ms_upload_dev", "ms_prep", "ms_bin", "ms_node_sweep", "ms_node_big", "ms_edge_fast", "ms_edge_slow", "ms_compact", "ms_build_span")
            reps = 4
            for i in range(reps + 1):                  # (the first one also loads the kernels' code)
                ub.upload(); ub.build()
                if i >= 1:
                    sb = ub.stats()
                    for key in keys:
                        kern[key] = kern.get(key, 0.0) + sb[key] / reps
            big_stats = ub.stats()
            # the same unit rebuilt while resident (what r01's headline timed): kernels only, no upload, no download, no walk
            t1 = time.perf_counter()
            for _ in range(16):
                ub.build()
            kern["ms_resident_rebuild"] = 1e3 * (time.perf_counter() - t1) / 16

    # ---- parity spot-check of what was timed + CPU baseline on rank 0 at N=1: the real reference binary (oracle/_ref, -O2, 1 thread) ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample_pairs > 0:
        total_len = sum(chroms)
        frac = args.cpu_sample_pairs / float(pairs)
        sg = max(20000, int(total_len * frac))
        srun = os.path.join(args.workdir, "cpu_sample")
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import harness as H                     # the checker: reference binary (oracle/_ref) or the oracle, CPU baseline leg only
        D.synth(srun, seed=999, chroms=str(sg), pairs=args.cpu_sample_pairs, L=L, k=k, coverage=args.coverage)
        sample = "%d pairs 2x%d on a %d bp unit (same depth as the timed workload), stages (1)-(5), text parsing included" % (args.cpu_sample_pairs, L, sg)
        if H.have_reference(True):
            ref_out, secs = H.run_reference(srun, opt=True)
            kind, val = "reference", 2.0 * args.cpu_sample_pairs / secs
            # the GPU engine must reproduce the reference bytes on this very sample
            with A.Unit(k=k, insert_variation=50, coverage=args.coverage, device=local_rank) as u2:
                u2.load_files(os.path.join(srun, "tmp"), 0)
                u2.build()
                got = u2.finish()
            assert all(got[k2] == ref_out[0][k2] for k2 in ("initial", "pre", "extended")), "engine output differs from the reference binary on the CPU sample"
        else:
            t1 = time.perf_counter()
            H.run_oracle(os.path.join(srun, "tmp"), 0, k, 50, args.coverage)
            secs = time.perf_counter() - t1
            kind, val = "port", 2.0 * args.cpu_sample_pairs / secs
        cpu = {"value": round(val, 1), "unit": "reads/s", "cores": 1, "kind": kind, "sample": sample, "seconds": round(secs, 2)}
        # SURVEY §8(d)'s second baseline: as many independent units as this process has CPUs, one reference process per unit (the path has no threading of
        # its own; units are what can run side by side), all started together
        try:
            from concurrent.futures import ThreadPoolExecutor as _TPE
            n_par = max(2, min(A.usable_cpus(), 32))
            sruns = [D.synth(os.path.join(args.workdir, "cpu_par_%d" % i), seed=2000 + i, chroms=str(sg), pairs=args.cpu_sample_pairs, L=L, k=k, coverage=args.coverage) for i in range(n_par)]
            t1 = time.perf_counter()
            with _TPE(max_workers=n_par) as ex:
                if kind == "reference":
                    list(ex.map(lambda r: H.run_reference(r, opt=True), sruns))
                else:
                    list(ex.map(lambda r: H.run_oracle(os.path.join(r, "tmp"), 0, k, 50, args.coverage), sruns))
            wall = time.perf_counter() - t1
            cpu["parallel"] = {"value": round(2.0 * args.cpu_sample_pairs * n_par / wall, 1), "unit": "reads/s", "cores": n_par, "kind": kind, "seconds": round(wall, 2),
                               "sample": "%d independent units of that size side by side, one %s process each, wall time from the first start to the last end (process start-up and text parsing included); cores = the CPUs this process can keep busy (affinity, cgroup quota)" % (n_par, "reference" if kind == "reference" else "oracle")}
            for r in sruns:
                shutil_rm(r); shutil_rm(r + ".ref")
        except Exception as e:                                   # the second baseline is a report, never a reason to fail the bench
            cpu["parallel"] = {"error": str(e)}

    if rank == 0:
        outs = {uu: bytes(v) for uu, v in gathered.items()}
        assert sorted(outs) == (sorted(mine) if emu else list(range(n_units))), "the gather did not deliver every unit"
        all_ext = b"".join(outs[uu] for uu in sorted(outs))
        reads_per_step = 2.0 * (sam_pairs_total if emu else pairs)
        sec_per_step = elapsed / args.steps
        value = reads_per_step / sec_per_step
        # roofline of the dominant kernel (agx_k_node_sweep<0>) over rank 0's units in the last timed step: SURVEY §8d's algorithmic bytes of
        # those units over the HIP-event time of their sweeps; next to it the HBM bytes the counters saw (profiles/) over the same time
        sw_ms = sum(unit_stats[uu]["ms_node_sweep"] for uu in mine)
        abytes = sum(algorithmic_bytes(unit_stats[uu]["sam_line_pairs"], L, k, unit_stats[uu]["n_pos"]) for uu in mine)
        cbytes = sum(sweep_compulsory_bytes(unit_stats[uu], L, k) for uu in mine)
        achieved = cbytes / (sw_ms * 1e-3) / 1e9 if sw_ms > 0 else 0.0
        up_bytes = sum(unit_stats[uu]["upload_bytes"] for uu in mine)
        down_bytes = sum(unit_stats[uu]["download_bytes"] for uu in mine)
        traffic = hbm_rate = None
        issue = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                tab = json.load(open(pmc))      # measured HBM bytes of the sweep per (tile, hit) list entry, for this configuration if it was profiled
                per_entry = tab.get("node_sweep_bytes_per_tile_entry_by_config", {}).get(args.config, tab.get("node_sweep_bytes_per_tile_entry"))
                entries = sum(unit_stats[uu]["n_tile_entries"] for uu in mine)
                if per_entry:
                    traffic = int(per_entry * entries)
                    hbm_rate = traffic / (sw_ms * 1e-3) / 1e9 if sw_ms > 0 else None
                ins = tab.get("node_sweep_insts_per_tile_entry_by_config", {}).get(args.config)
                if ins and sw_ms > 0:
                    # the roof the sweep is under: a SIMD issues one vector instruction of a wave64 every 4 cycles, a CU's scalar unit one instruction per cycle (MI355X: 256 CUs x 4 SIMDs, 2.4 GHz)
                    cyc = sw_ms * 1e-3 * 2.4e9
                    issue = {"valu_per_entry": ins["valu"], "salu_per_entry": ins["salu"], "frac_vector_port": round(ins["valu"] * entries * 4 / (1024 * cyc), 3), "frac_scalar_port": round(ins["salu"] * entries / (256 * cyc), 3),
                             "note": "instructions per tile-list entry from the SQ counters (profiles/%s, measured with rocprofv3 --pmc like `traffic`) x this run's list entries over the issue slots of the same HIP-event time: the node sweep is "
                                     "bound by instruction issue and the waits between dependent vector and scalar instructions, not by HBM" % ins.get("from", "pmc_traffic.json")}
            except Exception:
                traffic = None
        per_unit = {str(uu): {"positions": unit_stats[uu]["n_pos"], "sam_pairs": unit_stats[uu]["sam_line_pairs"], "hits": unit_stats[uu]["n_hits"],
                              "upload_MB": round(unit_stats[uu]["upload_bytes"] / 1e6, 1), "hbm_MB": round(unit_stats[uu]["device_bytes"] / 1e6, 1),
                              "ms_node_sweep": round(unit_stats[uu]["ms_node_sweep"], 3), "ms_download": round(unit_stats[uu]["ms_download"], 2),
                              "ms_walk": round(unit_stats[uu]["ms_walk"], 2), "ms_upload_to_built": round(1e3 * unit_stats[uu]["s_upload_build"], 2),
                              "ms_unit_total": round(1e3 * unit_stats[uu]["s_total"], 2), "download_MB": round(unit_stats[uu]["download_bytes"] / 1e6, 1),
                              "build_attempts": unit_stats[uu]["build_attempts"], "spilled_ids": unit_stats[uu]["n_spilled"],
                              "mid_tiles": unit_stats[uu]["n_mid_tiles"], "big_tiles": unit_stats[uu]["n_big_tiles"], "dense_lists": unit_stats[uu]["dense_lists"], "rows_by_reference": unit_stats[uu]["rows_by_reference"]} for uu in mine}
        line = {
            "metric": "aligned reads/sec through graph build+extend", "value": round(value, 1), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * sec_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic" if not share else "synthetic (AGX_BENCH_SHARE_GPU plumbing test: ranks share a GPU, gloo gather)",
            "config": {"workload": label + ", k=%d, --coverage %d, synthetic target at 1%% SNP + 0.1%% indel" % (k, args.coverage)
                       + (" [NON-STANDARD generator options: %s]" % os.environ["AGX_BENCH_SYNTH"] if extra else ""),
                       "name": args.config, "units": n_units, "unit_positions": unit_len,
                       "timed_region": "T_core per step: every unit new to the device (upload + first build + download + host walk), %s memory pools, %s" % (args.pool, "the same units uploaded again every step" if reupload else "a fresh set of one-shot units per step (loaded from the unit caches before the clock)"),
                       "units_in_flight_per_gpu": inflight,
                       "parallelism": "units sharded longest-first over %d GPU%s, one RCCL gather of extended contigs per job" % (world, "" if world == 1 else "s")},
            "host_cpu_ms_per_step": round(1e3 * cpu_s_per_step, 2),      # CPU time of rank 0's process inside the timed region, per step (a 16-CPU quota gives 16 x ms_per_step)
            "single_gpu_ms_same_config": round(single_ms, 3) if single_ms else None,
            "speedup_vs_1gpu": round(single_ms / (1e3 * sec_per_step), 3) if single_ms else None,
            "value_same_config_1gpu": round(reads_per_step / (single_ms * 1e-3), 1) if single_ms else None,      # the 1-GPU point of THIS configuration: a 1 -> N curve of `value` over driver runs with different default configurations (cfg3 at N = 1, the whole-human shape at N > 1) compares workloads, this pair does not
            "t_core_s": round(sec_per_step, 4),
            "t_unit_s": round(sec_per_step + t_parse_max, 4),
            "sam_text_bytes": sam_bytes_total, "sam_seq_columns": bool(sam_seq),
            "t_unit_note": "t_core_s + text parsing and staging of the per-unit input files: the five text files of every unit -> staged arrays in pinned memory (slowest rank; its units side by side, each on its share of the cores: %.3f s); the one-off index of tmp/_reads.fa (%.3f s on rank 0, shared by all units of a run) is not in it" % (t_parse_max, t_index),
            "load_ms_per_unit": {str(uu): load_ms[uu] for uu in mine},
            "load_note": "per unit, wall: contigs = unit sequence + contig threading (on a thread of its own, beside the read alignments); read_alignments = SAM parsing, left-mate decision, rows of 2-bit read bases straight into the pinned upload buffers; rest = what the load took beyond the read alignments",
            "reads_index_s": round(t_index, 3), "usable_cpus": cpus, "load_threads_per_unit": int(os.environ.get("AGX_LOAD_THREADS", "0")),
            "value_t_unit": round(reads_per_step / (sec_per_step + t_parse_max), 1),
            "t_unit_cached_s": round(sec_per_step + t_cached_max, 4),
            "t_unit_cached_note": "t_core_s + loading every unit from its binary cache file (tmp/_agx_unit.<u>.bin, written where the alignments are distributed; up to 4 units side by side: %.3f s) instead of parsing text — what the timed steps' units were loaded from" % t_cached_max,
            "value_t_unit_cached": round(reads_per_step / (sec_per_step + t_cached_max), 1),
            "roofline": {"bound": "hbm", "kernel": "agx_k_node_sweep<0>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "achieved_note": "bytes the node sweep cannot avoid (bench.py sweep_compulsory_bytes: tile records, vote codes, conti-mer heads read once; node table written once — node state lives in LDS) of rank 0's units / HIP-event time of their node sweeps in the last timed step",
                         "compulsory_bytes": cbytes, "kernel_ms": round(sw_ms, 4),
                         "traffic": traffic, "achieved_hbm": round(hbm_rate, 1) if hbm_rate else None, "issue": issue,
                         "frac_hbm": round(hbm_rate / HBM_PEAK_GBS, 4) if hbm_rate else None,
                         "traffic_note": "HBM bytes from the PMC counters (profiles/pmc_traffic.json: bytes per tile-list entry of the sweep, measured with rocprofv3 --pmc) x this run's list entries; achieved_hbm = traffic / kernel_ms",
                         "algorithmic_bytes_8d": abytes,
                         "algorithmic_bytes_8d_note": "SURVEY 8(d)'s algorithmic bytes of the WHOLE path (per pair (L-k)*40 + L + 32, plus 64 per position).  r01/r02 printed them / the node sweep's time / peak as `frac` (it reached 0.99-1.07: one kernel was charged the whole path's bytes); they are now only divided by the whole job's time (`job_frac`)",
                         "frac_8d_kernel": round(abytes / (sw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sw_ms > 0 else None,
                         "frac_8d_kernel_note": "the quotient SURVEY 8(d) prescribes, printed so that a reader sees why it is not used as `frac`: the WHOLE path's algorithmic bytes / the node sweep's time / peak.  It comes out near or above 1 — not a fraction of a roof — because one kernel is charged every byte of the path, 32 bytes per arrival of which (the node state) live in LDS and never touch HBM; `frac` (bytes the kernel cannot avoid) and `frac_hbm` (counter traffic) are the kernel's, `job_frac` is 8(d)'s bytes against the time of everything that moves them",
                         "job_frac": round(abytes / sec_per_step / 1e9 / HBM_PEAK_GBS, 4),
                         "job_frac_note": "SURVEY 8(d)'s algorithmic bytes / T_core per job / peak: the whole path against the whole job's time (upload, kernels, download, host walk)",
                         "pcie": {"up_bytes": up_bytes, "down_bytes": down_bytes, "peak_GBs": 64.0,
                                  "up_frac_of_job": round(up_bytes / sec_per_step / 64e9, 4), "down_frac_of_job": round(down_bytes / sec_per_step / 64e9, 4),
                                  "upload_GBs_largest_unit": round(big_stats["upload_bytes"] / kern["ms_upload_dev"] / 1e6, 1) if (big_stats and kern.get("ms_upload_dev")) else None,
                                  "note": "host -> HBM and HBM -> host bytes of rank 0's units per job over T_core over 64 GB/s (PCIe 5 x16, one direction): how much of the job the link would be busy if nothing else ran; upload_GBs_largest_unit = the rate of the largest unit's copies alone on the device's upload stream"}},
            "cpu_baseline": cpu,
            "emulated_rank": emu_note,
            "breakdown_ms_largest_unit": {k2: round(v, 3) for k2, v in kern.items()},
            "breakdown_note": "largest unit of rank 0, exclusive builds with section events after the timed region (4 uploads + first builds); ms_resident_rebuild = r01's headline quantity (rebuild of a resident unit, kernels only)",
            "units": per_unit,
            "host": numa_note,
            "untimed_s": {"generate": round(t_gen, 2), "parse_text": round(float(tot[1].item()), 2), "stage_of_which": round(float(tot[2].item()), 3)},
            "sam_line_pairs": sam_pairs_total,
            "output": {"extended_contigs": all_ext.count(b">"), "extended_bytes": len(all_ext), "N50": n50_of(all_ext)},
        }
        if big_stats is not None:
            line["graph_largest_unit"] = {"positions": big_stats["n_pos"], "hits": big_stats["n_hits"], "nodes": big_stats["n_nodes"], "tile_entries": big_stats["n_tile_entries"],
                                          "walk_ids": big_stats["n_walk_ids"], "edge_overflow": big_stats["n_edge_overflow"]}
        if os.environ.get("AGX_BENCH_DIGEST"):                                  # tests: what the job delivered to rank 0, unit by unit
            import hashlib
            with open(os.environ["AGX_BENCH_DIGEST"], "w") as f:
                json.dump({str(uu): hashlib.md5(outs[uu]).hexdigest() for uu in sorted(outs)}, f)
        print(json.dumps(line))
    for r in held.values():
        r.free()
    if rank == 0:
        try:
            os.remove(os.path.join(run, ".in_use.%d" % os.getpid()))
        except OSError:
            pass
    for one in unit_sets:
        for un in one.values():
            un.close()
    if not args.keep and rank == 0:
        import shutil
        if dist:
            dist.barrier()
        shutil.rmtree(run, ignore_errors=True)
        shutil.rmtree(os.path.join(args.workdir, "cpu_sample"), ignore_errors=True)
        shutil.rmtree(os.path.join(args.workdir, "cpu_sample.ref"), ignore_errors=True)
    elif dist:
        dist.barrier()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
