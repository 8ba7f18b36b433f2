"""The N>1 path on CPU: world_size-2 gloo processes shard units, compute them (the oracle stands in for the GPU build in this
CPU-only test), and gather the per-unit extended contigs on rank 0 through aligngraph_amd.shard — the same code bench.py runs over RCCL."""
import os
import socket

import pytest

import harness as H
from aligngraph_amd import shard


def test_assign_units_is_lpt_and_deterministic():
    sizes = [30, 20, 23, 19, 27]          # A. thaliana-like chromosome lengths (Mb)
    a = shard.assign_units(sizes, 4)
    assert sorted(u for r in a for u in r) == list(range(5))
    assert max(sum(sizes[u] for u in r) for r in a) == 39          # {20,19} share a GPU; makespan 39 vs ideal 29.75
    assert shard.assign_units(sizes, 4) == a
    assert shard.assign_units([5], 8) == [[0]] + [[]] * 7
    assert shard.assign_units([], 2) == [[], []]


def test_pack_roundtrip():
    assert shard.unit_header(7, 3) + b"abc" == shard.pack_units([7], [b"abc"])
    blob = shard.pack_units([3, 0], [b"abc", b""])
    assert shard.unpack_units(blob) == {3: b"abc", 0: b""}
    assert shard.unpack_units(shard.pack_units([], [])) == {}


def _worker(rank, world, port, run, meta, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.assign_units(meta["unit_len"], world)[rank]
    blobs = [H.run_oracle(os.path.join(run, "tmp"), u, meta["k"], meta["insert_variation"], meta["coverage"])["extended"] for u in mine]
    got = shard.gather_bytes(shard.pack_units(mine, blobs), dist, torch.device("cpu"), rank, world)
    # the per-step variant bench.py uses: persistent buffers, growing capacity, several steps
    g = shard.UnitGather(dist, torch.device("cpu"), rank, world)
    steps = [g.step(shard.pack_units(mine[:n], blobs[:n])).payloads() for n in (0, len(mine), 1)]
    # ... and the form that saves the host-side join: a short head in front of one unit's bytes
    import numpy as np
    headed = g.step(np.frombuffer(blobs[0], dtype=np.uint8), head=shard.unit_header(mine[0], len(blobs[0]))).payloads() if mine else g.step(b"", head=shard.pack_units([], [])).payloads()
    if rank == 0:
        merged = {}
        for payload in got:
            merged.update(shard.unpack_units(payload))
        again = {}
        for payload in steps[1]:
            again.update(shard.unpack_units(payload))
        assert again == merged and all(shard.unpack_units(p) == {} for p in steps[0])
        assert sum(len(shard.unpack_units(p)) for p in steps[2]) == sum(1 for r in shard.assign_units(meta["unit_len"], world) if r)
        firsts = {}
        for payload in headed:
            firsts.update(shard.unpack_units(payload))
        assert firsts == {r[0]: merged[r[0]] for r in shard.assign_units(meta["unit_len"], world) if r}
        q.put(merged)
    else:
        assert steps == [None, None, None] and headed is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_gather(built, tmp_path):
    import torch.multiprocessing as mp
    run = H.synth(str(tmp_path / "run"), seed=77, chroms="12000,9000,7000", pairs=4000, coverage=3, sam_seq=0)
    meta = H.read_meta(run)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, run, meta, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(merged) == [0, 1, 2]
    for u in range(3):
        assert merged[u] == H.run_oracle(os.path.join(run, "tmp"), u, meta["k"], meta["insert_variation"], meta["coverage"])["extended"]
