"""The N>1 path on CPU: world_size-2 gloo processes shard units, compute them (the oracle stands in for the GPU build in this
CPU-only test), and gather the per-unit extended contigs on rank 0 through aligngraph_amd.shard.run_job — the very function bench.py
--gpus N runs over RCCL with the HIP engine as run_unit."""
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
    blob = shard.pack_units([3, 0], [b"abc", b""])
    assert shard.unpack_units(blob) == {3: b"abc", 0: b""}
    assert shard.unpack_units(shard.pack_units([], [])) == {}


def test_plan_is_longest_first_within_the_rank():
    sizes = [30, 20, 23, 19, 27]
    plans = [shard.plan(sizes, r, 2) for r in range(2)]
    assert sorted(u for pl in plans for u in pl) == list(range(5))
    for pl in plans:
        assert [sizes[u] for u in pl] == sorted((sizes[u] for u in pl), reverse=True)
    assert shard.plan(sizes, 0, 1) == [0, 4, 2, 1, 3]                 # one GPU: all five, largest first (cfg3's order in bench.py)


def test_run_job_single_rank_needs_no_process_group():
    got = shard.run_job([3, 9, 5], 0, 1, lambda u: b"unit%d" % u, None, None, inflight=2)
    assert got == {0: b"unit0", 1: b"unit1", 2: b"unit2"}
    with pytest.raises(ZeroDivisionError):
        shard.run_job([1, 2], 0, 1, lambda u: 1 // 0, None, None)


def _worker(rank, world, port, run, meta, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ran = []

    def run_unit(u):
        ran.append(u)
        return H.run_oracle(os.path.join(run, "tmp"), u, meta["k"], meta["insert_variation"], meta["coverage"])["extended"]

    for job in range(2):                               # bench.py runs the job once per step
        merged = shard.run_job(meta["unit_len"], rank, world, run_unit, dist, torch.device("cpu"), inflight=2)
        assert sorted(ran[-len(shard.plan(meta["unit_len"], rank, world)):]) == sorted(shard.plan(meta["unit_len"], rank, world))
        assert (merged is not None) == (rank == 0)
    if rank == 0:
        q.put({u: bytes(v) for u, v in merged.items()})      # (run_job hands out views: of the C buffers for its own units, of its landing buffers — kept and overwritten by the next job — for the others')
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


def test_units_are_admitted_by_device_memory():
    """shard.run_job with hbm_need / hbm_budget: never more in flight than the device holds, plan order kept, a unit beyond the budget runs alone."""
    import threading
    import time
    from aligngraph_amd import shard
    sizes = [50, 48, 40, 30, 20, 10, 5, 120]
    need = {u: s for u, s in enumerate(sizes)}
    lock, now, peak, order, alone = threading.Lock(), [0], [0], [], []

    def start_unit(u):
        order.append(u)

    def run_unit(u):
        with lock:
            now[0] += need[u]
            peak[0] = max(peak[0], now[0])
            if need[u] > 100:
                alone.append(now[0] == need[u])
        time.sleep(0.01 + 0.0005 * need[u])
        with lock:
            now[0] -= need[u]
        return b"u%d" % u
    out = shard.run_job(sizes, 0, 1, run_unit, None, None, inflight=8, start_unit=start_unit, hbm_need=need, hbm_budget=100)
    assert sorted(out) == list(range(8)) and out[3] == b"u3"
    assert order == shard.plan(sizes, 0, 1)                 # largest first, as planned
    assert peak[0] <= 120 and alone == [True]               # 120 > budget: it ran with nothing beside it; everything else within 100
    with lock:
        assert now[0] == 0


def test_a_unit_that_gives_memory_back_early_lets_the_next_one_in():
    """r05: run_unit(u, release) — a unit that has trimmed its device memory after the download (agx_unit_trim) tells the job so, and the next unit of the plan is admitted
    while this one is still being walked.  The budget is never exceeded by what the units hold at any moment, and a unit cannot release more than it was admitted with."""
    import threading
    import time
    from aligngraph_amd import shard
    sizes = [60, 50, 40, 30]
    need = {u: s for u, s in enumerate(sizes)}
    lock, now, peak, started, t0 = threading.Lock(), [0], [0], {}, time.perf_counter()

    def start_unit(u):
        started[u] = time.perf_counter() - t0

    def run_unit(u, release):
        with lock:
            now[0] += need[u]
            peak[0] = max(peak[0], now[0])
        time.sleep(0.05)                                   # upload + build + download
        keep = need[u] // 4
        with lock:
            now[0] -= need[u] - keep
        release(need[u] - keep)
        time.sleep(0.30)                                   # the walk
        with lock:
            now[0] -= keep
        return b"u%d" % u
    run_unit.takes_release = True
    out = shard.run_job(sizes, 0, 1, run_unit, None, None, inflight=4, start_unit=start_unit, hbm_need=need, hbm_budget=100)
    assert sorted(out) == [0, 1, 2, 3]
    assert started[1] < 0.25 and started[2] < 0.30, started          # unit 1 (50) did not wait for unit 0's walk (0.35 s): it went in when unit 0 gave 45 back
    with lock:
        assert now[0] == 0


def _stream_worker(rank, world, port, case, q):
    """shard.UnitStream under gloo: made-up units whose completion order and timing the case dictates."""
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if case == "end":
        os.environ["AGX_GATHER"] = "end"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [90, 80, 70, 60, 50, 40, 30]                   # 2 ranks: rank 0 gets {0, 3, 4, 6}, rank 1 {1, 2, 5}
    mine = shard.plan(sizes, rank, world)
    took = {}

    def payload(u, job):
        return b"" if u == 5 else (b"unit %d of job %d;" % (u, job)) * (1000 * (u + 1))      # (unit 5: nothing extended — zero bytes travel as an announcement alone)

    for job in range(3):
        def run_unit(u, job=job):
            t0 = time.perf_counter()
            if case == "peer-first" and rank == 0:
                time.sleep(0.5)                            # the root's units are still being built when the peer has long finished all of its own
            if case == "reverse" and rank == 1:
                time.sleep(0.05 * (10 - u))                # the peer's units finish in the reverse of the plan's order
            if case == "fail" and rank == 1 and u == 2 and job == 1:
                raise ValueError("unit 2 cannot be built")
            took[u] = time.perf_counter() - t0
            b = payload(u, job)
            return np.frombuffer(b, dtype=np.uint8) if u % 2 else b      # (bench.py hands over numpy views of C memory, the tests bytes)
        try:
            merged = shard.run_job(sizes, rank, world, run_unit, dist, torch.device("cpu"), inflight=3)
        except Exception as e:
            q.put((rank, job, "raised", type(e).__name__, str(e)))
            break
        assert (merged is not None) == (rank == 0)
        if rank == 0:
            assert sorted(merged) == list(range(len(sizes)))
            for u in merged:
                assert bytes(merged[u]) == payload(u, job), (case, job, u)
            q.put((rank, job, "ok", None, None))
    dist.barrier() if case != "fail" else None
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["plain", "reverse", "peer-first", "end", "fail"])
def test_units_travel_to_the_root_as_they_finish(case):
    """r06: the gather runs while the units are computed — a peer announces every finished unit through the process group's store and sends it, the root
    posts the receive when it has read the announcement (shard.UnitStream).  Three jobs in a row per case (the job number is part of every key and every payload): units that
    finish out of the plan's order, a peer that is done before the root's first unit, a unit of zero bytes, numpy views and bytes, AGX_GATHER=end (the gather at the end, r05),
    and a peer whose unit fails: the root's job raises instead of waiting for bytes that will not come."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = []
    want = 3 if case != "fail" else 3                       # fail: job 0 ok on the root, then one "raised" from each rank
    while len(got) < want:
        got.append(q.get(timeout=120))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if case != "fail":
        assert [g[2] for g in got] == ["ok"] * 3
    else:
        assert sorted((g[0], g[1], g[2]) for g in got) == [(0, 0, "ok"), (0, 1, "raised"), (1, 1, "raised")], got
        assert any(g[3] == "ValueError" for g in got) and any("rank 1 failed" in (g[4] or "") for g in got), got
