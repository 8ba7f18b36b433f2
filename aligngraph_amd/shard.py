"""Unit sharding across the GPUs of one node and the path's only exchange step.

Units (chromosomes or --part slices) are independent (AG:4765-4783 shares no state between iterations), so
multi-GPU is: assign units to ranks, build each on its own device, gather the per-unit FASTA byte buffers on rank 0.
`torch.distributed` is plumbing only: backend "nccl" (= RCCL over xGMI) with device tensors on the GPU box, "gloo" with
CPU tensors in the CPU test-suite.
"""


def assign_units(sizes, world):
    """Longest-processing-time-first: returns world lists of unit indices; deterministic."""
    order = sorted(range(len(sizes)), key=lambda u: (-sizes[u], u))
    load = [0] * world
    out = [[] for _ in range(world)]
    for u in order:
        r = min(range(world), key=lambda i: (load[i], i))
        out[r].append(u)
        load[r] += sizes[u]
    return [sorted(x) for x in out]


def plan(unit_sizes, rank, world):
    """The units of `rank`: longest-processing-time-first over the ranks, and longest first within the rank (its host walk then overlaps
    the uploads and kernels of the smaller units)."""
    mine = assign_units(unit_sizes, world)[rank]
    return sorted(mine, key=lambda u: (-unit_sizes[u], u))


def run_job(unit_sizes, rank, world, run_unit, dist, device, inflight=8, dst=0, start_unit=None, hbm_need=None, hbm_budget=None, plan_as=None):
    """One whole job the way bench.py --gpus N and the gloo test run it: this rank's units (plan) through run_unit(u) -> bytes on up to
    `inflight` worker threads that take the next unit as they finish one, then the path's ONLY exchange — one gather of every rank's
    per-unit bytes to rank `dst` (north_star: "gather of extended contigs at the end").  run_unit may return anything with
    the buffer interface (bytes, a numpy uint8 view of C memory).  start_unit(u), if given, is called when a worker TAKES unit u, in plan
    order and one at a time (bench.py queues the unit's upload there: uploads then reach the device largest unit first, whatever the
    threads do next).  hbm_need / hbm_budget: admission by device memory — {unit: bytes its upload will take (agx_unit_hbm_needed)} and what this rank's
    device may hold at once: a unit is only taken when it fits beside the units in flight (a unit larger than the budget: when nothing else is in flight), in
    plan order — whole-human units take up to 57 GB of HBM each, eight of them do not fit one device.  A run_unit that declares `takes_release = True` is called as
    run_unit(u, release) and may call release(nbytes) when the unit has given part of its device memory back before it is done (agx_unit_trim after the download: the
    next unit is then admitted while this one is still being walked on the host).  Returns {unit: bytes-like} of ALL units on dst, None elsewhere."""
    import threading
    mine = plan(unit_sizes, *plan_as) if plan_as else plan(unit_sizes, rank, world)      # plan_as = (R, N): one process runs rank R's share of an N-rank job alone (bench.py --emulate-rank)
    out, errs = {}, []
    nxt, take = iter(mine), threading.Lock()
    held, room = [0], threading.Condition()

    def worker():
        try:
            while True:
                need = 0
                with take:
                    u = next(nxt, None)
                    if u is not None:
                        if hbm_need is not None and hbm_budget:
                            need = hbm_need[u]
                            with room:                 # (the next unit of the plan waits here, and everybody behind it at `take`: the order stands)
                                room.wait_for(lambda: errs or held[0] == 0 or held[0] + need <= hbm_budget)
                                held[0] += need
                        if start_unit is not None:
                            start_unit(u)
                if u is None:
                    return
                left = [need]                              # what this unit still holds of the budget

                def release(nbytes, left=left):
                    n = max(0, min(int(nbytes), left[0]))
                    if n:
                        with room:
                            held[0] -= n
                            left[0] -= n
                            room.notify_all()
                try:
                    out[u] = run_unit(u, release) if getattr(run_unit, "takes_release", False) else run_unit(u)
                finally:
                    release(left[0])
        except BaseException as e:                     # surfaces in the calling thread
            errs.append(e)
            with room:
                room.notify_all()

    threads = [threading.Thread(target=worker) for _ in range(max(1, min(inflight, len(mine))))]
    # r06: the exchange runs WHILE the units are computed (UnitStream below): a peer sends each unit's bytes as the unit finishes, the root receives them as they are
    # announced.  AGX_GATHER=end: one gather after the last walk (gather_units; r02-r05).
    import os
    stream = None
    if dist is not None and world > 1 and plan_as is None and os.environ.get("AGX_GATHER", "stream") != "end":
        stream = UnitStream(unit_sizes, dist, device, rank, world, dst)
        if rank != dst:
            done_unit = run_unit

            def run_and_send(u, *a, _inner=done_unit):
                b = _inner(u, *a)
                stream.send(u, b)
                return b
            run_and_send.takes_release = getattr(run_unit, "takes_release", False)
            run_unit = run_and_send
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if stream is not None:
        return stream.finish(mine, out, errs[0] if errs else None)
    if errs:
        raise errs[0]
    if dist is None or world == 1:
        return out                                     # one rank: the outputs are where they are wanted already — nothing is copied
    return gather_units(mine, [out[u] for u in mine], dist, device, rank, world, dst)


_streams = {"group": None, "job": 0, "arena": {}}


class _Arena:
    """Host landing memory for what one peer sends (pinned when the payload comes down from a device), kept between jobs: pinning costs 0.2 ms per MB and a job's sizes repeat."""

    def __init__(self, torch, pinned):
        self.torch, self.pinned, self.chunks, self.at = torch, pinned, [], (0, 0)

    def reset(self):
        self.at = (0, 0)

    def take(self, n):
        i, off = self.at
        while i < len(self.chunks) and off + n > self.chunks[i].numel():
            i, off = i + 1, 0
        if i == len(self.chunks):
            self.chunks.append(self.torch.empty(max(n + n // 8, 32 << 20), dtype=self.torch.uint8, pin_memory=self.pinned))
        self.at = (i, off + n)
        return self.chunks[i][off:off + n]


class UnitStream:
    """The gather as units finish.  One barrier gather after the last walk left the root's one PCIe link with everybody's bytes at once, behind the job's
    last walk: 2.7 GB = 45 ms of a whole-human job whose critical rank has 95 ms.  Here every unit's bytes leave their rank when the unit is done, beside the units still in flight:

      * the handshake is on the CPU: a peer ANNOUNCES a finished unit through the process group's key-value store ("<job>/<rank>/<n-th finished>" = "<unit>,<bytes>"), after it
        has put the bytes where its send will read them, and only then sends; the root posts the matching receive only when it has read the announcement.  A receive is never
        posted for a sender that is not ready: under RCCL a posted receive is a kernel on the root's GPU, and one that spins in front of the root's own node sweeps would stall
        the rank that is the job's critical path (why r05 did not send early);
      * the transfers use a process group of their own (nobody else's collectives interleave with them) from one sender thread per peer and one receiver thread per peer on the
        root, the root's receives one at a time (they share its link anyway);
      * nothing is gathered at the end: the root knows from the plan (assign_units is deterministic) which units each peer owes it and returns when it has them all.

    gloo in the CPU tests (host tensors), RCCL on the GPU box (device tensors: host -> peer's HBM -> xGMI -> root's HBM -> pinned landing memory)."""

    def __init__(self, unit_sizes, dist, device, rank, world, dst):
        import queue
        import threading
        import torch
        from torch.distributed import distributed_c10d as c10d
        self.dist, self.device, self.rank, self.world, self.dst, self.torch = dist, device, rank, world, dst, torch
        self.cuda = torch.device(device).type == "cuda"
        if _streams["group"] is None:
            _streams["group"] = dist.new_group()           # (collective: every rank's first job makes it)
        self.group, self.store = _streams["group"], c10d._get_default_store()
        _streams["job"] += 1
        self.job = _streams["job"]
        self.owed = assign_units(unit_sizes, world)         # what every rank computes
        self.errs, self.merged = [], {}
        self.lock = threading.Lock()                        # the root's receives, one at a time
        self.threads = []
        if rank != dst:
            self.q = queue.Queue()
            t = threading.Thread(target=self._sender)
            t.start(); self.threads.append(t)
        else:
            for peer in range(world):
                if peer != dst and self.owed[peer]:
                    t = threading.Thread(target=self._receiver, args=(peer,))
                    t.start(); self.threads.append(t)

    def _key(self, rank, n):
        return "agx/gather/%d/%d/%d" % (self.job, rank, n)

    def send(self, u, blob):
        """(a peer's worker thread, when unit u is done)"""
        self.q.put((u, blob))

    def _sender(self):
        import numpy as np
        torch, n_sent = self.torch, 0
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)          # (the current device is per thread)
            while True:
                item = self.q.get()
                if item is None:
                    return
                u, blob = item
                if isinstance(blob, BaseException):          # this rank failed: the root must not wait for the rest
                    self.store.set(self._key(self.rank, n_sent), "error,%s" % type(blob).__name__)
                    return
                v = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob.view(np.uint8).reshape(-1)
                t = torch.from_numpy(v if v.flags.writeable else v.copy()).to(self.device) if v.size else None      # (first where the send will read them ...)
                self.store.set(self._key(self.rank, n_sent), "%d,%d" % (u, v.size))                                 # (... then the announcement ...)
                n_sent += 1
                if t is not None:
                    self.dist.send(t, self.dst, group=self.group)                                                      # (... then the send: the root posts its receive when it has read it)
        except BaseException as e:
            self.errs.append(e)

    def _receiver(self, peer):
        torch = self.torch
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            arena = _streams["arena"].setdefault((peer, self.cuda), _Arena(torch, self.cuda))
            arena.reset()
            for n_got in range(len(self.owed[peer])):
                key = self._key(peer, n_got)
                self.store.wait([key])                       # asleep on the CPU until the peer has announced its next unit
                word = self.store.get(key).decode()
                try:
                    self.store.delete_key(key)
                except Exception:
                    pass
                if word.startswith("error"):
                    raise RuntimeError("rank %d failed (%s): its units will not arrive" % (peer, word.partition(",")[2]))
                u, n = (int(x) for x in word.split(","))
                if n == 0:
                    self.merged[u] = memoryview(b"")
                    continue
                host = arena.take(n)
                with self.lock:
                    if self.cuda:
                        buf = torch.empty(n, dtype=torch.uint8, device=self.device)
                        self.dist.recv(buf, peer, group=self.group)
                        host.copy_(buf, non_blocking=True)
                        torch.cuda.current_stream().synchronize()
                    else:
                        self.dist.recv(host, peer, group=self.group)      # (gloo: straight into the landing memory)
                self.merged[u] = memoryview(host.numpy())
        except BaseException as e:
            self.errs.append(e)

    def finish(self, mine, out, err):
        """(every rank, when its own units are done; err: what one of them raised) -> {unit: bytes-like} of ALL units on the root, None elsewhere"""
        import numpy as np
        if self.rank != self.dst:
            self.q.put((None, err) if err is not None else None)
        for t in self.threads:
            t.join()
        if err is not None:
            raise err
        if self.errs:
            raise self.errs[0]
        if self.rank != self.dst:
            return None
        for u in mine:
            b = out[u]
            self.merged[u] = memoryview(np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b.view(np.uint8).reshape(-1))
        return self.merged


_pinned = {}      # root's landing buffers for the gather, by peer: pinning host memory costs 0.2 ms per MB, a job's sizes repeat from step to step


def _landing(torch, peer, nbytes, cuda):
    """A host uint8 tensor of at least nbytes for what `peer` sends (pinned when the payload comes down from a device), kept between jobs."""
    have = _pinned.get((peer, cuda))
    if have is None or have.numel() < nbytes:
        have = torch.empty(max(nbytes, 1) + max(nbytes, 1) // 8, dtype=torch.uint8, pin_memory=cuda)
        _pinned[(peer, cuda)] = have
    return have


def gather_units(unit_ids, blobs, dist, device, rank, world, dst=0):
    """The path's one exchange: every rank's per-unit byte buffers to `dst`, unit by unit, without a copy of the payload beyond the transfers
    themselves — a whole-human job gathers 3.1 GB of extended contigs, and r03's pack / send / unpack (five Python-level copies of it) would have cost more
    than the eight ranks' units.  One all_gather_object of the (unit, length) lists; then every peer sends each of its buffers as it is (a view of the C memory
    agx_unit_finish left it in -> device -> peer link) and the root posts all its receives at once (grouped point-to-point: ncclGroupStart / ncclSend /
    ncclRecv / ncclGroupEnd under backend "nccl", one xGMI link per peer), copying each arrival into a pinned landing buffer that it keeps between jobs.
    Returns {unit: bytes-like} of ALL units on dst (memoryviews; the root's own units are what it passed in, the others' lie in the landing buffers and stay
    valid until the next gather), None elsewhere."""
    import numpy as np
    import torch
    views = [np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b.view(np.uint8).reshape(-1) for b in blobs]
    if dist is None or world == 1:
        return {u: memoryview(v) for u, v in zip(unit_ids, views)}
    cuda = torch.device(device).type == "cuda"
    meta = [None] * world
    dist.all_gather_object(meta, [(int(u), int(v.size)) for u, v in zip(unit_ids, views)])
    if rank != dst:
        ops, keep = [], []
        for v in views:
            if v.size:
                t = torch.from_numpy(v if v.flags.writeable else v.copy()).to(device)      # (device: the rank's GPU under nccl; gloo sends host tensors)
                keep.append(t); ops.append(dist.P2POp(dist.isend, t, dst))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return None
    merged = {u: memoryview(v) for u, v in zip(unit_ids, views)}
    ops, parts = [], []
    for r in range(world):
        if r == dst:
            continue
        total = sum(n for _, n in meta[r])
        land = _landing(torch, r, total, cuda) if total else None
        at = 0
        for u, n in meta[r]:
            if n == 0:
                merged[u] = memoryview(b"")
                continue
            host = land[at:at + n]; at += n
            buf = torch.empty(n, dtype=torch.uint8, device=device) if cuda else host      # (gloo: straight into the landing buffer)
            ops.append(dist.P2POp(dist.irecv, buf, r)); parts.append((u, buf, host))
    if ops:
        reqs = dist.batch_isend_irecv(ops)
        for req, (u, buf, host) in zip(reqs, parts) if len(reqs) == len(parts) else ():
            req.wait()
            if cuda:
                host.copy_(buf, non_blocking=True)
        if len(reqs) != len(parts):                    # (a backend that answers a batch with one request for the whole group)
            for req in reqs:
                req.wait()
            if cuda:
                for u, buf, host in parts:
                    host.copy_(buf, non_blocking=True)
        if cuda:
            torch.cuda.synchronize()
        for u, buf, host in parts:
            merged[u] = memoryview(host.numpy())
    return merged


def gather_bytes(payload, dist, device, rank, world, dst=0):
    """gather-v of one bytes object per rank to `dst`: an all_gather of the 8-byte sizes, then every peer sends exactly its bytes to the root and
    the root posts all its receives at once (grouped point-to-point: RCCL's ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd under backend
    "nccl").  xGMI is point-to-point, so every peer->root transfer rides its own link, and nobody pads to the largest payload.  Returns the
    list of payloads on dst, None elsewhere."""
    import torch
    if dist is None or world == 1:
        return [payload]
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(v) for v in torch.cat(sizes).cpu().tolist()]          # (one device -> host copy for all of them)
    if rank != dst:
        if len(payload):
            buf = torch.frombuffer(payload if isinstance(payload, bytearray) else bytearray(payload), dtype=torch.uint8).to(device)
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, dst)]):
                req.wait()
        return None
    bufs = {r: torch.empty(sizes[r], dtype=torch.uint8, device=device) for r in range(world) if r != dst and sizes[r]}
    if bufs:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, r) for r, b in bufs.items()]):
            req.wait()
    out = []
    for r in range(world):
        if r == dst:
            out.append(bytes(payload))
        else:
            out.append(bufs[r].cpu().numpy().tobytes() if r in bufs else b"")
    return out


def pack_units(unit_ids, blobs):
    """Length-prefixed concatenation of (unit id, bytes-like) so that one gather carries all units of a rank (one copy of the payload)."""
    import struct
    out = [struct.pack("<I", len(unit_ids))]
    for u, b in zip(unit_ids, blobs):
        out.append(struct.pack("<IQ", u, len(b)))
        out.append(b)
    return bytearray().join(out)


def unpack_units(payload):
    import struct
    (n,), p = struct.unpack_from("<I", payload, 0), 4
    out = {}
    for _ in range(n):
        u, ln = struct.unpack_from("<IQ", payload, p)
        p += 12
        out[u] = bytes(payload[p:p + ln])
        p += ln
    return out
