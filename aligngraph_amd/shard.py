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


def gather_bytes(payload, dist, device, rank, world, dst=0):
    """gather-v of one bytes object per rank to `dst`: all_gather of the 8-byte sizes, then one padded gather.
    xGMI is point-to-point, so every peer->root transfer rides its own link; the payload (<= one unit's FASTA) is tiny
    next to the build.  Returns the list of payloads on dst, None elsewhere."""
    import torch
    if dist is None or world == 1:
        return [payload]
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    outs = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, outs, dst=dst)
    if rank != dst:
        return None
    return [bytes(o[:s].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]


class UnitGather:
    """The same gather-v for a loop that runs it every step: persistent staging buffers (pinned on the GPU box), one all_reduce(MAX) of
    the payload size per step instead of an all_gather + world host reads, the transfers queued without blocking, and no bytes objects
    built on the root until someone asks.  Per step the calling thread pays one 8-byte device-to-host read plus a memcpy of its own
    payload; the root's device-to-host copy of the world x cap gather buffer completes behind an event.

    step(payload) -> handle; handle.payloads() on dst gives the list of payloads (bytes) of that step, None elsewhere.  A handle
    reads the persistent buffers, so it is only good until the next step()."""

    class _Handle:
        def __init__(self, owner, sizes_host, event, host, cap):
            self._o, self._sizes, self._event, self._host, self._cap = owner, sizes_host, event, host, cap

        def payloads(self):
            if self._host is None:
                return None
            if self._event is not None:
                self._event.synchronize()
            flat = self._host.numpy()
            out = []
            for r in range(self._o.world):
                n = int.from_bytes(flat[r * self._cap:r * self._cap + 8].tobytes(), "little")
                out.append(flat[r * self._cap + 8:r * self._cap + 8 + n].tobytes())
            return out

    def __init__(self, dist, device, rank, world, dst=0):
        self.dist, self.device, self.rank, self.world, self.dst = dist, device, rank, world, dst
        self.cap = 0
        self._stage = self._dev = self._recv = self._recv_host = self._staged = None
        self._side = None                     # root, GPU box: the device-to-host copies of the gathered buffers run on their own stream, so the
        self._copied = None                   # next step's size exchange (a host read) does not wait for them

    def _grow(self, cap):
        import torch
        pin = self.device.type == "cuda"
        if self._copied is not None:
            self._copied.synchronize()
        self.cap = cap
        self._staged = self._copied = None
        self._stage = torch.zeros(cap, dtype=torch.uint8, pin_memory=pin)
        self._dev = torch.zeros(cap, dtype=torch.uint8, device=self.device)
        if self.rank == self.dst:
            self._recv = [torch.zeros(cap, dtype=torch.uint8, device=self.device) for _ in range(self.world)]
            self._recv_host = torch.zeros(cap * self.world, dtype=torch.uint8, pin_memory=pin)

    def step(self, payload, head=b""):
        """payload — bytes or a 1-D uint8 numpy array — and an optional short head in front of it (so that a caller need not join the two)
        -> handle.  The payload is read before step() returns, except with a single rank, where the handle keeps a reference to it."""
        import numpy as np
        import torch
        if self.dist is None or self.world == 1:
            class _Local:
                def payloads(_s):
                    body = payload if isinstance(payload, (bytes, bytearray)) else payload.tobytes()
                    return [head + body if head else body]
            return _Local()
        total = len(head) + len(payload)
        need = torch.tensor([total + 8], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(need, op=self.dist.ReduceOp.MAX)
        need = int(need.item())
        if need > self.cap:
            self._grow((need + need // 8 + (1 << 20)) & ~((1 << 20) - 1))       # every rank sees the same maximum, so the same capacity
        if self._staged is not None:
            self._staged.synchronize()                                          # the previous step's host-to-device copy has read the staging buffer
        st = self._stage.numpy()
        st[:8] = np.frombuffer(total.to_bytes(8, "little"), dtype=np.uint8)
        if head:
            st[8:8 + len(head)] = np.frombuffer(head, dtype=np.uint8)
        if len(payload):
            st[8 + len(head):8 + total] = np.frombuffer(payload, dtype=np.uint8) if isinstance(payload, (bytes, bytearray)) else payload      # the one host copy of the payload
        self._dev.copy_(self._stage, non_blocking=True)
        if self.device.type == "cuda":
            self._staged = torch.cuda.Event()
            self._staged.record()
        cuda = self.device.type == "cuda"
        if cuda and self.rank == self.dst and self._copied is not None:
            torch.cuda.current_stream().wait_event(self._copied)                # the previous step's copies have read the receive buffers
        self.dist.gather(self._dev, self._recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return UnitGather._Handle(self, None, None, None, self.cap)
        host = self._recv_host
        if not cuda:
            for r in range(self.world):
                host[r * self.cap:(r + 1) * self.cap].copy_(self._recv[r])
            return UnitGather._Handle(self, None, None, host, self.cap)
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            for r in range(self.world):
                host[r * self.cap:(r + 1) * self.cap].copy_(self._recv[r], non_blocking=True)
            self._copied = torch.cuda.Event()
            self._copied.record()
        return UnitGather._Handle(self, None, self._copied, host, self.cap)


def pack_units(unit_ids, blobs):
    """Length-prefixed concatenation of (unit id, bytes) so that one gather carries all units of a rank."""
    import struct
    out = [struct.pack("<I", len(unit_ids))]
    for u, b in zip(unit_ids, blobs):
        out.append(struct.pack("<IQ", u, len(b)))
        out.append(b)
    return b"".join(out)


def unit_header(unit_id, length):
    """What pack_units([unit_id], [blob]) puts in front of the blob."""
    import struct
    return struct.pack("<IIQ", 1, unit_id, length)


def unpack_units(payload):
    import struct
    (n,), p = struct.unpack_from("<I", payload, 0), 4
    out = {}
    for _ in range(n):
        u, ln = struct.unpack_from("<IQ", payload, p)
        p += 12
        out[u] = payload[p:p + ln]
        p += ln
    return out
