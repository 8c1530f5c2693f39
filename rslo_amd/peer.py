"""Same-stream statistics exchange between the ranks of one node (csrc/peer.hip, include/rslo_hip.h rslo_peer_*).

The BEV head's SyncBatchNorm (reference: rslo/layers/SparseConv.py:96-132 over apex SyncBatchNorm, train_hdf5.py:463)
needs the per-channel sums of every layer summed over the ranks, between two dependent kernels of that layer: 90
exchanges of <= 513 doubles per training step.  `all_reduce_(t, group)` is that sum.  Where every rank of the group
lives on this host it is ONE small kernel on the current stream (rslo_peer_allreduce_f64: write own slice, spin on the
peers' flags, add in rank order) instead of an RCCL collective launch with its two stream hand-offs; anywhere else --
several hosts, an explicit sub-group, a CPU tensor, RSLO_SYNCBN_EXCHANGE=rccl -- it is `dist.all_reduce`.

Transport (RSLO_SYNCBN_EXCHANGE): "device" = every rank's slice in its own HBM, opened by the peers through HIP IPC
(polls and pulls are xGMI reads; 5.4 us per exchange between two processes on one GPU), "host" = a shared-memory segment
registered with the HIP runtime (any GPUs of a host; 9.3 us), "rccl" = the collective, "auto" (default) = device if every
rank can open every handle, else host, else rccl -- agreed over the group, never a per-rank choice.
Sums are formed in rank order on every rank: all ranks hold identical bits (an RCCL ring does not promise that)."""
import ctypes as C
import os
import socket

import torch
import torch.distributed as dist

from rslo_amd import capi

MAX_N = 1024
# how long an exchange waits for a peer before it poisons the result with NaN and raises the comm's status word (a rank held
# up by its data loader, a first-use module load, a debugger ...): the process-group scale, not a kernel scale
TIMEOUT_MS = int(os.environ.get("RSLO_PEER_TIMEOUT_MS", "600000"))
_COMMS = {}          # id of the default group object -> PeerComm | False (not eligible)


class PeerComm:
    def __init__(self, handle, rank, world, transport, shared_device=False):
        self.handle, self.rank, self.world, self.transport = handle, rank, world, transport
        # two ranks of the comm live on ONE GPU (development boxes): kernels whose every workgroup spins for its peer
        # (rslo_bn2d_fwd_peer) then need both processes' grids resident side by side, which nothing guarantees -- a full
        # GPU of waiting workgroups starves the peer's kernel until the timeout.  The single-workgroup exchange kernel is
        # fine there; the fused SyncBN path is only taken when every rank has its own device.
        self.shared_device = shared_device

    def all_reduce_(self, t):
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() <= MAX_N
        capi._chk(capi.lib().rslo_peer_allreduce_f64(self.handle, capi._ptr(t), t.numel(), capi._stream()),
                  "rslo_peer_allreduce_f64")
        return t

    def status(self):
        """(0, -1) while every exchange met its peers, else (sequence number of the first time-out, missing rank);
        reads pinned memory, lags the stream."""
        peer = C.c_int(-1)
        s = capi.lib().rslo_peer_status(self.handle, C.byref(peer))
        return int(s), int(peer.value)

    def check(self):
        s, peer = self.status()
        if s:
            raise capi.RsloHipError("peer exchange %d timed out waiting for rank %d (results were poisoned with NaN)" % (s, peer))

    def wait_stats(self):
        """p50 / p99 / max microseconds the recent exchanges (<= 4096) spent waiting for their slowest peer, and how many
        samples that is over; call after a synchronisation."""
        buf = (C.c_float * 4096)()
        n = capi.lib().rslo_peer_wait_samples(self.handle, buf, 4096)
        if n <= 0:
            return {"samples": 0}
        v = sorted(buf[:n])
        return {"samples": n, "p50_us": round(v[n // 2], 2), "p99_us": round(v[min(n - 1, (99 * n) // 100)], 2),
                "max_us": round(v[-1], 2)}

    # ---- exchanges inside a replayed stream capture (rslo_amd/headgraph.py) -------------------------------------------------
    def capture_begin(self):
        """Exchanges launched from here to capture_end() carry numbers relative to a device word (replay-invariant launches)."""
        capi._chk(capi.lib().rslo_peer_capture_begin(self.handle), "rslo_peer_capture_begin")

    def capture_end(self):
        """-> how many exchanges the capture holds."""
        n = C.c_int(0)
        capi._chk(capi.lib().rslo_peer_capture_end(self.handle, C.byref(n)), "rslo_peer_capture_end")
        return int(n.value)

    def replay_prepare(self, n):
        """In front of every replay of a capture with n exchanges, on the replaying (current) stream."""
        capi._chk(capi.lib().rslo_peer_replay_prepare(self.handle, int(n), capi._stream()), "rslo_peer_replay_prepare")

    def set_timeout_ms(self, ms):
        capi._chk(capi.lib().rslo_peer_set_timeout_ms(self.handle, int(ms)), "rslo_peer_set_timeout_ms")

    def close(self):
        if self.handle is not None:
            capi.lib().rslo_peer_destroy(self.handle)
            self.handle = None


def create(transport="host", group=None):
    """Collective over `group` (default group): every rank calls it at the same point.  Returns a PeerComm, or None when
    the ranks are not all on this host or the transport cannot be set up on some rank (the decision is agreed: all ranks
    get a comm or none does)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    hosts = [None] * world
    pr = torch.cuda.get_device_properties(torch.cuda.current_device())
    dev_id = tuple(getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    if dev_id == (None, None, None):
        dev_id = ("index", torch.cuda.current_device())
    dist.all_gather_object(hosts, (socket.gethostname(), dev_id), group=group)
    shared_device = len({h[1] for h in hosts}) < world
    if len({h[0] for h in hosts}) != 1 or world > 16:
        return None
    lib = capi.lib()
    handle = C.c_void_p()
    ok, comm = 1, None
    try:
        if transport == "host":
            name = [None]
            if rank == 0:
                name[0] = "/rslo_peer_%d_%s" % (os.getpid(), os.urandom(4).hex())
            dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            capi._chk(lib.rslo_peer_create_host(name[0].encode(), rank, world, MAX_N, C.byref(handle)), "rslo_peer_create_host")
        elif transport == "device":
            nb = lib.rslo_peer_ipc_handle_bytes()
            buf = C.create_string_buffer(nb)
            capi._chk(lib.rslo_peer_create_device_begin(rank, world, MAX_N, C.byref(handle), buf), "rslo_peer_create_device_begin")
            allh = [None] * world
            dist.all_gather_object(allh, bytes(buf.raw), group=group)
            capi._chk(lib.rslo_peer_create_device_finish(handle, b"".join(allh)), "rslo_peer_create_device_finish")
        else:
            raise ValueError("unknown peer transport %r" % (transport,))
        comm = PeerComm(handle, rank, world, transport, shared_device)
    except Exception as e:      # agreed below: one rank failing turns the exchange off everywhere
        ok = 0
        err = e
        if comm is None and handle.value:      # begun but not finished (a peer's handle would not open): give it back
            lib.rslo_peer_destroy(handle)
    flags = [None] * world
    dist.all_gather_object(flags, ok, group=group)
    if not all(flags):
        if comm is not None:
            comm.close()
        if not ok and os.environ.get("RSLO_SYNCBN_EXCHANGE", "auto") in ("host", "device"):
            raise err          # an explicitly requested transport that cannot be had is an error, not a silent fallback
        return None
    dist.barrier(group=group)   # every rank has the segment mapped before rank 0 may ever unlink it / anyone sends
    if transport == "host" and rank == 0:
        lib.rslo_peer_host_unlink(handle)      # the name goes now: a rank that dies later leaves nothing in /dev/shm
    # Self-test before the transport carries statistics: a few exchanges of rank-dependent values with a short timeout,
    # checked on the host on every rank.  A transport that maps but does not deliver on this machine (peer access, coherence
    # of the mapping, kernels of the ranks not co-scheduled) is dropped HERE, by agreement, and the next one is tried.
    good = 1
    try:
        comm.set_timeout_ms(3000)
        tri = world * (world + 1) // 2
        for n in (3, 513, MAX_N, 129):
            t = torch.arange(n, dtype=torch.float64, device="cuda") * float(rank + 1) + float(rank)
            comm.all_reduce_(t)
            want = torch.arange(n, dtype=torch.float64) * float(tri) + float(tri - world)
            if not torch.equal(t.cpu(), want):
                good = 0
        if comm.status()[0]:
            good = 0
        # the single-launch SyncBN kernels meet their peers per channel (rslo_bn2d_fwd_peer): rank-dependent constants
        # per channel, whose mean over the ranks is known
        fused_on = os.environ.get("RSLO_SYNCBN_FUSED_PEER", "1") != "0"
        for Cc, hw in (((8, 16), (64, 1056), (512, 4)) if not shared_device else ((8, 16),)):
            if not (fused_on and capi.bn2d_peer_supported(2, Cc, hw)):
                continue          # the fused kernels are switched off / do not take the shape: nothing of theirs to test
            x = (torch.arange(Cc, dtype=torch.float32, device="cuda").view(1, Cc, 1, 1) + float(rank + 1)).expand(2, Cc, hw, 1).contiguous()
            y, mean, invstd, cnt = capi.bn2d_fwd_peer(comm, x, None, None, None, None, None, 0.1, 1e-5, 1.0)
            want = torch.arange(Cc, dtype=torch.float32) + float(tri) / world
            if not torch.allclose(mean.cpu(), want, rtol=0, atol=1e-5) or float(cnt.item()) != 2.0 * hw * world:
                good = 0
        if comm.status()[0]:
            good = 0
        comm.set_timeout_ms(TIMEOUT_MS)
    except Exception:
        good = 0
    flags = [None] * world
    dist.all_gather_object(flags, good, group=group)
    if not all(flags):
        comm.close()
        if os.environ.get("RSLO_SYNCBN_EXCHANGE", "auto") in ("host", "device"):
            raise capi.RsloHipError("peer exchange (%s transport) failed its self-test on ranks %s" % (
                transport, [r for r, f in enumerate(flags) if not f]))
        return None
    return comm


def comm_for(group):
    """The PeerComm of the default process group (created on first use: a collective call -- every rank reaches its
    first SyncBN forward at the same point), or None: explicit sub-groups, RSLO_SYNCBN_EXCHANGE=rccl, several hosts."""
    mode = os.environ.get("RSLO_SYNCBN_EXCHANGE", "auto")
    if group is not None or mode == "rccl" or not (dist.is_available() and dist.is_initialized()):
        return None
    key = id(dist.group.WORLD)
    ent = _COMMS.get(key)
    if ent is None or ent[0] is not dist.group.WORLD:
        for k in list(_COMMS):               # a destroyed / re-made default group: the old comm is dead
            old = _COMMS.pop(k)
            if old[1]:
                old[1].close()
        if dist.get_world_size() < 2 and os.environ.get("RSLO_FORCE_SYNCBN_PATH", "0") != "1":
            c = None
        elif mode in ("host", "device"):
            c = create(mode, None)
        else:      # auto: slices in HBM read over xGMI if HIP IPC can be set up on every rank, else the host segment, else RCCL
            c = create("device", None) or create("host", None)
        ent = _COMMS[key] = (dist.group.WORLD, c)
    return ent[1]


def all_reduce_(t, group=None, fallback_group=None):
    """Sum of t (float64 CUDA, <= 1024 elements) over the ranks, in place, on the current stream."""
    c = comm_for(group) if (t.is_cuda and t.dtype == torch.float64 and t.numel() <= MAX_N) else None
    if c is None:
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            # (rslo_amd/headgraph.py then keeps the head on the eager pass)
            raise capi.RsloHipError("a SyncBN statistics exchange that needs an RCCL collective cannot go into a stream capture")
        dist.all_reduce(t, group=fallback_group if fallback_group is not None else group)
    else:
        c.all_reduce_(t)
    return t


def check_all():
    """Raise if any exchange of any live comm timed out (pinned-memory read, no synchronisation): the product path calls this
    once per step (apex DDP stand-in, bench.py), so a poisoned statistic is an error within a step, never a silent NaN."""
    for ent in _COMMS.values():
        if ent[1]:
            ent[1].check()


def shutdown():
    """Close every live comm.  A collective call: no rank may free or unmap its slice while a peer's exchange kernel can still
    be polling it, so each rank first drains its own device and the ranks meet at a barrier (skipped if the process group is
    already gone: the caller then vouches that every rank is past its last exchange)."""
    if any(ent[1] for ent in _COMMS.values()):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            try:
                dist.barrier()
            except Exception:
                pass
    for k in list(_COMMS):
        ent = _COMMS.pop(k)
        if ent[1]:
            ent[1].close()
