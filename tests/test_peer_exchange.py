"""rslo_peer_* (csrc/peer.hip, rslo_amd/peer.py): the same-stream SyncBN statistics exchange between the ranks of a node.
The test boxes have ONE GPU: two (three) processes share it -- the host transport is exactly what a multi-GPU node runs
(a registered shared-memory segment), the device transport opens HIP IPC handles on the same device."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rslo_amd  # noqa: F401

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, transport, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rslo_amd import peer
    comm = peer.create(transport)
    if comm is None:
        q.put((rank, "no-comm"))
        dist.barrier()
        dist.destroy_process_group()
        return
    out = {"transport": comm.transport}
    try:
        if mode == "sums":
            # 300 exchanges of the sizes the BEV head uses (2C + 1 / 2C doubles, C in 1..256 .. 512), with one rank lagging at
            # random points: the result must be the rank-ordered sum, identical bits on every rank, for every exchange
            rng = np.random.default_rng(5)
            sizes = [int(rng.choice([3, 15, 65, 129, 257, 513, 1024, 2, 64, 128, 256, 512])) for _ in range(300)]
            got, want = [], []
            comm.set_timeout_ms(3000)                 # a box that cannot run both processes' kernels side by side fails fast
            probe = torch.ones(3, dtype=torch.float64, device="cuda")
            comm.all_reduce_(probe)
            torch.cuda.synchronize()
            comm.check()
            assert probe.tolist() == [float(world)] * 3
            for k, n in enumerate(sizes):
                mine = torch.from_numpy(np.random.default_rng(1000 * k + rank).standard_normal(n)).cuda()
                ref = sum(torch.from_numpy(np.random.default_rng(1000 * k + r).standard_normal(n)) for r in range(world))
                if (k * 7 + rank) % 23 == 0:
                    torch.cuda.synchronize()
                    time.sleep(0.02)                  # this rank arrives late: the others spin
                t = mine.clone()
                comm.all_reduce_(t)
                got.append(t)
                want.append(ref)
            torch.cuda.synchronize()
            comm.check()
            out["exact"] = all(torch.equal(g.cpu(), w) for g, w in zip(got, want))
            # the collective it replaces gives the same numbers (two ranks: a + b in either order)
            t2 = torch.from_numpy(np.random.default_rng(rank).standard_normal(257)).cuda()
            t3 = t2.clone()
            comm.all_reduce_(t2)
            dist.all_reduce(t3)
            out["vs_collective"] = float((t2 - t3).abs().max())
            # latency of one exchange between ranks that arrive together (the 90-per-step cost)
            xs = torch.zeros(513, dtype=torch.float64, device="cuda")
            for _ in range(20):
                comm.all_reduce_(xs)
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                comm.all_reduce_(xs)
            e1.record()
            torch.cuda.synchronize()
            out["us_per_exchange"] = e0.elapsed_time(e1) * 1e3 / 200
            comm.check()
        elif mode == "fusedbn":
            # single-launch SyncBN (rslo_bn2d_fwd_peer / _bwd_peer: the channel's workgroup meets its peers inside the
            # kernel) against statistics kernel -> exchange kernel -> apply kernel, on the three register-cached map sizes
            # of the head, with one rank lagging; and a late rank in front of every second layer
            import apex.parallel as AP
            AP.FORCE_FUSED_ON_SHARED_DEVICE = True        # two ranks on ONE GPU here: small layers, both grids fit side by side
            comm.set_timeout_ms(5000)
            peer._COMMS[id(dist.group.WORLD)] = (dist.group.WORLD, comm)       # the comm under test is the group's comm
            outs = {}
            for name, (Cc, H, W) in {"12x22": (256, 12, 22), "24x44": (128, 24, 44), "48x88": (64, 48, 88)}.items():
                g = torch.Generator().manual_seed(17)
                full = torch.randn(4, Cc, H, W, generator=g) * 1.5 + 0.3
                res_full = torch.randn(4, Cc, H, W, generator=g)
                gy_full = torch.randn(4, Cc, H, W, generator=g)
                sl = slice(2 * rank, 2 * rank + 2) if world == 2 else slice(rank, rank + 1)
                got = []
                for fused in (True, False):
                    AP.FUSED_PEER_BN = fused
                    x = full[sl].cuda().requires_grad_(True)
                    res = res_full[sl].cuda().requires_grad_(True)
                    bn = AP.SyncBatchNorm(Cc, eps=1e-3, momentum=0.01).cuda().train()
                    if rank == 1:
                        torch.cuda.synchronize()
                        time.sleep(0.01)
                    y = bn(x, act_slope=0.0, residual=res)
                    (y * gy_full[sl].cuda()).sum().backward()
                    got.append([t.detach().cpu() for t in (y, x.grad, res.grad, bn.weight.grad, bn.bias.grad, bn.running_mean,
                                                            bn.running_var)])
                AP.FUSED_PEER_BN = True
                torch.cuda.synchronize()
                comm.check()
                outs[name] = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(got[0], got[1]))
                # every rank normalises with the same statistics, bit for bit
                rm = got[0][5].cuda().double()
                both = [torch.zeros_like(rm) for _ in range(world)]
                dist.all_gather(both, rm)
                outs[name + "_same_bits"] = all(torch.equal(both[0], b) for b in both)
            out["fusedbn"] = outs
            # latency of one fused layer forward (256 channels at 12x22, both ranks arriving together) against the three launches
            xs = torch.randn(2, 256, 12, 22, device="cuda")
            for fused in (True, False):
                AP.FUSED_PEER_BN = fused
                bn = AP.SyncBatchNorm(256, eps=1e-3, momentum=0.01).cuda().train()
                with torch.no_grad():
                    for _ in range(10):
                        AP.fused_bn_forward(bn, xs, None, bn.weight, bn.bias, 0.0, None, world)
                    torch.cuda.synchronize()
                    dist.barrier()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(100):
                        AP.fused_bn_forward(bn, xs, None, bn.weight, bn.bias, 0.0, None, world)
                    e1.record()
                    torch.cuda.synchronize()
                out["us_fused" if fused else "us_three_launches"] = e0.elapsed_time(e1) * 1e3 / 100
            AP.FUSED_PEER_BN = True
            peer._COMMS.pop(id(dist.group.WORLD), None)
            comm.check()
        elif mode == "captured":
            # exchanges inside a REPLAYED stream capture (rslo_peer_capture_begin / _end / _replay_prepare: numbers relative to a
            # device word): two plain exchanges + one single-launch SyncBN layer captured once, replayed 8 times with fresh
            # inputs, eager exchanges in between, one rank lagging -- every result is the rank-ordered sum / the full-batch mean
            from rslo_amd import capi
            import apex.parallel as AP
            AP.FORCE_FUSED_ON_SHARED_DEVICE = True
            comm.set_timeout_ms(5000)
            ta = torch.zeros(65, dtype=torch.float64, device="cuda")
            tb = torch.zeros(513, dtype=torch.float64, device="cuda")
            xs = torch.zeros(2, 8, 4, 4, device="cuda")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            ok, details = True, []
            with torch.cuda.stream(side):
                for _ in range(2):           # (eager first: lazy state of the launch code)
                    comm.all_reduce_(ta.clone())
                    capi.bn2d_fwd_peer(comm, xs, None, None, None, None, None, 0.1, 1e-5, 1.0)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                comm.capture_begin()
                try:
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        comm.all_reduce_(ta)
                        comm.all_reduce_(tb)
                        y, mean, invstd, cnt = capi.bn2d_fwd_peer(comm, xs, None, None, None, None, None, 0.1, 1e-5, 1.0)
                finally:
                    n = comm.capture_end()
                out["n_captured"] = n

                def vals(k, r, m):
                    return torch.from_numpy(np.random.default_rng(977 * k + 13 * r + m).standard_normal(m))
                for k in range(8):
                    ta.copy_(vals(k, rank, 65))
                    tb.copy_(vals(k, rank, 513))
                    xs.copy_(vals(k, rank, 256).float().view(2, 8, 4, 4))
                    if (k + rank) % 3 == 0:
                        torch.cuda.synchronize()
                        time.sleep(0.02)
                    comm.replay_prepare(n)
                    g.replay()
                    e = vals(k, rank, 129).cuda()
                    comm.all_reduce_(e)                                   # an eager exchange between two replays
                    torch.cuda.synchronize()
                    wa = sum(vals(k, r, 65) for r in range(world))
                    wb = sum(vals(k, r, 513) for r in range(world))
                    we = sum(vals(k, r, 129) for r in range(world))
                    full = torch.cat([vals(k, r, 256).float().view(2, 8, 4, 4) for r in range(world)], 0)
                    wm = full.double().mean(dim=(0, 2, 3)).float()
                    good = (torch.equal(ta.cpu(), wa) and torch.equal(tb.cpu(), wb) and torch.equal(e.cpu(), we)
                            and torch.allclose(mean.cpu(), wm, rtol=0, atol=1e-6) and float(cnt.item()) == 32.0 * world)
                    ok = ok and good
                    details.append(good)
            torch.cuda.current_stream().wait_stream(side)
            comm.check()
            out["captured_ok"], out["details"] = ok, details
        elif mode == "timeout":
            comm.set_timeout_ms(300)
            t = torch.ones(65, dtype=torch.float64, device="cuda")
            if rank == 0:
                comm.all_reduce_(t)                   # rank 1 never joins this exchange
                torch.cuda.synchronize()
                out["poisoned"] = bool(torch.isnan(t).all())
                out["status"] = comm.status()
            dist.barrier()
    finally:
        q.put((rank, out))
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


def _run(world, transport, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world", [2, 3])
def test_host_transport_sums_in_rank_order_under_skew(world):
    res = _run(world, "host", "sums")
    for r in range(world):
        assert res[r] != "no-comm"
        assert res[r]["exact"], res
        assert res[r]["vs_collective"] <= (0.0 if world == 2 else 1e-15), res
    print("host transport, %d ranks on one GPU: %.1f us per exchange of 513 doubles" % (world, res[0]["us_per_exchange"]))


def test_device_transport_over_hip_ipc():
    """Each rank's slice in its own device memory, opened by the peer through hipIpcOpenMemHandle (same device here; xGMI
    peers on a node).  If the runtime refuses IPC on this box the creation reports it and both ranks agree to fall back."""
    res = _run(2, "device", "sums")
    if res[0] == "no-comm":
        assert res[1] == "no-comm"
        pytest.skip("HIP IPC not available between two processes on this box")
    for r in range(2):
        assert res[r]["exact"] and res[r]["vs_collective"] == 0.0, res
    print("device transport: %.1f us per exchange of 513 doubles" % res[0]["us_per_exchange"])


def test_missing_peer_times_out_instead_of_hanging_the_gpu():
    res = _run(2, "host", "timeout")
    assert res[0]["poisoned"] is True
    seq, peer_rank = res[0]["status"]
    assert seq == 6 and peer_rank == 1          # five self-test exchanges at creation (four plain, one fused SyncBN layer: the ranks share a GPU), then the one rank 1 never joined


@pytest.mark.parametrize("transport", ["host", "device"])
def test_exchanges_inside_a_replayed_capture_keep_their_numbers(transport):
    """Round 6 (capturable SyncBN): exchanges launched between capture_begin() and capture_end() read their number as
    device word + k; replay_prepare() sets the word in front of every replay.  Eight replays of one hipGraph holding two
    plain exchanges and one single-launch SyncBN layer, eager exchanges in between, a lagging rank: all sums exact."""
    res = _run(2, transport, "captured")
    if res[0] == "no-comm":
        assert res[1] == "no-comm"
        pytest.skip("transport %s not available between two processes on this box" % transport)
    for r in range(2):
        assert res[r]["n_captured"] == 3, res
        assert res[r]["captured_ok"] is True, res


@pytest.mark.parametrize("transport", ["host", "device"])
def test_single_launch_syncbn_meets_its_peers_inside_the_kernel(transport):
    """rslo_bn2d_fwd_peer / _bwd_peer (round 5): two ranks on one GPU, each half a batch.  Same results as the three-launch
    path (statistics -> rslo_peer_allreduce_f64 -> apply) to rounding of the double sums, identical statistics on both
    ranks, no time-out with a lagging rank; prints what a layer costs either way."""
    res = _run(2, transport, "fusedbn")
    if res[0] == "no-comm":
        assert res[1] == "no-comm"
        pytest.skip("transport %s not available between two processes on this box" % transport)
    for r in range(2):
        for k, v in res[r]["fusedbn"].items():
            if k.endswith("_same_bits"):
                assert v is True, (r, k)
            else:
                assert v < 2e-6, (r, k, v)
    print("%s transport, SyncBN forward of 256 ch at 12x22 on 2 ranks: %.1f us in one launch, %.1f us as three launches"
          % (transport, res[0]["us_fused"], res[0]["us_three_launches"]))
