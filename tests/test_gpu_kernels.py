"""GPU parity: every HIP entry point (through the C ABI) against the CPU oracle on the same inputs.

Bar: bit-exact for integer/index work (voxel ids, coordinates, neighbour tables, chamfer idx) and for
the chamfer distances; fp32 tolerance (stated per test) for the convolution arithmetic.
"""
import os

import numpy as np
import pytest
import torch

import oracle as O
from rslo_amd import synthetic as S

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rand_sites(rng, batch, dims, n):
    vol = batch * dims[0] * dims[1] * dims[2]
    lin = rng.choice(vol, size=min(n, vol), replace=False)
    x = lin % dims[2]; r = lin // dims[2]
    y = r % dims[1]; r //= dims[1]
    z = r % dims[0]; b = r // dims[0]
    return np.stack([b, z, y, x], 1).astype(np.int32)


# ------------------------------------------------------------------------------- voxelizer
def _vox_both(hip, pts, rng_, vs, T, maxv):
    g = O.grid_size(rng_, vs)
    v, c, n, nv = hip.voxelize(dev(pts), rng_, vs, g, T, maxv)
    M = int(nv.item())
    ov, oc, on = O.voxelize(pts, rng_, vs, T, maxv)
    assert M == len(oc)
    assert (c[:M].cpu().numpy() == oc).all()
    assert (n[:M].cpu().numpy() == on).all()
    assert (v[:M].cpu().numpy() == ov).all()
    assert float(v[M:].abs().sum()) == 0.0 and int(n[M:].sum()) == 0
    return v[:M], c[:M], n[:M]


def test_voxelize_small_cases(hip):
    rng = np.random.default_rng(0)
    pts = (rng.random((5000, 7)) * np.array([10, 8, 4, 1, 1, 1, 1]) - np.array([1, 1, 0.5, 0, 0, 0, 0])).astype(np.float32)
    _vox_both(hip, pts, [0, 0, 0, 8, 6, 3], [0.5, 0.5, 0.5], 3, 10000)   # T truncation, out-of-range points
    _vox_both(hip, pts, [0, 0, 0, 8, 6, 3], [0.5, 0.5, 0.5], 5, 100)     # max_voxels break
    _vox_both(hip, pts[:1], [0, 0, 0, 8, 6, 3], [0.5, 0.5, 0.5], 5, 100)
    _vox_both(hip, pts[:0], [0, 0, 0, 8, 6, 3], [0.5, 0.5, 0.5], 5, 100)  # empty cloud


def test_voxelize_kitti_shaped_scan_and_vfe(hip):
    pts = S.scan()
    v, c, n = _vox_both(hip, pts, S.PC_RANGE, S.VOXEL_SIZE, S.MAX_POINTS_PER_VOXEL, S.MAX_VOXELS)
    assert len(c) > 30000
    f = hip.vfe_mean(v, n).cpu().numpy()
    np.testing.assert_allclose(f, O.vfe_mean(v.cpu().numpy(), n.cpu().numpy()), rtol=2e-6, atol=2e-6)
    # cap hit: 20000 voxels
    _vox_both(hip, pts, S.PC_RANGE, S.VOXEL_SIZE, S.MAX_POINTS_PER_VOXEL, 20000)


def test_host_voxelizer_equals_device_voxelizer(hip):
    """The two faces of spconv.utils.VoxelGenerator.generate -- numpy in (librslo_host.so, what forked DataLoader
    workers call) and CUDA tensor in (rslo_voxelize) -- return the same bits on full scans: 64-ring scan, the 20000-voxel
    cap (the loop's break), the dense 128-ring scan at 0.1 m voxels (C5)."""
    from spconv.utils import VoxelGenerator
    for pts, vs, maxv in ((S.scan(), S.VOXEL_SIZE, S.MAX_VOXELS), (S.scan(), S.VOXEL_SIZE, 20000),
                          (S.scan(n_el=128), [0.1, 0.1, 0.1], 1 << 18)):
        vg = VoxelGenerator(vs, S.PC_RANGE, S.MAX_POINTS_PER_VOXEL, maxv)
        hv, hc, hn = vg.generate(pts, maxv)
        assert isinstance(hv, np.ndarray)
        dv, dc, dn = vg.generate(torch.from_numpy(pts).cuda(), maxv)
        assert dv.is_cuda
        assert hv.shape == tuple(dv.shape) and len(hc) > 19000
        assert (hv == dv.cpu().numpy()).all() and (hc == dc.cpu().numpy()).all() and (hn == dn.cpu().numpy()).all()


# ------------------------------------------------------------------------------- rulebooks
def _encoder_levels(hip, coords, batch, dims):
    """Runs the rulebook chain of SpMiddleFHDWithCov2_3 on both sides; yields per-level data."""
    specs = [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]),
             ([3, 3, 3], [2, 2, 2], [0, 1, 1]), ([3, 1, 1], [2, 1, 1], [0, 0, 0])]
    idx = hip.SiteIndex(dev(coords), batch, dims)
    oc = coords
    od = list(dims)
    for li, (ks, st, pd) in enumerate(specs):
        if li < 4:
            nbr = hip.rulebook_subm(idx, [3, 3, 3]).cpu().numpy()
            assert (nbr == O.rulebook_subm(oc, batch, od)).all(), "subm table level %d" % li
        oidx, nb, nbT = hip.rulebook_conv(idx, ks, st, pd)
        oc2, od2, onb, onbT = O.rulebook_conv(oc, batch, od, ks, st, pd)
        assert oidx.dims == od2
        assert (oidx.coords.cpu().numpy() == oc2).all(), "out coords level %d" % li
        assert (nb.cpu().numpy() == onb).all() and (nbT.cpu().numpy() == onbT).all()
        idx, oc, od = oidx, oc2, od2
    return oc, od


def test_rulebooks_bitexact_random_sites(hip):
    rng = np.random.default_rng(1)
    dims = [17, 40, 36]
    coords = rand_sites(rng, 3, dims, 3000)
    _encoder_levels(hip, coords, 3, dims)


def test_rulebooks_bitexact_kitti_scan_batch2(hip):
    cs = []
    for b, seed in enumerate((0, 1)):
        _, c, _ = O.voxelize(S.scan(scan_seed=seed), S.PC_RANGE, S.VOXEL_SIZE, 10, 40000)
        cs.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    coords = np.concatenate(cs, 0)
    oc, od = _encoder_levels(hip, coords, 2, [41, 768, 1408])
    assert od == [2, 96, 176]


def _row_order_ref(nbr, flip):
    """numpy restatement of rslo_rulebook_row_order: windows of 2048 rows, ascending (mask, row)."""
    n, K = nbr.shape
    bits = (K - 1 - np.arange(K)) if flip else np.arange(K)
    mask = ((nbr >= 0).astype(np.int64) << bits).sum(1)
    out = np.empty(n, np.int32)
    for s in range(0, n, 2048):
        m = mask[s:s + 2048]
        out[s:s + 2048] = s + np.lexsort((np.arange(len(m)), m))
    return out


@pytest.mark.parametrize("flip", [False, True])
def test_row_order_is_the_windowed_mask_sort(hip, flip):
    rng = np.random.default_rng(5)
    dims = [9, 60, 70]
    coords = rand_sites(rng, 2, dims, 7000)          # 3.4 windows, ragged tail
    idx = hip.SiteIndex(dev(coords), 2, dims)
    nbr = hip.rulebook_subm(idx, [3, 3, 3])
    order = hip.rulebook_row_order(nbr, flip_k=flip).cpu().numpy()
    assert (np.sort(order) == np.arange(len(coords))).all()
    assert (order == _row_order_ref(nbr.cpu().numpy(), flip)).all()
    # K = 3 tables (the (3,1,1) convolution) and an empty table
    _, nb3, _ = hip.rulebook_conv(idx, [3, 1, 1], [2, 1, 1], [0, 0, 0])
    assert (hip.rulebook_row_order(nb3, flip_k=flip).cpu().numpy() == _row_order_ref(nb3.cpu().numpy(), flip)).all()
    assert hip.rulebook_row_order(nbr[:0], flip_k=flip).shape == (0,)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (64, 64), (32, 64), (16, 32)])
def test_conv_results_do_not_depend_on_row_order(hip, cin, cout):
    """The order only regroups rows into tiles: forward, data gradient (flip_k) and the bf16 path give the same BITS
    with the mask-sorted order, with a random permutation and without one (ragged sizes: 16/32-row tile tails)."""
    rng = np.random.default_rng(cin + cout)
    dims = [9, 60, 70]
    for n in (5000, 8200 + 17):
        coords = rand_sites(rng, 2, dims, n)
        idx = hip.SiteIndex(dev(coords), 2, dims)
        nbr = hip.rulebook_subm(idx, [3, 3, 3])
        x = torch.randn(len(coords), cin, device="cuda")
        W = torch.randn(27, cin, cout, device="cuda") * 0.1
        b = torch.randn(cout, device="cuda")
        g = torch.randn(len(coords), cout, device="cuda")
        orders = [hip.rulebook_row_order(nbr), hip.rulebook_row_order(nbr, flip_k=True),
                  torch.randperm(len(coords), device="cuda").int()]
        y0 = hip.spconv_fwd(x, W, b, nbr, act_slope=0.01)
        d0 = hip.spconv_dgrad(g, W, nbr, flip_k=True)
        for o in orders:
            assert torch.equal(hip.spconv_fwd(x, W, b, nbr, act_slope=0.01, order=o), y0)
            assert torch.equal(hip.spconv_dgrad(g, W, nbr, flip_k=True, order=o), d0)
        if cin >= 32 and cout >= 32:
            xb = x.to(torch.bfloat16)
            yb = hip.spconv_fwd_bf16(xb, W, b, nbr, act_slope=0.01)
            assert torch.equal(hip.spconv_fwd_bf16(xb, W, b, nbr, act_slope=0.01, order=orders[0]), yb)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (32, 64), (64, 32)])
def test_offset_split_tilings_agree(hip, cin, cout):
    """rslo_spconv_set_tiling: every (rows per tile, waves per tile) form of k_spconv_v6 computes the same products; only
    the order in which the per-offset sums meet differs (ks waves are added through LDS in wave order).  All forms
    against float64 within the fp32 bar of the kernel, each form run-to-run identical, ragged and tiny sizes included."""
    rng = np.random.default_rng(cin * 5 + cout)
    dims = [9, 60, 70]
    L = hip.lib()
    try:
        for n in (1, 31, 5000, 8200 + 17):
            coords = rand_sites(rng, 2, dims, n)
            idx = hip.SiteIndex(dev(coords), 2, dims)
            nbr = hip.rulebook_subm(idx, [3, 3, 3])
            x = torch.randn(len(coords), cin, device="cuda")
            W = torch.randn(27, cin, cout, device="cuda") * 0.1
            b = torch.randn(cout, device="cuda")
            order = hip.rulebook_row_order(nbr)
            # float64 reference: gather-matmul per offset
            nb = nbr.long()
            xz = torch.cat([x.double(), torch.zeros(1, cin, dtype=torch.float64, device="cuda")])
            ref = b.double()[None].repeat(len(coords), 1)
            for k in range(27):
                ref += xz[torch.where(nb[:, k] >= 0, nb[:, k], torch.full_like(nb[:, k], len(coords)))] @ W[k].double()
            scale = ref.abs().max().item() + 1e-30
            bf16_ref = None
            for rbw, ks in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (2, 4), (4, 1), (0, 0)):
                L.rslo_spconv_set_tiling(rbw, ks)
                y = hip.spconv_fwd(x, W, b, nbr)
                assert (y.double() - ref).abs().max().item() <= 2e-6 * scale, (rbw, ks, n)
                assert torch.equal(hip.spconv_fwd(x, W, b, nbr), y), (rbw, ks, n)
                yo = hip.spconv_fwd(x, W, b, nbr, order=order)
                assert (yo.double() - ref).abs().max().item() <= 2e-6 * scale, (rbw, ks, n, "order")
                if rbw != 4:       # the bf16 feature kernel (C4) takes the same tilings: bf16 rows, fp32 accumulation
                    xb = x.to(torch.bfloat16)
                    yb = hip.spconv_fwd_bf16(xb, W, b, nbr)
                    if bf16_ref is None:
                        bf16_ref = yb
                    # the partial sums meet in a different order: one bf16 ulp of the rounded output at most
                    assert (yb.float() - bf16_ref.float()).abs().max().item() <= 2 ** -7 * scale, (rbw, ks, n, "bf16")
                    assert torch.equal(hip.spconv_fwd_bf16(xb, W, b, nbr), yb)
    finally:
        L.rslo_spconv_set_tiling(0, 0)


def test_rulebook_empty_input(hip):
    idx = hip.SiteIndex(torch.zeros((0, 4), dtype=torch.int32, device="cuda"), 1, [5, 8, 8])
    assert hip.rulebook_subm(idx, [3, 3, 3]).shape == (0, 27)
    oidx, nb, nbT = hip.rulebook_conv(idx, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert oidx.coords.shape == (0, 4) and nb.shape == (0, 27) and nbT.shape == (0, 27)


# ------------------------------------------------------------------------------- conv arithmetic
CONV_TOL = dict(rtol=2e-5, atol=2e-5)   # fp32 MFMA accumulation vs the oracle's double accumulation

CHANNELS = [(7, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 32), (32, 16), (16, 7)]


@pytest.mark.parametrize("cin,cout", CHANNELS)
def test_subm_conv_fwd_bwd(hip, cin, cout):
    rng = np.random.default_rng(10 + cin + cout)
    dims, B = [9, 30, 28], 2
    coords = rand_sites(rng, B, dims, 2500)   # ~16 % occupancy: every offset occurs
    x = rng.normal(size=(len(coords), cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    gy = rng.normal(size=(len(coords), cout)).astype(np.float32)
    nbr = O.rulebook_subm(coords, B, dims)
    dn = dev(nbr)
    y = hip.spconv_fwd(dev(x), dev(W), dev(b), dn, act_slope=0.01).cpu().numpy()
    yo = O.spconv_fwd(x, W, b, nbr)
    yo = np.where(yo > 0, yo, yo * np.float32(0.01))
    np.testing.assert_allclose(y, yo, **CONV_TOL)
    y2 = hip.spconv_fwd(dev(x), dev(W), None, dn).cpu().numpy()
    np.testing.assert_allclose(y2, O.spconv_fwd(x, W, None, nbr), **CONV_TOL)
    gx = hip.spconv_dgrad(dev(gy), dev(W), dn, flip_k=True).cpu().numpy()
    np.testing.assert_allclose(gx, O.spconv_dgrad(gy, W, nbr[:, ::-1].copy()), **CONV_TOL)
    # the dedicated entry point (weights read in place) and the forward kernel on transposed weights agree
    gx2 = hip.spconv_dgrad_direct(dev(gy), dev(W), dn, flip_k=True).cpu().numpy()
    np.testing.assert_allclose(gx2, gx, rtol=1e-5, atol=1e-5)
    Wt = hip.weight_transpose(dev(W)).cpu().numpy()
    assert (Wt == np.ascontiguousarray(W.transpose(0, 2, 1))).all()
    gw, gb = hip.spconv_wgrad(dev(x), dev(gy), dn, cin, cout)
    ow, ob = O.spconv_wgrad(x, gy, nbr, cin, cout)
    np.testing.assert_allclose(gw.cpu().numpy(), ow, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gb.cpu().numpy(), ob, rtol=1e-4, atol=1e-4)
    # pair-list rulebook (spconv layout, ascending output row per offset): bit-exact, and the wgrad over it
    pin, pout, koff = hip.rulebook_pairs(dn)
    P = int(koff[-1].item())
    assert P == int((nbr >= 0).sum())
    ko = koff.cpu().numpy()
    for k in (0, 13, 26):
        rows = np.nonzero(nbr[:, k] >= 0)[0]
        assert (pout[ko[k]:ko[k + 1]].cpu().numpy() == rows).all()
        assert (pin[ko[k]:ko[k + 1]].cpu().numpy() == nbr[rows, k]).all()
    gw2, gb2 = hip.spconv_wgrad_pairs(dev(x), dev(gy), (pin, pout, koff), len(coords), 27, cin, cout)
    np.testing.assert_allclose(gw2.cpu().numpy(), ow, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gb2.cpu().numpy(), ob, rtol=1e-4, atol=1e-4)


def test_strided_and_inverse_conv(hip):
    rng = np.random.default_rng(3)
    dims, B, cin, cout = [11, 24, 26], 2, 32, 64
    ks, st, pd = [3, 3, 3], [2, 2, 2], [0, 1, 1]
    coords = rand_sites(rng, B, dims, 2000)
    x = rng.normal(size=(len(coords), cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) * 0.1).astype(np.float32)
    oc, od, nbr, nbrT = O.rulebook_conv(coords, B, dims, ks, st, pd)
    y = hip.spconv_fwd(dev(x), dev(W), None, dev(nbr)).cpu().numpy()
    yo = O.spconv_fwd(x, W, None, nbr)
    np.testing.assert_allclose(y, yo, **CONV_TOL)
    gy = rng.normal(size=yo.shape).astype(np.float32)
    gx = hip.spconv_dgrad(dev(gy), dev(W), dev(nbrT)).cpu().numpy()
    np.testing.assert_allclose(gx, O.spconv_dgrad(gy, W, nbrT), **CONV_TOL)
    gw, _ = hip.spconv_wgrad(dev(x), dev(gy), dev(nbr), cin, cout, with_bias=False)
    np.testing.assert_allclose(gw.cpu().numpy(), O.spconv_wgrad(x, gy, nbr, cin, cout)[0], rtol=1e-4, atol=1e-4)
    # inverse conv: forward over the transposed table with its own weight [K, cout, c2]
    Wi = (rng.normal(size=(27, cout, 32)) * 0.1).astype(np.float32)
    z = hip.spconv_fwd(dev(yo), dev(Wi), None, dev(nbrT)).cpu().numpy()
    np.testing.assert_allclose(z, O.spconv_fwd(yo, Wi, None, nbrT), **CONV_TOL)


def test_conv_ragged_sizes_and_empty(hip):
    rng = np.random.default_rng(4)
    for n in (1, 15, 63, 64, 65, 129):
        dims = [4, 8, 8]
        coords = rand_sites(rng, 1, dims, n)
        x = rng.normal(size=(len(coords), 16)).astype(np.float32)
        W = (rng.normal(size=(27, 16, 16)) * 0.1).astype(np.float32)
        nbr = O.rulebook_subm(coords, 1, dims)
        y = hip.spconv_fwd(dev(x), dev(W), None, dev(nbr)).cpu().numpy()
        np.testing.assert_allclose(y, O.spconv_fwd(x, W, None, nbr), **CONV_TOL)
    e = hip.spconv_fwd(torch.zeros((0, 16), device="cuda"), dev(W), None,
                       torch.zeros((0, 27), dtype=torch.int32, device="cuda"))
    assert e.shape == (0, 16)


def test_conv_linearity_at_full_size(hip):
    """Size-independent property at the BASELINE scan size: conv(a*x1 + x2) == a*conv(x1) + conv(x2)."""
    _, c, _ = O.voxelize(S.scan(), S.PC_RANGE, S.VOXEL_SIZE, 10, 40000)
    coords = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    idx = hip.SiteIndex(dev(coords), 1, [41, 768, 1408])
    nbr = hip.rulebook_subm(idx, [3, 3, 3])
    g = torch.Generator(device="cuda").manual_seed(0)
    x1 = torch.randn((len(c), 16), device="cuda", generator=g)
    x2 = torch.randn((len(c), 16), device="cuda", generator=g)
    W = torch.randn((27, 16, 16), device="cuda", generator=g) * 0.1
    lhs = hip.spconv_fwd((2.5 * x1 + x2).contiguous(), W, None, nbr)
    rhs = 2.5 * hip.spconv_fwd(x1, W, None, nbr) + hip.spconv_fwd(x2, W, None, nbr)
    assert float((lhs - rhs).abs().max()) < 1e-4
    # and against the oracle on the full-size table
    yo = O.spconv_fwd(x1.cpu().numpy(), W.cpu().numpy(), None, nbr.cpu().numpy())
    np.testing.assert_allclose(hip.spconv_fwd(x1, W, None, nbr).cpu().numpy(), yo, **CONV_TOL)


@pytest.mark.parametrize("tiled", [1, 0])
@pytest.mark.parametrize("Cc,n", [(64, 200), (64, 1), (16, 67), (48, 129), (8, 200)])
def test_dense_scatter_gather(hip, tiled, Cc, n):
    """dense() and its backward gather against the oracle: the 64-row x 64-channel LDS-tiled kernel (dense_tiled = 1, C >= 16)
    and the thread-per-element one, ragged last tiles, and the pair layout (frames = 2) the two must agree on bit for bit."""
    rng = np.random.default_rng(5)
    dims, B = [2, 12, 10], 4
    coords = rand_sites(rng, B, dims, n)
    f = rng.normal(size=(len(coords), Cc)).astype(np.float32)
    with hip.tuning(dense_tiled=tiled):
        d = hip.dense_scatter(dev(f), dev(coords), B, dims)
        assert (d.cpu().numpy() == O.dense(f, coords, B, dims)).all()
        back = hip.dense_gather(d, dev(coords), Cc, B, dims)
        assert (back.cpu().numpy() == f).all()
        d2 = hip.dense_scatter(dev(f), dev(coords), B, dims, frames=2)
        back2 = hip.dense_gather(d2, dev(coords), Cc, B, dims, frames=2)
    assert (back2.cpu().numpy() == f).all()
    with hip.tuning(dense_tiled=1 - tiled):
        assert torch.equal(hip.dense_scatter(dev(f), dev(coords), B, dims, frames=2), d2)


@pytest.mark.parametrize("B,T,Cg,H,W", [(4, 2, 128, 96, 176), (1, 2, 48, 7, 5), (3, 1, 100, 33, 17)])
def test_bev_channel_sums_and_display_match_torch_bits(hip, B, T, Cg, H, W):
    """rslo_bev_channel_sums + rslo_bev_display = the reference's logged extras (voxel_odom_net.py:455-464: occupancy of the
    concatenated frames, per-frame channel mean normalised to [0, 1]) with torch's own arithmetic, bit for bit."""
    g = torch.Generator(device="cuda").manual_seed(3)
    bev = torch.randn((B, T * Cg, H, W), device="cuda", generator=g)
    bev = bev * (torch.rand((B, 1, H, W), device="cuda", generator=g) < 0.3)       # mostly empty cells, as a BEV map
    sums = hip.bev_channel_sums(bev, T)
    sums2, mf, mb, ob = hip.bev_channel_sums(bev, T, masks=True)       # the head's input mask from the same launch
    occ0 = bev[:, :Cg].sum(dim=1, keepdim=True) != 0
    assert torch.equal(sums2, sums) and mb.dtype == torch.bool and ob.dtype == torch.bool
    assert torch.equal(mb, sums[:, 0:1] != 0) and torch.equal(ob, ~mb) and torch.equal(mf, mb.float())
    assert int((mb != occ0).sum()) == 0
    mask, disp = hip.bev_display(sums, Cg)
    frames = list(bev.split(Cg, dim=1))
    ref_mask = (sums.sum(dim=1, keepdim=True) != 0).float()
    assert torch.equal(mask, ref_mask)
    assert torch.equal(mask, (torch.sum(torch.cat(frames, dim=1), dim=1, keepdim=True) != 0).float())
    for t in range(T):
        d = sums[:, t:t + 1] / float(Cg)
        ref = (d - d.min()) / (d.max() - d.min() + 1e-12)
        assert torch.equal(disp[t], ref)
        m = frames[t].mean(dim=1, keepdim=True)       # the reference's expression: same up to the summation order
        ref2 = (m - m.min()) / (m.max() - m.min() + 1e-12)
        assert float((disp[t] - ref2).abs().max()) < 1e-5


@pytest.mark.parametrize("B,Ca,Cb,H,W,s", [(2, 5, 3, 7, 9, 2), (1, 64, 128, 24, 44, 2), (2, 3, 4, 5, 6, 1), (1, 2, 2, 4, 5, 3)])
def test_cat_upsample_matches_torch_bits(hip, B, Ca, Cb, H, W, s):
    """rslo_cat_upsample_fwd / _bwd = nn.Upsample(scale_factor=s)(torch.cat([a, b], 1)) and its autograd backward
    (reference odom_pred.py:219-221), bit for bit, including one-sided gradients."""
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn((B, Ca, H, W), device="cuda", generator=g, requires_grad=True)
    b = torch.randn((B, Cb, H, W), device="cuda", generator=g, requires_grad=True)
    ref = torch.nn.Upsample(scale_factor=s)(torch.cat([a, b], dim=1))
    go = torch.randn(ref.shape, device="cuda", generator=g)
    ref.backward(go)
    out = hip.cat_upsample_fwd(a.detach(), b.detach(), s)
    assert torch.equal(out, ref.detach())
    da, db = hip.cat_upsample_bwd(go, Ca, Cb, s)
    assert torch.equal(da, a.grad) and torch.equal(db, b.grad)
    da, db = hip.cat_upsample_bwd(go, Ca, Cb, s, need_a=False)
    assert da is None and torch.equal(db, b.grad)


def test_pose_tail_matches_torch(hip):
    """rslo_pose_tail_fwd / _bwd = (o[:, :3], o[:, 3:] / (|o[:, 3:]| + 1e-12)) and its autograd gradient (reference
    odom_pred.py:279-288) to fp32 rounding; one-sided and zero-quaternion rows included."""
    g = torch.Generator(device="cuda").manual_seed(5)
    o = torch.randn((6, 7), device="cuda", generator=g)
    o[2, 3:] = 0                                     # |q| = 0: torch.norm's gradient is zero there
    o[3, 3:] *= 1e-3
    o = o.requires_grad_(True)
    t_ref, q = o[:, :3], o[:, 3:]
    r_ref = q / (torch.norm(q, dim=1, keepdim=True) + 1e-12)
    gt, gr = torch.randn((6, 3), device="cuda", generator=g), torch.randn((6, 4), device="cuda", generator=g)
    (t_ref * gt).sum().add((r_ref * gr).sum()).backward()
    t, r = hip.pose_tail_fwd(o.detach())
    assert torch.equal(t, t_ref.detach())
    assert torch.allclose(r, r_ref.detach(), rtol=2e-6, atol=1e-7)
    d = hip.pose_tail_bwd(o.detach(), gt, gr)
    scale = float(o.grad.abs().max())
    assert float((d - o.grad).abs().max()) <= 2e-6 * scale
    d_t_only = hip.pose_tail_bwd(o.detach(), gt, None)
    assert torch.equal(d_t_only[:, :3], gt) and float(d_t_only[:, 3:].abs().max()) == 0.0


def test_leaky_bwd(hip):
    y = torch.randn(1001, device="cuda")
    g = torch.randn(1001, device="cuda")
    out = hip.leaky_bwd(y, g, 0.01)
    assert torch.equal(out, torch.where(y > 0, g, g * 0.01))


# ------------------------------------------------------------------------------- chamfer
def test_chamfer_golden_and_oracle_bitexact(hip):
    g = np.load(os.path.join(GOLD, "chamfer_ref.npz"))
    d, i = hip.chamfer_nn(dev(g["xyz1"]), dev(g["xyz2"]))
    assert (i.cpu().numpy() == g["idx1"]).all() and (d.cpu().numpy() == g["dist1"]).all()
    g1, g2 = hip.chamfer_grad(dev(g["xyz1"]), dev(g["xyz2"]), dev(g["graddist1"]), i)
    np.testing.assert_allclose(g1.cpu().numpy(), g["gradxyz1"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(g2.cpu().numpy(), g["gradxyz2"], rtol=1e-5, atol=1e-5)
    rng = np.random.default_rng(7)
    for (b, n, m) in [(1, 1, 1), (1, 5, 3000), (3, 1025, 777), (1, 4097, 4096)]:
        a = (rng.normal(size=(b, n, 3)) * 10).astype(np.float32)
        c = (rng.normal(size=(b, m, 3)) * 10).astype(np.float32)
        d, i = hip.chamfer_nn(dev(a), dev(c))
        od, oi = O.chamfer_nn(a, c)
        assert (i.cpu().numpy() == oi).all() and (d.cpu().numpy() == od).all()


def test_chamfer_two_direction_entry_points_match_reference(hip):
    """cd.forward_cuda / cd.backward_cuda and the ChamferDistance / ChamferDistanceWithIdx autograd modules
    (thirdparty/chamfer_distance/chamfer_distance.py:47-130,234-246) against vectors of the reference's own CPU build:
    distances and indices bit-exact in both directions, gradients with both upstream gradients non-zero."""
    from thirdparty.chamfer_distance.chamfer_distance import ChamferDistance, ChamferDistanceWithIdx, cd
    g = np.load(os.path.join(GOLD, "chamfer_ref.npz"))
    a, c = dev(g["xyz1"]), dev(g["xyz2"])
    B, N, M = a.shape[0], a.shape[1], c.shape[1]
    d1, d2 = torch.zeros(B, N, device="cuda"), torch.zeros(B, M, device="cuda")
    i1 = torch.zeros(B, N, dtype=torch.int32, device="cuda")
    i2 = torch.zeros(B, M, dtype=torch.int32, device="cuda")
    cd.forward_cuda(a, c, d1, d2, i1, i2)
    for got, key in ((d1, "dist1"), (d2, "dist2"), (i1, "idx1"), (i2, "idx2")):
        assert (got.cpu().numpy() == g[key]).all(), key
    g1, g2 = torch.zeros_like(a), torch.zeros_like(c)
    cd.backward_cuda(a, c, g1, g2, dev(g["graddist1"]), dev(g["graddist2"]), i1, i2)
    np.testing.assert_allclose(g1.cpu().numpy(), g["gradxyz1_both"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(g2.cpu().numpy(), g["gradxyz2_both"], rtol=1e-6, atol=1e-6)
    # autograd modules
    ar, cr = a.clone().requires_grad_(True), c.clone().requires_grad_(True)
    e1, e2, j1, j2 = ChamferDistanceWithIdx()(ar, cr)
    assert torch.equal(e1, d1) and torch.equal(e2, d2) and torch.equal(j1, i1) and torch.equal(j2, i2)
    ((e1 * dev(g["graddist1"])).sum() + (e2 * dev(g["graddist2"])).sum()).backward()
    np.testing.assert_allclose(ar.grad.cpu().numpy(), g["gradxyz1_both"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cr.grad.cpu().numpy(), g["gradxyz2_both"], rtol=1e-6, atol=1e-6)
    ar2, cr2 = a.clone().requires_grad_(True), c.clone().requires_grad_(True)
    f1, f2 = ChamferDistance()(ar2, cr2)
    (f1 * dev(g["graddist1"])).sum().backward()           # one upstream gradient only: the one-direction vectors
    np.testing.assert_allclose(ar2.grad.cpu().numpy(), g["gradxyz1"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cr2.grad.cpu().numpy(), g["gradxyz2"], rtol=1e-6, atol=1e-6)


def test_chamfer_full_size_properties(hip):
    """At the BASELINE size (~31k x 31k): self-query gives idx == arange and dist == 0; a sampled
    subset agrees bit-exactly with the oracle."""
    v, c, n = O.voxelize(S.scan(), S.PC_RANGE, S.VOXEL_SIZE, 10, 40000)
    p = O.vfe_mean(v, n)[:, :3][None].copy()
    t = dev(p)
    d, i = hip.chamfer_nn(t, t)
    assert torch.equal(i[0].long(), torch.arange(p.shape[1], device="cuda")) and float(d.abs().max()) == 0.0
    q = (p[:, ::37] + np.float32(0.05)).copy()
    d, i = hip.chamfer_nn(dev(q), t)
    od, oi = O.chamfer_nn(q, p)
    assert (i.cpu().numpy() == oi).all() and (d.cpu().numpy() == od).all()


@pytest.mark.parametrize("method", ["brute", "grid"])
def test_chamfer_methods_bitexact_vs_oracle(hip, method):
    """Both searches behind rslo_chamfer_nn (exhaustive scan / spatially pruned) give the oracle's bits: random clouds,
    clustered clouds with exact duplicates (ties -> lowest index), far-apart clouds, points outside the key range,
    ragged batches with +inf padding rows."""
    rng = np.random.default_rng(11)
    cases = []
    cases.append(((rng.normal(size=(2, 3000, 3)) * 10).astype(np.float32), (rng.normal(size=(2, 2500, 3)) * 10).astype(np.float32)))
    base = (rng.normal(size=(1, 400, 3)) * 20).astype(np.float32)
    dup = np.concatenate([base, base[:, ::-1], base + np.float32(1e-3)], 1)               # duplicates: ties
    cases.append((np.concatenate([base, base * np.float32(1.0001)], 1), dup))
    cases.append((rng.uniform(-70, 70, size=(1, 2000, 3)).astype(np.float32) + np.float32(300.0),
                  rng.uniform(-70, 70, size=(1, 2100, 3)).astype(np.float32)))             # no overlap, outside the grid
    cases.append((rng.uniform(-2000, 2000, size=(1, 1500, 3)).astype(np.float32),
                  rng.uniform(-2000, 2000, size=(1, 1700, 3)).astype(np.float32)))         # clamped keys everywhere
    for a, c in cases:
        d, i = hip.chamfer_nn(dev(a), dev(c), method=method)
        od, oi = O.chamfer_nn(a, c)
        assert (i.cpu().numpy() == oi).all() and (d.cpu().numpy() == od).all()
    # ragged: pair b uses its first cnt[b] points on both sides; padding rows answer (+inf, 0)
    a = (rng.normal(size=(3, 2048, 3)) * 15).astype(np.float32)
    c = (rng.normal(size=(3, 2304, 3)) * 15).astype(np.float32)
    ncnt, mcnt = np.array([2048, 1000, 1], np.int32), np.array([2304, 700, 3], np.int32)
    d, i = hip.chamfer_nn(dev(a), dev(c), ncnt=dev(ncnt), mcnt=dev(mcnt), method=method)
    d, i = d.cpu().numpy(), i.cpu().numpy()
    for b in range(3):
        od, oi = O.chamfer_nn(a[b:b + 1, :ncnt[b]], c[b:b + 1, :mcnt[b]])
        assert (i[b, :ncnt[b]] == oi[0]).all() and (d[b, :ncnt[b]] == od[0]).all()
        assert np.isinf(d[b, ncnt[b]:]).all() and (i[b, ncnt[b]:] == 0).all()


def test_chamfer_grid_equals_brute_at_full_size(hip):
    """BASELINE size, ragged batch of 4 KITTI-shaped pairs: the pruned search returns the exhaustive scan's bits, for
    well-aligned clouds and for a grossly wrong pose (early-training case: most queries far from every target)."""
    v, c, n = O.voxelize(S.scan(), S.PC_RANGE, S.VOXEL_SIZE, 10, 40000)
    p = O.vfe_mean(v, n)[:, :3].copy()
    rng = np.random.default_rng(3)
    N = len(p)
    a = np.stack([p, p[rng.permutation(N)], p, p]).astype(np.float32)
    th = np.float32(2.2)
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
    Rx = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]], np.float32)
    c = np.stack([p + np.float32([0.8, 0.05, 0.0]), p + np.float32([0.3, -0.2, 0.02]), p @ Rz.T, p @ (Rx @ Rz).T + np.float32(5.0)])
    cnt = np.array([N, N - 777, N, N - 5000], np.int32)
    db, ib = hip.chamfer_nn(dev(a), dev(c.astype(np.float32)), ncnt=dev(cnt), mcnt=dev(cnt), method="brute")
    dg, ig = hip.chamfer_nn(dev(a), dev(c.astype(np.float32)), ncnt=dev(cnt), mcnt=dev(cnt), method="grid")
    assert torch.equal(ib, ig) and torch.equal(db, dg)


def test_segmented_batchnorm_matches_torch_per_frame(hip):
    """rslo_segbn_fwd/bwd == nn.BatchNorm1d applied frame by frame (+ LeakyReLU), incl. running statistics."""
    torch.manual_seed(0)
    offs = [0, 700, 700, 1900, 2500]          # one empty frame
    S, Cc = 4, 32
    x = torch.randn(offs[-1], Cc, device="cuda") * 2 + 0.5
    gy = torch.randn(offs[-1], Cc, device="cuda")
    bn = torch.nn.BatchNorm1d(Cc).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref_bn = torch.nn.BatchNorm1d(Cc).cuda().train()
    ref_bn.load_state_dict(bn.state_dict())
    xr = x.clone().requires_grad_(True)
    parts = [torch.nn.functional.leaky_relu(ref_bn(xr[offs[b]:offs[b + 1]]), 0.01) for b in range(S) if offs[b + 1] > offs[b]]
    yr = torch.cat(parts, 0)
    (yr * gy).sum().backward()
    seg = torch.tensor(offs, dtype=torch.int32, device="cuda")
    y, mean, invstd = hip.segbn_fwd(x, seg, S, 1200, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                    bn.running_var, bn.momentum, bn.eps, 0.01)
    assert float((y - yr).abs().max()) < 2e-5
    assert float((bn.running_mean - ref_bn.running_mean).abs().max()) < 1e-6
    assert float((bn.running_var - ref_bn.running_var).abs().max()) < 1e-5
    gx, dg, db = hip.segbn_bwd(x, y, gy, seg, S, 1200, bn.weight.detach(), mean, invstd, 0.01)
    assert float((gx - xr.grad).abs().max()) < 2e-5
    assert float((dg - ref_bn.weight.grad).abs().max()) < 2e-4 and float((db - ref_bn.bias.grad).abs().max()) < 2e-4


def test_c5_dense_scan_geometry(hip):
    """BASELINE config C5 geometry: 128-ring scan (~263k points), 0.1 m cubic voxels, sparse shape [81,768,1408],
    max_voxels 2^18.  Voxel ids and the level-0/1 rulebooks are bit-exact vs the oracle; one conv agrees."""
    pts = S.scan(n_el=128)
    assert len(pts) > 260000
    v, c, n = _vox_both(hip, pts, S.PC_RANGE, S.VOXEL_SIZE_DENSE, S.MAX_POINTS_PER_VOXEL, 1 << 18)
    assert len(c) > 50000
    coords = np.concatenate([np.zeros((len(c), 1), np.int32), c.cpu().numpy()], 1)
    dims = [81, 768, 1408]
    idx = hip.SiteIndex(dev(coords), 1, dims)
    nbr = hip.rulebook_subm(idx, [3, 3, 3])
    onbr = O.rulebook_subm(coords, 1, dims)
    assert (nbr.cpu().numpy() == onbr).all()
    oidx, nb, nbT = hip.rulebook_conv(idx, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    oc, od, onb, onbT = O.rulebook_conv(coords, 1, dims, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert od == [41, 384, 704] and (oidx.coords.cpu().numpy() == oc).all()
    assert (nb.cpu().numpy() == onb).all() and (nbT.cpu().numpy() == onbT).all()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((len(c), 16), device="cuda", generator=g)
    W = torch.randn((27, 16, 32), device="cuda", generator=g) * 0.1
    y = hip.spconv_fwd(x, W, None, nb)
    np.testing.assert_allclose(y.cpu().numpy(), O.spconv_fwd(x.cpu().numpy(), W.cpu().numpy(), None, onb), **CONV_TOL)


def test_pad_rows_fwd_bwd(hip):
    """rslo_pad_rows_fwd/_bwd: ragged rows -> zero-padded batch and its inverse copy (exact)."""
    import torch
    from rslo.core import losses
    g = torch.Generator().manual_seed(0)
    src = torch.randn(1000, 13, generator=g)
    off = torch.tensor([0, 300, 450, 900], dtype=torch.int32)
    length = torch.tensor([250, 150, 400, 100], dtype=torch.int32)
    ref_in = src.clone().requires_grad_(True)
    ref = losses.pad_rows(ref_in, off, length, 400)            # CPU formulation (slicing)
    w = torch.randn(4, 400, 13, generator=g)
    (ref * w).sum().backward()
    dev_in = src.cuda().requires_grad_(True)
    out = losses.pad_rows(dev_in, off.cuda(), length.cuda(), 400)
    (out * w.cuda()).sum().backward()
    assert torch.equal(out.cpu(), ref) and torch.equal(dev_in.grad.cpu(), ref_in.grad)


@pytest.mark.parametrize("F", [7, 6])
def test_pair_rows_equals_the_slice_cat_pad_formulation(hip, F):
    """rslo_pair_rows_fwd (one launch per frame) == column selection + concatenation with the covariance rows + pad_rows +
    the column slices of the result (voxel_odom_net.py:630-660), values and the covariance gradient, exactly."""
    import torch
    from rslo.core import losses
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(1000, F, generator=g)
    conf = torch.randn(1000, 7, generator=g)
    off = torch.tensor([0, 300, 450, 900], dtype=torch.int32)
    length = torch.tensor([250, 150, 400, 100], dtype=torch.int32)
    cols = torch.cat([feats[:, 0:3], feats[:, 4:7]], 1) if F > 6 else feats[:, 0:6]
    cref = conf.clone().requires_grad_(True)
    both = losses.pad_rows(torch.cat([cols, cref], 1), off, length, 401)      # CPU formulation
    w = torch.randn(4, 401, 7, generator=g)
    (both[:, :, 6:] * w).sum().backward()
    cdev = conf.cuda().requires_grad_(True)
    xyz, nrm, cov = losses.pair_rows(feats.cuda(), cdev, off.cuda(), length.cuda(), 401)
    (cov * w.cuda()).sum().backward()
    assert xyz.is_contiguous() and nrm.is_contiguous() and cov.is_contiguous()
    assert torch.equal(xyz.cpu(), both[:, :, 0:3]) and torch.equal(nrm.cpu(), both[:, :, 3:6])
    assert torch.equal(cov.cpu(), both[:, :, 6:].detach()) and torch.equal(cdev.grad.cpu(), cref.grad)
    assert not xyz.requires_grad and not nrm.requires_grad


def test_rigid_move_fwd_bwd(hip):
    """rslo_transform_rows / _bwd on a column slice of a wider tensor == x @ R^T + t and its pose gradients."""
    from rslo.core import losses
    g = torch.Generator().manual_seed(4)
    wide = torch.randn(3, 1500, 6, generator=g)
    R = torch.randn(3, 3, 3, generator=g, requires_grad=True)
    t = torch.randn(3, 3, generator=g, requires_grad=True)
    w = torch.randn(3, 1500, 3, generator=g)
    ref = losses.rigid_move(wide[:, :, 3:], R, t)               # CPU formulation
    (ref * w).sum().backward()
    Rd, td = R.detach().cuda().requires_grad_(True), t.detach().cuda().requires_grad_(True)
    out = losses.rigid_move(wide.cuda()[:, :, 3:], Rd, td)
    (out * w.cuda()).sum().backward()
    assert float((out.cpu() - ref).abs().max()) < 1e-5
    assert float((Rd.grad.cpu() - R.grad).abs().max() / R.grad.abs().max()) < 1e-5
    assert float((td.grad.cpu() - t.grad).abs().max() / t.grad.abs().max()) < 1e-5
    out2 = losses.rigid_move(wide.cuda()[:, :, :3], Rd.detach())
    assert float((out2.cpu() - wide[:, :, :3] @ R.detach().transpose(-1, -2)).abs().max()) < 1e-5


@pytest.mark.parametrize("slope,with_res", [(None, False), (0.0, False), (0.0, True), (0.1, True)])
def test_fused_bn2d_act_matches_torch(hip, slope, with_res):
    """rslo_bn2d_* (stats -> apply, reduce -> apply) == nn.BatchNorm2d(train) [+ residual] [+ (Leaky)ReLU]: outputs,
    running statistics, input / residual / affine gradients; odd spatial size exercises the scalar tails."""
    import apex.parallel as AP
    torch.manual_seed(0)
    for shape in [(4, 64, 24, 44), (2, 32, 7, 9), (2, 16, 96, 176)]:      # one-launch forms (two block sizes), slice form
        x = (torch.randn(*shape, device="cuda") * 2 + 0.3).requires_grad_(True)
        res = torch.randn(*shape, device="cuda", requires_grad=True) if with_res else None
        gy = torch.randn(*shape, device="cuda")
        ref = torch.nn.BatchNorm2d(shape[1], eps=1e-3, momentum=0.01).cuda().train()
        mine = AP.SyncBatchNorm(shape[1], eps=1e-3, momentum=0.01).cuda().train()
        with torch.no_grad():
            ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-0.5, 0.5)
        mine.load_state_dict(ref.state_dict())
        y = ref(x)
        if with_res:
            y = y + res
        if slope is not None:
            y = torch.nn.functional.leaky_relu(y, slope)
        (y * gy).sum().backward()
        want = [x.grad.clone(), res.grad.clone() if with_res else None, ref.weight.grad.clone(), ref.bias.grad.clone()]
        x.grad = None
        if with_res:
            res.grad = None
        AP.FUSED_BN = "1"          # force the fused kernels on this single rank
        try:
            assert mine.fusable(x)
            y2 = mine(x, act_slope=slope, residual=res)
        finally:
            AP.FUSED_BN = os.environ.get("RSLO_FUSED_BN", "1")
        (y2 * gy).sum().backward()
        assert float((y2 - y).abs().max()) < 2e-5
        assert float((mine.running_mean - ref.running_mean).abs().max()) < 1e-6
        assert float((mine.running_var - ref.running_var).abs().max()) < 1e-5
        assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1
        got = [x.grad, res.grad if with_res else None, mine.weight.grad, mine.bias.grad]
        for a, b in zip(got, want):
            if b is not None:
                assert float((a - b).abs().max() / (b.abs().max() + 1e-12)) < 2e-4


def test_fused_bn2d_cross_rank_statistics_by_hand(hip):
    """The multi-rank path all-reduces the [2C+1] statistics between the two kernels: summing the statistics of two
    half batches by hand and applying them to each half must equal BatchNorm over the whole batch."""
    torch.manual_seed(1)
    for shape in [(4, 16, 12, 22), (4, 8, 96, 176)]:          # direct per-channel sums / slice partials + finish
        C_ = shape[1]
        x = torch.randn(*shape, device="cuda") * 1.5 + 0.2
        g = torch.ones(C_, device="cuda"); b = torch.zeros(C_, device="cuda")
        sa, sb = hip.bn2d_stats(x[:2].contiguous()), hip.bn2d_stats(x[2:].contiguous())
        tot = sa + sb                                     # what dist.all_reduce(SUM) produces on every rank
        rm, rv = torch.zeros(C_, device="cuda"), torch.ones(C_, device="cuda")
        ya, mean, invstd = hip.bn2d_apply(x[:2].contiguous(), None, tot, g, b, rm, rv, 0.1, 1e-5, 1.0)
        yb, _, _ = hip.bn2d_apply(x[2:].contiguous(), None, tot, g, b, None, None, 0.1, 1e-5, 1.0)
        ref = torch.nn.functional.batch_norm(x, None, None, g, b, True, 0.1, 1e-5)
        assert float((torch.cat([ya, yb]) - ref).abs().max()) < 2e-5
        assert float((mean - x.mean((0, 2, 3))).abs().max()) < 1e-5
        # backward sums: both halves' reductions added by hand == the full-batch gradient
        gy = torch.randn_like(x)
        reds = [hip.bn2d_bwd_reduce(gy[h].contiguous(), None, x[h].contiguous(), mean, invstd, 1.0, False)[0]
                for h in (slice(0, 2), slice(2, 4))]
        red = reds[0] + reds[1]
        cnt = float(x.shape[0] * x.shape[2] * x.shape[3])
        dx = torch.cat([hip.bn2d_bwd_apply(gy[h].contiguous(), None, x[h].contiguous(), g, mean, invstd, red, cnt, 1.0,
                                           False, False)[0] for h in (slice(0, 2), slice(2, 4))])
        xr = x.clone().requires_grad_(True)
        torch.nn.functional.batch_norm(xr, None, None, g, b, True, 0.1, 1e-5).backward(gy)
        assert float((dx - xr.grad).abs().max()) < 2e-5 * max(1.0, float(xr.grad.abs().max()))


def _syncbn_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # two ranks on ONE GPU: gloo moves the CUDA tensors
    import rslo_amd  # noqa: F401
    import apex.parallel as AP
    torch.manual_seed(0)
    full = torch.randn(4, 16, 12, 22) * 1.5 + 0.2
    res_full = torch.randn(4, 16, 12, 22)
    gy_full = torch.randn(4, 16, 12, 22)
    sl = slice(2 * rank, 2 * rank + 2)
    x = full[sl].cuda().requires_grad_(True)
    res = res_full[sl].cuda().requires_grad_(True)
    bn = AP.SyncBatchNorm(16, eps=1e-3, momentum=0.01).cuda().train()
    assert bn.fusable(x)                                             # world size 2 -> fused path by default
    y = bn(x, act_slope=0.0, residual=res)
    (y * gy_full[sl].cuda()).sum().backward()
    q.put((rank, y.detach().cpu().numpy(), x.grad.cpu().numpy(), res.grad.cpu().numpy(), bn.weight.grad.cpu().numpy(),
           bn.bias.grad.cpu().numpy(), bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _block_worker(rank, world, port, q):
    import copy
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # two ranks on ONE GPU
    import rslo_amd  # noqa: F401
    from rslo.layers import hip_conv2d
    from rslo.layers.MaskConv import MaskConv
    from rslo.layers.SparseConv import FusedSequential, SPC_SyncBN2d
    from rslo.models import custom_resnet_spc as R
    torch.manual_seed(21)
    down = FusedSequential(R.conv1x1(64, 128, 2, Conv2d=MaskConv), SPC_SyncBN2d(128))
    blk = R.BasicBlock(64, 128, 2, down, BN=SPC_SyncBN2d, Conv2d=MaskConv).cuda().train()
    for p in blk.parameters():
        torch.nn.init.normal_(p, 0.0, 0.2)
    ref = copy.deepcopy(blk)
    full = torch.randn(4, 64, 24, 44)
    g_full = torch.randn(4, 128, 12, 22)
    sl = slice(2 * rank, 2 * rank + 2)
    outs = []
    for b, fused in ((blk, True), (ref, False)):
        os.environ["RSLO_FUSED_BLOCK"] = "1" if fused else "0"
        b.__dict__.pop("_fused_ok", None)
        hip_conv2d.presplit(b)
        x = full[sl].cuda().requires_grad_(True)
        y = b([x, None])[0]
        assert (type(y.grad_fn).__name__ == "_BasicBlockFnBackward") == fused
        y.backward(g_full[sl].cuda())
        outs.append([y.detach().cpu(), x.grad.cpu()] + [p.grad.cpu() for p in b.parameters()] + [t.cpu() for t in b.buffers()])
    same = all(torch.equal(a, c) for a, c in zip(*outs))
    q.put((rank, same, float(outs[0][0].abs().mean())))
    dist.barrier()
    dist.destroy_process_group()


def test_fused_block_node_with_two_ranks_equals_layer_nodes(hip):
    """_BasicBlockFn under SyncBN with two ranks (gloo, one GPU): the statistics exchanges happen inside the node in the
    same order as the layer-by-layer nodes issue them -> identical bits on every rank."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_block_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1] for o in out) and all(o[2] > 0 for o in out), out


def test_fused_syncbn_two_ranks_equals_full_batch(hip):
    """Two processes (gloo, same GPU) each hold half of a batch: the fused SyncBatchNorm path must reproduce
    BatchNorm over the whole batch -- outputs, input / residual gradients, running statistics; the affine gradients are
    the LOCAL sums (data parallel averages parameter gradients afterwards)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    full = (torch.randn(4, 16, 12, 22) * 1.5 + 0.2).requires_grad_(True)
    res_full = torch.randn(4, 16, 12, 22).requires_grad_(True)
    gy_full = torch.randn(4, 16, 12, 22)
    ref = torch.nn.BatchNorm2d(16, eps=1e-3, momentum=0.01).train()
    y = torch.relu(ref(full) + res_full)
    (y * gy_full).sum().backward()
    y_got = np.concatenate([out[0][1], out[1][1]])
    gx_got = np.concatenate([out[0][2], out[1][2]])
    gr_got = np.concatenate([out[0][3], out[1][3]])
    np.testing.assert_allclose(y_got, y.detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(gx_got, full.grad.numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(gr_got, res_full.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out[0][4] + out[1][4], ref.weight.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(out[0][5] + out[1][5], ref.bias.grad.numpy(), rtol=1e-3, atol=1e-4)
    for r in range(2):
        np.testing.assert_allclose(out[r][6], ref.running_mean.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out[r][7], ref.running_var.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 64), (64, 32), (32, 32)])
def test_bf16_feature_conv_matches_oracle_on_rounded_operands(hip, cin, cout):
    """BASELINE config C4 (bf16 features, int32 rulebook, fp32 accumulate): rslo_spconv_fwd_bf16 on bf16-rounded
    inputs / weights equals the double-precision oracle on the SAME rounded operands up to the final bf16 rounding of
    the output (half an ulp = 2^-9 relative) -- forward with bias + LeakyReLU, and the data gradient (transposed)."""
    rng = np.random.default_rng(cin * 3 + cout)
    dims, B = [9, 30, 28], 2
    coords = rand_sites(rng, B, dims, 2500)
    nbr = O.rulebook_subm(coords, B, dims)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16)             # RNE
    x = rng.normal(size=(len(coords), cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    xr, Wr = bf(x).float().numpy(), bf(W).float().numpy()
    y = hip.spconv_fwd_bf16(bf(x).cuda(), dev(W), dev(b), dev(nbr), act_slope=0.01)
    assert y.dtype == torch.bfloat16
    yo = O.spconv_fwd(xr, Wr, b, nbr)
    yo = np.where(yo > 0, yo, yo * np.float32(0.01))
    err = np.abs(y.float().cpu().numpy() - yo)
    assert float((err / (np.abs(yo) + 1e-2)).max()) < 2.0 ** -8 + 1e-3
    gy = rng.normal(size=(len(coords), cout)).astype(np.float32)
    gx = hip.spconv_fwd_bf16(bf(gy).cuda(), dev(W), None, dev(nbr), flip_k=True, transpose=True)
    go = O.spconv_dgrad(bf(gy).float().numpy(), Wr, nbr[:, ::-1].copy())
    err = np.abs(gx.float().cpu().numpy() - go)
    assert gx.shape == (len(coords), cin) and float((err / (np.abs(go) + 1e-2)).max()) < 2.0 ** -8 + 1e-3


@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 64), (64, 32), (32, 32)])
def test_bf16_backward_kernels_match_oracle_on_rounded_operands(hip, cin, cout):
    """C4 backward of a sparse layer on bf16 rows: rslo_leaky_bwd_colsum_bf16 (g bit-exact vs the round-to-nearest
    bf16 of the fp32 formula, column partials = fp32 sums of the ROUNDED g) and rslo_spconv_wgrad_pairs_bf16 (products of
    bf16 values are exact in fp32, fp32 accumulation: 2e-5 of the largest entry vs the double-precision oracle on the
    same rounded operands; bias gradient from the partials)."""
    rng = np.random.default_rng(cin * 5 + cout)
    dims, B = [9, 30, 28], 2
    coords = rand_sites(rng, B, dims, 5000)
    nbr = O.rulebook_subm(coords, B, dims)
    n = len(coords)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16)             # RNE
    x = bf(rng.normal(size=(n, cin)).astype(np.float32))
    y = bf(rng.normal(size=(n, cout)).astype(np.float32))
    y[::7] = 0.0                                                       # exact zeros take the slope branch (y > 0 false)
    gy = bf(rng.normal(size=(n, cout)).astype(np.float32))
    g, part = hip.leaky_bwd_bf16(y.cuda(), gy.cuda(), 0.01, colsum=True)
    g_ref = torch.where(y.float() > 0, gy.float(), gy.float() * 0.01).to(torch.bfloat16)
    assert g.dtype == torch.bfloat16 and torch.equal(g.cpu(), g_ref)
    np.testing.assert_allclose(part.sum(0).cpu().numpy(), g_ref.float().double().sum(0).numpy(), rtol=1e-4, atol=1e-3)
    pairs = hip.rulebook_pairs(dev(nbr))
    dW, db = hip.spconv_wgrad_pairs_bf16(x.cuda(), g, pairs, n, 27, cin, cout, bias_partial=part)
    oW, ob = O.spconv_wgrad(x.float().numpy(), g_ref.float().numpy(), nbr, cin, cout)
    assert np.abs(dW.cpu().numpy() - oW).max() <= 2e-5 * np.abs(oW).max()
    np.testing.assert_allclose(db.cpu().numpy(), ob, rtol=1e-4, atol=1e-3)
    dW2, db2 = hip.spconv_wgrad_pairs_bf16(x.cuda(), g, pairs, n, 27, cin, cout)      # no bias requested
    assert db2 is None and torch.equal(dW2, dW)


@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 32, 64, 12, 22), (1, 64, 64, 96, 176), (2, 96, 64, 25, 23)])
def test_conv2d_bf16_operand_mode_matches_float64_on_rounded_operands(hip, B, cin, cout, H, W):
    """C4 dense head: rslo_conv2d_fwd_bf16 / rslo_conv2d_wgrad_bf16 = the product of the round-to-nearest bf16 values of
    both operands with fp32 accumulation -> 2e-5 of the largest entry vs float64 on the SAME rounded operands (forward
    with bias, data gradient, weight + bias gradient)."""
    rng = np.random.default_rng(B + cin + H)
    r16 = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    g = rng.standard_normal((B, cout, H, W)).astype(np.float32)
    y = hip.conv2d_fwd(dev(x), hip.conv2d_wsplit(dev(w), False), dev(bias), cout, lp=True).cpu().numpy()
    dx = hip.conv2d_fwd(dev(g), hip.conv2d_wsplit(dev(w), True), None, cin, lp=True).cpu().numpy()
    dw, db = hip.conv2d_wgrad(dev(x), dev(g), 1, want_bias=True, lp=True)
    ry, rdx = O.conv2d_fwd(r16(x), r16(w), bias), O.conv2d_dgrad(r16(g), r16(w))
    rdw = O.conv2d_wgrad(r16(x), r16(g))
    for got, ref in ((y, ry), (dx, rdx), (dw.cpu().numpy(), rdw)):
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    np.testing.assert_allclose(db.cpu().numpy(), g.astype(np.float64).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)   # fp32 sum of dout
    # and it really is the low-precision product: differs from the fp32-accurate path at the bf16 level
    y32 = hip.conv2d_fwd(dev(x), hip.conv2d_wsplit(dev(w), False), dev(bias), cout).cpu().numpy()
    assert 1e-4 < np.abs(y - y32).max() / np.abs(y32).max() < 3e-2


def test_roi_threshold_radix_select_is_exact(hip):
    """rslo_roi_threshold == sort-based k-th value (bit-exact), ragged counts, duplicates, +inf padding, tiny rows."""
    from rslo.core import losses
    g = torch.Generator().manual_seed(5)
    d = torch.rand(5, 4000, generator=g) ** 3 * 40
    d[1, :2000] = d[1, 2000:4000]                     # duplicates
    d[2] = 0.25                                        # all equal -> threshold clamps to 1
    cnt = torch.tensor([4000, 3999, 4000, 1, 37], dtype=torch.int32)
    for b in range(5):
        d[b, cnt[b]:] = float("inf")
    for ratio in (0.97, 0.5, 1.0):
        want = losses.roi_threshold_ragged(d, cnt, ratio).reshape(-1)
        got = hip.roi_threshold(d.cuda(), cnt.cuda(), ratio).cpu()
        assert torch.equal(got, want), (ratio, got, want)
    want = losses.roi_threshold(d[:1], 0.97).reshape(-1)
    assert torch.equal(hip.roi_threshold(d[:1].cuda().contiguous(), None, 0.97).cpu(), want)
    # the kernel keeps 40 values per thread (40960 per pair) in registers and reads any further ones from memory in every
    # pass: counts at the step's size, exactly at the register capacity, one past it and well past it
    d = torch.rand(4, 45000, generator=g) ** 2 * 9
    cnt = torch.tensor([31000, 40960, 40961, 45000], dtype=torch.int32)
    for b in range(4):
        d[b, cnt[b]:] = float("inf")
    for ratio in (0.97, 0.3):
        want = losses.roi_threshold_ragged(d, cnt, ratio).reshape(-1)
        assert torch.equal(hip.roi_threshold(d.cuda(), cnt.cuda(), ratio).cpu(), want), ratio


def test_pose_algebra_kernels_match_kornia_restatement(hip):
    """rslo_quat_to_rot (+ backward) and rslo_pose_targets vs the op-by-op formulation (kornia 0.4.0 restatement,
    torchplus.roll, sign fix), all four matrix->quaternion branches exercised."""
    from rslo.core import losses
    g = torch.Generator().manual_seed(8)
    q = torch.randn(64, 4, generator=g)
    q[:4] *= 3.0                                           # un-normalised inputs are normalised first
    w = torch.randn(64, 3, 3, generator=g)
    qc = q.clone().requires_grad_(True)
    Rc = losses.quat_wxyz_to_rot(qc)                       # CPU: kornia formulation
    (Rc * w).sum().backward()
    qd = q.cuda().requires_grad_(True)
    Rd = losses.quat_wxyz_to_rot(qd)
    (Rd * w.cuda()).sum().backward()
    assert float((Rd.cpu() - Rc).abs().max()) < 1e-6
    assert float((qd.grad.cpu() - qc.grad).abs().max() / qc.grad.abs().max()) < 1e-5
    # rotations around all axes by angles up to pi: every trace branch
    ang = torch.rand(64, generator=g) * 3.1
    axis = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    axis[:6] = torch.eye(3).repeat(2, 1); ang[:6] = 3.1
    qa = torch.cat([torch.cos(ang / 2)[:, None], axis * torch.sin(ang / 2)[:, None]], 1)
    res_r = losses.quat_wxyz_to_rot(qa).detach()
    res_t, Tp = torch.randn(64, 3, generator=g), torch.randn(64, 3, generator=g)
    Rp = losses.quat_wxyz_to_rot(torch.randn(64, 4, generator=g)).detach()
    rc, tc = losses.icp_pose_targets(res_r, res_t, Rp, Tp)
    rd, td = losses.icp_pose_targets(res_r.cuda(), res_t.cuda(), Rp.cuda(), Tp.cuda())
    assert float((td.cpu() - tc).abs().max()) < 1e-5
    assert float((rd.cpu() - rc).abs().max()) < 2e-5
    # the [B,7] (t*, q*) rows from the same launch hold the same bits
    rd2, td2, tq = losses.icp_pose_targets(res_r.cuda(), res_t.cuda(), Rp.cuda(), Tp.cuda(), with_tq=True)
    assert torch.equal(rd2, rd) and torch.equal(td2, td) and torch.equal(tq, torch.cat([td, rd], 1))


# ------------------------------------------------------------------------------- dense conv2d weight gradient (BEV head)
@pytest.mark.parametrize("B,cin,cout,H,W,stride", [
    (2, 32, 64, 12, 22, 1),      # plane of 264 pixels: ragged last 32-pixel chunk
    (1, 16, 32, 24, 44, 1),      # 32-wide cout block
    (3, 48, 64, 25, 23, 1),      # odd sizes, rows wrap inside a lane's 8 pixels
    (2, 48, 64, 25, 23, 2),      # stride 2, odd input
    (2, 32, 128, 24, 44, 2),     # stride 2, even input (last input column / row never read by kx = ky = 2 ... by the pad)
    (1, 64, 64, 96, 176, 1),     # full-resolution BEV map
    (1, 256, 128, 96, 176, 2),   # the full-resolution stride-2 layer (the library's until round 3)
])
def test_conv2d_wgrad_matches_float64_oracle(hip, B, cin, cout, H, W, stride):
    """fp32 tolerance: |err| <= 2e-5 * max|dW| against the float64 restatement (the split-bf16 products are exact to
    2^-24 relative per term; the rest is fp32 accumulation over B*H*W terms); torch's own fp32 result must be no
    closer than 4x."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    g = rng.standard_normal((B, cout, Ho, Wo)).astype(np.float32)
    assert hip.conv2d_wgrad_supported(cin, cout, H, W, stride)
    got = hip.conv2d_wgrad(dev(x), dev(g), stride).cpu().numpy().astype(np.float64)
    ref = O.conv2d_wgrad(x, g, stride)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= 2e-5 * scale, (err, scale)
    tw = torch.zeros((cout, cin, 3, 3), device="cuda", requires_grad=True)
    torch.nn.functional.conv2d(dev(x), tw, None, stride, 1).backward(dev(g))
    err_lib = np.abs(tw.grad.cpu().numpy().astype(np.float64) - ref).max()
    assert err <= 4 * err_lib + 1e-6 * scale, (err, err_lib)
    # bit-reproducible (fixed summation order)
    again = hip.conv2d_wgrad(dev(x), dev(g), stride).cpu().numpy().astype(np.float64)
    assert np.array_equal(got, again)
    if stride == 1:     # bias gradient from the same pass: per-channel sum of dout
        dw2, db = hip.conv2d_wgrad(dev(x), dev(g), 1, want_bias=True)
        assert np.array_equal(dw2.cpu().numpy().astype(np.float64), got)
        ref_b = g.astype(np.float64).sum((0, 2, 3))
        assert np.abs(db.cpu().numpy() - ref_b).max() <= 2e-5 * max(np.abs(ref_b).max(), np.sqrt(g[:, 0].size))


@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 64, 32, 12, 30), (1, 32, 64, 13, 21), (4, 256, 128, 96, 176), (2, 128, 128, 48, 88)])
def test_conv1x1s2_wgrad_matches_float64_oracle(hip, B, cin, cout, H, W):
    """rslo_conv1x1s2_wgrad (weight gradient of the 1x1 / stride-2 downsample layers, the centre tap of the stride-2 kernel):
    |err| <= 2e-5 * max|dW| against float64, no further than 4x the library's fp32 result, bit-reproducible."""
    rng = np.random.default_rng(17)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = rng.standard_normal((B, cout, Ho, Wo)).astype(np.float32)
    got_t = hip.conv1x1s2_wgrad(dev(x), dev(g))
    assert got_t is not None and tuple(got_t.shape) == (cout, cin, 1, 1)
    got = got_t.cpu().numpy().astype(np.float64).reshape(cout, cin)
    ref = np.einsum("bopq,bipq->oi", g.astype(np.float64), x[:, :, ::2, ::2].astype(np.float64))
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= 2e-5 * scale, (err, scale)
    tw = torch.zeros((cout, cin, 1, 1), device="cuda", requires_grad=True)
    torch.nn.functional.conv2d(dev(x), tw, None, 2, 0).backward(dev(g))
    err_lib = np.abs(tw.grad.cpu().numpy().astype(np.float64).reshape(cout, cin) - ref).max()
    assert err <= 4 * err_lib + 1e-6 * scale, (err, err_lib)
    assert torch.equal(hip.conv1x1s2_wgrad(dev(x), dev(g)), got_t)


def test_hip_conv2d_module_gradients_match_library(hip):
    from rslo.layers.hip_conv2d import Conv2d
    torch.manual_seed(5)
    for stride, bias in [(1, True), (2, False)]:
        m = Conv2d(32, 64, 3, stride=stride, padding=1, bias=bias).cuda()
        x = torch.randn(2, 32, 24, 44, device="cuda", requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        got = [x.grad.clone(), m.weight.grad.clone()] + ([m.bias.grad.clone()] if bias else [])
        x.grad = None; m.zero_grad()
        m.hip_wgrad = False
        y2 = m(x)
        y2.backward(gy)
        ref = [x.grad, m.weight.grad] + ([m.bias.grad] if bias else [])
        assert torch.allclose(y, y2, rtol=1e-4, atol=1e-5 * float(y2.abs().max()))
        for a, b in zip(got, ref):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))


def test_leaf_stream_weight_gradients_accumulate_and_share_safely(hip):
    """rslo_amd.streams.leaf: the dense weight gradients run on a side stream the autograd engine knows nothing about.
    (a) two backward() calls without zero_grad (AccumulateGrad ADDS on the training stream), (b) zero_grad(set_to_none=
    False), (c) one weight used by two layers in one pass: every case must give the bits of the run with the side stream
    switched off, on many repetitions with a long-running kernel in front of the leaf work (a race shows as a mismatch)."""
    from rslo.layers.hip_conv2d import Conv2d
    from rslo_amd import streams
    assert streams.ENABLED
    torch.manual_seed(11)
    m = Conv2d(64, 64, 3, stride=1, padding=1, bias=True).cuda()
    m2 = Conv2d(64, 64, 3, stride=1, padding=1, bias=False).cuda()
    m2.weight = m.weight                                   # (c) shared weight
    x = torch.randn(4, 64, 96, 176, device="cuda")

    def run(mode):
        m.zero_grad(set_to_none=True)
        if mode == "zero_fill":
            m(x).sum().backward()
            m.zero_grad(set_to_none=False)
        xin = x.clone().requires_grad_(True)
        if mode == "shared":
            (m2(torch.relu(m(xin))) * 0.5).sum().backward()
        else:
            (m(xin) * 0.25).sum().backward()
            if mode == "twice":
                (m(xin) * 0.75).sum().backward()
        torch.cuda.synchronize()
        return [m.weight.grad.clone(), m.bias.grad.clone(), xin.grad.clone()]

    for mode in ("twice", "zero_fill", "shared"):
        streams.ENABLED = False
        try:
            ref = run(mode)
        finally:
            streams.ENABLED = True
        for _ in range(5):
            got = run(mode)
            for a, b in zip(got, ref):
                assert torch.equal(a, b), mode
    assert not any(st["pending"] or st["keep"] or st["targets"] for st in streams._state.values())


@pytest.mark.parametrize("B,cin,cout,H,W,cfg", [
    (2, 32, 64, 12, 22, None),       # map smaller than one 16-wide tile pair, 4-row tiles
    (1, 64, 32, 24, 44, None),       # 32 output channels: one 16-block per wave
    (2, 96, 64, 25, 23, None),       # odd sizes: ragged tiles in both directions
    (1, 64, 64, 96, 176, None),      # full-resolution BEV map
    (2, 64, 128, 16, 40, "8,2"), (2, 64, 128, 16, 40, "8,1"), (2, 64, 128, 16, 40, "4,2"), (2, 64, 128, 16, 40, "4,1"),
    (2, 64, 128, 16, 40, "6,1"),     # 6-row tiles on a map they do not divide
    (1, 256, 256, 12, 22, None),     # the library's own choice here: two wave sets per workgroup (forward and gradient)
    (2, 96, 64, 25, 23, "kc=2"),     # two wave sets forced on an ODD number of channel chunks and ragged tiles
    (1, 64, 32, 9, 17, "kc=2"),      # ... and on one chunk per set
])
def test_conv2d_fwd_and_dgrad_match_float64_oracle(hip, B, cin, cout, H, W, cfg):
    """Forward (with bias) and data gradient through rslo_conv2d_wsplit + rslo_conv2d_fwd against the float64
    restatement: |err| <= 2e-5 * max|ref| (split-bf16 products exact to 2^-24 per term, fp32 accumulation over 9 Cin
    terms), and no further from it than 4x the library's own fp32 result.  cfg forces each tile configuration."""
    if cfg is not None:
        # a forced tile configuration: explicit switches of the launch code (rslo_tuning_set), restored afterwards
        kw = {"conv2d_fwd_kc": 2} if cfg == "kc=2" else dict(zip(("conv2d_fwd_tr", "conv2d_fwd_mtw"), map(int, cfg.split(","))))
        rng = np.random.default_rng(2)
        x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
        w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
        with hip.tuning(**kw):
            y = hip.conv2d_fwd(dev(x), hip.conv2d_wsplit(dev(w), False), None, cout).cpu().numpy()
        r = O.conv2d_fwd(x, w)
        assert np.abs(y - r).max() / np.abs(r).max() < 2e-5
        return
    rng = np.random.default_rng(12)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    g = rng.standard_normal((B, cout, H, W)).astype(np.float32)
    assert hip.conv2d_fwd_supported(cin, cout, H, W)
    y = hip.conv2d_fwd(dev(x), hip.conv2d_wsplit(dev(w), False), dev(bias), cout).cpu().numpy().astype(np.float64)
    dx = hip.conv2d_fwd(dev(g), hip.conv2d_wsplit(dev(w), True), None, cin).cpu().numpy().astype(np.float64)
    ry, rdx = O.conv2d_fwd(x, w, bias), O.conv2d_dgrad(g, w)
    tx = dev(x).requires_grad_(True)
    ty = torch.nn.functional.conv2d(tx, dev(w), dev(bias), 1, 1)
    ty.backward(dev(g))
    for got, ref, lib in [(y, ry, ty.detach().cpu().numpy()), (dx, rdx, tx.grad.cpu().numpy())]:
        scale = np.abs(ref).max()
        err = np.abs(got - ref).max()
        err_lib = np.abs(lib.astype(np.float64) - ref).max()
        assert err <= 2e-5 * scale, (err, scale)
        assert err <= 4 * err_lib + 1e-6 * scale, (err, err_lib)
    # rslo_conv2d_fwd_add: the residual read in the epilogue = the convolution followed by a separate add, bit for bit
    res = dev(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    wt = hip.conv2d_wsplit(dev(w), True)
    joined = hip.conv2d_fwd(dev(g), wt, None, cin, residual=res)
    apart = hip.conv2d_fwd(dev(g), wt, None, cin).add_(res)
    assert torch.equal(joined, apart)
    joined = hip.conv2d_fwd(dev(g), wt, None, cin, lp=True, residual=res)
    assert torch.equal(joined, hip.conv2d_fwd(dev(g), wt, None, cin, lp=True).add_(res))


@pytest.mark.parametrize("B,cin,cout,H,W,cfg", [
    (2, 32, 64, 12, 22, {}), (2, 96, 64, 25, 23, {}),                 # small / odd maps: ragged tiles, borders in every tile
    (4, 128, 128, 48, 88, {}), (1, 64, 64, 96, 176, {}),              # the one-tap-ahead variant (>= 512 workgroups)
    (2, 64, 128, 16, 40, {"conv2d_fwd_tr": 8, "conv2d_fwd_mtw": 2}), (2, 64, 128, 16, 40, {"conv2d_fwd_tr": 8}),
    (2, 64, 128, 16, 40, {"conv2d_fwd_mtw": 2}), (2, 64, 128, 48, 88, {"conv2d_fwd_lean": 0}),
])
def test_conv2d_planes_kernel_is_bit_identical_to_the_fp32_staged_kernel(hip, B, cin, cout, H, W, cfg):
    """rslo_opl_from_nchw + rslo_conv2d_fwd_p (pre-split operand planes, round 5) against rslo_conv2d_fwd on the same
    tensor: the producer's split is the split the staged kernel performs and the six products are accumulated in the same
    order, so forward (with bias), data gradient and the residual epilogue are equal BIT FOR BIT; and the planes
    themselves recompose to the fp32 values exactly (hi + mid + lo)."""
    rng = np.random.default_rng(21)
    x = dev(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    g = dev(rng.standard_normal((B, cout, H, W)).astype(np.float32))
    w = dev((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32))
    bias = dev(rng.standard_normal(cout).astype(np.float32))
    res = dev(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    wf, wt = hip.conv2d_wsplit(w, False), hip.conv2d_wsplit(w, True)
    assert hip.conv2d_fwd_p_supported(cin, cout, H, W)
    px, pg = hip.opl_from_nchw(x), hip.opl_from_nchw(g)
    # planes [B, C/8, 3, H, W, 8] bf16 bit patterns -> fp32: exact recomposition
    back = (px.view(torch.bfloat16).float().sum(dim=2)).permute(0, 1, 4, 2, 3).reshape(B, cin, H, W)
    assert torch.equal(back, x)
    ref_y = hip.conv2d_fwd(x, wf, bias, cout)
    ref_dx = hip.conv2d_fwd(g, wt, None, cin, residual=res)
    with hip.tuning(**cfg):
        y = hip.conv2d_fwd_p(px, wf, bias, cout)
        dx = hip.conv2d_fwd_p(pg, wt, None, cin, residual=res)
    assert torch.equal(y, ref_y) and torch.equal(dx, ref_dx)
    r = O.conv2d_fwd(x.cpu().numpy(), w.cpu().numpy(), bias.cpu().numpy())
    assert np.abs(y.cpu().numpy() - r).max() <= 2e-5 * np.abs(r).max()


@pytest.mark.parametrize("B,cin,cout,H,W,k", [
    (2, 64, 32, 12, 22, 3), (2, 64, 32, 12, 22, 1),          # small even map
    (1, 32, 64, 13, 21, 3), (1, 32, 64, 13, 21, 1),          # odd sizes: ragged parity classes
    (2, 128, 128, 48, 88, 3), (2, 128, 128, 48, 88, 1),      # stage 2 of the BEV encoder
    (1, 256, 128, 96, 176, 3),                               # stage 1: full-resolution map
])
def test_conv2d_stride2_fwd_and_dgrad_match_float64_oracle(hip, B, cin, cout, H, W, k):
    """The stride-2 layers (3x3 / padding 1 and the 1x1 downsample) through rslo_conv2d_wsplit_k + rslo_conv2d_fwd_s2 /
    rslo_conv2d_dgrad_s2 against the float64 restatement: |err| <= 2e-5 max|ref| and no further from it than 4x the
    library's fp32 result; the data gradient writes every element (the buffer is pre-filled with NaN)."""
    rng = np.random.default_rng(31 + k)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(k * k * cin)).astype(np.float32)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = rng.standard_normal((B, cout, Ho, Wo)).astype(np.float32)
    assert hip.conv2d_s2_supported(cin, cout, k)
    y = hip.conv2d_fwd_s2(dev(x), hip.conv2d_wsplit_k(dev(w), False), cout, k)
    assert y.shape == (B, cout, Ho, Wo)
    dx = hip.conv2d_dgrad_s2(dev(g), hip.conv2d_wsplit_k(dev(w), True), cin, H, W, k)
    assert dx.shape == (B, cin, H, W) and bool(torch.isfinite(dx).all())
    ry, rdx = O.conv2d_s2_fwd(x, w), O.conv2d_s2_dgrad(g, w, H, W)
    tx = dev(x).requires_grad_(True)
    ty = torch.nn.functional.conv2d(tx, dev(w), None, 2, (k - 1) // 2)
    ty.backward(dev(g))
    for got, ref, lib in [(y.cpu().numpy().astype(np.float64), ry, ty.detach().cpu().numpy()),
                          (dx.cpu().numpy().astype(np.float64), rdx, tx.grad.cpu().numpy())]:
        scale = np.abs(ref).max()
        err = np.abs(got - ref).max()
        err_lib = np.abs(lib.astype(np.float64) - ref).max()
        assert err <= 2e-5 * scale, (err, scale)
        assert err <= 4 * err_lib + 1e-6 * scale, (err, err_lib)
    # rslo_conv2d_dgrad_s2_add: the other branch's gradient read in the epilogue = a separate add, bit for bit
    res = dev(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    joined = hip.conv2d_dgrad_s2(dev(g), hip.conv2d_wsplit_k(dev(w), True), cin, H, W, k, residual=res)
    assert torch.equal(joined, res.clone().add_(dx))
    if k == 1:      # in place: only the pixels (2y, 2x) of the residual are read and written, same bits
        buf = res.clone()
        out = hip.conv2d_dgrad_s2(dev(g), hip.conv2d_wsplit_k(dev(w), True), cin, H, W, k, residual=buf, inplace=True)
        assert out.data_ptr() == buf.data_ptr() and torch.equal(out, joined)


def test_hip_conv2d_stride2_layers_match_library_through_autograd(hip):
    """rslo.layers.hip_conv2d.Conv2d on the two stride-2 layer kinds of a BEV stage (3x3 / padding 1 and the bias-free
    1x1 downsample): outputs, input gradients and weight gradients against torch.nn.functional.conv2d, with the operands
    coming from the one-launch presplit of the module tree."""
    from rslo.layers import hip_conv2d
    torch.manual_seed(4)
    root = torch.nn.Sequential(hip_conv2d.Conv2d(64, 32, 3, stride=2, padding=1, bias=False),
                               hip_conv2d.Conv2d(64, 32, 1, stride=2, bias=False)).cuda()
    hip_conv2d.presplit(root)
    for m in root:
        assert getattr(m.weight, "_hip_split", None) is not None
        x = torch.randn(2, 64, 24, 44, device="cuda", requires_grad=True)
        y = m(x)
        assert y.grad_fn is not None and type(y.grad_fn).__name__ in ("_Conv3x3FnBackward", "_Conv1x1S2FnBackward")
        g = torch.randn_like(y)
        y.backward(g)
        xr = x.detach().clone().requires_grad_(True)
        wr = m.weight.detach().clone().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, None, 2, m.padding)
        yr.backward(g)
        for a, b in ((y, yr), (x.grad, xr.grad), (m.weight.grad, wr.grad)):
            assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()), type(y.grad_fn).__name__


def test_hip_conv2d_all_passes_with_presplit_match_library(hip, monkeypatch):
    """RSLO_CONV2D_PASSES=wfd: forward and data gradient through rslo_conv2d_fwd with the operands of all layers split
    in one launch (presplit), refreshed after an in-place weight update."""
    from rslo.layers import hip_conv2d
    monkeypatch.setattr(hip_conv2d, "HIP_PASSES", "wfd")
    torch.manual_seed(6)
    net = torch.nn.Sequential(hip_conv2d.Conv2d(32, 64, 3, padding=1), torch.nn.ReLU(),
                              hip_conv2d.Conv2d(64, 32, 3, padding=1, bias=False)).cuda()
    x = torch.randn(2, 32, 24, 44, device="cuda", requires_grad=True)
    for it in range(2):
        hip_conv2d.presplit(net)
        y = net(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        got = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        x.grad = None; net.zero_grad()
        monkeypatch.setattr(hip_conv2d, "HIP_PASSES", "")
        for m in net:
            if hasattr(m, "_hip_ok"):
                del m._hip_ok
        y2 = net(x)
        y2.backward(gy)
        ref = [y2.detach(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        for a, b in zip(got, ref):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
        x.grad = None; net.zero_grad()
        monkeypatch.setattr(hip_conv2d, "HIP_PASSES", "wfd")
        for m in net:
            if hasattr(m, "_hip_ok"):
                del m._hip_ok
        with torch.no_grad():                   # an "optimizer step": stale operands must not be used
            for p in net.parameters():
                p.mul_(0.5)


@pytest.mark.parametrize("B,cin,cout,H,W", [(4, 64, 7, 24, 44), (2, 32, 1, 96, 176), (3, 32, 7, 7, 9)])
def test_conv1x1_kernels_match_oracle_and_autograd(hip, B, cin, cout, H, W):
    """1x1 output convolutions: forward vs the float64 restatement (1e-5 relative to max|out|); data / weight / bias
    gradients vs torch autograd of the library conv (2e-4 relative), weight gradient bit-reproducible."""
    from rslo.layers.hip_conv2d import Conv2d
    torch.manual_seed(3)
    m = Conv2d(cin, cout, 1).cuda()
    x = torch.randn(B, cin, H, W, device="cuda", requires_grad=True)
    y = m(x)
    assert y.grad_fn is not None and "Conv1x1" in type(y.grad_fn).__name__
    ref = O.conv1x1_fwd(x.detach().cpu().numpy(), m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy())
    assert np.abs(y.detach().cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    gy = torch.randn_like(y)
    y.backward(gy)
    got = [x.grad.clone(), m.weight.grad.clone(), m.bias.grad.clone()]
    x.grad = None; m.zero_grad()
    m.hip_wgrad = False
    m(x).backward(gy)
    for a, b in zip(got, [x.grad, m.weight.grad, m.bias.grad]):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-6
    dW1, _ = hip.conv1x1_wgrad(x.detach(), gy)
    dW2, _ = hip.conv1x1_wgrad(x.detach(), gy)
    assert torch.equal(dW1, dW2)


@pytest.mark.parametrize("B,cin,cout,H,W", [(4, 64, 7, 96, 176), (2, 32, 1, 96, 176), (3, 32, 7, 7, 9), (1, 50, 8, 5, 13),
                                            (2, 3, 2, 4, 4)])
def test_conv1x1_channel_split_kernels_keep_the_bits(hip, B, cin, cout, H, W):
    """k_conv1x1_fwd4 / _dgrad4 (a pixel's channels dealt to the four waves of a workgroup, switch conv1x1_split) add a
    pixel's products in channel order like the one-thread-per-pixel kernels: forward (with and without bias) and data
    gradient are equal bit for bit, ragged channel groups and pixel counts included."""
    torch.manual_seed(5)
    x = torch.randn(B, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 1, 1, device="cuda")
    bias = torch.randn(cout, device="cuda")
    gy = torch.randn(B, cout, H, W, device="cuda")
    with hip.tuning(conv1x1_split=0):
        ref = [hip.conv1x1_fwd(x, w, bias), hip.conv1x1_fwd(x, w, None), hip.conv1x1_dgrad(gy, w)]
    with hip.tuning(conv1x1_split=1):
        got = [hip.conv1x1_fwd(x, w, bias), hip.conv1x1_fwd(x, w, None), hip.conv1x1_dgrad(gy, w)]
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("stride", [1, 2])
def test_basic_block_as_one_autograd_node_gives_the_same_bits(hip, stride, monkeypatch):
    """custom_resnet_spc._BasicBlockFn (one autograd node per BasicBlock) against the layer-by-layer nodes
    (RSLO_FUSED_BLOCK=0) on the same block: outputs, input gradient, every parameter gradient and the BatchNorm running
    statistics are bit-identical -- same kernels, same order; the shortcut gradient is added in place of the engine's sum."""
    import copy
    from rslo.layers import hip_conv2d
    from rslo.layers.MaskConv import MaskConv
    from rslo.layers.SparseConv import FusedSequential, SPC_SyncBN2d
    from rslo.models import custom_resnet_spc as R
    torch.manual_seed(11 + stride)
    inp, planes = (64, 64) if stride == 1 else (64, 128)
    down = None
    if stride == 2:
        down = FusedSequential(R.conv1x1(inp, planes, 2, Conv2d=MaskConv), SPC_SyncBN2d(planes))
    blk = R.BasicBlock(inp, planes, stride, down, BN=SPC_SyncBN2d, Conv2d=MaskConv).cuda().train()
    for p in blk.parameters():
        torch.nn.init.normal_(p, 0.0, 0.2)
    ref = copy.deepcopy(blk)
    x = torch.randn(2, inp, 24, 44, device="cuda")
    g = torch.randn(2, planes, 24 // stride, 44 // stride, device="cuda")

    def run(b, fused):
        monkeypatch.setenv("RSLO_FUSED_BLOCK", "1" if fused else "0")
        b.__dict__.pop("_fused_ok", None)
        hip_conv2d.presplit(b)
        xi = x.clone().requires_grad_(True)
        y = b([xi, None])[0]
        assert (type(y.grad_fn).__name__ == "_BasicBlockFnBackward") == fused
        y.backward(g)
        return y.detach(), xi.grad
    ya, ga = run(blk, True)
    yb, gb = run(ref, False)
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    for (n, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and torch.equal(p.grad, q.grad), n
    for (n, p), (_, q) in zip(blk.named_buffers(), ref.named_buffers()):
        assert torch.equal(p, q), n


def test_conv_bn_act_as_one_autograd_node_gives_the_same_bits(hip, monkeypatch):
    """FusedSequential(Conv2d 3x3 (+bias), SyncBN, ReLU) -- the skip / deblock / prediction trunks of the BEV head --
    as one autograd node (_ConvBNActFn) against the two nodes it replaces: identical outputs, gradients, statistics."""
    import copy
    from rslo.layers import hip_conv2d
    from rslo.layers.SparseConv import FusedSequential, SPC_ReLU, SPC_SyncBN2d
    torch.manual_seed(5)
    seq = FusedSequential(hip_conv2d.Conv2d(64, 32, kernel_size=3, padding=1), SPC_SyncBN2d(32), SPC_ReLU(),
                          hip_conv2d.Conv2d(32, 7, 1)).cuda().train()
    ref = copy.deepcopy(seq)
    x = torch.randn(2, 64, 24, 44, device="cuda")
    g = torch.randn(2, 7, 24, 44, device="cuda")

    def run(m, fused):
        monkeypatch.setenv("RSLO_FUSED_BLOCK", "1" if fused else "0")
        m[0].__dict__.pop("_convbn_ok", None)
        hip_conv2d.presplit(m)
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        y.backward(g)
        names = set()
        fn, todo = y.grad_fn, [y.grad_fn]
        while todo:
            f = todo.pop()
            if f is None:
                continue
            names.add(type(f).__name__)
            todo += [n for n, _ in f.next_functions]
        assert ("_ConvBNActFnBackward" in names) == fused
        return y.detach(), xi.grad
    ya, ga = run(seq, True)
    yb, gb = run(ref, False)
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    for (n, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):
        assert torch.equal(p.grad, q.grad), n
    for (n, p), (_, q) in zip(seq.named_buffers(), ref.named_buffers()):
        assert torch.equal(p, q), n


def test_forked_input_gradient_joins_in_the_data_gradient_kernel(hip):
    """FusedSequential.forward_fork: the map a branch reads is handed on to its other consumers as an output of the branch's
    first node, and their gradient is added in that node's data-gradient epilogue.  Same outputs, input gradient, parameter
    gradients and statistics as the plain form, where the autograd engine adds the two contributions with a launch of its
    own -- also with a chain of two forks (three consumers) and when only the handed-on map is used."""
    import copy
    from rslo.layers import hip_conv2d
    from rslo.layers.SparseConv import FusedSequential, SPC_ReLU, SPC_SyncBN2d
    torch.manual_seed(9)

    def trunk():
        return FusedSequential(hip_conv2d.Conv2d(64, 32, kernel_size=3, padding=1), SPC_SyncBN2d(32), SPC_ReLU(),
                               hip_conv2d.Conv2d(32, 7, 1)).cuda().train()
    a, b = trunk(), trunk()
    ra, rb = copy.deepcopy(a), copy.deepcopy(b)
    x = torch.randn(2, 64, 24, 44, device="cuda")
    w = torch.randn(2, 64, 24, 44, device="cuda")
    g = torch.randn(2, 7, 24, 44, device="cuda")

    def run(ma, mb, fork):
        for m in (ma, mb):
            hip_conv2d.presplit(m)
        xi = x.clone().requires_grad_(True)
        h = xi * 1.0                                  # a non-leaf map, as in the head
        if fork:
            ya, h1 = ma.forward_fork(h)
            yb, h2 = mb.forward_fork(h1)
            assert h1 is not h and h1.data_ptr() == h.data_ptr() and h2.data_ptr() == h.data_ptr()
        else:
            ya, yb, h2 = ma(h), mb(h), h
        ((ya * g).sum() + (yb * g).sum() * 0.5 + (h2 * w).sum()).backward()
        return ya.detach(), yb.detach(), xi.grad
    out_f = run(a, b, True)
    out_p = run(ra, rb, False)
    for u, v in zip(out_f, out_p):
        assert torch.equal(u, v)
    for m, r in ((a, ra), (b, rb)):
        for (n, p), (_, q) in zip(m.named_parameters(), r.named_parameters()):
            assert torch.equal(p.grad, q.grad), n
        for (n, p), (_, q) in zip(m.named_buffers(), r.named_buffers()):
            assert torch.equal(p, q), n
    # only the handed-on map reaches the loss: the branch's own output is dropped
    xi = x.clone().requires_grad_(True)
    _, h1 = a.forward_fork(xi * 1.0)
    (h1 * w).sum().backward()
    assert torch.equal(xi.grad, w)
    # without autograd (evaluation) nothing is forked
    with torch.no_grad():
        y, h = a.forward_fork(x)
    assert h is x and y.shape == (2, 7, 24, 44)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (64, 64), (32, 64), (7, 16)])
def test_sparse_wgrad_xcd_order_keeps_the_bits(hip, cin, cout):
    """The XCD-aware deal of (pair chunk, offset) workgroups (wg2_assign, spconv.hip) only changes WHICH workgroup id
    computes a chunk: dW and dbias are bit-identical to the plain (chunk, offset) grid, fp32 and bf16 rows, also when an
    offset has fewer chunks than XCDs and when the row count is not a multiple of the chunk."""
    rng = np.random.default_rng(77)
    dims, B = [9, 40, 44], 2
    coords = rand_sites(rng, B, dims, 5300)
    n = len(coords)
    x = dev(rng.normal(size=(n, cin)).astype(np.float32))
    gy = dev(rng.normal(size=(n, cout)).astype(np.float32))
    pairs = hip.rulebook_pairs(dev(O.rulebook_subm(coords, B, dims)))
    res = {}
    for mode in ("0", "1"):
        with hip.tuning(wgrad_xcd=int(mode)):
            gw, gb = hip.spconv_wgrad_pairs(x, gy, pairs, n, 27, cin, cout)
            res[mode] = [gw.clone(), gb.clone()]
            if cin in (32, 64) and cout in (32, 64):
                gwb, _ = hip.spconv_wgrad_pairs_bf16(x.bfloat16(), gy.bfloat16(), pairs, n, 27, cin, cout)
                res[mode].append(gwb.clone())
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)
    assert float(res["1"][0].abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,H,W", [(4, 128, 128, 24, 44), (4, 256, 256, 12, 22), (2, 64, 32, 20, 37),
                                             (1, 32, 64, 9, 16), (3, 128, 64, 13, 50)])
def test_conv2d_fwd_xcd_orders_keep_the_bits(hip, B, cin, cout, H, W):
    """conv2d_xcd_tile (conv2d.hip) arranges the 8 XCDs as channel classes x pixel ranges; every arrangement (forced
    through the switch conv2d_fwd_xsc = 1, 2, 4, 8 where it divides the channel groups, 0 = automatic) computes the same
    tiles as the plain grid (-1): identical bits, forward operand and data-gradient operand, fp32 and bf16 operands,
    maps whose tile count is not a multiple of the pixel ranges."""
    torch.manual_seed(5)
    x = torch.randn(B, cin, H, W, device="cuda")
    g = torch.randn(B, cout, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    bias = torch.randn(cout, device="cuda")
    ws, wst = hip.conv2d_wsplit(w, False), hip.conv2d_wsplit(w, True)

    def run():
        return [hip.conv2d_fwd(x, ws, bias, cout).clone(), hip.conv2d_fwd(g, wst, None, cin).clone(),
                hip.conv2d_fwd(x, ws, bias, cout, lp=True).clone()]
    with hip.tuning(conv2d_fwd_xsc=-1):
        ref = run()
    for xsc in (0, 1, 2, 4, 8):
        with hip.tuning(conv2d_fwd_xsc=xsc):
            for a, b in zip(ref, run()):
                assert torch.equal(a, b), xsc
    lib = torch.nn.functional.conv2d(x, w, bias, 1, 1)
    assert float((ref[0] - lib).abs().max()) <= 2e-4 * float(lib.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,H,W", [(4, 128, 128, 48, 88), (1, 64, 64, 96, 176), (2, 96, 64, 25, 23), (4, 256, 128, 48, 88)])
def test_conv2d_fwd_tile_heights_keep_the_bits(hip, B, cin, cout, H, W):
    """4-, 6- and 8-row tiles (switch conv2d_fwd_tr) accumulate every output element over the same (channel chunk, tap, product) sequence: forward with bias and the
    residual epilogue are equal bit for bit."""
    torch.manual_seed(9)
    x = torch.randn(B, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    bias = torch.randn(cout, device="cuda")
    res = torch.randn(B, cout, H, W, device="cuda")
    ws = hip.conv2d_wsplit(w, False)
    with hip.tuning(conv2d_fwd_tr=4):
        ref = [hip.conv2d_fwd(x, ws, bias, cout).clone(), hip.conv2d_fwd(x, ws, None, cout, residual=res).clone()]
    for tr in (0, 6, 8):
        with hip.tuning(conv2d_fwd_tr=tr):
            got = [hip.conv2d_fwd(x, ws, bias, cout), hip.conv2d_fwd(x, ws, None, cout, residual=res)]
        for a, b in zip(ref, got):
            assert torch.equal(a, b), tr


@pytest.mark.gpu
def test_eval_batchnorm1d_with_activation_in_one_launch(hip):
    """rslo_bn1d_eval_act (round 6): eval-mode nn.BatchNorm1d + LeakyReLU of the covariance branch (rslo/models/middle.py:181-213
    under evaluate.py's net.eval()) against torch's F.batch_norm(training=False) + leaky_relu in float64; with a device-side
    live-row count the rows past it stay untouched."""
    import torch.nn.functional as F
    torch.manual_seed(3)
    for n, C in ((31495, 16), (1777, 32), (5, 7)):
        x = torch.randn(n, C, device="cuda") * 2 + 0.5
        rm, rv = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.3
        g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        for slope in (0.01, 0.0, 1.0):
            y = hip.bn1d_eval_act(x, rm, rv, g, b, 1e-3, slope)
            ref = F.batch_norm(x.double(), rm.double(), rv.double(), g.double(), b.double(), False, 0.0, 1e-3)
            ref = F.leaky_relu(ref, slope) if slope != 1.0 else ref
            assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        live = torch.tensor([n // 2], dtype=torch.int32, device="cuda")
        y2 = hip.bn1d_eval_act(x, rm, rv, g, b, 1e-3, 0.01, n_live=live)
        y = hip.bn1d_eval_act(x, rm, rv, g, b, 1e-3, 0.01)
        assert torch.equal(y2[:n // 2], y[:n // 2])


SPCONV_SKIP_CASES = [(32, 32), (64, 64), (32, 64), (64, 32)]


def spconv_skip_case(hip, cin, cout, ks):
    """k_spconv_v6 forward (bias + LeakyReLU) on 32-row tiles, `ks` waves per tile, over a seeded table with whole missing
    16-row blocks, isolated rows and a ragged last tile."""
    g = torch.Generator().manual_seed(5)
    n_in, n_out, K = 5000, 4133, 27
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    keep = torch.rand((n_out, K), generator=g) < 0.45
    blk = torch.arange(n_out) // 16
    keep &= ~(((blk[:, None] * 7 + torch.arange(K)[None]) % 3) == 0)        # a third of the (block, offset) pairs are empty
    keep[torch.arange(0, n_out, 97)] = False                                  # rows without any neighbour
    nbr = torch.where(keep, nbr, torch.full_like(nbr, -1)).cuda().contiguous()
    x = torch.randn(n_in, cin, generator=g).cuda()
    W = (torch.randn(K, cin, cout, generator=g) * 0.1).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    with hip.tuning(spconv_rbw=2, spconv_ks=ks):
        return hip.spconv_fwd(x, W, bias, nbr, act_slope=0.01).clone()


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", SPCONV_SKIP_CASES)
def test_spconv_dead_row_block_skip_keeps_the_recorded_bits(hip, cin, cout):
    """k_spconv_v6 on 32-row tiles skips a 16-row block none of whose rows has the offset (round 6).  A row's sum keeps its
    order: the result equals, bit for bit, what the kernel that always computed both blocks gave -- on a table with whole missing
    blocks, isolated rows, a ragged last tile, one and two waves per tile.  That kernel is deleted; its bits are in
    tests/golden/kernel_bits.json (written by tests/golden/make_golden_kernel_bits.py while both forms existed and agreed)."""
    gold = kernel_bits()
    for ks in (1, 2):
        key = "spconv_skip/%d_%d/ks%d" % (cin, cout, ks)
        assert tensor_bits(spconv_skip_case(hip, cin, cout, ks)) == gold[key], key


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,H,W", [(4, 128, 128, 48, 88), (1, 64, 64, 96, 176), (2, 96, 64, 25, 23), (4, 256, 128, 48, 88),
                                            (1, 32, 32, 7, 9), (3, 64, 192, 12, 22), (1, 512, 128, 24, 44)])
def test_conv2d_lds_weight_kernel_keeps_the_bits(hip, B, cin, cout, H, W):
    """k_conv2d_wl (csrc/conv2d_wl.hip: a workgroup's weight operands copied once into LDS by LDS-DMA, kernel-row stages in a
    two-deep ring, a wave = both 16-channel blocks x 2 rows) accumulates every output element over the same (channel chunk, tap,
    product) sequence as k_conv2d_fwd: forward with bias, the data gradient (transposed operand) and the residual epilogue are
    equal bit for bit -- 8-row and 6-row tiles, ragged edges in both directions, maps smaller than one tile, 1 to 16 chunks."""
    torch.manual_seed(11)
    x = torch.randn(B, cin, H, W, device="cuda")
    g = torch.randn(B, cout, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    bias = torch.randn(cout, device="cuda")
    res = torch.randn(B, cout, H, W, device="cuda")
    ws, wst = hip.conv2d_wsplit(w, False), hip.conv2d_wsplit(w, True)

    def run():
        return [hip.conv2d_fwd(x, ws, bias, cout).clone(), hip.conv2d_fwd(x, ws, None, cout, residual=res).clone(),
                hip.conv2d_fwd(g, wst, None, cin).clone()]
    with hip.tuning(conv2d_fwd_wl=0, conv2d_fwd_kc=1):      # (two wave sets add their halves of the chunks: another order)
        ref = run()
    for mode in (1, 3):
        with hip.tuning(conv2d_fwd_wl=mode):
            for i in range(3):      # repeated launches: a stale weight stage or halo would show as a difference between runs
                for a, b in zip(ref, run()):
                    assert torch.equal(a, b), (mode, i)
    lib = torch.nn.functional.conv2d(x, w, bias, 1, 1)
    assert float((ref[0] - lib).abs().max()) <= 2e-4 * float(lib.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,H,W,k", [(4, 128, 128, 48, 88, 3), (2, 64, 128, 21, 37, 3), (4, 128, 256, 24, 44, 1),
                                               (1, 256, 128, 96, 176, 3)])
def test_conv2d_stride2_xcd_orders_keep_the_bits(hip, B, cin, cout, H, W, k):
    """Same statement for the stride-2 kernels (forward and the four-class data gradient), switch conv2d_s2_xsc."""
    torch.manual_seed(6)
    x = torch.randn(B, cin, H, W, device="cuda")
    g = torch.randn(B, cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") / (k * cin ** 0.5)
    ws, wst = hip.conv2d_wsplit_k(w, False), hip.conv2d_wsplit_k(w, True)

    def run():
        return [hip.conv2d_fwd_s2(x, ws, cout, k).clone(), hip.conv2d_dgrad_s2(g, wst, cin, H, W, k).clone()]
    with hip.tuning(conv2d_s2_xsc=-1):
        ref = run()
    for xsc in (0, 1, 2, 4, 8):
        with hip.tuning(conv2d_s2_xsc=xsc):
            for a, b in zip(ref, run()):
                assert torch.equal(a, b), xsc


def tensor_bits(*tensors):
    """sha256 over the bytes of the tensors: what tests/golden/kernel_bits.json records."""
    import hashlib
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def kernel_bits():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kernel_bits.json")) as f:
        return json.load(f)


CONV2D_S2_CASES = [(4, 128, 128, 48, 88, 3), (2, 64, 128, 21, 37, 3), (4, 128, 256, 24, 44, 1),
                   (1, 256, 128, 96, 176, 3), (1, 32, 64, 13, 21, 1), (2, 64, 64, 5, 3, 3)]


def conv2d_s2_case(hip, B, cin, cout, H, W, k, mtw):
    """Forward, data gradient and data gradient with the joined residual of one stride-2 layer on seeded operands, `mtw`
    16-channel blocks per wave -> the three tensors."""
    torch.manual_seed(8)
    x = torch.randn(B, cin, H, W, device="cuda")
    g = torch.randn(B, cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device="cuda")
    res = torch.randn(B, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") / (k * cin ** 0.5)
    ws, wst = hip.conv2d_wsplit_k(w, False), hip.conv2d_wsplit_k(w, True)
    with hip.tuning(conv2d_s2_mtw=mtw):
        return [hip.conv2d_fwd_s2(x, ws, cout, k).clone(), hip.conv2d_dgrad_s2(g, wst, cin, H, W, k).clone(),
                hip.conv2d_dgrad_s2(g, wst, cin, H, W, k, residual=res).clone()]


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,H,W,k", CONV2D_S2_CASES)
def test_conv2d_stride2_kernels_keep_the_recorded_bits(hip, B, cin, cout, H, W, k):
    """k_conv2d_str2 (compile-time tap lists, operands one tap ahead, parity-ordered halo columns) forms the same products in the
    same order as its run-time-loop predecessor k_conv2d_str did: forward and data gradient (with and without the joined
    residual), one and two 16-channel blocks per wave, ragged maps.  Until round 6 this test ran both kernels; k_conv2d_str is
    deleted, its bits are in tests/golden/kernel_bits.json (written by tests/golden/make_golden_kernel_bits.py while both
    kernels existed and agreed)."""
    gold = kernel_bits()
    for mtw in (1, 2):
        if mtw == 2 and (cin % 64 or cout % 64):
            continue
        key = "conv2d_s2/%d_%d_%d_%d_%d_k%d/mtw%d" % (B, cin, cout, H, W, k, mtw)
        assert tensor_bits(*conv2d_s2_case(hip, B, cin, cout, H, W, k, mtw)) == gold[key], key


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 63, 64, 65, 300, 4097])
def test_vfe_mean_wave_tail_sizes_match_oracle(hip, M):
    """k_vfe_mean_lds stages the rows of 64 voxels per wave through LDS: voxel counts around the wave / block boundaries
    (last wave partly filled, last block with idle waves) against the oracle's sequential sums."""
    rng = np.random.default_rng(M)
    T, F = 10, 7
    num = rng.integers(1, T + 1, size=M).astype(np.int32)
    vox = (rng.normal(size=(M, T, F)) * 10).astype(np.float32)
    vox *= (np.arange(T)[None, :, None] < num[:, None, None])
    got = hip.vfe_mean(dev(vox), dev(num)).cpu().numpy()
    np.testing.assert_allclose(got, O.vfe_mean(vox, num), rtol=2e-6, atol=2e-6)
