"""CPU oracle for the RSLO hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (rslo_amd/) never does; it fails loudly when its HIP
library is missing instead of falling back to anything here.

numpy front-end over librslo_oracle.so (oracle/rslo_oracle.c, built by oracle/Makefile).
Parity status of each function is stated in the C file's header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(verbose=False):
    """(Re)build librslo_oracle.so with gcc if missing or stale."""
    so = os.path.join(_HERE, "librslo_oracle.so")
    src = os.path.join(_HERE, "rslo_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
        if verbose:
            print(out.stdout)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_voxelize.restype = C.c_int64
        _LIB.orc_conv_out_coords.restype = C.c_int64
    return _LIB


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def _i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def grid_size(pc_range, voxel_size):
    """spconv VoxelGenerator.grid_size: round((hi - lo) / vs) in float32 (x, y, z)."""
    r = np.asarray(pc_range, np.float32)
    v = np.asarray(voxel_size, np.float32)
    return np.round((r[3:] - r[:3]) / v).astype(np.int64)


def voxelize(points, pc_range, voxel_size, max_points, max_voxels):
    """-> voxels [M,T,F] f32, coords [M,3] i32 (z,y,x), num_points [M] i32."""
    pts = _f32(points)
    P, F = pts.shape
    g = _i32(grid_size(pc_range, voxel_size))
    vox = np.empty((max_voxels, max_points, F), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    npts = np.empty((max_voxels,), np.int32)
    r = _f32(pc_range)
    v = _f32(voxel_size)
    M = lib().orc_voxelize(_p(pts), C.c_int64(P), F, _p(r), _p(v), _p(g), int(max_points),
                           int(max_voxels), _p(vox), _p(coords), _p(npts))
    if M < 0:
        raise MemoryError
    return vox[:M].copy(), coords[:M].copy(), npts[:M].copy()


def vfe_mean(voxels, num_points):
    """SimpleVoxel_XYZINormalC.forward (rslo/models/voxel_encoder.py:272-280), fp32."""
    v = _f32(voxels)
    n = np.asarray(num_points).astype(np.float32).reshape(-1, 1)
    m = v.sum(axis=1, dtype=np.float32) / n
    nrm = np.sqrt((m[:, 4:7] * m[:, 4:7]).sum(-1, keepdims=True, dtype=np.float32))
    m[:, 4:7] = m[:, 4:7] / (nrm + np.float32(1e-12))
    return m.astype(np.float32)


def conv_out_dims(in_dims, ks, stride, pad):
    return [(int(d) + 2 * int(p) - int(k)) // int(s) + 1 for d, k, s, p in zip(in_dims, ks, stride, pad)]


def rulebook_subm(coords, batch, dims, ks=(3, 3, 3)):
    c = _i32(coords)
    N = c.shape[0]
    K = int(np.prod(ks))
    nbr = np.empty((N, K), np.int32)
    rc = lib().orc_rulebook_subm(_p(c), C.c_int64(N), int(batch), _p(_i32(dims)), _p(_i32(ks)), _p(nbr))
    assert rc == 0
    return nbr


def rulebook_conv(coords, batch, in_dims, ks, stride, pad):
    """-> out_coords [M,4], out_dims, nbr [M,K] (in row per (out,k)), nbrT [N,K] (out row per (in,k))."""
    c = _i32(coords)
    N = c.shape[0]
    K = int(np.prod(ks))
    od = conv_out_dims(in_dims, ks, stride, pad)
    cap = int(min(N * K, batch * od[0] * od[1] * od[2]))
    oc = np.empty((max(cap, 1), 4), np.int32)
    ks_, st_, pd_, id_, od_ = _i32(ks), _i32(stride), _i32(pad), _i32(in_dims), _i32(od)
    M = lib().orc_conv_out_coords(_p(c), C.c_int64(N), int(batch), _p(id_), _p(ks_), _p(st_), _p(pd_),
                                  _p(od_), _p(oc), C.c_int64(cap))
    assert 0 <= M <= cap
    oc = oc[:M].copy()
    nbr = np.empty((M, K), np.int32)
    nbrT = np.empty((N, K), np.int32)
    rc = lib().orc_conv_tables(_p(c), C.c_int64(N), _p(oc), C.c_int64(M), int(batch), _p(id_), _p(ks_),
                               _p(st_), _p(pd_), _p(od_), _p(nbr), _p(nbrT))
    assert rc == 0
    return oc, od, nbr, nbrT


def pairs_from_nbrT(nbrT):
    """spconv-1.x layout: indice_pairs [K,2,cap] (-1 padded), indice_num [K]."""
    t = _i32(nbrT)
    N, K = t.shape
    cap = max(int((t >= 0).sum(0).max()) if N else 0, 1)
    pairs = np.full((K, 2, cap), -1, np.int32)
    num = np.zeros((K,), np.int32)
    lib().orc_pairs_from_nbrT(_p(t), C.c_int64(N), K, C.c_int64(cap), _p(pairs), _p(num))
    return pairs, num


def spconv_fwd(x, W, bias, nbr):
    x, W = _f32(x), _f32(W)
    nbr = _i32(nbr)
    M, K = nbr.shape
    Cin, Cout = W.shape[-2], W.shape[-1]
    W = W.reshape(K, Cin, Cout)
    out = np.empty((M, Cout), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_spconv_fwd(_p(x), Cin, _p(W), _p(b) if b is not None else None, _p(nbr), C.c_int64(M), K, Cout,
                         _p(out))
    return out


def spconv_dgrad(dout, W, nbrT):
    dout, W = _f32(dout), _f32(W)
    nbrT = _i32(nbrT)
    N, K = nbrT.shape
    Cin, Cout = W.shape[-2], W.shape[-1]
    W = W.reshape(K, Cin, Cout)
    din = np.empty((N, Cin), np.float32)
    lib().orc_spconv_dgrad(_p(dout), Cout, _p(W), _p(nbrT), C.c_int64(N), K, Cin, _p(din))
    return din


def spconv_wgrad(x, dout, nbr, Cin, Cout):
    x, dout = _f32(x), _f32(dout)
    nbr = _i32(nbr)
    M, K = nbr.shape
    dW = np.empty((K, Cin, Cout), np.float32)
    db = np.empty((Cout,), np.float32)
    lib().orc_spconv_wgrad(_p(x), Cin, _p(dout), Cout, _p(nbr), C.c_int64(M), K, _p(dW), _p(db))
    return dW, db


def dense(feat, coords, batch, dims):
    f = _f32(feat)
    c = _i32(coords)
    M, Cc = f.shape
    out = np.empty((batch, Cc, dims[0], dims[1], dims[2]), np.float32)
    lib().orc_dense(_p(f), _p(c), C.c_int64(M), Cc, int(batch), _p(_i32(dims)), _p(out))
    return out


def chamfer_nn(xyz1, xyz2):
    """[B,N,3],[B,M,3] -> dist [B,N] f32, idx [B,N] i32 (chamfer_distance.cpp:116-144)."""
    a, b = _f32(xyz1), _f32(xyz2)
    B, N, _ = a.shape
    M = b.shape[1]
    dist = np.empty((B, N), np.float32)
    idx = np.empty((B, N), np.int32)
    lib().orc_chamfer_nn(B, N, M, _p(a), _p(b), _p(dist), _p(idx))
    return dist, idx


def chamfer_grad(xyz1, xyz2, graddist1, idx1):
    a, b = _f32(xyz1), _f32(xyz2)
    g = _f32(graddist1)
    i = _i32(idx1)
    B, N, _ = a.shape
    M = b.shape[1]
    g1 = np.empty((B, N, 3), np.float32)
    g2 = np.empty((B, M, 3), np.float32)
    lib().orc_chamfer_grad(B, N, M, _p(a), _p(b), _p(g), _p(i), _p(g1), _p(g2))
    return g1, g2


def conv2d_wgrad(x, dout, stride=1):
    """Weight gradient of the BEV head's dense 3x3 / padding-1 Conv2d layers (torch.nn.Conv2d built in
    rslo/models/odom_pred.py:65-134 and rslo/layers/MaskConv.py:33-37; cross-correlation, zero padding):
    dW[o,i,ky,kx] = sum_{b,y,x} dout[b,o,y,x] * xpad[b,i,S*y+ky,S*x+kx], accumulated in float64.
    x [B,Cin,H,W], dout [B,Cout,Ho,Wo] -> [Cout,Cin,3,3] float64."""
    x = np.asarray(x, np.float64)
    g = np.asarray(dout, np.float64)
    B, Cin, H, W = x.shape
    Ho, Wo = g.shape[2], g.shape[3]
    xp = np.zeros((B, Cin, H + 2, W + 2))
    xp[:, :, 1:H + 1, 1:W + 1] = x
    dW = np.empty((g.shape[1], Cin, 3, 3))
    for ky in range(3):
        for kx in range(3):
            win = xp[:, :, ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride]
            dW[:, :, ky, kx] = np.einsum("boyx,biyx->oi", g, win, optimize=True)
    return dW


def conv2d_fwd(x, w, bias=None):
    """Dense 3x3 / stride-1 / padding-1 Conv2d forward (cross-correlation, zero padding; torch.nn.Conv2d as built in
    rslo/models/odom_pred.py:65-134), float64.  x [B,Cin,H,W], w [Cout,Cin,3,3] -> [B,Cout,H,W]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    B, Cin, H, W = x.shape
    xp = np.zeros((B, Cin, H + 2, W + 2))
    xp[:, :, 1:H + 1, 1:W + 1] = x
    out = np.zeros((B, w.shape[0], H, W))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("oi,biyx->boyx", w[:, :, ky, kx], xp[:, :, ky:ky + H, kx:kx + W], optimize=True)
    if bias is not None:
        out += np.asarray(bias, np.float64)[None, :, None, None]
    return out


def conv2d_dgrad(dout, w):
    """Data gradient of the same layer: dx[b,i,y,x] = sum_{o,ky,kx} dout[b,o,y-ky+1,x-kx+1] w[o,i,ky,kx], float64."""
    g = np.asarray(dout, np.float64)
    w = np.asarray(w, np.float64)
    B, Cout, H, W = g.shape
    gp = np.zeros((B, Cout, H + 2, W + 2))
    gp[:, :, 1:H + 1, 1:W + 1] = g
    dx = np.zeros((B, w.shape[1], H, W))
    for ky in range(3):
        for kx in range(3):
            dx += np.einsum("oi,boyx->biyx", w[:, :, ky, kx], gp[:, :, 2 - ky:2 - ky + H, 2 - kx:2 - kx + W], optimize=True)
    return dx


def conv2d_s2_fwd(x, w):
    """Bias-free stride-2 Conv2d of the BEV stages, float64: 3x3 / padding 1 (BasicBlock.conv1 of a stage's first block,
    rslo/models/custom_resnet_spc.py:224-260) or 1x1 / padding 0 (its downsample branch, odom_pred.py:404-406).
    x [B,Cin,H,W], w [Cout,Cin,k,k] -> [B,Cout,(H-1)//2+1,(W-1)//2+1]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    B, Cin, H, W = x.shape
    k = w.shape[2]
    pad = (k - 1) // 2
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xp = np.zeros((B, Cin, H + 2 * pad + 1, W + 2 * pad + 1))
    xp[:, :, pad:pad + H, pad:pad + W] = x
    out = np.zeros((B, w.shape[0], Ho, Wo))
    for ky in range(k):
        for kx in range(k):
            out += np.einsum("oi,biyx->boyx", w[:, :, ky, kx], xp[:, :, ky:ky + 2 * Ho:2, kx:kx + 2 * Wo:2], optimize=True)
    return out


def conv2d_s2_dgrad(dout, w, H, W):
    """Data gradient of conv2d_s2_fwd for an [H, W] input: dx[b,i,2y+ky-pad,2x+kx-pad] += dout[b,o,y,x] w[o,i,ky,kx]."""
    g = np.asarray(dout, np.float64)
    w = np.asarray(w, np.float64)
    B, Cout, Ho, Wo = g.shape
    k = w.shape[2]
    pad = (k - 1) // 2
    dxp = np.zeros((B, w.shape[1], H + 2 * pad + 1, W + 2 * pad + 1))
    for ky in range(k):
        for kx in range(k):
            dxp[:, :, ky:ky + 2 * Ho:2, kx:kx + 2 * Wo:2] += np.einsum("oi,boyx->biyx", w[:, :, ky, kx], g, optimize=True)
    return dxp[:, :, pad:pad + H, pad:pad + W]


def conv1x1_fwd(x, w, bias=None):
    """1x1 Conv2d (torch.nn.Conv2d(c, 7, 1), rslo/models/odom_pred.py:71): out[b,o,y,x] = bias[o] + sum_i w[o,i] x[b,i,y,x]."""
    out = np.einsum("oi,biyx->boyx", np.asarray(w, np.float64)[:, :, 0, 0], np.asarray(x, np.float64))
    if bias is not None:
        out = out + np.asarray(bias, np.float64)[None, :, None, None]
    return out
