"""thirdparty.chamfer_distance.chamfer_distance on the MI355X nearest-neighbour kernel.

Same surface as the reference module (thirdparty/chamfer_distance/chamfer_distance.py:234-246 and the
pybind module `cd`, chamfer_distance.cpp:237-244): autograd functions ChamferDistance,
ChamferDistanceWithIdx, OneDirectionChamferDistanceWithIdx, and a `cd` namespace with the six raw
entry points that write into caller-allocated tensors.  Device tensors run through rslo_chamfer_nn /
rslo_chamfer_grad of the C ABI; the libraries are prebuilt, nothing is JIT-compiled at import.
Dispatch is the reference's: `cd.forward` / `cd.backward` and the two-direction autograd functions
take HOST tensors too (the reference computes those on the host, chamfer_distance.cpp:147-234,
chosen at chamfer_distance.py:34,61) -- served by librslo_host.so (rslo_host_chamfer_forward /
_backward, plain C, include/rslo_host.h), same dist / idx bits as the device kernel; the
one-direction functions the loss uses are CUDA-only and raise NotImplementedError for CPU tensors, as
the reference does (chamfer_distance.py:174-175).  A device tensor never takes the host route.
"""
import torch

from rslo_amd import capi


def _on_host(*tensors):
    host = [not t.is_cuda for t in tensors]
    if any(host) and not all(host):
        raise ValueError("chamfer_distance: tensors on different devices")
    return all(host)


class _CD:
    """The six `cd.*` entry points (caller-allocated outputs, return None)."""

    @staticmethod
    def forward_cuda_one_direction(xyz1, xyz2, dist1, idx1):
        capi.chamfer_nn(xyz1, xyz2, dist1, idx1)

    @staticmethod
    def forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2):
        capi.chamfer_nn(xyz1, xyz2, dist1, idx1)
        capi.chamfer_nn(xyz2, xyz1, dist2, idx2)

    @staticmethod
    def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
        """The reference's `cd.forward` is its HOST implementation (chamfer_distance.cpp:147-190 `nnsearch` on CPU
        tensors, selected at chamfer_distance.py:34 when the inputs are not CUDA): host tensors -> librslo_host.so, device
        tensors -> the kernel (same result as forward_cuda)."""
        if _on_host(xyz1, xyz2, dist1, dist2, idx1, idx2):
            from rslo_amd import hostlib
            return hostlib.chamfer_forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        _CD.forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)

    @staticmethod
    def backward_cuda_one_direction(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, idx1):
        capi.chamfer_grad(xyz1, xyz2, graddist1, idx1, gradxyz1, gradxyz2)

    @staticmethod
    def backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        capi.chamfer_grad(xyz1, xyz2, graddist1, idx1, gradxyz1, gradxyz2)
        g2b, g1b = torch.empty_like(xyz2), torch.empty_like(xyz1)
        capi.chamfer_grad(xyz2, xyz1, graddist2, idx2, g2b, g1b)
        gradxyz1 += g1b
        gradxyz2 += g2b

    @staticmethod
    def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        """Reference: the HOST backward (chamfer_distance.cpp:192-234, chamfer_distance.py:61); see forward."""
        if _on_host(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
            from rslo_amd import hostlib
            return hostlib.chamfer_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
        _CD.backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)


cd = _CD()


def _prep(x):
    return x.contiguous().float()


class OneDirectionChamferDistanceWithIdxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = _prep(xyz1), _prep(xyz2)
        dist1, idx1 = capi.chamfer_nn(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1)
        ctx.mark_non_differentiable(idx1)
        return dist1, idx1

    @staticmethod
    def backward(ctx, graddist1, _gradidx1):
        xyz1, xyz2, idx1 = ctx.saved_tensors
        g1, g2 = capi.chamfer_grad(xyz1, xyz2, graddist1.contiguous(), idx1)
        return g1, g2


class ChamferDistanceWithIdxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = _prep(xyz1), _prep(xyz2)
        if _on_host(xyz1, xyz2):             # chamfer_distance.py:34: the module's host branch
            b, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
            dist1, dist2 = torch.zeros(b, n), torch.zeros(b, m)
            idx1, idx2 = torch.zeros(b, n, dtype=torch.int32), torch.zeros(b, m, dtype=torch.int32)
            cd.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        else:
            dist1, idx1 = capi.chamfer_nn(xyz1, xyz2)
            dist2, idx2 = capi.chamfer_nn(xyz2, xyz1)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, _g1, _g2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        g1 = torch.empty_like(xyz1)
        g2 = torch.empty_like(xyz2)
        if _on_host(xyz1, xyz2):             # chamfer_distance.py:61
            cd.backward(xyz1, xyz2, g1, g2, graddist1.contiguous(), graddist2.contiguous(), idx1, idx2)
        else:
            cd.backward_cuda(xyz1, xyz2, g1, g2, graddist1.contiguous(), graddist2.contiguous(), idx1, idx2)
        return g1, g2


class ChamferDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        d1, d2, i1, i2 = ChamferDistanceWithIdxFunction.forward(ctx, xyz1, xyz2)
        return d1, d2

    @staticmethod
    def backward(ctx, graddist1, graddist2):
        return ChamferDistanceWithIdxFunction.backward(ctx, graddist1, graddist2, None, None)


class ChamferDistance(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        return ChamferDistanceFunction.apply(xyz1, xyz2)


class ChamferDistanceWithIdx(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        return ChamferDistanceWithIdxFunction.apply(xyz1, xyz2)


class OneDirectionChamferDistanceWithIdx(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        return OneDirectionChamferDistanceWithIdxFunction.apply(xyz1, xyz2)
