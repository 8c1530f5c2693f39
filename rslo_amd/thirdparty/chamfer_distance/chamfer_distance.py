"""thirdparty.chamfer_distance.chamfer_distance on the MI355X nearest-neighbour kernel.

Same surface as the reference module (thirdparty/chamfer_distance/chamfer_distance.py:234-246 and the
pybind module `cd`, chamfer_distance.cpp:237-244): autograd functions ChamferDistance,
ChamferDistanceWithIdx, OneDirectionChamferDistanceWithIdx, and a `cd` namespace with the six raw
entry points that write into caller-allocated tensors.  Everything runs through rslo_chamfer_nn /
rslo_chamfer_grad of the C ABI; the library is prebuilt, nothing is JIT-compiled at import, and CPU
tensors are refused (the reference raises NotImplementedError for them too,
chamfer_distance.py:174-175).
"""
import torch

from rslo_amd import capi


def _refuse_host(name, *tensors):
    if any(not t.is_cuda for t in tensors):
        raise NotImplementedError(
            "%s: in the reference this entry point computes on the HOST for CPU tensors (chamfer_distance.cpp:147-234, "
            "chosen at chamfer_distance.py:34,61); rslo_amd serves the chamfer path on the GPU only and has no CPU "
            "fallback -- move the tensors to the device (same results: the kernel is bit-exact against the reference's "
            "host code, tests/golden/chamfer_ref.npz)" % name)


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
        tensors, selected at chamfer_distance.py:34 when the inputs are not CUDA).  This package has no CPU compute path:
        device tensors are served (same result as forward_cuda), host tensors are refused with the difference named."""
        _refuse_host("cd.forward", xyz1, xyz2)
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
        _refuse_host("cd.backward", xyz1, xyz2)
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
