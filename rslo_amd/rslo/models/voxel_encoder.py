"""Voxel feature extractor of the hot path (reference: rslo/models/voxel_encoder.py:11-26,258-280)."""
import torch
from torch import nn

from rslo_amd import capi

REGISTERED_VFE_CLASSES = {}


def register_vfe(cls, name=None):
    name = cls.__name__ if name is None else name
    assert name not in REGISTERED_VFE_CLASSES, f"exist class: {REGISTERED_VFE_CLASSES}"
    REGISTERED_VFE_CLASSES[name] = cls
    return cls


def get_vfe_class(name):
    assert name in REGISTERED_VFE_CLASSES, f"available class: {REGISTERED_VFE_CLASSES}"
    return REGISTERED_VFE_CLASSES[name]


@register_vfe
class SimpleVoxel_XYZINormalC(nn.Module):
    """Mean of the (<=T) points of each voxel over the first `num_input_features` channels; the normal
    channels 4:7 are re-normalised (+1e-12).  Parameter-free; runs on rslo_vfe_mean."""

    def __init__(self, num_input_features=8, use_norm=True, num_filters=[32, 128], with_distance=False,
                 voxel_size=(0.2, 0.2, 4), pc_range=(0, -40, -3, 70.4, 40, 1), name="VoxelFeatureExtractor"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        f = features[:, :, :self.num_input_features]
        if f.shape[-1] != 7:
            raise NotImplementedError("SimpleVoxel_XYZINormalC expects 7 point features (x,y,z,i,nx,ny,nz)")
        # rslo_vfe_mean is an fp32 kernel; other dtypes only occur when the test harness swaps the backend
        f = f.contiguous()
        return capi.vfe_mean(f.float() if f.is_cuda else f, num_voxels.int().contiguous())


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
