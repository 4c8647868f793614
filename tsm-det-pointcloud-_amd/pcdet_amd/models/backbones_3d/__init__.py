from .spconv_backbone import VoxelBackBone8x, VoxelResBackBone8x
from .spconv_unet import UNetV2

__all__ = {
    'VoxelBackBone8x': VoxelBackBone8x,
    'VoxelResBackBone8x': VoxelResBackBone8x,
    'UNetV2': UNetV2,
}
