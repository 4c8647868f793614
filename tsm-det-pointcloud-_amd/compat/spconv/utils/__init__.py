"""`from spconv.utils import Point2VoxelCPU3d` (data_processor.py:26) -> the HIP voxeliser."""
from spx.voxel import Point2VoxelCPU3d, PointToVoxel  # noqa: F401
