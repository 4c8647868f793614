"""spx — MI355X-native sparse 3-D convolution operators (libspx.so behind a C ABI, include/spx.h).

`import spx as spconv` gives the `spconv.pytorch` surface the reference uses
(pcdet/utils/spconv_utils.py:3-6): SparseConvTensor, SubMConv3d, SparseConv3d, SparseInverseConv3d,
SparseSequential, SparseModule, and `spconv.conv.SparseConvolution`.
"""
from . import _lib, ops  # noqa: F401
from . import modules as conv  # `spconv.conv.SparseConvolution`
from .modules import (SparseConv3d, SparseConvolution, SparseInverseConv3d, SparseModule, SparseSequential,  # noqa: F401
                      SubMConv3d, ToDense, is_spconv_module)
from .tensor import SparseConvTensor  # noqa: F401
from .voxel import Point2VoxelCPU3d, PointToVoxel  # noqa: F401

__version__ = "0.1.0"
