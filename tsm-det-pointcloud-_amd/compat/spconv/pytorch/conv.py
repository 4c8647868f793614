from spx.modules import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d  # noqa: F401
