"""`import spconv.pytorch as spconv` -> libspx-backed modules."""
from spx import (SparseConv3d, SparseConvolution, SparseConvTensor, SparseInverseConv3d, SparseModule,  # noqa: F401
                 SparseSequential, SubMConv3d, ToDense, conv, ops)
from spx import modules  # noqa: F401
