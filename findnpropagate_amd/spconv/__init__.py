"""spconv-shaped API (the subset pcdet uses) on the MI355X kernels.

Mirrors: spconv.SparseConvTensor, SubMConv3d, SparseConv3d, SparseSequential, SparseModule,
spconv.conv.SparseConvolution, spconv.constants, spconv.utils.Point2VoxelCPU3d / VoxelGenerator,
spconv.__version__ (parsed by pcdet/utils/spconv_utils.py:4 as float(__version__[2:])).
"""
__version__ = "2.3.6"

from . import constants  # noqa: E402,F401
from .core import SparseConvTensor  # noqa: E402,F401
from .modules import SparseModule, SparseSequential  # noqa: E402,F401
from .conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d  # noqa: E402,F401
from . import conv, modules, utils  # noqa: E402,F401
from . import pytorch  # noqa: E402,F401  (INTEGRATION.md §2 aliases `spconv.pytorch` from this attribute)
