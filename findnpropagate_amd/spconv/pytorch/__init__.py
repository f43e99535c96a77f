"""`import spconv.pytorch as spconv` (pcdet/utils/spconv_utils.py:7-10) resolves here."""
from ..core import SparseConvTensor  # noqa: F401
from ..modules import SparseModule, SparseSequential  # noqa: F401
from ..conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d  # noqa: F401
from .. import conv, constants, modules, utils  # noqa: F401
from .. import __version__  # noqa: F401
