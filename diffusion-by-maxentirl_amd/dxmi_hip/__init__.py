"""dxmi_hip — Python host side of the gfx950 kernel library (see include/dxmi_hip.h)."""
from ._lib import DxmiError, LIB_PATH, load  # noqa: F401
from . import ops  # noqa: F401
