"""Net builders: lower the reference's dense nets into libcsm355 layer programs."""
from .isnet import build_isnet  # noqa: F401
from .leres import build_leres  # noqa: F401
from .rtmdet import build_rtmdet, RTMDetConfig  # noqa: F401
