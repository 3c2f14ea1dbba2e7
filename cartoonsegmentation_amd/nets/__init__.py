"""Net builders: lower the reference's dense nets into libcsm355 layer programs."""
from .isnet import build_isnet  # noqa: F401
