"""cartoonsegmentation_amd -- MI355X (gfx950) native hot path of CartoonSegmentation.

The arithmetic lives in libcsm355.so (hand-written HIP, C ABI in include/csm355.h);
this package is the thin Python host layer that mirrors the reference's operator
signatures.  PyTorch is used for device memory, streams and torch.distributed only.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
