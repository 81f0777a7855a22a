"""htscodecs_amd — MI355X (gfx950) implementation of the CRAM 3.1 rANS 4x16 codec.

The product is the C-ABI shared library ``librans4x16_hip.so`` (see include/rans4x16_hip.h);
this package is the thin Python mirror of that interface used by the tests and the benchmark:

* :mod:`htscodecs_amd.lib`  — ctypes binding of every exported symbol (fails loudly if the
  library has not been built; there is no Python or CPU implementation to fall back to);
* :mod:`htscodecs_amd.codec` — the reference's five functions on ``bytes`` plus batch calls on
  device-resident torch tensors.
"""
from .lib import load, LibraryNotBuilt, STATUS_NAMES  # noqa: F401
from .codec import (  # noqa: F401
    rans_compress_bound_4x16, rans_compress_4x16, rans_uncompress_4x16,
    compress_batch, compress_best_batch, uncompress_batch, DeviceCodec, MultiCodec,
    rans_compress, rans_uncompress, compress_batch_4x8, uncompress_batch_4x8,
)

__all__ = [
    "load", "LibraryNotBuilt", "STATUS_NAMES",
    "rans_compress_bound_4x16", "rans_compress_4x16", "rans_uncompress_4x16",
    "compress_batch", "compress_best_batch", "uncompress_batch", "DeviceCodec", "MultiCodec",
    "rans_compress", "rans_uncompress", "compress_batch_4x8", "uncompress_batch_4x8",
]
