"""ctypes access to the CPU checkers: the oracle (our C restatement) and, when it has been
built from /root/reference by `make -C oracle ref`, the real reference library.

Test infrastructure only: nothing in htscodecs_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_u8p = C.POINTER(C.c_ubyte)


class _Codec:
    """Wraps one library exporting the five htscodecs entry points (optionally prefixed)."""

    def __init__(self, path, prefix=""):
        self.lib = C.CDLL(path)
        self.path = path
        self._prefix = prefix
        g = lambda n: getattr(self.lib, prefix + n)
        self.bound = g("rans_compress_bound_4x16")
        self.bound.restype = C.c_uint
        self.bound.argtypes = [C.c_uint, C.c_int]
        self.compress_to = g("rans_compress_to_4x16")
        self.compress_to.restype = C.c_void_p
        self.compress_to.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.POINTER(C.c_uint), C.c_int]
        self.uncompress_to = g("rans_uncompress_to_4x16")
        self.uncompress_to.restype = C.c_void_p
        self.uncompress_to.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.POINTER(C.c_uint)]

    def uncompress_malloc(self, comp):
        """The out == NULL entry (rans_uncompress_4x16, rANS_static4x16pr.c:1638): the callee sizes and mallocs
        the result from the stream's own size field.  bytes -> bytes (None on failure)."""
        fn = getattr(self.lib, self._prefix + "rans_uncompress_4x16")
        fn.restype = C.c_void_p
        fn.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]
        src = np.ascontiguousarray(np.frombuffer(bytes(comp), dtype=np.uint8))
        n = C.c_uint(0)
        p = fn(src.ctypes.data, len(src), C.byref(n))
        if not p:
            return None
        res = C.string_at(p, n.value)
        C.CDLL(None).free(C.c_void_p(p))
        return res

    def compress(self, data, order):
        """bytes-like -> bytes (None on failure)."""
        src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        cap = self.bound(len(src), order)
        out = np.zeros(cap + 16, dtype=np.uint8)
        n = C.c_uint(cap)
        r = self.compress_to(src.ctypes.data if len(src) else out.ctypes.data, len(src),
                             out.ctypes.data, C.byref(n), order)
        if not r:
            return None
        return out[:n.value].tobytes()

    def uncompress(self, comp, capacity=None, out_size_hint=None):
        """bytes -> bytes (None on failure).  capacity: size of the caller's buffer."""
        src = np.ascontiguousarray(np.frombuffer(bytes(comp), dtype=np.uint8))
        if capacity is None:
            capacity = max(peek_ulen(comp), 0)
        out = np.zeros(capacity + 16, dtype=np.uint8)
        n = C.c_uint(capacity if out_size_hint is None else out_size_hint)
        r = self.uncompress_to(src.ctypes.data, len(src), out.ctypes.data, C.byref(n))
        if not r:
            return None
        return out[:n.value].tobytes()


def peek_ulen(comp):
    """Uncompressed length stored in a container header (0 if X_NOSZ)."""
    if len(comp) < 2 or (comp[0] & 0x10):
        return 0
    v, i = 0, 1
    while i < len(comp):
        v = ((v << 7) | (comp[i] & 0x7F)) & 0xFFFFFFFF
        if not comp[i] & 0x80:
            break
        i += 1
    return v


def _build(target):
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, target], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


_oracle = None


def oracle():
    """The C restatement (built on demand with gcc; ~1 s)."""
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, "liboracle4x16.so")
        src = os.path.join(ORACLE_DIR, "rans4x16_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            _build("all")
        _oracle = _Codec(so, prefix="orc_")
        lib = _oracle.lib
        for name in ("orc_compress_many", "orc_uncompress_many"):
            getattr(lib, name).restype = C.c_int
    return _oracle
