"""ctypes binding of librans4x16_hip.so (include/rans4x16_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# R4X16_LIB: another build of the SAME library (tools/build_variant.sh -> htscodecs_amd/variants/lib<name>.so) for A/B
# measurements; it must exist and export the whole ABI, there is no fallback either way.
LIB_PATH = os.environ.get("R4X16_LIB") or os.path.join(_HERE, "librans4x16_hip.so")

STATUS_NAMES = {0: "OK", 1: "CAPACITY", 2: "TRUNCATED", 3: "TABLE", 4: "STATE", 5: "SIZE",
                6: "UNSUPPORTED", 7: "CONTEXT", 8: "RLE", 9: "EMPTY"}

# name -> (restype, argtypes); the single source of truth that tests/test_cabi.py checks against
# the declarations in include/rans4x16_hip.h.
_u8p = C.c_void_p
SIGNATURES = {
    "rans_compress_bound_4x16": (C.c_uint, [C.c_uint, C.c_int]),
    "rans_compress_to_4x16": (C.c_void_p, [_u8p, C.c_uint, _u8p, C.POINTER(C.c_uint), C.c_int]),
    "rans_compress_4x16": (C.c_void_p, [_u8p, C.c_uint, C.POINTER(C.c_uint), C.c_int]),
    "rans_uncompress_to_4x16": (C.c_void_p, [_u8p, C.c_uint, _u8p, C.POINTER(C.c_uint)]),
    "rans_uncompress_4x16": (C.c_void_p, [_u8p, C.c_uint, C.POINTER(C.c_uint)]),
    "rans4x16_hip_create": (C.c_void_p, [C.c_int]),
    "rans4x16_hip_destroy": (None, [C.c_void_p]),
    "rans4x16_hip_last_error": (C.c_char_p, [C.c_void_p]),
    "rans4x16_hip_compress_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "rans4x16_hip_uncompress_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    "rans4x16_hip_compress_best_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rans4x16_hip_compress_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rans4x16_hip_uncompress_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_uint32, C.c_uint32, C.c_void_p]),
    "rans4x16_hip_compress_dev_sized": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]),
    "rans4x16_hip_uncompress_dev_sized": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]),
    "rans4x16_hip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_long]),
    "rans4x16_hip_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]),
    "rans4x16_hip_option_name": (C.c_char_p, [C.c_int]),
    "rans4x16_hip_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "rans4x16_hip_timing": (None, [C.c_void_p, C.c_int]),
    "rans4x16_hip_timing_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]),
    "rans4x16_hip_version": (C.c_char_p, []),
    "rans4x16_hip_set_dev_stripe_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_uint]),
    "rans4x16_hip_device_clock_khz": (C.c_int, [C.c_void_p]),
    "rans4x16_hip_residency": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_uint, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rans4x16_hip_partition": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rans4x16_hip_cpulist_parse": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int]),
    "rans4x16_hip_multi_numa_node": (C.c_int, [C.c_void_p, C.c_int]),
    "rans4x16_hip_multi_create": (C.c_void_p, [C.c_int, C.c_void_p]),
    "rans4x16_hip_multi_destroy": (None, [C.c_void_p]),
    "rans4x16_hip_multi_devices": (C.c_int, [C.c_void_p]),
    "rans4x16_hip_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "rans4x16_hip_compress_batch_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "rans4x16_hip_uncompress_batch_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p]),
    # include/rans4x8_hip.h
    "rans_compress": (C.c_void_p, [_u8p, C.c_uint, C.POINTER(C.c_uint), C.c_int]),
    "rans_uncompress": (C.c_void_p, [_u8p, C.c_uint, C.POINTER(C.c_uint)]),
    "rans4x8_hip_compress_bound": (C.c_uint, [C.c_uint]),
    "rans4x8_hip_compress_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "rans4x8_hip_uncompress_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "rans4x8_hip_compress_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rans4x8_hip_uncompress_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


class LibraryNotBuilt(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library.  Raises LibraryNotBuilt if it is missing — the product path never
    substitutes another implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryNotBuilt(
                f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C htscodecs_amd/csrc` (needs hipcc)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
