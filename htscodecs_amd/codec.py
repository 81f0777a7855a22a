"""Python mirror of the reference interface (htscodecs/rANS_static4x16.h:41-50) on top of the
C ABI, plus the batch calls.  Same names and argument meaning as the C functions; errors come
back as ``None`` exactly where the C functions return NULL."""
import ctypes as C

import threading

import numpy as np

from . import lib as _lib


def rans_compress_bound_4x16(size, order):
    return _lib.load().rans_compress_bound_4x16(size, order)


def rans_compress_4x16(data, order):
    """bytes -> compressed bytes, or None (C: NULL).  rANS_static4x16pr.c:1347."""
    L = _lib.load()
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    cap = L.rans_compress_bound_4x16(len(src), order)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_uint(cap)
    r = L.rans_compress_to_4x16(src.ctypes.data if len(src) else out.ctypes.data, len(src),
                                out.ctypes.data, C.byref(n), order)
    return out[:n.value].tobytes() if r else None


def rans_uncompress_4x16(comp, out_size=None):
    """compressed bytes -> bytes, or None.  out_size: capacity of the caller buffer (needed for
    X_NOSZ streams, rANS_static4x16pr.c:1456)."""
    L = _lib.load()
    src = np.frombuffer(bytes(comp), dtype=np.uint8)
    if out_size is None:
        n = C.c_uint(0)
        p = L.rans_uncompress_4x16(src.ctypes.data, len(src), C.byref(n))
        if not p:
            return None
        res = C.string_at(p, n.value)
        C.CDLL(None).free(C.c_void_p(p))
        return res
    out = np.empty(out_size + 1, dtype=np.uint8)
    n = C.c_uint(out_size)
    r = L.rans_uncompress_to_4x16(src.ctypes.data, len(src), out.ctypes.data, C.byref(n))
    return out[:n.value].tobytes() if r else None


# ---- rANS 4x8 (include/rans4x8_hip.h; htscodecs/rANS_static.h:41-44) ---------------------------------------

def rans_compress(data, order):
    """bytes -> rANS 4x8 stream, or None (C: NULL).  rANS_static.c:927."""
    L = _lib.load()
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    n = C.c_uint(0)
    dummy = np.zeros(1, dtype=np.uint8)
    p = L.rans_compress(src.ctypes.data if len(src) else dummy.ctypes.data, len(src), C.byref(n), order)
    if not p:
        return None
    res = C.string_at(p, n.value)
    C.CDLL(None).free(C.c_void_p(p))
    return res


def rans_uncompress(comp):
    """rANS 4x8 stream -> bytes, or None.  rANS_static.c:934."""
    L = _lib.load()
    src = np.frombuffer(bytes(comp) + bytes(16), dtype=np.uint8)
    n = C.c_uint(0)
    p = L.rans_uncompress(src.ctypes.data, len(comp), C.byref(n))
    if not p:
        return None
    res = C.string_at(p, n.value)
    C.CDLL(None).free(C.c_void_p(p))
    return res


def _host_batch8(blocks, decode, orders=None, caps=None):
    ctx = _thread_ctx()
    L = ctx.L
    n = len(blocks)
    srcs = [np.frombuffer(bytes(b), dtype=np.uint8) for b in blocks]
    capv = list(caps) if decode else [L.rans4x8_hip_compress_bound(len(s)) for s in srcs]
    outs = [np.empty(max(c, 1), dtype=np.uint8) for c in capv]
    dummy = np.zeros(1, dtype=np.uint8)
    in_p = (C.c_void_p * n)(*[(s.ctypes.data if len(s) else dummy.ctypes.data) for s in srcs])
    out_p = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    in_sz = (C.c_uint * n)(*[len(s) for s in srcs])
    out_sz = (C.c_uint * n)(*capv)
    status = (C.c_int * n)()
    if decode:
        rc = L.rans4x8_hip_uncompress_batch(ctx.h, n, in_p, in_sz, out_p, out_sz, status)
    else:
        ords = (C.c_int * n)(*orders)
        rc = L.rans4x8_hip_compress_batch(ctx.h, n, in_p, in_sz, out_p, out_sz, ords, status)
    if rc < 0:
        raise RuntimeError("rans4x8 batch call failed: " + ctx.error())
    return [outs[i][:out_sz[i]].tobytes() if status[i] == 0 else None for i in range(n)], list(status)


def compress_batch_4x8(blocks, orders):
    return _host_batch8(blocks, False, orders=orders)


def uncompress_batch_4x8(blocks, caps):
    return _host_batch8(blocks, True, caps=caps)


class _Ctx:
    def __init__(self, device=-1):
        self.L = _lib.load()
        self.h = self.L.rans4x16_hip_create(device)
        if not self.h:
            raise RuntimeError("rans4x16_hip_create failed: no usable HIP device (no CPU path exists)")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.rans4x16_hip_destroy(self.h)
        except Exception:
            pass

    def error(self):
        return self.L.rans4x16_hip_last_error(self.h).decode()

    def set_option(self, name, value):
        if self.L.rans4x16_hip_set_option(self.h, name.encode(), int(value)) != 0:
            raise KeyError(f"unknown option {name!r}")

    def get_option(self, name):
        v = C.c_long(0)
        if self.L.rans4x16_hip_get_option(self.h, name.encode(), C.byref(v)) != 0:
            raise KeyError(f"unknown option {name!r}")
        return v.value


_tls = threading.local()


def _thread_ctx():
    """One context per host thread, kept between calls: it owns the device arenas, the pinned bounce buffers
    and the lane contexts of the host-batch pipeline, which are far too expensive to rebuild per call."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = _tls.ctx = _Ctx()
    return ctx


def set_option(name, value):
    """rans4x16_hip_set_option on the calling thread's context (the one the host-batch helpers below use)."""
    _thread_ctx().set_option(name, value)


def get_option(name):
    return _thread_ctx().get_option(name)


def get_default_option(name):
    v = C.c_long(0)
    if _lib.load().rans4x16_hip_get_option(None, name.encode(), C.byref(v)) != 0:
        raise KeyError(f"unknown option {name!r}")
    return v.value


def set_default_option(name, value):
    """Process-wide default (ctx == NULL): contexts created from now on, and the combiner if it has not started."""
    if _lib.load().rans4x16_hip_set_option(None, name.encode(), int(value)) != 0:
        raise KeyError(f"unknown option {name!r}")


def _host_batch(blocks, decode, orders=None, caps=None):
    ctx = _thread_ctx()
    L = ctx.L
    n = len(blocks)
    srcs = [np.frombuffer(bytes(b), dtype=np.uint8) for b in blocks]
    if decode:
        capv = list(caps)
    else:
        capv = [L.rans_compress_bound_4x16(len(s), o) for s, o in zip(srcs, orders)]
    outs = [np.empty(max(c, 1), dtype=np.uint8) for c in capv]
    dummy = np.zeros(1, dtype=np.uint8)
    in_p = (C.c_void_p * n)(*[(s.ctypes.data if len(s) else dummy.ctypes.data) for s in srcs])
    out_p = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    in_sz = (C.c_uint * n)(*[len(s) for s in srcs])
    out_sz = (C.c_uint * n)(*capv)
    status = (C.c_int * n)()
    if decode:
        rc = L.rans4x16_hip_uncompress_batch(ctx.h, n, in_p, in_sz, out_p, out_sz, status)
    else:
        ords = (C.c_int * n)(*orders)
        rc = L.rans4x16_hip_compress_batch(ctx.h, n, in_p, in_sz, out_p, out_sz, ords, status)
    if rc < 0:
        raise RuntimeError("batch call failed: " + ctx.error())
    res = [outs[i][:out_sz[i]].tobytes() if status[i] == 0 else None for i in range(n)]
    return res, list(status)


class MultiCodec:
    """Host-buffer batches over several GPUs of one node (include/rans4x16_hip.h part 3): the library cuts the
    batch into contiguous ranges, one per device.  devices=None: every visible device; a device may be listed
    twice (two pipelines on one card)."""

    def __init__(self, devices=None):
        self.L = _lib.load()
        if devices is None:
            self.h = self.L.rans4x16_hip_multi_create(0, None)
        else:
            arr = (C.c_int * len(devices))(*devices)
            self.h = self.L.rans4x16_hip_multi_create(len(devices), arr)
        if not self.h:
            raise RuntimeError("rans4x16_hip_multi_create failed: no usable HIP device (no CPU path exists)")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.rans4x16_hip_multi_destroy(self.h)
        except Exception:
            pass

    def devices(self):
        return self.L.rans4x16_hip_multi_devices(self.h)

    def _run(self, blocks, decode, orders=None, caps=None):
        L = self.L
        n = len(blocks)
        srcs = [np.frombuffer(bytes(b), dtype=np.uint8) for b in blocks]
        capv = list(caps) if decode else [L.rans_compress_bound_4x16(len(s), o) for s, o in zip(srcs, orders)]
        outs = [np.empty(max(c, 1), dtype=np.uint8) for c in capv]
        dummy = np.zeros(1, dtype=np.uint8)
        in_p = (C.c_void_p * n)(*[(s.ctypes.data if len(s) else dummy.ctypes.data) for s in srcs])
        out_p = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        in_sz = (C.c_uint * n)(*[len(s) for s in srcs])
        out_sz = (C.c_uint * n)(*capv)
        status = (C.c_int * n)()
        if decode:
            rc = L.rans4x16_hip_uncompress_batch_multi(self.h, n, in_p, in_sz, out_p, out_sz, status)
        else:
            ords = (C.c_int * n)(*orders)
            rc = L.rans4x16_hip_compress_batch_multi(self.h, n, in_p, in_sz, out_p, out_sz, ords, status)
        if rc < 0:
            raise RuntimeError("multi-device batch failed: " + L.rans4x16_hip_multi_last_error(self.h).decode())
        return [outs[i][:out_sz[i]].tobytes() if status[i] == 0 else None for i in range(n)], list(status)

    def compress_batch(self, blocks, orders):
        return self._run(blocks, False, orders=orders)

    def uncompress_batch(self, blocks, caps):
        return self._run(blocks, True, caps=caps)


def compress_batch(blocks, orders):
    """list of bytes, list of int -> (list of bytes|None, list of status)."""
    return _host_batch(blocks, False, orders=orders)


def compress_best_batch(blocks, methods):
    """The reference's caller-side "try several methods, keep the smallest" loop (tokenise_name3.c:1246-1300)
    as one call: list of bytes, list of order values -> (list of bytes|None, chosen order per block, statuses)."""
    ctx = _thread_ctx()
    L = ctx.L
    n, k = len(blocks), len(methods)
    srcs = [np.frombuffer(bytes(b), dtype=np.uint8) for b in blocks]
    capv = [max(L.rans_compress_bound_4x16(len(s), m) for m in methods) for s in srcs]
    outs = [np.empty(max(c, 1), dtype=np.uint8) for c in capv]
    dummy = np.zeros(1, dtype=np.uint8)
    in_p = (C.c_void_p * n)(*[(s.ctypes.data if len(s) else dummy.ctypes.data) for s in srcs])
    out_p = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    in_sz = (C.c_uint * n)(*[len(s) for s in srcs])
    out_sz = (C.c_uint * n)(*capv)
    meth = (C.c_int * k)(*methods)
    chosen = (C.c_int * n)()
    status = (C.c_int * n)()
    rc = L.rans4x16_hip_compress_best_batch(ctx.h, n, in_p, in_sz, out_p, out_sz, k, meth, chosen, status)
    if rc < 0:
        raise RuntimeError("batch call failed: " + ctx.error())
    res = [outs[i][:out_sz[i]].tobytes() if status[i] == 0 else None for i in range(n)]
    return res, list(chosen), list(status)


def uncompress_batch(blocks, caps):
    """list of compressed bytes, list of output capacities -> (list of bytes|None, statuses)."""
    return _host_batch(blocks, True, caps=caps)


class DeviceCodec:
    """Device-resident batches on torch tensors (torch is used for device memory and streams
    only).  All tensors must live on the context's device."""

    def __init__(self, device_index=0):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device_index)
        with torch.cuda.device(self.dev):
            self.ctx = _Ctx(device_index)
        self.L = self.ctx.L

    def timing(self, enable=True):
        self.L.rans4x16_hip_timing(self.ctx.h, 1 if enable else 0)

    def set_option(self, name, value):
        self.ctx.set_option(name, value)

    def get_option(self, name):
        return self.ctx.get_option(name)

    def timing_read(self, which, reset=True):
        ms = C.c_double(0)
        k = C.c_int(0)
        rc = self.L.rans4x16_hip_timing_read(self.ctx.h, which, C.byref(ms), C.byref(k), 1 if reset else 0)
        if rc != 0:
            raise RuntimeError("timing_read failed")
        return ms.value, k.value

    def workspace_bytes(self):
        return self.L.rans4x16_hip_workspace_bytes(self.ctx.h)

    def residency(self, decode, nsym, order, shift=10):
        """(streams per CU, live lanes per wave, CUs) of the chain kernel for this kind of stream."""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        if self.L.rans4x16_hip_residency(self.ctx.h, 1 if decode else 0, nsym, order, shift,
                                         C.byref(a), C.byref(b), C.byref(c)) != 0:
            raise RuntimeError("residency query failed")
        return a.value, b.value, c.value

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def compress(self, d_in, in_off, in_size, d_out, out_off, out_cap, out_size, status, order,
                 max_in_size, d_order=None, total_in_size=0):
        t = self.torch
        assert d_in.dtype == t.uint8 and d_out.dtype == t.uint8
        assert in_off.dtype == t.int64 and out_off.dtype == t.int64
        assert in_size.dtype == t.int32 and out_cap.dtype == t.int32
        assert out_size.dtype == t.int32 and status.dtype == t.int32
        n = in_off.numel()
        rc = self.L.rans4x16_hip_compress_dev_sized(
            self.ctx.h, n, d_in.data_ptr(), in_off.data_ptr(), in_size.data_ptr(),
            d_out.data_ptr(), out_off.data_ptr(), out_cap.data_ptr(), out_size.data_ptr(),
            status.data_ptr(), int(order), d_order.data_ptr() if d_order is not None else None,
            int(max_in_size), int(total_in_size), self._stream())
        if rc != 0:
            raise RuntimeError("compress_dev: " + self.ctx.error())

    def uncompress(self, d_in, in_off, in_size, d_out, out_off, out_cap, out_size, status,
                   max_in_size, max_out_cap, total_out_cap=0):
        t = self.torch
        assert d_in.dtype == t.uint8 and d_out.dtype == t.uint8
        n = in_off.numel()
        rc = self.L.rans4x16_hip_uncompress_dev_sized(
            self.ctx.h, n, d_in.data_ptr(), in_off.data_ptr(), in_size.data_ptr(),
            d_out.data_ptr(), out_off.data_ptr(), out_cap.data_ptr(), out_size.data_ptr(),
            status.data_ptr(), int(max_in_size), int(max_out_cap), int(total_out_cap), self._stream())
        if rc != 0:
            raise RuntimeError("uncompress_dev: " + self.ctx.error())
