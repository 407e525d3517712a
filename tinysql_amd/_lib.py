"""ctypes binding of libtsq.so — the reference-side analogue is the cgo stub in INTEGRATION.md.

Fails loudly: a missing library or a missing GPU raises; nothing here computes on the CPU.
"""
import ctypes as C
import os

from . import _abi as abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtsq.so")


class TsqError(RuntimeError):
    """Error returned by libtsq (the Go shim maps these to `error` values)."""

    def __init__(self, status, message):
        super().__init__("tsq status %d (%s): %s" % (status, abi.STATUS_NAMES.get(status, "?"), message))
        self.status = status
        self.message = message


_lib = None


def load():
    """Loads libtsq.so once; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libtsq.so not found at %s: build it with `make -C tinysql_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in abi.SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.tsq_abi_version() != abi.TSQ_ABI_VERSION:
        raise RuntimeError("libtsq.so ABI version mismatch")
    _lib = lib
    return lib


def last_error(handle=None):
    msg = load().tsq_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(status, handle=None):
    if status != abi.OK:
        raise TsqError(status, last_error(handle))


class Context:
    """tsq_ctx: one per (process, device)."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.tsq_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device

    def set_stream(self, stream_ptr):
        check(self.lib.tsq_ctx_set_stream(self.h, C.c_void_p(stream_ptr)), self.h)

    def sync(self):
        check(self.lib.tsq_ctx_sync(self.h), self.h)

    def reserve(self, nbytes):
        """tsq_ctx_reserve: one slab of HBM that every operator buffer of this context is carved from (0 gives it back)"""
        check(self.lib.tsq_ctx_reserve(self.h, nbytes), self.h)

    def set_knob(self, knob, value=abi.KNOB_DEFAULT):
        """tsq_ctx_set_knob: a test / measurement knob (abi.KNOB_*); no value = back to the default"""
        check(self.lib.tsq_ctx_set_knob(self.h, knob, value), self.h)

    def reset_knobs(self):
        for k in range(48):
            self.set_knob(k)

    def knobs(self, **kv):
        """context manager: `with ctx.knobs(AGG_TAG_BITS=7): ...` sets the knobs and restores the defaults afterwards"""
        ctx = self

        class _K:
            def __enter__(self_inner):
                for k, v in kv.items():
                    ctx.set_knob(getattr(abi, "KNOB_" + k), int(v))

            def __exit__(self_inner, *a):
                for k in kv:
                    ctx.set_knob(getattr(abi, "KNOB_" + k))
        return _K()

    def arena_stats(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.tsq_ctx_arena_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), self.h)
        return {"size": a.value, "used": b.value, "peak": c.value}

    def alloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.tsq_dev_alloc(self.h, nbytes, C.byref(p)), self.h)
        return p.value

    def free(self, ptr):
        if ptr:
            check(self.lib.tsq_dev_free(self.h, C.c_void_p(ptr)), self.h)

    def host_array(self, n, dtype):
        """numpy array of n elements over PINNED host memory (tsq_host_alloc): what a host keeps the chunks in that it pushes or
        pulls — the DMA engines reach it directly.  Give it back with host_release(arr) (or it goes with the context)."""
        import numpy as np
        dt = np.dtype(dtype)
        p = C.c_void_p()
        check(self.lib.tsq_host_alloc(self.h, max(int(n), 1) * dt.itemsize, C.byref(p)), self.h)
        buf = (C.c_uint8 * (max(int(n), 1) * dt.itemsize)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt, count=int(n))
        self._host_blocks = getattr(self, "_host_blocks", {})
        self._host_blocks[arr.ctypes.data] = p.value
        return arr

    def host_release(self, arr):
        p = getattr(self, "_host_blocks", {}).pop(arr.ctypes.data, None)
        if p:
            check(self.lib.tsq_host_free(self.h, C.c_void_p(p)), self.h)

    def memset(self, ptr, byte, nbytes):
        check(self.lib.tsq_dev_memset(self.h, C.c_void_p(ptr), byte, nbytes), self.h)

    def h2d(self, dst, src_np):
        check(self.lib.tsq_copy_h2d(self.h, C.c_void_p(dst), src_np.ctypes.data_as(C.c_void_p), src_np.nbytes), self.h)

    def d2h(self, dst_np, src):
        check(self.lib.tsq_copy_d2h(self.h, dst_np.ctypes.data_as(C.c_void_p), C.c_void_p(src), dst_np.nbytes), self.h)

    def timer_start(self):
        check(self.lib.tsq_timer_start(self.h), self.h)

    def timer_stop_ms(self):
        ms = C.c_double()
        check(self.lib.tsq_timer_stop_ms(self.h, C.byref(ms)), self.h)
        return ms.value

    def gen_column(self, spec, nrows, dst, null_bitmap=None, src=None):
        check(self.lib.tsq_gen_column(self.h, C.byref(spec), nrows, C.c_void_p(dst),
                                      C.c_void_p(null_bitmap) if null_bitmap else None,
                                      C.c_void_p(src) if src else None), self.h)

    def close(self):
        if self.h:
            self.lib.tsq_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
