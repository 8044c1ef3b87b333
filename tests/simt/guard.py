"""TEST INFRASTRUCTURE ONLY.  Guard-page allocations for the kernel-source emulation: a tensor whose storage ends (mode "end") or
begins (mode "start") exactly at a PROT_NONE page, so that a kernel reading or writing even one byte past that side of a buffer dies
with SIGSEGV on the host -- the emulated kernels address host memory directly, an out-of-bounds access that would be a silent read of
a neighbouring allocation (or a memory fault that kills the process) on the GPU becomes a hard, attributable failure here.
`guarded_ops(mode)` additionally routes the output / workspace allocations of mq_det_amd.ops (torch.empty / empty_like / zeros / full)
through the same allocator."""
import contextlib
import ctypes
import mmap

import torch

_libc = ctypes.CDLL(None, use_errno=True)
_libc.mmap.restype = ctypes.c_void_p
_libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
_libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
_PAGE = mmap.PAGESIZE
_KEEP = []                       # mappings stay alive for the life of the test process (small, bounded by the tests that use them)


def alloc(shape, dtype, mode="end"):
    """An uninitialised CPU tensor of `shape` / `dtype` with a PROT_NONE page directly after ("end") or before ("start") its data."""
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
    n = 1
    for s in shape:
        n *= s
    nbytes = max(n * torch.empty(0, dtype=dtype).element_size(), 1)
    body = -(-nbytes // _PAGE) * _PAGE
    base = _libc.mmap(None, body + 2 * _PAGE, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    if base in (None, ctypes.c_void_p(-1).value):
        raise MemoryError("mmap failed")
    if _libc.mprotect(base, _PAGE, 0) or _libc.mprotect(base + _PAGE + body, _PAGE, 0):
        raise OSError(ctypes.get_errno(), "mprotect failed")
    es = torch.empty(0, dtype=dtype).element_size()
    # "end": the last byte of the tensor is the last byte before the guard page (tensors whose byte size is not a multiple of 16 are
    # then not 16-byte aligned at their start -- exactly the situation of a slice in the middle of a device allocation)
    addr = base + _PAGE + (body - nbytes if mode == "end" else 0)
    buf = (ctypes.c_char * nbytes).from_address(addr)
    _KEEP.append(buf)
    t = torch.frombuffer(buf, dtype=torch.uint8).view(dtype)[:n].reshape(shape) if n else torch.empty(shape, dtype=dtype)
    return t


def guarded(t, mode="end"):
    """A copy of `t` (same shape, dtype, values; contiguous) in guard-page storage."""
    if t is None:
        return None
    g = alloc(t.shape, t.dtype, mode)
    g.copy_(t)
    return g


@contextlib.contextmanager
def guarded_ops(mode="end"):
    """Inside: every tensor mq_det_amd.ops allocates (outputs, workspaces) sits against a guard page."""
    from mq_det_amd import ops

    class _Torch:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def empty(*size, dtype=torch.float32, device=None, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            return alloc(size, dtype, mode)

        @staticmethod
        def empty_like(t, **kw):
            return alloc(t.shape, kw.get("dtype", t.dtype), mode)

        @staticmethod
        def zeros(*size, dtype=torch.float32, device=None, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            return alloc(size, dtype, mode).zero_()

    saved = ops.torch
    ops.torch = _Torch()
    try:
        yield
    finally:
        ops.torch = saved


@contextlib.contextmanager
def pointer_guard(mode="end"):
    """Inside: every tensor handed to the library through mq_det_amd.ops._ptr is replaced, for the duration of the call, by a copy of
    its WHOLE storage placed against a guard page (views keep their offset inside it), and copied back afterwards (outputs).  The
    parity checks run unchanged; a kernel that touches a byte outside the storage of any of its arguments on the guarded side
    kills the process with SIGSEGV (run it in a subprocess).  Pointers inside ctypes structs (the grouped DCNv2 / coefficient launches)
    bypass _ptr and are not guarded."""
    from mq_det_amd import ops
    live = {}                                             # storage data_ptr -> (uint8 view of the original storage, guarded uint8 copy)

    def _ptr(t):
        if t is None:
            return ctypes.c_void_p(0)
        st = t.untyped_storage()
        key = st.data_ptr()
        if key not in live:
            orig = torch.empty(0, dtype=torch.uint8).set_(st)
            g = alloc((orig.numel(),), torch.uint8, mode)
            g.copy_(orig)
            live[key] = (orig, g)
        return ctypes.c_void_p(live[key][1].data_ptr() + (t.data_ptr() - key))

    real_chk = ops._chk

    def _chk(rc, name):
        for orig, g in live.values():
            orig.copy_(g)
        live.clear()
        return real_chk(rc, name)

    saved = (ops._ptr, ops._chk)
    ops._ptr, ops._chk = _ptr, _chk
    try:
        yield
    finally:
        ops._ptr, ops._chk = saved
