"""TEST INFRASTRUCTURE ONLY.  `install()` points mq_det_amd.ops at the lane-by-lane host emulation of the kernel SOURCES
(tests/simt/build_emu.py) so that the thin torch wrappers, the C ABI marshalling and the kernels themselves can be checked on a
machine without a GPU, on CPU tensors.  Nothing under mq_det_amd/ imports this package; the product raises without a GPU."""
import contextlib
import ctypes

from . import build_emu

_LIB = None


def library():
    global _LIB
    if _LIB is None:
        from mq_det_amd import ops
        lib = ctypes.CDLL(build_emu.build())
        for name, (res, args) in ops._SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib.simt_set_schedule.restype, lib.simt_set_schedule.argtypes = None, [ctypes.c_int, ctypes.c_ulong]
        lib.simt_set_lds_limit.restype, lib.simt_set_lds_limit.argtypes = None, [ctypes.c_ulong]
        for name in ops.BF16_TWINS:                       # the fp32-operand build of the same sources (build_emu.py: *_f32)
            fn = getattr(lib, name + "_f32", None)
            if fn is not None:
                fn.restype, fn.argtypes = ops._SIGNATURES[name]
        _LIB = lib
    return _LIB


def set_schedule(mode="ascending", seed=0):
    """Order in which the emulation resumes runnable fibers: "ascending" thread ids (default), "descending", or "random" (seeded).
    A race-free kernel is insensitive to it; a missing barrier between a producer and a consumer wave is not."""
    library().simt_set_schedule({"ascending": 0, "descending": 1, "random": 2}[mode], seed)


@contextlib.contextmanager
def installed(f32=False):
    """with installed(): mq_det_amd.ops.* run on CPU tensors through the emulated kernels.
    f32=True: through the fp32-OPERAND build of the kernel sources (entry points *_f32: every 16-bit operand is a float, the emulated
    MFMA multiplies exactly) -- the wrappers then take float32 tensors wherever they take fp16 / bf16 on the device.  What is left
    between such a run and the fp32 oracle is the kernels' logic and summation order, not operand rounding."""
    import torch
    from mq_det_amd import ops
    saved = (ops._LIB, ops._need_gpu, ops._stream, ops._fn, ops._H16)
    lib = library()
    ops._LIB, ops._need_gpu, ops._stream = lib, (lambda *ts: None), (lambda: ctypes.c_void_p(0))
    if f32:
        ops._fn = lambda lib_, name, *ts: getattr(lib_, name + "_f32")
        ops._H16 = (torch.float32,)
        lib.simt_set_lds_limit(320 << 10)                 # every 16-bit LDS tile is twice as large
    try:
        yield ops
    finally:
        ops._LIB, ops._need_gpu, ops._stream, ops._fn, ops._H16 = saved
        lib.simt_set_lds_limit(160 << 10)
