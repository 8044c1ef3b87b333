"""TEST INFRASTRUCTURE ONLY.  `install()` points mq_det_amd.ops at the lane-by-lane host emulation of the kernel SOURCES
(tests/simt/build_emu.py) so that the thin torch wrappers, the C ABI marshalling and the kernels themselves can be checked on a
machine without a GPU, on CPU tensors.  Nothing under mq_det_amd/ imports this package; the product raises without a GPU."""
import contextlib
import ctypes
import os

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
    f32 = 1 / True: the PRECISE mode exactly as the device runs it (KERNELS["F32_OPERANDS"] = 1: entry points *_f32 -- every operand a
    float, one 16x16x32 MFMA = eight exact 16x16x4 fp32 MFMAs --, 160 KB of LDS per workgroup, the wrappers' choice of variants that fit);
    f32 = 2: the same operand type with a 320 KB LDS limit, i.e. EVERY kernel in the shape the 16-bit modes launch it (kernel logic
    against the fp32 oracle).  The wrappers then take float32 tensors wherever they take fp16 / bf16 on the device."""
    from mq_det_amd import ops
    saved = (ops._LIB, ops._need_gpu, ops._stream, os.environ.get("MQ_F32_OPERANDS"))
    lib = library()
    ops._LIB, ops._need_gpu, ops._stream = lib, (lambda *ts: None), (lambda: ctypes.c_void_p(0))
    if f32:
        os.environ["MQ_F32_OPERANDS"] = str(int(f32))
        lib.simt_set_lds_limit((160 if int(f32) == 1 else 320) << 10)
    ops.configure()
    try:
        yield ops
    finally:
        ops._LIB, ops._need_gpu, ops._stream = saved[:3]
        if saved[3] is None:
            os.environ.pop("MQ_F32_OPERANDS", None)
        else:
            os.environ["MQ_F32_OPERANDS"] = saved[3]
        ops.configure()
        lib.simt_set_lds_limit(160 << 10)
