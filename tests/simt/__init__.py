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
        _LIB = lib
    return _LIB


def set_schedule(mode="ascending", seed=0):
    """Order in which the emulation resumes runnable fibers: "ascending" thread ids (default), "descending", or "random" (seeded).
    A race-free kernel is insensitive to it; a missing barrier between a producer and a consumer wave is not."""
    library().simt_set_schedule({"ascending": 0, "descending": 1, "random": 2}[mode], seed)


@contextlib.contextmanager
def installed():
    """with installed(): mq_det_amd.ops.* run on CPU tensors through the emulated kernels."""
    from mq_det_amd import ops
    saved = (ops._LIB, ops._need_gpu, ops._stream)
    ops._LIB, ops._need_gpu, ops._stream = library(), (lambda *ts: None), (lambda: ctypes.c_void_p(0))
    try:
        yield ops
    finally:
        ops._LIB, ops._need_gpu, ops._stream = saved
