"""TEST INFRASTRUCTURE ONLY: run the product's HIP kernels under the CPU emulator.

`with emulated():` swaps the ctypes handle inside score_sde_pytorch_amd._lib for
tests/emu/build/libssde_emu.so (same sources, same C ABI, host pointers) and relaxes the
is_cuda guards of the thin Python wrappers for the duration of the block.  The product package
has no such switch: outside this context manager a CPU tensor is refused, as always.
"""
import contextlib
import ctypes as C

import torch

from . import build_emu

available = build_emu.available

_handle = None


def lib():
    global _handle
    if _handle is None:
        from score_sde_pytorch_amd import _lib as L
        _handle = L.bind(C.CDLL(build_emu.build()))
    return _handle


@contextlib.contextmanager
def emulated():
    from score_sde_pytorch_amd import _lib as L, hipops, engine
    saved = (L._lib, hipops._need_cuda, hipops._stream, engine.Program._launch)
    L._lib = lib()
    hipops._need_cuda = lambda *ts: None
    hipops._stream = lambda: C.c_void_p(0)

    def _launch(self, ops_ptr, count, stream=None):
        L.check(L._lib.ssde_program_run(ops_ptr, count, C.c_void_p(0)), "ssde_program_run[emu]")
    engine.Program._launch = _launch
    try:
        yield L._lib
    finally:
        L._lib, hipops._need_cuda, hipops._stream, engine.Program._launch = saved
