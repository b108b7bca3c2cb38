"""Helpers for the hipemu tests: build the emulator library and call the C ABI on host (numpy) memory."""
import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "hipemu"))


def emu_lib():
    import build_emu
    from howl_amd.lib import Library
    return Library(build_emu.build())


def ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


class emulated_package:
    """Context manager: routes ``howl_amd`` (ops, models, trainers) through the hipemu build of the SAME kernels on host
    memory -- the C ABI is called with CPU tensor pointers and a null stream.  Test infrastructure: lets the CPU suite run
    the product's host logic end to end (collate, FusedTrainer, data-parallel all-reduce over gloo) without a GPU.  The
    product itself has no such path: ``ops.on_device`` is ``tensor.is_cuda`` there."""

    def __enter__(self):
        import torch
        from howl_amd import lib, ops
        self._saved = (lib._LIB, ops.on_device, ops._stream, torch.cuda.current_stream)
        lib._LIB = emu_lib()
        ops.on_device = lambda t: True
        ops._stream = lambda: None
        return self

    def __exit__(self, *exc):
        from howl_amd import lib, ops
        lib._LIB, ops.on_device, ops._stream, _ = self._saved
        return False
