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
