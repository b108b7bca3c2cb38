"""CPU checks of the drop-in boundary: the built library loads and exports every entry point that
``include/howl_hip.h`` declares, and the ctypes table in ``howl_amd/lib.py`` covers exactly those (no compute calls)."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    return ge.LIB


def header_functions():
    text = (ROOT / "include" / "howl_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(howl_[a-z0-9_]+)\s*\(", text))


def test_library_exports_header(built):
    from howl_amd import lib
    hdr = header_functions()
    out = subprocess.run(["nm", "-D", "--defined-only", str(built)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (howl_[a-z0-9_]+)\n", out))
    assert hdr <= exported, hdr - exported
    table = set(lib.SIGNATURES) | set(lib.SIZE_FUNCS) | {"howl_last_error"}
    assert table == hdr, (table ^ hdr)
    lb = lib.Library(built)   # resolves every symbol and sets argtypes
    import ctypes
    major, minor = ctypes.c_int(-1), ctypes.c_int(-1)
    lb.call("howl_version", ctypes.byref(major), ctypes.byref(minor))
    assert (major.value, minor.value) == (0, 1)
    assert lb.cdll.howl_res8_workspace_bytes(4, 81) > 0


def test_missing_library_is_loud(tmp_path):
    from howl_amd import lib
    with pytest.raises(lib.HowlHipError):
        lib.Library(tmp_path / "nope.so")


def test_argument_errors_have_messages(built):
    from howl_amd import lib
    lb = lib.Library(built)
    with pytest.raises(lib.HowlHipError) as e:
        lb.call("howl_logmel_fwd", None, 1, 16000, 16000, None, 40, 1e-7, None, None, 0, None)
    assert "null pointer" in str(e.value)
