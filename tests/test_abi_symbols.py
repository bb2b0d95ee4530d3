"""The C-ABI libraries load and export every symbol their headers declare (no compute, no GPU)."""
import ctypes
import os
import re

from conftest import ROOT
from swimm_amd import hip_backend, host


def _declared(header, prefix):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_hip_abi_exports_every_declared_symbol():
    names = _declared(os.path.join(ROOT, "include", "swimm_hip.h"), "swimm_hip_")
    assert sorted(hip_backend.ABI_SYMBOLS) == names
    lib = ctypes.CDLL(hip_backend.LIB_PATH)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.swimm_hip_abi_version() == 1


def test_host_lib_exports_every_declared_symbol():
    names = _declared(os.path.join(ROOT, "swimm_amd", "csrc", "host", "swimm_host.h"), "swimm_")
    lib = ctypes.CDLL(host.LIB_PATH)
    for n in names:
        assert getattr(lib, n) is not None
    assert set(host.HOST_SYMBOLS) <= set(names)
