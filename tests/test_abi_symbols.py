"""The C-ABI libraries load and export every symbol their headers declare (no compute, no GPU)."""
import ctypes
import os
import re
import subprocess

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


def test_header_is_plain_c_and_matches_the_reference_call(tmp_path):
    """gcc -std=c99 -pedantic -Werror on the binding a SWIMM maintainer would write"""
    src = os.path.join(ROOT, "tests", "data", "dropin_binding.c")
    for std in ("c99", "c11"):
        subprocess.check_call(["gcc", "-std=" + std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                               "-c", src, "-o", str(tmp_path / ("b_" + std + ".o"))])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-I", os.path.join(ROOT, "include"), "-c", src,
                           "-o", str(tmp_path / "b_cxx.o")])
