"""CPU tests of the drop-in boundary: libsavont_hip.so loads without a GPU, exports exactly the symbols
include/savont_hip.h declares, and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "savont_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svt_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    from savont_amd import hip
    L = hip.load()
    names = _header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libsavont_hip.so does not export %s" % n
    assert sorted(hip.SYMBOLS) == names


def test_only_c_abi_symbols_leak():
    """the exported svt_* surface is extern "C" (unmangled)"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "savont_amd", "libsavont_hip.so")]).decode()
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert set(_header_symbols()) <= set(exported)


def test_no_device_is_a_loud_error():
    from savont_amd import hip
    L = hip.load()
    if L.svt_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert L.svt_create(0, C.byref(h)) == hip.SVT_ERR_NODEVICE and not h.value
    with pytest.raises(hip.SavontHipError):
        hip.Device(0)
    from savont_amd.pipeline import AsvPipeline
    with pytest.raises(hip.SavontHipError):
        AsvPipeline(0)


def test_product_never_references_the_oracle():
    """nothing under savont_amd/ or include/ may import, link or name the oracle"""
    bad = []
    for base in ("savont_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in d or "__pycache__" in d:
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(d, f), errors="ignore").read()
                for m in re.finditer(r"(oracle_lib|savont_oracle|libsavont_oracle|orc_[a-z_]+\()", txt):
                    bad.append((f, m.group(0)))
    assert not bad, bad
    ldd = subprocess.check_output(["ldd", os.path.join(ROOT, "savont_amd", "libsavont_asv.so")]).decode()
    assert "oracle" not in ldd
