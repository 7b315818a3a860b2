"""The C-ABI library loads and exports every symbol include/tilemotion.h declares (no compute calls: runs without a GPU),
and the product path cannot reach the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "tilemotion.h")
LIB = os.path.join(ROOT, "tiler_amd", "lib", "libtilemotion.so")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        subprocess.check_call(["bash", os.path.join(ROOT, "tiler_amd", "csrc", "build.sh")])
    return ctypes.CDLL(LIB)


def _declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    return sorted(set(re.findall(r"TM_API[^;(]*?\b(\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported(built):
    names = _declared()
    assert len(names) > 30
    missing = [n for n in names if not hasattr(built, n)]
    assert not missing, f"declared in tilemotion.h but not exported: {missing}"


def test_only_declared_symbols_are_exported():
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    extra = sorted(n for n in exported if not n.startswith("_") and n not in set(_declared()))
    assert not extra, f"exported but undeclared: {extra}"


def test_no_device_fails_loudly(built):
    """without a GPU the product must refuse, not fall back (this container has none; on the GPU box it is skipped)"""
    built.tm_device_count.restype = ctypes.c_int
    if built.tm_device_count() > 0:
        pytest.skip("a GPU is present")
    built.tm_create.restype = ctypes.c_void_p
    assert not built.tm_create()
    built.tm_last_error.restype = ctypes.c_char_p
    assert b"no CPU path" in built.tm_last_error()
    built.tm_stage_features_rgb.restype = ctypes.c_int
    assert built.tm_stage_features_rgb(None, ctypes.c_int64(1), None, 1, 0, None, None) == -2  # TM_E_NODEVICE


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "tiler_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "libtm_oracle" not in txt and "oracle_binding" not in txt and "tm_oracle.h" not in txt, fn
    out = subprocess.check_output(["ldd", LIB], text=True)
    assert "oracle" not in out
