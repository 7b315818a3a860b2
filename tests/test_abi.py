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


def test_optimize_palettes_host_matches_oracle(built, oracle):
    """A11 runs on the host in both builds: the product's C++ Powell/OptimizePalettes against the oracle's C restatement"""
    import numpy as np
    rng = np.random.default_rng(4)
    for pc, ps in ((1, 16), (6, 16), (3, 5), (9, 64)):
        pals = rng.integers(0, 1 << 24, size=(pc, ps)).astype(np.int32)
        pals[0, ps - 1] = -65281  # a null slot takes part as magenta (FromRGB of $FFFF00FF)
        a, b = pals.copy(), pals.copy()
        sw = ctypes.c_int()
        assert built.tm_optimize_palettes_host(a.ctypes.data_as(ctypes.c_void_p), pc, ps, ctypes.byref(sw)) == 0
        sweeps = oracle.L.tmo_optimize_palettes(b.ctypes.data_as(ctypes.c_void_p), pc, ps)
        assert np.array_equal(a, b) and sw.value == sweeps
        assert all(sorted(x) == sorted(y) for x, y in zip(a.tolist(), pals.tolist()))  # a permutation of each palette
