"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol that
include/waveletsext_hip.h declares, rejects bad arguments with the reference's error classes
before touching a device, and fails loudly (no CPU fallback) when no HIP device exists."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "waveletsext_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(wx):
    lib = ctypes.CDLL(wx.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), "declared in include/waveletsext_hip.h but not exported: " + s
    assert lib.wx_version() >= 100


def test_argument_errors_mirror_the_reference(wx):
    wt = wx.wavelet(wx.WT.db4)
    x = np.zeros((8, 3))
    with pytest.raises(AssertionError):
        wx.wpdall(x, wt, 4)                                   # dwt_all.jl:266
    with pytest.raises(AssertionError):
        wx.wpdall(np.zeros(8), wt)                            # dwt_all.jl:265 ndims > 1
    with pytest.raises(AssertionError):
        wx.wpd(np.zeros(12), wt)                              # DWT.jl:64
    with pytest.raises(AssertionError):
        wx.wptall(x, wt, np.array([0, 1, 0, 0, 0, 0, 0], dtype=bool))   # invalid tree
    with pytest.raises(AssertionError):
        wx.wptall(x, wt, np.ones(4, dtype=bool))              # wrong tree length
    with pytest.raises(AssertionError):
        wx.iwpdall(np.zeros((8, 5, 2)), wt)                   # k-1 <= maxtransformlevels (Utils.jl:110)
    with pytest.raises(wx.ArgumentError):
        wx.iwpdall(np.zeros((8, 2, 2)), wt, 3)                # not enough levels (Utils.jl:120)
    with pytest.raises(AssertionError):
        wx.iwpd_(np.zeros(4), np.zeros((8, 4)), wt)           # DWT.jl:344 size mismatch
    with pytest.raises(wx.ArgumentError):
        wx.OrthoFilter([1.0, 2.0, 3.0])                       # odd-length filter


def test_no_cpu_fallback_without_gpu(wx):
    if wx.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(wx.WxError) as ei:
        wx.wpdall(np.zeros((8, 2)), wx.wavelet(wx.WT.haar))
    assert ei.value.code == -10
