"""CPU: the C-ABI library loads and exports every symbol include/*.h declares; without a GPU the
product fails loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsi_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(hotlib):
    from rsicnv_amd import api
    names = declared_functions("rsi_hot.h") + declared_functions("rsi_synth.h")
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(hotlib, n)]
    assert not missing, f"librsi_hot.so lacks: {missing}"
    assert set(api.EXPORTS) <= set(names)


def test_struct_layouts_match_header(hotlib):
    from rsicnv_amd import api
    assert C.sizeof(api.RsiParams) == 6 * 4 + 7 * 8
    assert C.sizeof(api.RsiCall) == 8 * 4 + 8 * 8
    p = api.RsiParams()
    hotlib.rsi_default_params(C.byref(p))
    assert (p.m, p.gcadjust, p.trans, p.merge, p.maxchkbp, p.cap, p.epsilon, p.chklen, p.minmlen, p.buffer, p.p) == \
        (101, 1, 0, 1, 100000, 4.0, 1.5, 2.5, 3.01, 0.05, 0.05)      # rsi.cpp:34-98
    assert api.make_params(m=100).m == 101                             # rsi.cpp:2061-2064


def test_no_cpu_fallback(hotlib):
    """With no HIP device the context cannot be created and says why; with one it can."""
    from rsicnv_amd import api
    try:
        h = api.RsiHot(0)
    except api.RsiError as e:
        assert e.code == -1 and "no CPU fallback" in str(e)
    else:
        h.close()


def test_cli_binary_exists_and_prints_usage():
    import subprocess
    exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
    assert os.path.exists(exe), "build with make -f rsicnv_amd/csrc/Makefile"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert "rsicnv rsi <options> [-b BAMFILE | -d RDFILE -c RNAME ] -f REFFILE" in r.stderr
