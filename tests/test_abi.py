"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU and
exports exactly the entry points include/acoss_mi355x.h declares.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "acoss_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_library_loads_and_exports_every_declared_symbol():
    from acoss_amd import _lib, build
    build.build()
    lib = _lib.load()
    names = declared_functions()
    assert {"qmax_c", "dmax_c", "swalignimpconstrained"} <= set(names)   # pySeqAlign.pxd:3-10
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "declared in the header but not exported: " + n
    # and the Python binding table covers the same set
    assert sorted(_lib.SIGNATURES) == names
    assert lib.acoss_abi_version() == 1


def test_struct_layouts_match_header():
    from acoss_amd import _lib
    assert _lib.PAIR_DESC.itemsize == 64 and _lib.MAT_DESC.itemsize == 32
    assert ctypes.sizeof(_lib.AlignParams) == 24
    p = _lib.AlignParams()
    _lib.load().acoss_default_align_params(ctypes.byref(p))
    # SequenceAlignment.c:45-46, 57-58, 105-106
    assert (p.gamma_onset, p.gamma_extension, p.sw_match, p.sw_mismatch) == (0.5, 0.5, 1.0, -1.0)
    assert p.sw_gap_open == -0.5 and abs(p.sw_gap_ext - (-0.7)) < 1e-7


def test_plan_pairs_host_helper():
    from acoss_amd import _lib
    lib = _lib.load()
    off = np.array([0, 100, 250, 259], dtype=np.int64)
    pairs = np.array([[0, 1], [1, 0], [2, 2]], dtype=np.int32)
    descs = np.zeros(3, dtype=_lib.PAIR_DESC)
    tc, tr = ctypes.c_int64(), ctypes.c_int64()
    rc = lib.acoss_plan_pairs(off.ctypes.data, 3, pairs.ctypes.data, 3, 9, 16, descs.ctypes.data,
                              ctypes.byref(tc), ctypes.byref(tr))
    assert rc == 0
    assert list(descs["nx"]) == [100, 150, 9] and list(descs["ny"]) == [150, 100, 9]
    assert list(descs["x_row0"]) == [0, 100, 250]
    assert list(descs["crp_pitch"]) == [144, 96, 16] and list(descs["csm_pitch"]) == [160, 112, 16]
    assert all(o % 16 == 0 for o in descs["crp_off"]) and all(o % 16 == 0 for o in descs["csm_off"])
    assert tr.value >= 92 * 144 + 142 * 96 + 16
    # a song shorter than the window is an error, not a silent empty matrix
    rc = lib.acoss_plan_pairs(off.ctypes.data, 3, pairs.ctypes.data, 3, 10, 16, descs.ctypes.data,
                              ctypes.byref(tc), ctypes.byref(tr))
    assert rc == -22 and b"shorter" in lib.acoss_last_error()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from acoss_amd import CRPUtils, _lib
    with pytest.raises(_lib.AcossError):
        CRPUtils.get_csm(np.zeros((4, 12)), np.zeros((5, 12)))


def test_pyseqalign_argument_checks():
    from acoss_amd import pySeqAlign
    S = np.zeros(12, dtype=np.uint8)
    D = np.zeros(12, dtype=np.float32)
    with pytest.raises(ValueError):
        pySeqAlign.qmax(S.astype(np.int32), D, 3, 4)           # dtype mismatch
    with pytest.raises(ValueError):
        pySeqAlign.qmax(S.reshape(3, 4), D, 3, 4)              # ndim
    with pytest.raises(ValueError):
        pySeqAlign.dmax(S, np.zeros(24, np.float32)[::2], 3, 4)  # not contiguous
    with pytest.raises(TypeError):
        pySeqAlign.swconstrained(None, D, 3, 4)
    # degenerate sizes return 0.0 without touching the GPU (SequenceAlignment.c:117-119)
    assert pySeqAlign.qmax(np.zeros(4, np.uint8), np.zeros(4, np.float32), 2, 2) == 0.0
    assert pySeqAlign.dmax(np.zeros(9, np.uint8), np.zeros(9, np.float32), 3, 3) == 0.0
    assert pySeqAlign.swconstrained(np.zeros(4, np.uint8), np.zeros(9, np.float32), 2, 2) == 0.0
