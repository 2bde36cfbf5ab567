"""
ctypes binding of libacoss_mi355x.so (include/acoss_mi355x.h).  There is no CPU fallback: if the
library is missing or was not built, every entry point of the package fails loudly here.
"""
import ctypes
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# (ACOSS_LIB_PATH: another build of the library -- tools/build_variant.py -- for A/B runs of the tests and tools)
LIB_PATH = os.environ.get("ACOSS_LIB_PATH") or os.path.join(PKG, "libacoss_mi355x.so")

PAIR_DESC = np.dtype([
    ("x_row0", "<i8"), ("y_row0", "<i8"), ("csm_off", "<i8"), ("crp_off", "<i8"),
    ("nx", "<i4"), ("ny", "<i4"), ("csm_pitch", "<i4"), ("crp_pitch", "<i4"),
    ("shift", "<i4"), ("song_x", "<i4"), ("song_y", "<i4"), ("reserved", "<i4"),
])
MAT_DESC = np.dtype([
    ("s_off", "<i8"), ("d_off", "<i8"), ("rows", "<i4"), ("cols", "<i4"),
    ("s_pitch", "<i4"), ("d_pitch", "<i4"),
])
assert PAIR_DESC.itemsize == 64 and MAT_DESC.itemsize == 32


class AlignParams(ctypes.Structure):
    _fields_ = [("gamma_onset", ctypes.c_float), ("gamma_extension", ctypes.c_float),
                ("sw_match", ctypes.c_float), ("sw_mismatch", ctypes.c_float),
                ("sw_gap_open", ctypes.c_float), ("sw_gap_ext", ctypes.c_float)]


class AcossError(RuntimeError):
    pass


_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_dbl = ctypes.c_double
_sz = ctypes.c_size_t

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "acoss_abi_version": (_i, []),
    "acoss_last_error": (ctypes.c_char_p, []),
    "acoss_device_count": (_i, []),
    "acoss_set_device": (_i, [_i]),
    "acoss_default_align_params": (None, [ctypes.POINTER(AlignParams)]),
    "qmax_c": (ctypes.c_float, [_vp, _vp, _i, _i]),
    "dmax_c": (ctypes.c_float, [_vp, _vp, _i, _i]),
    "swalignimpconstrained": (ctypes.c_float, [_vp, _vp, _i, _i]),
    "acoss_plan_pairs": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "acoss_frame_norms_f64": (_i, [_vp, _i64, _i, _vp, _vp]),
    "acoss_frame_norms_f32": (_i, [_vp, _i64, _i, _vp, _vp]),
    "acoss_oti_batch": (_i, [_vp, _i, _vp, _i, _vp]),
    "acoss_csm_batch_f64": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "acoss_csm_batch_f32": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "acoss_xpack_elems": (_i64, [_i, _i]),
    "acoss_pack_x_f64": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "acoss_pack_x_f32": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "acoss_csm_packed_batch_f64": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "acoss_csm_packed_batch_f32": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "acoss_csm_strip_batch_f64": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "acoss_csm_rows_batch_f64": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "acoss_crp_batch_f64": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "acoss_crp_planar_batch_f64": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "acoss_crp_planar32_batch": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "acoss_crp_batch_f32": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "acoss_sliding_batch_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "acoss_sliding_batch_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "acoss_binarize_work_bytes": (_sz, [_i, _i, _i, _i]),
    "acoss_binarize_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _vp, _sz, _vp]),
    "acoss_align_bits_qd_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "acoss_mask_bits_words": (_i, [_i, _i, _i]),
    "acoss_mask_bits_work_bytes": (_sz, [_i, _i, _i, _i]),
    "acoss_mask_bits_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _vp, _sz, _vp]),
    "acoss_mask_bits_planar_batch": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _vp, _sz, _vp]),
    "acoss_mask_bits_planar32_batch": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _vp, _sz, _vp]),
    "acoss_crp_keys16_batch": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "acoss_mask_bits_keys16_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _vp, _sz, _vp]),
    "acoss_keys16_koff_f32_batch": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "acoss_mask_bits_keys16_f32_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _vp, _sz, _vp]),
    "acoss_mask_bits_keys16_stats": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "acoss_mask_bits_keys16_unresolved": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "acoss_radix16_enabled": (_i, []),
    "acoss_radix16_work_bytes": (_sz, [_i, _i, _i, _i]),
    "acoss_radix16_layout": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "acoss_radix16_stage": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _vp, _sz, _vp]),
    "acoss_pack_frames_f32": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "acoss_mask_bits_fused_supported": (_i, [_i, _i, _i, _i]),
    "acoss_mask_bits_fused_work_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "acoss_mask_bits_fused_batch": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _vp, _sz, _i, _vp]),
    "acoss_mask_bits_fused_counter": (_vp, [_vp]),
    "acoss_align_bits_batch": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "acoss_thresholds_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _dbl, _i, _vp, _sz, _vp]),
    "acoss_align_fused_batch": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp, _vp, _vp]),
    "acoss_qmax_batch": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "acoss_dmax_batch": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "acoss_ftm2d_scratch_bytes": (_sz, [_i64, _i64, _i]),
    "acoss_ftm2d_shingles": (_i, [_vp, _vp, _i, _dbl, _dbl, _vp, _sz, _vp, _vp]),
    "acoss_ftm2d_pairs": (_i, [_vp, _vp, _i, _vp, _vp]),
    "acoss_ftm2d_gram": (_i, [_vp, _i, _vp, _vp]),
    "acoss_snf_scratch_bytes": (_sz, [_vp, _vp, _i, _i]),
    "acoss_snf_cross_batch": (_i, [_vp, _i, _i, _vp, _vp, _dbl, _dbl, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "acoss_eval_ranks": (_i, [_vp, _i, _i64, _vp, _vp, _i, _vp, _vp]),
    "acoss_corpus_create": (_i, [_vp, _vp, _i, _i, _vp, _i, ctypes.POINTER(_vp)]),
    "acoss_corpus_wrap": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, ctypes.POINTER(_vp)]),
    "acoss_corpus_destroy": (None, [_vp]),
    "acoss_serra09_scratch_bytes": (_sz, [_vp, _vp, _i, _i, _i]),
    "acoss_serra09_scores": (_i, [_vp, _vp, _i, _i, _dbl, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "acoss_swc_batch": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
}

_LIB = None


def load():
    """The loaded library.  Raises AcossError when it is absent -- build it with
    `python -m acoss_amd.build` (hipcc, gfx950)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise AcossError(
                "libacoss_mi355x.so is not built (%s).  Run `python -m acoss_amd.build`; this "
                "package has no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm ships its own HIP runtime; it must be in the process first so that the
        # library binds to the same runtime instance whose device pointers and streams it is handed.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.acoss_abi_version() != 1:
            raise AcossError("libacoss_mi355x.so ABI version mismatch")
        _LIB = lib
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = load().acoss_last_error().decode("utf-8", "replace")
        raise AcossError("%s failed (%d): %s" % (what or "acoss call", rc, msg))


def last_error():
    return load().acoss_last_error().decode("utf-8", "replace")
