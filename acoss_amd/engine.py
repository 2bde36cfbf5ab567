"""
Device-side plumbing between the Python plugin layer and the C ABI (include/acoss_mi355x.h).

PyTorch is used only for what it is good at here: device memory, streams and (in sharding.py)
torch.distributed.  All arithmetic happens in libacoss_mi355x.so; tensors cross the boundary as
raw device pointers.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import MAT_DESC, PAIR_DESC, AcossError, check


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise AcossError("no MI355X visible to PyTorch-ROCm: acoss_amd has no CPU fallback")


def to_device_bytes(arr, device):
    """numpy structured/plain array -> uint8 device tensor holding the same bytes."""
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    return torch.from_numpy(raw.copy()).to(device)


class DeviceCorpus(object):
    """
    A set of songs resident in HBM: frames-major concatenated features (total_frames, d), their
    per-frame squared norms, and (for chroma) the per-song global chroma used by the OTI.
    Mirrors the per-song dict of Serra09.load_features (Serra09.py:154) for one feature type.
    """

    def __init__(self, feats, frame_off, gchroma=None, device=None):
        require_gpu()
        lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        feats = np.ascontiguousarray(feats)
        if feats.dtype not in (np.float32, np.float64):
            feats = feats.astype(np.float64)
        self.dtype = feats.dtype
        self.d = int(feats.shape[1])
        self.frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
        self.n_songs = len(self.frame_off) - 1
        self.n_frames = int(self.frame_off[-1])
        assert feats.shape[0] == self.n_frames
        with torch.cuda.device(self.device):
            self.feats = torch.from_numpy(feats).to(self.device)
            self.norms = torch.empty(self.n_frames, dtype=self.feats.dtype, device=self.device)
            fn = lib.acoss_frame_norms_f64 if self.dtype == np.float64 else lib.acoss_frame_norms_f32
            check(fn(_ptr(self.feats), self.n_frames, self.d, _ptr(self.norms), _stream()), "frame_norms")
            self.gchroma = None
            if gchroma is not None:
                self.gchroma = torch.from_numpy(np.ascontiguousarray(gchroma, dtype=np.float64)).to(self.device)

    def lengths(self):
        return np.diff(self.frame_off)

    def __del__(self):
        for h in getattr(self, "_chandles", []):
            try:
                _lib.load().acoss_corpus_destroy(h)
            except Exception:
                pass

    def song_wmax(self, win):
        """Per song, the largest sum of squared (centred: float32_copy) frame norms over `win` consecutive frames (host
        float64): the scale of the error bound of the float32-approximate windowed sums (crp_planar32)."""
        cache = getattr(self, "_wmax", None)
        if cache is None or cache[0] != win:
            float32_copy(self)
            nrm = self._f32_norms64.astype(np.float64)
            out = np.zeros(self.n_songs)
            for s_ in range(self.n_songs):
                a, b = int(self.frame_off[s_]), int(self.frame_off[s_ + 1])
                if b - a >= win:
                    c = np.concatenate([[0.0], np.cumsum(nrm[a:b])])
                    out[s_] = float(np.max(c[win:] - c[:-win])) * (1.0 + 1e-12)
                else:
                    out[s_] = float(nrm[a:b].sum())
            self._wmax = (win, out)
        return self._wmax[1]


class PairBatch(object):
    """The plan of one launch batch: K pair descriptors on host and device."""

    def __init__(self, frame_off, pairs, win, device, pitch_align=16):
        lib = _lib.load()
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
        self.K = int(pairs.shape[0])
        self.win = int(win)
        self.descs = np.zeros(self.K, dtype=PAIR_DESC)
        tc, tr = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib.acoss_plan_pairs(frame_off.ctypes.data, len(frame_off) - 1, pairs.ctypes.data, self.K,
                                   self.win, int(pitch_align), self.descs.ctypes.data,
                                   ctypes.byref(tc), ctypes.byref(tr)), "plan_pairs")
        self.total_csm, self.total_crp = int(tc.value), int(tr.value)
        self.max_nx = int(self.descs["nx"].max()) if self.K else 0
        self.max_ny = int(self.descs["ny"].max()) if self.K else 0
        self.device = device
        self.descs_dev = to_device_bytes(self.descs, device)

    @property
    def M(self):
        return self.descs["nx"] - self.win + 1

    @property
    def N(self):
        return self.descs["ny"] - self.win + 1

    def set_shifts(self, shifts):
        self.descs["shift"] = np.asarray(shifts, dtype=np.int32)
        self.descs_dev = to_device_bytes(self.descs, self.device)

    def fetch_shifts(self):
        host = self.descs_dev.cpu().numpy().view(PAIR_DESC)
        self.descs["shift"] = host["shift"]
        return self.descs["shift"].copy()

    def mats(self, with_d=False, sw=False):
        """acoss_mat_desc array for the alignment kernels over this batch's CRP matrices.
        Returns (mats numpy, total D elements)."""
        m = np.zeros(self.K, dtype=MAT_DESC)
        m["s_off"] = self.descs["crp_off"]
        m["rows"] = self.M
        m["cols"] = self.N
        m["s_pitch"] = self.descs["crp_pitch"]
        total = 0
        if with_d:
            extra = 1 if sw else 0
            pitch = m["cols"].astype(np.int64) + extra
            sizes = (m["rows"].astype(np.int64) + extra) * pitch
            offs = np.zeros(self.K, dtype=np.int64)
            if self.K:
                offs[1:] = np.cumsum(sizes)[:-1]
            m["d_off"] = offs
            m["d_pitch"] = pitch
            total = int(sizes.sum())
        return m, total


# ---------------------------------------------------------------------------------------------
# stage wrappers (device tensors in, device tensors out)
# ---------------------------------------------------------------------------------------------
def oti(corpus, batch):
    """CRPUtils.py:109 for every pair of the batch; fills the descriptors' shift on the device."""
    if corpus.gchroma is None:
        raise AcossError("oti: the corpus has no global chroma")
    check(_lib.load().acoss_oti_batch(_ptr(corpus.gchroma), corpus.gchroma.shape[1], _ptr(batch.descs_dev),
                                      batch.K, _stream()), "oti_batch")


def csm(corpus, batch, out=None):
    """CRPUtils.py:67 for every pair; returns the flat CSM buffer (dtype of the features)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(max(batch.total_csm, 1), dtype=corpus.feats.dtype, device=corpus.device)
    fn = lib.acoss_csm_batch_f64 if corpus.dtype == np.float64 else lib.acoss_csm_batch_f32
    check(fn(_ptr(corpus.feats), _ptr(corpus.norms), corpus.d, _ptr(batch.descs_dev), batch.K,
             batch.max_nx, batch.max_ny, _ptr(out), _stream()), "csm_batch")
    return out


def pack_x(corpus, batch, out=None):
    """Rotated, line-packed x frames of every pair (input of csm_packed / crp)."""
    lib = _lib.load()
    need = int(lib.acoss_xpack_elems(batch.K, batch.max_nx))
    if out is None or out.numel() < need:
        out = torch.empty(max(need, 1), dtype=corpus.feats.dtype, device=corpus.device)
    fn = lib.acoss_pack_x_f64 if corpus.dtype == np.float64 else lib.acoss_pack_x_f32
    check(fn(_ptr(corpus.feats), _ptr(corpus.norms), corpus.d, _ptr(batch.descs_dev), batch.K, batch.max_nx,
             _ptr(out), _stream()), "pack_x")
    return out


def csm_packed(corpus, batch, xp, out=None):
    """CRPUtils.py:67 on packed x frames (same output as csm())."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(max(batch.total_csm, 1), dtype=corpus.feats.dtype, device=corpus.device)
    fn = lib.acoss_csm_packed_batch_f64 if corpus.dtype == np.float64 else lib.acoss_csm_packed_batch_f32
    check(fn(_ptr(xp), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d, _ptr(batch.descs_dev), batch.K,
             batch.max_nx, batch.max_ny, _ptr(out), _stream()), "csm_packed_batch")
    return out


def csm_strip(corpus, batch, xp, out=None):
    """CRPUtils.py:67 for float64 features through the persistent matrix-core strip kernel."""
    lib = _lib.load()
    if corpus.dtype != np.float64:
        raise AcossError("csm_strip: float64 features only")
    if out is None:
        out = torch.empty(max(batch.total_csm, 1), dtype=torch.float64, device=corpus.device)
    check(lib.acoss_csm_strip_batch_f64(_ptr(xp), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d,
                                        _ptr(batch.descs_dev), batch.K, batch.max_nx, batch.max_ny, _ptr(out),
                                        _stream()), "csm_strip_batch")
    return out


def csm_rows(corpus, batch, xp, out=None):
    """CRPUtils.py:67 for float64 features through the row-band matrix-core kernel (csrc/csm_rows_kernels.hip): the same
    matrix as csm / csm_packed / csm_strip, bit for bit; the HBM-bound form (whole-line stores straight from the accumulators)."""
    lib = _lib.load()
    if corpus.dtype != np.float64:
        raise AcossError("csm_rows: float64 features only")
    if out is None:
        out = torch.empty(max(batch.total_csm, 1), dtype=torch.float64, device=corpus.device)
    check(lib.acoss_csm_rows_batch_f64(_ptr(xp), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d,
                                       _ptr(batch.descs_dev), batch.K, batch.max_nx, batch.max_ny, _ptr(out),
                                       _stream()), "csm_rows_batch")
    return out


def crp(corpus, batch, xp, sqrt_out=False, out=None, force_valu=False, force_tile=False):
    """get_csm + sliding_csm fused (CRPUtils.py:67 + :24): windowed sums of squared distances
    (sqrt_out=False) or their square roots = sliding_csm's output (sqrt_out=True); float64."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(max(batch.total_crp, 1), dtype=torch.float64, device=corpus.device)
    fn = lib.acoss_crp_batch_f64 if corpus.dtype == np.float64 else lib.acoss_crp_batch_f32
    check(fn(_ptr(xp), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d, _ptr(batch.descs_dev), batch.K,
             batch.win, batch.max_nx, batch.max_ny, int(bool(sqrt_out)) | (2 if force_valu else 0) | (4 if force_tile else 0), _ptr(out),
             _stream()), "crp_batch")
    return out


def crp_planar(corpus, batch, xp, out=None):
    """The key high words of the windowed sums of crp() as an int32 device vector with the float64 matrix's element
    indexing (include/acoss_mi355x.h) -- the input of mask_bits_planar().  float64 features, win == 9, d in {12, 13}."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(planar_elems(batch), dtype=torch.int32, device=corpus.device)
    check(lib.acoss_crp_planar_batch_f64(_ptr(xp), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d,
                                         _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx, batch.max_ny,
                                         _ptr(out), _stream()), "crp_planar_batch")
    return out


def float32_copy(corpus):
    """(features, squared norms) of a float64 corpus, centred and rounded to float32 (of a float32 corpus: the corpus itself,
    see below), cached on the corpus: the operands
    of the approximate strip kernel (crp_planar32).  Every entry has the corpus mean subtracted first: distances between
    frames do not change (the shift is the same in every bin, so it commutes with the OTI rotation), but the squared
    norms -- the scale of the float32 error bound -- shrink to the variance part (about a third for chroma)."""
    if getattr(corpus, "_f32", None) is None and corpus.dtype == np.float32:
        # a float32 corpus (the reference's mfcc_htk / hpcp) IS its own filter operand: its exact path runs in float32 on
        # these very values (CRPUtils.py:82, :40-41), so the filter's cross-similarity values are bit for bit the exact ones
        n64 = corpus.norms.to(torch.float64)
        top = float(n64.max().item()) if n64.numel() else 0.0
        corpus._f32_ok = bool(np.isfinite(top) and top > 0.0 and bool(torch.isfinite(corpus.feats).all().item()))
        corpus._f32_scale2 = 1.0
        corpus._f32 = (corpus.feats, corpus.norms)
        corpus._f32_norms64 = n64.cpu().numpy()
    if getattr(corpus, "_f32", None) is None:
        mu = corpus.feats.mean()
        centred = corpus.feats - mu
        n64 = (centred * centred).sum(1)
        top = float(n64.max().item()) if n64.numel() else 0.0
        # a power-of-two scale brings the largest norm to ~1: exact in both precisions, the order of the windowed sums
        # (all the selection looks at) does not change, and float32 neither overflows nor loses small corpora to denormals
        corpus._f32_ok = bool(np.isfinite(top) and top > 0.0)
        scale = 2.0 ** -round(0.5 * np.log2(top)) if corpus._f32_ok else 1.0
        centred = centred * scale
        n64 = n64 * (scale * scale)
        corpus._f32_scale2 = scale * scale          # approximate values = this x the windowed sums of the original corpus
        corpus._f32 = (centred.to(torch.float32), n64.to(torch.float32))
        corpus._f32_norms64 = n64.cpu().numpy()
    return corpus._f32


def song_variation(corpus):
    """Per song, the mean squared distance of its frames from the song's mean frame (host float64, cached): the scale the
    float32 filter's resolution is measured against when two songs of very different loudness meet (serra09_scores_py)."""
    v = getattr(corpus, "_variation", None)
    if v is None:
        # segmented sums on the device, a few million frames at a time (whole songs), one download at the end
        off = np.asarray(corpus.frame_off, dtype=np.int64)
        lens = np.diff(off)
        parts = []
        s0 = 0
        while s0 < corpus.n_songs:
            s1 = int(np.searchsorted(off, off[s0] + (1 << 21), side="right")) - 1
            s1 = min(max(s1, s0 + 1), corpus.n_songs)
            a, b = int(off[s0]), int(off[s1])
            n = torch.as_tensor(np.maximum(lens[s0:s1], 1), dtype=torch.float64, device=corpus.device)
            if b > a:
                x = corpus.feats[a:b].to(torch.float64)
                seg = torch.repeat_interleave(torch.arange(s1 - s0, device=corpus.device), torch.as_tensor(lens[s0:s1], device=corpus.device))
                mean = torch.zeros((s1 - s0, x.shape[1]), dtype=torch.float64, device=corpus.device).index_add_(0, seg, x) / n[:, None]
                sq = ((x - mean[seg]) ** 2).sum(1)
                parts.append(torch.zeros(s1 - s0, dtype=torch.float64, device=corpus.device).index_add_(0, seg, sq) / n)
            else:
                parts.append(torch.zeros(s1 - s0, dtype=torch.float64, device=corpus.device))
            s0 = s1
        v = torch.cat(parts).cpu().numpy() if parts else np.zeros(0)
        corpus._variation = v
    return v


def planar32_usable(corpus):
    """False for corpora the float32 copy cannot represent (all-zero or non-finite features): those stay on float64."""
    float32_copy(corpus)
    return corpus._f32_ok


def pack_x32(corpus, batch, out=None):
    """pack_x() of the float32 copy of a float64 corpus."""
    lib = _lib.load()
    f32, n32 = float32_copy(corpus)
    need = int(lib.acoss_xpack_elems(batch.K, batch.max_nx))
    if out is None or out.numel() < need:
        out = torch.empty(max(need, 1), dtype=torch.float32, device=corpus.device)
    check(lib.acoss_pack_x_f32(_ptr(f32), _ptr(n32), corpus.d, _ptr(batch.descs_dev), batch.K, batch.max_nx,
                               _ptr(out), _stream()), "pack_x_f32")
    return out


def crp_planar32(corpus, batch, xp32, out=None):
    """Float32 approximation of the windowed sums as order-preserving uint32 keys (float32 bits | sign bit), same
    element indexing as crp_planar(); |approx - exact| <= 40 * 2^-24 * (window sums of the squared norms)."""
    lib = _lib.load()
    f32, n32 = float32_copy(corpus)
    if out is None:
        out = torch.empty(planar_elems(batch), dtype=torch.int32, device=corpus.device)
    check(lib.acoss_crp_planar32_batch(_ptr(xp32), _ptr(f32), _ptr(n32), corpus.d, _ptr(batch.descs_dev), batch.K,
                                       batch.win, batch.max_nx, batch.max_ny, _ptr(out), _stream()), "crp_planar32_batch")
    return out


# |approx - exact| <= PLANAR32_BOUND_W * (window sums of squared norms) + PLANAR32_BOUND_T * exact.  First-order analysis
# for a chain of round-to-nearest FMAs: 16 u W + 9 u T; that v_mfma_f32_16x16x4_f32 accumulates this
# way is pinned bit for bit by tests/test_gpu_fast_path.py::test_float32_strip_kernel_is_a_round_to_nearest_fma_chain.
# The constants carry 3-5 % for the second-order terms and the rounding of the band itself.
PLANAR32_BOUND_W = 16.5 * 2.0 ** -24
PLANAR32_BOUND_T = 9.5 * 2.0 ** -24


def planar32_bound_w(d, fused):
    """Coefficient of the window norm sums in the float32 error bound.  A d-step FMA chain costs (d + 2) u |x||y| on the
    dot product, i.e. (d + 2) u N on 2 x.y (N = |x|^2 + |y|^2); the strip kernel adds 2 u N for the norms (d = 12: 16 u N,
    DESIGN.md section 4); the fused band kernel carries the norms through the matrix product, which costs one u N more
    (band_kernels.hip).  Half a unit of margin for second-order terms."""
    return (d + (5.5 if fused else 4.5)) * 2.0 ** -24


def planar32_band(corpus, batch, fused=False):
    """Per pair of the batch two float32 (base, slope): twice the error bound of a value v of crp_planar32 (fused: of the
    band kernel) is base + slope * v, both rounded up (device tensor of 2 K floats)."""
    w = corpus.song_wmax(batch.win)
    bound_w = planar32_bound_w(corpus.d, fused)
    sx, sy = batch.descs["song_x"].astype(np.int64), batch.descs["song_y"].astype(np.int64)
    up = 1.0 + 2.0 ** -10        # covers the float32 rounding of base, slope and of base + slope * v in the kernels
    band = np.empty((batch.K, 2), dtype=np.float64)
    band[:, 0] = 2.0 * bound_w * (w[sx] + w[sy]) * up
    band[:, 1] = 2.0 * PLANAR32_BOUND_T * up
    b32 = band.astype(np.float32)
    b32 = np.where(b32.astype(np.float64) < band, np.nextafter(b32, np.float32(np.inf)), b32).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(b32.reshape(-1))).to(corpus.device)


def mask_bits_planar32(keys, band, corpus, batch, kappa, mutual=True, out=None, work=None):
    """mask_bits_planar() on the float32-approximate keys of crp_planar32(): rows / columns with another value inside the
    error band of their k-th smallest are finished exactly in float64; identical masks."""
    lib = _lib.load()
    max_m = batch.max_nx - batch.win + 1
    if out is None:
        out = torch.zeros(max(batch.K * max_m * bits_words(batch), 1), dtype=torch.int64, device=keys.device)
    need = int(lib.acoss_mask_bits_work_bytes(batch.K, batch.max_nx, batch.max_ny, batch.win))
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.uint8, device=keys.device)
    check(lib.acoss_mask_bits_planar32_batch(_ptr(keys), _ptr(band), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d,
                                             _ptr(batch.descs_dev), batch.K, batch.win,
                                             batch.max_nx, batch.max_ny, float(kappa), int(bool(mutual)), _ptr(out),
                                             _ptr(work), work.numel(), _stream()), "mask_bits_planar32_batch")
    return out, work


def keys16_koff(corpus, batch):
    """Per pair of the batch the uint32 offset of its 16-bit keys (csrc/keys16.h): the float32 bit pattern of 2 W 2^-7, W =
    the pair's bound on the window norm sums (song_wmax of x + of y; no windowed sum exceeds 2 W).  With k' = bits -sat koff,
    key16 = min(max(k' >> 11, (k' >> 9) -sat 49152), 0xFFFE): 14 mantissa bits over the three octaves below 2 W, 12 over the
    four below those.  The pattern is that of the float32 NOT BELOW 2 W (the kernels rely on koff >= 2 W 2^-7).  Device int32
    tensor (K)."""
    if corpus.dtype == np.float32:
        raise AcossError("keys16_koff: a float32 corpus takes its key range from the data (keys16_koff_f32)")
    w = corpus.song_wmax(batch.win)
    sx, sy = batch.descs["song_x"].astype(np.int64), batch.descs["song_y"].astype(np.int64)
    W64 = 2.0 * (w[sx] + w[sy])
    W = W64.astype(np.float32)
    W = np.where(W.astype(np.float64) < W64, np.nextafter(W, np.float32(np.inf)), W).astype(np.float32)
    bits = W.view(np.uint32).astype(np.int64) - (7 << 23)
    koff = np.where(np.isfinite(W) & (W > np.float32(2.0 ** -100)), bits, 0).astype(np.uint32)
    return torch.from_numpy(koff.view(np.int32).copy()).to(corpus.device)


def keys16_koff_f32(corpus, batch, xp32):
    """koff for a FLOAT32 corpus (its own filter operand, not centred): the key range hung on squared norms centred PER PAIR
    (|x - y|^2 <= 2 (|x - m|^2 + |y - m|^2) for any m) -- the raw norm sums of features with a common offset (real MFCC) lie
    far above every distance.  Device int32 tensor (K)."""
    lib = _lib.load()
    f32, n32 = float32_copy(corpus)
    koff = torch.empty(max(batch.K, 1), dtype=torch.int32, device=corpus.device)
    check(lib.acoss_keys16_koff_f32_batch(_ptr(xp32), _ptr(f32), _ptr(n32), corpus.d, _ptr(batch.descs_dev), batch.K, batch.win,
                                          batch.max_nx, batch.max_ny, _ptr(koff), _stream()), "keys16_koff_f32_batch")
    return koff[:batch.K]


def keys16_band_f32(corpus, batch):
    """(base, slope) of the error band for a float32 corpus: the filter's cross-similarity values are the exact path's bit for
    bit, so only the root-square (3 u) and the float32 window sum (4 u) separate the two: a purely relative band (9.5 u as
    for float64 corpora: margin), plus a few denormal steps."""
    up = 1.0 + 2.0 ** -10
    band = np.empty((batch.K, 2), dtype=np.float32)
    band[:, 0] = np.float32(2.0 ** -140)
    band[:, 1] = np.nextafter(np.float32(2.0 * PLANAR32_BOUND_T * up), np.float32(np.inf))
    return torch.from_numpy(np.ascontiguousarray(band.reshape(-1))).to(corpus.device)


def crp_keys16(corpus, batch, xp32, koff, out=None):
    """crp_planar32()'s windowed sums as 16-bit keys (int16 device vector, the uint32 matrix's element indexing)."""
    lib = _lib.load()
    f32, n32 = float32_copy(corpus)
    if out is None:
        out = torch.empty(planar_elems(batch) + 64, dtype=torch.int16, device=corpus.device)
    check(lib.acoss_crp_keys16_batch(_ptr(xp32), _ptr(f32), _ptr(n32), corpus.d, _ptr(batch.descs_dev), batch.K, batch.win,
                                     batch.max_nx, batch.max_ny, _ptr(koff), _ptr(out), _stream()), "crp_keys16_batch")
    return out


def mask_bits_keys16(keys16, band, koff, xp32, corpus, batch, kappa, mutual=True, out=None, work=None):
    """mask_bits_planar32() on the 16-bit keys of crp_keys16(): identical masks."""
    lib = _lib.load()
    f32, n32 = float32_copy(corpus)
    max_m = batch.max_nx - batch.win + 1
    if out is None:
        out = torch.zeros(max(batch.K * max_m * bits_words(batch), 1), dtype=torch.int64, device=keys16.device)
    need = int(lib.acoss_mask_bits_work_bytes(batch.K, batch.max_nx, batch.max_ny, batch.win))
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.uint8, device=keys16.device)
    mcode = {"rows_kernel_only": 2, "cols_kernel_only": 3}.get(mutual) if isinstance(mutual, str) else int(bool(mutual))
    if corpus.dtype == np.float32:
        # float32 features: the exact values behind the masks are the float32-input ones (sum of (double)(sqrtf(c)^2))
        check(lib.acoss_mask_bits_keys16_f32_batch(_ptr(keys16), _ptr(band), _ptr(koff), _ptr(xp32), _ptr(f32), _ptr(n32), corpus.d,
                                                   _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx, batch.max_ny, float(kappa),
                                                   mcode, _ptr(out), _ptr(work), work.numel(), _stream()), "mask_bits_keys16_f32_batch")
        return out, work
    check(lib.acoss_mask_bits_keys16_batch(_ptr(keys16), _ptr(band), _ptr(koff), _ptr(xp32), _ptr(f32), _ptr(n32), _ptr(corpus.feats),
                                           _ptr(corpus.norms), corpus.d, _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx,
                                           batch.max_ny, float(kappa), mcode, _ptr(out), _ptr(work),
                                           work.numel(), _stream()), "mask_bits_keys16_batch")
    return out, work


def mask_bits_keys16_unresolved(work, batch):
    """Pairs of the batch the last mask_bits_keys16() call on `work` left unresolved (int32 ndarray; synchronises).  Matrices up
    to 1024 x 1024: always empty; beyond (the long form of the radix selection, up to 2048 x 2048): pairs with exact ties --
    their masks are undefined and the caller takes them through mask_bits() on the float64 sums."""
    import ctypes
    lst = np.zeros(max(batch.K, 1), dtype=np.int32)
    n = ctypes.c_int(0)
    check(_lib.load().acoss_mask_bits_keys16_unresolved(_ptr(work), batch.K, batch.max_nx, batch.max_ny, batch.win,
                                                        lst.ctypes.data_as(ctypes.c_void_p), batch.K, ctypes.byref(n), _stream()),
          "mask_bits_keys16_unresolved")
    return lst[:n.value].copy()


def radix16_work(batch):
    """Workspace of radix16_stage() for this batch (uint8 device tensor)."""
    need = int(_lib.load().acoss_radix16_work_bytes(batch.K, batch.max_nx, batch.max_ny, batch.win))
    return torch.empty(need, dtype=torch.uint8, device=batch.descs_dev.device)


def radix16_stage(what, keys16, band, koff, corpus, batch, kappa, bits, rwork, mutual=True):
    """Stages of the radix selection that mask_bits_keys16() runs (csrc/radix16_kernels.hip; float64 corpora): what = 1 the
    column kernel, 2 the row kernel (+ the mask's base bits), 4 exact values of the work items and their cells; combinable."""
    check(_lib.load().acoss_radix16_stage(int(what), _ptr(keys16), _ptr(band), _ptr(koff), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d,
                                          _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx, batch.max_ny, float(kappa),
                                          1 if mutual else 0, _ptr(bits), _ptr(rwork), rwork.numel(), _stream()), "radix16_stage")


def keys16_supported(corpus, batch):
    """float64 or float32 chroma / MFCC-sized features, the reference's window, matrices up to 2048 x 2048 (beyond 1024 x 1024:
    the long forms of the radix selection -- ACOSS_RADIX16 on --, whose unresolved pairs the caller redoes)."""
    side = 2048 if _lib.load().acoss_radix16_enabled() else 1024
    return (corpus.d in (12, 13) and batch.win == 9 and batch.max_nx - batch.win + 1 <= side
            and batch.max_ny - batch.win + 1 <= side and planar32_usable(corpus))


def packed32(corpus):
    """The float32 copy of a float64 corpus (float32_copy) as 16-float packed frames [d values | squared norm | 0 ...]:
    the operand of the fused band kernel (mask_bits_fused); cached on the corpus."""
    if getattr(corpus, "_pk32", None) is None:
        f32, n32 = float32_copy(corpus)
        out = torch.empty((max(corpus.n_frames, 1), 16), dtype=torch.float32, device=corpus.device)
        check(_lib.load().acoss_pack_frames_f32(_ptr(f32), _ptr(n32), corpus.d, corpus.n_frames, _ptr(out), _stream()),
              "pack_frames_f32")
        corpus._pk32 = out
    return corpus._pk32


def fused_supported(corpus, batch):
    """The product path: float64 chroma / MFCC-sized features, the reference's window, matrices up to 1014 x 1014."""
    return (corpus.dtype == np.float64 and batch.K > 0
            and bool(_lib.load().acoss_mask_bits_fused_supported(corpus.d, batch.win, batch.max_nx, batch.max_ny))
            and planar32_usable(corpus))


def fused_side_rows(batch):
    """Default capacity of the side buffer of mask_bits_fused: 2 % of the batch's rows and columns (0.5-1 % ask for a slot
    on chroma features)."""
    rows = batch.K * (batch.max_nx + batch.max_ny - 2 * batch.win + 2)
    return int(max(1024, 0.02 * rows))


def mask_bits_fused(corpus, batch, kappa, mutual=True, band=None, out=None, work=None, side_rows=None, verify=True):
    """get_csm + sliding_csm + csm_to_binary_mutual of every pair without a matrix in HBM (csrc/band_kernels.hip): returns
    (bits, work) like mask_bits().  verify=True reads the undecided-row counter back (one synchronising 4-byte copy) and
    repeats the call with a larger side buffer if rows were dropped; verify=False leaves that check to the caller
    (fused_counter)."""
    lib = _lib.load()
    max_m = batch.max_nx - batch.win + 1
    if band is None:
        band = planar32_band(corpus, batch, fused=True)
    if out is None:
        out = torch.zeros(max(batch.K * max_m * 16, 1), dtype=torch.int64, device=corpus.device)
    pk = packed32(corpus)
    side_rows = int(side_rows or fused_side_rows(batch))
    while True:
        need = int(lib.acoss_mask_bits_fused_work_bytes(batch.K, batch.max_nx, batch.max_ny, batch.win, side_rows))
        if work is None or work.numel() < need:
            work = torch.empty(need, dtype=torch.uint8, device=corpus.device)
        check(lib.acoss_mask_bits_fused_batch(_ptr(pk), _ptr(band), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d,
                                              _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx, batch.max_ny,
                                              float(kappa), int(bool(mutual)), _ptr(out), _ptr(work), work.numel(),
                                              side_rows, _stream()), "mask_bits_fused_batch")
        if not verify:
            return out, work
        asked = int(fused_counter(work).item())
        if asked <= side_rows:
            return out, work
        side_rows = asked + 64          # rows were dropped: once more with room for all of them
        work = None


def fused_counter(work):
    """Device int32 view of the undecided-row counter of the last mask_bits_fused call on `work`."""
    off = int(_lib.load().acoss_mask_bits_fused_counter(_ptr(work))) - work.data_ptr()
    return work[off:off + 4].view(torch.int32)


def planar_elems(batch):
    """Words of the high-word matrix of a batch."""
    return max(batch.total_crp, 2)


PLANAR_PITCH_ALIGN = 32     # PairBatch(pitch_align=...) of the fast path: rows start on 128-byte lines


def planar_supported(corpus, batch):
    """The fast path: float64 chroma / MFCC-sized features, the reference's window, matrices up to 2048 x 2048."""
    return (corpus.dtype == np.float64 and corpus.d in (12, 13) and batch.win == 9
            and batch.max_nx - batch.win + 1 <= 2048 and batch.max_ny - batch.win + 1 <= 2048)


def bits_words(batch):
    """uint64 words per row of the bit-packed mask of this batch (16 up to 1024 x 1024, else 32)."""
    return int(_lib.load().acoss_mask_bits_words(batch.max_nx, batch.max_ny, batch.win))


def crp_supported(corpus, win):
    return corpus.d in (12, 13) and 1 <= win <= 16


def sliding(csm_buf, batch, out=None):
    """CRPUtils.py:24 for every pair; float64 out."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(max(batch.total_crp, 1), dtype=torch.float64, device=csm_buf.device)
    fn = lib.acoss_sliding_batch_f64 if csm_buf.dtype == torch.float64 else lib.acoss_sliding_batch_f32
    check(fn(_ptr(csm_buf), _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx, batch.max_ny,
             _ptr(out), _stream()), "sliding_batch")
    return out


def binarize(S_buf, batch, kappa, mutual=True, out=None, work=None):
    """CRPUtils.py:169 / :201 for every pair; uint8 out, same layout as S."""
    lib = _lib.load()
    if out is None:
        out = torch.zeros(max(batch.total_crp, 1), dtype=torch.uint8, device=S_buf.device)
    need = int(lib.acoss_binarize_work_bytes(batch.K, batch.max_nx, batch.max_ny, batch.win))
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.uint8, device=S_buf.device)
    check(lib.acoss_binarize_batch(_ptr(S_buf), _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx,
                                   batch.max_ny, float(kappa), int(bool(mutual)), _ptr(out), _ptr(work),
                                   work.numel(), _stream()), "binarize_batch")
    return out


def thresholds(S_buf, batch, kappa, mutual=True, work=None):
    """Row (and column) kNN thresholds of every pair's matrix; returns the workspace tensor."""
    lib = _lib.load()
    need = int(lib.acoss_binarize_work_bytes(batch.K, batch.max_nx, batch.max_ny, batch.win))
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.uint8, device=S_buf.device)
    check(lib.acoss_thresholds_batch(_ptr(S_buf), _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx,
                                     batch.max_ny, float(kappa), int(bool(mutual)), _ptr(work), work.numel(),
                                     _stream()), "thresholds_batch")
    return work


def align_fused(kind, T_buf, batch, work, mutual=True, boundary=0, params=None, scores=None):
    """qmax / dmax straight from the windowed sums and their thresholds (no mask in memory)."""
    lib = _lib.load()
    if scores is None:
        scores = torch.empty(max(batch.K, 1), dtype=torch.float32, device=T_buf.device)
    pp = ctypes.byref(params) if params is not None else None
    check(lib.acoss_align_fused_batch({"qmax": 0, "dmax": 1}[kind], _ptr(T_buf), _ptr(batch.descs_dev), batch.K,
                                      batch.win, batch.max_nx, batch.max_ny, int(bool(mutual)), _ptr(work),
                                      work.numel(), int(boundary), pp, _ptr(scores), _stream()), "align_fused_batch")
    return scores[:batch.K]


def mask_bits(S_buf, batch, kappa, mutual=True, out=None, work=None):
    """Bit-packed kNN mask of every pair ((K, max_m, bits_words) uint64; bit c of word w = column 64w + c)."""
    lib = _lib.load()
    max_m = batch.max_nx - batch.win + 1
    if out is None:
        out = torch.zeros(max(batch.K * max_m * bits_words(batch), 1), dtype=torch.int64, device=S_buf.device)
    need = int(lib.acoss_mask_bits_work_bytes(batch.K, batch.max_nx, batch.max_ny, batch.win))
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.uint8, device=S_buf.device)
    check(lib.acoss_mask_bits_batch(_ptr(S_buf), _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx,
                                    batch.max_ny, float(kappa), int(bool(mutual)), _ptr(out), _ptr(work),
                                    work.numel(), _stream()), "mask_bits_batch")
    return out, work


def mask_bits_planar(planes, corpus, batch, kappa, mutual=True, out=None, work=None):
    """mask_bits() from the key high words of crp_planar() (the corpus and the batch they were built from): identical
    result, half the bytes written and read."""
    lib = _lib.load()
    max_m = batch.max_nx - batch.win + 1
    if out is None:
        out = torch.zeros(max(batch.K * max_m * bits_words(batch), 1), dtype=torch.int64, device=planes.device)
    need = int(lib.acoss_mask_bits_work_bytes(batch.K, batch.max_nx, batch.max_ny, batch.win))
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.uint8, device=planes.device)
    check(lib.acoss_mask_bits_planar_batch(_ptr(planes), _ptr(corpus.feats), _ptr(corpus.norms), corpus.d,
                                           _ptr(batch.descs_dev), batch.K, batch.win,
                                           batch.max_nx, batch.max_ny, float(kappa), int(bool(mutual)), _ptr(out),
                                           _ptr(work), work.numel(), _stream()), "mask_bits_planar_batch")
    return out, work


def align_bits(kind, bits, batch, boundary=0, params=None, scores=None):
    """qmax / dmax / swalignimpconstrained from the bit-packed mask."""
    lib = _lib.load()
    if scores is None:
        scores = torch.empty(max(batch.K, 1), dtype=torch.float32, device=bits.device)
    pp = ctypes.byref(params) if params is not None else None
    check(lib.acoss_align_bits_batch({"qmax": 0, "dmax": 1, "swc": 2}[kind], _ptr(bits), _ptr(batch.descs_dev), batch.K,
                                     batch.win, batch.max_nx, batch.max_ny, int(boundary), pp, _ptr(scores),
                                     _stream()), "align_bits_batch")
    return scores[:batch.K]


def align_bits_qd(bits, batch, boundary=1, params=None):
    """(qmax, dmax) from the bit-packed mask in one sweep; boundary=1: dmax on the D qmax leaves (Serra09.py:173-175)."""
    lib = _lib.load()
    q = torch.empty(max(batch.K, 1), dtype=torch.float32, device=bits.device)
    d = torch.empty(max(batch.K, 1), dtype=torch.float32, device=bits.device)
    pp = ctypes.byref(params) if params is not None else None
    check(lib.acoss_align_bits_qd_batch(_ptr(bits), _ptr(batch.descs_dev), batch.K, batch.win, batch.max_nx, batch.max_ny,
                                        int(boundary), pp, _ptr(q), _ptr(d), _stream()), "align_bits_qd_batch")
    return q[:batch.K], d[:batch.K]


def bits_path_supported(batch):
    return batch.max_nx - batch.win + 1 <= 1024 and batch.max_ny - batch.win + 1 <= 1024


def unpack_mask_bits(bits, batch, p):
    """Host uint8 (M, N) view of pair p's bit-packed mask (tests)."""
    max_m = batch.max_nx - batch.win + 1
    M, N = int(batch.M[p]), int(batch.N[p])
    W = bits_words(batch)
    words = bits[p * max_m * W:(p * max_m + M) * W].cpu().numpy().view(np.uint64).reshape(M, W)
    b = np.unpackbits(words.view(np.uint8).reshape(M, 8 * W), axis=1, bitorder="little")
    return b[:, :N]


def fused_align_supported(batch):
    return batch.max_ny - batch.win + 1 <= 1024


def align(kind, B_buf, mats, D=None, boundary=0, params=None, max_cols=None, mats_dev=None, scores=None):
    """SequenceAlignment.c recurrences over a batch of matrices.  kind in {'qmax','dmax','swc'}.
    mats: numpy MAT_DESC array (mats_dev: the same bytes already on the device, to keep the
    upload out of a timed loop).  Returns the float32 score tensor (max cell per matrix)."""
    lib = _lib.load()
    K = len(mats)
    dev = B_buf.device
    if mats_dev is None:
        mats_dev = to_device_bytes(mats, dev)
    if scores is None:
        scores = torch.empty(max(K, 1), dtype=torch.float32, device=dev)
    if max_cols is None:
        max_cols = int(mats["cols"].max()) if K else 0
    pp = ctypes.byref(params) if params is not None else None
    if kind == "qmax":
        rc = lib.acoss_qmax_batch(_ptr(B_buf), _ptr(mats_dev), K, max_cols, _ptr(D), pp, _ptr(scores), _stream())
    elif kind == "dmax":
        rc = lib.acoss_dmax_batch(_ptr(B_buf), _ptr(mats_dev), K, max_cols, _ptr(D), int(boundary), pp,
                                  _ptr(scores), _stream())
    elif kind == "swc":
        rc = lib.acoss_swc_batch(_ptr(B_buf), _ptr(mats_dev), K, max_cols, _ptr(D), pp, _ptr(scores), _stream())
    else:
        raise ValueError("unknown alignment kind %r" % (kind,))
    check(rc, kind + "_batch")
    return scores[:K]


# ---------------------------------------------------------------------------------------------
# the Serra09 chain
# ---------------------------------------------------------------------------------------------
_SCRATCH = {}


def _scratch(name, numel, dtype, device, zero=False):
    """Grow-only device scratch buffers shared by successive calls of the chain functions."""
    key = (name, str(device), dtype)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < numel:
        _SCRATCH.pop(key, None)
        buf = None
        torch.cuda.empty_cache() if numel * torch.empty((), dtype=dtype).element_size() > (1 << 30) else None
        buf = (torch.zeros if zero else torch.empty)(int(numel * 1.05) + 16, dtype=dtype, device=device)
        _SCRATCH[key] = buf
    elif zero:
        buf[:numel].zero_()
    return buf[:numel]


def _scorer_scratch(need, device):
    """The scorer's device scratch: ONE plain allocation, grow-only, shared by successive calls.  (Rounds 2-3 timed trial
    batches in several candidate allocations here, because the column-strip form of the strip kernel ran 5-10 % faster or
    slower depending on the allocation's physical backing; with the row-band strip kernel the spread is 0-3 % (DESIGN.md
    appendix A) and the trials went.)"""
    key = ("scorer", str(device), torch.uint8)
    buf = _SCRATCH.get(key)
    if buf is not None and buf.numel() >= need:
        return buf[:need]
    _SCRATCH.pop(key, None)
    buf = None
    torch.cuda.empty_cache()
    buf = torch.empty(int(need * 1.05) + 16, dtype=torch.uint8, device=device)
    _SCRATCH[key] = buf
    return buf[:need]


def release_scratch():
    """Drop the cached scratch buffers (tests, memory-tight callers)."""
    _SCRATCH.clear()
    torch.cuda.empty_cache()


def keys16_default():
    """Whether the float32 filter's keys are 16 bits wide (csrc/keys16.h: the default since round 3) or 32 (round 2's form,
    ACOSS_KEYS16=0: the C scorer reads the same name when a corpus handle is made)."""
    return os.environ.get("ACOSS_KEYS16", "1") not in ("0", "", "false", "no")


def planar32_default():
    """Whether the chain uses the float32-filter form of the strip kernel (crp_planar32 + mask_bits_planar32: float32 keys,
    rows and columns inside the error band refined exactly in float64: identical masks and scores) when the caller does
    not say.  On by default; ACOSS_PLANAR32=0 keeps every windowed sum in float64."""
    return os.environ.get("ACOSS_PLANAR32", "1") not in ("0", "", "false", "no")


def fused_default():
    """ACOSS_FUSED=1: masks from the fused band kernel (mask_bits_fused) instead of the materialising kernels.  Off by
    default: bit-identical, but 1.5 ms slower per 4096 pairs of 1000-frame songs."""
    return os.environ.get("ACOSS_FUSED", "0") not in ("0", "", "false", "no")


def _c_corpus(corpus, with32):
    """The C-side handle of a float64 corpus (acoss_corpus_wrap around this object's device arrays): what
    acoss_serra09_scores takes.  with32: hand over the centred float32 copy too (float32 filter on)."""
    attr = "_chandle32" if with32 else "_chandle64"
    h = getattr(corpus, attr, None)
    if h is None:
        lib = _lib.load()
        f32 = n32 = ns = None
        if with32:
            f32, n32 = float32_copy(corpus)
            ns = np.ascontiguousarray(corpus._f32_norms64, dtype=np.float64)
        h = ctypes.c_void_p()
        nb = int(corpus.gchroma.shape[1]) if corpus.gchroma is not None else 0
        check(lib.acoss_corpus_wrap(_ptr(corpus.feats), _ptr(corpus.norms), _ptr(corpus.gchroma), nb, _ptr(f32), _ptr(n32),
                                    ns.ctypes.data if ns is not None else None, corpus.frame_off.ctypes.data,
                                    corpus.n_songs, corpus.d, ctypes.byref(h)), "corpus_wrap")
        setattr(corpus, attr, h)
        corpus._chandles = getattr(corpus, "_chandles", []) + [h]
    return h


def serra09_scores(corpus, pairs, m=9, kappa=0.095, do_oti=True, want=("qmax", "dmax"), batch_pairs=None, approx32=None):
    """
    Serra09.py:166-175 for every pair: OTI -> cross-similarity + sliding window -> mutual kNN mask -> qmax [-> dmax on the
    D qmax leaves behind] [-> swalignimpconstrained, BASELINE config 3: the reference's plugins call it as
    `swconstrained(B, D, M, N) / (M + N)`, EarlySNF_Old.py:199-203].  Scores are divided by (M + N).

    float64 corpora go through the library's own scorer (acoss_serra09_scores, csrc/scorer.hip: batch planning, size
    classes, the float32 filter, one synchronisation at the end); this function only hands over the pair list and a
    scratch buffer.  approx32=False (or ACOSS_PLANAR32=0) keeps the windowed sums in float64.
    """
    bad = [k for k in want if k not in ("qmax", "dmax", "swc")]
    if bad:
        raise AcossError("serra09_scores: unknown score(s) %r (qmax, dmax, swc)" % (bad,))
    if corpus.dtype == np.float64 and not fused_default():
        lib = _lib.load()
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        K = pairs.shape[0]
        out = {k: np.zeros(K) for k in want}
        if K == 0:
            return out
        use32 = (planar32_default() if approx32 is None else bool(approx32)) and planar32_usable(corpus)
        h = _c_corpus(corpus, use32)
        bp = int(batch_pairs) if batch_pairs else 0
        need = int(lib.acoss_serra09_scratch_bytes(h, pairs.ctypes.data, K, int(m), bp))
        if need == 0:
            raise AcossError("serra09_scores: %s" % _lib.last_error())
        scratch = _scorer_scratch(need, corpus.device)
        mask = sum(b for k, b in (("qmax", 1), ("dmax", 2), ("swc", 4)) if k in want)
        ptr = lambda k: out[k].ctypes.data if k in out else None
        check(lib.acoss_serra09_scores(h, pairs.ctypes.data, K, int(m), float(kappa), int(bool(do_oti)), mask, bp,
                                       _ptr(scratch), scratch.numel(), ptr("qmax"), ptr("dmax"), ptr("swc"), _stream()),
              "serra09_scores")
        return out
    return serra09_scores_py(corpus, pairs, m, kappa, do_oti, want, batch_pairs, approx32)


def serra09_scores_py(corpus, pairs, m=9, kappa=0.095, do_oti=True, want=("qmax", "dmax"), batch_pairs=None, approx32=None):
    """
    The same chain composed in Python from the stage entry points (float32 corpora; ACOSS_FUSED=1: masks from the fused
    band kernel; and the tests that compare the two compositions).  Falls back to the staged chain for shapes the fused
    kernels do not cover.
    """
    if not crp_supported(corpus, m):
        return serra09_scores_staged(corpus, pairs, m, kappa, do_oti, want, batch_pairs)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    K = pairs.shape[0]
    out = {k: np.zeros(K) for k in want}
    if K == 0:
        return out
    # size classes go through separate batches: matrices up to 1024 x 1024 (16 values per lane in the selection and
    # alignment kernels), up to 2048 x 2048 (32 per lane), and beyond (byte-mask path), so that one long song does not
    # push a whole batch onto the slower kernels
    lens = corpus.lengths()
    side = np.maximum(lens[pairs[:, 0]], lens[pairs[:, 1]]) - m + 1
    cls = (side > 1024).astype(np.int8) + (side > 2048)
    if cls.min() != cls.max():
        for c in np.unique(cls):
            part = np.flatnonzero(cls == c)
            res = serra09_scores_py(corpus, pairs[part], m, kappa, do_oti, want, batch_pairs, approx32)
            for k in want:
                out[k][part] = res[k]
        return out
    if corpus.dtype == np.float32 and approx32 is not False and planar32_default():
        # the float32 filter cannot tell the cells of a row apart when one song of the pair has (next to) no variation
        # against the other's scale -- a silent or constant track: every row or column is then refined exactly, cell by cell,
        # which costs far more than the plain chain: such pairs take the plain float32-input chain
        v = song_variation(corpus)
        a, b = v[pairs[:, 0]], v[pairs[:, 1]]
        flat = np.minimum(a, b) < 2.0 ** -20 * np.maximum(a, b)
        if flat.any():
            for part, a32 in ((np.flatnonzero(~flat), approx32), (np.flatnonzero(flat), False)):
                if len(part):
                    res = serra09_scores_py(corpus, pairs[part], m, kappa, do_oti, want, batch_pairs, a32)
                    for k in want:
                        out[k][part] = res[k]
            return out
    if batch_pairs is None:
        # ~4096 pairs of 1000-frame songs per launch batch (34 GB of the 288): the one-wave-per-pair alignment kernel
        # needs thousands of pairs in flight
        per_pair = float(max(lens[pairs[:, 0]].max(), lens[pairs[:, 1]].max())) ** 2 * 9.2
        batch_pairs = int(max(1, min(K, (36 << 30) // max(per_pair, 1.0))))
        if batch_pairs >= 4096 and K > batch_pairs:
            batch_pairs -= batch_pairs % 4096          # whole waves-per-SIMD rounds for the one-wave-per-pair alignment kernels
    # The scores of a batch travel to pinned host memory asynchronously, behind its kernels on the stream, and are read after ONE
    # synchronisation at the end: the host plans and launches batch after batch while the GPU works (every scratch buffer is
    # reused in stream order), instead of waiting for each batch's scores before it plans the next.
    pending = []
    redo = []
    long_off = False

    def fetch_later(lo_, n_, denom_, got_):
        host = {}
        for kind, t in got_.items():
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            host[kind] = h
        pending.append((lo_, n_, denom_, host))

    def flush():
        if pending:
            torch.cuda.synchronize()
        while pending:
            lo_, n_, denom_, host = pending.pop(0)
            for kind, h in host.items():
                out[kind][lo_:lo_ + n_] = h.numpy().astype(np.float64) / denom_
    for lo in range(0, K, batch_pairs):
        if long_off:
            # (a corpus whose thresholds fall below the keys' seven octaves -- unbounded random walks -- left most pairs of the last
            #  batch of long songs unresolved: the rest of the call goes without the filter)
            redo.append(np.arange(lo, K, dtype=np.int64))
            break
        sel = pairs[lo:lo + batch_pairs]
        batch = PairBatch(corpus.frame_off, sel, m, corpus.device, pitch_align=PLANAR_PITCH_ALIGN)
        if do_oti:
            oti(corpus, batch)
        planar = planar_supported(corpus, batch)
        use32 = planar and (planar32_default() if approx32 is None else bool(approx32)) and planar32_usable(corpus)
        # float32 corpora (the reference's mfcc_htk): the same filter on 16-bit keys with the corpus itself as its operand,
        # refined with the exact float32-input arithmetic (round 4); everything else about the batch as for float64
        k16_ok = keys16_supported(corpus, batch)
        filter32 = (corpus.dtype == np.float32 and (planar32_default() if approx32 is None else bool(approx32)) and keys16_default()
                    and not fused_default() and k16_ok)
        if filter32:
            planar = use32 = True
        fused = use32 and fused_default() and fused_supported(corpus, batch)
        # scratch buffers live across calls (grow-only): a fresh 16-34 GB allocation per call costs more than the batch
        xp = None if use32 else pack_x(corpus, batch, out=_scratch("xp", int(_lib.load().acoss_xpack_elems(batch.K, batch.max_nx)), corpus.feats.dtype, corpus.device))
        T = _scratch("T", (batch.total_crp + 1) // 2 + 1 if planar else max(batch.total_crp, 1), torch.float64, corpus.device)
        work = _scratch("work", int(_lib.load().acoss_mask_bits_work_bytes(batch.K, batch.max_nx, batch.max_ny, m)), torch.uint8, corpus.device)
        B = None if planar or bits_path_supported(batch) or fused_align_supported(batch) \
            else _scratch("B", max(batch.total_crp, 1), torch.uint8, corpus.device, zero=True)
        bits_buf = _scratch("bits", max(batch.K * (batch.max_nx - m + 1) * bits_words(batch), 1), torch.int64, corpus.device) \
            if (planar or bits_path_supported(batch)) else None
        denom = (batch.M + batch.N).astype(np.float64)
        if "swc" in want and not (planar or bits_path_supported(batch)):
            # no bit-mask path for this batch: byte mask + dp_wave_kernel / dp_block_kernel
            B = _scratch("B", max(batch.total_crp, 1), torch.uint8, corpus.device, zero=True)
            crp(corpus, batch, xp, sqrt_out=False, out=T)
            binarize(T, batch, kappa, mutual=True, out=B, work=work)
            mats, _ = batch.mats()
            for kind, kw in (("qmax", {}), ("dmax", {"boundary": 1}), ("swc", {})):
                if kind in want:
                    out[kind][lo:lo + len(sel)] = align(kind, B, mats, **kw).cpu().numpy().astype(np.float64) / denom
            continue
        if fused:
            bits, _ = mask_bits_fused(corpus, batch, kappa, mutual=True, out=bits_buf)
        elif use32 and keys16_default() and k16_ok:
            xp32 = pack_x32(corpus, batch, out=_scratch("xp32", int(_lib.load().acoss_xpack_elems(batch.K, batch.max_nx)), torch.float32, corpus.device))
            koff = keys16_koff_f32(corpus, batch, xp32) if filter32 else keys16_koff(corpus, batch)
            band = keys16_band_f32(corpus, batch) if filter32 else planar32_band(corpus, batch)
            k16 = crp_keys16(corpus, batch, xp32, koff, out=T.view(torch.int16)[:planar_elems(batch) + 64])
            bits, work = mask_bits_keys16(k16, band, koff, xp32, corpus, batch, kappa, mutual=True, out=bits_buf, work=work)
            if bits_words(batch) == 32:
                # the long form leaves pairs with exact ties unresolved (one synchronisation per batch of long songs): scored
                # again below without the filter
                un = mask_bits_keys16_unresolved(work, batch)
                if len(un):
                    redo.append(lo + un.astype(np.int64))
                long_off = len(un) > batch.K // 4
        elif use32:
            xp32 = pack_x32(corpus, batch, out=_scratch("xp32", int(_lib.load().acoss_xpack_elems(batch.K, batch.max_nx)), torch.float32, corpus.device))
            keys = crp_planar32(corpus, batch, xp32, out=T.view(torch.int32)[:planar_elems(batch)])
            bits, work = mask_bits_planar32(keys, planar32_band(corpus, batch), corpus, batch, kappa, mutual=True, out=bits_buf, work=work)
        elif planar:
            planes = crp_planar(corpus, batch, xp, out=T.view(torch.int32)[:planar_elems(batch)])
            bits, work = mask_bits_planar(planes, corpus, batch, kappa, mutual=True, out=bits_buf, work=work)
        else:
            crp(corpus, batch, xp, sqrt_out=False, out=T)
        if planar or bits_path_supported(batch):
            if not planar:
                bits, work = mask_bits(T, batch, kappa, mutual=True, out=bits_buf, work=work)
            got = {}
            if "qmax" in want and "dmax" in want:
                got["qmax"], got["dmax"] = align_bits_qd(bits, batch, boundary=1)
            elif "qmax" in want:
                got["qmax"] = align_bits("qmax", bits, batch)
            elif "dmax" in want:
                got["dmax"] = align_bits("dmax", bits, batch, boundary=1)
            if "swc" in want:
                got["swc"] = align_bits("swc", bits, batch)
            fetch_later(lo, len(sel), denom, got)
            continue
        flush()
        if fused_align_supported(batch):
            work = thresholds(T, batch, kappa, mutual=True, work=work)
            if "qmax" in want:
                out["qmax"][lo:lo + len(sel)] = align_fused("qmax", T, batch, work).cpu().numpy().astype(np.float64) / denom
            if "dmax" in want:
                out["dmax"][lo:lo + len(sel)] = align_fused("dmax", T, batch, work, boundary=1).cpu().numpy().astype(np.float64) / denom
            continue
        binarize(T, batch, kappa, mutual=True, out=B, work=work)
        mats, _ = batch.mats()
        if "qmax" in want:
            out["qmax"][lo:lo + len(sel)] = align("qmax", B, mats).cpu().numpy().astype(np.float64) / denom
        if "dmax" in want:
            out["dmax"][lo:lo + len(sel)] = align("dmax", B, mats, boundary=1).cpu().numpy().astype(np.float64) / denom
    flush()
    if redo:
        part = np.concatenate(redo)
        res = serra09_scores_py(corpus, pairs[part], m, kappa, do_oti, want, batch_pairs, False)       # (no filter: float64 keys / the plain chain)
        for k in want:
            out[k][part] = res[k]
    return out


# every intermediate materialised in HBM, one kernel per reference function
def serra09_scores_staged(corpus, pairs, m=9, kappa=0.095, do_oti=True, want=("qmax", "dmax"),
                          batch_pairs=None, keep=None):
    """
    Serra09.py:166-175 for every pair through the stage kernels: oti -> csm -> sliding ->
    mutual binarise -> qmax [-> dmax on the boundary qmax leaves behind].
    Returns {kind: float64 ndarray(K)} of scores already divided by (M+N).
    keep: optional dict that receives the last batch's device buffers (for tests).
    """
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    K = pairs.shape[0]
    out = {k: np.zeros(K) for k in want}
    if K == 0:
        return out
    if batch_pairs is None:
        lens = corpus.lengths()
        per_pair = float(lens.max()) ** 2 * (corpus.feats.element_size() + 9)
        batch_pairs = int(max(1, min(K, (8 << 30) // max(per_pair, 1.0))))
    for lo in range(0, K, batch_pairs):
        sel = pairs[lo:lo + batch_pairs]
        batch = PairBatch(corpus.frame_off, sel, m, corpus.device)
        if do_oti:
            oti(corpus, batch)
        C = csm(corpus, batch)
        S = sliding(C, batch)
        B = binarize(S, batch, kappa, mutual=True)
        mats, _ = batch.mats()
        denom = (batch.M + batch.N).astype(np.float64)
        if "qmax" in want:
            out["qmax"][lo:lo + len(sel)] = align("qmax", B, mats).cpu().numpy().astype(np.float64) / denom
        if "dmax" in want:
            out["dmax"][lo:lo + len(sel)] = align("dmax", B, mats, boundary=1).cpu().numpy().astype(np.float64) / denom
        if "swc" in want:
            out["swc"][lo:lo + len(sel)] = align("swc", B, mats).cpu().numpy().astype(np.float64) / denom
        if keep is not None:
            keep.update(batch=batch, C=C, S=S, B=B)
    return out


# ---------------------------------------------------------------------------------------------
# FTM2D (benchmarking/FTM2D.py): 2D Fourier transform magnitude shingles and their similarity
# ---------------------------------------------------------------------------------------------
def ftm2d_shingles(btchromas, pwr=1.96, C=5):
    """One 900-d shingle per song (FTM2D.py:92-100) from beat-synchronous chroma: btchromas is a list of
    (12, nbeats) arrays (what librosa.util.sync returns at :91).  Returns a (n_songs, 900) float64 device tensor;
    songs with fewer than 75 beats get zeros (:87-90)."""
    lib = _lib.load()
    require_gpu()
    dev = "cuda:%d" % torch.cuda.current_device()
    n = len(btchromas)
    nb = np.array([b.shape[1] for b in btchromas], dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum(nb)
    for b in btchromas:
        if b.shape[0] != 12:
            raise AssertionError('beat-aligned matrix transposed?')          # FTM2D.py:37
    flat = np.concatenate([np.ascontiguousarray(b.T, dtype=np.float64) for b in btchromas], axis=0) if n else np.zeros((0, 12))
    bt = torch.from_numpy(np.ascontiguousarray(flat)).to(dev) if flat.size else torch.zeros((1, 12), dtype=torch.float64, device=dev)
    windows = int(np.maximum(nb - 74, 0).sum())
    scratch = torch.empty(int(lib.acoss_ftm2d_scratch_bytes(int(off[-1]), windows, n)), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 900), dtype=torch.float64, device=dev)
    check(lib.acoss_ftm2d_shingles(_ptr(bt), off.ctypes.data, n, float(pwr), float(C), _ptr(scratch), scratch.numel(),
                                   _ptr(out), _stream()), "ftm2d_shingles")
    return out


def ftm2d_pairs(shingles, pairs):
    """exp(-|s_i - s_j|^2) for the listed pairs (FTM2D.py:117-127): float64 ndarray(K)."""
    lib = _lib.load()
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    K = pairs.shape[0]
    if K == 0:
        return np.zeros(0)
    pd = torch.from_numpy(pairs).to(shingles.device)
    out = torch.empty(K, dtype=torch.float64, device=shingles.device)
    check(lib.acoss_ftm2d_pairs(_ptr(shingles), _ptr(pd), K, _ptr(out), _stream()), "ftm2d_pairs")
    return out.cpu().numpy()


def ftm2d_gram(shingles):
    """All n x n similarities as one product on the float64 matrix cores: (n, n) float64 device tensor."""
    lib = _lib.load()
    n = shingles.shape[0]
    out = torch.empty((n, n), dtype=torch.float64, device=shingles.device)
    check(lib.acoss_ftm2d_gram(_ptr(shingles), n, _ptr(out), _stream()), "ftm2d_gram")
    return out


# ---------------------------------------------------------------------------------------------
# Early similarity network fusion (benchmarking/EarlySNF.py:41-90, SimilarityFusion.py)
# ---------------------------------------------------------------------------------------------
class _SnfFeature(ctypes.Structure):
    _fields_ = [("ssma", ctypes.c_void_p), ("ssmb", ctypes.c_void_p), ("csm", ctypes.c_void_p),
                ("da", ctypes.c_void_p), ("db", ctypes.c_void_p), ("dc", ctypes.c_void_p), ("win", ctypes.c_int32)]


def snf_cross(features, M, N, kappa, dout, out=None, niters=3, mu=0.5, debug=False):
    """-fused[0:M, M:] of every pair (EarlySNF.py:83-85) from per-feature distance matrices.
    features: list of dicts {ssma, ssmb, csm: float64 device buffers; da, db, dc: PairBatch (layouts); win}.
    M, N: int arrays (K).  dout: PairBatch whose crp layout receives the result (float64 buffer `out`).
    debug=True also returns (W [n_feat][sum L^2], fused [sum L^2]) device tensors."""
    lib = _lib.load()
    K = len(M)
    M = np.ascontiguousarray(M, dtype=np.int32)
    N = np.ascontiguousarray(N, dtype=np.int32)
    dev = dout.descs_dev.device
    arr = (_SnfFeature * len(features))()
    for f, ft in enumerate(features):
        arr[f].ssma, arr[f].ssmb, arr[f].csm = _ptr(ft["ssma"]), _ptr(ft["ssmb"]), _ptr(ft["csm"])
        arr[f].da, arr[f].db, arr[f].dc = _ptr(ft["da"].descs_dev), _ptr(ft["db"].descs_dev), _ptr(ft["dc"].descs_dev)
        arr[f].win = int(ft["win"])
    need = int(lib.acoss_snf_scratch_bytes(M.ctypes.data, N.ctypes.data, K, len(features)))
    scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.zeros(max(dout.total_crp, 1), dtype=torch.float64, device=dev)
    tot = int(((M.astype(np.int64) + N) ** 2).sum())
    dW = torch.empty((len(features), tot), dtype=torch.float64, device=dev) if debug else None
    dF = torch.empty(tot, dtype=torch.float64, device=dev) if debug else None
    check(lib.acoss_snf_cross_batch(ctypes.addressof(arr), len(features), K, M.ctypes.data, N.ctypes.data, float(kappa),
                                    float(mu), int(niters), _ptr(scratch), need, _ptr(dout.descs_dev), _ptr(out),
                                    _ptr(dW) if debug else None, _ptr(dF) if debug else None, _stream()), "snf_cross_batch")
    return (out, dW, dF) if debug else out


def early_snf_scores(chroma, ssms, pairs, m=9, kappa=0.095, do_oti=True, batch_pairs=None, want=("qmax", "dmax")):
    """
    EarlySNF.py:41-90 for every pair: the chroma block affinity (CSM and both SSMs with the sliding window) and the
    'ssms' feature block affinity (no window) are fused by 3 cross-diffusion steps; the negated cross block goes
    through the mutual kNN mask and qmax [/ dmax on the same D].  chroma: DeviceCorpus of (n, 12) frames;
    ssms: DeviceCorpus of the per-song (n - m + 1, d) features.  Scores are divided by (M + N).
    """
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    K = pairs.shape[0]
    out = {k: np.zeros(K) for k in want}
    if K == 0:
        return out
    if batch_pairs is None:
        L = 2.0 * float(chroma.lengths().max())
        batch_pairs = int(max(1, min(K, (24 << 30) // max(8.5 * 8.0 * L * L, 1.0))))
    dev = chroma.device
    for lo in range(0, K, batch_pairs):
        sel = pairs[lo:lo + batch_pairs]
        aa = np.stack([sel[:, 0], sel[:, 0]], axis=1)
        bb = np.stack([sel[:, 1], sel[:, 1]], axis=1)
        feats = []
        # chroma: windowed distances, OTI on the cross pair only (a common roll does not change the self distances)
        bc, ba, bbb = (PairBatch(chroma.frame_off, q, m, dev) for q in (sel, aa, bb))
        if do_oti:
            oti(chroma, bc)
        mats = []
        for b in (ba, bbb, bc):
            xp = pack_x(chroma, b)
            mats.append(crp(chroma, b, xp, sqrt_out=True))
        feats.append(dict(ssma=mats[0], ssmb=mats[1], csm=mats[2], da=ba, db=bbb, dc=bc, win=m))
        # 'ssms' features: plain Euclidean distances (get_csm / get_ssm, EarlySNF.py:72-74)
        sc, sa, sb = (PairBatch(ssms.frame_off, q, 1, dev) for q in (sel, aa, bb))
        feats.append(dict(ssma=csm(ssms, sa), ssmb=csm(ssms, sb), csm=csm(ssms, sc), da=sa, db=sb, dc=sc, win=1))
        if not (np.array_equal(sc.descs["nx"], bc.M) and np.array_equal(sc.descs["ny"], bc.N)):
            raise AcossError("early_snf: the 'ssms' features must have nframes - m + 1 rows per song")
        cross = snf_cross(feats, bc.M, bc.N, kappa, bc)
        denom = (bc.M + bc.N).astype(np.float64)
        if bits_path_supported(bc):
            bits, _ = mask_bits(cross, bc, kappa, mutual=True)
            if "qmax" in want:
                out["qmax"][lo:lo + len(sel)] = align_bits("qmax", bits, bc).cpu().numpy().astype(np.float64) / denom
            if "dmax" in want:
                out["dmax"][lo:lo + len(sel)] = align_bits("dmax", bits, bc, boundary=1).cpu().numpy().astype(np.float64) / denom
        else:
            B = binarize(cross, bc, kappa, mutual=True)
            mats_d, _ = bc.mats()
            if "qmax" in want:
                out["qmax"][lo:lo + len(sel)] = align("qmax", B, mats_d).cpu().numpy().astype(np.float64) / denom
            if "dmax" in want:
                out["dmax"][lo:lo + len(sel)] = align("dmax", B, mats_d, boundary=1).cpu().numpy().astype(np.float64) / denom
    return out
