"""numpy-facing wrappers over the C ABI, one per third-party call on the KARIOS hot path.

Names and argument meaning follow the calls the reference makes (cv2.Laplacian,
cv2.goodFeaturesToTrack, cv2.calcOpticalFlowPyrLK, skimage phase_cross_correlation,
karios.core.image.shift_image); file:line citations are relative to the reference tree.
All compute happens in libkarios_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Context, KariosHipError, KltParams, as_image, default_context, dtype_code, ptr, row_stride


def _ctx(ctx):
    return ctx if ctx is not None else default_context()


def to_uint8(arr, invert: bool = False, ctx: Context | None = None, return_minmax: bool = False):
    """_to_uint8 (matcher/klt.py:42-49) [+ `255 - x`, klt.py:419]."""
    c = _ctx(ctx)
    a = as_image(arr)
    out = np.empty(a.shape, np.uint8)
    mm = (C.c_double * 2)()
    c.check(c.lib.km_to_uint8(c.handle, ptr(a), dtype_code(a), a.shape[0], a.shape[1], row_stride(a),
                              int(bool(invert)), ptr(out), mm), "km_to_uint8")
    return (out, (mm[0], mm[1])) if return_minmax else out


def auto_mask(mon, ref, nodata_mon=None, nodata_ref=None, ctx: Context | None = None):
    """Automatic validity mask (klt.py:268-273) -> (uint8 mask, valid pixel count)."""
    c = _ctx(ctx)
    m, r = as_image(mon), as_image(ref)
    if m.shape != r.shape or m.dtype != r.dtype:
        raise KariosHipError("auto_mask: mon/ref shape or dtype mismatch")
    mask = np.empty(m.shape, np.uint8)
    valid = C.c_int64()
    nm = C.byref(C.c_double(float(nodata_mon))) if nodata_mon is not None else None
    nr = C.byref(C.c_double(float(nodata_ref))) if nodata_ref is not None else None
    c.check(c.lib.km_auto_mask(c.handle, ptr(m), ptr(r), dtype_code(m), m.shape[0], m.shape[1], row_stride(m),
                               row_stride(r), nm, nr, ptr(mask), C.byref(valid)), "km_auto_mask")
    return mask, int(valid.value)


def laplacian_u8(img, ksize: int, ctx: Context | None = None):
    """cv2.Laplacian(img, cv2.CV_8U, ksize=ksize) (klt.py:433-434)."""
    c = _ctx(ctx)
    a = np.ascontiguousarray(img, np.uint8)
    if a.ndim != 2:
        raise KariosHipError("laplacian_u8: expected a 2-D uint8 image")
    out = np.empty_like(a)
    c.check(c.lib.km_laplacian_u8(c.handle, ptr(a), a.shape[0], a.shape[1], int(ksize), ptr(out)), "km_laplacian_u8")
    return out


def min_eigen(img, block_size: int, ctx: Context | None = None):
    """cornerMinEigenVal map used inside goodFeaturesToTrack."""
    c = _ctx(ctx)
    a = np.ascontiguousarray(img, np.uint8)
    out = np.empty(a.shape, np.float32)
    c.check(c.lib.km_min_eigen(c.handle, ptr(a), a.shape[0], a.shape[1], int(block_size), ptr(out)), "km_min_eigen")
    return out


def good_features_to_track(image, maxCorners, qualityLevel, minDistance, mask=None, blockSize=3,
                           ctx: Context | None = None):
    """cv2.goodFeaturesToTrack(image, mask=, maxCorners, qualityLevel, minDistance, blockSize)
    (klt.py:120, 494) -> (N,1,2) float32, or None when no corner is found."""
    c = _ctx(ctx)
    a = np.ascontiguousarray(image, np.uint8)
    if a.ndim != 2:
        raise KariosHipError("goodFeaturesToTrack: expected a 2-D uint8 image")
    m = None
    if mask is not None:
        m = np.ascontiguousarray(mask, np.uint8)
        if m.shape != a.shape:
            raise KariosHipError("goodFeaturesToTrack: mask shape mismatch")
    cap = int(maxCorners) if maxCorners > 0 else max(1, (a.shape[0] * a.shape[1]) // 4)
    out = np.empty((cap, 2), np.float32)
    n = C.c_int()
    c.check(c.lib.km_good_features(c.handle, ptr(a), ptr(m), a.shape[0], a.shape[1], int(maxCorners),
                                   float(qualityLevel), float(minDistance), int(blockSize), ptr(out), cap, C.byref(n)),
            "km_good_features")
    if n.value == 0:
        return None
    return out[:n.value].reshape(-1, 1, 2).copy()


def pyr_down(img, ctx: Context | None = None):
    c = _ctx(ctx)
    a = np.ascontiguousarray(img, np.uint8)
    out = np.empty(((a.shape[0] + 1) // 2, (a.shape[1] + 1) // 2), np.uint8)
    c.check(c.lib.km_pyrdown_u8(c.handle, ptr(a), a.shape[0], a.shape[1], ptr(out)), "km_pyrdown_u8")
    return out


def calc_optical_flow_pyr_lk(prev_img, next_img, prev_pts, winSize=(25, 25), maxLevel=1, maxCount=30, epsilon=0.03,
                             ctx: Context | None = None):
    """cv2.calcOpticalFlowPyrLK(prev, next, pts, None, winSize=, maxLevel=, criteria=(EPS|COUNT, maxCount, epsilon))
    (klt.py:128-140) -> next points (N,1,2) float32.  status / err are not produced: KARIOS overwrites
    status and never uses err (klt.py:142-153)."""
    c = _ctx(ctx)
    a = np.ascontiguousarray(prev_img, np.uint8)
    b = np.ascontiguousarray(next_img, np.uint8)
    if a.shape != b.shape or a.ndim != 2:
        raise KariosHipError("calcOpticalFlowPyrLK: image shape mismatch")
    if winSize[0] != winSize[1]:
        raise KariosHipError("calcOpticalFlowPyrLK: only square windows (KARIOS uses (w, w))")
    p = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    out = np.empty_like(p)
    c.check(c.lib.km_pyrlk(c.handle, ptr(a), ptr(b), a.shape[0], a.shape[1], ptr(p), p.shape[0], int(winSize[0]),
                           int(maxLevel), int(maxCount), float(epsilon), ptr(out)), "km_pyrlk")
    return out.reshape(-1, 1, 2)


def lk_oscillation_probe(quads, ctx: Context | None = None):
    """Test hook: the LK kernels' oscillation stop (OpenCV: float32 |delta + prevDelta| against the double literal 0.01,
    klt.py:134-140) evaluated on the device for (n, 4) float32 rows (ddx, pdx, ddy, pdy) -> bool array."""
    c = _ctx(ctx)
    q = np.ascontiguousarray(quads, np.float32).reshape(-1, 4)
    out = np.zeros(q.shape[0], np.uint8)
    c.check(c.lib.km_lk_oscillation_probe(c.handle, ptr(q), q.shape[0], ptr(out)), "km_lk_oscillation_probe")
    return out.astype(bool)


def make_params(conf, mon_ksize=1, ref_ksize=1, invert_mon=False) -> KltParams:
    """KLTConfiguration duck type (core/configuration.py:36-50) -> km_klt_params with the fixed
    LK criteria of klt.py:128-132."""
    p = KltParams()
    p.max_corners = int(conf.maxCorners)
    p.block_size = int(conf.blocksize)
    p.win_size = int(conf.matching_winsize)
    p.max_level = 1
    p.max_count = 30
    p.ksize_mon = int(mon_ksize)
    p.ksize_ref = int(ref_ksize)
    p.invert_mon = int(bool(invert_mon))
    p.quality_level = float(conf.qualityLevel)
    p.min_distance = float(conf.minDistance)
    p.epsilon = 0.03
    return p


def _track_outputs(cap):
    return (np.empty((cap, 2), np.float32), np.empty((cap, 2), np.float32), np.empty((cap, 2), np.float32))


def klt_track(ref_lap, mon_lap, mask, conf, p0=None, ctx: Context | None = None):
    """GFTT on ref (unless p0) + LK ref->mon + LK mon->ref (klt.py:103-140).
    -> (p0, p1, p0r) each (N,1,2) float32, or None when no feature was extracted."""
    c = _ctx(ctx)
    a = np.ascontiguousarray(ref_lap, np.uint8)
    b = np.ascontiguousarray(mon_lap, np.uint8)
    if a.shape != b.shape or a.ndim != 2:
        raise KariosHipError("klt_track: image shape mismatch")
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    prm = make_params(conf)
    p0_in, n_p0 = None, 0
    if p0 is not None:
        p0_in = np.ascontiguousarray(p0, np.float32).reshape(-1, 2)
        n_p0 = p0_in.shape[0]
        cap = max(n_p0, 1)
    else:
        cap = prm.max_corners if prm.max_corners > 0 else max(1, a.size // 4)
    o0, o1, o2 = _track_outputs(cap)
    n = C.c_int()
    c.check(c.lib.km_klt_track(c.handle, ptr(a), ptr(b), ptr(m), a.shape[0], a.shape[1], C.byref(prm), ptr(p0_in), n_p0,
                               ptr(o0), ptr(o1), ptr(o2), cap, C.byref(n)), "km_klt_track")
    if n.value == 0:
        return None
    k = n.value
    return o0[:k].reshape(-1, 1, 2), o1[:k].reshape(-1, 1, 2), o2[:k].reshape(-1, 1, 2)


def klt_tile(ref_box, mon_box, conf, mask_box=None, nodata_ref=None, nodata_mon=None, mon_ksize=1, ref_ksize=1,
             invert_mon=False, ctx: Context | None = None):
    """Numeric core of KLT._match_tile for one box (klt.py:252-301, 407-436), fused on the GPU:
    uint8 stretch -> Laplacians -> (auto) mask -> GFTT -> LK fwd/bwd.
    -> ("ok", (p0, p1, p0r)) | ("no_valid_pixels", None) | ("no_features", None)."""
    c = _ctx(ctx)
    r, m = as_image(ref_box), as_image(mon_box)
    if r.shape != m.shape or r.dtype != m.dtype:
        raise KariosHipError("klt_tile: ref/mon shape or dtype mismatch")
    mk = None
    if mask_box is not None:
        mk = np.ascontiguousarray(mask_box, np.uint8)
        if mk.shape != r.shape:
            raise KariosHipError("klt_tile: mask shape mismatch")
    prm = make_params(conf, mon_ksize, ref_ksize, invert_mon)
    cap = prm.max_corners if prm.max_corners > 0 else max(1, r.size // 4)
    o0, o1, o2 = _track_outputs(cap)
    n = C.c_int()
    nr = C.byref(C.c_double(float(nodata_ref))) if nodata_ref is not None else None
    nm = C.byref(C.c_double(float(nodata_mon))) if nodata_mon is not None else None
    c.check(c.lib.km_klt_tile(c.handle, ptr(r), ptr(m), dtype_code(r), r.shape[0], r.shape[1], row_stride(r), row_stride(m),
                              ptr(mk), nr, nm, C.byref(prm), ptr(o0), ptr(o1), ptr(o2), cap, C.byref(n)), "km_klt_tile")
    if n.value == 0:
        return ("no_valid_pixels" if c.stats().valid_pixels == 0 else "no_features"), None
    k = n.value
    return "ok", (o0[:k].reshape(-1, 1, 2), o1[:k].reshape(-1, 1, 2), o2[:k].reshape(-1, 1, 2))


def tile_prefilter(ref_box, mon_box, nodata_ref=None, nodata_mon=None, ref_ksize=1, mon_ksize=1, invert_mon=False,
                   with_mask=True, ctx: Context | None = None):
    """Pre-filter of KLT._match_tile (klt.py:268-273, 407-436) in the fused kernel of the tile path:
    -> (laplacian(uint8(ref)), laplacian(uint8(mon) or its inverse), auto mask | None, valid pixel count | None)."""
    c = _ctx(ctx)
    r, m = as_image(ref_box), as_image(mon_box)
    if r.shape != m.shape or r.dtype != m.dtype:
        raise KariosHipError("tile_prefilter: ref/mon shape or dtype mismatch")
    lr, lm = np.empty(r.shape, np.uint8), np.empty(r.shape, np.uint8)
    mk = np.empty(r.shape, np.uint8) if with_mask else None
    nv = C.c_int64()
    nr = C.byref(C.c_double(float(nodata_ref))) if nodata_ref is not None else None
    nm = C.byref(C.c_double(float(nodata_mon))) if nodata_mon is not None else None
    c.check(c.lib.km_tile_prefilter(c.handle, ptr(r), ptr(m), dtype_code(r), r.shape[0], r.shape[1], row_stride(r), row_stride(m),
                                    nr, nm, int(ref_ksize), int(mon_ksize), int(bool(invert_mon)), ptr(lr), ptr(lm), ptr(mk),
                                    C.byref(nv)), "km_tile_prefilter")
    return lr, lm, mk, (int(nv.value) if with_mask else None)


def zncc_batch(ref, mon, x0, y0, dx, dy, ctx: Context | None = None):
    """ZNCCService._compute_zncc for every keypoint (zncc_service.py:186-238) -> float64[n]."""
    c = _ctx(ctx)
    r, m = as_image(ref), as_image(mon)
    if r.dtype != m.dtype:
        raise KariosHipError("zncc_batch: dtype mismatch")
    x0, y0, dx, dy = (np.ascontiguousarray(v, np.float32) for v in (x0, y0, dx, dy))
    n = len(x0)
    out = np.empty(n, np.float64)
    if n == 0:
        return out
    c.check(c.lib.km_zncc_batch(c.handle, ptr(r), ptr(m), dtype_code(r), r.shape[0], r.shape[1], m.shape[0], m.shape[1],
                                row_stride(r), row_stride(m), ptr(x0), ptr(y0), ptr(dx), ptr(dy), n, ptr(out)),
            "km_zncc_batch")
    return out


def zncc_windows(img1, img2, u1, v1, u2, v2, half_size: int, ctx: Context | None = None):
    """`_zncc2(img1, img2, u1, v1, u2, v2, n)` (zncc_service.py:45-126) for arrays of window centres (rows u, columns v),
    any half-size n >= 0 and any two numeric pixel types (other types are read as float64).
    -> (float64 values, bool mask of the windows that leave their image: the reference raises IndexError for those)."""
    c = _ctx(ctx)
    a, b = np.asarray(img1), np.asarray(img2)
    if a.dtype not in _lib._DTYPES and a.dtype not in _lib._ANY_DTYPES:
        a = a.astype(np.float64)
    if b.dtype not in _lib._DTYPES and b.dtype not in _lib._ANY_DTYPES:
        b = b.astype(np.float64)
    a, b = as_image(a), as_image(b)
    uv = np.ascontiguousarray(np.stack([np.asarray(v).ravel() for v in (u1, v1, u2, v2)]), np.int32)
    count = uv.shape[1]
    out, outside = np.empty(count, np.float64), np.zeros(count, np.uint8)
    if count:
        c.check(c.lib.km_zncc_windows(c.handle, ptr(a), ptr(b), _lib.any_dtype_code(a), _lib.any_dtype_code(b), a.shape[0], a.shape[1],
                                      b.shape[0], b.shape[1], row_stride(a), row_stride(b), ptr(uv), int(half_size), count, ptr(out),
                                      ptr(outside)), "km_zncc_windows")
    return out, outside.astype(bool)


def mi_batch(ref, mon, x0, y0, dx, dy, ctx: Context | None = None):
    """Per-keypoint mutual-information scores on the 57x57 chips -> (studholme, nmi) float64 arrays:
    `MutualInfoService._compute_mutual_info` (mutual_info_service.py:99-130, (H(X)+H(Y))/H(X,Y) in [1,2]) and
    `ZNCCService._compute_mi` (zncc_service.py:260-287, 2*MI/(H(X)+H(Y)) in [0,1])."""
    c = _ctx(ctx)
    r, m = as_image(ref), as_image(mon)
    if r.dtype != m.dtype:
        raise KariosHipError("mi_batch: dtype mismatch")
    x0, y0, dx, dy = (np.ascontiguousarray(v, np.float32) for v in (x0, y0, dx, dy))
    n = len(x0)
    a, b = np.empty(n, np.float64), np.empty(n, np.float64)
    if n == 0:
        return a, b
    c.check(c.lib.km_mi_batch(c.handle, ptr(r), ptr(m), dtype_code(r), r.shape[0], r.shape[1], m.shape[0], m.shape[1],
                              row_stride(r), row_stride(m), ptr(x0), ptr(y0), ptr(dx), ptr(dy), n, ptr(a), ptr(b)), "km_mi_batch")
    return a, b


def phase_cross_correlation(reference_image, moving_image, ctx: Context | None = None):
    """skimage.registration.phase_cross_correlation(reference_image, moving_image)[0] with the 0.24
    defaults (large_offset.py:39) -> array([row, col]) float64 holding integers."""
    c = _ctx(ctx)
    a, b = as_image(reference_image), as_image(moving_image)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise KariosHipError("phase_cross_correlation: images must have the same shape and dtype")
    out = (C.c_double * 2)()
    c.check(c.lib.km_phase_shift(c.handle, ptr(a), ptr(b), dtype_code(a), a.shape[0], a.shape[1], row_stride(a), row_stride(b),
                                 out), "km_phase_shift")
    return np.array([out[0], out[1]], np.float64)


def shift_image(img, y_off=0, x_off=0, ctx: Context | None = None):
    """shift_image (core/image.py:70-101): integer shift, zero fill, dtype/shape preserved."""
    c = _ctx(ctx)
    a = np.asarray(img)
    if a.ndim != 2 or a.itemsize not in (1, 2, 4, 8):
        raise KariosHipError("shift_image: expected a 2-D array of 1/2/4/8-byte elements")
    a = as_image(a)
    y_off, x_off = int(round(y_off)), int(round(x_off))
    out = np.empty(a.shape, a.dtype)
    c.check(c.lib.km_shift_image(c.handle, ptr(a), a.itemsize, a.shape[0], a.shape[1], row_stride(a), y_off, x_off, ptr(out)),
            "km_shift_image")
    return out


__all__ = ["Context", "KariosHipError", "to_uint8", "auto_mask", "laplacian_u8", "min_eigen", "good_features_to_track",
           "pyr_down", "calc_optical_flow_pyr_lk", "klt_track", "klt_tile", "zncc_batch", "zncc_windows", "mi_batch", "phase_cross_correlation",
           "shift_image", "make_params", "_lib"]
