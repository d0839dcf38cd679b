"""Deterministic numpy-only synthetic image pairs.

There is no GDAL and no Sentinel-2 data on the GPU box; every bench / parity
workload is generated here.  The monitored image is the reference content moved
by (+sx, +sy) pixels, so a KARIOS run must report dx ~ +sx, dy ~ +sy
(dx = x_mon - x_ref, reference klt.py:167).
"""
from __future__ import annotations

import numpy as np

PAD = 64


def _box_blur(a: np.ndarray, radius: int, passes: int = 3) -> np.ndarray:
    """`passes` separable running-mean passes (cumsum based, edge-replicated)."""
    k = 2 * radius + 1
    for _ in range(passes):
        for axis in (0, 1):
            pad = [(0, 0), (0, 0)]
            pad[axis] = (radius + 1, radius)
            c = np.cumsum(np.pad(a, pad, mode="edge"), axis=axis, dtype=np.float64)
            n = a.shape[axis]
            hi = np.take(c, np.arange(k, k + n), axis=axis)
            lo = np.take(c, np.arange(0, n), axis=axis)
            a = ((hi - lo) / k).astype(np.float32)
    return a


def _standardise(a: np.ndarray) -> np.ndarray:
    a = a - a.mean(dtype=np.float64)
    return (a / a.std(dtype=np.float64)).astype(np.float32)


def make_base(H: int, W: int, seed: int = 20260101) -> np.ndarray:
    """Multi-scale texture field of shape (H+2*PAD, W+2*PAD), float32."""
    rng = np.random.default_rng(seed)
    n = rng.standard_normal((H + 2 * PAD, W + 2 * PAD), dtype=np.float32)
    base = np.full(n.shape, 3000.0, np.float32)
    for radius, gain in ((1, 600.0), (4, 900.0), (16, 1200.0)):
        base += gain * _standardise(_box_blur(n, radius))
    return base


def _quantise(a: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(a), 1, 16000).astype(np.uint16)


def _sample_shifted(base: np.ndarray, H: int, W: int, sx: float, sy: float) -> np.ndarray:
    """Bilinear sample of base at (PAD + y - sy, PAD + x - sx)."""
    fy, fx = PAD - sy, PAD - sx
    iy, ix = int(np.floor(fy)), int(np.floor(fx))
    ay, ax = np.float32(fy - iy), np.float32(fx - ix)
    if iy < 0 or ix < 0 or iy + H + 1 > base.shape[0] or ix + W + 1 > base.shape[1]:
        raise ValueError("shift exceeds synthetic padding")
    b00 = base[iy:iy + H, ix:ix + W]
    b01 = base[iy:iy + H, ix + 1:ix + W + 1]
    b10 = base[iy + 1:iy + H + 1, ix:ix + W]
    b11 = base[iy + 1:iy + H + 1, ix + 1:ix + W + 1]
    return ((1 - ay) * ((1 - ax) * b00 + ax * b01) + ay * ((1 - ax) * b10 + ax * b11)).astype(np.float32)


def make_pair(H: int, W: int, sx: float = 0.5, sy: float = 0.0, seed: int = 20260101,
              noise_sigma: float = 15.0, nodata_wedge: bool = False):
    """-> (mon uint16, ref uint16).  mon = ref content moved by (+sx, +sy) + noise."""
    base = make_base(H, W, seed)
    ref = _quantise(base[PAD:PAD + H, PAD:PAD + W])
    mon_f = _sample_shifted(base, H, W, sx, sy)
    if noise_sigma > 0:
        mon_f = mon_f + np.random.default_rng(seed + 1).normal(0, noise_sigma, (H, W)).astype(np.float32)
    mon = _quantise(mon_f)
    if nodata_wedge:
        yy, xx = np.ogrid[:H, :W]
        wedge = (xx + yy) < 0.45 * W
        ref[wedge] = 0
        mon[wedge] = 0
    return mon, ref


def make_cross_sensor_pair(H: int, W: int, sx: float = 0.4, sy: float = -0.3, seed: int = 20260101):
    """BASELINE config 5 stand-in: mon is 3x3 block-averaged / nearest-upsampled,
    gamma 0.8 radiometry; plus a user mask zeroing ~20 % in random rectangles."""
    base = make_base(H, W, seed)
    ref = _quantise(base[PAD:PAD + H, PAD:PAD + W])
    m = _sample_shifted(base, H, W, sx, sy)
    h3, w3 = (H // 3) * 3, (W // 3) * 3
    blk = m[:h3, :w3].reshape(h3 // 3, 3, w3 // 3, 3).mean(axis=(1, 3))
    m[:h3, :w3] = np.repeat(np.repeat(blk, 3, axis=0), 3, axis=1)
    m = 16000.0 * np.power(np.clip(m, 1, 16000) / 16000.0, 0.8)
    mon = _quantise(m)
    rng = np.random.default_rng(seed + 2)
    mask = np.ones((H, W), np.uint8)
    target = 0.2 * H * W
    while (mask == 0).sum() < target:
        h, w = rng.integers(H // 16 + 1, H // 4 + 2), rng.integers(W // 16 + 1, W // 4 + 2)
        y, x = rng.integers(0, H), rng.integers(0, W)
        mask[y:y + h, x:x + w] = 0
    return mon, ref, mask
