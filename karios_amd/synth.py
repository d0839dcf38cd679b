"""Deterministic numpy-only synthetic image pairs.

There is no GDAL and no Sentinel-2 data on the GPU box; every bench / parity
workload is generated here.  The monitored image is the reference content moved
by (+sx, +sy) pixels, so a KARIOS run must report dx ~ +sx, dy ~ +sy
(dx = x_mon - x_ref, reference klt.py:167).
"""
from __future__ import annotations

import os

import numpy as np

PAD = 64


def _box_blur(a: np.ndarray, radius: int, passes: int = 3) -> np.ndarray:
    """`passes` separable running-mean passes (edge-replicated), rows split over threads."""
    from concurrent.futures import ThreadPoolExecutor

    from scipy import ndimage as ndi

    k = 2 * radius + 1
    nthr = min(8, os.cpu_count() or 1)

    def run(axis, src):
        dst = np.empty_like(src)
        # filter along `axis`, chunk along the other axis so the chunks are independent
        chunks = np.array_split(np.arange(src.shape[1 - axis]), nthr)

        def work(idx):
            sl = (slice(None), slice(idx[0], idx[-1] + 1)) if axis == 0 else (slice(idx[0], idx[-1] + 1), slice(None))
            ndi.uniform_filter1d(src[sl], k, axis=axis, mode="nearest", output=dst[sl])

        with ThreadPoolExecutor(nthr) as ex:
            list(ex.map(work, [c for c in chunks if len(c)]))
        return dst

    for _ in range(passes):
        a = run(0, a)
        a = run(1, a)
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


def _base_torch(H: int, W: int, seed: int, device):
    """Device version of `make_base`: float32 texture field of shape (H + 2 PAD, W + 2 PAD)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = torch.randn((H + 2 * PAD, W + 2 * PAD), generator=g, device=device, dtype=torch.float32)

    def blur(a, radius):
        k = 2 * radius + 1
        for _ in range(3):
            for axis in (0, 1):
                idx_lo = torch.arange(-radius - 1, a.shape[axis] - radius - 1, device=device).clamp_(0, a.shape[axis] - 1)
                idx_hi = torch.arange(radius, a.shape[axis] + radius, device=device).clamp_(0, a.shape[axis] - 1)
                # edge-replicated running mean via cumulative sums (float64)
                pad_lo = torch.arange(-radius - 1, 0, device=device).clamp_(min=0)
                c = torch.cumsum(a.double(), dim=axis)
                hi = c.index_select(axis, idx_hi)
                lo = c.index_select(axis, idx_lo)
                first = a.double().index_select(axis, torch.zeros(1, dtype=torch.long, device=device))
                last = a.double().index_select(axis, torch.full((1,), a.shape[axis] - 1, dtype=torch.long, device=device))
                pos = torch.arange(a.shape[axis], device=device)
                n_before = (radius - pos).clamp_(min=0).double()           # replicated first samples
                n_after = (pos + radius - (a.shape[axis] - 1)).clamp_(min=0).double()  # replicated last samples
                shape = [1, 1]
                shape[axis] = -1
                lo = torch.where((pos - radius - 1 < 0).view(shape), torch.zeros_like(lo), lo)
                s = hi - lo + n_before.view(shape) * first + n_after.view(shape) * last
                a = (s / k).float()
                del c, hi, lo, s, pad_lo
        return a

    base = torch.full_like(n, 3000.0)
    for radius, gain in ((1, 600.0), (4, 900.0), (16, 1200.0)):
        t = blur(n, radius)
        t = (t - t.mean()) / t.std()
        base += gain * t
        del t
    del n
    return base


def _quant_torch(a):
    import torch
    return torch.clamp(torch.round(a), 1, 16000).to(torch.int32).to(torch.int16)  # values < 32768: same bits as uint16


def _sample_shifted_torch(base, H: int, W: int, sx: float, sy: float):
    fy, fx = PAD - sy, PAD - sx
    iy, ix = int(np.floor(fy)), int(np.floor(fx))
    ay, ax = float(fy - iy), float(fx - ix)
    b00 = base[iy:iy + H, ix:ix + W]
    b01 = base[iy:iy + H, ix + 1:ix + W + 1]
    b10 = base[iy + 1:iy + H + 1, ix:ix + W]
    b11 = base[iy + 1:iy + H + 1, ix + 1:ix + W + 1]
    return (1 - ay) * ((1 - ax) * b00 + ax * b01) + ay * ((1 - ax) * b10 + ax * b11)


def make_pair_torch(H: int, W: int, sx: float = 0.5, sy: float = 0.25, seed: int = 20260101, noise_sigma: float = 15.0,
                    device="cuda"):
    """GPU version of `make_pair` for full-size bench inputs (same construction -- three box-blurred
    noise scales, bilinear sub-pixel shift, additive noise, uint16 quantisation -- generated by torch on
    the device; NOT bit-identical to the numpy generator, which the parity tests use).
    -> (mon, ref) int16-storage torch tensors holding uint16 bit patterns, dense (H, W)."""
    import torch

    base = _base_torch(H, W, seed, device)
    ref = _quant_torch(base[PAD:PAD + H, PAD:PAD + W]).contiguous()
    mon_f = _sample_shifted_torch(base, H, W, sx, sy)
    if noise_sigma > 0:
        g2 = torch.Generator(device=device)
        g2.manual_seed(seed + 1)
        mon_f = mon_f + noise_sigma * torch.randn((H, W), generator=g2, device=device, dtype=torch.float32)
    mon = _quant_torch(mon_f).contiguous()
    return mon, ref


def make_cross_sensor_pair_torch(H: int, W: int, sx: float = 0.4, sy: float = -0.3, seed: int = 20260101, device="cuda"):
    """GPU version of `make_cross_sensor_pair` (BASELINE config 5 stand-in) for full-size bench inputs: the monitored image is the
    shifted texture 3x3 block-averaged and nearest-upsampled x3 (a 30 m sensor on the 10 m grid) with gamma-0.8 radiometry; the
    user mask zeroes ~20 % of the pixels in seeded random rectangles.  -> (mon, ref, mask): int16-storage uint16 bit patterns, uint8."""
    import torch

    base = _base_torch(H, W, seed, device)
    ref = _quant_torch(base[PAD:PAD + H, PAD:PAD + W]).contiguous()
    m = _sample_shifted_torch(base, H, W, sx, sy).contiguous()
    del base
    h3, w3 = (H // 3) * 3, (W // 3) * 3
    blk = m[:h3, :w3].reshape(h3 // 3, 3, w3 // 3, 3).mean(dim=(1, 3))
    m[:h3, :w3] = blk.repeat_interleave(3, dim=0).repeat_interleave(3, dim=1)
    m = 16000.0 * torch.pow(torch.clamp(m, 1, 16000) / 16000.0, 0.8)
    mon = _quant_torch(m).contiguous()
    rng = np.random.default_rng(seed + 2)
    mask = torch.ones((H, W), dtype=torch.uint8, device=device)
    target = 0.2 * H * W
    while int((mask == 0).sum().item()) < target:
        h, w = rng.integers(H // 16 + 1, H // 4 + 2), rng.integers(W // 16 + 1, W // 4 + 2)
        y, x = rng.integers(0, H), rng.integers(0, W)
        mask[y:y + h, x:x + w] = 0
    return mon, ref, mask


# ---------------------------------------------------------------------------------------------------- non-best-case workloads
# The pairs above are the friendliest content there is: every corner survives the forward-backward test and LK stops after two
# iterations.  The reference's one real run keeps 37 448 of <= 80 000 corners (tests/end_to_end/ref_data/test_full/KLT_matcher_*.csv,
# klt.py:134-144), and SURVEY App. A.1 warns that k = 7 Laplacians of real scenes are near-binary.  The two generators below produce
# that kind of content; torch on any device (the CPU build serves the tests and tools/investigations/calibrate_workloads.py).
def _smooth_field_torch(H: int, W: int, cell: int, seed: int, device):
    """Smooth random field in [0, 1], correlation length ~`cell` px: a coarse uniform grid, bicubically enlarged."""
    import torch

    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    gh, gw = H // cell + 3, W // cell + 3
    coarse = torch.rand((1, 1, gh, gw), generator=g, dtype=torch.float32).to(device)
    big = torch.nn.functional.interpolate(coarse, size=(gh * cell, gw * cell), mode="bicubic", align_corners=False)
    return big[0, 0, cell:cell + H, cell:cell + W].clamp_(0.0, 1.0).contiguous()


def make_hard_pair_torch(H: int, W: int, sx: float = 0.5, sy: float = 0.25, seed: int = 20260101, mix: float = 0.555,
                         noise_sigma: float = 200.0, warp: float = 0.6, device="cuda"):
    """`hard_content`: the monitored image is decorrelated from the reference until about half of the tracks fail the 0.1-px round
    trip: (1) a smooth sub-pixel warp - the content moves by (sx + u warp, sy + v warp) with u, v smooth fields in [0, 1] of ~400 px
    correlation length (a blend of the four corner shifts with bilinear weights); (2) `mix` of an independent texture of the same
    spectrum; (3) strong additive noise.  Defaults calibrated on the MI355X at 10980^2 (tools/investigations/hard_probe.py): 48 % of the 20 000
    corners survive (mix 0.45 / sigma 120: 79 %, 0.57 / 250: 42 %, 0.6 / 300: 34 %).  -> (mon, ref) int16-storage uint16 bit patterns like `make_pair_torch`."""
    import torch

    base = _base_torch(H, W, seed, device)
    ref = _quant_torch(base[PAD:PAD + H, PAD:PAD + W]).contiguous()
    u = _smooth_field_torch(H, W, 400, seed + 3, device)
    v = _smooth_field_torch(H, W, 400, seed + 4, device)
    mon_f = torch.zeros((H, W), dtype=torch.float32, device=device)
    for du, dv, wgt in ((0.0, 0.0, (1 - u) * (1 - v)), (warp, 0.0, u * (1 - v)), (0.0, warp, (1 - u) * v), (warp, warp, u * v)):
        mon_f += wgt * _sample_shifted_torch(base, H, W, sx + du, sy + dv)
        del wgt
    del base, u, v
    if mix > 0:
        other = _base_torch(H, W, seed + 5, device)[PAD:PAD + H, PAD:PAD + W]
        mon_f = (1.0 - mix) * mon_f + mix * other
        del other
    if noise_sigma > 0:
        g2 = torch.Generator(device=device)
        g2.manual_seed(seed + 1)
        mon_f = mon_f + noise_sigma * torch.randn((H, W), generator=g2, device=device, dtype=torch.float32)
    return _quant_torch(mon_f).contiguous(), ref


def make_tie_heavy_pair_torch(H: int, W: int, sx: float = 0.5, sy: float = 0.25, seed: int = 20260101, levels: int = 6,
                              noise_sigma: float = 15.0, period: int = 96, device="cuda"):
    """`tie_heavy`: both rasters quantised to `levels` grey levels, so that their uint8 stretch is a staircase of ~255 / levels and
    the k = 7 Laplacian (gain 16 x 64) saturates at nearly every edge - a near-binary image; and the texture REPEATS
    with `period` px (fields, roofs, a regular street grid: every corner of the motif recurs with exactly the same minimum eigenvalue),
    so the candidate list is made of exact ties that only the raster index orders (App. A.2 step 6), flat maxima give several
    candidates per 3x3 neighbourhood, and the value bins of the synchronisation-free selection are few and full.  Binary Laplacians
    alone do not tie: a 15x15 sum of Sobel products of a 0/255 image still takes ~as many values as there are corners (measured:
    694 distinct values among 695 corners)."""
    import torch

    base = _base_torch(min(H, period) if period else H, min(W, period) if period else W, seed, device)
    if period:
        cell = base[PAD:PAD + min(H, period), PAD:PAD + min(W, period)]
        ry, rx = -(-(H + 2 * PAD) // cell.shape[0]), -(-(W + 2 * PAD) // cell.shape[1])
        base = cell.repeat(ry, rx)[:H + 2 * PAD, :W + 2 * PAD].contiguous()
    lo, hi = 3000.0 - 2200.0, 3000.0 + 2200.0
    step = (hi - lo) / levels

    def stair(a):
        q = torch.floor((a - lo) / step).clamp_(0, levels - 1)
        return _quant_torch(lo + (q + 0.5) * step).contiguous()

    ref = stair(base[PAD:PAD + H, PAD:PAD + W])
    mon_f = _sample_shifted_torch(base, H, W, sx, sy)
    del base
    if noise_sigma > 0:
        g2 = torch.Generator(device=device)
        g2.manual_seed(seed + 1)
        mon_f = mon_f + noise_sigma * torch.randn((H, W), generator=g2, device=device, dtype=torch.float32)
    return stair(mon_f), ref
