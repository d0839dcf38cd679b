"""CPU oracle for the KARIOS matching hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package ``karios_amd`` never does.

Thin ctypes binding over ``libkarios_oracle.so`` (karios_oracle.c, the dense
restatement) plus numpy restatements of the reference's numpy glue.  Every
function cites the reference file:line (relative to /root/reference) it follows.

Parity status: see the header of karios_oracle.c ("parity unpinned" for the
OpenCV-defined arithmetic; numpy-defined pieces are pinned by tests/golden).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KARIOS_ORACLE_SO: another build of the same sources, e.g. libkarios_oracle_asan.so (`make -C oracle asan`, tests/test_host_asan.py)
_SO = os.environ.get("KARIOS_ORACLE_SO") or os.path.join(_HERE, "libkarios_oracle.so")

_DT = {np.dtype("uint8"): 0, np.dtype("uint16"): 1, np.dtype("int16"): 2,
       np.dtype("float32"): 3, np.dtype("float64"): 4}


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    if os.environ.get("KARIOS_ORACLE_SO"):
        return _SO                      # (built by whoever chose it)
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("karios_oracle.c", "karios_oracle_cvlit.c", "Makefile"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


_lib = None


def _cgroup_cpu_limit() -> float | None:
    """CPU quota of this process's cgroup (and its ancestors), in CPUs: cgroup v2 `cpu.max`, cgroup v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`.  None when no quota is visible."""
    best = None

    def note(quota, period):
        nonlocal best
        if quota > 0 and period > 0:
            v = quota / period
            best = v if best is None else min(best, v)

    rel_v2, rel_v1 = "", ""
    try:
        for line in open("/proc/self/cgroup"):
            _, ctrl, path = line.rstrip("\n").split(":", 2)
            if ctrl == "":
                rel_v2 = path
            elif "cpu" in ctrl.split(","):
                rel_v1 = path
    except (OSError, ValueError):
        pass
    # v2: the process's own directory and every ancestor up to the mount point (inside a cgroup namespace the own
    # directory IS the mount point)
    parts = [p for p in rel_v2.split("/") if p]
    for k in range(len(parts), -1, -1):
        try:
            quota, period = open(os.path.join("/sys/fs/cgroup", *parts[:k], "cpu.max")).read().split()[:2]
            if quota != "max":
                note(int(quota), int(period))
        except (OSError, ValueError):
            pass
    parts = [p for p in rel_v1.split("/") if p]
    for root in ("/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
        for k in range(len(parts), -1, -1):
            try:
                d = os.path.join(root, *parts[:k])
                note(int(open(os.path.join(d, "cpu.cfs_quota_us")).read()), int(open(os.path.join(d, "cpu.cfs_period_us")).read()))
            except (OSError, ValueError):
                pass
    return best


def usable_cpus() -> int:
    """CPUs this process may actually use: min(logical CPUs, affinity mask, cgroup quota).  A container that shows
    256 logical CPUs under a 16-CPU quota runs an all-cores OpenMP team slower than a 16-thread one (throttled spinning)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    q = _cgroup_cpu_limit()
    if q is not None:
        n = min(n, max(1, int(q + 0.999)))
    return max(1, n)


# The parity tests call the oracle thousands of times on small images: the team is capped (a quota this module cannot see -
# a parent cgroup outside the namespace, a hypervisor - would otherwise make every call spin against the throttle; the
# round-1 GPU suite took 855 s on the driver's box and 38 s on the builder's for that reason) and the OpenMP runtime is
# told to block instead of spinning.  bench.py's cpu_baseline sets its team explicitly.
DEFAULT_TEAM_CAP = 16


def team_size() -> int:
    env = os.environ.get("KARIOS_ORACLE_THREADS")
    if env:
        return max(1, int(env))
    return max(1, min(usable_cpus(), DEFAULT_TEAM_CAP))


def lib():
    global _lib
    if _lib is None:
        build()
        # read by libgomp when it is first loaded (a libgomp bundled with torch is a different library instance)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        os.environ.setdefault("GOMP_SPINCOUNT", "0")
        L = C.CDLL(_SO)
        L.ko_auto_mask.restype = C.c_long
        L.ko_set_threads(min(int(L.ko_max_threads()), team_size()))
        _lib = L
    return _lib


def describe_cpu() -> str:
    """One line for test / bench headers: what the oracle sees of the host."""
    q = _cgroup_cpu_limit()
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = -1
    return (f"oracle: logical CPUs {os.cpu_count()}, affinity {aff}, cgroup quota {q if q is not None else 'none'}, usable {usable_cpus()}, "
            f"team {min(max_threads(), team_size())} (libgomp max {max_threads()}), OMP_WAIT_POLICY={os.environ.get('OMP_WAIT_POLICY')}, "
            f"GOMP_SPINCOUNT={os.environ.get('GOMP_SPINCOUNT')}")


def max_threads() -> int:
    return int(lib().ko_max_threads())


def set_threads(n: int) -> None:
    lib().ko_set_threads(int(n))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _check_img(a):
    a = np.asarray(a)
    if a.ndim != 2 or a.dtype not in _DT:
        raise TypeError(f"unsupported image {a.dtype} ndim={a.ndim}")
    if a.strides[1] != a.itemsize:
        a = np.ascontiguousarray(a)
    return a, _DT[a.dtype], a.strides[0] // a.itemsize


# ----------------------------------------------------------------- numpy glue
def minmax(arr):
    """np.nanmin/np.nanmax as floats (klt.py:46)."""
    a, dt, st = _check_img(arr)
    mn, mx = C.c_double(), C.c_double()
    lib().ko_minmax(_p(a), dt, a.shape[0], a.shape[1], C.c_ssize_t(st), C.byref(mn), C.byref(mx))
    return mn.value, mx.value


def to_uint8(arr, invert: bool = False):
    """_to_uint8 (klt.py:42-49), optional 255-x (klt.py:419)."""
    a, dt, st = _check_img(arr)
    mn, mx = (0.0, 0.0) if dt == 0 else minmax(a)
    out = np.empty(a.shape, np.uint8)
    lib().ko_to_uint8(_p(a), dt, a.shape[0], a.shape[1], C.c_ssize_t(st),
                      C.c_double(mn), C.c_double(mx), int(bool(invert)), _p(out))
    return out


def auto_mask(mon, ref, nodata_mon=None, nodata_ref=None):
    """Automatic validity mask (klt.py:268-273) -> (uint8 mask, valid count)."""
    m, dt, sm = _check_img(mon)
    r, dt2, sr = _check_img(ref)
    if dt != dt2 or m.shape != r.shape:
        raise TypeError("mon/ref dtype or shape mismatch")
    mask = np.empty(m.shape, np.uint8)
    nm = C.byref(C.c_double(float(nodata_mon))) if nodata_mon is not None else None
    nr = C.byref(C.c_double(float(nodata_ref))) if nodata_ref is not None else None
    valid = lib().ko_auto_mask(_p(m), _p(r), dt, m.shape[0], m.shape[1], C.c_ssize_t(sm),
                               C.c_ssize_t(sr), nm, nr, _p(mask))
    return mask, int(valid)


def sobel_kernel(ksize: int, order: int):
    out = np.zeros(ksize, np.int32)
    if lib().ko_sobel_kernel(ksize, order, _p(out)) != 0:
        raise ValueError("bad ksize")
    return out


def laplacian_u8(img, ksize: int):
    """cv2.Laplacian(u8, cv2.CV_8U, ksize=k) (klt.py:433-434)."""
    a = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(a)
    if lib().ko_laplacian_u8(_p(a), a.shape[0], a.shape[1], int(ksize), _p(out)) != 0:
        raise ValueError(f"bad Laplacian ksize {ksize}")
    return out


def min_eigen(img, block: int):
    """cornerMinEigenVal map inside goodFeaturesToTrack."""
    a = np.ascontiguousarray(img, np.uint8)
    out = np.empty(a.shape, np.float32)
    if lib().ko_min_eigen(_p(a), a.shape[0], a.shape[1], int(block), _p(out)) != 0:
        raise ValueError("min_eigen failed")
    return out


def min_eigen_cv(img, block: int, fma: bool = False):
    """cornerMinEigenVal the way OpenCV rounds it (float32 Sobel with the scale in the smoothing taps, float32 products,
    double box sums cast to float32): karios_oracle_cvlit.c.  NOT the definition the kernels follow - the second opinion that
    tools/investigations/oracle_sensitivity.py compares against."""
    a = np.ascontiguousarray(img, np.uint8)
    out = np.empty(a.shape, np.float32)
    if lib().kl_min_eigen_cv(_p(a), a.shape[0], a.shape[1], int(block), _p(out), int(bool(fma))) != 0:
        raise ValueError("min_eigen_cv failed")
    return out


def select_corners(eig, mask=None, maxCorners=0, qualityLevel=0.1, minDistance=10):
    """Steps 4-8 of goodFeaturesToTrack on a given min-eigenvalue map -> (N,1,2) float32 | None."""
    e = np.ascontiguousarray(eig, np.float32)
    H, W = e.shape
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    cap = int(maxCorners) if maxCorners > 0 else H * W
    out = np.empty((max(cap, 1), 2), np.float32)
    n = lib().ko_select_corners(_p(e), None if m is None else _p(m), H, W, int(maxCorners), C.c_double(qualityLevel),
                                C.c_double(minDistance), _p(out), cap, None)
    if n < 0:
        raise RuntimeError(f"ko_select_corners rc={n}")
    return None if n == 0 else out[:n].reshape(n, 1, 2).copy()


def pyr_lk_cv(prev, nxt, pts, win=25, max_level=1, max_count=30, eps=0.03):
    """calcOpticalFlowPyrLK with OpenCV's four-lane float32 accumulation (karios_oracle_cvlit.c): second opinion only."""
    a = np.ascontiguousarray(prev, np.uint8)
    b = np.ascontiguousarray(nxt, np.uint8)
    p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = np.empty_like(p)
    rc = lib().kl_pyrlk_cv(_p(a), _p(b), a.shape[0], a.shape[1], _p(p), p.shape[0], int(win), int(max_level), int(max_count),
                           C.c_double(eps), _p(out))
    if rc != 0:
        raise RuntimeError(f"kl_pyrlk_cv rc={rc}")
    return out.reshape(-1, 1, 2)


def good_features(img, mask=None, maxCorners=0, qualityLevel=0.1, minDistance=10, blockSize=3,
                  return_stats=False):
    """cv2.goodFeaturesToTrack (klt.py:120) -> (N,1,2) float32 or None."""
    a = np.ascontiguousarray(img, np.uint8)
    H, W = a.shape
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    cap = int(maxCorners) if maxCorners > 0 else H * W
    out = np.empty((max(cap, 1), 2), np.float32)
    stats = np.zeros(2, np.float64)
    n = lib().ko_good_features(_p(a), None if m is None else _p(m), H, W, int(maxCorners),
                               C.c_double(qualityLevel), C.c_double(minDistance), int(blockSize),
                               _p(out), cap, _p(stats))
    if n < 0:
        raise RuntimeError(f"ko_good_features rc={n}")
    res = None if n == 0 else out[:n].reshape(n, 1, 2).copy()
    return (res, stats) if return_stats else res


def pyrdown_u8(img):
    a = np.ascontiguousarray(img, np.uint8)
    out = np.empty(((a.shape[0] + 1) // 2, (a.shape[1] + 1) // 2), np.uint8)
    lib().ko_pyrdown_u8(_p(a), a.shape[0], a.shape[1], _p(out))
    return out


def pyr_lk(prev, nxt, pts, win=25, max_level=1, max_count=30, eps=0.03, return_iters=False):
    """cv2.calcOpticalFlowPyrLK(prev, next, pts, None, winSize=(w,w), maxLevel=1,
    criteria=(EPS|COUNT,30,0.03)) (klt.py:128-140) -> next points (N,1,2) f32."""
    a = np.ascontiguousarray(prev, np.uint8)
    b = np.ascontiguousarray(nxt, np.uint8)
    if a.shape != b.shape:
        raise ValueError("shape mismatch")
    p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = p.shape[0]
    out = np.empty_like(p)
    it = np.zeros(2 * max(n, 1), np.int32)      # [level 0 | level 1] iteration counts (diagnostic)
    rc = lib().ko_pyrlk(_p(a), _p(b), a.shape[0], a.shape[1], _p(p), n, int(win), int(max_level),
                        int(max_count), C.c_double(eps), _p(out), _p(it))
    if rc != 0:
        raise RuntimeError(f"ko_pyrlk rc={rc}")
    out = out.reshape(n, 1, 2)
    if return_iters == "levels":
        return out, it[:n], it[max(n, 1):max(n, 1) + n]
    return (out, it[:n]) if return_iters else out


def lk_oscillates(ddx, pdx, ddy, pdy, literal_oracle: bool = False) -> bool:
    """The oscillation stop of calcOpticalFlowPyrLK's iteration: float32 sums and magnitudes against OpenCV's
    DOUBLE literal 0.01.  `literal_oracle`: the second (OpenCV-literal) oracle's copy of the predicate."""
    f = lib().kl_lk_oscillates if literal_oracle else lib().ko_lk_oscillates
    f.argtypes = [C.c_float] * 4
    return bool(f(float(ddx), float(pdx), float(ddy), float(pdy)))


def filter_outliers(x0, y0, x1, y1, score):
    """__filter_outliers (klt.py:52-71)."""
    dx = x1 - x0
    dy = y1 - y0
    while True:
        ind = ((np.abs(dx - dx.mean()) < 3 * dx.std()) & (np.abs(dy - dy.mean()) < 3 * dy.std())
               & (np.abs(dx - dx.mean()) < 20) & (np.abs(dy - dy.mean()) < 20))
        if int(ind.sum()) == len(dx):
            break
        dx, dy, x0, x1, y0, y1, score = dx[ind], dy[ind], x0[ind], x1[ind], y0[ind], y1[ind], score[ind]
    return x0, y0, x1, y1, score


def klt_tracker(ref_data, image_data, mask, conf, p0=None):
    """klt_tracker (klt.py:83-172) -> (dict of float32 columns, Ninit) | None."""
    if p0 is None:
        p0 = good_features(ref_data, mask, conf.maxCorners, conf.qualityLevel, conf.minDistance,
                           conf.blocksize)
    if p0 is None:
        return None
    w = conf.matching_winsize
    p1 = pyr_lk(ref_data, image_data, p0, w)
    p0r = pyr_lk(image_data, ref_data, p1, w)
    d = abs(p0 - p0r).reshape(-1, 2).max(-1)
    back_threshold = 0.1
    st = d < back_threshold
    ninit = len(p0)
    p0s, p1s, d = p0[st], p1[st], d[st]
    score = 1 - d / back_threshold
    x0, y0 = p0s[:, 0, 0], p0s[:, 0, 1]
    x1, y1 = p1s[:, 0, 0], p1s[:, 0, 1]
    if getattr(conf, "outliers_filtering", False):
        x0, y0, x1, y1, score = filter_outliers(x0, y0, x1, y1, score)
    return {"x0": x0, "y0": y0, "dx": x1 - x0, "dy": y1 - y0, "score": score}, ninit


def resolve_ksize(ksize):
    """mon/ref kernel size from int | dict (klt.py:431-432)."""
    if isinstance(ksize, dict):
        return ksize.get("mon", ksize.get("ref", 1)), ksize.get("ref", ksize.get("mon", 1))
    return ksize, ksize


def klt_tile(mon_box, ref_box, conf, mask_box=None, nodata_mon=None, nodata_ref=None,
             x_off=0, y_off=0, invert_mon=False):
    """KLT._match_tile for fixed ksize / fixed polarity (klt.py:236-349, 407-436):
    -> dict of float32 columns sorted by (x0, y0) plus 'Ninit', or None."""
    if mask_box is None:
        mask_box, valid = auto_mask(mon_box, ref_box, nodata_mon, nodata_ref)
    else:
        mask_box = np.asarray(mask_box)
        valid = int((mask_box > 0).sum())
    if valid == 0:
        return None
    mon_k, ref_k = resolve_ksize(conf.laplacian_kernel_size)
    mon_u8 = to_uint8(mon_box, invert=invert_mon)
    lap_mon = laplacian_u8(mon_u8, mon_k)
    lap_ref = laplacian_u8(to_uint8(ref_box), ref_k)
    res = klt_tracker(lap_ref, lap_mon, mask_box, conf)
    if res is None:
        return None
    cols, ninit = res
    cols["x0"] = cols["x0"] + x_off
    cols["y0"] = cols["y0"] + y_off
    order = np.lexsort((cols["y0"], cols["x0"]))
    out = {k: v[order] for k, v in cols.items()}
    out["Ninit"] = ninit
    out["lap_mon"], out["lap_ref"], out["mask"] = lap_mon, lap_ref, mask_box
    return out


def radial_angle(dx, dy):
    """_handle_klt_results columns (core.py:872-873), float32 in -> float32 out."""
    return np.sqrt(dx ** 2 + dy ** 2), np.degrees(np.arctan2(dy, dx))


def zncc_batch(ref, mon, x0, y0, dx, dy):
    """ZNCCService.compute_zncc per keypoint (zncc_service.py:186-238) -> float64[n]."""
    r, dt, sr = _check_img(ref)
    m, dt2, sm = _check_img(mon)
    if dt != dt2:
        raise TypeError("dtype mismatch")
    x0, y0, dx, dy = (np.ascontiguousarray(v, np.float32) for v in (x0, y0, dx, dy))
    n = len(x0)
    out = np.empty(max(n, 1), np.float64)
    lib().ko_zncc_batch(_p(r), _p(m), dt, r.shape[0], r.shape[1], m.shape[0], m.shape[1],
                        C.c_ssize_t(sr), C.c_ssize_t(sm), _p(x0), _p(y0), _p(dx), _p(dy), n, _p(out))
    return out[:n]


def _kp_chips(ref, mon, x0, y0, dx, dy):
    """Chip extraction rules shared by _compute_zncc / _compute_mutual_info / _compute_mi
    (zncc_service.py:186-218, mutual_info_service.py:99-123): yields (k, chip_ref, chip_mon) or (k, None, None)."""
    m = 28
    for k in range(len(x0)):
        X0, Y0 = int(x0[k]), int(y0[k])
        sx, sy = np.float32(x0[k]) + np.float32(dx[k]), np.float32(y0[k]) + np.float32(dy[k])
        if not (np.isfinite(sx) and np.isfinite(sy)):
            yield k, None, None
            continue
        X1, Y1 = round(sx), round(sy)
        if X0 - m < 0 or Y0 - m < 0 or X1 - m < 0 or Y1 - m < 0 or X0 >= ref.shape[1] - m or Y0 >= ref.shape[0] - m \
                or X1 >= mon.shape[1] - m or Y1 >= mon.shape[0] - m:
            yield k, None, None
            continue
        yield k, ref[Y0 - m:Y0 + m + 1, X0 - m:X0 + m + 1], mon[Y1 - m:Y1 + m + 1, X1 - m:X1 + m + 1]


def mi_batch(ref, mon, x0, y0, dx, dy):
    """(studholme, nmi) per keypoint: `_mutual_info` (mutual_info_service.py:32-63) and `_mutual_information`
    (zncc_service.py:129-151) on the 57x57 chips, 32-bin np.histogram2d.  Pure numpy loop: small cases only."""
    n = len(x0)
    st, nmi = np.full(n, np.nan), np.full(n, np.nan)
    for k, c1, c2 in _kp_chips(np.asarray(ref), np.asarray(mon), x0, y0, dx, dy):
        if c1 is None:
            continue
        # float64 samples -> float64 bin edges, whatever numpy runs here: the reference's environment (pandas 2.1 / opencv 4.8 =>
        # numpy 1.x) evaluates `np.linspace(float32 min, float32 max, 33)` in float64 for float32 chips as well, and
        # `_mutual_information` casts explicitly (zncc_service.py:134-135); numpy >= 2 would give float32 edges for float32 chips
        h, _, _ = np.histogram2d(c1.ravel().astype(np.float64), c2.ravel().astype(np.float64), bins=32)
        pxy = h / h.sum()
        px, py = pxy.sum(axis=1), pxy.sum(axis=0)
        ent = lambda p, lg: -np.sum(p[p > 0] * lg(p[p > 0]))
        hx, hy, hxy = ent(px, np.log), ent(py, np.log), ent(pxy, np.log)
        st[k] = np.nan if hxy == 0 else (hx + hy) / hxy
        hx2, hy2, hxy2 = ent(px, np.log2), ent(py, np.log2), ent(pxy.ravel(), np.log2)
        nmi[k] = np.nan if hx2 + hy2 == 0 else 2.0 * (hx2 + hy2 - hxy2) / (hx2 + hy2)
    return st, nmi


def shift_image(img, y_off=0, x_off=0):
    """shift_image (image.py:70-101)."""
    y_off, x_off = int(round(y_off)), int(round(x_off))
    new = np.zeros(img.shape, img.dtype)
    if x_off > 0:
        new[:, :-x_off] = img[:, x_off:]
    elif x_off < 0:
        new[:, -x_off:] = img[:, :x_off]
    if x_off != 0:
        img, new = new, np.zeros(img.shape, img.dtype)
    if y_off > 0:
        new[:-y_off, :] = img[y_off:, :]
    elif y_off < 0:
        new[-y_off:, :] = img[:y_off, :]
    if y_off != 0:
        img = new
    return img


def phase_cross_correlation(reference_image, moving_image):
    """skimage.registration.phase_cross_correlation (0.24 defaults:
    upsample_factor=1, space='real', normalization='phase'), shift only
    (large_offset.py:39).  scikit-image is a third-party
    dependency absent from /root/reference (environment.yml:11)."""
    import scipy.fft as sfft
    src = sfft.fftn(np.asarray(reference_image, np.float64))
    tgt = sfft.fftn(np.asarray(moving_image, np.float64))
    prod = src * tgt.conj()
    eps = np.finfo(prod.real.dtype).eps
    prod /= np.maximum(np.abs(prod), 100 * eps)
    cc = sfft.ifftn(prod)
    maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
    mid = np.array([np.fix(s / 2) for s in cc.shape])
    shift = np.stack(maxima).astype(np.float64)
    shift[shift > mid] -= np.array(cc.shape)[shift > mid]
    for d in range(cc.ndim):
        if cc.shape[d] == 1:
            shift[d] = 0
    return shift


def large_offset(mon, ref, min_threshold=2):
    """LargeOffsetMatcher.match + thresholding of _detect_large_offset
    (large_offset.py:32-41, core.py:756-769) -> [row_off, col_off]."""
    off = phase_cross_correlation(mon, ref)
    if abs(off[1]) < min_threshold:
        off[1] = 0
    if abs(off[0]) < min_threshold:
        off[0] = 0
    return off


def default_conf(**kw):
    """processing_configuration.json:8-19 defaults as a duck-typed config."""
    d = dict(minDistance=10, blocksize=15, maxCorners=20000, matching_winsize=25, qualityLevel=0.1,
             xStart=0, tile_size=20000, laplacian_kernel_size=7, outliers_filtering=False,
             laplacian_invert_polarity=False)
    d.update(kw)
    return SimpleNamespace(**d)


def filter_by_dn_values(x0, y0, ref, mon, no_values=None, ref_nd=None, mon_nd=None):
    """Keep mask of `KariosAPI._filter_by_dn_values` (api/core.py:650-737): pixel under (int(x0), int(y0)) of either image
    equal to one of `no_values`, or equal to its own image's no-data value -> drop."""
    x = np.asarray(x0).astype(int)
    y = np.asarray(y0).astype(int)
    rv, mv = np.asarray(ref)[y, x], np.asarray(mon)[y, x]
    keep = np.ones(len(x), bool)
    for v in no_values or []:
        keep &= ~((rv == v) | (mv == v))
    if ref_nd is not None:
        keep &= ~(rv == ref_nd)
    if mon_nd is not None:
        keep &= ~(mv == mon_nd)
    return keep


def handle_klt_results_columns(frame, ref, mon, confidence_threshold=0.4, large_shift_applied=False):
    """Columns `_handle_klt_results` adds to one tile frame (api/core.py:872-907): dict of numpy arrays in CSV order.
    `frame`: dict with float32 x0, y0, dx, dy, score."""
    dx, dy = np.asarray(frame["dx"], np.float32), np.asarray(frame["dy"], np.float32)
    out = {k: np.asarray(frame[k]) for k in ("x0", "y0", "dx", "dy", "score")}
    out["radial error"] = np.sqrt(dx ** 2 + dy ** 2)
    out["angle"] = np.degrees(np.arctan2(dy, dx))
    if large_shift_applied:
        return out
    n = len(dx)
    keep = np.asarray(frame["score"]) >= confidence_threshold
    z, st, nmi = np.full(n, np.nan), np.full(n, np.nan), np.full(n, np.nan)
    if keep.any():
        sel = [np.asarray(frame[k])[keep] for k in ("x0", "y0", "dx", "dy")]
        z[keep] = zncc_batch(ref, mon, *sel)
        st[keep], nmi[keep] = mi_batch(ref, mon, *sel)
    out["zncc_score"], out["mutual_info_score"], out["mi_score"] = z, st, nmi
    return out
