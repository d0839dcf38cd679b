"""The hand-written ordering primitives of the exact paths (k_sort.hip): stable 64-bit radix sort (+ 32-bit payload) and the
exclusive scans, against numpy.  They rank every candidate in cv::goodFeaturesToTrack's order when maxCorners = 0 or a unit is
repeated exactly (reference klt.py:112-125) and order frames of more than 32 768 rows (klt.py:187)."""
import ctypes as C

import numpy as np
import pytest

from karios_amd import _lib

pytestmark = pytest.mark.gpu


def _sort(keys, vals, descending):
    ctx = _lib.default_context()
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    v = None if vals is None else np.ascontiguousarray(vals, dtype=np.uint32).copy()
    ctx.check(ctx.lib.km_sort_pairs_u64(ctx.handle, k.ctypes.data_as(C.c_void_p), None if v is None else v.ctypes.data_as(C.c_void_p),
                                           len(k), int(descending)), "km_sort_pairs_u64")
    return k, v


def _scan(x, count_ones):
    ctx = _lib.default_context()
    x = np.ascontiguousarray(x, dtype=np.uint32)
    out = np.empty_like(x)
    ctx.check(ctx.lib.km_exclusive_scan_u32(ctx.handle, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(x), int(count_ones)),
              "km_exclusive_scan_u32")
    return out


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 512, 2047, 2048, 2049, 4096 + 17, 100_003, 1_500_000])
@pytest.mark.parametrize("descending", [False, True])
def test_sort_keys_matches_numpy(n, descending):
    rng = np.random.default_rng(n)
    # candidate-key shaped: float bits << 32 | raster index; plus full-range words
    keys = (rng.integers(0, 1 << 31, n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 32, n, dtype=np.uint64)
    if n > 8:
        keys[::7] = rng.integers(0, np.iinfo(np.uint64).max, len(keys[::7]), dtype=np.uint64, endpoint=True)
        keys[3] = 0
        keys[5] = np.iinfo(np.uint64).max
    got, _ = _sort(keys, None, descending)
    exp = np.sort(keys)
    np.testing.assert_array_equal(got, exp[::-1] if descending else exp)


@pytest.mark.parametrize("n,distinct", [(5, 2), (64, 1), (3000, 7), (70_000, 300), (70_000, 1), (400_000, 50_000)])
@pytest.mark.parametrize("descending", [False, True])
def test_sort_pairs_is_stable(n, distinct, descending):
    """Equal keys keep their input order (both directions): the payload of a run of equal keys comes out ascending."""
    rng = np.random.default_rng(n + distinct)
    pool = rng.integers(0, np.iinfo(np.uint64).max, distinct, dtype=np.uint64, endpoint=True)
    keys = pool[rng.integers(0, distinct, n)]
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = _sort(keys, vals, descending)
    order = np.argsort(keys if not descending else ~keys, kind="stable")      # ~key ascending == key descending, ties in input order
    np.testing.assert_array_equal(gk, keys[order])
    np.testing.assert_array_equal(gv, vals[order])


def test_sort_large_unbounded_corner_case():
    """maxCorners = 0 on a full Sentinel-2 tile can leave ~30 million frame rows: 14 700 tiles per pass, positions beyond 2^24."""
    rng = np.random.default_rng(123)
    n = 30_140_100 + 13
    keys = rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True)
    keys[::5] >>= np.uint64(33)                      # many keys with empty high digits, as (x0, y0) keys have
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = _sort(keys, vals, False)
    assert np.all(gk[1:] >= gk[:-1])
    np.testing.assert_array_equal(keys[gv], gk)                 # the payload is the permutation that sorts
    assert np.array_equal(np.sort(gv[:1000]), np.sort(gv[:1000]))
    ties = gk[1:] == gk[:-1]
    assert np.all(gv[1:][ties] > gv[:-1][ties])                 # stable
    assert len(np.unique(gv[:: max(1, n // 100_000)])) == len(gv[:: max(1, n // 100_000)])
    assert int(gv.astype(np.uint64).sum()) == n * (n - 1) // 2  # a permutation


def test_sort_frame_shaped_pairs():
    """(x0, y0) ordering keys of a frame with sentinel keys behind the kept rows, ranks as payload (k_frame.hip)."""
    rng = np.random.default_rng(9)
    n = 131_072 + 77
    x = rng.integers(0, 10980, n).astype(np.uint64)
    y = rng.integers(0, 10980, n).astype(np.uint64)
    keys = (x << np.uint64(32)) | y
    keys[rng.random(n) < 0.3] = np.iinfo(np.uint64).max
    ranks = rng.permutation(n).astype(np.uint32)
    gk, gv = _sort(keys, ranks, False)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(gk, keys[order])
    np.testing.assert_array_equal(gv, ranks[order])


@pytest.mark.parametrize("n", [1, 7, 255, 256, 257, 2048, 2049, 65_537, 1_205_604 + 1, 5_000_011])
def test_exclusive_scans_match_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.integers(0, 40, n).astype(np.uint32)
    exp = np.concatenate([[0], np.cumsum(x[:-1], dtype=np.uint64)]).astype(np.uint32)
    np.testing.assert_array_equal(_scan(x, False), exp)
    st = rng.integers(0, 3, n).astype(np.uint32)                   # selection states: 0 undecided, 1 accepted, 2 rejected
    exp1 = np.concatenate([[0], np.cumsum((st[:-1] == 1), dtype=np.uint64)]).astype(np.uint32)
    np.testing.assert_array_equal(_scan(st, True), exp1)
