"""CPU suite, part 2: host-side logic of karios_amd (no GPU compute) and the C ABI surface."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_abi_library_loads_and_exports_every_declared_symbol():
    from karios_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "karios_hip.h")).read()
    declared = set(re.findall(r"\b(km_[a-z0-9_]+)\s*\(", header))
    declared -= {"km_ctx", "km_klt_params", "km_klt_stats"}
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in karios_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.km_version() >= 100
    assert lib.km_stage_name(1).decode() == "stretch_laplacian_mask"


def test_struct_layout_matches_header():
    from karios_amd._lib import KltParams, KltStats
    assert ctypes.sizeof(KltParams) == 8 * 4 + 3 * 8
    assert ctypes.sizeof(KltStats) == 2 * 8 + 2 * 4 + 4 * 8 + 2 * 4 + 2 * 4 and KltStats.path_flags.offset == 64
    assert KltParams.quality_level.offset == 32 and KltStats.min_ref.offset == 24
    from karios_amd._lib import KmUnit                            # km_unit of include/karios_hip.h: 8 pointers / strides, 10 int32 / float, mask pointer + stride
    assert ctypes.sizeof(KmUnit) == 8 * 8 + 10 * 4 + 2 * 8 and KmUnit.H.offset == 64 and KmUnit.win_ox.offset == 88 and KmUnit.d_mask.offset == 104


def test_no_device_fails_loudly_without_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from karios_amd import _lib
    with pytest.raises(_lib.KariosHipError, match="no CPU fallback"):
        _lib.Context(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "karios_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
                assert "karios_oracle" not in src and "libkarios_oracle" not in src, f


def test_tile_boxes_reference_order():
    from karios_amd.core import KLTConfiguration
    from karios_amd.matcher.klt import KLT
    boxes = KLT(KLTConfiguration(tile_size=6000)).tile_boxes(10980, 10980)
    assert boxes == [(0, 0, 6000, 6000), (0, 6000, 6000, 4980), (6000, 0, 4980, 6000), (6000, 6000, 4980, 4980)]
    assert KLT(KLTConfiguration(tile_size=20000)).tile_boxes(10980, 10980) == [(0, 0, 10980, 10980)]
    skipped = KLT(KLTConfiguration(tile_size=100, xStart=150)).tile_boxes(300, 100)
    assert [b[0] for b in skipped] == [200]                      # x_off 0 and 100 are < xStart
    assert len(KLT(KLTConfiguration(tile_size=100)).tile_boxes(200, 200)) == 4   # reference test_klt_match_method


def test_frame_from_tracks_matches_reference_fb_arithmetic():
    from karios_amd import frames
    g = np.load(os.path.join(G, "fb_score.npz"))
    cols, ninit = frames.track_columns(g["p0"], g["p1"], g["p0r"])
    frame = frames.assemble(cols, ordered=False)
    assert ninit == int(g["ninit"])
    assert list(frame.columns) == ["x0", "y0", "dx", "dy", "score"]
    for c in frame.columns:
        assert frame[c].dtype == np.float32
        np.testing.assert_array_equal(frame[c].to_numpy(), g[c])
    assert len(frame) < ninit and frame["score"].max() == 1.0


def test_filter_outliers_matches_reference():
    from karios_amd import frames
    g = np.load(os.path.join(G, "outliers.npz"))
    keep = frames.sigma_clip(g["x1"] - g["x0"], g["y1"] - g["y0"])
    assert 0 < len(keep) < len(g["x0"])
    for i, name in enumerate(("x0", "y0", "x1", "y1", "score")):
        np.testing.assert_array_equal(g[name][keep], g[f"out_{i}"])
    assert len(frames.sigma_clip(np.zeros(0, np.float32), np.zeros(0, np.float32))) == 0


def test_ordered_frame_keeps_the_labels_of_an_inplace_sort():
    """`assemble(ordered=True)` == DataFrame + offsets + sort_values(["x0", "y0"], inplace=True) (klt.py:341-348)."""
    from karios_amd import frames
    g = np.load(os.path.join(G, "fb_score.npz"))
    cols, _ = frames.track_columns(g["p0"], g["p1"], g["p0r"])
    got = frames.assemble(cols, x_off=130, y_off=260)
    want = pd.DataFrame({k: v.copy() for k, v in cols.items()})
    want["x0"] = want["x0"] + 130
    want["y0"] = want["y0"] + 260
    want.sort_values(by=["x0", "y0"], inplace=True)
    pd.testing.assert_frame_equal(got, want)


def test_resolve_ksize():
    from karios_amd.matcher.klt import KLT
    assert KLT._resolve_ksize(7) == (7, 7)
    assert KLT._resolve_ksize({"mon": 5, "ref": 9}) == (5, 9)
    assert KLT._resolve_ksize({"ref": 3}) == (3, 3) and KLT._resolve_ksize({"mon": 11}) == (11, 11) and KLT._resolve_ksize({}) == (1, 1)


def test_numpy_raster_image_duck_type():
    from karios_amd.core import NumpyRasterImage
    a = np.arange(60, dtype=np.uint16).reshape(6, 10)
    im = NumpyRasterImage(a, no_data_value=0, filepath="/x/y/img.tif")
    assert (im.x_size, im.y_size, im.file_name) == (10, 6, "img.tif")
    np.testing.assert_array_equal(im.read(1, 2, 1, 3, 4), a[1:5, 2:5])
    im.clear_cache()
    assert im.array is a


def test_zncc2_mirror_error_contract():
    from karios_amd.matcher.zncc_service import _zncc2
    img = np.ones((5, 5))
    with pytest.raises(ValueError, match="must be non-negative"):
        _zncc2(img, img, 2, 2, 2, 2, -1)
    with pytest.raises(IndexError):
        _zncc2(img, img, 0, 0, 0, 0, 3)


def test_pack_unpack_and_unit_enumeration():
    from karios_amd.core import KLTConfiguration
    from karios_amd.parallel import enumerate_units, pack_frame, unpack_frame, units_of_rank
    f = pd.DataFrame(np.random.default_rng(0).random((7, 5)).astype(np.float32), columns=["x0", "y0", "dx", "dy", "score"],
                     index=[5, 0, 3, 6, 1, 2, 4])              # the permuted labels an in-place sort leaves
    back = unpack_frame(pack_frame(f, 10))
    pd.testing.assert_frame_equal(back, f)
    scored = f.assign(zncc_score=np.linspace(0, 1, 7))
    scored.loc[3, "zncc_score"] = np.nan
    pd.testing.assert_frame_equal(unpack_frame(pack_frame(scored, 10, True), with_zncc=True), scored)
    assert unpack_frame(pack_frame(None, 4)) is None
    assert len(unpack_frame(pack_frame(f.iloc[:0], 4))) == 0          # corners found, none survived: an empty frame, not None
    with pytest.raises(ValueError):
        pack_frame(f, 3)
    units = enumerate_units(4, 10980, 10980, KLTConfiguration(tile_size=5490))
    assert len(units) == 16 and [u.index for u in units] == list(range(16))
    assert (units[1].x_off, units[1].y_off) == (0, 5490) and (units[2].x_off, units[2].y_off) == (5490, 0)   # x outer, y inner
    parts = [units_of_rank(units, r, 8) for r in range(8)]
    assert all(len(p) == 2 for p in parts) and sorted(u.index for p in parts for u in p) == list(range(16))


def test_gather_frames_world_size_2_gloo(tmp_path):
    """N>1 path: two CPU processes (gloo) exchange their per-unit frames; both end with every frame in
    reference order."""
    script = tmp_path / "worker.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, pandas as pd, torch.distributed as dist
from karios_amd.parallel import gather_frames
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
def frame(u):
    n = 3 + u
    return pd.DataFrame(np.full((n, 5), u, np.float32) + np.arange(5, dtype=np.float32), columns=["x0", "y0", "dx", "dy", "score"])
local = {{u: (None if u == 2 else frame(u)) for u in range(5) if u % 2 == rank}}
out = gather_frames(local, 5, 16)
assert [None if f is None else len(f) for f in out] == [3, 4, None, 6, 7], out
for u, f in enumerate(out):
    if f is not None:
        pd.testing.assert_frame_equal(f, frame(u))
dist.destroy_process_group()
print("rank", rank, "ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_gather_blocks_world_size_2_gloo(tmp_path):
    """Raw frame blocks (the device pipeline's layout) gathered across two gloo ranks and turned into frames."""
    script = tmp_path / "worker2.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch.distributed as dist
from karios_amd.parallel import gather_blocks, blocks_to_frames, block_len
rank = int(os.environ["RANK"]); cap = 8
dist.init_process_group("gloo", rank=rank, world_size=2)
def block(u):
    b = np.zeros(block_len(cap, True), np.float32)
    n = 2 + u
    b[:4].view(np.int32)[:2] = (n, n + 1)
    body = b[4:]
    for c in range(5):
        body[c * cap:c * cap + n] = 10 * u + c + np.arange(n)
    body[5 * cap:5 * cap + n].view(np.int32)[:] = np.arange(n)[::-1]
    body[6 * cap:8 * cap].view(np.float64)[:n] = 0.5 + u
    return b
local = {{u: (None if u == 1 else block(u)) for u in range(4) if u % 2 == rank}}
out = gather_blocks(local, 4, cap, True)
assert out.shape == (4, block_len(cap, True))
frames = blocks_to_frames(out, cap, True)
assert [None if f is None else len(f) for f in frames] == [2, None, 4, 5], frames
assert list(frames[2].index) == [3, 2, 1, 0] and float(frames[3]["zncc_score"].iloc[0]) == 3.5 and float(frames[2]["dy"].iloc[1]) == 24.0
dist.destroy_process_group()
print("rank", rank, "ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_gather_rank_blocks_world_size_2_gloo(tmp_path):
    """The lean per-step exchange of `bench.py --gpus N` (one frame block per rank, gathered tensor left on the device)."""
    script = tmp_path / "worker3.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch, torch.distributed as dist
from karios_amd.parallel import gather_rank_blocks, blocks_to_frames, block_len
rank = int(os.environ["RANK"]); cap = 6
dist.init_process_group("gloo", rank=rank, world_size=2)
b = np.zeros(block_len(cap, True) + 7, np.float32)          # longer than needed, like the host ring buffer
n = 3 + rank
b[:4].view(np.int32)[:2] = (n, n + 2)
b[4:4 + n] = 100 * rank + np.arange(n)
t, total = gather_rank_blocks(b, cap, True)
assert tuple(t.shape) == (2, block_len(cap, True)) and total == 7, (t.shape, total)
frames = blocks_to_frames(t.cpu().numpy(), cap, True)
assert [len(f) for f in frames] == [3, 4] and float(frames[1]["x0"].iloc[2]) == 102.0
t2, total2 = gather_rank_blocks(None if rank == 0 else b, cap, True)
assert total2 == 4 and blocks_to_frames(t2.cpu().numpy(), cap, True)[0] is None
dist.destroy_process_group()
print("rank", rank, "ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_rank_block_exchange_world_size_2_gloo(tmp_path):
    """`RankBlockExchange` - the streamed exchange of `bench.py --gpus N` (one frame block per rank and step, asynchronous all-gather,
    counts kept with the gathered headers and summed at the end) - on two gloo ranks with host blocks: more steps than ring slots,
    a flagged block (its rows must not be counted), rank-ordered gathered blocks, counters that restart."""
    script = tmp_path / "worker4.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch, torch.distributed as dist
from karios_amd.parallel import RankBlockExchange, block_len
rank = int(os.environ["RANK"]); cap = 5
dist.init_process_group("gloo", rank=rank, world_size=2)
ex = RankBlockExchange(None, cap, True, device="cpu", slots=3)
assert not ex.on_gpu and ex.grouped and ex.ws == 2
L = block_len(cap, True)
def block(step, flagged=False):
    b = np.zeros(L + 3, np.float32)
    b[:4].view(np.int32)[:] = (1 + step + 10 * rank, 20, 32 if flagged else 0, 99)
    b[4] = 1000 * rank + step
    return b
want = 0
for step in range(8):                        # 8 steps through 3 slots
    flagged = step == 5 and rank == 1
    ex.arm(step)                             # (no-op without a device)
    ex.issue(step, host_block=block(step, flagged))
    want += (1 + step) + (0 if step == 5 else 11 + step)      # rank 0's rows + rank 1's rows (unless flagged)
rows, nflag = ex.finish()
assert (rows, nflag) == (want, 1), (rows, nflag, want)
last = ex.last_blocks(7).numpy()
assert last.shape == (2, L) and last[0, 4] == 7.0 and last[1, 4] == 1007.0          # rank order
ex.reset_counts()
ex.issue(8, host_block=None)                 # a rank without a block contributes zeros
assert ex.finish() == (0, 0)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29546", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_stretch_exact_multiples():
    """The integer formulation of `_to_uint8` the HIP kernels use (k_dense.hip, "uint8 stretch of 16-bit integer images"):
    numpy's trunc(fl(fl(d / r) * 255)) equals floor(255 * d / r) for every integer d in [0, r], r <= 65535 - in
    particular at the exact multiples 255 * d == k * r, where only fl(k / 255) * 255 >= k keeps the truncation at k."""
    k = np.arange(256, dtype=np.float64)
    assert np.all((k / 255.0) * 255.0 >= k)
    for r in (1, 2, 3, 254, 255, 256, 510, 765, 1020, 4095, 10000, 12345, 32767, 32768, 65025, 65534, 65535):
        d = np.arange(r + 1, dtype=np.int64)
        want = ((d.astype(np.float64) / float(r)) * 255.0).astype(np.uint8)       # numpy semantics of klt.py:42-49
        got = (255 * d) // r
        assert np.array_equal(want, got.astype(np.uint8)), r
        fma_form = np.floor(d.astype(np.float64) * (255.0 / r) + 0.5 / r)          # what the kernel evaluates (one fma)
        assert np.array_equal(fma_form.astype(np.int64), got), r


def test_oracle_thread_team_respects_the_cpu_quota(O):
    """The oracle's OpenMP team is sized from the CPUs the process may use (affinity, cgroup quota), not from the logical
    CPU count: on the GPU box 256 logical CPUs hide a 16-CPU quota and an all-cores team ran ~50x slower on small tiles."""
    import os
    n = O.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        assert n <= len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    assert 1 <= O.max_threads() <= max(n, 1) or O.max_threads() == 1


def test_block_to_frame_with_radial_columns_equals_the_two_step_construction():
    """`frames.block_to_frame(..., radial=True, own=True)` - the host half of the pipelined bench: one DataFrame construction, columns
    as views of the block - gives the frame of `block_to_frame` + `radial_angle_columns` (reference api/core.py:872-873), bit for bit,
    index labels included; `radial_angle_columns` leaves such a frame alone."""
    import pandas as pd
    from karios_amd import frames
    cap, rows = 500, 321
    rng = np.random.default_rng(7)
    block = np.zeros(4 + 8 * cap, np.float32)
    block[:4].view(np.int32)[:] = (rows, 400, 0, 0)
    body = block[4:]
    body[0:cap] = rng.integers(0, 3000, cap)
    body[cap:2 * cap] = rng.integers(0, 3000, cap)
    body[2 * cap:3 * cap] = rng.normal(0.5, 2.0, cap)
    body[3 * cap:4 * cap] = rng.normal(-0.25, 2.0, cap)
    body[4 * cap:5 * cap] = rng.uniform(0, 1, cap)
    body[5 * cap:6 * cap].view(np.int32)[:] = rng.permutation(cap)
    z = rng.uniform(-1, 1, cap)
    z[::7] = np.nan
    body[6 * cap:8 * cap].view(np.float64)[:] = z
    two_step = frames.radial_angle_columns(frames.block_to_frame(block, cap, True))
    private = block.copy()
    one_step = frames.block_to_frame(private, cap, True, radial=True, own=True)
    pd.testing.assert_frame_equal(one_step, two_step, check_exact=True)
    assert list(one_step.columns) == ["x0", "y0", "dx", "dy", "score", "zncc_score", "radial error", "angle"]
    assert one_step["radial error"].dtype == np.float32 and one_step["angle"].dtype == np.float32
    assert frames.radial_angle_columns(one_step) is one_step
    assert len(one_step) == rows and one_step.index.dtype == np.int64
    block[:2].view(np.int32)[:] = (0, 0)
    assert frames.block_to_frame(block, cap, True, radial=True) is None


# ---------------------------------------------------------------------------- bench.py launcher
def _run_bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_n_spawns_n_ranks_and_fails_when_a_rank_fails():
    """`python bench.py --gpus 2` with no launcher environment starts two ranks itself; without a GPU each of them refuses to run
    (no CPU path) and the parent must report the failure with a non-zero status instead of printing a one-rank line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_bench.py")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                  # no JSON line from a failed job
    assert r.stderr.count("needs an MI355X") == 2, r.stderr        # both ranks were started
    assert "rank 0 exited with status" in r.stderr or "rank 1 exited with status" in r.stderr


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    r = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""


# ---------------------------------------------------------------------------- FrameStream (host logic with a stand-in pair)
class _FakePair:
    """Duck type of ResidentPair for the stream's host logic: `match_tile_raw` hands back a finished block at once."""

    def __init__(self, cap=8):
        self.cap, self.calls, self.ctx = cap, [], None

    def match_tile_raw(self, conf, box=None, zncc_threshold=None, origin=None, mutual_info=False):
        from karios_amd.parallel import pack_frame
        from karios_amd.resident import RawFrame
        k = len(self.calls)
        self.calls.append((box, zncc_threshold, origin))
        n = 1 + k % 3
        f = pd.DataFrame({c: np.full(n, float(k), np.float32) for c in ("x0", "y0", "dx", "dy", "score")})
        f["zncc_score"] = np.linspace(0.0, 1.0, n)
        return RawFrame(pack_frame(f, self.cap, zncc_threshold is not None), self.cap, zncc_threshold is not None)

    def score_frame(self, frame, thr, mutual_info=False):
        from karios_amd import frames
        return frames.radial_angle_columns(frame)


def test_frame_stream_orders_results_bounds_depth_and_restores_the_switch_interval():
    from types import SimpleNamespace
    from karios_amd.stream import FrameStream
    conf = SimpleNamespace(maxCorners=0)
    before = sys.getswitchinterval()
    pair = _FakePair()
    seen = []
    with FrameStream(0.4, depth=2, host_stage=lambda f, p: f.assign(stage=1)) as s:
        assert sys.getswitchinterval() == pytest.approx(1e-4, abs=5e-6)
        got = []
        for k in range(7):
            done = s.submit(pair, conf, box=(k, 0, 1, 1), tag=k)
            seen.append(len(done))
            got += done
        got += s.drain()
    assert sys.getswitchinterval() == pytest.approx(before)
    assert seen == [0, 0, 1, 1, 1, 1, 1]                              # two units stay pending behind each submission
    assert [d.tag for d in got] == list(range(7))                     # submission order
    assert [len(d.frame) for d in got] == [1 + k % 3 for k in range(7)]
    assert all({"radial error", "angle", "zncc_score", "stage"} <= set(d.frame.columns) for d in got)
    assert all(float(d.frame["x0"].iloc[0]) == d.tag and not d.redone for d in got)
    with pytest.raises(RuntimeError):
        s.submit(pair, conf)
    # bare frames: no threshold -> no ZNCC call, no score columns
    with FrameStream(None, depth=0, gil_switch_interval=None) as s0:
        d = s0.submit(_FakePair(), conf)
        assert len(d) == 1 and list(d[0].frame.columns) == ["x0", "y0", "dx", "dy", "score"]


def test_switch_interval_survives_overlapping_streams_closed_out_of_order():
    """the interval is process-global; stream A (opened first) closing before stream B must not leave it lowered."""
    from karios_amd.stream import FrameStream
    before = sys.getswitchinterval()
    a = FrameStream(None)
    b = FrameStream(None)
    assert sys.getswitchinterval() == pytest.approx(1e-4, abs=5e-6)
    a.close()
    assert sys.getswitchinterval() == pytest.approx(1e-4, abs=5e-6)   # B is still open
    b.close()
    b.close()                                                     # idempotent
    assert sys.getswitchinterval() == pytest.approx(before)
    c = FrameStream(None)
    del c                                                         # an abandoned stream gives the interval back too
    import gc
    gc.collect()
    assert sys.getswitchinterval() == pytest.approx(before)


def test_finalizers_on_a_foreign_thread_leave_their_library_calls_to_the_owning_thread():
    """`ResidentPair.__del__` / `DeviceBuffer.__del__` may run on FrameStream's worker (cyclic GC); the context is not
    thread-safe, so their calls are queued and run by the owner's next call."""
    import threading
    from karios_amd._lib import Context
    ctx = object.__new__(Context)                                # no device needed: only the bookkeeping is exercised
    ctx._owner, ctx._abandoned, ctx._abandoned_lock, ctx.handle = threading.get_ident(), [], threading.Lock(), None
    ran = []
    ctx.run_or_defer(lambda: ran.append(("own", threading.get_ident())))
    assert ran == [("own", threading.get_ident())]
    t = threading.Thread(target=lambda: ctx.run_or_defer(lambda: ran.append(("deferred", threading.get_ident()))))
    t.start(); t.join()
    assert len(ran) == 1 and len(ctx._abandoned) == 1             # not run on the foreign thread
    ctx.drain()
    assert ran[1] == ("deferred", threading.get_ident()) and not ctx._abandoned


# ---------------------------------------------------------------------------- inline-assembly hazards of the built kernels
def test_inline_assembly_hazard_scan_of_the_device_code(tmp_path):
    """The kernels write DOT / DPP / packed instructions as inline assembly, which LLVM's hazard recogniser cannot see into: the
    required wait states are padded by hand.  `tools/hazard_scan.py` re-derives them from the gfx950 assembly of every source that
    contains inline assembly (cross-compiled here, no GPU needed) - a violated hazard would be a silent, data-dependent wrong value."""
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "karios_amd", "csrc")
    srcs = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip") and ("asm" in open(os.path.join(csrc, f)).read() or f in ("k_eig2.hip", "k_eig3.hip"))]
    assert {"k_eig3.hip", "k_lk.hip", "k_dense.hip", "k_fft.hip"} <= set(srcs), srcs
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-fno-gpu-flush-denormals-to-zero", "-I/opt/rocm/include", "--cuda-device-only", "-S"]      # (the Makefile's flags)

    def build(f):
        out = str(tmp_path / (f[:-4] + ".s"))
        subprocess.run([hipcc, *flags, "-o", out, os.path.join(csrc, f)], check=True, capture_output=True, timeout=900)
        return out

    with ThreadPoolExecutor(max_workers=4) as ex:
        asm = list(ex.map(build, srcs))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hazard_scan.py"), *asm], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("0 potential DOT hazards") == len(asm) and r.stdout.count("0 potential DPP hazards") == len(asm), r.stdout[-2000:]


def _levels_forward(x, levels):
    """numpy model of the forward levels of k_fft64.hip: level l replaces, in every block of n_l R_l consecutive elements and for
    every r < R_l, the n_l elements at stride R_l by their DFT times W_{n_l R_l}^(r k) - in place."""
    x = x.copy()
    N = len(x)
    tw = np.exp(-2j * np.pi * np.arange(N) / N)
    R = N
    for n in levels:
        Nl, R = R, R // n
        for blk in range(N // Nl):
            for r in range(R):
                idx = blk * Nl + r + R * np.arange(n)
                x[idx] = np.fft.fft(x[idx]) * tw[(r * np.arange(n)) * (N // Nl)]
    return x


def test_phase_plan_levels_multiply_to_the_side_and_pair_every_frequency_with_its_negative():
    """km_phase_plan (host only): the level plan of the double-precision phase correlation (large_offset.py:39 is complex128) for
    Sentinel-2 sides, smooth sides, sides with other small primes and sides that need Bluestein; the positions of the negated
    frequencies are checked against a numpy model of the level arithmetic and numpy's own FFT."""
    import ctypes as C
    from karios_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(4)
    for n, cols in [(10980, 0), (10980, 1), (5490, 0), (1830, 1), (3721, 0), (1, 0), (2, 1), (7, 0), (61, 0), (122, 1), (1000, 0), (1000, 1),
                    (4096, 0), (4096, 1), (65536, 1), (2 * 3 * 5 * 7 * 11 * 13, 0), (127 * 4, 1), (131, 0), (10007, 1), (2 * 257, 0), (6000, 1)]:
        levels = (C.c_int * 32)()
        nl, blue = C.c_int(-1), C.c_int(-1)
        neg = np.full(n, -1, np.int32)
        assert lib.km_phase_plan(n, cols, levels, 16, C.byref(nl), C.byref(blue), neg.ctypes.data_as(C.c_void_p)) == 0
        lv = [(levels[2 * i], levels[2 * i + 1]) for i in range(nl.value)]
        big_prime = max([p for p in range(2, n + 1) if n % p == 0 and all(p % q for q in range(2, int(p ** 0.5) + 1))], default=1) if n < 70000 else 0
        if big_prime > 127:
            assert blue.value >= 2 * n - 1 and blue.value & (blue.value - 1) == 0 and not lv
            np.testing.assert_array_equal(neg, (n - np.arange(n)) % n)
            continue
        assert blue.value == 0 and int(np.prod([a for a, _ in lv] or [1])) == n
        for i, (a, kind) in enumerate(lv):
            if kind == 1:
                assert 11 <= a <= 127 and all(a % q for q in range(2, a))
            else:
                assert all(p in (2, 3, 5, 7) for p in range(2, a + 1) if a % p == 0 and all(p % q for q in range(2, p)))
                assert a <= (2048 if (not cols and i == len(lv) - 1) else 256)
        np.testing.assert_array_equal(neg[neg], np.arange(n))                       # an involution
        if n <= 11000 and lv:
            x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            X, ref = _levels_forward(x, [a for a, _ in lv]), np.fft.fft(x)
            scale = np.abs(ref).max()
            pos_of = np.empty(n, np.int64)                                           # frequency at each position, from the spectrum itself
            for pos in rng.integers(0, n, 40):
                f = int(np.argmin(np.abs(ref - X[pos])))
                assert abs(ref[f] - X[pos]) < 1e-9 * scale
                assert abs(X[neg[pos]] - ref[(-f) % n]) < 1e-9 * scale
    assert lib.km_phase_plan(0, 0, levels, 16, None, None, None) < 0
    assert lib.km_phase_plan(10980, 0, levels, 1, None, None, None) < 0                 # two levels do not fit one slot


def test_the_library_links_no_math_library():
    """Both FFTs, the sort and the scans are hand-written (k_fft.hip, k_fft64.hip, k_sort.hip): the shared object needs the HIP runtime
    and the C / C++ runtimes, nothing else - no rocFFT / hipFFT / rocPRIM / rocBLAS / MIOpen behind the C ABI."""
    import subprocess
    from karios_amd import _lib
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout.lower()
    for name in ("rocfft", "hipfft", "rocprim", "hipcub", "rocblas", "hipblas", "miopen", "rocrand", "rocsparse", "rocsolver"):
        assert name not in out, f"{name} in the library's dependencies:\n{out}"
    assert "libamdhip64" in out


def test_release_library_rejects_development_option_names():
    """the release build's km_set_option accepts only the names include/karios_hip.h documents; nothing that was
    measured-and-not-adopted (or that returned wrong results by design: `fft_dbg`) is reachable through the public ABI.  Host-only:
    the accepted names are read from the source of km_set_option and looked up in the strings of the compiled library."""
    import re
    from karios_amd import _lib
    lib = _lib.load()
    assert lib.km_is_dev_build() == 0, "the in-tree library must be the release build (make -C karios_amd/csrc, no DEV=1)"
    header = open(os.path.join(ROOT, "include", "karios_hip.h")).read()
    documented = set(re.findall(r'^ \*   "([a-z0-9_]+)"', header, re.M))
    assert {"fused_eig", "speculative", "frame_mi", "phase_fp64", "spec_flag", "profile_stage"} <= documented
    source = open(os.path.join(ROOT, "karios_amd", "csrc", "api.hip")).read()
    body = source[source.index("int km_set_option("):source.index("// development build: the counters of")]
    release, dev = body.split("#ifdef KM_DEV")
    assert "eig3_count" in dev
    accepted = set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', release))
    assert accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))
    dev_names = set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', dev))
    gone = {"fft_dbg", "fft_ts", "lk_pair", "lk_order", "tail_overlap", "mm_early_at"}
    assert not (gone & (accepted | dev_names))
    # the compiled library agrees
    strings = open(os.path.join(ROOT, "karios_amd", "libkarios_hip.so"), "rb").read()
    for name in sorted(gone | dev_names):
        assert (b"\0" + name.encode() + b"\0") not in strings, f"development option name {name!r} is compiled into the release library"
    for name in sorted(accepted):
        assert (b"\0" + name.encode() + b"\0") in strings or name.encode() in strings, name


# ---------------------------------------------------------------------------- bench.py: the driver's line stays small
def _r05_detail():
    import json
    return json.loads(open(os.path.join(ROOT, "profiles", "bench_r05.json")).read().strip().splitlines()[-1])


def test_bench_small_line_is_small_and_carries_the_contract():
    """Round 5's 21 KB line did not fit the driver's record (BENCH_r05.json `parsed: null`).  `benchkit.summary.small_line` turns the full
    detail - here round 5's own line, 21 KB - into the line that is printed LAST: < 8000 B, the contract keys, roofline + cpu_baseline."""
    import json
    sys.path.insert(0, ROOT)
    from benchkit import summary
    detail = _r05_detail()
    assert len(json.dumps(detail)) > 20000
    line = summary.small_line(detail)
    text = json.dumps(line)
    assert len(text) < 8000 and len(text) <= summary.MAX_BYTES
    for key in summary.CONTRACT_KEYS:
        assert key in line, key
    assert line["value"] == detail["value"] and line["steps"] == detail["steps"] and line["warmup"] == detail["warmup"]
    assert isinstance(line["config"]["workload"], str) and 0 < len(line["config"]["workload"]) <= 118
    roof = line["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_8d_model", "valu_pipe_busy"):
        assert key in roof, key
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["parity"]["passed"] is True and len(cb["sample"]) <= 118
    for key in ("one_pair_per_submission_ms", "end_to_end_ms", "config3_fp64_ms", "config4_ms"):
        assert key in line["summary"], key
    assert line["gates_all_passed"] is True and line["gates_passed"]["config3"] is True
    # no string of the line is long enough for the driver to cut it
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(x) for x in strings(line)) <= 140


def test_bench_small_line_guard_drops_optional_parts_first():
    import json
    sys.path.insert(0, ROOT)
    from benchkit import summary
    detail = _r05_detail()
    detail["stage_ms"] = {f"stage_{i}": 0.1234 + i for i in range(400)}         # a leg that grew: the guard sheds it, the contract stays
    line = summary.small_line(detail)
    assert len(json.dumps(line)) <= summary.MAX_BYTES and "stage_ms" not in line
    assert all(k in line for k in summary.CONTRACT_KEYS)
    failed = dict(_r05_detail())
    failed["config3"] = dict(failed["config3"], gate=dict(failed["config3"]["gate"], passed=False))
    assert summary.small_line(failed)["gates_all_passed"] is False


def test_committed_bench_records_are_single_small_lines_with_the_contract():
    """The round's committed records (profiles/bench_r06*.json = the LAST stdout line of the three bench commands of tools/final_measure.sh):
    one JSON line each, < 6 KB, every contract key, the roofline and cpu_baseline objects, every in-run gate passed."""
    import json
    sys.path.insert(0, ROOT)
    from benchkit import summary
    for name, with_cpu in (("bench_r06.json", True), ("bench_r06_default.json", True), ("bench_r06_config3.json", False)):
        raw = open(os.path.join(ROOT, "profiles", name)).read()
        assert raw.count("\n") <= 1 and len(raw) < summary.MAX_BYTES, name
        line = json.loads(raw)
        for key in summary.CONTRACT_KEYS:
            assert key in line, (name, key)
        assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None and line["data"] == "synthetic"
        assert line["gates_all_passed"] is True and all(line["gates_passed"].values()), name
        if with_cpu:
            roof, cb = line["roofline"], line["cpu_baseline"]
            assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4 and roof["traffic"] > 0
            assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["parity"]["passed"] is True and cb["parity"]["pairs_gated"] == 4
            assert abs(line["value"] - 10980 * 10980 / 1e6 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    first = json.loads(open(os.path.join(ROOT, "profiles", "bench_r06.json")).read())
    assert first["steps"] == 20 and first["warmup"] == 5                     # (the driver's flags)
    stdout = open(os.path.join(ROOT, "profiles", "bench_r06_stdout.txt")).read().strip().splitlines()
    assert json.loads(stdout[-1]) == first and all(l.startswith("detail ") for l in stdout[:-1] if l.strip())


def test_every_tool_is_listed_in_the_tools_readme():
    readme = open(os.path.join(ROOT, "tools", "README.md")).read()
    for sub in ("", "investigations"):
        for f in sorted(os.listdir(os.path.join(ROOT, "tools", sub))):
            if f.endswith((".py", ".sh")):
                assert f"`{f}" in readme, f"tools/{sub}/{f} is not described in tools/README.md"


# ---------------------------------------------------------------------------- shared_pair cannot return stale pixels
def test_shared_pair_token_never_serves_stale_pixels(monkeypatch):
    """The resident copy the matcher services share is keyed on the uploaded buffers + a token, not on a sparse sample of the pixels:
    while it exists the host arrays are read-only (an in-place edit fails loudly), `NumpyRasterImage.clear_cache()` by the owner drops
    it, and a pixel the old 61 x 67-strided fingerprint never looked at is re-uploaded.  CPU only: the upload is counted, not made."""
    from karios_amd import resident as R
    from karios_amd.core import NumpyRasterImage
    uploads = []

    class FakePair:
        pass

    monkeypatch.setattr(R.ResidentPair, "upload", classmethod(lambda cls, mon, ref, ctx=None, **kw: uploads.append((mon.copy(), ref.copy())) or FakePair()))
    R.forget_shared_pairs()
    ctx = object()
    mon, ref = np.zeros((610, 670), np.uint16), np.ones((610, 670), np.uint16)
    mon_img, ref_img = NumpyRasterImage(mon), NumpyRasterImage(ref)
    # KLT.match publishes the whole-image tile it uploaded (a VIEW of the raster's array, as `read()` returns it) ...
    pair = R.shared_pair(mon_img.read(1, 0, 0, 670, 610), ref_img.read(1, 0, 0, 670, 610), ctx, publish=FakePair(), rasters=(mon_img, ref_img))
    # ... and the services find it through `.array`
    assert R.shared_pair(mon_img.array, ref_img.array, ctx) is pair and not uploads
    with pytest.raises(ValueError, match="read-only"):
        mon[3, 5] = 7                                             # (3, 5) is off the old fingerprint's grid (rows % 10, columns % 10)
    # the services' own closing clear_cache() calls keep the entry of an in-memory raster ...
    with R.keep_shared_across(mon_img, ref_img):
        mon_img.clear_cache()
        ref_img.clear_cache()
    assert R.shared_pair(mon_img.array, ref_img.array, ctx) is pair and not uploads
    # ... the OWNER's clear_cache() is the token's other half: write access comes back, the edited pixel is uploaded
    mon_img.clear_cache()
    mon[3, 5] = 7
    fresh = R.shared_pair(mon_img.array, ref_img.array, ctx, rasters=(mon_img, ref_img))
    assert fresh is not pair and len(uploads) == 1 and uploads[0][0][3, 5] == 7
    # somebody lifts the guard behind the library's back and edits: the lookup notices the writeable array and uploads again
    mon.flags.writeable = True
    mon[4, 6] = 9
    again = R.shared_pair(mon_img.array, ref_img.array, ctx)
    assert again is not fresh and len(uploads) == 2 and uploads[1][0][4, 6] == 9
    # a raster whose clear_cache() drops its array (GdalRasterImage, image.py:445-447) comes back with a NEW buffer: a miss
    class Reloading:
        def __init__(self, a):
            self._a = a

        @property
        def array(self):
            return self._a

        def clear_cache(self):
            self._a = self._a.copy()

    g_mon, g_ref = Reloading(np.zeros((64, 64), np.uint16)), Reloading(np.ones((64, 64), np.uint16))
    first = R.shared_pair(g_mon.array, g_ref.array, ctx, rasters=(g_mon, g_ref))
    with R.keep_shared_across(g_mon, g_ref):
        g_mon.clear_cache()
        g_ref.clear_cache()
    assert R.shared_pair(g_mon.array, g_ref.array, ctx) is not first
    R.forget_shared_pairs()
    assert mon.flags.writeable and ref.flags.writeable
    # rasters over page-locked memory (karios_amd.pinned_empty: np.frombuffer(...).reshape - the raster's `.array` is a VIEW object of its
    # own, not on the base chain of what `read()` returns): found again through `.array`, guarded, released
    import ctypes
    bufs = [(ctypes.c_char * (2 * 48 * 64))() for _ in range(2)]
    p_mon, p_ref = (NumpyRasterImage(np.frombuffer(b, dtype=np.uint16, count=48 * 64).reshape(48, 64)) for b in bufs)
    n_up = len(uploads)
    pinned = R.shared_pair(p_mon.read(1, 0, 0, 64, 48), p_ref.read(1, 0, 0, 64, 48), ctx, publish=FakePair(), rasters=(p_mon, p_ref))
    assert R.shared_pair(p_mon.array, p_ref.array, ctx) is pinned and len(uploads) == n_up and not p_mon.array.flags.writeable
    R.forget_shared_pairs()
    assert p_mon.array.flags.writeable and p_ref.array.flags.writeable
