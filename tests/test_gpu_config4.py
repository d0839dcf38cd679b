"""BASELINE config 4 - 4 band pairs of 10980^2, tile_size 5490 = 16 independent work units - on one GPU, and
the distributed entry point with two ranks (gloo) sharing that GPU.  Each unit is uploaded alone (its box + a halo for the
ZNCC chips); its frame must equal the tile loop over the fully resident band, and the oracle for one 5490^2 unit."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, TILE, BANDS = 10980, 5490, 4


def _band(dev, b):
    import torch
    from karios_amd import synth
    mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * b, device=dev)   # seeds per SURVEY 8d
    torch.cuda.synchronize()
    return mon_t, ref_t                                   # int16 storage of the uint16 bit patterns, resident in HBM


def test_config4_sixteen_units_on_one_gpu(ops, O):
    import torch
    from karios_amd.core import DeviceRasterImage, KLTConfiguration
    from karios_amd.parallel import enumerate_units, match_distributed
    from karios_amd.resident import ResidentPair
    dev = torch.device("cuda", 0)
    conf = KLTConfiguration(tile_size=TILE)
    units = enumerate_units(BANDS, S, S, conf)
    assert len(units) == 16 and [(u.band, u.x_off, u.y_off) for u in units[:5]] == [(0, 0, 0), (0, 0, TILE), (0, TILE, 0), (0, TILE, TILE), (1, 0, 0)]
    # the bands stay in HBM (a unit's region is a device-to-device copy): the test moves no image over PCIe except the one
    # 5490^2 unit the oracle needs
    dev_bands = {b: _band(dev, b) for b in range(BANDS)}
    bands = {b: (DeviceRasterImage(m, np.uint16), DeviceRasterImage(r, np.uint16)) for b, (m, r) in dev_bands.items()}
    got = match_distributed(bands, BANDS, S, S, conf, score=True)
    assert len(got) == 16 and all(f is not None and len(f) > 10000 for f in got)
    for u, f in zip(units, got):                                           # unit order = band major, x outer, y inner
        assert f["x0"].min() >= u.x_off + 1 and f["x0"].max() <= u.x_off + u.x_size - 2
        assert f["y0"].min() >= u.y_off + 1 and f["y0"].max() <= u.y_off + u.y_size - 2
        assert list(f.columns) == ["x0", "y0", "dx", "dy", "score", "zncc_score", "radial error", "angle"]
        assert abs(float(np.median(f["dx"])) - 0.5) < 0.03 and abs(float(np.median(f["dy"])) - 0.25) < 0.03
    # every unit == the same tile of the fully resident band (same kernels, ZNCC chips cut from the whole image)
    for b in (0, 3):
        m_t, r_t = dev_bands[b]
        pair = ResidentPair.from_device_pointers(m_t.data_ptr(), r_t.data_ptr(), np.uint16, S, S, keepalive=(m_t, r_t))
        for u in units[4 * b:4 * b + 4]:
            want = pair.score_frame(pair.match_tile(conf, u.box, zncc_threshold=0.4), 0.4)
            pd.testing.assert_frame_equal(got[u.index], want[list(got[u.index].columns)], check_exact=True)
    # one 5490^2 unit against the oracle
    u = units[6]                                                           # band 1, (5490, 0)
    rx, ry = u.x_off - 128, 0                                               # the unit plus the margin its ZNCC chips may reach into
    m, r = (t[ry:u.y_off + u.y_size + 128, rx:].contiguous().cpu().numpy().view(np.uint16) for t in dev_bands[1])
    sl = (slice(u.y_off - ry, u.y_off - ry + u.y_size), slice(u.x_off - rx, u.x_off - rx + u.x_size))
    exp = O.klt_tile(np.ascontiguousarray(m[sl]), np.ascontiguousarray(r[sl]), conf, x_off=u.x_off, y_off=u.y_off)
    f = got[6]
    assert len(f) == len(exp["x0"])
    for col in ("x0", "y0", "dx", "dy", "score"):
        np.testing.assert_array_equal(f[col].to_numpy(), exp[col])
    keep = exp["score"] >= np.float32(0.4)
    # ZNCC by the oracle on the cut-out (its right and top borders are the image's, the others lie >= 128 px outside the
    # unit): the monitored centre round(x0 + dx) is taken in IMAGE coordinates - the float32 sum depends on the magnitude of
    # x0 - and handed over as an integer displacement, which the cut-out's smaller coordinates cannot round differently
    x0, y0 = exp["x0"][keep], exp["y0"][keep]
    dxi = np.rint(x0 + exp["dx"][keep]) - x0
    dyi = np.rint(y0 + exp["dy"][keep]) - y0
    z = O.zncc_batch(r, m, x0 - np.float32(rx), y0 - np.float32(ry), dxi.astype(np.float32), dyi.astype(np.float32))
    gz = f["zncc_score"].to_numpy()
    assert np.all(np.isnan(gz[~keep])) and np.array_equal(np.isnan(gz[keep]), np.isnan(z)) and np.nanmax(np.abs(gz[keep] - z)) <= 1e-9


def test_match_distributed_two_ranks_share_the_gpu(tmp_path):
    """`match_distributed` with world_size 2 (gloo, both ranks on GPU 0): the units are split round-robin, each rank uploads
    only its own regions, and both ranks end with all frames, equal to the single-process result."""
    script = tmp_path / "worker.py"
    script.write_text(f'''
import os, sys, pickle
sys.path.insert(0, {ROOT!r})
import numpy as np, torch, torch.distributed as dist
from karios_amd import synth
from karios_amd.core import KLTConfiguration, NumpyRasterImage
from karios_amd.parallel import enumerate_units, match_distributed, units_of_rank
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
conf = KLTConfiguration(tile_size=450, maxCorners=700, laplacian_kernel_size=5)
H, W, NB = 800, 900, 3
units = enumerate_units(NB, W, H, conf)
mine = {{u.band for u in units_of_rank(units, rank, world)}}
reads = []
class Img(NumpyRasterImage):
    def read(self, band, x, y, w, h):
        reads.append((x, y, w, h))
        return super().read(band, x, y, w, h)
bands = {{}}
for b in sorted(mine):
    mon, ref = synth.make_pair(H, W, 0.3 + 0.1 * b, -0.2, seed=50 + b, nodata_wedge=(b == 1))
    bands[b] = (Img(mon), Img(ref))
frames = match_distributed(bands, NB, W, H, conf, score=True, halo=64, device="cpu" if world > 1 else None)
assert len(reads) == 2 * len(units_of_rank(units, rank, world))          # one region per image and owned unit, nothing else
assert all(w <= 450 + 128 and h <= 450 + 128 for (_, _, w, h) in reads)
pickle.dump(frames, open(os.environ["OUT"], "wb"))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    single = tmp_path / "single.pkl"
    subprocess.run([sys.executable, str(script)], env=dict(env, RANK="0", WORLD_SIZE="1", OUT=str(single)), check=True, timeout=600)
    outs = [tmp_path / f"r{r}.pkl" for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="2", OUT=str(outs[r])),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    want = pd.read_pickle(single)
    assert len(want) == 12 and sum(f is not None for f in want) >= 10
    for path in outs:
        got = pd.read_pickle(path)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert (g is None) == (w is None)
            if g is not None:
                pd.testing.assert_frame_equal(g, w, check_exact=True)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_single_tile_matched_by_several_ranks_exactly(tmp_path, world):
    """the reference's default single-tile configuration matched by `world` ranks (gloo, sharing GPU 0), each
    holding only its row band + halo: global min / max and maximum eigenvalue by all-reduce, one ranked selection on the
    gathered candidate keys, tracks in image coordinates on a virtual row origin.  The frame on every rank must be the
    single-GPU frame of `ResidentPair.match_tile` - rows, order, index labels, every float32 bit, ZNCC to 1e-12."""
    script = tmp_path / "band_worker.py"
    script.write_text(f'''
import os, sys, pickle
sys.path.insert(0, {ROOT!r})
import numpy as np, torch, torch.distributed as dist
from karios_amd import synth
from karios_amd.core import KLTConfiguration, NumpyRasterImage
from karios_amd.parallel import match_tile_banded, band_rows
from karios_amd.resident import ResidentPair
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
H, W = 1180, 760
out = {{}}
for case, kw in (("wedge", dict(maxCorners=2500, laplacian_kernel_size=7)), ("mask_inv", dict(maxCorners=600, minDistance=14.3, blocksize=7, laplacian_kernel_size={{"mon": 5, "ref": 9}}, laplacian_invert_polarity=True, matching_winsize=31)),
                 ("all", dict(maxCorners=0, qualityLevel=0.3, laplacian_kernel_size=3))):
    mon, ref = synth.make_pair(H, W, 0.6, -0.35, seed=77, nodata_wedge=(case == "wedge"))
    mask = None
    if case == "mask_inv":
        mon = (mon.max() - mon).astype(np.uint16)                 # negative radiometry: what laplacian_invert_polarity is for
        mask = np.full((H, W), 255, np.uint8); mask[300:700, 100:400] = 0; mask[560:640, :] = 0
    conf = KLTConfiguration(**kw)
    reads = []
    class Img(NumpyRasterImage):
        def read(self, band, x, y, w, h):
            reads.append((y, h)); return super().read(band, x, y, w, h)
    frame = match_tile_banded(Img(mon), Img(ref), None if mask is None else Img(mask), conf, zncc_threshold=0.4, device="cpu")
    edges = band_rows(H, world)
    assert all(h <= edges[rank + 1] - edges[rank] + 2 * 96 for (_, h) in reads), reads      # only the band and its halo were read
    if rank == 0 and world == 1:
        pair = ResidentPair.upload(mon, ref, mask)
        want = pair.match_tile(conf, zncc_threshold=0.4)
        out[case + "_single"] = want
    out[case] = frame
    if case == "wedge" and world > 1:                              # a halo the 25x25 windows of pyramid level 1 do not fit in
        try:
            match_tile_banded(Img(mon), Img(ref), None, conf, halo=32, device="cpu"); out["small_halo"] = "accepted"
        except Exception as e:
            out["small_halo"] = f"{{type(e).__name__}}: {{e}}"
pickle.dump(out, open(os.environ["OUT"], "wb"))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
''')
    import pandas as pd
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29550 + world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    single = tmp_path / "single.pkl"
    subprocess.run([sys.executable, str(script)], env=dict(env, RANK="0", WORLD_SIZE="1", OUT=str(single)), check=True, timeout=600)
    ref = pd.read_pickle(single)
    for case in ("wedge", "mask_inv", "all"):
        want, one_band = ref[case + "_single"], ref[case]
        assert want is not None and len(want) > 100
        pd.testing.assert_frame_equal(one_band, want, check_exact=False, rtol=0, atol=1e-12)   # one band == the plain single-GPU call
        assert all(np.array_equal(one_band[c].to_numpy(), want[c].to_numpy()) for c in ("x0", "y0", "dx", "dy", "score"))
    if world == 1:
        return
    outs = [tmp_path / f"r{r}.pkl" for r in range(world)]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE=str(world), OUT=str(outs[r])),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    for path in outs:
        got = pd.read_pickle(path)
        assert got["small_halo"].startswith("KariosHipError") and "halo" in got["small_halo"], got["small_halo"]   # on EVERY rank
        for case in ("wedge", "mask_inv", "all"):
            want = ref[case + "_single"]
            g = got[case]
            assert g is not None and len(g) == len(want)
            assert np.array_equal(g.index.to_numpy(), want.index.to_numpy())
            for c in ("x0", "y0", "dx", "dy", "score"):
                np.testing.assert_array_equal(g[c].to_numpy(), want[c].to_numpy(), err_msg=f"{case} {c}")
            gz, wz = g["zncc_score"].to_numpy(), want["zncc_score"].to_numpy()
            assert np.array_equal(np.isnan(gz), np.isnan(wz)) and np.nanmax(np.abs(gz - wz)) <= 1e-12
