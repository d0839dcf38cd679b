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
