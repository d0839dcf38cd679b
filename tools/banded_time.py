"""Single-tile row-band mode (karios_amd.parallel.match_tile_banded) at Sentinel-2 size on ONE rank: checks the
frame against ResidentPair.match_tile and prints where the time goes.  World sizes > 1 are covered by
tests/test_gpu_config4.py::test_single_tile_matched_by_several_ranks_exactly (gloo ranks sharing one GPU)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from karios_amd import synth, pinned_empty
from karios_amd.core import KLTConfiguration, NumpyRasterImage
from karios_amd.parallel import match_tile_banded
from karios_amd.resident import ResidentPair

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 10980
import torch
mon_t, ref_t = synth.make_pair_torch(H, W, 0.5, 0.25)
mon, ref = pinned_empty((H, W), np.uint16), pinned_empty((H, W), np.uint16)
mon[...] = mon_t.cpu().numpy().view(np.uint16); ref[...] = ref_t.cpu().numpy().view(np.uint16)
del mon_t, ref_t
conf = KLTConfiguration()
pair = ResidentPair.upload(mon, ref)
want = pair.match_tile(conf, zncc_threshold=0.4)
for it in range(3):
    t0 = time.perf_counter()
    got = match_tile_banded(NumpyRasterImage(mon), NumpyRasterImage(ref), None, conf, zncc_threshold=0.4, device="cpu")
    dt = time.perf_counter() - t0
    same = len(got) == len(want) and all(np.array_equal(got[c].to_numpy(), want[c].to_numpy()) for c in ("x0", "y0", "dx", "dy", "score"))
    dz = np.nanmax(np.abs(got.zncc_score.to_numpy() - want.zncc_score.to_numpy()))
    print(f"banded, 1 rank, {H}x{W}: {dt * 1e3:.1f} ms (upload of the pair included), rows {len(got)}, identical {same}, max |dZNCC| {dz:.1e}")
