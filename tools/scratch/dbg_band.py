import sys, numpy as np
sys.path.insert(0, "/root/repo")
from karios_amd import synth
from karios_amd.core import KLTConfiguration, NumpyRasterImage
from karios_amd.parallel import match_tile_banded
from karios_amd.resident import ResidentPair
H, W = 1180, 760
for case, kw in (("wedge", dict(maxCorners=2500, laplacian_kernel_size=7)), ("plain", dict(maxCorners=2500, laplacian_kernel_size=7)),
                 ("all", dict(maxCorners=0, qualityLevel=0.3, laplacian_kernel_size=3))):
    mon, ref = synth.make_pair(H, W, 0.6, -0.35, seed=77, nodata_wedge=(case == "wedge"))
    conf = KLTConfiguration(**kw)
    pair = ResidentPair.upload(mon, ref, None)
    w0 = pair.match_tile(conf, zncc_threshold=0.4)
    print(case, "before", None if w0 is None else (len(w0), float(w0.dx.mean()), float(w0.dy.mean())))
    f = match_tile_banded(NumpyRasterImage(mon), NumpyRasterImage(ref), None, conf, zncc_threshold=0.4, device="cpu")
    print(case, "banded", None if f is None else (len(f), float(f.dx.mean()), float(f.dy.mean())))
    w1 = pair.match_tile(conf, zncc_threshold=0.4)
    print(case, "after", None if w1 is None else (len(w1), float(w1.dx.mean()), float(w1.dy.mean())))
    if f is not None and w0 is not None and len(f) == len(w0):
        for c in ("x0", "y0", "dx", "dy", "score"):
            print("   ", c, np.array_equal(f[c].to_numpy(), w0[c].to_numpy()))
        print("    zncc", np.nanmax(np.abs(f.zncc_score.to_numpy() - w0.zncc_score.to_numpy())))
