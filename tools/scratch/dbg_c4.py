import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, pandas as pd
from karios_amd import synth
from karios_amd.core import KLTConfiguration, NumpyRasterImage
from karios_amd.parallel import enumerate_units, match_distributed, ResidentUnit
from karios_amd.resident import ResidentPair
S=4000; T=2000
dev=torch.device("cuda",0)
mon_t, ref_t = synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101, device=dev); torch.cuda.synchronize()
m, r = mon_t.cpu().numpy().view(np.uint16), ref_t.cpu().numpy().view(np.uint16)
conf=KLTConfiguration(tile_size=T, maxCorners=5000)
units=enumerate_units(1,S,S,conf)
got=match_distributed({0:(NumpyRasterImage(m),NumpyRasterImage(r))},1,S,S,conf,score=True)
pair=ResidentPair.upload(m,r)
for u in units:
    want=pair.score_frame(pair.match_tile(conf,u.box,zncc_threshold=0.4),0.4)
    g=got[u.index]
    a,b=g["zncc_score"].to_numpy(), want["zncc_score"].to_numpy()
    bad=np.flatnonzero(~((a==b)|(np.isnan(a)&np.isnan(b))))
    print("unit",u.index,"rows",len(g),"bad",len(bad), "other cols equal", all(np.array_equal(g[c].to_numpy(),want[c].to_numpy()) for c in ("x0","y0","dx","dy","score")))
    for i in bad[:8]:
        print("   ", g.iloc[i].to_dict(), "want", b[i], "diff", a[i]-b[i], hex(np.float64(a[i]).view(np.uint64)), hex(np.float64(b[i]).view(np.uint64)))
    # direct
    ru=ResidentUnit.load(u,NumpyRasterImage(m),NumpyRasterImage(r))
    f=ru.match(conf,0.4)
    a2=f["zncc_score"].to_numpy()
    print("   direct ResidentUnit.match bad:", int((~((a2==b)|(np.isnan(a2)&np.isnan(b)))).sum()))
