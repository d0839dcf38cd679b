for i in 1 2; do
  for lib in libkarios_hip_fftA.so libkarios_hip.so; do
    KARIOS_HIP_LIB=$PWD/karios_amd/$lib python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.resident import ResidentPair
S = 10980
dev = torch.device("cuda", 0)
mon_t, ref_t = synth.make_pair_torch(S, S, 37.25, -20.75, device=dev)
torch.cuda.synchronize()
ctx = Context(0)
pair = ResidentPair.from_device_pointers(mon_t.data_ptr(), ref_t.data_ptr(), np.uint16, S, S, ctx=ctx)
ctx.set_profiling(True)
ts = []
for _ in range(8):
    off = pair.phase_offset()
    ts.append(ctx.stage_ms().get("phase_correlation", 0.0))
print(os.path.basename(os.environ["KARIOS_HIP_LIB"]), "phase ms min %.3f median %.3f" % (min(ts), sorted(ts)[len(ts)//2]), off)
PY
  done
done
