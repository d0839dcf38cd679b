import os, sys, time
sys.path.insert(0,'.')
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
os.system("lscpu | egrep 'Model name|^CPU\\(s\\)|Thread|Socket|NUMA node\\(s\\)'")
import numpy as np
from oracle import oracle as O
O.build()
from karios_amd import synth
mon, ref = synth.make_pair(2048, 4096, 0.5, 0.25)
conf = O.default_conf(maxCorners=3000)
small = synth.make_pair(300, 300, 0.5, 0.25)
MAXT=O.max_threads(); print('omp max', MAXT)
for n in (1, 4, 8, 16, 32, 64, 128):
    if n > MAXT: break
    O.set_threads(n)
    t=time.time(); e=O.klt_tile(mon, ref, conf); dt=time.time()-t
    t=time.time(); e2=O.klt_tile(small[0], small[1], conf); dt2=time.time()-t
    print(f"threads {n:4d}: 2048x4096 tile {dt:.3f} s ({2048*4096/1e6/dt:.1f} Mpx/s)   300x300 tile {dt2:.3f} s", flush=True)
