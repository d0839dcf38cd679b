#!/usr/bin/env python3
"""Does it matter which socket the process runs on?  PIN=local|remote|none: the process (before any thread or page-locked buffer exists) is
bound to the CPUs of the GPU's NUMA node (sysfs local_cpulist of its PCI device), to the other CPUs, or left alone; then the bench's
end_to_end object (482 MB per pair from page-locked host rasters) and the headline's loop at 5 and 15 submissions."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out.update(range(int(a), int(b) + 1))
        elif part:
            out.add(int(part))
    return out


def gpu_local_cpus(index=0):
    import torch
    p = torch.cuda.get_device_properties(index)          # (does not initialise the GPU)
    bus = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    path = f"/sys/bus/pci/devices/{bus}/local_cpulist"
    return cpulist(open(path).read()), bus


pin = os.environ.get("PIN", "none")
if pin != "none":
    local, bus = gpu_local_cpus(0)
    allc = os.sched_getaffinity(0)
    want = (local & allc) if pin == "local" else (allc - local)
    os.sched_setaffinity(0, want)
    print(f"PIN={pin}: GPU {bus}, {len(want)} CPUs", flush=True)
import numpy as np
import torch
from benchkit import legs
from karios_amd import synth
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair
from karios_amd.stream import FrameStream

S = 10980
dev = torch.device("cuda", 0)
ctx = Context(0)
conf = KLTConfiguration()
data = [synth.make_pair_torch(S, S, 0.5, 0.25, seed=20260101 + 10 * b, device=dev) for b in range(4)]
torch.cuda.synchronize()
pairs = [ResidentPair.from_device_pointers(m.data_ptr(), r.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(m, r)) for m, r in data]
for subs in (5, 15):
    with FrameStream(0.4, depth=2) as s:
        def go(n):
            for _ in range(n):
                s.submit_many([(p, None, None) for p in pairs], conf)
            s.drain()
            ctx.sync()
        go(6)
        w = []
        for _ in range(5):
            t0 = time.perf_counter()
            go(subs)
            w.append((time.perf_counter() - t0) / (subs * 4) * 1e3)
    print(f"PIN={pin}: {subs} submissions per window: {sorted(w)[2]:.4f} ms per pair {[round(v, 4) for v in w]}", flush=True)
mon, ref = data[0][0].cpu().numpy().view(np.uint16), data[0][1].cpu().numpy().view(np.uint16)
del data, pairs
print(f"PIN={pin}: end_to_end {legs.end_to_end(mon, ref, ctx, 10)['ms_per_pair']:.3f} ms per pair", flush=True)
