"""Where a rank's host CPU goes in the batched headline loop: CPU time of the worker's frame wait against its wall time (a spinning wait
shows cpu == wall), per-thread CPU of the process.  python tools/investigations/host_cpu_probe.py  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from karios_amd import synth, resident
from karios_amd._lib import Context
from karios_amd.core import KLTConfiguration
from karios_amd.resident import ResidentPair, PendingBatch
from karios_amd.stream import FrameStream
S = 10980
dev = torch.device("cuda", 0)
ctx = Context(0)
mon, ref = synth.make_pair_torch(S, S, 0.5, 0.25, device=dev); torch.cuda.synchronize()
pair = ResidentPair.from_device_pointers(mon.data_ptr(), ref.data_ptr(), np.uint16, S, S, ctx=ctx, keepalive=(mon, ref))
conf = KLTConfiguration()
acc = {"wait_cpu": 0.0, "wait_wall": 0.0}
orig = PendingBatch.wait
def wait(self):
    import threading
    acc["tid"] = threading.get_native_id()
    c, w = time.thread_time(), time.perf_counter()
    r = orig(self)
    acc["wait_cpu"] += time.thread_time() - c; acc["wait_wall"] += time.perf_counter() - w
    return r
PendingBatch.wait = wait
with FrameStream(0.4, depth=2) as s:
    def go(n):
        for _ in range(n):
            s.submit_many([(pair, None, None)] * 4, conf)
        s.drain(); ctx.sync()
    go(4)
    acc["wait_cpu"] = acc["wait_wall"] = 0.0; s.worker_cpu_s = 0.0
    def threads():
        out = {}
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open(f"/proc/self/task/{tid}/stat").read()
                name = f[f.index("(") + 1:f.rindex(")")]
                fields = f[f.rindex(")") + 2:].split()
                out[tid] = (name, (int(fields[11]) + int(fields[12])) / os.sysconf("SC_CLK_TCK"))
            except Exception:
                pass
        return out
    th0 = threads()
    p0, t0 = time.process_time(), time.perf_counter()
    go(20)
    th1 = threads()
    busy = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1], th1[t][0], t) for t in th1), reverse=True)[:8]
    # which library does the busiest OTHER thread execute in?  (instruction pointer from /proc, resolved against the process map)
    other = [t for b, n, t in busy if t != str(acc.get("tid")) and t != str(os.getpid())][:1]
    if other:
        maps = []
        for line in open("/proc/self/maps"):
            f = line.split()
            if len(f) >= 6 and "x" in f[1]:
                lo, hi = (int(v, 16) for v in f[0].split("-"))
                maps.append((lo, hi, f[5]))
        import threading
        def sampler():
            seen = {}
            for _ in range(200):
                try:
                    st = open(f"/proc/self/task/{other[0]}/stat").read()
                    ip = int(st[st.rindex(")") + 2:].split()[27])
                    lib = next((m[2] for m in maps if m[0] <= ip < m[1]), hex(ip))
                    seen[lib] = seen.get(lib, 0) + 1
                    sy = open(f"/proc/self/task/{other[0]}/syscall").read().split()[0]
                    seen["syscall " + sy] = seen.get("syscall " + sy, 0) + 1
                except Exception as e:
                    seen[repr(e)[:60]] = seen.get(repr(e)[:60], 0) + 1
                time.sleep(0.0005)
            print("other busy thread", other[0], "samples:", seen)
        th = threading.Thread(target=sampler); th.start(); go(20); th.join()
    print("main tid", os.getpid(), "waiting thread tid", acc.get("tid"))
    print("threads (cpu s over the window, name, tid):", [(round(b, 3), n, t) for b, n, t in busy if b > 0], "window s", round(time.perf_counter() - t0, 3))
    dt = time.perf_counter() - t0
    print("ms/pair", dt / 80 * 1e3, "worker cpu ms/pair", s.worker_cpu_s / 80 * 1e3, "wait cpu ms/pair", acc["wait_cpu"] / 80 * 1e3, "wait wall ms/pair", acc["wait_wall"] / 80 * 1e3,
          "process cpu ms/pair", (time.process_time() - p0) / 80 * 1e3)
