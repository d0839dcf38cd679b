import os, sys, time
import torch
n = 241_120_800
a = torch.empty(n, dtype=torch.uint8).pin_memory()
b = torch.empty(n, dtype=torch.uint8).pin_memory()
pa = torch.empty(n, dtype=torch.uint8)
d1 = torch.empty(n, dtype=torch.uint8, device="cuda")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def one():
    d1.copy_(a, non_blocking=True)
def two_streams():
    with torch.cuda.stream(s1): d1.copy_(a, non_blocking=True)
    with torch.cuda.stream(s2): d2.copy_(b, non_blocking=True)
def chunks4():
    q = n // 4
    for i, s in enumerate((s1, s2, s1, s2)):
        with torch.cuda.stream(s): d1[i*q:(i+1)*q].copy_(a[i*q:(i+1)*q], non_blocking=True)
def pageable():
    d1.copy_(pa)
print("HSA_ENABLE_SDMA", os.environ.get("HSA_ENABLE_SDMA"))
print("pinned 241MB one stream   %.2f ms  %.1f GB/s" % (t(one)*1e3, n/t(one)/1e9))
x = t(two_streams); print("pinned 2x241MB two streams %.2f ms  %.1f GB/s" % (x*1e3, 2*n/x/1e9))
x = t(chunks4); print("pinned 241MB 4 chunks/2 streams %.2f ms  %.1f GB/s" % (x*1e3, n/x/1e9))
x = t(pageable); print("pageable 241MB            %.2f ms  %.1f GB/s" % (x*1e3, n/x/1e9))
