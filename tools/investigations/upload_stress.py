#!/usr/bin/env python3
"""Control experiment for the stale-input mismatch of the blocking host-buffer tile call (profiles/r03_fuzz_parity.md,
DESIGN 10.8; reference semantics at stake: klt.py:252-253, a tile is matched as read).

Several processes share the one GPU and keep every host core busy, as the unforced soaks did when the mismatch was seen:

    tile workers   km_klt_tile on pageable, row-strided boxes of changing pixel type and size (contexts are re-created now and
                   then, so workspace slots are freed / re-allocated / regrown all the time)
    aux workers    phase correlation (FFT-heavy kernels), ZNCC / MI batches, shift_image from host buffers
    burners        numpy loops that do nothing but occupy the remaining cores

No oracle runs: the detector is KARIOS_HIP_UPLOAD_CHECKSUM=1 (csrc/staging.hip) - a row-checksum kernel right behind every upload
on the same stream, compared with the host's checksum of the source when the call completes, and re-evaluated after the stream
has drained on a miss.  That is ~50x more tile calls per minute than an oracle-checked soak.

    python tools/investigations/upload_stress.py --mode old  --seconds 600     # the ORIGINAL upload: hipMemcpy2DAsync from pageable rows
    python tools/investigations/upload_stress.py --mode ring --seconds 300     # the library's page-locked staging ring (default build)

Writes gpurun_out/upload_stress_<mode>.json and one log per worker.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def tile_worker(seed: int, seconds: float, max_size: int):
    import numpy as np
    from karios_amd import ops, synth
    from karios_amd._lib import Context
    from karios_amd.core import KLTConfiguration
    rng = np.random.default_rng(seed)
    bases = []
    t0 = time.time()
    for k in range(10):                                         # a pool of rasters of every pixel type the boundary accepts
        H, W = int(rng.integers(300, max_size + 1)), int(rng.integers(300, max_size + 1))
        mon, ref = synth.make_pair(H, W, float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), seed=seed * 100 + k, noise_sigma=15.0)
        dt = (np.uint16, np.int16, np.int16, np.float32, np.uint8)[k % 5]
        if dt is np.uint8:
            mon, ref = (mon >> 5).astype(np.uint8), (ref >> 5).astype(np.uint8)
        elif dt is np.int16:
            mon, ref = (mon.astype(np.int32) - 4000).astype(np.int16), (ref.astype(np.int32) - 4000).astype(np.int16)
        elif dt is np.float32:
            mon, ref = mon.astype(np.float32) * np.float32(0.37), ref.astype(np.float32) * np.float32(0.37)
        mask = np.full((H, W), 255, np.uint8)
        mask[H // 3:H // 3 + 40, W // 4:W // 4 + 90] = 0
        bases.append((mon, ref, mask))
    calls = armed = missed = ctxs = 0
    ctx = None
    while time.time() - t0 < seconds:
        if ctx is None or calls % 150 == 0:
            if ctx is not None:
                a, m = ctx.upload_check_stats()
                armed += a; missed += m
                ctx.close()
            ctx = Context(); ctxs += 1
        mon, ref, mask = bases[int(rng.integers(len(bases)))]
        H, W = mon.shape
        if rng.random() < 0.75:                                  # inner box: row-strided views, odd offsets
            bx, by = int(rng.integers(40, W + 1)), int(rng.integers(40, H + 1))
            x0, y0 = int(rng.integers(0, W - bx + 1)), int(rng.integers(0, H - by + 1))
        else:
            x0, y0, bx, by = 0, 0, W, H
        sl = (slice(y0, y0 + by), slice(x0, x0 + bx))
        conf = KLTConfiguration(maxCorners=int(rng.choice([40, 500, 2000])), laplacian_kernel_size=int(rng.choice([3, 5, 7])))
        use_mask = rng.random() < 0.3
        ops.klt_tile(ref[sl], mon[sl], conf, mask_box=mask[sl] if use_mask else None, mon_ksize=conf.laplacian_kernel_size,
                     ref_ksize=conf.laplacian_kernel_size, ctx=ctx)
        calls += 1
    a, m = ctx.upload_check_stats()
    armed += a; missed += m
    print(json.dumps({"role": "tile", "seed": seed, "calls": calls, "uploads_checked": armed, "uploads_missed": missed, "contexts": ctxs,
                      "seconds": round(time.time() - t0, 1)}), flush=True)


def aux_worker(seed: int, seconds: float):
    import numpy as np
    from karios_amd import ops, synth
    from karios_amd._lib import default_context
    rng = np.random.default_rng(seed)
    base, _ = synth.make_pair(900, 900, 0.0, 0.0, seed=seed, noise_sigma=0.0)
    t0 = time.time()
    calls = 0
    while time.time() - t0 < seconds:
        H, W = int(rng.integers(64, 700)), int(rng.integers(64, 700))
        y, x = int(rng.integers(0, 900 - H)), int(rng.integers(0, 900 - W))
        a = base[y:y + H, x:x + W]
        b = np.roll(a, (int(rng.integers(-9, 10)), int(rng.integers(-9, 10))), (0, 1))
        ops.phase_cross_correlation(b, a)
        n = 200
        x0 = rng.integers(30, W - 30, n).astype(np.float32)
        y0 = rng.integers(30, H - 30, n).astype(np.float32)
        d = rng.uniform(-1, 1, (2, n)).astype(np.float32)
        ops.zncc_batch(a, b, x0, y0, d[0], d[1])
        ops.mi_batch(a, b, x0, y0, d[0], d[1])
        ops.shift_image(a, int(rng.integers(-20, 21)), int(rng.integers(-20, 21)))
        calls += 1
    armed, missed = default_context().upload_check_stats()
    print(json.dumps({"role": "aux", "seed": seed, "calls": calls, "uploads_checked": armed, "uploads_missed": missed,
                      "seconds": round(time.time() - t0, 1)}), flush=True)


def burner(seconds: float):
    import numpy as np
    rng = np.random.default_rng(os.getpid())
    a = rng.standard_normal((600, 600))
    t0 = time.time()
    while time.time() - t0 < seconds:
        a = np.sort(a @ a.T * 1e-3, axis=0)[:, ::-1].copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("old", "ring"), default="old")
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--tile-workers", type=int, default=4)
    ap.add_argument("--aux-workers", type=int, default=2)
    ap.add_argument("--burners", type=int, default=-1, help="-1: usable CPUs minus the workers (at least 2)")
    ap.add_argument("--max-size", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=4100)
    ap.add_argument("--role", choices=("tile", "aux", "burn"), default=None)
    a = ap.parse_args()
    if a.role == "tile":
        return tile_worker(a.seed, a.seconds, a.max_size)
    if a.role == "aux":
        return aux_worker(a.seed, a.seconds)
    if a.role == "burn":
        return burner(a.seconds)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, KARIOS_HIP_UPLOAD_CHECKSUM="1", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    if a.mode == "old":
        env["KARIOS_HIP_ASYNC_HOST_UPLOAD"] = "1"
    else:
        env.pop("KARIOS_HIP_ASYNC_HOST_UPLOAD", None)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 8
    ncpu = min(ncpu, 32)
    burners = a.burners if a.burners >= 0 else max(2, min(ncpu, 16) - a.tile_workers - a.aux_workers)
    procs = []
    me = [sys.executable, os.path.abspath(__file__), "--seconds", str(a.seconds), "--max-size", str(a.max_size)]
    for i in range(a.tile_workers):
        log = open(os.path.join(out, f"upload_stress_{a.mode}_tile{i}.log"), "w")
        procs.append(("tile", subprocess.Popen(me + ["--role", "tile", "--seed", str(a.seed + i)], env=env, stdout=log, stderr=subprocess.STDOUT), log))
    for i in range(a.aux_workers):
        log = open(os.path.join(out, f"upload_stress_{a.mode}_aux{i}.log"), "w")
        procs.append(("aux", subprocess.Popen(me + ["--role", "aux", "--seed", str(a.seed + 100 + i)], env=env, stdout=log, stderr=subprocess.STDOUT), log))
    for i in range(burners):
        procs.append(("burn", subprocess.Popen(me + ["--role", "burn"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL), None))
    rc = 0
    for role, p, log in procs:
        try:
            r = p.wait(timeout=a.seconds + 300)
        except subprocess.TimeoutExpired:
            p.kill(); r = -9
        if log:
            log.close()
        if r != 0 and role != "burn":
            rc = 1
    total = {"mode": a.mode, "seconds": a.seconds, "tile_workers": a.tile_workers, "aux_workers": a.aux_workers, "burners": burners,
             "tile_calls": 0, "aux_calls": 0, "uploads_checked": 0, "uploads_missed": 0, "miss_lines": [], "worker_failed": rc != 0}
    for name in sorted(os.listdir(out)):
        if not name.startswith(f"upload_stress_{a.mode}_") or not name.endswith(".log"):
            continue
        for line in open(os.path.join(out, name), errors="replace"):
            if line.startswith("{"):
                try:
                    d = json.loads(line)
                except ValueError:
                    continue
                total["tile_calls" if d["role"] == "tile" else "aux_calls"] += d["calls"]
                total["uploads_checked"] += d["uploads_checked"]
                total["uploads_missed"] += d["uploads_missed"]
            elif "UPLOAD_CHECKSUM MISS" in line:
                total["miss_lines"].append(line.strip()[:600])
    json.dump(total, open(os.path.join(out, f"upload_stress_{a.mode}.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in total.items() if k != "miss_lines"}), flush=True)
    for m in total["miss_lines"][:20]:
        print(m)
    return rc


if __name__ == "__main__":
    sys.exit(main() or 0)
