#!/usr/bin/env python3
"""Randomised self-consistency sweep of the software-pipelined batched submissions (csrc/api_units.hip, "units_pipeline"): every round
draws a few resident pairs (size, pixel type, shift, no-data wedge, with or without a user mask), a configuration (Laplacian kernel
<= 7, blockSize, winSize, maxCorners, qualityLevel, minDistance >= 1) and a stream of batched submissions of random boxes, runs the
stream through FrameStream (pipeline on, depth 1 or 2, every third submission artificially flagged) and compares every unit's frame
block with the same unit submitted ALONE through the exact path (pipeline off) - bit for bit, score columns included.  The single-unit
path itself is what tools/fuzz_parity.py holds against the oracle.

    python tools/fuzz_pipeline.py --rounds 40 --seed 1 [--budget-s 120]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


KSIZES = [int(v) for v in os.environ["KARIOS_FUZZ_KSIZES"].split(",")] if os.environ.get("KARIOS_FUZZ_KSIZES") else [1, 3, 5, 7, 7, 9, 11]

def same_rows(a, b) -> bool:
    ia, ib = a.block.view(np.int32), b.block.view(np.int32)
    # (header word 3 - the candidate count of the synchronisation-free corner path - is a diagnostic: 0 for a unit that was repeated exactly)
    if not np.array_equal(ia[:3], ib[:3]) or a.cap != b.cap or a.with_zncc != b.with_zncc:
        return False
    n, cap = int(ia[0]), a.cap
    ok = all(np.array_equal(ia[4 + k * cap:4 + k * cap + n], ib[4 + k * cap:4 + k * cap + n]) for k in range(6))
    base = 4 + 6 * cap
    return ok and all(np.array_equal(ia[base + 2 * k * cap:base + 2 * k * cap + 2 * n], ib[base + 2 * k * cap:base + 2 * k * cap + 2 * n])
                      for k in range(int(a.with_zncc)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=0.0)
    a = ap.parse_args()
    from karios_amd import _lib, synth
    from karios_amd.core import KLTConfiguration
    from karios_amd.resident import ResidentPair
    from karios_amd.stream import FrameStream
    ctx = _lib.default_context()
    t_start, units_checked, bad, rounds_done, redone_total = time.time(), 0, 0, 0, 0
    for rnd in range(a.rounds):
        if a.budget_s and time.time() - t_start > a.budget_s:
            break
        rng = np.random.default_rng(7919 * (a.seed + rnd) + 3)
        dtype = [np.uint16, np.uint16, np.uint8, np.int16][rng.integers(4)]
        masked = bool(rng.random() < 0.35)
        pairs = []
        for _ in range(int(rng.integers(1, 4))):
            H, W = int(rng.integers(300, 1300)), int(rng.integers(560, 1500))
            mon, ref = synth.make_pair(H, W, float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-1.2, 1.2)), seed=int(rng.integers(1 << 30)),
                                       nodata_wedge=bool(rng.random() < 0.3))
            if dtype == np.uint8:
                mon, ref = (mon >> 6).astype(np.uint8), (ref >> 6).astype(np.uint8)
            elif dtype == np.int16:
                mon, ref = (mon.astype(np.int32) - 9000).astype(np.int16), (ref.astype(np.int32) - 9000).astype(np.int16)
            mask = None
            if masked:
                mask = np.ones((H, W), np.uint8)
                for _k in range(int(rng.integers(2, 9))):
                    y, x = int(rng.integers(0, H - 20)), int(rng.integers(0, W - 20))
                    mask[y:y + int(rng.integers(20, 300)), x:x + int(rng.integers(20, 400))] = 0
            pairs.append(ResidentPair.upload(mon, ref, mask, ctx=ctx))
        conf = KLTConfiguration(maxCorners=int(rng.choice([40, 500, 3000])), laplacian_kernel_size=int(rng.choice(KSIZES)),
                                blocksize=int(rng.choice([3, 5, 7, 9, 15, 15])), matching_winsize=int(rng.choice([9, 15, 21, 25, 25, 31])),
                                qualityLevel=float(rng.choice([0.01, 0.1, 0.1, 0.3])), minDistance=int(rng.choice([1, 3, 10, 10, 14])),
                                laplacian_invert_polarity=bool(rng.random() < 0.2))
        thr = None if rng.random() < 0.2 else 0.4
        mi = thr is not None and bool(rng.random() < 0.4)

        def draw_box(p):
            if rng.random() < 0.25:
                return None
            w = int(rng.integers(520, p.x_size + 1))
            h = int(rng.integers(max(2 * conf.blocksize + 8, 2 * conf.matching_winsize + 2, 64), p.y_size + 1))
            return (int(rng.integers(0, p.x_size - w + 1)), int(rng.integers(0, p.y_size - h + 1)), w, h)

        subs = []
        for _ in range(int(rng.integers(3, 8))):
            units = []
            for _u in range(int(rng.integers(2, 9))):
                p = pairs[rng.integers(len(pairs))]
                units.append((p, draw_box(p), None))
            subs.append(units)
        # reference: every unit alone through the exact path, pipeline off
        ctx.set_option("units_pipeline", 0)
        want = [[p.submit_tile(conf, box=b, zncc_threshold=thr, origin=o, mutual_info=mi).result() for p, b, o in units] for units in subs]
        got = []
        with FrameStream(thr, depth=int(rng.integers(1, 3)), mutual_info=mi) as s:
            for k, units in enumerate(subs):
                ctx.set_option("spec_flag", 32 if k % 3 == 2 else 0)
                got += s.submit_many(units, conf)
                ctx.set_option("spec_flag", 0)
            got += s.drain()
            redone_total += s.units_redone
        flat = [w for sub in want for w in sub]
        if len(got) != len(flat):
            print(f"FAIL round {rnd} (seed {a.seed + rnd}): {len(got)} results for {len(flat)} units", flush=True)
            bad += 1
            continue
        for k, (g, w) in enumerate(zip(got, flat)):
            units_checked += 1
            if not same_rows(g.raw, w):
                bad += 1
                print(f"FAIL round {rnd} (seed {a.seed + rnd}) unit {k}: dtype {np.dtype(dtype).name} masked {masked} conf k={conf.laplacian_kernel_size} "
                      f"block={conf.blocksize} win={conf.matching_winsize} maxCorners={conf.maxCorners} thr={thr} mi={mi} "
                      f"hdr {g.raw.block[:4].view(np.int32).tolist()} vs {w.block[:4].view(np.int32).tolist()}", flush=True)
        rounds_done += 1
        del pairs
    print(f"fuzz_pipeline: {rounds_done} rounds, {units_checked} units compared, {redone_total} repeated exactly, {bad} FAILED, "
          f"{time.time() - t_start:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
