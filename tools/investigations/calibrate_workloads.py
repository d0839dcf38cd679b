#!/usr/bin/env python3
"""CPU calibration of the non-best-case bench workloads (karios_amd.synth.make_hard_pair_torch / make_tie_heavy_pair_torch) with the
oracle: forward-backward survival, LK iteration histograms per level and direction, how binary the Laplacians are, how many exact
ties the candidate list holds.  `python tools/investigations/calibrate_workloads.py hard --size 3072 --mix 0.45 --noise 120`"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def lk_report(O, lap_ref, lap_mon, p0, win=25):
    p1, f0, f1 = O.pyr_lk(lap_ref, lap_mon, p0, win, return_iters="levels")
    p0r, b0, b1 = O.pyr_lk(lap_mon, lap_ref, p1, win, return_iters="levels")
    d = np.abs(p0 - p0r).reshape(-1, 2).max(-1)
    keep = d < np.float32(0.1)
    out = {"corners": int(len(p0)), "survive": int(keep.sum()), "survival": float(keep.mean())}
    for name, it in (("fwd_L1", f1), ("fwd_L0", f0), ("bwd_L1", b1), ("bwd_L0", b0)):
        h = np.bincount(it, minlength=31)[:31]
        out[name] = {"mean": float(it.mean()), "p50": int(np.median(it)), "p90": int(np.percentile(it, 90)), "at_cap_30": int(h[30]), "hist_0_30": h.tolist()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=("hard", "tie", "plain"))
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--mix", type=float, default=0.555)
    ap.add_argument("--noise", type=float, default=200.0)
    ap.add_argument("--warp", type=float, default=0.6)
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--period", type=int, default=96)
    ap.add_argument("--ksize", type=int, default=7)
    a = ap.parse_args()
    import torch
    from karios_amd import synth
    from oracle import oracle as O
    S = a.size
    if a.kind == "hard":
        mon, ref = synth.make_hard_pair_torch(S, S, mix=a.mix, noise_sigma=a.noise, warp=a.warp, device="cpu")
    elif a.kind == "tie":
        mon, ref = synth.make_tie_heavy_pair_torch(S, S, levels=a.levels, period=a.period, device="cpu")
    else:
        mon, ref = synth.make_pair_torch(S, S, device="cpu")
    mon, ref = mon.numpy().view(np.uint16), ref.numpy().view(np.uint16)
    conf = O.default_conf(maxCorners=max(50, int(20000 * (S / 10980.0) ** 2)), laplacian_kernel_size=a.ksize)
    lap_ref, lap_mon = O.laplacian_u8(O.to_uint8(ref), a.ksize), O.laplacian_u8(O.to_uint8(mon), a.ksize)
    binary = float(((lap_ref == 0) | (lap_ref == 255)).mean())
    eig = O.min_eigen(lap_ref, conf.blocksize)
    p0 = O.good_features(lap_ref, None, conf.maxCorners, conf.qualityLevel, conf.minDistance, conf.blocksize)
    rep = {"size": S, "maxCorners": conf.maxCorners, "laplacian_0_or_255": binary, "eig_distinct_over_pixels": float(len(np.unique(eig)) / eig.size)}
    if p0 is not None:
        v = eig[p0[:, 0, 1].astype(int), p0[:, 0, 0].astype(int)]
        rep["corner_values_distinct"] = int(len(np.unique(v)))
        rep.update(lk_report(O, lap_ref, lap_mon, p0, conf.matching_winsize))
    import json
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
