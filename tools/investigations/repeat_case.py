#!/usr/bin/env python3
"""Run ONE case of tools/fuzz_parity.py many times on the GPU and compare every run with the oracle's (single) result:
hunts timing-dependent defects.    python tools/investigations/repeat_case.py --seed 40917 --max-size 1200 --reps 3000"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tools/: fuzz_parity
import fuzz_parity as F   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--max-size", type=int, default=700)
    ap.add_argument("--reps", type=int, default=1000)
    a = ap.parse_args()
    from oracle import oracle as O
    O.build()
    from karios_amd import ops
    from karios_amd.resident import ResidentPair
    case = F.draw_case(a.seed, a.max_size)
    mon, ref, mask = F.make_inputs(case)
    conf = O.default_conf(maxCorners=case["maxCorners"], blocksize=case["blocksize"], matching_winsize=case["winsize"],
                          qualityLevel=case["qualityLevel"], minDistance=case["minDistance"],
                          laplacian_kernel_size={"mon": case["mon_k"], "ref": case["ref_k"]}, laplacian_invert_polarity=case["invert"])
    box = case.get("box")
    x_off, y_off, bx, by = box if box is not None else (0, 0, case["W"], case["H"])
    sl = (slice(y_off, y_off + by), slice(x_off, x_off + bx))
    mon_b, ref_b, mask_b = mon[sl], ref[sl], (None if mask is None else mask[sl])
    exp = O.klt_tile(np.ascontiguousarray(mon_b), np.ascontiguousarray(ref_b), conf,
                     mask_box=None if mask_b is None else np.ascontiguousarray(mask_b), nodata_mon=case["nodata_mon"],
                     nodata_ref=case["nodata_ref"], x_off=x_off, y_off=y_off, invert_mon=case["invert"])
    assert exp is not None, "pick a case with corners"
    p0e = O.good_features(exp["lap_ref"], exp["mask"], conf.maxCorners, conf.qualityLevel, conf.minDistance, conf.blocksize)
    p1e = O.pyr_lk(exp["lap_ref"], exp["lap_mon"], p0e, case["winsize"])
    p0re = O.pyr_lk(exp["lap_mon"], exp["lap_ref"], p1e, case["winsize"])
    pair = ResidentPair.upload(mon, ref, mask=mask)
    pair.no_data_mon, pair.no_data_ref = case["nodata_mon"], case["nodata_ref"]
    bad = 0
    t0 = time.time()
    for it in range(a.reps):
        status, tr = ops.klt_tile(ref_b, mon_b, conf, mask_box=mask_b, nodata_ref=case["nodata_ref"], nodata_mon=case["nodata_mon"],
                                  mon_ksize=case["mon_k"], ref_ksize=case["ref_k"], invert_mon=case["invert"])
        frame = pair.match_tile(conf, box=box, zncc_threshold=0.4)
        msgs = []
        if status != "ok" or tr[0].shape != p0e.shape or not np.array_equal(tr[0], p0e):
            msgs.append("host path p0")
        elif not np.array_equal(tr[1], p1e) or not np.array_equal(tr[2], p0re):
            msgs.append("host path tracks")
        if frame is None or len(frame) != len(exp["x0"]):
            msgs.append(f"frame rows {None if frame is None else len(frame)} vs {len(exp['x0'])}")
        else:
            for col in ("x0", "y0", "dx", "dy", "score"):
                if not np.array_equal(frame[col].to_numpy(), exp[col]):
                    msgs.append(f"frame {col}")
        if msgs:
            bad += 1
            print(f"rep {it}: {msgs}", flush=True)
            if bad == 1:
                np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"repeat_fail_{a.seed}.npz"),
                                    gpu_p0=tr[0] if status == "ok" else np.zeros(0), exp_p0=p0e,
                                    **({f"gpu_{c}": frame[c].to_numpy() for c in frame.columns} if frame is not None else {}),
                                    **{f"exp_{c}": exp[c] for c in ("x0", "y0", "dx", "dy", "score")})
    print(f"repeat_case seed {a.seed}: {a.reps} reps, {bad} mismatching, {time.time() - t0:.1f} s, case {case}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
