#!/usr/bin/env python3
"""Mine the ONE real-OpenCV output the reference tree holds - the golden CSV of its end-to-end test
(`tests/end_to_end/ref_data/test_full/KLT_matcher_*.csv`, produced by `karios process` with
`tests/end_to_end/processing_configuration.json:10-20`: tile_size 6000, maxCorners 20000, minDistance 10, blocksize 15,
winsize 25, Laplacian k=5; the input rasters are stripped, `.MISSING_LARGE_BLOBS`) - for facts about cv2.goodFeaturesToTrack /
calcOpticalFlowPyrLK and the numpy glue around them that hold WITHOUT the input images.

Runs in the build container only (reads /root/reference).  Writes `tests/golden/e2e_csv_facts.npz`: aggregated facts as a
JSON string plus a row SAMPLE of the numeric columns (arrays - never the file's text) for the recomputation tests of
`tests/test_oracle_golden.py::test_real_opencv_*`.
"""
import glob
import json
import os

import numpy as np
import pandas as pd
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = glob.glob("/root/reference/tests/end_to_end/ref_data/test_full/KLT_matcher_*.csv")[0]
SIZE, TILE, MAXC, MIN_DIST, MARGIN = 10980, 6000, 20000, 10.0, 28


def main():
    df = pd.read_csv(SRC, sep=";", dtype=np.float32)            # float32 text (shortest repr) -> the float32 values the run wrote
    x, y = df["x0"].to_numpy(), df["y0"].to_numpy()
    tid = (x >= TILE).astype(int) * 2 + (y >= TILE).astype(int)  # tile number in the reference's x-outer / y-inner order
    facts = {"source": "tests/end_to_end/ref_data/test_full/" + os.path.basename(SRC), "rows": int(len(df)), "columns": list(df.columns),
             "image_size": SIZE, "tile_size": TILE, "max_corners": MAXC, "min_distance": MIN_DIST,
             "tile_sequence_in_file": [int(tid[0])] + [int(tid[i + 1]) for i in np.flatnonzero(np.diff(tid) != 0)],
             "corners_are_integers": bool(np.all(x == np.round(x)) and np.all(y == np.round(y))), "tiles": []}
    for t in range(4):
        m = tid == t
        ox, oy = (t // 2) * TILE, (t % 2) * TILE
        w, h = min(TILE, SIZE - ox), min(TILE, SIZE - oy)
        xs, ys = x[m].astype(np.float64), y[m].astype(np.float64)
        nn = cKDTree(np.c_[xs, ys]).query(np.c_[xs, ys], k=2)[0][:, 1]
        facts["tiles"].append({"tile": t, "origin": [ox, oy], "size": [w, h], "rows": int(m.sum()),
                               "sorted_by_x0_y0": bool(np.array_equal(np.lexsort((ys, xs)), np.arange(m.sum()))),
                               "local_x_range": [float(xs.min() - ox), float(xs.max() - ox)],
                               "local_y_range": [float(ys.min() - oy), float(ys.max() - oy)],
                               "nearest_neighbour_min": float(nn.min()), "nearest_neighbour_equal_min_distance": int((nn == MIN_DIST).sum()),
                               "nearest_neighbour_below_min_distance": int((nn < MIN_DIST).sum()),
                               "smallest_distances": [float(v) for v in np.unique(nn)[:4]]})
    # rows whose local coordinate is a power of two (the float32 grid below it is finer), the two rows at y0 == SIZE - MARGIN
    # (scored by the run that wrote the CSV, NaN under today's `>=` rule, zncc_service.py:212-215), and every 8th row
    xl, yl = np.where(x >= TILE, x - TILE, x), np.where(y >= TILE, y - TILE, y)
    pow2 = (np.log2(np.maximum(xl, 1)) % 1 == 0) | (np.log2(np.maximum(yl, 1)) % 1 == 0)
    edge = (y == SIZE - MARGIN) | (x == SIZE - MARGIN)
    take = np.zeros(len(df), bool)
    take[::8] = True
    take |= pow2 | edge
    facts["sample_rows"] = int(take.sum())
    facts["zncc_nan_rows"] = int(df["zncc_score"].isna().sum())
    facts["rows_with_score_at_least_0.4"] = int((df["score"].to_numpy() >= np.float32(0.4)).sum())
    s = df[take]
    np.savez_compressed(os.path.join(HERE, "e2e_csv_facts.npz"), facts=json.dumps(facts),
                        row=np.flatnonzero(take).astype(np.int32), tile=tid[take].astype(np.int8),
                        **{c.replace(" ", "_"): s[c].to_numpy() for c in df.columns})
    print(json.dumps(facts, indent=1))


if __name__ == "__main__":
    main()
