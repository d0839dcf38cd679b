#!/usr/bin/env python3
"""Per-kernel mean of each PMC counter from a rocprofv3 counter_collection.csv (library kernels only)."""
import csv, sys, collections, re
KEEP = ("eig3_kernel", "eig2_kernel", "f_sweep", "f_acc_emit", "eigc_kernel", "lap_kernel", "lap_march", "eig_kernel", "eig_march", "tk_", "fb_", "cand_kernel", "lk_kernel", "lk2_kernel", "lk_order", "f_cut", "f_scatter", "f_cells", "f_acc", "tk_hist", "fb_compact", "fb_place", "argmax", "shift_kernel", "shift_rows_kernel", "pyrdown_kernel", "minmax_partial", "zncc_kernel", "sel_sweep", "fft_rows", "transpose_kernel", "cross_power_f32")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
    k = next((x for x in KEEP if x in name), None)
    if not k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (round(sum(v) / len(v)), len(v)) for c, v in sorted(d.items())})
