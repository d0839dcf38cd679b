#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats.csv into a short
table of this library's kernels (torch kernels of the synthetic-data generator are dropped).

usage: summarize_rocprof.py <kernel_stats.csv> <out.md> [steps_profiled]
"""
import csv
import re
import sys

OURS = ("lk2_", "valid_sum_units", "pyrdown_units", "fft61", "f61_", "eig3_", "f_", "kf_header", "fb_place", "eig2_", "eigc_", "dn_keep", "fb_count", "mi_kernel", "mi_int_kernel", "row_checksum", "lap_kernel", "lap_march", "eig_kernel", "eig_march", "tk_", "fb_", "pyr_", "cand_kernel", "lk_kernel", "pyrdown_kernel", "minmax_", "zncc_kernel", "zncc_int_kernel", "sel_", "select_kernel",
        "take_first", "sum_u32", "max_u32", "to_uint8", "auto_mask", "count_nonzero", "shift_kernel", "shift_rows_kernel", "rocprim", "cross_power",
        "absmax", "first_index", "to_f64", "f64_", "blue_", "fft", "stretch", "lut_", "transpose_kernel", "argmax_f32", "zncc_win")


def short(name: str) -> str:
    if "rocprim" in name:
        m = re.search(r"(radix_sort_\w+|scan_\w+|lookback_scan\w*|init_lookback\w*|onesweep\w*|histogram\w*)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = {}
    for r in csv.DictReader(open(src)):
        if not any(k in r["Name"] for k in OURS):
            continue
        k = short(r["Name"])
        e = rows.setdefault(k, [0, 0.0, 1e30, 0.0])
        e[0] += int(r["Calls"]); e[1] += float(r["TotalDurationNs"]); e[2] = min(e[2], float(r["MinNs"])); e[3] = max(e[3], float(r["MaxNs"]))
    tot = sum(v[1] for v in rows.values())
    with open(dst, "w") as f:
        f.write(f"source: {src.split('/')[-1]} (rocprofv3 --kernel-trace --stats), library kernels only")
        f.write(f"; {steps} warm-up+timed steps profiled\n\n" if steps else "\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % of library GPU time |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            f.write(f"| {k} | {v[0]} | {v[1]/1e6:.3f} | {v[1]/v[0]/1e3:.1f} | {v[2]/1e3:.1f} | {v[3]/1e3:.1f} | {100*v[1]/tot:.1f} |\n")
        if steps:
            f.write(f"\nlibrary GPU time per step: {tot/1e6/steps:.3f} ms\n")


if __name__ == "__main__":
    main()
