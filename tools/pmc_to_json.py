#!/usr/bin/env python3
"""Fold the passes of tools/pmc_collect.sh into profiles/pmc_traffic.json.

    python tools/pmc_to_json.py gpurun_out/pmc_<tag> <size> [--config3]

Per STAGE of the bench line (the keys `bench.py` looks up): HBM bytes per pair = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over
every launch of the stage's kernels, divided by the number of pairs profiled (gfx950: FETCH_SIZE counts 64 B per 128-B request,
MI355X_MICROARCH.md; WRITE_SIZE in KB), plus VALU / LDS wave-instructions, waves and the VALU pipe utilisation
4 * SQ_INSTS_VALU / (1024 SIMDs * kernel cycles).  Every entry carries the commit it was measured at.
"""
import collections
import csv
import json
import os
import sys

STAGES = {   # stage key -> kernel-name fragments
    "minmax": ("minmax_partial", "minmax_final"),            # (incl. the *_units_kernel forms of a batched submission)
    "stretch_laplacian_mask": ("lap_march", "lap_kernel", "sum_u32", "valid_sum_units"),
    "min_eigen_candidates_fused": ("eig3_kernel", "eig3_units_kernel", "eig2_kernel", "eig3_max"),
    "rank_select": ("f_hist_cut", "f_scatter_cells", "f_sweep", "f_acc_", "tk_hist", "f_cut", "f_cells"),
    "pyramid": ("pyrdown_kernel", "pyrdown_units_kernel"),
    "lk_fwd_bwd": ("lk2_kernel", "lk2_units_kernel", "lk_kernel"),
    "fb_frame": ("fb_compact", "fb_place", "fb_gather"),
    "zncc": ("zncc_kernel", "zncc_int_kernel", "zncc_int_units_kernel"),
    "phase_correlation_f32": ("fft_rows", "fft61_", "transpose_kernel", "cross_power_f32", "argmax_f32", "fft_"),
    "shift_image": ("shift_kernel", "shift_rows_kernel"),
    "mi_kernel": ("mi_kernel", "mi_int_kernel", "mi_int_units_kernel"),
    "dn_keep": ("dn_keep_kernel",),
    "phase_correlation_f64": ("f64_prime_kernel", "f64_smooth_kernel", "f64_cross_kernel", "f64_best_reduce", "f64_pack", "f64_absmax", "f64_first_index"),
    "phase_f64_prime_level": ("f64_prime_kernel",),
    "phase_f64_smooth_level": ("f64_smooth_kernel",),
    "phase_f64_cross_power": ("f64_cross_kernel",),
}
ONCE_PER_PAIR = {"config2": "lk2_kernel", "config2_batched": "lk2_units_kernel", "config3": "f61_top2_reduce", "scoring": "mi_int_kernel", "dn": "dn_keep_kernel", "f64": "f64_best_reduce"}


def short_name(kernel: str) -> str:
    """'void (anonymous namespace)::eig3_kernel<15>(unsigned char const*, ...)' -> 'eig3_kernel<15>': the full kernel name with its
    template arguments, without return type, anonymous namespace and parameter list."""
    n = kernel.strip()
    if n.startswith("void "):
        n = n[5:]
    n = n.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(n):          # the parameter list starts at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.strip()[:120]


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(name, r["Counter_Name"])] += 1
    return per, launches


def durations(trace):
    d = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(trace)):
        d[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        n[r["Kernel_Name"]] += 1
    return d, n


def main():
    root, size = sys.argv[1], sys.argv[2]
    mode = "config3" if "--config3" in sys.argv else "config2"
    for arg in sys.argv[3:]:
        if arg.startswith("--mode="):
            mode = arg.split("=", 1)[1]          # which kernel counts the units profiled (ONCE_PER_PAIR)
    units_per_launch = next((int(a.split("=", 1)[1]) for a in sys.argv[3:] if a.startswith("--units-per-launch=")), 1)   # batched submissions: pairs one marker launch serves
    only = [a.split("=", 1)[1].split(",") for a in sys.argv[3:] if a.startswith("--only=")]
    only = only[0] if only else None
    commit = open(os.path.join(root, "commit.txt")).read().strip()
    # Every pass is a run of its own and the workload's settle phases are time-bound: the passes differ in the number of launches.
    # Counters are therefore averaged PER LAUNCH inside their own pass and scaled by the kernel's launches per unit of the trace pass.
    merged = collections.defaultdict(lambda: collections.defaultdict(float))     # kernel -> counter -> mean per launch
    launch_count = collections.Counter()                                          # kernel -> launches in the trace pass (v)
    for p in ("f", "w", "v"):
        per, launches = load(os.path.join(root, p, "p_counter_collection.csv"))
        for name, cs in per.items():
            for c, v in cs.items():
                merged[name][c] = v / max(1, launches[(name, c)])
        if p == "v":
            for (name, c), k in launches.items():
                launch_count[name] = max(launch_count[name], k)
    dur, dn = durations(os.path.join(root, "v", "p_kernel_trace.csv"))
    marker = next((n for n in launch_count if ONCE_PER_PAIR[mode] in n), None)
    pairs = launch_count[marker] * units_per_launch if marker else 1
    for name in list(merged):                       # mean per launch -> per unit (pair / call)
        scale = launch_count.get(name, 0) / pairs
        for c in merged[name]:
            merged[name][c] *= scale
    pairs_for_sums = 1                              # (the sums below are per unit already)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    db = json.load(open(dst)) if os.path.exists(dst) else {}
    for stage, frags in STAGES.items():
        if only is not None and stage not in only:
            continue
        names = [n for n in merged if any(f in n for f in frags)]
        if not names:
            continue
        tot = collections.defaultdict(float)
        for n in names:
            for c, v in merged[n].items():
                tot[c] += v
        fetch, write = tot.get("FETCH_SIZE", 0.0), tot.get("WRITE_SIZE", 0.0)
        us = sum(dur[n] for n in names) / pairs
        cycles = tot.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # the counter sums the 8 XCDs
        valu = tot.get("SQ_INSTS_VALU", 0.0)
        entry = {str(size): int(round((2 * fetch + write) * 1024)), "measured_at": commit,
                 "_detail": {"kernels": sorted({short_name(n) for n in names}), "launches_per_pair": round(sum(launch_count[n] for n in names) / pairs, 2),
                             "pairs_profiled": pairs, "FETCH_SIZE_KB": round(fetch), "WRITE_SIZE_KB": round(write),
                             "SQ_INSTS_VALU": round(valu), "SQ_INSTS_LDS": round(tot.get("SQ_INSTS_LDS", 0.0)),
                             "SQ_WAVES": round(tot.get("SQ_WAVES", 0.0)), "kernel_cycles": round(cycles),
                             "kernel_us_under_counters": round(us, 1),
                             "valu_pipe_busy": round(4 * valu / (1024 * cycles), 3) if cycles else None,
                             "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per pair (gfx950: FETCH_SIZE counts 64 B per 128-B request); separate rocprofv3 "
                                     "--pmc passes (tools/pmc_collect.sh); valu_pipe_busy = 4 * SQ_INSTS_VALU / (1024 SIMDs * kernel cycles)"}}
        db[stage] = entry
        print(stage, entry[str(size)], entry["_detail"]["valu_pipe_busy"], entry["_detail"]["kernel_us_under_counters"])
    json.dump(db, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
