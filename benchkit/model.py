"""Byte models of the stages and the roofline objects built from stage spans."""
from __future__ import annotations

import json
import os
import statistics
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# algorithmic HBM bytes per pixel of one pair, per dense stage
STAGE_BYTES_PER_PX = {
    "minmax": 4.0,                   # read mon 2 + ref 2
    "stretch_laplacian_mask": 7.0,   # read 2+2, write lap_mon 1 + lap_ref 1 + mask 1
    "min_eigen": 5.0,                # read lap_ref 1, write eig 4            (two-kernel path only)
    "candidates": 5.0,               # read eig 4 + mask 1                    (two-kernel path only)
    "pyramid": 2.5,                  # read 1+1, write 1/4+1/4
}
FUSED_EIG_BYTES_PER_PX = 2.0         # fused K3+K4: read lap_ref 1 + mask 1; the eig map is never written (+ 8 B per emitted key)
LK_BYTES_PER_POINT = 6272.0          # 2 directions x 2 levels x (28x28 I-patch + 28x28 J-patch), u8
ZNCC_BYTES_PER_POINT = 7396.0        # 2 x 43x43 x 2 B
SELECT_BYTES_PER_CANDIDATE = 16.0    # candidate ranking
MI_BYTES_PER_POINT = 12996.0         # 2 x 57x57 x 2 B chips per scored key point
PHASE_BYTES_PER_PX_F64 = 116.0       # SURVEY 8(d) large-shift model executed in fp64 (reference precision)
PHASE_BYTES_PER_PX_F32 = 60.0        # SURVEY 8(d) large-shift model in float32 (28 forward + 12 cross power + 16 inverse + 4 arg-max)
SHIFT_BYTES_PER_PX = 4.0
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured streaming copy)
PMC_FILE = os.path.join("profiles", "pmc_traffic.json")
SENS_FILE = os.path.join("profiles", "r02_oracle_sensitivity.json")


# ---------------------------------------------------------------------------------------------------- roofline helpers
def pmc_traffic(kernel: str, S: int) -> dict:
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json: separate --pmc runs, gfx950
    correction 2 * FETCH_SIZE + WRITE_SIZE).  PRECOMPUTED - measured with rocprofv3 on an earlier run of the same command, not by
    this process - and labelled so."""
    path = os.path.join(ROOT, PMC_FILE)
    try:
        db = json.load(open(path))
    except Exception:
        return {"traffic": None}
    ent = db.get(kernel)
    if not isinstance(ent, dict) or str(S) not in ent:
        return {"traffic": None}
    out = {"traffic": ent[str(S)], "traffic_source": f"precomputed: {PMC_FILE}"
           + (f" (measured at commit {ent['measured_at']})" if ent.get("measured_at") else " (round-1 counters)")}
    busy = (ent.get("_detail") or {}).get("valu_pipe_busy")
    if busy is not None:
        # what actually bounds the stencil / tracker kernels: instruction issue.  4 cycles per VALU wave-instruction over the 1024 SIMDs'
        # cycles of the launch, from the same precomputed passes
        out["valu_pipe_busy"] = busy
        out["valu_pipe_busy_note"] = "4 * SQ_INSTS_VALU / (1024 SIMDs * kernel cycles), kernels serialised by the counter run (precomputed)"
    return out


def roofline_of(stage_ms: dict, S: int, n_init: int, n_cand: int, n_zncc: int, timed_stage: str | None, minmax_early: bool = False) -> dict:
    """Every stage's bytes-it-must-move / span, and the object for the LARGEST one.  Dense stages: SURVEY 8(d)'s per-pixel figures;
    the fused minimum-eigenvalue + candidate kernel is priced on what IT moves (source 1 B/px + mask 1 B/px + 8 B per emitted key) -
    SURVEY's 10 B/px for the two unfused steps counts an eigenvalue-map round trip the fusion removed and is reported next to it as
    `unfused_model`; LK 6272 B per corner, ZNCC 7396 B per scored row, corner ranking + selection 16 B per candidate."""
    px = float(S) * S
    model = {"minmax": STAGE_BYTES_PER_PX["minmax"] * px, "stretch_laplacian_mask": STAGE_BYTES_PER_PX["stretch_laplacian_mask"] * px,
             "pyramid": STAGE_BYTES_PER_PX["pyramid"] * px, "lk_fwd_bwd": LK_BYTES_PER_POINT * n_init, "zncc": ZNCC_BYTES_PER_POINT * n_zncc}
    fused = stage_ms.get("candidates", 0) == 0 and stage_ms.get("min_eigen", 0) > 0
    eig_name = "min_eigen_candidates_fused" if fused else "min_eigen"
    if fused:
        model[eig_name] = FUSED_EIG_BYTES_PER_PX * px + 8.0 * n_cand
    else:
        model["min_eigen"] = STAGE_BYTES_PER_PX["min_eigen"] * px
        model["candidates"] = STAGE_BYTES_PER_PX["candidates"] * px
    ms = dict(stage_ms)
    if fused:
        ms[eig_name] = ms.pop("min_eigen")
        ms.pop("candidates", None)
    if "sort" in ms or "select" in ms:
        ms["rank_select"] = ms.pop("sort", 0.0) + ms.pop("select", 0.0)
        model["rank_select"] = SELECT_BYTES_PER_CANDIDATE * n_cand
    table = {}
    for k, b in model.items():
        t = ms.get(k, 0.0)
        if t > 0 and b > 0:
            table[k] = {"ms": round(t, 4), "bytes": b, "achieved": b / (t * 1e-3) / 1e9, "frac": b / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # the pyramids run on the library's second stream beside the fused eigenvalue pass (they fill what that issue-bound kernel leaves):
    # their span is stretched by the sharing and is not on the critical path - never the "largest kernel"
    if "pyramid" in table:
        table["pyramid"]["overlapped"] = "second stream, beside min_eigen: the span is stretched by the sharing (0.10 ms alone)"
    # likewise the min / max of a unit submitted behind another one (KM_PATH_MM_EARLY): second stream, beside the PREVIOUS unit's LK /
    # FB test / ZNCC - HBM-bound work under instruction-bound kernels; its span covers that whole window
    hidden = {"pyramid"}
    if minmax_early and "minmax" in table:
        table["minmax"]["overlapped"] = "second stream, beside the previous unit's LK .. ZNCC (0.083 ms alone at 5.8 TB/s); LK pays ~0.03 ms for the sharing"
        hidden.add("minmax")
    dom = max((k for k in table if k not in hidden), key=lambda k: table[k]["ms"])
    d = table[dom]
    out = {"bound": "hbm", "kernel": dom, "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
           **pmc_traffic(dom, S), "algorithmic_bytes_per_launch": d["bytes"], "kernel_ms": d["ms"],
           "kernel_ms_source": ("HIP events over the timed steps" if dom == timed_stage else "HIP events over an untimed pass of the same loop")}
    t = out.get("traffic")
    if t:
        out["frac_of_measured_traffic"] = t / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS      # what the kernel really moved (PMC, precomputed) / time / peak
    if fused and eig_name in table:
        unf = (STAGE_BYTES_PER_PX["min_eigen"] + STAGE_BYTES_PER_PX["candidates"]) * px
        e = table[eig_name]
        e["unfused_model"] = {"bytes": unf, "achieved": unf / (e["ms"] * 1e-3) / 1e9, "frac": unf / (e["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "SURVEY 8(d) P3 + P4 = 10 B/px: what the two unfused steps would move (eig map written and read back)"}
    out["kernels"] = table
    dense = [k for k in ("minmax", "stretch_laplacian_mask", eig_name, "candidates") if k in table]
    db, dm = sum(table[k]["bytes"] for k in dense), sum(table[k]["ms"] for k in dense if k not in hidden)
    if "pyramid" in table:
        db += table["pyramid"]["bytes"]            # (their bytes count, their time hides under the eigenvalue pass)
    out["dense_path"] = {"bytes": db, "ms": dm, "achieved": db / (dm * 1e-3) / 1e9, "frac": db / (dm * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "minmax + stretch/Laplacian/mask + fused eigenvalue pass (+ the pyramids' bytes, hidden beside it"
                                 + ("; the min / max bytes likewise: hidden beside the previous unit's LK)" if "minmax" in hidden else ")")}
    tb = sum(v["bytes"] for v in table.values())
    tm = sum(v["ms"] for k, v in table.items() if k not in hidden)
    out["all_stages"] = {"bytes": tb, "ms_serial_sum": tm, "achieved": tb / (tm * 1e-3) / 1e9, "frac": tb / (tm * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return out


def roofline_units(c4: dict) -> dict | None:
    """Roofline object of config 4's dominant kernel - the fused minimum-eigenvalue + candidate pass over a rank's units in ONE launch
    (`eig3_units_kernel`) - from the HIP-event spans `benchkit.config4` sampled inside its timed region (rank 0's launches)."""
    if not c4.get("stage_launch_ms"):
        return None
    px, cand, ms = c4["px_per_launch_rank0"], c4.get("candidates_per_launch_rank0", 0.0), c4["stage_launch_ms"]
    b = FUSED_EIG_BYTES_PER_PX * px + 8.0 * cand
    unf = (STAGE_BYTES_PER_PX["min_eigen"] + STAGE_BYTES_PER_PX["candidates"]) * px
    return {"bound": "hbm", "kernel": "min_eigen_candidates_fused", "achieved": b / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "frac_8d_model": unf / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "launch_ms": ms, "units_per_launch": c4["units_per_launch"], "algorithmic_bytes_per_launch": b,
            "kernel_ms_source": f"HIP events over {c4['stage_launches_timed']} launches of the timed steps (rank 0)"}
