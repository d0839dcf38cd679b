"""The ONE line the driver parses: small on purpose (< 6 KB).  Everything else the bench measures is `detail` - written to
`bench_detail.json` and printed on earlier stdout lines (`detail <name> <json>`), never on the last one.

`small_line(detail)` is a pure function of the detail dict so that the contract (keys, size) is testable without a GPU
(tests/test_host_logic.py::test_bench_small_line_*)."""
from __future__ import annotations

import json

MAX_BYTES = 6000
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "frac_8d_model", "frac_of_measured_traffic",
                 "valu_pipe_busy", "launch_ms", "pairs_per_launch", "units_per_launch", "algorithmic_bytes_per_launch", "kernel_ms", "kernel_ms_source",
                 "kernel_ms_alone", "frac_alone")
GATED = ("hard_content", "tie_heavy", "e2e_shape", "config3", "config5", "auto_ksize")


def _short(s, n=118):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _r(v, nd=4):
    return None if v is None else round(float(v), nd)


def small_roofline(roof: dict | None) -> dict | None:
    if not roof:
        return None
    out = {k: roof[k] for k in ROOFLINE_KEYS if k in roof and roof[k] is not None}
    out.setdefault("traffic", roof.get("traffic"))
    k = (roof.get("kernels") or {}).get(roof.get("kernel"), {})
    unf = k.get("unfused_model")
    if unf and "frac_8d_model" not in out:
        out["frac_8d_model"] = unf["frac"]          # SURVEY 8(d) P3 + P4 (10 B/px): what the two unfused steps would move
    for key in ("achieved", "frac", "frac_8d_model", "frac_of_measured_traffic", "launch_ms", "kernel_ms", "kernel_ms_alone", "frac_alone"):
        if key in out:
            out[key] = _r(out[key], 5)
    ws = roof.get("whole_step")
    if ws:
        out["whole_step_frac"] = _r(ws["frac"], 4)          # every stage's algorithmic bytes / the step's wall time / peak
        out["whole_step_GBps"] = _r(ws["achieved"], 1)
        if "frac_8d_model" in ws:
            out["whole_step_frac_8d_model"] = _r(ws["frac_8d_model"], 4)
    if "traffic_source" in out:
        out["traffic_source"] = _short(out["traffic_source"], 90)
    return out


def small_cpu_baseline(cb: dict | None) -> dict | None:
    if not cb:
        return None
    out = {"value": _r(cb["value"], 3), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": _short(cb.get("sample", ""))}
    par = cb.get("parity") or {}
    out["parity"] = {k: par[k] for k in ("passed", "pairs_gated", "keypoints_identical_and_in_order", "max_abs_ddx_px", "max_abs_ddy_px",
                                         "max_abs_dscore", "max_abs_dzncc") if k in par}
    if "single_thread" in cb:
        out["single_thread_value"] = _r(cb["single_thread"]["value"], 3)
    if "opencv_live" in cb:
        out["opencv_live_Mpx_s"] = _r(cb["opencv_live"].get("Mpx_per_s"), 3)
    return out


def gates(detail: dict) -> dict:
    """name -> passed, for every in-run oracle gate the detail carries."""
    out = {}
    cb = detail.get("cpu_baseline") or {}
    if "parity" in cb and cb["parity"].get("checked", True):
        out["headline_pairs_vs_oracle"] = bool(cb["parity"].get("passed"))
    fs = detail.get("full_scoring") or {}
    if (fs.get("parity") or {}).get("checked"):
        out["full_scoring"] = bool(fs["parity"].get("passed"))
    for name in GATED:
        g = (detail.get(name) or {}).get("gate")
        if g is not None:
            out[name] = bool(g.get("passed"))
    return out


def small_line(detail: dict) -> dict:
    """The driver's line from the full detail: the contract keys, the roofline and cpu_baseline objects, one-number summaries."""
    line = {k: detail.get(k) for k in CONTRACT_KEYS if k not in ("roofline", "cpu_baseline", "config")}
    cfg = dict(detail.get("config") or {})
    cfg["workload"] = _short(cfg.get("workload", ""))
    if "parallelism" in cfg:
        cfg["parallelism"] = _short(cfg["parallelism"], 90)
    line["config"] = cfg
    line["dtype"] = _short(line.get("dtype"), 60)
    line["roofline"] = small_roofline(detail.get("roofline"))
    line["cpu_baseline"] = small_cpu_baseline(detail.get("cpu_baseline"))
    for k in ("matched_keypoints_per_sec", "matched_keypoints_per_pair", "speedup_vs_cpu_port"):
        if detail.get(k) is not None:
            line[k] = _r(detail[k], 1)
    for k in ("world", "launcher", "backend", "rccl_ranks_seen", "units_per_rank", "speculative_tiles_redone"):
        if detail.get(k) is not None:
            line[k] = detail[k]
    if detail.get("stage_ms"):
        line["stage_ms"] = {k: _r(v) for k, v in detail["stage_ms"].items() if v}
    g = gates(detail)
    line["gates_passed"] = g
    line["gates_all_passed"] = all(g.values()) if g else None        # (None: no gate ran - N > 1 lines, --headline-only)

    def ms(name, key="ms_per_pair"):
        v = (detail.get(name) or {}).get(key)
        return None if v is None else _r(v)

    c3 = detail.get("config3") or {}
    summary = {
        "one_pair_per_submission_ms": ms("one_pair_per_submission"),
        "same_pair_repeated_ms": ms("same_pair_repeated"),
        "full_scoring_ms": ms("full_scoring"),
        "full_scoring_batched_ms": _r(((detail.get("full_scoring") or {}).get("batched") or {}).get("ms_per_pair")),
        "in_flight_ms": ms("in_flight"),
        "end_to_end_ms": ms("end_to_end"),
        "end_to_end_note": "PCIe-inclusive (482 MB per pair from page-locked host rasters through KLT.match): what an unmodified KARIOS sees; never `value`" if detail.get("end_to_end") else None,
        "config3_fp64_ms": (c3.get("phase_fp64") or {}).get("config3_ms_per_step_at_reference_precision"),
        "config3_phase_fp64_ms": (c3.get("phase_fp64") or {}).get("ms"),
        "config3_f32_ms": _r(c3.get("ms_per_step")) if c3 else None,
        "config4_ms": ms("config4", "ms_per_step"),
        "config4_Mpx_s": _r((detail.get("config4") or {}).get("value"), 1) if detail.get("config4") else None,
        "config5_ms": ms("config5", "ms_per_step"),
        "config5_tiled_mask_ms": ((detail.get("config5") or {}).get("tiled_with_mask") or {}).get("ms_per_pair"),
        "auto_ksize_ms": ms("auto_ksize"),
        "hard_content_ms": ms("hard_content"), "tie_heavy_ms": ms("tie_heavy"), "e2e_shape_ms": ms("e2e_shape"),
        "weak_pairs_Mpx_s": _r((detail.get("weak_pairs") or {}).get("value"), 1) if detail.get("weak_pairs") else None,
        "weak_pairs_ms_per_step": ms("weak_pairs", "ms_per_step"),
    }
    line["summary"] = {k: v for k, v in summary.items() if v is not None}
    line["detail"] = detail.get("detail_where", "bench_detail.json; `detail <name> {...}` lines above this one")
    # guard: the line must stay small whatever a leg adds - optional parts go first
    for drop in ("stage_ms", "launcher", "backend", "speedup_vs_cpu_port"):
        if len(json.dumps(line)) <= MAX_BYTES:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) > MAX_BYTES:
        line["summary"] = {k: v for k, v in line["summary"].items() if not k.endswith("_note")}
    return line
