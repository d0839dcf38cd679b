"""The bench's multi-rank path on ONE GPU: `python3 bench.py --gpus 2 --share-gpu` spawns two ranks itself (gloo collectives,
both ranks on the visible GPU) - the launch, the world-size check, the per-rank weak-scaling headline and config 4's fixed
16-unit workload gathered over the two ranks (reference tiles: karios/matcher/klt.py:220-253)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, timeout=900):
    """-> (the LAST stdout line = the driver's small line, the detail objects of the `detail <name> {...}` lines above it)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1 and out[-1] == lines[0], r.stdout[-2000:]
    assert len(lines[0]) < 8000                                   # the driver's record holds the whole line
    detail = {}
    for ln in out:
        if ln.startswith("detail "):
            _, name, body = ln.split(" ", 2)
            detail[name] = json.loads(body)
    return json.loads(lines[0]), detail


def test_bench_gpus_2_value_is_config4_strong_scaling_over_two_ranks():
    """with N > 1 `value` is BASELINE config 4 - the fixed 16 units (karios/matcher/klt.py:220-253) dealt to the ranks,
    batched per rank, one all-gather per step - and the band-per-rank stream is the side number `weak_pairs`."""
    line, detail = _bench(["--gpus", "2", "--share-gpu", "--steps", "4", "--warmup", "1"])
    assert line["n_gpus"] == 2 and line["world"] == 2 and line["rccl_ranks_seen"] == 2
    assert line["launcher"].startswith("bench.py") and line["backend"] == "gloo"
    assert line["scaling"] == "strong" and line["steps"] == 4 and line["warmup"] == 1
    assert line["config"]["workload"].startswith("BASELINE config 4") and line["config"]["units"] == 16
    assert line["units_per_rank"] == [8, 8] and line["config"]["units_per_rank"] == [8, 8]
    c4 = detail["config4"]
    assert c4["n_gpus"] == 2 and c4["units"] == 16 and c4["units_gathered"] == 16 and c4["steps"] == 4
    assert c4["matched_keypoints_per_step"] > 16 * 10000 and c4["scaling"] == "strong"
    assert abs(line["value"] - 4 * 10980 * 10980 / 1e6 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["kernel"] == "min_eigen_candidates_fused" and 0 < roof["frac"] < 1 and roof["units_per_launch"] == 8
    wk = detail["weak_pairs"]                                                       # the stream of independent pairs, one per rank and step
    assert wk["scaling"] == "weak" and wk["config"]["pairs_per_step"] == 2 and wk["matched_keypoints_per_pair"] > 10000
    assert line["summary"]["weak_pairs_Mpx_s"] > 0
    assert sorted(d["rank"] for d in detail["devices"]) == [0, 1]
    assert line["cpu_baseline"] is None and "in_flight" not in detail and "config3" not in detail     # rank-0, N = 1 legs only


def test_bench_gpus_8_shares_the_gpu_between_eight_ranks_two_units_each():
    """The launch an 8-GPU node would see - eight ranks, ports, the rendezvous, config 4's sixteen units dealt two per rank and gathered,
    no rank starved - with the ranks time-slicing the one GPU a builder has (gloo collectives: RCCL wants a device per rank).  What it
    cannot show is xGMI; what it does show is that eight ranks' worth of host work fits the box."""
    line, detail = _bench(["--gpus", "8", "--share-gpu", "--steps", "4", "--warmup", "1"], timeout=1500)
    assert line["n_gpus"] == 8 and line["world"] == 8 and line["rccl_ranks_seen"] == 8 and line["backend"] == "gloo"
    assert sorted(d["rank"] for d in detail["devices"]) == list(range(8))
    assert line["scaling"] == "strong" and line["units_per_rank"] == [2] * 8
    c4 = detail["config4"]
    assert c4["n_gpus"] == 8 and c4["units"] == 16 and c4["units_gathered"] == 16 and c4["units_per_rank"] == [2] * 8
    assert c4["matched_keypoints_per_step"] > 16 * 10000
    wk = detail["weak_pairs"]
    assert wk["config"]["pairs_per_step"] == 8
    cpu = wk["host_cpu_ms_per_step"]
    assert len(cpu["process_ms_per_step_per_rank"]) == 8 and all(v > 0 for v in cpu["process_ms_per_step_per_rank"])       # every rank worked


def test_bench_default_line_is_small_and_the_detail_carries_every_leg():
    line, detail = _bench(["--steps", "8", "--warmup", "2", "--cpu-runs", "2"])
    assert line["n_gpus"] == 1 and line["launcher"] == "single process" and line["scaling"] == "weak"
    assert line["config"]["pairs_per_submission"] == 4 and line["config"]["distinct_pairs_resident"] == 4 and len(set(line["config"]["seeds"])) == 4
    for key in ("roofline", "cpu_baseline", "end_to_end", "in_flight", "config3", "config4", "config5", "full_scoring", "one_pair_per_submission",
                "same_pair_repeated", "stage_ms", "step_spread"):
        assert key in detail or key in line, key
    cb = line["cpu_baseline"]
    assert cb["parity"]["passed"] is True and cb["parity"]["pairs_gated"] == 2 and cb["kind"] == "port"
    assert detail["cpu_baseline"]["parity"]["by_pair"]["1"]["keypoints_identical_and_in_order"] is True        # a DISTINCT pair of the submission
    fs = detail["full_scoring"]       # the whole scoring of _handle_klt_results in the tile call, gated against the oracle on every row
    assert fs["parity"]["passed"] is True and fs["parity"]["rows"] == fs["matched_keypoints_per_pair"] > 10000
    assert {"zncc_score", "mutual_info_score", "mi_score"} <= set(fs["columns"]) and fs["roofline"]["frac"] > 0
    assert set(detail["config4"]["contexts_in_flight_ab"]) >= {"1", "3", "batched"}
    for key in ("hard_content", "tie_heavy", "e2e_shape"):          # the step on content that is not the best case, gated inside the run
        assert detail[key]["gate"]["passed"] is True and detail[key]["gate"]["keypoints_identical_and_in_order"] is True, key
        assert line["gates_passed"][key] is True
    assert line["gates_all_passed"] is True
    assert 0.4 <= detail["hard_content"]["forward_backward_survival"] <= 0.6
    assert detail["e2e_shape"]["tiles"] == 4 and detail["host_cpu_ms_per_step"]["process_ms_per_step"] > 0
    g = detail["config3"]["gate"]
    assert g["passed"] is True and g["gpu_crop_row_col"] == g["oracle_crop_row_col"]
    assert detail["config3"]["detected_offset_row_col"] == [-21.0, 37.0]
    assert detail["in_flight"]["pairs_in_flight"] == 3 and detail["in_flight"]["ms_per_pair"] > 0
    roof, full = line["roofline"], detail["roofline"]
    assert 0 < roof["frac"] < 1 and full["kernel"] in full["kernels"] and roof["pairs_per_launch"] == 4
    if roof.get("traffic") is not None:
        assert roof["traffic_source"].startswith("precomputed: profiles/")
    assert detail["oracle_sensitivity"]["source"].startswith("precomputed: profiles/")
    assert detail["config5"]["matched_keypoints_per_pair"] > 5000
    s = line["summary"]
    for key in ("one_pair_per_submission_ms", "same_pair_repeated_ms", "end_to_end_ms", "config3_fp64_ms", "config4_ms", "config5_ms"):
        assert s[key] > 0, key
