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
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_and_gathers_sixteen_units():
    line = _bench(["--gpus", "2", "--share-gpu", "--steps", "4", "--warmup", "1"])
    assert line["n_gpus"] == 2 and line["world"] == 2 and line["rccl_ranks_seen"] == 2
    assert line["launcher"].startswith("bench.py") and line["backend"] == "gloo"
    assert sorted(d["rank"] for d in line["devices"]) == [0, 1]
    assert line["config"]["pairs_per_step"] == 2 and line["scaling"] == "weak"
    assert line["matched_keypoints_per_pair"] > 10000
    c4 = line["config4"]
    assert c4["n_gpus"] == 2 and c4["units"] == 16 and c4["units_gathered"] == 16 and c4["units_per_rank"] == [8, 8]
    assert c4["matched_keypoints_per_step"] > 16 * 10000 and c4["scaling"] == "strong"
    assert line["cpu_baseline"] is None and "in_flight" not in line and "config3" not in line     # rank-0, N = 1 legs only


def test_bench_gpus_8_shares_the_gpu_between_eight_ranks_two_units_each():
    """the launch an 8-GPU node would see - eight ranks, ports, the rendezvous, config 4's sixteen units dealt two per
    rank and gathered, no rank starved - with the ranks time-slicing the one GPU a builder has (gloo collectives: RCCL wants a device
    per rank).  What it cannot show is xGMI; what it does show is that eight ranks' worth of host work fits the box."""
    line = _bench(["--gpus", "8", "--share-gpu", "--steps", "4", "--warmup", "1", "--no-sensitivity"], timeout=1500)
    assert line["n_gpus"] == 8 and line["world"] == 8 and line["rccl_ranks_seen"] == 8 and line["backend"] == "gloo"
    assert sorted(d["rank"] for d in line["devices"]) == list(range(8))
    assert line["config"]["pairs_per_step"] == 8
    c4 = line["config4"]
    assert c4["n_gpus"] == 8 and c4["units"] == 16 and c4["units_gathered"] == 16 and c4["units_per_rank"] == [2] * 8
    assert c4["matched_keypoints_per_step"] > 16 * 10000
    cpu = line["host_cpu_ms_per_step"]
    assert len(cpu["process_ms_per_step_per_rank"]) == 8 and all(v > 0 for v in cpu["process_ms_per_step_per_rank"])       # every rank worked


def test_bench_default_line_carries_config3_in_flight_and_labels_precomputed_parts():
    line = _bench(["--steps", "6", "--warmup", "2", "--cpu-runs", "1"])
    assert line["n_gpus"] == 1 and line["launcher"] == "single process"
    for key in ("roofline", "cpu_baseline", "end_to_end", "in_flight", "config3", "config4", "config5", "full_scoring"):
        assert key in line, key
    assert line["cpu_baseline"]["parity"]["passed"] is True
    fs = line["full_scoring"]       # the whole scoring of _handle_klt_results in the tile call, gated against the oracle on every row
    assert fs["parity"]["passed"] is True and fs["parity"]["rows"] == fs["matched_keypoints_per_pair"] > 10000
    assert {"zncc_score", "mutual_info_score", "mi_score"} <= set(fs["columns"]) and fs["roofline"]["frac"] > 0
    assert set(line["config4"]["contexts_in_flight_ab"]) >= {"1", "3", "batched"}
    for key in ("hard_content", "tie_heavy", "e2e_shape"):          # the step on content that is not the best case, gated inside the run
        assert line[key]["gate"]["passed"] is True and line[key]["gate"]["keypoints_identical_and_in_order"] is True, key
    assert 0.4 <= line["hard_content"]["forward_backward_survival"] <= 0.6
    assert line["e2e_shape"]["tiles"] == 4 and line["host_cpu_ms_per_step"]["process_ms_per_step"] > 0
    g = line["config3"]["gate"]
    assert g["passed"] is True and g["gpu_crop_row_col"] == g["oracle_crop_row_col"]
    assert line["config3"]["detected_offset_row_col"] == [-21.0, 37.0]
    assert line["in_flight"]["pairs_in_flight"] == 3 and line["in_flight"]["ms_per_pair"] > 0
    roof = line["roofline"]
    assert 0 < roof["frac"] < 1 and roof["kernel"] in roof["kernels"]
    if roof.get("traffic") is not None:
        assert roof["traffic_source"].startswith("precomputed: profiles/")
    assert line["oracle_sensitivity"]["source"].startswith("precomputed: profiles/")
    assert line["config5"]["matched_keypoints_per_pair"] > 5000
