"""A fixed slice of the randomised parity sweep (`tools/fuzz_parity.py`) inside the GPU suite: random sizes, pixel types,
no-data patterns, user masks, kernel sizes, polarity and tracker parameters; corners, tracks and frame columns
bit-identical with the oracle, ZNCC within 1e-9."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


def test_case_generator_is_deterministic():
    a, b = fuzz.draw_case(5), fuzz.draw_case(5)
    assert a == b and fuzz.draw_case(6) != a
    m1, r1, k1 = fuzz.make_inputs(a)
    m2, r2, k2 = fuzz.make_inputs(b)
    assert m1.tobytes() == m2.tobytes() and r1.tobytes() == r2.tobytes() and (k1 is None) == (k2 is None)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 121))
def test_random_tile_case_matches_oracle(ops, O, seed):
    from karios_amd.resident import ResidentPair
    case = fuzz.draw_case(seed, max_size=420)
    fails = fuzz.run_case(case, ops, O, ResidentPair)
    assert not fails, f"{fails} for {case}"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 25))
def test_random_aux_case_matches_oracle(ops, O, seed):
    """Phase correlation, shift_image, ZNCC / MI on random (also out-of-range) key points, outlier-filter tile."""
    from karios_amd.resident import ResidentPair
    fails = fuzz.run_aux_case(seed, ops, O, ResidentPair)
    assert not fails, fails


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 17))
def test_random_batched_submission_matches_oracle_unit_by_unit(ops, O, seed):
    """2 .. 16 random units (boxes of one or two resident pairs, widths off the dword grid) through km_klt_units_frame_submit, every
    unit's frame against the oracle on its box (flagged units through the exact repeat, as FrameStream does); a submission the batch
    form declines for a documented reason counts as covered by the unit-by-unit tests."""
    from karios_amd.resident import ResidentPair
    fails = fuzz.run_units_case(seed, ops, O, ResidentPair)
    assert not fails, fails


@pytest.mark.gpu
def test_random_pipelined_streams_equal_the_units_alone():
    """A fixed slice of tools/fuzz_pipeline.py: random streams of batched submissions (masked or not, every third one artificially
    flagged) through the software pipeline of csrc/api_units.hip against every unit submitted alone through the exact path."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_pipeline.py"), "--rounds", "24", "--seed", "7001"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 FAILED" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
