"""The stdout line of bench.py is what the driver parses: it must stay a compact object (round 5's 20.7 KB line came back
`parsed: null`).  Built here from canned full results -- the committed full line of round 5 (profiles/r05_bench.json) and
synthetic worst cases -- without a GPU."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.fixture(scope="module")
def canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))


def test_compact_line_from_the_round5_result(canned):
    assert len(json.dumps(canned)) > 20000                    # the object the driver could not parse
    text = bench.compact_line(canned)
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 6144
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == canned["value"] and line["ms_per_step"] == canned["ms_per_step"]      # not rounded
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert line["config"]["workload"].startswith("batch 65536 filters/GPU") and "model" not in line["config"]
    roof = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "launches", "achieved_api", "hbm_resident"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-4)
    assert roof["hbm_resident"]["batch"] == 1048576 and 0 < roof["hbm_resident"]["frac"] < 1
    cpu = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] == "port" and cpu["cores"] == 16
    assert set(line["north_star"]) == {"pixels_m4", "pixels_m4_stereo", "fused_frame_pixels_m4"}
    assert line["fp64"]["value"] == pytest.approx(canned["fp64"]["value"], rel=1e-4)
    # the consistency the driver checks: ms_per_step x steps is the timed region, value = filters x 230 x steps / it
    assert line["value"] == pytest.approx(65536 * 230 * line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3), rel=1e-9)


def test_compact_line_of_a_multi_gpu_result(canned):
    """an N > 1 line has no side legs: nulls / absent blocks, still every contract key"""
    out = copy.deepcopy(canned)
    out.update({"n_gpus": 8, "cpu_baseline": None, "fp64": None, "north_star_rows": None, "roofline_hbm_resident": None,
                "legs_skipped": [f"{leg}: single-GPU side leg, not run with 8 ranks" for leg in ("fp64", "north_star_rows", "cpu_baseline")]})
    line = json.loads(bench.compact_line(out))
    for k in CONTRACT:
        assert k in line, k
    assert line["cpu_baseline"] is None and "hbm_resident" not in line["roofline"] and "north_star" not in line
    assert line["legs_skipped"] == ["fp64", "north_star_rows", "cpu_baseline"]


def test_compact_line_never_outgrows_the_limit(canned):
    """free text blown up 100x: optional blocks go first, then text is cut; roofline / cpu_baseline / config survive"""
    out = copy.deepcopy(canned)
    out["config"]["workload"] *= 30
    out["cpu_baseline"]["path"] *= 30
    out["cpu_baseline"]["sample"] *= 30
    out["legs_skipped"] = ["x: " + "y" * 500] * 40
    text = bench.compact_line(out)
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["config"]["workload"]


def test_non_finite_numbers_do_not_break_the_line(canned):
    out = copy.deepcopy(canned)
    out["correct_kernel"]["avg_launch_us"] = float("nan")
    line = json.loads(bench.compact_line(out))                # strict JSON: no NaN token
    assert line["correct_kernel"]["avg_launch_us"] is None
    assert "NaN" not in bench.compact_line(out)
