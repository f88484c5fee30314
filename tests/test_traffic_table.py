"""The whole-step HBM figure of bench.py's line (`roofline.whole_step.counter_GBs` / `counter_frac`) is a profile
constant: the counters' bytes of one step from profiles/r06_traffic_step.json times the run's steps per second.  This
file checks, without a GPU, that the table reproduces from the committed counter rows (profiles/r06_pmc/: the
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over tools/kprof.py 4k 128) and that the committed bench
line's figure follows from the table and the line's own step time."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = os.path.join(ROOT, "profiles", "r06_pmc")
TABLE = os.path.join(ROOT, "profiles", "r06_traffic_step.json")


def test_table_reproduces_from_the_committed_counter_rows(tmp_path):
    out = tmp_path / "again.json"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_step.py"), PMC, PMC, str(out), "4k", "128", "7"],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    again, table = json.load(open(out)), json.load(open(TABLE))
    assert again["bytes_per_step"] == pytest.approx(table["bytes_per_step"], rel=1e-12)
    assert set(again["kernels"]) == set(table["kernels"])
    for k, v in table["kernels"].items():
        assert again["kernels"][k]["bytes_per_step"] == pytest.approx(v["bytes_per_step"], rel=1e-12), k
        assert again["kernels"][k]["launches_per_step"] == v["launches_per_step"], k
    # every kernel of a step is in it: 15 iteration launches (levels 0 - 4 x 3), 128 remap steps, the two fused expansions
    assert table["kernels"]["k_flow_iter_pc"]["launches_per_step"] == 15
    assert table["kernels"]["k_remap_step_px"]["launches_per_step"] == 128
    assert table["kernels"]["k_level0_polyexp_t"]["launches_per_step"] == table["kernels"]["k_level1_polyexp_t"]["launches_per_step"] == 1
    assert 0.8 < table["counter_over_built"] < 1.0          # below the built bytes: consecutive pairs share a frame in L2


def test_the_lines_figure_follows_from_the_table_and_its_own_step_time():
    sys.path.insert(0, ROOT)
    import bench
    from transflow_amd import roofline as rf
    table = json.load(open(TABLE))
    wl = bench.WORKLOADS["4k"]
    got = bench.step_counters(rf, "4k", wl, 128, steps_per_s=10.0)
    assert got["counter_bytes"] == table["bytes_per_step"]
    assert got["counter_GBs"] == pytest.approx(table["bytes_per_step"] * 10.0 / 1e9)
    assert got["counter_frac"] == pytest.approx(table["bytes_per_step"] * 10.0 / 8e12)
    assert "r06_traffic_step.json" in got["counter_source"]
    # another pass size or workload: no table, no figure (nothing is scaled)
    for args in (("4k", wl, 32), ("1080p", bench.WORKLOADS["1080p"], 128)):
        none = bench.step_counters(rf, *args, steps_per_s=10.0)
        assert none["counter_GBs"] is None and none["counter_frac"] is None
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_4k_default.json")).read().strip().splitlines()[-1])
    ws = line["roofline"]["whole_step"]
    assert ws["counter_frac"] == pytest.approx(table["bytes_per_step"] / (line["ms_per_step"] * 1e-3) / 8e12, rel=1e-9)
    assert 0.45 < ws["counter_frac"] < ws["frac"] < 0.65    # counters below built bytes below the kernel's own figure


def test_rocprof_summary_agrees_with_the_line_printed_under_it():
    """profiles/r06_bench_4k_one_lane_kernel_stats.csv is `rocprofv3 --kernel-trace --stats` of the very command whose line
    is profiles/r06_bench_4k_one_lane_under_rocprof.json: the dominant kernel's average launch duration by the profiler
    (every launch of the process) and by HIP events (the timed region's launches) agree, and the line's `achieved` /
    `frac` are its algorithmic bytes per launch over that duration."""
    import csv
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_bench_4k_one_lane_kernel_stats.csv"))))
    it = [r for r in rows if "k_flow_iter_pc" in r["Name"]]
    assert len(it) == 2                                        # <7, 1> and <7, 2> (the first iteration of a level)
    calls = sum(int(r["Calls"]) for r in it)
    avg_ms = sum(float(r["TotalDurationNs"]) for r in it) / calls / 1e6
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_4k_one_lane_under_rocprof.json")).read().strip().splitlines()[-1])
    r = line["roofline"]
    assert r["kernel"] == "fb_flow_iter" and calls % 15 == 0   # 15 launches per pass: levels 0 - 4 x 3 iterations
    assert avg_ms == pytest.approx(r["avg_launch_ms"], rel=0.01)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-9)
    assert r["frac"] == pytest.approx(r["achieved"] / 8000.0, rel=1e-9) and r["peak"] == 8000.0
    share = sum(float(x["Percentage"]) for x in it)
    assert share > 55.0                                        # it IS the dominant kernel of the profiled process


def test_dominant_kernels_traffic_is_its_row_of_the_step_table():
    sys.path.insert(0, ROOT)
    from transflow_amd import roofline as rf
    table = json.load(open(TABLE))
    t = rf.profile_step_traffic("4k", 3840, 2160, 5, 128)
    row = table["kernels"]["k_flow_iter_pc"]
    assert t["kernels_bytes_per_step"]["k_flow_iter_pc"] == row["bytes_per_step"] and t["kernels_launches_per_step"]["k_flow_iter_pc"] == 15
    per_launch = row["bytes_per_step"] / 15
    built = rf.built_kernel_bytes("fb_flow_iter", 3840, 2160, 5, 128) / 15
    assert 0.7 < per_launch / built < 0.9                      # 12.3 GB by the counters against 15.3 GB loaded and stored
