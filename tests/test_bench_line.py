"""bench.py's output contract: ONE line on stdout, < 6 KB whatever the report holds, the contract's keys never
dropped; the full report goes to bench_detail.json.  CPU only: compact_line() on synthetic reports, and the stdout guard."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline", "detail")


def report(blow_up=1):
    big = "x" * (400 * blow_up)
    cfg5 = {"value": 7.3, "ms_per_step": 137.0, "steps": 3, "roofline": {"frac": 0.34, "hbm_equivalent_frac": 1.36, "traffic": 3.7e11, "issue": {"active_valu_frac": 0.5},
                                                                           "note": big}, "cpu_baseline": {"value": 0.016}, "config": {"mapping": big}}
    return {"metric": "iLQG iterations/sec, 65k-batch CarParking (n=4,m=2,N=500)", "value": 188.0, "unit": "iterations/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 5.3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_definition": big, "config": {"workload": "CarParking batch=65536 per GPU x 1 GPU", "mapping": "one lane per trajectory", "batch_per_gpu": 65536, "n_hor": 500,
                                                 "full_ddp": 0, "stream_groups": 4, "parallelism": "dp1", "env_switches": {"ILQG_X": big}},
            "roofline": {"bound": "hbm", "kernel": "k_backward[fused derivs]", "achieved": 2980.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.37, "traffic": 1.9e9,
                         "hbm_utilisation_frac": 0.087, "limiter": "valu_issue", "note": big * 3,
                         "dominant_launch": {"avg_launch_ms": 2.8, "launches": 80, "trajectories_per_launch": 16384.0, "hbm_equivalent": {"algorithmic_bytes_per_launch": 8.4e9},
                                             "valu_fp64": {"frac_of_peak": 0.03}, "pmc": {"source": "collected in this run: ..." + big}},
                         "issue": {"valu_insts_per_step": 1359.2, "active_valu_frac": 0.67, "source": big}},
            "iteration_roofline": {"achieved_GBs": 7390.0, "frac_of_peak": 0.92, "algorithmic_bytes_per_iteration": 39321600000, "note": big},
            "cpu_baseline": {"value": 0.62, "unit": "iterations/s (65536-trajectory batch equivalent)", "cores": 256, "kind": "reference",
                             "single_core_ms_per_trajectory_iteration": 0.19, "sample": "32768 trajectories x 20 iterations in 16.3 s on 256 threads" + big * (blow_up > 1)},
            "config5": cfg5, "config5_stored": dict(cfg5, value=1.8), "config2": {"lane_mapping": {"value": 327.0}, "wave_mapping": {"value": 172.0}, "note": big},
            "dropin_b1": {"ms_per_iteration": 2.5, "iterations": 20, "note": big}, "full_solve": {"value": 35000.0, "unit": "solves/s", "plain": {"occupancy_over_time": [big] * 16}},
            "kernels_ms_per_iteration_overlapping": {"k": 1.0}}


def test_the_line_holds_the_contract_and_stays_under_the_cap():
    line = bench.compact_line(report(), "bench_detail.json")
    assert len(line) < 3500
    j = json.loads(line)
    for k in REQUIRED:
        assert k in j, k
    assert j["roofline"]["frac"] == 0.37 and j["roofline"]["launches"] == 80 and j["roofline"]["valu_insts_per_step"] == 1359.2 and j["roofline"]["traffic_source"] == "live"
    assert j["config5"] == {"value": 7.3, "ms_per_step": 137.0, "steps": 3, "frac": 0.34, "hbm_equivalent_frac": 1.36, "traffic": 3.7e11, "active_valu_frac": 0.5,
                            "cpu_baseline_value": 0.016}
    assert j["config2"] == {"lane_mapping": 327.0, "wave_mapping": 172.0} and j["dropin_b1"] == {"ms_per_iteration": 2.5, "iterations": 20}
    assert "note" not in j["roofline"] and "env_switches" not in j["config"]  # (the long texts live in the report)


def test_optional_objects_go_before_the_cap_is_broken():
    """a report whose kept strings have grown tenfold: optional objects are dropped, the contract's keys stay, the cap holds"""
    line = bench.compact_line(report(blow_up=12), "bench_detail.json")
    assert len(line) < bench.LINE_CAP
    j = json.loads(line)
    for k in REQUIRED:
        assert k in j, k
    assert "full_solve" not in j


def test_a_failed_secondary_object_is_reported_in_place():
    r = report()
    r["config5"] = {"error": "child `--object config5` rc 1: " + "e" * 1000}
    j = json.loads(bench.compact_line(r, "bench_detail.json"))
    assert set(j["config5"]) == {"error"} and len(j["config5"]["error"]) <= 160 and j["value"] == 188.0


def test_nothing_but_the_line_reaches_stdout(tmp_path):
    """the guard: what C libraries and children print to fd 1 during the run goes to the log, the line goes to stdout"""
    code = ("import os, sys\nsys.path.insert(0, %r)\nimport bench\nbench.ROOT = %r\ng = bench.StdoutGuard()\n"
            "os.system('echo noise from a child')\nos.write(1, b'noise from C\\n')\nprint('noise from python')\ng.emit('{\"the\": \"line\"}')\n" % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout == '{"the": "line"}\n', (r.stdout, r.stderr)
    log = (tmp_path / "bench_detail.log").read_text()
    assert "noise from a child" in log and "noise from C" in log and "noise from python" in log
