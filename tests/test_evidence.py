"""The stamp that ties committed profiles to the sources they were collected on (ddp-generator_amd/evidence.py): bench.py
quotes a figure from profiles/traffic*.json / issue*.json only while the digest matches."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _evidence():
    import __graft_entry__ as g
    return g.load_package().evidence


def test_digest_covers_kernel_sources_and_benchmark_pairs():
    ev = _evidence()
    files = [os.path.relpath(f, ROOT) for f in ev.source_files()]
    assert files == sorted(files)
    for needed in ("ddp-generator_amd/csrc/ilqg_kernels.hip", "ddp-generator_amd/csrc/k_wave_backward.inc", "ddp-generator_amd/csrc/ilqg_quad.hpp",
                   "ddp-generator_amd/csrc/ilqg_shim_impl.inc", "ddp-generator_amd/csrc/Makefile", "problems/carparking/iLQG_func.c",
                   "problems/synth16x8/iLQG_func.c", "problems/synth16x8/iLQG_problem.h"):
        assert needed in files, needed
    h = hashlib.sha256()
    for f in files:
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(ROOT, f), "rb").read())
    assert ev.source_sha() == h.hexdigest()[:16]


def test_stamped_profile_is_used_only_on_the_sources_it_was_collected_on(tmp_path):
    ev = _evidence()
    good, bad, bare = tmp_path / "good.json", tmp_path / "bad.json", tmp_path / "bare.json"
    good.write_text(json.dumps({"_source_sha": ev.source_sha(), "k": {"hbm_bytes_per_launch": 1.0}}))
    bad.write_text(json.dumps({"_source_sha": "0123456789abcdef", "k": {"hbm_bytes_per_launch": 1.0}}))
    bare.write_text(json.dumps({"k": {"hbm_bytes_per_launch": 1.0}}))
    j, why = ev.load_stamped(str(good))
    assert why is None and j["k"]["hbm_bytes_per_launch"] == 1.0
    for p in (bad, bare, tmp_path / "absent.json"):
        j, why = ev.load_stamped(str(p))
        assert j is None and why and ("collected on sources" in why or "not present" in why)


def test_bench_reports_null_and_the_reason_for_a_stale_profile(tmp_path, monkeypatch):
    import bench
    ev = _evidence()
    monkeypatch.setattr(bench, "stamped", lambda name: ev.load_stamped(str(tmp_path / name)))
    (tmp_path / "issue.json").write_text(json.dumps({"_source_sha": "0123456789abcdef", "_steps": 500,
                                                      "k_backward<2>": {"valu_insts_per_wave_and_step": 1600.0, "active_valu_frac": 0.4}}))
    o = bench.issue_object("k_backward<2>")
    assert o["valu_insts_per_step"] is None and "collected on sources" in o["stale"]
    (tmp_path / "issue.json").write_text(json.dumps({"_source_sha": ev.source_sha(), "_steps": 500,
                                                      "k_backward<2>": {"valu_insts_per_wave_and_step": 1600.0, "active_valu_frac": 0.4}}))
    o = bench.issue_object("k_backward<2>")
    assert o["valu_insts_per_step"] == 1600.0 and o["active_valu_frac"] == 0.4 and "stale" not in o
    (tmp_path / "issue_config5.json").write_text(json.dumps({"_source_sha": ev.source_sha(), "_steps": 1000, "k_backward_quad<true>": {
        "valu_insts_per_wave_and_step": 9.9e4, "valu_insts_per_wavefront_step": 5200.0, "trajectories_per_wavefront": 4,
        "valu_insts_per_trajectory_step": 1300.0, "active_valu_frac": 0.6}}))
    o = bench.issue_object("k_backward_quad<true>", "issue_config5.json")
    assert o["valu_insts_per_step"] == 1300.0 and o["per"].startswith("step of a trajectory")
