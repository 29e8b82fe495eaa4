"""bench.py's rules for turning simulations/s into games/s, on the committed probe records: a probe counts only at
the workload's OWN simulation budget, in f32 arithmetic, under the random-stream specification the library implements; without one `value` is null."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_probe_of_the_current_specification_wins():
    ignored = []
    d = bench.committed_stones_probe("config2", 400, "C6RNG v5", ignored)
    assert d is not None and d["file"] == "r04_stones_probe_s400.json" and d["num_simulation"] == 400 and d["precision"] == "fp32" and d["rng"] == "C6RNG v5"
    assert d["games"] >= 256 and 120 < d["stones_per_game"] < 160
    why = " ".join(ignored)
    assert "r03_stones_probe_s400.json: random-stream specification C6RNG v4" in why      # another specification: void
    assert "r02_stones_probe.json: played at 8 simulations per move" in why               # another budget: never a divisor
    assert "r02_stones_probe_s400.json: precision bf16x3" in why


def test_records_from_before_round_4_are_c6rng_v4():
    d = bench.committed_stones_probe("config2", 400, "C6RNG v4")
    assert d is not None and d["file"] == "r03_stones_probe_s400.json" and "rng" not in json.load(open(os.path.join(ROOT, "profiles", d["file"])))


def test_config3_has_a_probe_at_its_own_budget():
    """r5: 200 complete games of the 8-block net at 800 simulations per move -- a small probe, so its standard error is part of the record"""
    d = bench.committed_stones_probe("config3", 800, "C6RNG v5")
    assert d is not None and d["file"] == "r05_stones_probe_config3_s800.json" and d["net_blocks"] == 8 and d["num_simulation"] == 800 and d["precision"] == "fp32"
    assert d["games"] >= 64 and 130 < d["stones_per_game"] < 190 and 0.0 < d["standard_error"] < 8.0


def test_no_probe_at_the_workloads_own_budget_means_no_value():
    assert bench.committed_stones_probe("config3", 400, "C6RNG v5") is None      # bench.py then prints value: null and sims_per_s
    assert bench.committed_stones_probe("config3", 800, "C6RNG v4") is None
    assert bench.committed_stones_probe("config2", 800, "C6RNG v5") is None
    assert bench.committed_stones_probe("config2", 400, "C6RNG v6") is None


def test_pooled_figure_is_reported_beside_value_only():
    p = bench.pooled_stones_probes("config2", 400)
    assert p["games"] >= 864 + 256 and len(p["probes"]) == 2 and p["standard_error"] < 2.0
    d = bench.committed_stones_probe("config2", 400, "C6RNG v5")
    lo, hi = sorted((d["stones_per_game"], 135.88))
    assert lo <= p["stones_per_game"] <= hi


def test_pmc_summary_is_keyed_to_the_convolution_sources():
    s = json.load(open(sorted(p for p in (os.path.join(ROOT, "profiles", f) for f in os.listdir(os.path.join(ROOT, "profiles"))) if p.endswith("_pmc_summary.json"))[-1]))
    assert s["src_hash_of"] == list(bench.CONV_SOURCES)
    # (equal while conv_wino6.hip / c6_wino.h / net_kernels.hip are the profiled ones: then bench.py replays the counters as `traffic`)
    if s["src_hash"] == bench.kernel_source_hash(bench.CONV_SOURCES):
        rec = bench.pmc_record("c6_w6_gemm_kernel<false, false, false>")
        assert rec["hbm_bytes_per_launch"] > 3e10 and 0.8 < rec["mfma_util"] < 1.0
    else:
        assert bench.pmc_record("c6_w6_gemm_kernel<false, false, false>") == {}
