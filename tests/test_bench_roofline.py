"""bench.py's roofline bookkeeping against the committed counter summaries (no GPU): the per-class sums `load_pmc` forms from
profiles/r02_pmc_<workload>.json and the figures `class_roofline` derives from them."""
import importlib.util
import json
import os

import pytest

from conftest import ROOT

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


@pytest.mark.parametrize("workload", ["cornell", "colonnade", "classroom"])
def test_committed_counters_load_and_price(workload):
    pmc, meta = bench.load_pmc(workload)
    assert pmc is not None and meta.get("workload") == workload and meta.get("commit")
    assert {"traversal", "shade"} <= set(pmc)
    raw = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_%s.json" % workload)))
    trav = [k for k in raw if k.startswith("k_extend") or k.startswith("k_connect")]
    assert abs(pmc["traversal"]["launches"] - sum(raw[k]["calls"] for k in trav) / max(int(meta.get("frames", 1)), 1)) < 1e-6
    t = pmc["traversal"]
    ms = sum(raw[k]["total_ms"] for k in trav)
    r = bench.class_roofline("traversal", ms, t["launches"], ms, t["launches"], pmc, algorithmic_bytes=1e9)
    assert r["counters"]["matches_live_launch_count"]
    assert 0.0 < r["hbm_counter_frac"] < 1.0 and 16.0 < r["lanes_per_valu_inst"] <= 64.0
    assert 0.0 < r["valu_lane_frac"] < r["valu_issue_frac"] < 1.0
    assert 0.3 < r["valu_pipe_busy_frac_profiled"] <= 1.0
    # a different launch count means another pipeline: the counters must not be used
    r2 = bench.class_roofline("traversal", ms, t["launches"] + 2, ms, t["launches"] + 2, pmc)
    assert not r2["counters"]["matches_live_launch_count"] and "valu_lane_frac" not in r2
