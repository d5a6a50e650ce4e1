"""bench.py's roofline bookkeeping against the committed counter summaries (no GPU): the per-class sums `load_pmc` forms from
profiles/r04_pmc_<workload>.json, the provenance check (a summary is used only when it carries the hash of the kernel sources
in this tree) and the figures `class_roofline` derives -- every fraction bounded by 1."""
import importlib
import importlib.util
import json
import os

import pytest

from conftest import ROOT

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _source_hash():
    return importlib.import_module("pathtracer-rs_amd.build").source_hash()


@pytest.mark.parametrize("workload", ["cornell", "colonnade", "classroom"])
def test_committed_counters_load_and_price(workload):
    path = os.path.join(ROOT, "profiles", "r04_pmc_%s.json" % workload)
    raw = json.load(open(path))
    h = raw["_meta"]["source_hash"]
    pmc, meta = bench.load_pmc(workload, h)  # priced as on the tree they were measured on
    assert pmc is not None and meta["usable"] and meta.get("workload") == workload and meta.get("source_hash")
    assert {"traversal", "shade"} <= set(pmc)
    trav = [k for k in raw if k.startswith("k_extend") or k.startswith("k_connect")]
    frames = max(int(meta.get("frames", 1)), 1)
    assert abs(pmc["traversal"]["launches"] - sum(raw[k]["calls"] for k in trav) / frames) < 1e-6
    for cls in ("traversal", "shade"):
        c = pmc[cls]
        ms = c["total_ms"]
        r = bench.class_roofline(cls, ms, c["launches"], pmc, algorithmic_bytes=1e9 if cls == "traversal" else None)
        assert r["counters"]["matches_live_launch_count"]
        lo, hi = r["valu_pipe_frac_lo_hi"]
        assert 0.0 < lo <= r["valu_pipe_frac"] <= hi < 1.0            # the vector-ALU time the instructions need, against measured issue rates
        assert 0.0 < r["hbm_counter_frac"] < 1.0 and 16.0 < r["lanes_per_valu_inst"] <= 64.0
        assert abs(r["wave_issue_frac"] + r["wave_wait_frac"] + r["wave_stall_frac"] - 1.0) < 0.05  # a wave issues, waits in s_waitcnt or waits for a slot
        assert 0.0 < r["salu_per_valu"] < 1.0
        assert r["lds_busy_frac"] is None or 0.0 <= r["lds_busy_frac"] < 1.0
        assert r["bound"] in ("hbm", "valu-issue", "lds", "waitcnt", "mixed")
        assert r["valu_ceiling_ginst_per_s"] > r["valu_ginst_per_s"]
    # a different launch count means another pipeline: the counters must not be used
    t = pmc["traversal"]
    r2 = bench.class_roofline("traversal", t["total_ms"], t["launches"] + 2, pmc)
    assert not r2["counters"]["matches_live_launch_count"] and "valu_pipe_frac" not in r2
    # ... and so does a summary measured on other kernel sources
    none, meta2 = bench.load_pmc(workload, "0123456789abcdef")
    assert none is None and meta2["usable"] is False and meta2["source_hash"] == h


def test_committed_counters_belong_to_this_tree(ptrs):
    """The summaries bench.py will use on the GPU box were measured with a library built from exactly the kernel sources and flags of
    this tree (tools/prof.sh records ptrs_build_id(); a later edit of csrc/ or include/ without re-profiling makes bench.py drop the
    counters, and this test fail) -- and the library in the tree is that build."""
    h = _source_hash()
    assert ptrs.build_id() == h, "libptrs_hip.so was not built from this tree: run __graft_entry__.build()"
    stale = [w for w in ("cornell", "colonnade", "classroom") if json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_%s.json" % w)))["_meta"]["source_hash"] != h]
    if stale:  # not a defect of the product: bench.py then reports its live timings without counters ("counters_usable": false)
        pytest.skip("profiles/r04_pmc_{%s}.json were measured on other kernel sources than this tree's (%s): re-run tools/final_profiles.sh" % (",".join(stale), h))


def test_valu_ceiling_table_is_the_measured_one():
    j = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_ceiling.json")))
    best = lambda inst: max(r["per_cycle_per_simd"] for r in j["rows"] if r["inst"] == inst and r["chains"].startswith("8"))
    assert 0.40 < best("v_fma_f32") < 0.55 and 0.40 < best("v_add_f32") < 0.55          # full-rate class: ~1 per 2.2 cycles and SIMD
    assert 0.20 < best("v_max_f32") < 0.36 and 0.20 < best("v_cndmask_b32_e64 (mask in an SGPR pair)") < 0.36  # half-rate class (0.24 at 2-6 waves, up to 0.32 at 8)
    at4 = lambda inst: next(r["per_cycle_per_simd"] for r in j["rows"] if r["inst"] == inst and r["chains"].startswith("8") and r["waves_per_simd"] == 4)
    assert abs(1.0 / at4("v_max_f32") - bench.CLASS_B_CYCLES) < 0.5 and abs(1.0 / at4("v_cmp_lt_f32_e64 -> SGPR pair") - bench.CLASS_B_CYCLES) < 0.5  # the model's 4.2 cycles: the rate at 2-6 waves per SIMD
    assert at4("mix v_max_f32 : v_fma_f32 = 1 : 1") > 0.38  # the two classes issue side by side


def test_fetch_size_calibration_is_what_the_reports_apply():
    """profiles/r04_fetch_calib.json (tools/fetch_calib.hip under rocprofv3 on MI355X): FETCH_SIZE is half of TCC_MISS x 128 B for this
    library's two access shapes, Infinity-Cache-resident tables included -- and every r04 counter summary doubles it accordingly."""
    j = json.load(open(os.path.join(ROOT, "profiles", "r04_fetch_calib.json")))
    c = j["_conclusions"]
    assert abs(c["factor_stream_16B_per_lane"] - 2.0) < 0.01
    assert all(abs(v - 2.0) < 0.01 for v in c["factor_gather_128B_records_vs_l2_misses"].values()) and c["infinity_cache_hits_counted"] is True
    for case in ("stream_2GB", "gather_64MB_dependent", "gather_2GB_dependent"):
        k = j[case]["counters"]
        assert abs(k["FETCH_SIZE"] * 1024.0 / (k["TCC_MISS_sum"] * 64.0) - 1.0) < 0.01            # = TCC_MISS x 64 B
        assert abs(k["TCC_EA0_RDREQ_128B_sum"] / k["TCC_EA0_RDREQ_sum"] - 1.0) < 0.01                # every memory-side read is a 128-byte request
        assert abs(k["TCC_EA0_RDREQ_DRAM_sum"] / k["TCC_EA0_RDREQ_sum"] - 1.0) < 0.01                # "DRAM" names the address space: no counter separates the Infinity Cache
    assert j["gather_64MB_dependent"]["record_visits_per_s"] > j["gather_2GB_dependent"]["record_visits_per_s"] > 3e10
    for w in ("cornell", "colonnade", "classroom", "trace-colonnade", "trace-classroom"):
        raw = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_%s.json" % w)))
        for name, k in raw.items():
            if not name.startswith("_"):
                assert k["fetch_size_factor"] == 2.0 and abs(k["hbm_bytes"] - (k["fetch_bytes_calibrated"] + k["write_bytes"])) <= 1e-6 * max(k["hbm_bytes"], 1.0)
