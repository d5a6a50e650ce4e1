#!/usr/bin/env python3
"""Per-kernel digest of one tools/prof.sh directory: time, HBM bytes, VALU issue, lanes, waits.

    python tools/prof_report.py gpurun_out/prof_TAG [--json OUT.json] [--frames N]

Units (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (FETCH_SIZE is NOT doubled here:
the gfx950 x2 correction is calibrated for 16 B/lane streaming reads only, these kernels gather); SQ_WAVE_CYCLES,
SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed over waves; SQ_BUSY_CYCLES summed over the 32 shader engines;
GRBM_GUI_ACTIVE summed over the 8 XCDs.
"""
import csv
import json
import re
import sys
from collections import defaultdict

SIMDS = 1024.0  # 256 CUs x 4


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name[:40]


def load(d):
    k = defaultdict(dict)
    for r in csv.DictReader(open(d + "/kernel_stats.csv")):
        n = short(r["Name"])
        k[n]["calls"] = int(r["Calls"]); k[n]["total_ms"] = float(r["TotalDurationNs"]) / 1e6; k[n]["avg_ms"] = float(r["AverageNs"]) / 1e6
        k[n]["pct"] = float(r["Percentage"])
    for r in csv.DictReader(open(d + "/pmc.csv")):
        k[r["kernel"]][r["counter"]] = float(r["sum"])
        k[r["kernel"]]["pmc_launches"] = int(r["launches"])
    return k


def digest(k):
    out = {}
    for n, v in k.items():
        if not n.startswith("k_") or "total_ms" not in v or "SQ_WAVE_CYCLES" not in v:
            continue
        g = lambda c: v.get(c, 0.0)
        cyc = g("GRBM_GUI_ACTIVE") / 8.0  # cycles the kernel's dispatches were in flight (profiled run)
        hbm = (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
        o = {
            "calls": v["calls"], "total_ms": v["total_ms"], "avg_ms": v["avg_ms"], "pct_of_gpu_time": v["pct"],
            "hbm_bytes": hbm, "hbm_bytes_per_launch": hbm / max(v["calls"], 1), "hbm_gbs": hbm / (v["total_ms"] * 1e-3) / 1e9 if v["total_ms"] else 0.0,
            "hbm_counter_frac_of_8TBs": hbm / (v["total_ms"] * 1e-3) / 8e12 if v["total_ms"] else 0.0,
            "lanes_per_valu_inst": g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_INSTS_VALU"), 1.0),
            "valu_busy_frac": 4.0 * g("SQ_ACTIVE_INST_VALU") / SIMDS / max(cyc, 1.0),
            "valu_insts": g("SQ_INSTS_VALU"), "salu_insts": g("SQ_INSTS_SALU"), "lds_insts": g("SQ_INSTS_LDS"), "vmem_rd_insts": g("SQ_INSTS_VMEM_RD"), "vmem_wr_insts": g("SQ_INSTS_VMEM_WR"),
            "wave_wait_frac": g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0), "wave_issue_frac": g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0),
            "wave_stall_frac": g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0),
            "waves_resident_per_simd": 4.0 * g("SQ_WAVE_CYCLES") / max(cyc, 1.0) / SIMDS,
            "lds_conflict_cycles_per_lds_inst": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_INSTS_LDS"), 1.0),
            "l2_hit_rate": g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0),
            "clock_ghz_profiled": cyc / (v["total_ms"] * 1e-3) / 1e9 if v["total_ms"] else 0.0,
        }
        out[n] = o
    return out


def main():
    d = sys.argv[1]
    dg = digest(load(d))
    if "--json" in sys.argv:
        arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
        dg["_meta"] = {"frames": int(arg("--frames", "1")), "workload": arg("--workload", "?"), "commit": arg("--commit", "?"),
                       "command": "tools/prof.sh: rocprofv3 --kernel-trace --stats and separate --pmc passes of `python3 bench.py --workload W --steps FRAMES --profile` (one pipeline lane)",
                       "units": "hbm_bytes = (FETCH_SIZE + WRITE_SIZE) KiB x 1024 as reported (FETCH_SIZE not doubled: 16-byte gathers, uncalibrated); *_frac / lanes from the SQ counters, see tools/prof_report.py"}
        json.dump(dg, open(arg("--json", "pmc.json"), "w"), indent=1, sort_keys=True)
        dg.pop("_meta")
    print("%-34s %5s %8s %7s %6s %7s %6s %6s %6s %6s %6s %6s %5s" % ("kernel", "calls", "total_ms", "avg_ms", "pct", "HBM GB/s", "lanes", "VALUbz", "wait", "issue", "w/SIMD", "LDScf", "L2hit"))
    for n, o in sorted(dg.items(), key=lambda kv: -kv[1]["total_ms"]):
        print("%-34s %5d %8.2f %7.3f %6.2f %7.0f %6.1f %6.2f %6.2f %6.2f %6.2f %6.2f %5.2f" % (n, o["calls"], o["total_ms"], o["avg_ms"], o["pct_of_gpu_time"], o["hbm_gbs"], o["lanes_per_valu_inst"],
              o["valu_busy_frac"], o["wave_wait_frac"], o["wave_issue_frac"], o["waves_resident_per_simd"], o["lds_conflict_cycles_per_lds_inst"], o["l2_hit_rate"]))


if __name__ == "__main__":
    main()
