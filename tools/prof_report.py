#!/usr/bin/env python3
"""Per-kernel digest of one tools/prof.sh directory: time, HBM bytes, what the waves did, what bounds the kernel.

    python tools/prof_report.py gpurun_out/prof_TAG [--json OUT.json] [--frames N] [--workload W]

Units (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them.  FETCH_SIZE IS DOUBLED here: calibrated on this
library's own access shapes (tools/fetch_calib.hip -> profiles/r04_fetch_calib.json) it reports exactly half of TCC_MISS x 128 B --
for 16 B/lane streams and for the traversal's gathers of 128-byte records alike, for tables in HBM and for tables that live in the
Infinity Cache alike: it counts the L2's memory-side read requests (128 B tallied as 64), Infinity-Cache hits included.  `hbm_bytes`
is therefore what crossed the fabric below L2 (HBM + Infinity Cache), an UPPER bound of the HBM bytes; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*
in quad-cycles summed over waves; SQ_BUSY_CYCLES summed over the 32 shader engines; GRBM_GUI_ACTIVE summed over the 8 XCDs.

Every fraction printed here is bounded by 1 by construction:
  issue / wait / stall  a wave's cycles spent issuing (SQ_ACTIVE_INST_ANY), in s_waitcnt (SQ_WAIT_ANY), waiting for an issue
                        slot (SQ_WAIT_INST_ANY), each / SQ_WAVE_CYCLES;
  VALU lo..hi           the share of the SIMDs' vector-ALU time the kernel's instructions need, from the MEASURED issue rates of
                        gfx950 (profiles/r03_valu_ceiling.json, tools/valu_ceiling.hip): fp32 add / mul / fma and plain bit
                        operations ("class A") issue beside everything else ("class B": comparisons, v_cndmask, min / max,
                        shifts, integer mads, conversions, fp64, transcendental), one class-B instruction per ~4.2 cycles and
                        SIMD, so n instructions of which nB are class B need 4.2 x max(nB, n / 2) cycles.  The counters split
                        the instructions by kind, not by class: `lo` counts every integer instruction as class A, `hi` as
                        class B; fp32 add / mul / fma are A, conversions, transcendental and fp64 are B, the rest (comparisons,
                        selects, min / max: SQ_INSTS_VALU minus the kinds counted) B.
  LDS                   SQ_LDS_IDX_ACTIVE (LDS-array cycles incl. conflict cycles) / CU-cycles, and the conflict share of them.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

SIMDS, CUS = 1024.0, 256.0  # 256 CUs x 4
CLASS_B_CYCLES = 4.2        # measured: one class-B wave64 instruction per 4.2 cycles and SIMD (v_max_f32, v_cndmask_b32_e64, v_cmp_*: 0.24 / cycle)


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
        fetch = 2.0 * g("FETCH_SIZE") * 1024.0  # calibrated (profiles/r04_fetch_calib.json): FETCH_SIZE = 1/2 x TCC_MISS x 128 B for streams and gathers alike
        hbm = fetch + g("WRITE_SIZE") * 1024.0
        nv = g("SQ_INSTS_VALU")
        fp32 = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32")
        ints = g("SQ_INSTS_VALU_INT32") + g("SQ_INSTS_VALU_INT64")
        have_kinds = "SQ_INSTS_VALU_ADD_F32" in v
        nb_lo = max(nv - fp32 - ints, 0.0) if have_kinds else 0.0   # every integer instruction taken as class A
        nb_hi = max(nv - fp32, 0.0) if have_kinds else nv           # ... as class B
        simd_cycles = max(cyc, 1.0) * SIMDS
        need = lambda nb: CLASS_B_CYCLES * max(nb, nv / 2.0) / simd_cycles
        wc = max(g("SQ_WAVE_CYCLES"), 1.0)
        lds_active = g("SQ_LDS_IDX_ACTIVE")
        o = {
            "calls": v["calls"], "total_ms": v["total_ms"], "avg_ms": v["avg_ms"], "pct_of_gpu_time": v["pct"],
            "hbm_bytes": hbm, "fetch_bytes_calibrated": fetch, "write_bytes": g("WRITE_SIZE") * 1024.0, "l2_miss_bytes": g("TCC_MISS_sum") * 128.0, "fetch_size_factor": 2.0,
            "hbm_bytes_per_launch": hbm / max(v["calls"], 1), "hbm_gbs": hbm / (v["total_ms"] * 1e-3) / 1e9 if v["total_ms"] else 0.0,
            "hbm_counter_frac_of_8TBs": hbm / (v["total_ms"] * 1e-3) / 8e12 if v["total_ms"] else 0.0,
            "lanes_per_valu_inst": g("SQ_THREAD_CYCLES_VALU") / max(nv, 1.0),
            "valu_insts": nv, "valu_fp32_add_mul_fma": fp32, "valu_int": ints, "valu_trans": g("SQ_INSTS_VALU_TRANS_F32"), "valu_cvt": g("SQ_INSTS_VALU_CVT"),
            "valu_fp64": g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64"),
            "valu_pipe_frac_lo": need(nb_lo), "valu_pipe_frac_hi": need(nb_hi), "valu_class_b_share_lo": nb_lo / max(nv, 1.0), "valu_class_b_share_hi": nb_hi / max(nv, 1.0),
            "valu_insts_per_cycle_simd": nv / simd_cycles,
            "salu_insts": g("SQ_INSTS_SALU"), "salu_per_valu": g("SQ_INSTS_SALU") / max(nv, 1.0), "smem_insts": g("SQ_INSTS_SMEM"), "branch_insts": g("SQ_INSTS_BRANCH"),
            "lds_insts": g("SQ_INSTS_LDS"), "vmem_rd_insts": g("SQ_INSTS_VMEM_RD"), "vmem_wr_insts": g("SQ_INSTS_VMEM_WR"),
            "wave_wait_frac": g("SQ_WAIT_ANY") / wc, "wave_issue_frac": g("SQ_ACTIVE_INST_ANY") / wc, "wave_stall_frac": g("SQ_WAIT_INST_ANY") / wc,
            "wave_wait_lds_frac": g("SQ_WAIT_INST_LDS") / wc,
            "waves_resident_per_simd": 4.0 * g("SQ_WAVE_CYCLES") / max(cyc, 1.0) / SIMDS,
            "lds_busy_frac": lds_active / (max(cyc, 1.0) * CUS) if lds_active else None,
            "lds_conflict_share": g("SQ_LDS_BANK_CONFLICT") / lds_active if lds_active else None,
            "lds_conflict_cycles_per_lds_inst": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_INSTS_LDS"), 1.0),
            "l2_hit_rate": g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0),
            "clock_ghz_profiled": cyc / (v["total_ms"] * 1e-3) / 1e9 if v["total_ms"] else 0.0,
        }
        # what bounds the kernel, from the bounded quantities above
        mid = 0.5 * (o["valu_pipe_frac_lo"] + o["valu_pipe_frac_hi"])
        if o["hbm_counter_frac_of_8TBs"] >= 0.5:
            o["bound"] = "hbm"
        elif mid >= 0.6:
            o["bound"] = "valu-issue"
        elif o["lds_busy_frac"] and o["lds_busy_frac"] >= 0.6:
            o["bound"] = "lds"
        elif o["wave_wait_frac"] >= 0.5:
            o["bound"] = "waitcnt (memory / LDS latency the resident waves do not cover)"
        else:
            o["bound"] = "mixed: vector issue %.2f, waitcnt %.2f of wave time" % (mid, o["wave_wait_frac"])
        out[n] = o
    return out


def main():
    d = sys.argv[1]
    dg = digest(load(d))
    arg = lambda k, dflt: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else dflt
    if "--json" in sys.argv:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        try:
            import importlib
            src = importlib.import_module("pathtracer-rs_amd").build_id()  # of the library the profiled runs loaded (PTRS_LIB or the in-tree build)
        except Exception:
            src = "?"
        dg["_meta"] = {"frames": int(arg("--frames", "1")), "workload": arg("--workload", "?"), "source_hash": src,
                       "command": "tools/prof.sh: rocprofv3 --kernel-trace --stats and separate --pmc passes of `python3 bench.py --workload W --steps FRAMES --profile` (one pipeline lane)",
                       "units": "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: FETCH_SIZE calibrated on this library's access shapes (profiles/r04_fetch_calib.json: exactly half of TCC_MISS x 128 B for streams and for gathers of 128-byte records, Infinity-Cache hits included) -- bytes below L2 (HBM + Infinity Cache), an upper bound of HBM bytes; fractions and their bounds: tools/prof_report.py"}
        json.dump(dg, open(arg("--json", "pmc.json"), "w"), indent=1, sort_keys=True)
        dg.pop("_meta")
    f = lambda x: "  -  " if x is None else "%5.2f" % x
    print("%-34s %5s %8s %7s %6s %8s %6s %11s %5s %5s %5s %6s %6s %6s %6s %6s %6s %5s  %s" % ("kernel", "calls", "total_ms", "avg_ms", "pct", "belowL2", "lanes", "VALU lo..hi", "issue", "wait", "stall", "LDSstl", "w/SIMD", "S/V", "VM/V", "LDSbz", "LDScf", "L2hit", "bound"))
    for n, o in sorted(dg.items(), key=lambda kv: -kv[1]["total_ms"]):
        print("%-34s %5d %8.2f %7.3f %6.2f %8.0f %6.1f %5.2f..%4.2f %5.2f %5.2f %5.2f %6.3f %6.2f %6.2f %6.3f %s %6.2f %5.2f  %s" % (
            n, o["calls"], o["total_ms"], o["avg_ms"], o["pct_of_gpu_time"], o["hbm_gbs"], o["lanes_per_valu_inst"], o["valu_pipe_frac_lo"], o["valu_pipe_frac_hi"],
            o["wave_issue_frac"], o["wave_wait_frac"], o["wave_stall_frac"], o["wave_wait_lds_frac"], o["waves_resident_per_simd"], o["salu_per_valu"],
            (o["vmem_rd_insts"] + o["vmem_wr_insts"]) / max(o["valu_insts"], 1.0), f(o["lds_busy_frac"]),
            o["lds_conflict_cycles_per_lds_inst"], o["l2_hit_rate"], o["bound"]))
    print("(belowL2: GB/s of 2 x FETCH_SIZE + WRITE_SIZE, bytes below L2 = HBM + Infinity Cache; LDSstl: SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, the LDS share of `stall`; VM/V: vector-memory per vector-ALU instruction)")


if __name__ == "__main__":
    main()
