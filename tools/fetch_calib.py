#!/usr/bin/env python3
"""tools/fetch_calib.py -- runs tools/bin/fetch_calib's cases (known byte counts in this library's two access shapes) plainly for the
timing and under `rocprofv3 --pmc` (one counter set per run; no trace domains beside --pmc) for the L2 / fabric counters, and writes
profiles/r04_fetch_calib.json: per case what FETCH_SIZE, TCC_MISS, TCC_EA0_RDREQ* report against the bytes the kernel is known to
move, and from them (a) the factor that turns FETCH_SIZE into bytes for each shape, (b) whether requests served by the Infinity Cache
are in it.  GPU box only:   python3 tools/fetch_calib.py [OUT.json]"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin", "fetch_calib")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04_fetch_calib.json")
CASES = [("stream_2GB", ["stream", "2048"]),
         ("gather_16MB", ["gather", "16", "8", "0"]), ("gather_64MB", ["gather", "64", "8", "0"]), ("gather_2GB", ["gather", "2048", "8", "0"]),
         ("gather_64MB_dependent", ["gather", "64", "8", "1"]), ("gather_2GB_dependent", ["gather", "2048", "8", "1"]),
         ("gather_2GB_6vec_dependent", ["gather", "2048", "6", "1"]), ("gather_64MB_6vec_dependent", ["gather", "64", "6", "1"])]
SETS = [["FETCH_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_READ_sum", "TCC_REQ_sum"],
        ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
        ["TCC_EA0_RDREQ_DRAM_sum", "TCC_READ_SECTORS_sum", "TCC_BUBBLE_sum"]]


def pmc(args, counters, tag):
    d = os.path.join(ROOT, "gpurun_out", "fc_" + tag)
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", BIN] + args, capture_output=True, text=True, env=env, timeout=300)
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            k = row["Kernel_Name"]
            if "k_stream" in k or "k_gather" in k:  # (not the fill / evict kernels)
                vals[row["Counter_Name"]] = vals.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    subprocess.run(["rm", "-rf", d])
    if not vals:
        vals["_error"] = (p.stderr or p.stdout)[-300:]
    return vals


def main():
    res = {}
    for name, args in CASES:
        runs = [json.loads(subprocess.run([BIN] + args, capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1]) for _ in range(3)]
        c = min(runs, key=lambda r: r["ms"])  # unprofiled timing: best of three
        c["counters"] = {}
        for i, s in enumerate(SETS):
            c["counters"].update(pmc(args, s, "%s_%d" % (name, i)))
        k = c["counters"]
        known = c.get("bytes_read_known") or c["record_visits"] * 128.0  # gather: every visit touches one whole 128-byte line (8 vectors; 6 vectors touch 96 B of it)
        c["bytes_lines_known"] = known
        if "FETCH_SIZE" in k:
            c["fetch_size_bytes"] = k["FETCH_SIZE"] * 1024.0  # rocprofv3 reports KiB
            c["fetch_size_over_known_line_bytes"] = c["fetch_size_bytes"] / known
        if "TCC_MISS_sum" in k:
            c["l2_miss_x128_bytes"] = k["TCC_MISS_sum"] * 128.0
            c["l2_hit_rate"] = k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k["TCC_MISS_sum"], 1.0)
            if "fetch_size_bytes" in c:
                c["fetch_size_over_l2_miss_x128"] = c["fetch_size_bytes"] / max(c["l2_miss_x128_bytes"], 1.0)
        if "TCC_EA0_RDREQ_sum" in k:
            c["ea_rdreq_bytes_by_size"] = 32.0 * k.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0 * k.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 128.0 * k.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        res[name] = c
        print(name, json.dumps({q: c.get(q) for q in ("ms", "gbs", "requested_gbs", "record_visits_per_s", "fetch_size_over_known_line_bytes", "fetch_size_over_l2_miss_x128", "l2_hit_rate")}), flush=True)
    s, g64, g2 = res["stream_2GB"], res["gather_64MB"], res["gather_2GB"]
    concl = {"what": "factor = known bytes / FETCH_SIZE bytes for each shape; infinity_cache_hits_counted: the 64 MB table is served on-die (it is far below 256 MiB and re-read ~40 times) yet its FETCH_SIZE per L2 miss equals the 2 GB table's"}
    if "fetch_size_bytes" in s:
        concl["factor_stream_16B_per_lane"] = s["bytes_read_known"] / s["fetch_size_bytes"]
    if "fetch_size_bytes" in g2 and "l2_miss_x128_bytes" in g2:
        concl["factor_gather_128B_records_vs_l2_misses"] = {"2GB": g2["l2_miss_x128_bytes"] / g2["fetch_size_bytes"], "64MB": g64["l2_miss_x128_bytes"] / max(g64.get("fetch_size_bytes", 0.0), 1.0)}
        concl["infinity_cache_hits_counted"] = bool(g64.get("fetch_size_bytes", 0.0) > 0.5 * g64["l2_miss_x128_bytes"] / concl["factor_gather_128B_records_vs_l2_misses"]["2GB"])
    res["_conclusions"] = concl
    json.dump(res, open(OUT, "w"), indent=1)
    print(json.dumps(concl))


if __name__ == "__main__":
    main()
