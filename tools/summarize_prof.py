#!/usr/bin/env python3
"""Condense a gpurun_out/prof_* directory (rocprofv3 --kernel-trace --stats and separate --pmc
FETCH_SIZE / WRITE_SIZE passes) into small committed files under profiles/.

usage: tools/summarize_prof.py gpurun_out/prof_r01 profiles/r01_cfg2
HBM traffic per launch follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes, so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")[:90]


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    out = {"source": src}
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    rows = []
    if stats:
        for r in csv.DictReader(open(stats[0])):
            rows.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                         "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"]), "pct": float(r["Percentage"])})
    out["kernel_stats"] = rows
    traffic = collections.defaultdict(dict)
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        fs = glob.glob(os.path.join(src, tag, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == ctr:
                k = short(r["Kernel_Name"])
                agg[k].append(float(r["Counter_Value"]))
                meta[k] = {"grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]),
                           "lds": int(r["LDS_Block_Size"]), "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"])}
        for k, v in agg.items():
            traffic[k][ctr + "_KiB_avg"] = sum(v) / len(v)
            traffic[k]["launches_" + ctr] = len(v)
            traffic[k].update(meta[k])
    for k, t in traffic.items():
        f = t.get("FETCH_SIZE_KiB_avg")
        w = t.get("WRITE_SIZE_KiB_avg")
        if f is not None and w is not None:
            t["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    out["pmc"] = traffic
    with open(dst + ".json", "w") as fjson:
        json.dump(out, fjson, indent=1)
    with open(dst + ".md", "w") as md:
        md.write("# rocprofv3 summary (%s)\n\n" % src)
        md.write("## --kernel-trace --stats\n\n| kernel | calls | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|\n")
        for r in rows:
            md.write("| `%s` | %d | %.3f | %.3f | %.3f | %.2f |\n" % (r["kernel"], r["calls"], r["avg_ns"] / 1e6,
                                                                      r["min_ns"] / 1e6, r["max_ns"] / 1e6, r["pct"]))
        md.write("\n## --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)\n\n"
                 "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE half-count correction).\n\n"
                 "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch | grid | wg | LDS B | VGPR | SGPR |\n|---|---|---|---|---|---|---|---|---|\n")
        for k, t in traffic.items():
            if "hbm_bytes_per_launch" in t:
                md.write("| `%s` | %.1f | %.1f | %.4e | %d | %d | %d | %d | %d |\n" % (
                    k, t["FETCH_SIZE_KiB_avg"], t["WRITE_SIZE_KiB_avg"], t["hbm_bytes_per_launch"], t["grid"], t["wg"],
                    t["lds"], t["vgpr"], t["sgpr"]))
    print("wrote", dst + ".json", dst + ".md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
