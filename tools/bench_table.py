#!/usr/bin/env python3
"""One table of the committed bench lines: profiles/<prefix>_bench_*.json -> profiles/<prefix>_summary.md
usage: python tools/bench_table.py r03"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(pfx):
    rows = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pfx + "_bench_*.json"))):
        w = os.path.basename(f)[len(pfx) + 7:-5]
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r, i = d["roofline"], d.get("inverse") or {}
        tr = r.get("traffic")
        alg = r.get("algorithmic_bytes_per_launch")
        first = "%.3f ms, %s = %.3f of %s" % (r["avg_launch_ms"], r["kernel"], r["frac"], "HBM peak" if r["bound"] == "hbm" else "FP64 peak (minimal flops)")
        if tr and alg:
            first += ", PMC traffic %.3f x" % (tr / (alg * r.get("launches_per_step", 1) if False else alg))
        second = "%.3f ms" % i["avg_launch_ms"] if i else "-"
        if i.get("frac") is not None and r["bound"] == "hbm":
            second += " = %.3f" % i["frac"]
        cpu = d.get("cpu_baseline", {})
        rows.append("| %s | %.0f | %.3f | %s | %s | %s |" % (w, d["value"], d["ms_per_step"], first, second,
                                                           ("%.4g (%d thread)" % (cpu["value"], cpu.get("cores", 1))) if cpu else "-"))
    out = ["# bench.py lines of round %s (one MI355X), from profiles/%s_bench_*.json" % (pfx, pfx), "",
           "| workload | Msamples/s | ms per step | first leg (dominant kernel, fraction of its roofline) | second leg | CPU port, Msamples/s |",
           "|---|---|---|---|---|---|"] + rows
    open(os.path.join(ROOT, "profiles", pfx + "_summary.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
