#!/usr/bin/env python3
"""Summary of tools/fp64_issue.sh: per kernel the launch time, the GPU clock it ran at (GRBM_GUI_ACTIVE / time) and the share of the
launch's cycles the vector ALUs were issuing (SQ_INSTS_VALU x 4 cycles per wavefront instruction / 1024 SIMDs / cycles).  rocprofv3
sums GRBM_GUI_ACTIVE over the 8 XCDs: cycles = GRBM_GUI_ACTIVE / 8 (checked on the pure-FMA loop: 2.13-2.37 GHz).
usage: python tools/fp64_issue.py <dir>"""
import collections
import csv
import glob
import re
import sys

csv.field_size_limit(1 << 30)
SIMDS = 256 * 4
XCDS = 8


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][-60:]


def main():
    root = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        run = f[len(root):].strip("/").split("/")[0].rsplit("_p", 1)[0]
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not (k.startswith("k_lat") or "fma64" in k or "k_fma" in k or k.startswith("k_acwpd")):
                continue
            if run == "ubench":
                k += " grid %s" % r["Grid_Size"]
            acc[(run, k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[(run, k)]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("%-10s %-36s %8s %9s %9s %11s %9s %9s %9s" % ("run", "kernel", "ms", "Mcycles", "clk GHz", "VALU/wave", "VALU busy", "LDS/wave", "any/wave"))
    for (run, k), c in sorted(acc.items()):
        avg = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else float("nan")
        ms, gui, valu, waves = avg("_ms"), avg("GRBM_GUI_ACTIVE") / XCDS, avg("SQ_INSTS_VALU"), avg("SQ_WAVES")
        print("%-10s %-36s %8.3f %9.3f %9.3f %11.0f %9.3f %9.0f %9.0f" % (
            run, k[:36], ms, gui / 1e6, gui / (ms * 1e6), valu / waves, valu * 4 / SIMDS / gui,
            avg("SQ_INSTS_LDS") / waves, avg("SQ_ACTIVE_INST_ANY") / waves))


if __name__ == "__main__":
    main()
