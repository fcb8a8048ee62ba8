"""Timeline of HIP API calls and kernels from a `rocprofv3 --hip-trace --kernel-trace --output-format csv` directory: for the LAST
occurrence of every kernel sequence between two hipEventRecord calls (one leg of bench.py) print each kernel's start, duration and the
idle gap on the device before it, and the host API calls issued inside the leg with their durations.  usage: trace_gaps.py <dir> [n_legs]"""
import csv
import glob
import sys


def load(d, pat):
    f = glob.glob(d + "/**/*" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def main():
    d = sys.argv[1]
    nlegs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    api = load(d, "hip_api_trace.csv")
    ker = load(d, "kernel_trace.csv")
    if not api or not ker:
        print("missing csv in", d)
        return
    A = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in api))
    K = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in ker))
    # legs = spans between consecutive hipEventRecord calls on the host
    rec = [a for a in A if a[2] == "hipEventRecord"]
    print("api calls %d, kernels %d, hipEventRecord %d" % (len(A), len(K), len(rec)))
    legs = [(rec[i][0], rec[i + 1][1]) for i in range(len(rec) - 1)]
    legs = [l for l in legs if any(l[0] <= a[0] <= l[1] and a[2].startswith("hipLaunch") or a[2] == "hipModuleLaunchKernel" and l[0] <= a[0] <= l[1] for a in A)]
    ki = 0
    # host order == device order on one stream: the kernels of a leg are the next len(launches) kernels
    launches_before = 0
    islaunch = lambda n: ("Launch" in n) and ("Kernel" in n or "GGL" in n)
    all_launch = [a for a in A if islaunch(a[2])]
    for (t0, t1) in legs[-nlegs:]:
        calls = [a for a in A if t0 <= a[0] <= t1]
        nl_before = sum(1 for a in all_launch if a[0] < t0)
        nl = sum(1 for a in calls if islaunch(a[2]))
        ks = K[nl_before:nl_before + nl] if len(K) == len(all_launch) else []
        print("\n=== leg: host span %.3f ms, %d API calls, %d launches" % ((t1 - t0) / 1e6, len(calls), nl))
        for a in calls:
            print("  host +%9.3f ms  %-40s %9.3f ms" % ((a[0] - t0) / 1e6, a[2], (a[1] - a[0]) / 1e6))
        if ks:
            prev = None
            print("  device: first kernel starts %.3f ms after the leg's first host call; span %.3f ms; busy %.3f ms" % (
                (ks[0][0] - t0) / 1e6, (ks[-1][1] - ks[0][0]) / 1e6, sum(k[1] - k[0] for k in ks) / 1e6))
            for k in ks:
                gap = (k[0] - prev) / 1e6 if prev else 0.0
                print("  dev  +%9.3f ms  gap %8.3f  dur %9.3f ms  %s" % ((k[0] - ks[0][0]) / 1e6, gap, (k[1] - k[0]) / 1e6, k[2][:90]))
                prev = k[1]
        else:
            print("  (kernel count %d != launch count %d: no per-leg device rows)" % (len(K), len(all_launch)))


main()
