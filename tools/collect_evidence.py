#!/usr/bin/env python3
"""File the output of tools/refresh_evidence.sh under profiles/:
usage: tools/collect_evidence.py <tag> <round-prefix>      e.g.  tools/collect_evidence.py r01c r01
 profiles/<prefix>_<workload>.{md,json}   rocprofv3 summaries (tools/summarize_prof.py)
 profiles/<prefix>_bench_<workload>.json  the bench.py line of the same build
 profiles/traffic.json                    HBM bytes per launch of every wx kernel (PMC passes), read by bench.py"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import summarize_prof  # noqa: E402


def build_src():
    import re
    sys.path.insert(0, ROOT)
    import waveletsext_jl_amd as wx
    m = re.search(r"src=(\w+)", wx.build_info())
    return m.group(1) if m else None


def main(tag, prefix):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}      # workloads not in this run keep their entries
    for w in ("cfg2", "wpt_db8", "target", "target_n2048", "target_n1024", "target_haar", "tree_random", "tree_pyramid", "cfg3", "cfg3_sdwt", "swpt_db4",
              "cfg4", "cfg4_256", "cfg4_1024", "cfg5", "bb", "ldb", "siwt", "dwt_long", "target_f32", "denoise"):
        src = os.path.join(ROOT, "gpurun_out", "prof_" + tag, w)
        if os.path.isdir(src):
            dst = os.path.join(ROOT, "profiles", "%s_%s" % (prefix, w))
            summarize_prof.main(os.path.relpath(src, ROOT), dst)
            pm = json.load(open(dst + ".json")).get("pmc", {})
            traffic[w] = {k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                              "source": "profiles/%s_%s.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                                        "FETCH_SIZE doubled per MI355X_MICROARCH.md)" % (prefix, w)}
                          for k, v in pm.items() if k.startswith("k_") and "hbm_bytes_per_launch" in v}
            # the counters belong to the build that was profiled -- the library in this tree, this script runs right behind the
            # profiler (tools/refresh_evidence.sh): its source digest goes with them, bench.py reports `traffic` for that digest only
            traffic[w]["_build_src"] = build_src()
        b = os.path.join(ROOT, "gpurun_out", "bench_" + tag, w + ".json")
        if os.path.exists(b) and os.path.getsize(b):
            line = open(b).read().strip().splitlines()[-1]
            json.loads(line)
            open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (prefix, w)), "w").write(line + "\n")

    json.dump(traffic, open(tpath, "w"), indent=1)
    log = os.path.join(ROOT, "gpurun_out", "pytest_gpu_%s.log" % tag)
    if os.path.exists(log):
        shutil.copy(log, os.path.join(ROOT, "profiles", "%s_pytest_gpu.log" % prefix))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
