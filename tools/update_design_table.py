#!/usr/bin/env python3
"""Rewrite the rows of DESIGN.md section 6's table from profiles/r01_bench_*.json (the committed bench lines)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def B(w):
    return json.loads(open(os.path.join(ROOT, "profiles", "r01_bench_%s.json" % w)).read())


def main():
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    rows = {}
    d = B('cfg2'); r = d['roofline']
    rows['cfg2'] = f"| cfg2 `wpdall`+`iwpdall` 65536x4096 f64 db8 L=12 | **{d['value']:.0f}** | {r['avg_launch_ms']:.2f} ms, {r['achieved']/1000:.2f} TB/s | **{100*r['frac']:.1f} % of HBM peak** | {d['inverse']['avg_launch_ms']:.2f} ms | {d['cpu_baseline']['value']:.1f} |"
    d = B('target'); r = d['roofline']
    rows['target'] = f"| target `wptall`+`iwptall` 65536x4096 f64 db4 L=10 | {d['value']:.0f} | {r['avg_launch_ms']:.2f} ms, {r['achieved']/1000:.2f} TB/s | {100*r['frac']:.0f} % HBM (FP64-bound) | {d['inverse']['avg_launch_ms']:.2f} ms | {d['cpu_baseline']['value']:.1f} |"
    d = B('target_haar'); r = d['roofline']
    rows['target_haar'] = f"| target_haar `wptall`+`iwptall` 65536x4096 f64 haar L=10 | **{d['value']:.0f}** | {r['avg_launch_ms']:.2f} ms, {r['achieved']/1000:.2f} TB/s | **{100*r['frac']:.1f} % of HBM peak** | {d['inverse']['avg_launch_ms']:.2f} ms ({d['inverse']['achieved_GBs']/80:.0f} %) | {d['cpu_baseline']['value']:.1f} |"
    d = B('cfg3'); r = d['roofline']
    rows['cfg3'] = f"| cfg3 `swptall`+`iswptall` 64x16384 f64 haar L=12 | {d['value']:.1f} | {r['avg_launch_ms']:.2f} ms, {r['achieved']/1000:.2f} TB/s | {100*r['frac']:.0f} % HBM | {d['inverse']['avg_launch_ms']:.2f} ms | {d['cpu_baseline']['value']:.4f} |"
    d = B('cfg4'); r = d['roofline']
    rows['cfg4'] = f"| cfg4 2-D `wptall`+`iwptall` 512x(512x512) f32 db4 L=6 | {d['value']:.0f} | {r['avg_launch_ms']:.2f} ms | {100*r['frac']:.1f} % HBM (2 passes, LDS/FP bound) | {d['inverse']['avg_launch_ms']:.2f} ms | {d['cpu_baseline']['value']:.1f} |"
    d = B('cfg5'); r = d['roofline']
    rows['cfg5'] = f"| cfg5 acwpd+JBB 2048x2048 f64 coif6 L=11 | {d['value']:.0f} | {r['avg_launch_ms']:.1f} ms, {r['achieved']:.0f} TFLOP/s-equiv. | {100*r['frac']:.0f} % FP64 peak (LDS-issue bound) | {d['inverse']['avg_launch_ms']:.2f} ms (costs+tree) | {d['cpu_baseline']['value']:.4f} |"
    d = B('bb'); r = d['roofline']
    rows['bb'] = f"| bb `bestbasistreeall(·, BB())` 16384x4096 f64 (+`wpdall`) | {d['value']:.0f} | {r['avg_launch_ms']:.2f} ms (costs + all trees) | {100*r['frac']:.0f} % HBM (VALU-bound entropy terms) | {d['inverse']['avg_launch_ms']:.2f} ms (`wpdall`) | {d['cpu_baseline']['value']:.1f} |"
    d = B('ldb'); r = d['roofline']
    rows['ldb'] = f"| ldb `energy_map` 4 classes 16384x4096 f64 (+`wpdall`) | {d['value']:.0f} | {r['avg_launch_ms']:.2f} ms (class kernel + norms + host labels) | {100*r['frac']:.0f} % HBM | {d['inverse']['avg_launch_ms']:.2f} ms (`wpdall`) | {d['cpu_baseline']['value']:.1f} |"
    d = B('siwt'); r = d['roofline']
    rows['siwt'] = f"| siwt `siwpdall` (+costs) / best basis + `isiwpdall` 4096x1024 f64 db4 L=10 d=3 | {d['value']:.0f} | {r['avg_launch_ms']:.2f} ms, {r['achieved']/1000:.2f} TB/s | {100*r['frac']:.0f} % HBM (entropy terms VALU-bound) | {d['inverse']['avg_launch_ms']:.2f} ms (trees + inverse) | {d['cpu_baseline']['value']:.4f} |"
    out = []
    for ln in s.split('\n'):
        m = re.match(r'\| (cfg2|target_haar|target|cfg3|cfg4|cfg5|bb|ldb|siwt) ', ln)
        out.append(rows[m.group(1)] if m else ln)
    open(p, "w").write('\n'.join(out))


if __name__ == "__main__":
    main()
