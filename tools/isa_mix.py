#!/usr/bin/env python3
"""Static instruction mix of selected kernels of a gfx950 assembly listing (hipcc -S --cuda-device-only): FP64 arithmetic (with /
without a DPP operand), DPP moves, other VALU, LDS, VMEM, scalar.  usage: python tools/isa_mix.py file.s substring [substring ...]"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    wants = sys.argv[2:]
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if wants and not any(w in name for w in wants):
            continue
        c = collections.Counter()
        for line in body.split("\n"):
            line = line.strip()
            if not line or line[0] in ";." or line.endswith(":"):
                continue
            op = line.split()[0]
            dpp = "row_" in line or "quad_perm" in line or "wave_" in line or "_dpp" in op
            if re.match(r"v_(fma|mul|add|fmac)_f64", op):
                c["FP64 fma/mul/add" + (", DPP operand" if dpp else "")] += 1
            elif op.startswith("v_mov") and dpp:
                c["v_mov with DPP"] += 1
            elif op.startswith("v_"):
                c["other VALU"] += 1
            elif op.startswith("ds_"):
                c["LDS"] += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"):
                c["VMEM"] += 1
            elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
                c["s_waitcnt / s_nop / barrier"] += 1
            elif op.startswith("s_"):
                c["SALU / branch"] += 1
            else:
                c[op] += 1
        print("%s: %d instructions" % (name, sum(c.values())))
        for k, v in c.most_common():
            print("    %-32s %6d" % (k, v))


if __name__ == "__main__":
    main()
