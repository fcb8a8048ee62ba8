#!/usr/bin/env python3
"""Instruction histogram per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only).

usage: python tools/isa_hist.py file.s [name-substring]
"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        c = collections.Counter()
        for line in body.split("\n"):
            line = line.strip()
            if not line or line[0] in ";." or line.endswith(":"):
                continue
            c[line.split()[0]] += 1
        cls = collections.Counter()
        for k, v in c.items():
            if k.startswith("v_pk_fma") or k.startswith("v_fma") or k.startswith("v_pk_mul") or k.startswith("v_mul"):
                cls["fma/mul"] += v
            elif "dpp" in k:
                cls["dpp"] += v
            elif k.startswith("ds_"):
                cls["lds"] += v
            elif k.startswith("global_") or k.startswith("buffer_") or k.startswith("scratch_"):
                cls["vmem:" + k.split("_")[0]] += v
            elif k.startswith("v_"):
                cls["valu other"] += v
            elif k.startswith("s_"):
                cls["salu"] += v
            else:
                cls["other"] += v
        print(name, sum(c.values()), dict(cls))
        for k, v in c.most_common(22):
            print("    %-28s %d" % (k, v))


if __name__ == "__main__":
    main()
