#!/usr/bin/env python3
"""Writes tests/golden/oracle_outputs.json: outputs of the CPU oracle on seeded inputs at reduced sizes of the five
BASELINE configurations (plus the north-star target at its real length), stored bit-exactly as hexadecimal IEEE words
together with the inputs.  SURVEY 8(c): "Golden fixtures to commit ... Float64 hex".

The reference itself cannot run here (pure Julia, no Julia in the image), so these vectors pin the ORACLE: a silent
change to oracle/wx_oracle.c moves tests/test_golden_cpu.py (bit-exact), and the HIP path is checked against the same
committed numbers (tests/test_gpu_golden.py) rather than against whatever the oracle computes that day.

    python tools/gen_golden.py            # rewrites the fixture (only after a deliberate oracle change)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
OUT = os.path.join(ROOT, "tests", "golden", "oracle_outputs.json")


def enc(a):
    a = np.asfortranarray(a)
    if a.dtype == np.bool_:
        return {"dtype": "bool", "shape": list(a.shape), "bits": "".join("1" if v else "0" for v in a.ravel(order="F"))}
    w = {np.dtype(np.float64): (np.uint64, 16), np.dtype(np.float32): (np.uint32, 8)}[a.dtype]
    words = a.ravel(order="F").view(w[0])
    return {"dtype": str(a.dtype), "shape": list(a.shape), "hex": "".join(format(int(v), "0%dx" % w[1]) for v in words)}


def dec(d):
    if d["dtype"] == "bool":
        return np.array([c == "1" for c in d["bits"]], dtype=bool).reshape(d["shape"], order="F")
    w = {"float64": (np.uint64, 16), "float32": (np.uint32, 8)}[d["dtype"]]
    h = d["hex"]
    words = np.array([int(h[i:i + w[1]], 16) for i in range(0, len(h), w[1])], dtype=w[0])
    return np.asfortranarray(words.view(d["dtype"]).reshape(d["shape"], order="F"))


def cases(O, wx):
    """name -> (description, inputs dict, function producing the outputs dict)"""
    q = lambda name: np.asarray(wx.wavelet(getattr(wx.WT, name)).qmf, dtype=np.float64)
    rng = np.random.default_rng(20261003)
    out = {}

    def add(name, desc, inputs, fn):
        out[name] = {"desc": desc, "inputs": inputs, "fn": fn}

    x = rng.standard_normal(1024)
    add("cfg1_wpt", "config 1: wpt / iwpt n=1024 f64 db4 L=8, one signal", {"x": x, "wavelet": "db4", "L": 8},
        lambda i: {"wpt": O.wpt(i["x"], q("db4"), 8), "iwpt_of_wpt": O.iwpt(O.wpt(i["x"], q("db4"), 8), q("db4"), 8)})
    x = np.asfortranarray(rng.standard_normal((128, 2)))
    add("cfg2_wpdall", "config 2 reduced: wpdall / iwpdall n=128 f64 db8 full tree L=7, 2 signals", {"x": x, "wavelet": "db8", "L": 7},
        lambda i: {"wpd": O.wpdall(i["x"], q("db8"), 7), "iwpd_of_wpd": O.iwpdall(O.wpdall(i["x"], q("db8"), 7), q("db8"), 7)})
    x = np.asfortranarray(rng.standard_normal((4096, 1)))
    add("target_wptall", "north-star target at its real length: wptall / iwptall n=4096 f64 db4 L=10, 1 signal",
        {"x": x, "wavelet": "db4", "L": 10},
        lambda i: {"wpt": O.wptall(i["x"], q("db4"), 10)})
    x = np.asfortranarray(rng.standard_normal((64, 2)))
    add("cfg3_swptall", "config 3 reduced: swpt / iswpt (average-based) n=64 f64 haar L=4, 2 signals", {"x": x, "wavelet": "haar", "L": 4},
        lambda i: {"swpt": np.stack([O.swpt(i["x"][:, b], q("haar"), 4) for b in range(2)], axis=-1),
                   "iswpt_of_swpt": np.stack([O.iswpt(O.swpt(i["x"][:, b], q("haar"), 4), q("haar")) for b in range(2)], axis=-1)})
    x = np.asfortranarray(rng.standard_normal((32, 32, 2)).astype(np.float32))
    add("cfg4_wpt2d", "config 4 reduced: 2-D wptall / iwptall 32x32 f32 db4 L=3, 2 images", {"x": x, "wavelet": "db4", "L": 3},
        lambda i: {"wpt": O.wptall(i["x"], q("db4"), 3)})
    x = np.asfortranarray(rng.standard_normal((64, 8)))
    def cfg5(i):
        X = np.asfortranarray(np.stack([O.acwpd(i["x"][:, b], q("coif6"), 5) for b in range(8)], axis=-1))
        return {"acwpd_signal0": X[:, :, 0], "sum": X.sum(axis=2), "sumsq": (X ** 2).sum(axis=2),
                "costs": O.tree_costs_jbb(X, redundant=True), "tree": O.bestbasistree_jbb(X, redundant=True)}
    add("cfg5_acwpd_jbb", "config 5 reduced: acwpd + JBB costs + tree n=64 f64 coif6 L=5, 8 signals", {"x": x, "wavelet": "coif6", "L": 5}, cfg5)
    return out


def compute():
    import wx_oracle as O
    import waveletsext_jl_amd as wx
    O.build()
    res = {}
    for name, c in cases(O, wx).items():
        outs = c["fn"](c["inputs"])
        res[name] = {"desc": c["desc"],
                     "inputs": {k: (enc(v) if isinstance(v, np.ndarray) else v) for k, v in c["inputs"].items()},
                     "outputs": {k: enc(np.asarray(v)) for k, v in outs.items()}}
    return res


def recompute_from(fixture):
    """outputs of today's oracle on the fixture's stored inputs (used by the CPU test)"""
    import wx_oracle as O
    import waveletsext_jl_amd as wx
    O.build()
    res = {}
    fns = cases(O, wx)
    for name, c in fixture.items():
        inputs = {k: (dec(v) if isinstance(v, dict) else v) for k, v in c["inputs"].items()}
        res[name] = fns[name]["fn"](inputs)
    return res


if __name__ == "__main__":
    res = compute()
    with open(OUT, "w") as f:
        json.dump({"generator": "tools/gen_golden.py (oracle/wx_oracle.c, seed 20261003)", "cases": res}, f, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
