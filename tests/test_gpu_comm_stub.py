"""The C entry points over RCCL (csrc/wx_comm.hip) against a RECORDING stub of the RCCL table (tests/stubs/stub_rccl.c): no box of this
pool has two GPUs, so `wx_allgatherv_out_*`, `wx_allgather_out_*` and `wx_allreduce_moments_*` have only ever run with a one-rank
communicator.  Here a fresh process loads the library with WX_RCCL_LIB pointing at the stub, builds
communicators of 2, 3 and 5 ranks for EVERY rank in turn, calls the entry points with ragged counts on real device buffers and checks
what reached ncclSend / ncclRecv: peers, element counts, byte offsets into `recv`, data types, one group, send posted before the
matching recv slot -- and that the rank's own piece was copied to its offset (a real device copy) while nothing else of `recv` changed.
The schedule is the one of distributed.OverlappedAllGather, whose data movement the gloo tests cover on CPU (tests/test_distributed_gloo.py);
what is checked here is the index arithmetic of the C side (C1 of SURVEY 8e: all-gather of the reconstructed output only)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch
import waveletsext_jl_amd as wx
from waveletsext_jl_amd import _lib
L = _lib.lib()
stub = ctypes.CDLL(os.environ["WX_RCCL_LIB"], mode=ctypes.RTLD_GLOBAL)     # the same handle the library dlopens
stub.stub_call.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6
out = []
idb = ctypes.create_string_buffer(128)
assert L.wx_comm_unique_id(idb) == 0
for dt, suf, esz, nccl_dt in ((torch.float64, "f64", 8, 8), (torch.float32, "f32", 4, 7)):
    for nranks, counts in ((2, [5, 3]), (3, [4, 0, 7]), (5, [1, 2, 3, 4, 5]), (3, [0, 0, 6]), (2, [0, 0])):
        total = sum(counts)
        for rank in range(nranks):
            comm = ctypes.c_void_p()
            assert L.wx_comm_init(nranks, rank, idb, ctypes.byref(comm)) == 0
            recv = torch.full((max(total, 1),), -1.0, dtype=dt, device="cuda")
            send = torch.arange(1, counts[rank] + 1, dtype=dt, device="cuda") + 100 * rank if counts[rank] else torch.empty(1, dtype=dt, device="cuda")
            cnt = (ctypes.c_int64 * nranks)(*counts)
            stub.stub_reset()
            fn = getattr(L, "wx_allgatherv_out_" + suf)
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            rc = fn(ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), cnt, nranks, comm,
                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            calls = []
            for i in range(stub.stub_ncalls()):
                kind, ptr, count, dtype, peer, depth = ctypes.c_int(), ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                stub.stub_call(i, ctypes.byref(kind), ctypes.byref(ptr), ctypes.byref(count), ctypes.byref(dtype), ctypes.byref(peer), ctypes.byref(depth))
                base = "send" if ptr.value == send.data_ptr() else "recv"
                calls.append({"kind": kind.value, "base": base, "off": (ptr.value or 0) - (send.data_ptr() if base == "send" else recv.data_ptr()),
                              "count": count.value, "dtype": dtype.value, "peer": peer.value, "depth": depth.value})
            out.append({"suf": suf, "esz": esz, "nccl_dt": nccl_dt, "nranks": nranks, "rank": rank, "counts": counts, "rc": rc,
                        "groups": stub.stub_ngroups(), "depth_after": stub.stub_depth(), "calls": calls,
                        "recv": recv.cpu().numpy().astype(np.float64).tolist(), "send": send.cpu().numpy().astype(np.float64).tolist()})
            assert L.wx_comm_destroy(comm) == 0
    # the equal-count forms
    comm = ctypes.c_void_p()
    assert L.wx_comm_init(4, 2, idb, ctypes.byref(comm)) == 0
    send = torch.ones(6, dtype=dt, device="cuda")
    recv = torch.zeros(24, dtype=dt, device="cuda")
    stub.stub_reset()
    g = getattr(L, "wx_allgather_out_" + suf)
    g.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    rc1 = g(ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), 6, comm, None)
    r = getattr(L, "wx_allreduce_moments_" + suf)
    r.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    rc2 = r(ctypes.c_void_p(recv.data_ptr()), 24, comm, None)
    kinds = []
    for i in range(stub.stub_ncalls()):
        kind, ptr, count, dtype, peer, depth = ctypes.c_int(), ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        stub.stub_call(i, ctypes.byref(kind), ctypes.byref(ptr), ctypes.byref(count), ctypes.byref(dtype), ctypes.byref(peer), ctypes.byref(depth))
        kinds.append([kind.value, count.value, dtype.value, peer.value, ptr.value == (send.data_ptr() if kind.value == 3 else recv.data_ptr())])
    out.append({"suf": suf, "nccl_dt": nccl_dt, "equal": True, "rc": [rc1, rc2], "kinds": kinds})
    assert L.wx_comm_destroy(comm) == 0
print("STUBLOG " + json.dumps(out))
'''


@pytest.fixture(scope="module")
def stub_so(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("stub") / "librccl_stub.so")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "stubs", "stub_rccl.c")])
    return so


def test_comm_entry_points_against_the_recording_stub(stub_so):
    env = dict(os.environ, WX_RCCL_LIB=stub_so)
    env.pop("WX_KNOBS", None)                     # the path override is not a knob: it must work without WX_KNOBS=1
    p = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("STUBLOG ")][-1]
    recs = json.loads(line[len("STUBLOG "):])
    ragged = [r for r in recs if not r.get("equal")]
    assert len(ragged) == 2 * (2 + 3 + 5 + 3 + 2)
    for r in ragged:
        counts, rank, nranks, esz = r["counts"], r["rank"], r["nranks"], r["esz"]
        assert r["rc"] == 0 and r["depth_after"] == 0
        total, off = sum(counts), sum(counts[:rank])
        mine = counts[rank]
        # the own piece sits at its offset, every other element of recv is untouched (the stub moves nothing)
        exp = [-1.0] * max(total, 1)
        exp[off:off + mine] = r["send"][:mine]
        assert r["recv"] == exp, (nranks, rank, counts)
        if total == 0:
            assert r["calls"] == [] and r["groups"] == 0
            continue
        assert r["groups"] == 1 and all(c["depth"] == 1 for c in r["calls"])
        sends = [c for c in r["calls"] if c["kind"] == 1]
        rcvs = [c for c in r["calls"] if c["kind"] == 2]
        peers = [q for q in range(nranks) if q != rank]
        # one send of the whole own piece to every other rank (none for an empty piece), from `send` itself
        assert [c["peer"] for c in sends] == (peers if mine else [])
        assert all(c["count"] == mine and c["base"] == "send" and c["off"] == 0 and c["dtype"] == r["nccl_dt"] for c in sends)
        # one receive per other rank with a non-empty piece, straight into recv at that rank's element offset
        want = [(q, counts[q], sum(counts[:q]) * esz) for q in peers if counts[q]]
        assert [(c["peer"], c["count"], c["off"]) for c in rcvs] == want
        assert all(c["base"] == "recv" and c["dtype"] == r["nccl_dt"] for c in rcvs)
        # per peer the send is posted before the receive (both sides of a pair post in the same order: no deadlock inside the group)
        for q in peers:
            idx = [i for i, c in enumerate(r["calls"]) if c["peer"] == q]
            assert [r["calls"][i]["kind"] for i in idx] == ([1] if mine else []) + ([2] if counts[q] else [])
    for r in (r for r in recs if r.get("equal")):
        assert r["rc"] == [0, 0]
        # ncclAllGather(send, recv, 6, dt), then ncclAllReduce(recv, recv, 24, dt, sum) in place
        assert r["kinds"] == [[3, 6, r["nccl_dt"], -1, True], [4, 6, r["nccl_dt"], -1, True], [5, 24, r["nccl_dt"], 0, True], [6, 24, r["nccl_dt"], 0, True]]
