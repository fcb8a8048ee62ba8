"""The persistent 2-D launch (k_lat2d_fused_f32, csrc/wx_lattice2d.h) under the conditions a spin-waiting kernel must survive: ragged and tiny
batches (groups that are not full, batches that are their own ring, batches that reuse ring slots many times), odd batches of 256 x 256 images,
repeated calls on a dirtied scratch pool, and TWO launches in flight at once on two streams from two host threads (each has its own ring
and counters; a workgroup only ever waits for smaller tickets of its own launch, which are held by running workgroups -- whatever share of the chip
the other launch occupies).  Results: round trip <= 2e-6, identical from call to call, identical to the same call made alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rt(wx, torch, m, L, B, wt, seed):
    x = wx.jl_empty((m, m, B), torch.float32, "cuda")
    x.normal_(generator=torch.Generator(device="cuda").manual_seed(seed))
    y = wx.wptall(x, wt, L)
    xr = wx.iwptall(y, wt, L)
    err = float(((xr - x).abs().amax(dim=(0, 1)) / x.abs().max()).max())
    return x, y, err


@pytest.mark.parametrize("geom", [(512, 6), (256, 5), (1024, 7)])
def test_ragged_batches_round_trip_and_repeat(wx, geom):
    import torch
    m, L = geom
    wt = wx.wavelet(wx.WT.db4)
    rng = np.random.default_rng(m)
    per_gib = (1 << 30) // (4 * m * m)
    sizes = [1, 2, 3, 7, 8, 9, 63, 64, 65, 191, 192, 193, 257, 385, 600]
    sizes = [b for b in sizes if b <= per_gib * 2] + [int(v) for v in rng.integers(1, min(700, per_gib * 2), 6)]
    for B in sizes:
        x, y0, err = _rt(wx, torch, m, L, B, wt, B)
        assert err <= 3e-6, (m, B, err)
        junk = torch.full((4 * 1024 * 1024,), 3.0, device="cuda")            # dirty what the pool hands out next
        del junk
        y1 = wx.wptall(x, wt, L)
        assert torch.equal(y0, y1), (m, B)


def test_two_launches_in_flight(wx):
    import torch
    wt = wx.wavelet(wx.WT.db4)
    cases = [(512, 6, 300, 11), (256, 5, 777, 12), (1024, 7, 40, 13), (512, 6, 129, 14)]
    alone = {}
    xs = {}
    for m, L, B, seed in cases:
        x, y, err = _rt(wx, torch, m, L, B, wt, seed)
        assert err <= 3e-6
        alone[(m, B)] = y
        xs[(m, B)] = x
    torch.cuda.synchronize()
    bad = []

    def worker(keys, reps):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(reps):
                for (m, L, B, seed) in keys:
                    y = wx.wptall(xs[(m, B)], wt, L)
                    xr = wx.iwptall(y, wt, L)
                    st.synchronize()
                    if not torch.equal(y, alone[(m, B)]):
                        bad.append((m, B, "forward differs"))
                    e = float(((xr - xs[(m, B)]).abs().amax(dim=(0, 1)) / xs[(m, B)].abs().max()).max())
                    if e > 3e-6:
                        bad.append((m, B, e))

    t1 = threading.Thread(target=worker, args=(cases[:2], 6))
    t2 = threading.Thread(target=worker, args=(cases[2:], 6))
    t1.start(); t2.start()
    t1.join(300); t2.join(300)
    assert not t1.is_alive() and not t2.is_alive(), "a launch did not finish"
    assert not bad, bad[:5]
