"""Prototype (numpy, CPU) of the lattice formulation used by csrc/wx_lattice.hip.

A level of the packet transform (dwt/dwt_one_level.jl:79-107) is the 2x2 paraunitary polyphase matrix
    [a]   [ Qe(w)      Qo(w)    ] [v_even]        Qe(w) = sum q[2m] w^m,  Qo(w) = sum q[2m+1] w^m,
    [d] = [-Qo(1/w)    Qe(1/w)  ] [v_odd ]        w = advance by one pair (periodic)
which factors into J+1 = F/2 plane rotations c_j [[1, t_j], [-t_j, 1]] separated by "advance the odd channel by one
pair" -- F multiply-adds per pair instead of 2F.  The whole depth-L full-tree transform is then L in-place stencils
on the signal (level l acts on index bit l-1 with dilation 2^(l-1), period n) followed by a bit reversal of the
packet index, one common scale (prod c_j)^L at the end.

Run:  python tools/lattice_proto.py      (compares with oracle.wpt / iwpt; test infrastructure only)
"""
import sys, os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def lattice_factor(q):
    """q -> (t[0..J], scale) with G(w) = prod_j c_j [[1,t_j],[-t_j,1]] (advance of channel 1 between stages)."""
    q = np.asarray(q, dtype=np.longdouble)
    F = q.size
    assert F % 2 == 0
    J = F // 2 - 1
    # G(w) = [[Qe, Qo], [-w^J Qo(1/w), w^J Qe(1/w)]], coefficient matrices G[k] of w^k
    G = np.zeros((J + 1, 2, 2), dtype=np.longdouble)
    for m in range(J + 1):
        G[m, 0, 0] = q[2 * m]
        G[m, 0, 1] = q[2 * m + 1]
        G[J - m, 1, 0] = -q[2 * m + 1]
        G[J - m, 1, 1] = q[2 * m]
    ts, cs = [], []
    for deg in range(J, 0, -1):
        top = G[deg]
        # rotation R = [[c, s], [-s, c]]; R^T G: row0 = c*G0 - s*G1 must lose its w^deg term
        k = np.argmax(np.abs(top[0]) + np.abs(top[1]))
        a0, a1 = top[0, k], top[1, k]
        r = np.hypot(a0, a1)
        # c*a0 - s*a1 = 0  ->  (c, s) = (a1, a0)/r
        c, s = a1 / r, a0 / r
        Rt = np.array([[c, -s], [s, c]], dtype=np.longdouble)
        H = np.einsum("ij,kjl->kil", Rt, G[: deg + 1])
        # H row 0 has degree deg-1, H row 1 has no constant term; G' = Lambda(w)^-1 H
        Gn = np.zeros((deg, 2, 2), dtype=np.longdouble)
        Gn[:, 0, :] = H[:deg, 0, :]
        Gn[:, 1, :] = H[1 : deg + 1, 1, :]
        resid = max(np.abs(H[deg, 0]).max(), np.abs(H[0, 1]).max())
        assert resid < 1e-12, resid
        G = Gn
        ts.append(s / c)
        cs.append(c)
    R0 = G[0]
    c, s = R0[0, 0], R0[0, 1]
    assert abs(R0[1, 0] + s) < 1e-12 and abs(R0[1, 1] - c) < 1e-12, R0
    ts.append(s / c)
    cs.append(c)
    ts.reverse()
    cs.reverse()
    return np.array(ts, dtype=np.float64), float(np.prod(np.array(cs, dtype=np.longdouble)))


def level_fwd(x, b, t):
    """one packet level in place on index bit b (x: (n,) natural slots)"""
    n = x.size
    s = 1 << b
    J = len(t) - 1
    p = np.arange(n)
    up = p[(p >> b) & 1 == 0]            # u slots; partner v slot = up + s
    u = x[up].copy()
    v = x[up + s].copy()
    # "advance v by one pair" = v at u-slot p comes from slot p + s + 2s
    def adv(v, k):
        src = (up + s + k * 2 * s) % n
        full = np.zeros(n)
        full[up + s] = v
        return full[src]
    for j in range(J + 1):
        u, v = u + t[j] * v, v - t[j] * u
        if j < J:
            v = adv(v, 1)
    v = adv(v, -J)
    y = np.empty_like(x)
    y[up] = u
    y[up + s] = v
    return y


def level_inv(x, b, t):
    n = x.size
    s = 1 << b
    J = len(t) - 1
    p = np.arange(n)
    up = p[(p >> b) & 1 == 0]
    u = x[up].copy()
    v = x[up + s].copy()
    def adv(v, k):
        src = (up + s + k * 2 * s) % n
        full = np.zeros(n)
        full[up + s] = v
        return full[src]
    v = adv(v, J)
    for j in range(J, -1, -1):
        u, v = u - t[j] * v, v + t[j] * u
        if j > 0:
            v = adv(v, -1)
    y = np.empty_like(x)
    y[up] = u
    y[up + s] = v
    return y


def out_perm(n, L):
    p = np.arange(n)
    f = p & ((1 << L) - 1)
    j = np.zeros(n, dtype=np.int64)
    for k in range(L):
        j |= ((f >> k) & 1) << (L - 1 - k)
    return j * (n >> L) + (p >> L)


def wpt_lattice(x, q, L):
    t, sc = lattice_factor(q)
    y = np.array(x, dtype=np.float64)
    for b in range(L):
        y = level_fwd(y, b, t)
    out = np.empty_like(y)
    out[out_perm(y.size, L)] = y * sc ** L
    return out


def iwpt_lattice(w, q, L):
    t, sc = lattice_factor(q)
    y = np.asarray(w, dtype=np.float64)[out_perm(len(w), L)]
    for b in range(L - 1, -1, -1):
        y = level_inv(y, b, t)
    return y * sc ** L


if __name__ == "__main__":
    from oracle import wx_oracle as O
    from waveletsext_jl_amd import filters as Fm
    rng = np.random.default_rng(0)
    for name in ["haar", "db2", "db3", "db4", "db5", "db6", "db8", "db10", "coif2", "coif4", "coif6", "sym4", "sym8", "batt4"]:
        try:
            q = np.asarray(Fm.wavelet(name).qmf, dtype=np.float64)
        except Exception as e:  # filter family not in the table
            print(name, "n/a", e)
            continue
        t, sc = lattice_factor(q)
        for n, L in [(64, 6), (256, 5), (4096, 10), (4096, 12)]:
            x = rng.standard_normal(n)
            ref = O.wpt(x, q, L)
            got = wpt_lattice(x, q, L)
            e1 = np.abs(got - ref).max() / np.abs(ref).max()
            back = iwpt_lattice(ref, q, L)
            e2 = np.abs(back - x).max() / np.abs(x).max()
            print(f"{name:6s} F={q.size:2d} n={n:5d} L={L:2d} fwd {e1:.2e} inv {e2:.2e}  max|t|={np.abs(t).max():.3g} scale={sc:.4g}")
