"""Lane-and-register level emulation (numpy) of csrc/wx_lattice.hip: the same layouts, LDS slots and cross-lane moves
as the kernel, one Python statement per kernel statement.  Used to debug the kernel's data movement on the CPU and
as an executable specification of the layouts (python tools/lattice_emu.py compares with the oracle)."""
import os, sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools.lattice_proto import lattice_factor

LANES = np.arange(64)


def rev6(v):
    r = np.zeros_like(v)
    for k in range(6):
        r |= ((v >> k) & 1) << (5 - k)
    return r


def nbr(v, H, d):
    """value held by lane (i + d) within groups of 2^H lanes (what lat_nbr<H, d> returns in lane i)"""
    if H == 0:
        return v.copy()
    g = 1 << H
    src = (LANES & ~(g - 1)) | ((LANES + d) & (g - 1))
    return v[src]


def shear_coefs(t):
    """the rotation c [[1, t], [-t, 1]] as two in-place shears on (u, w = sigma v): u += p w; w -= kap u
    (sigma_0 = 1, sigma_{j+1} = sigma_j / (1 + t_j^2)); returns (p, kap)"""
    t = np.asarray(t, dtype=np.longdouble)
    sig = np.longdouble(1)
    p, kap = [], []
    for tj in t:
        p.append(tj / sig)
        kap.append(sig * tj / (1 + tj * tj))
        sig = sig / (1 + tj * tj)
    return np.array(p, dtype=np.float64), np.array(kap, dtype=np.float64)


def level(x, K, H, t, inv):
    """x: (64 regs, 64 lanes); t = (p, kap)"""
    NSEQ, M, S = 1 << K, 32 >> K, 1 << K
    U = lambda s, m: s + ((2 * m) << K)
    pc, kc = t
    NS = len(pc)

    def advance():
        for s in range(NSEQ):
            first = x[U(s, 0) + S].copy()
            for m in range(M - 1):
                x[U(s, m) + S] = x[U(s, m + 1) + S]
            x[U(s, M - 1) + S] = nbr(first, H, +1)

    def delay():
        for s in range(NSEQ):
            last = x[U(s, M - 1) + S].copy()
            for m in range(M - 1, 0, -1):
                x[U(s, m) + S] = x[U(s, m - 1) + S]
            x[U(s, 0) + S] = nbr(last, H, -1)

    if not inv:
        for j in range(NS):
            for s in range(NSEQ):
                for m in range(M):
                    x[U(s, m)] = x[U(s, m)] + pc[j] * x[U(s, m) + S]
                    x[U(s, m) + S] = x[U(s, m) + S] - kc[j] * x[U(s, m)]
            if j + 1 < NS:
                advance()
        for j in range(NS - 1):
            delay()
    else:
        for j in range(NS - 1):
            advance()
        for j in range(NS - 1, -1, -1):
            for s in range(NSEQ):
                for m in range(M):
                    x[U(s, m) + S] = x[U(s, m) + S] + kc[j] * x[U(s, m)]
                    x[U(s, m)] = x[U(s, m)] - pc[j] * x[U(s, m) + S]
            if j > 0:
                delay()


def lat_pi(L, r):
    e = 0
    for k in range(12 - L):
        e |= ((r >> (L - 6 + k)) & 1) << k
    for j in range(L - 6):
        e |= ((r >> (L - 7 - j)) & 1) << (12 - L + j)
    return e


def lat_pi_inv(L, e):
    for r in range(64):
        if lat_pi(L, r) == e:
            return r
    raise ValueError


def wpt_emu(xsig, t, gain1, L):
    lane = LANES
    lds = np.full(1104, np.nan)
    # L0 loads: instruction (hi3, f) = eight complete 128-byte lines
    xo = 64 * (lane >> 3) + 2 * (lane & 7)
    r = np.empty((32, 2, 64))
    for hi3 in range(8):
        for f in range(4):
            for e in range(2):
                r[4 * hi3 + f, e] = xsig[512 * hi3 + 16 * f + xo + e]
    a = np.empty((64, 64))
    wa, ra = 17 * (lane >> 3) + 2 * (lane & 7), 17 * lane
    for f in range(4):
        for hi3 in range(8):
            for e in range(2):
                lds[wa + 136 * hi3 + e] = r[4 * hi3 + f, e]
        for m in range(16):
            a[16 * f + m] = lds[ra + m]
    level(a, 0, 6, t, False)
    level(a, 1, 6, t, False)
    bb = np.empty((64, 64))
    sw = lane ^ ((lane >> 5) << 1)
    wa0, wa1 = sw, sw ^ 1
    H, p10 = lane & 15, lane >> 4
    lam0 = 4 * H
    sg = (p10 & 1) | ((lam0 >> 5) << 1)
    rah = [64 * p10 + ((lam0 + h) ^ sg) for h in range(4)]
    for f in range(4):
        for j in range(16):
            lds[(wa1 if j & 1 else wa0) + 64 * j] = a[16 * f + j]
        for h in range(4):
            for g in range(4):
                bb[16 * h + 4 * f + g] = lds[rah[h] + 256 * g]
    for K in range(4):
        level(bb, K, 4, t, False)
    c = np.empty((64, 64))
    wa = lane + (lane >> 5)
    ra = 66 * (lane >> 2) + 16 * (lane & 1) + 33 * ((lane >> 1) & 1)
    for f in range(4):
        for j in range(16):
            lds[wa + 66 * j] = bb[16 * f + j]
        for Hh in range(16):
            c[4 * Hh + f] = lds[ra + Hh]
    for K in range(6):
        if L > 6 + K:
            level(c, K, 0, t, False)
    # store
    y = np.full(4096, np.nan)
    ch = rev6(lane)
    c0, c1, c2 = ch & 1, (ch >> 1) & 1, (ch >> 2) & 1
    wrow = (c0 ^ c2) | ((ch >> 3) << 1) | (c1 << 4) | (c2 << 5)
    wa = 17 * wrow
    q = lane >> 3
    q0, q1, q2 = q & 1, (q >> 1) & 1, q >> 2
    ra = 17 * ((q0 ^ q2) + 16 * q1 + 32 * q2) + 2 * (lane & 7)
    yo = 64 * q + 2 * (lane & 7)
    pcl = np.array([bin(v).count("1") for v in lane])
    base = gain1 ** L * (gain1 ** -2.0) ** pcl                       # per lane: g^(L - 2 popcount(lane))
    mask = (1 << (L - 6)) - 1
    for k in range(4):
        for e4 in range(16):
            rr = lat_pi_inv(L, 16 * k + e4)
            lds[wa + e4] = c[rr] * (base * (gain1 ** -2.0) ** bin(rr & mask).count("1"))
        for i in range(8):
            y[512 * i + 16 * k + yo] = lds[ra + 34 * i]
            y[512 * i + 16 * k + yo + 1] = lds[ra + 34 * i + 1]
    return y


def iwpt_emu(w, t, gain1, L):
    lane = LANES
    lds = np.full(1104, np.nan)
    pcl = np.array([bin(v).count("1") for v in lane])
    base = gain1 ** -L * (gain1 ** 2.0) ** pcl
    mask = (1 << (L - 6)) - 1
    ch = rev6(lane)
    rrow = (((ch >> 1) ^ ch) & 1) | ((ch >> 2) << 1) | ((ch & 1) << 5)
    ra = 17 * rrow
    lanepart = (((lane >> 4) ^ (lane >> 3)) & 1) | (((lane >> 5) & 1) << 1) | (((lane >> 3) & 1) << 5)
    wa = 17 * lanepart + 2 * (lane & 7)
    xo = 64 * (lane >> 3) + 2 * (lane & 7)
    c = np.empty((64, 64))
    for k in range(4):
        for i in range(8):
            lds[wa + 68 * i] = w[512 * i + 16 * k + xo]
            lds[wa + 68 * i + 1] = w[512 * i + 16 * k + xo + 1]
        for e4 in range(16):
            rr = lat_pi_inv(L, 16 * k + e4)
            c[rr] = lds[ra + e4] * (base * (gain1 ** 2.0) ** bin(rr & mask).count("1"))
    for K in range(5, -1, -1):
        if L > 6 + K:
            level(c, K, 0, t, True)
    bb = np.empty((64, 64))
    wa = 34 * (lane >> 1) + (lane & 1)
    H, p0, p1 = lane & 15, (lane >> 4) & 1, lane >> 5
    ra = 34 * p1 + 2 * H + p0
    for f in range(4):
        for Hh in range(16):
            lds[wa + 2 * Hh] = c[4 * Hh + f]
        for j in range(16):
            bb[16 * f + j] = lds[ra + 68 * j]
    for K in range(3, -1, -1):
        level(bb, K, 4, t, True)
    a = np.empty((64, 64))
    H, p10 = lane & 15, lane >> 4
    wah = [((h ^ (H >> 2)) | ((H & 3) << 2) | (((H >> 2) & 1) << 4) | ((H >> 3) << 5)) + 64 * p10 for h in range(4)]
    Ha, ha = lane >> 2, lane & 3
    ra = (ha ^ (Ha >> 2)) | ((Ha & 3) << 2) | (((Ha >> 2) & 1) << 4) | ((Ha >> 3) << 5)
    for f in range(4):
        for h in range(4):
            for g in range(4):
                lds[wah[h] + 256 * g] = bb[16 * h + 4 * f + g]
        for j in range(16):
            a[16 * f + j] = lds[ra + 64 * j]
    level(a, 1, 6, t, True)
    level(a, 0, 6, t, True)
    wa, ra = 17 * lane, 17 * (lane >> 3) + 4 * (lane & 7)
    yo = 64 * (lane >> 3) + 2 * (lane & 7)
    y = np.full(4096, np.nan)
    for f in range(4):
        for m in range(16):
            lds[wa + 4 * (m >> 1) + 2 * (m & 1)] = a[16 * f + m]
        for hi3 in range(8):
            for e in range(2):
                y[512 * hi3 + 16 * f + yo + e] = lds[ra + 136 * hi3 + 2 * e]
    return y


if __name__ == "__main__":
    from oracle import wx_oracle as O
    from waveletsext_jl_amd import filters as Fm
    rng = np.random.default_rng(1)
    for name in ("db2", "db4", "db8"):
        q = np.asarray(Fm.wavelet(name).qmf, dtype=np.float64)
        t, g1 = lattice_factor(q)
        t = shear_coefs(t)
        for L in (6, 7, 9, 10, 12):
            x = rng.standard_normal(4096)
            ref = O.wpt(x, q, L)
            got = wpt_emu(x, t, g1, L)
            back = iwpt_emu(ref, t, g1, L)
            print(name, L, "fwd", np.abs(got - ref).max() / np.abs(ref).max(), "inv", np.abs(back - x).max() / np.abs(x).max())


# ---- wpd: every level leaves the registers through an LDS transposition (k_lat_wpd_f64) ---------------------------------
def src_of_pbit(lay, t):
    """('reg' | 'lane', bit index) that holds sample-index bit t in layout lay (0 = A, 2 = B, 6 = C)"""
    if lay == 0:
        return ("reg", t) if t < 6 else ("lane", t - 6)
    if lay == 2:
        if 2 <= t < 8:
            return ("reg", t - 2)
        return ("lane", t - 8) if t >= 8 else ("lane", 4 + t)
    return ("reg", t - 6) if t >= 6 else ("lane", t)


def obit_of_pbit(l, t):
    """where sample-index bit t lands in the position inside column l of the packet table"""
    return 11 - t if t < l else t - l


def emit_plan(lay, l):
    """routing of one level's output: round bits, order of the line-id bits, per-bit weights"""
    reg_o = {}
    lane_o = {}
    for t in range(12):
        kind, i = src_of_pbit(lay, t)
        (reg_o if kind == "reg" else lane_o)[i] = (obit_of_pbit(l, t), t)
    round_regbits = [i for i in range(6) if reg_o[i][0] >= 4][:2]
    assert len(round_regbits) == 2
    vbits = [i for i in range(6) if i not in round_regbits]
    line = []                                        # (kind, bit, obit) in compress order
    for k in range(4):
        if lane_o[k][0] >= 4:
            line.append(("lane", k, lane_o[k][0]))
    for k in (4, 5):
        if lane_o[k][0] >= 4:
            line.append(("lane", k, lane_o[k][0]))
    for i in vbits:
        if reg_o[i][0] >= 4:
            line.append(("reg", i, reg_o[i][0]))
    assert len(line) == 6, (lay, l, line)
    return dict(reg_o=reg_o, lane_o=lane_o, round_regbits=round_regbits, vbits=vbits, line=line)


def emit_level(x, lay, l, g, lds, ycol):
    """x: (64 regs, 64 lanes) after level l in layout lay -> column l (4096 values) of the packet table"""
    P = emit_plan(lay, l)
    lane = LANES
    low = (1 << l) - 1
    # lane parts
    hi_lane = np.zeros(64, dtype=np.int64)
    pos_lane = np.zeros(64, dtype=np.int64)
    pc_lane = np.zeros(64, dtype=np.int64)
    for k in range(6):
        ob, t = P["lane_o"][k]
        bit = (lane >> k) & 1
        if ob < 4:
            pos_lane += bit << ob
        if t < l:
            pc_lane += bit
    for q, (kind, b, ob) in enumerate(P["line"]):
        if kind == "lane":
            hi_lane += ((lane >> b) & 1) << q
    base = g ** l * (g ** -2.0) ** pc_lane
    for rho in range(4):
        # registers of the round: the two round bits fixed
        regs = []
        for v in range(16):
            r = 0
            for j, i in enumerate(P["vbits"]):
                r |= ((v >> j) & 1) << i
            for j, i in enumerate(P["round_regbits"]):
                r |= ((rho >> j) & 1) << i
            regs.append(r)
        o_round = 0
        for j, i in enumerate(P["round_regbits"]):
            o_round |= ((rho >> j) & 1) << P["reg_o"][i][0]
        for r in regs:
            hi_reg = pos_reg = pc_reg = 0
            for i in range(6):
                ob, t = P["reg_o"][i]
                bit = (r >> i) & 1
                if ob < 4:
                    pos_reg |= bit << ob
                if t < l:
                    pc_reg += bit
            for q, (kind, b, ob) in enumerate(P["line"]):
                if kind == "reg":
                    hi_reg |= ((r >> b) & 1) << q
            slot = 17 * (hi_lane + hi_reg) + pos_lane + pos_reg
            lds[slot] = x[r] * (base * (g ** -2.0) ** pc_reg)
        for i in range(8):
            for e in range(2):
                hi = 8 * i + (lane >> 3)
                pos = 2 * (lane & 7) + e
                o = np.full(64, o_round | 0, dtype=np.int64) + pos
                for q, (kind, b, ob) in enumerate(P["line"]):
                    o += ((hi >> q) & 1) << ob
                ycol[o] = lds[17 * hi + pos]


def wpd_emu(xsig, t, gain1, L):
    """(4096, L+1) packet table through the kernel's layouts; column 0 = the signal"""
    lane = LANES
    lds = np.full(1104, np.nan)
    y = np.full((4096, L + 1), np.nan)
    y[:, 0] = xsig
    xo = 64 * (lane >> 3) + 2 * (lane & 7)
    r = np.empty((32, 2, 64))
    for hi3 in range(8):
        for f in range(4):
            for e in range(2):
                r[4 * hi3 + f, e] = xsig[512 * hi3 + 16 * f + xo + e]
    a = np.empty((64, 64))
    wa, ra = 17 * (lane >> 3) + 2 * (lane & 7), 17 * lane
    for f in range(4):
        for hi3 in range(8):
            for e in range(2):
                lds[wa + 136 * hi3 + e] = r[4 * hi3 + f, e]
        for m in range(16):
            a[16 * f + m] = lds[ra + m]
    for K in range(2):
        if L > K:
            level(a, K, 6, t, False)
            emit_level(a, 0, K + 1, gain1, lds, y[:, K + 1])
    if L <= 2:
        return y
    bb = np.empty((64, 64))
    sw = lane ^ ((lane >> 5) << 1)
    wa0, wa1 = sw, sw ^ 1
    H, p10 = lane & 15, lane >> 4
    lam0 = 4 * H
    sg = (p10 & 1) | ((lam0 >> 5) << 1)
    rah = [64 * p10 + ((lam0 + h) ^ sg) for h in range(4)]
    for f in range(4):
        for j in range(16):
            lds[(wa1 if j & 1 else wa0) + 64 * j] = a[16 * f + j]
        for h in range(4):
            for g in range(4):
                bb[16 * h + 4 * f + g] = lds[rah[h] + 256 * g]
    for K in range(4):
        if L > 2 + K:
            level(bb, K, 4, t, False)
            emit_level(bb, 2, 3 + K, gain1, lds, y[:, 3 + K])
    if L <= 6:
        return y
    c = np.empty((64, 64))
    wa = lane + (lane >> 5)
    ra = 66 * (lane >> 2) + 16 * (lane & 1) + 33 * ((lane >> 1) & 1)
    for f in range(4):
        for j in range(16):
            lds[wa + 66 * j] = bb[16 * f + j]
        for Hh in range(16):
            c[4 * Hh + f] = lds[ra + Hh]
    for K in range(6):
        if L > 6 + K:
            level(c, K, 0, t, False)
            emit_level(c, 6, 7 + K, gain1, lds, y[:, 7 + K])
    return y
