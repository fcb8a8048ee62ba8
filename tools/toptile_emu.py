"""CPU emulation of the index arithmetic of csrc/wx_toptile.h (plan, windows, halos, which nodes leave / enter where),
written from the kernel so that a wrong offset shows up here, without a GPU: tests/test_toptile_cpu.py compares it with the
oracle for full trees, pyramids and random trees.  Not a product path."""
import numpy as np


def plan(inverse, n, NL, F, split, deep, TS):
    TL = min(TS >> NL, n >> NL)
    assert TL >= 2 and (n >> NL) % TL == 0
    HF = F // 2
    g = [0] * 5
    if not inverse:
        for d in range(NL - 1, -1, -1):
            u = 1 << (NL - d)
            g[d] = g[d + 1] + ((F - 2) + u - 1) // u
    else:
        for d in range(NL):
            u = 1 << (NL - d - 1)
            g[d + 1] = g[d] + ((HF - 1) + u - 1) // u
        g[NL] += g[NL] & 1
    W = [(TL + 2 * g[d]) << (NL - d) if d <= NL else 0 for d in range(5)]
    ex = 1
    for i in range(1, 1 << NL):
        if (ex >> (i - 1)) & 1 and (split >> (i - 1)) & 1:
            ex |= (1 << (2 * i - 1)) | (1 << (2 * i))
    split &= ex & ((1 << ((1 << NL) - 1)) - 1)
    return dict(NL=NL, TL=TL, g=g, W=W, split=split, exists=ex, deep=deep)


def fwd(x, q, NL, split, deep, TS=64):
    """one signal: returns (dst, deeparr) in wpt layout"""
    n, F = x.size, len(q)
    P = plan(False, n, NL, F, split, deep, TS)
    H = (F - 2) // 2
    dst, dp = np.full(n, np.nan), np.full(n, np.nan)
    TL, g, W = P["TL"], P["g"], P["W"]
    for t0 in range(0, n >> NL, TL):
        cur = np.array([x[((t0 - g[0]) * (1 << NL) + e) % n] for e in range(W[0])])
        for l in range(1, NL + 1):
            Wc, Wp = W[l], W[l - 1]
            nxt = np.full((1 << l) * Wc, np.nan)
            r = ((g[l - 1] - g[l]) << (NL - l + 1)) - (F - 2)
            first = (1 << (l - 1)) - 1
            for j in range(1 << (l - 1)):
                if not (P["split"] >> (first + j)) & 1:
                    continue
                pb, ab, db = j * Wp + r, 2 * j * Wc, 2 * j * Wc + Wc
                groups = (Wc + H + 3) // 4
                for gi in range(groups):
                    w = [cur[pb + 8 * gi + e] if pb + 8 * gi + e < cur.size else np.nan for e in range(8 + F - 2)]
                    c0 = gi * 4 - H
                    for p in range(4):
                        a = sum(q[k] * w[2 * p + k] for k in range(F))
                        d = sum((-q[k] if k & 1 else q[k]) * w[2 * p + F - 1 - k] for k in range(F))
                        c = c0 + p
                        if 0 <= c < Wc:
                            nxt[ab + c] = a
                        if c + H < Wc:
                            nxt[db + c + H] = d
            firstc = (1 << l) - 1
            core = TL << (NL - l)
            for j in range(1 << l):
                if not (P["exists"] >> (firstc + j)) & 1:
                    continue
                if l < NL and (P["split"] >> (firstc + j)) & 1:
                    continue
                out = dp if (l == NL and (deep >> j) & 1) else dst
                o = j * (n >> l) + (t0 << (NL - l))
                b = j * Wc + (g[l] << (NL - l))
                out[o:o + core] = nxt[b:b + core]
            cur = nxt
    return dst, dp


def inv(src, dp, q, NL, split, deep, TS=64):
    n, F = src.size, len(q)
    P = plan(True, n, NL, F, split, deep, TS)
    HF = F // 2
    TL, g, W = P["TL"], P["g"], P["W"]
    out = np.full(n, np.nan)
    for t0 in range(0, n >> NL, TL):
        bufs = {}

        def enter(l):
            b = bufs.setdefault(l, np.full((1 << l) * W[l], np.nan))
            firstc, np_ = (1 << l) - 1, n >> l
            for j in range(1 << l):
                if not (P["exists"] >> (firstc + j)) & 1:
                    continue
                if l < NL and (P["split"] >> (firstc + j)) & 1:
                    continue
                arr = dp if (l == NL and (deep >> j) & 1) else src
                v0 = (t0 - g[l]) << (NL - l)
                for e in range(W[l]):
                    b[j * W[l] + e] = arr[j * np_ + (v0 + e) % np_]
        enter(NL)
        for l in range(NL, 0, -1):
            if l - 1 >= 1:
                enter(l - 1)
            chi = bufs[l]
            par = bufs.setdefault(l - 1, np.full((1 << (l - 1)) * W[l - 1], np.nan))
            Wc, Wp = W[l], W[l - 1]
            offi = (g[l] - g[l - 1]) << (NL - l)
            pairs = Wp >> 1
            first = (1 << (l - 1)) - 1
            for j in range(1 << (l - 1)):
                if not (P["split"] >> (first + j)) & 1:
                    continue
                pb, ab, db = j * Wp, 2 * j * Wc + offi - (HF - 1), 2 * j * Wc + Wc + offi
                for k in range(pairs):
                    v0 = sum(q[2 * m] * chi[ab + k + HF - 1 - m] - q[2 * m + 1] * chi[db + k + m] for m in range(HF))
                    v1 = sum(q[2 * m + 1] * chi[ab + k + HF - 1 - m] + q[2 * m] * chi[db + k + m] for m in range(HF))
                    par[pb + 2 * k], par[pb + 2 * k + 1] = v0, v1
        out[t0 << NL:(t0 + TL) << NL] = bufs[0][:TL << NL]
    return out
