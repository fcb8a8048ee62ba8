// wx_toptile.h -- the top levels of a binary wavelet-packet tree on LONG signals, several levels per pass.
//
// Signals that do not fit a wavefront's registers (4096 samples) or a CU's LDS start with "top" levels whose nodes are
// longer than what the register / LDS kernels hold.  Rounds 2-3 ran those one level per launch (k_level1_tile: every level
// reads and writes the whole signal); here up to four levels run in ONE pass through LDS:
//
//   a tile is a time window: TL coefficients of every depth-NL node, i.e. TS = TL << NL input samples.  The workgroup loads
//   the window plus the halo the NL levels need, runs level after level between two LDS buffers, and writes each node's
//   coefficients out when the node is a leaf of the tree (or a depth-NL node).  Windows are kept in units of the node's
//   dilation -- depth d holds [(t0 - g[d]) << (NL - d), (t0 + TL + g[d]) << (NL - d)) of every node -- so every start is
//   aligned and the halo sizes are g[d] = g[d+1] + ceil((F - 2) / 2^(NL - d)) (forward; the inverse grows downwards with
//   HF - 1 = F/2 - 1 instead).
//
//   The tree decides what a node does: `split` bit i-1 (heap index i, depth < NL) = the node is decomposed; a node that
//   exists and is not split is a leaf and leaves / enters at its own depth, at its own place of the output (wpt layout:
//   node j of depth d occupies [j n >> d, (j+1) n >> d), Utils.jl:101-134).  Depth-NL nodes whose bit is set in `deep` go to /
//   come from a second array (the caller's scratch) in the same layout: they are the 4096-sample nodes the lattice
//   kernels continue with.  Full trees, pyramids (dwt / idwt) and any other tree are the same kernel with other masks.
//
//   Arithmetic: direct form, same tap order as dwt_step! / idwt_step! (dwt/dwt_one_level.jl:94-105, 207-221), Float64
//   accumulation, intermediate levels stored in the signal's type (Float32 signals round once per level like the
//   reference's stores).  Each thread produces P = 4 consecutive pairs out of one register window of 2 P + F - 2 LDS
//   reads (a[c] and d[c + (F-2)/2] read the same window), the LDS index is skewed by one word per 2 P so that the lanes'
//   windows start in different banks.
//
//   What it took to get there (each measured on 2 GiB of 8192 ... 65536-sample signals, round 4):
//   * the geometry -- F, NL, tile size, hence every halo, window and slot -- is a template parameter: the offsets inside a
//     window are immediates of the ds_read / ds_write instructions and a group costs one address computation (with run-time
//     geometry the index arithmetic was 5 x the multiply-adds: 730 vector instructions per wavefront and tile for one level);
//   * one LDS array and integer offsets for the two buffers (a `T *bufs[2]` indexed by the level's parity turns every LDS
//     access into a flat instruction, which also waits for the outstanding global loads);
//   * a workgroup walks its tiles in a loop with the NEXT tile's input in registers while this one is computed;
//   * node slots are padded instead of predicating the first / last outputs of a node; lengths are powers of two, so the
//     periodic wrap is a mask.
#pragma once
#include "wx_common.h"
#include <type_traits>
#include <atomic>

constexpr int WX_TT_P = 4;          // pairs per thread and window
constexpr int WX_TT_NT = 256;

// run-time part of a pass: the tree
struct WxTopTree {
    unsigned split;         // bit i-1: heap node i (depth < NL) is decomposed
    unsigned deep;          // bit j: depth-NL node j lives in the `deep` array
    int ns[5];              // ns[l]: split nodes of depth l - 1 (the parents level l works on), l = 1 .. NL
    int nf[5];              // nf[l]: nodes of depth l that are final in this pass (leaves, and every node of depth NL)
    unsigned sp[5];         // the split parents of depth l - 1, four bits each
    unsigned long long fl[5];   // the final nodes of depth l, four bits each
    int pf[7];              // inverse: element pairs that enter a tile, depth by depth: pf[l] .. pf[l+1] belong to depth l
    long long lstride;      // forward, wpd (wx_dev_top_levels_wpd): depth l leaves at dst + l * lstride (slice l of the packet table), and depth 0 -- the
                            // signal itself -- is stored too; 0: wpt layout, only the final nodes leave
};

// compile-time part: the geometry of a tile
template <int F, int NL, int TS, bool INVERSE> struct WxTTGeo {
    static constexpr int H = (F - 2) / 2, HF = F / 2, TL = TS >> NL;
    static constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
    static constexpr int g(int d)
    {
        int v = 0;
        if (!INVERSE) {
            for (int e = NL - 1; e >= d; --e) v += cdiv(F - 2, 1 << (NL - e));
            return v;
        }
        for (int e = 0; e < d; ++e) v += cdiv(HF - 1, 1 << (NL - e - 1));
        if (d == NL) v += v & 1;                    // the depth-NL windows start at t0 - g: even, for the 2-element loads
        return v;
    }
    static constexpr int W(int d) { return (TL + 2 * g(d)) << (NL - d); }                  // window of a node of depth d
    // slot of a node in LDS: its window, room in front for the H outputs a node's first group writes before its window (forward)
    // and behind for what the last group writes past it; a multiple of 8 so that the skew of a slot's start is a constant
    static constexpr int O = INVERSE ? 0 : ((H + 7) / 8) * 8;                               // the window starts O words into its slot
    static constexpr int S(int d) { return ((O + W(d) + H + 2 * WX_TT_P + F + 15) / 8) * 8; }
    static constexpr int bufw()
    {
        int m = 0;
        for (int d = 0; d <= NL; ++d) m = (S(d) << d) > m ? (S(d) << d) : m;
        return m;
    }
    static constexpr int SK = INVERSE ? 2 : 3;                                             // skew: one word per 1 << SK
    static constexpr int boff = bufw() + (bufw() >> SK) + 8;                               // second buffer
    static constexpr size_t lds_bytes(size_t esz) { return (size_t)(2 * boff + 8) * esz; }
    static constexpr int held = cdiv(W(0) / 2, WX_TT_NT);                                  // forward: 2-element loads per lane and tile
};

template <typename T, int NB> struct WxTTRegs {
    typename WxVec2<T>::type v[NB];
};

template <typename T, int F, int NL, int TS>
__global__ __launch_bounds__(WX_TT_NT) void k_top_tile_fwd(const T *__restrict__ src, T *__restrict__ dst, T *__restrict__ deep, int n,
                                                            int64_t sstride, int64_t dstride, int64_t deepstride, unsigned ntiles,
                                                            WxTopTree P, WxFilt filt)
{
    typedef typename WxVec2<T>::type V2;
    typedef WxTTGeo<F, NL, TS, false> G;
    // Float32 signals filter in Float32 like the reference (dwt/dwt_one_level.jl:79-83 is generic in T); until late in round 4 they
    // were widened to Float64 per LDS read: the pass is instruction-bound for Float32 (twice the samples per byte)
    typedef typename std::conditional<sizeof(T) == 8, double, float>::type AT;
    AT qq[F];
#pragma unroll
    for (int kq = 0; kq < F; ++kq) qq[kq] = (AT)filt.q[kq];
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    constexpr int PP = WX_TT_P, H = G::H, WIN = 2 * PP + F - 2, NB = G::held;
    T *lds = reinterpret_cast<T *>(wx_smem);
    const unsigned tiles = (unsigned)(n >> NL) / G::TL;
    const int nm = n - 1;

    WxTTRegs<T, NB> R;
    auto prefetch = [&](unsigned tile) {
        const unsigned sig = tile / tiles;
        const int t0 = (int)(tile - sig * tiles) * G::TL;
        const T *x = src + (int64_t)sig * sstride;
        const int s = (t0 - G::g(0)) << NL;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = threadIdx.x + u * WX_TT_NT;
            if ((u + 1) * WX_TT_NT <= G::W(0) / 2 || f < G::W(0) / 2) R.v[u] = *reinterpret_cast<const V2 *>(x + ((s + 2 * f) & nm));
        }
    };
    unsigned tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile);
    while (tile < ntiles) {
        const unsigned sig = tile / tiles;
        const int t0 = (int)(tile - sig * tiles) * G::TL;
        T *yo = dst + (int64_t)sig * dstride, *yd = deep + (int64_t)sig * deepstride;
        __syncthreads();                                    // the previous tile's last stores have read their buffer
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = threadIdx.x + u * WX_TT_NT;
            if ((u + 1) * WX_TT_NT <= G::W(0) / 2 || f < G::W(0) / 2) {
                const int a = G::O + 2 * f;
                lds[a + (a >> 3)] = R.v[u].x;
                lds[a + 1 + ((a + 1) >> 3)] = R.v[u].y;
            }
        }
        const unsigned next = tile + gridDim.x;
        if (next < ntiles) prefetch(next);                  // in flight while this tile's levels run
        __syncthreads();
        if (P.lstride) {
            // wpd: slice 0 of the table is the signal (DWT.jl:164-209 writes it first): the tile's own samples out of the staged window
            constexpr int c0 = G::O + (G::g(0) << NL);
            T *y0 = yo + ((int64_t)t0 << NL);
            for (int f = threadIdx.x; f < TS / 2; f += WX_TT_NT) {
                const int a = c0 + 2 * f;
                V2 v;
                v.x = lds[a + (a >> 3)];
                v.y = lds[a + 1 + ((a + 1) >> 3)];
                *reinterpret_cast<V2 *>(y0 + 2 * f) = v;
            }
        }
        auto level = [&](auto lc) {
            constexpr int l = decltype(lc)::value;
            constexpr int co = ((l - 1) & 1) ? G::boff : 0, no = (l & 1) ? G::boff : 0;
            constexpr int Wc = G::W(l), Sc = G::S(l), Sp = G::S(l - 1);
            constexpr int r = ((G::g(l - 1) - G::g(l)) << (NL - l + 1)) - (F - 2);   // first parent word of a node's first window
            constexpr int groups = (Wc + H + PP - 1) / PP;
            constexpr int base = G::O + r;                  // in words of the parent's slot
            const int total = P.ns[l] * groups;
            const unsigned spl = P.sp[l];
            for (int idx = threadIdx.x; idx < total; idx += WX_TT_NT) {
                const int k = idx / groups, gi = idx - k * groups;
                const int j = (int)((spl >> (4 * k)) & 15u);
                // parent slot j starts at a multiple of 8: ph(j Sp + base + 8 gi + e) = j (Sp + Sp / 8) + 9 gi + ph-part of (base + e)
                const T *pw = lds + co + j * (Sp + Sp / 8) + 9 * gi;
                AT w[WIN];
#pragma unroll
                for (int e = 0; e < WIN; ++e) w[e] = (AT)pw[base + e + ((base + e) >> 3)];
                // children slots 2 j and 2 j + 1; their windows start O words in; this group writes a[c0 ..] and d[c0 + H ..]
                const int cw = G::O + gi * PP - H;          // >= 0 by the choice of O
                T *aw = lds + no + 2 * j * (Sc + Sc / 8), *dw = aw + (Sc + Sc / 8);
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    AT a = 0, d = 0;
#pragma unroll
                    for (int kk = 0; kk < F; ++kk) {
                        a = fma(qq[kk], w[2 * p + kk], a);
                        d = fma((kk & 1) ? -qq[kk] : qq[kk], w[2 * p + F - 1 - kk], d);
                    }
                    const int ca = cw + p, cd = cw + p + H;
                    aw[ca + (ca >> 3)] = (T)a;               // the first H and the last few land in the slot's padding
                    dw[cd + (cd >> 3)] = (T)d;
                }
            }
            __syncthreads();
            // nodes of depth l that are final in this pass leave: leaves of the tree, and at depth NL every node
            constexpr int lc2 = NL - l;                                       // core of a node: TL << lc2 elements
            constexpr int core2 = (G::TL << lc2) / 2;                        // ... in 2-element stores
            constexpr int gl = G::O + (G::g(l) << lc2);
            const int tot2 = P.nf[l] * core2;
            const unsigned long long fll = P.fl[l];
            for (int f = threadIdx.x; f < tot2; f += WX_TT_NT) {
                const int k = f / core2, e = 2 * (f - k * core2);
                const int j = (int)((fll >> (4 * k)) & 15ull);
                const int a = j * Sc + gl + e;
                V2 v;
                v.x = lds[no + a + (a >> 3)];
                v.y = lds[no + a + 1 + ((a + 1) >> 3)];
                T *o = ((l == NL && ((P.deep >> j) & 1u)) ? yd : yo) + (int64_t)l * P.lstride + (int64_t)j * (n >> l) + ((int64_t)t0 << lc2);
                *reinterpret_cast<V2 *>(o + e) = v;
            }
        };
        level(std::integral_constant<int, 1>{});
        if constexpr (NL >= 2) level(std::integral_constant<int, 2>{});
        if constexpr (NL >= 3) level(std::integral_constant<int, 3>{});
        if constexpr (NL >= 4) level(std::integral_constant<int, 4>{});
        tile = next;
    }
}

template <typename T, int F, int NL, int TS>
__global__ __launch_bounds__(WX_TT_NT) void k_top_tile_inv(const T *__restrict__ src, T *__restrict__ dst, const T *__restrict__ deep, int n,
                                                            int64_t sstride, int64_t dstride, int64_t deepstride, unsigned ntiles,
                                                            WxTopTree P, WxFilt filt)
{
    typedef typename WxVec2<T>::type V2;
    typedef WxTTGeo<F, NL, TS, true> G;
    // Float32 signals filter in Float32 like the reference (dwt/dwt_one_level.jl:79-83 is generic in T); until late in round 4 they
    // were widened to Float64 per LDS read: the pass is instruction-bound for Float32 (twice the samples per byte)
    typedef typename std::conditional<sizeof(T) == 8, double, float>::type AT;
    AT qq[F];
#pragma unroll
    for (int kq = 0; kq < F; ++kq) qq[kq] = (AT)filt.q[kq];
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    constexpr int PP = WX_TT_P, HF = G::HF, WIN = PP + HF - 1;
    // every window that enters a tile -- the leaves of each depth, at depth NL every node -- is one flat list of element pairs
    // (P.pf[l] .. P.pf[l+1] belong to depth l), fetched into registers a tile ahead; `enter(l)` moves depth l's pairs into LDS.
    // At most all 2^NL nodes of depth NL enter (a tree with leaves higher up holds fewer elements).
    constexpr int NB = G::cdiv((G::W(NL) / 2) << NL, WX_TT_NT);
    T *lds = reinterpret_cast<T *>(wx_smem);
    const unsigned tiles = (unsigned)(n >> NL) / G::TL;

    WxTTRegs<T, NB> R;
    int at[NB];                                             // LDS word (unskewed, inside its buffer) | depth << 24; -1: empty
    auto prefetch = [&](unsigned tile) {
        const unsigned sig = tile / tiles;
        const int t0 = (int)(tile - sig * tiles) * G::TL;
        const T *xs = src + (int64_t)sig * sstride, *xd = deep + (int64_t)sig * deepstride;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = threadIdx.x + u * WX_TT_NT;
            at[u] = -1;
            if (f < P.pf[NL + 1]) {
                auto fetch = [&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    constexpr int wl2 = G::W(l) / 2;
                    const int fl = f - P.pf[l];
                    const int k = fl / wl2, e = 2 * (fl - k * wl2);
                    const int j = (int)((P.fl[l] >> (4 * k)) & 15ull);
                    const int np = n >> l;
                    const int i = (((t0 - G::g(l)) << (NL - l)) + e) & (np - 1);
                    const T *node = ((l == NL && ((P.deep >> j) & 1u)) ? xd : xs) + (int64_t)j * np;
                    R.v[u] = *reinterpret_cast<const V2 *>(node + i);
                    at[u] = (j * G::S(l) + e) | (l << 24);
                };
                if (NL >= 4 && f >= P.pf[4]) fetch(std::integral_constant<int, (NL >= 4 ? 4 : NL)>{});
                else if (NL >= 3 && f >= P.pf[3]) fetch(std::integral_constant<int, (NL >= 3 ? 3 : NL)>{});
                else if (NL >= 2 && f >= P.pf[2]) fetch(std::integral_constant<int, (NL >= 2 ? 2 : NL)>{});
                else fetch(std::integral_constant<int, 1>{});
            }
        }
    };
    auto enter = [&](int l) {
        const int bo = (l & 1) ? G::boff : 0;
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (at[u] >= 0 && (at[u] >> 24) == l) {
                const int a = at[u] & 0xffffff;
                lds[bo + a + (a >> 2)] = R.v[u].x;
                lds[bo + a + 1 + ((a + 1) >> 2)] = R.v[u].y;
            }
    };
    unsigned tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile);
    while (tile < ntiles) {
        const unsigned sig = tile / tiles;
        const int t0 = (int)(tile - sig * tiles) * G::TL;
        const unsigned next = tile + gridDim.x;
        __syncthreads();                                    // the previous tile's store has read buffer 0
        enter(NL);
        auto level = [&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if (l - 1 >= 1 && P.nf[l - 1]) enter(l - 1);    // the other buffer: no hazard with this level's reads
            if (l == 1 && next < ntiles) prefetch(next);    // every register of this tile is in LDS: fetch the next tile's
            __syncthreads();
            constexpr int co = (l & 1) ? G::boff : 0, po = ((l - 1) & 1) ? G::boff : 0;
            constexpr int Sc = G::S(l), Sp = G::S(l - 1);
            constexpr int offi = (G::g(l) - G::g(l - 1)) << (NL - l);
            constexpr int pairs = G::W(l - 1) / 2;
            constexpr int groups = (pairs + PP - 1) / PP;
            constexpr int ab = offi - (HF - 1), db = offi;
            const int total = P.ns[l] * groups;
            const unsigned spl = P.sp[l];
            for (int idx = threadIdx.x; idx < total; idx += WX_TT_NT) {
                const int k = idx / groups, gi = idx - k * groups;
                const int j = (int)((spl >> (4 * k)) & 15u);
                // children slots 2 j, 2 j + 1 (multiples of 8 words): ph(slot + x + 4 gi) = slot + slot / 4 + 5 gi + x + (x >> 2)
                const T *aw = lds + co + 2 * j * (Sc + Sc / 4) + 5 * gi, *dw = aw + (Sc + Sc / 4);
                AT wa[WIN], wd[WIN];
#pragma unroll
                for (int e = 0; e < WIN; ++e) {
                    wa[e] = (AT)aw[ab + e + ((ab + e) >> 2)];
                    wd[e] = (AT)dw[db + e + ((db + e) >> 2)];
                }
                T *pw = lds + po + j * (Sp + Sp / 4) + 10 * gi;         // parent words 8 gi ..: the skew adds 2 gi
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    // a[k - m] = wa[p + HF - 1 - m], d[k + m] = wd[p + m]
                    AT v0 = 0, v1 = 0;
#pragma unroll
                    for (int m = 0; m < HF; ++m) {
                        const AT av = wa[p + HF - 1 - m], dv = wd[p + m];
                        v0 = fma(qq[2 * m], av, v0);
                        v0 = fma(-qq[2 * m + 1], dv, v0);
                        v1 = fma(qq[2 * m + 1], av, v1);
                        v1 = fma(qq[2 * m], dv, v1);
                    }
                    pw[2 * p + ((2 * p) >> 2)] = (T)v0;      // the last group of a node runs into the slot's padding
                    pw[2 * p + 1 + ((2 * p + 1) >> 2)] = (T)v1;
                }
            }
            __syncthreads();
        };
        if constexpr (NL >= 4) level(std::integral_constant<int, 4>{});
        if constexpr (NL >= 3) level(std::integral_constant<int, 3>{});
        if constexpr (NL >= 2) level(std::integral_constant<int, 2>{});
        level(std::integral_constant<int, 1>{});
        T *yo = dst + (int64_t)sig * dstride + ((int64_t)t0 << NL);
        constexpr int tot2 = TS / 2;
        for (int f = threadIdx.x; f < tot2; f += WX_TT_NT) {
            V2 v;
            v.x = lds[2 * f + ((2 * f) >> 2)];
            v.y = lds[2 * f + 1 + ((2 * f + 1) >> 2)];
            *reinterpret_cast<V2 *>(yo + 2 * f) = v;
        }
        tile = next;
    }
}

// host side --------------------------------------------------------------------------------------------------------------------
// tile size (samples) of a pass: the largest whose two LDS buffers leave room for two workgroups per CU
template <typename T, int F, int NL, bool INVERSE> constexpr int wx_tt_ts()
{
    if (WxTTGeo<F, NL, 4096, INVERSE>::lds_bytes(sizeof(T)) <= 80 * 1024) return 4096;
    if (WxTTGeo<F, NL, 2048, INVERSE>::lds_bytes(sizeof(T)) <= 80 * 1024) return 2048;
    return 1024;
}
// the run-time part of a pass out of the masks (W2[l] = element pairs of a depth-l window); false: not a tree of NL levels
bool wx_top_tree(int NL, unsigned split, unsigned deep, const int *W2, WxTopTree *P, bool wpd = false);
int64_t wx_top_grid(int64_t ntiles, size_t lds);
// one pass over `batch` signals; see the header comment for the layouts
template <typename T>
int wx_dev_top_levels(bool inverse, const T *src, T *dst, T *deep, int64_t n, int NL, int64_t batch, int64_t sstride, int64_t dstride,
                      int64_t deepstride, unsigned split, unsigned deepmask, const WxFilt &filt, hipStream_t st);
bool wx_top_levels_ok(int F);
// wpd of the top NL levels in one pass: x (signals sstride apart) -> slices 0 .. NL of the packet tables at dst (tables dstride apart, slices
// lstride apart inside a table); full tree
template <typename T>
int wx_dev_top_levels_wpd(const T *src, T *dst, int64_t n, int NL, int64_t batch, int64_t sstride, int64_t dstride, int64_t lstride,
                          const WxFilt &filt, hipStream_t st);
