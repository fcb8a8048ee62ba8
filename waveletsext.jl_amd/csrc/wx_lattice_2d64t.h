#pragma once
// wx_lattice_2d64t.h -- 2-D wpt / iwpt of 64 x 64 images along ANY quad tree (pyramids = dwt / idwt of images, best bases, ...) in ONE pass:
// the masked form of wx_lattice_2d64.h, built from the pieces of wx_lattice_tree_sc.h.  Reference: 2-D wpt / iwpt with a quad tree
// (DWT.jl:500-548, 662-710: node i is decomposed iff tree[i], children 4 i - 2 + 2 (row half) + (column half), Utils.jl:31-60), each step the
// 2-D dwt_step! / idwt_step! of the node (dwt/dwt_one_level.jl:319-354, 401-436): down the columns, then along the rows.
//
// A quad tree is not separable (level l + 1 of a node needs BOTH passes of level l), so the levels alternate between the two register
// layouts of wx_lattice_2d64.h: level l = the column level on register bit l - 1 of layout A (a column per lane), the exchanges T2, T3, the row
// level on register bit l - 1 of layout C (a row per lane), and T3^-1, T2^-1 back if a deeper level follows.  Both are lane-local masked
// levels (lat_level_cm): the node a butterfly belongs to is (row path, column path) = (low register bits, low lane bits) in layout A and
// (low lane bits, low register bits) in layout C, so one 64-bit lane mask per (level, register class) says where the node is split
// (k_lat2d64t_prep); every level normalises its own gains under its mask, so a coefficient keeps its final value whatever its depth.
// The output position of slot (row, column) with leaf depth d is (row path reversed on top of row >> d, the same for the column): a table of
// 4096 LDS addresses; the wavefront writes its registers to a column-major image of the result in LDS (two halves of 16 KiB: the first
// column branch is register bit 0 of layout C) and streams it out with 16-byte-per-lane stores.  The inverse mirrors it.
// Cost next to the full-tree kernel: 4 exchanges per level instead of 2 per transform -- the kernel is LDS-issue bound, not FP64 bound.
#include "wx_lattice_dev.h"
#include "wx_host.h"
#include "wx_lattice_tree_sc.h"

struct WxLat2dTree {
    unsigned short perm[64 * 64];                 // [r >> 3][lane][r & 7], r = register of layout C: address in the 16 KiB half image
    alignas(64) unsigned long long mA[6 * 32];    // [32 (l - 1) + s]: column level l, register class s of layout A: lanes whose node is split
    alignas(64) unsigned long long mC[6 * 32];    // the same for the row level l in layout C
    int depth;                                    // levels that have a split node
};

namespace {

// one workgroup of 256 threads: the tree as "exists and is split" flags in LDS (quad heap, node i at eff[i]), then the tables
__global__ __launch_bounds__(256) void k_lat2d64t_prep(const uint8_t *__restrict__ status, int64_t nstatus, int L, WxLat2dTree *__restrict__ tab)
{
    __shared__ uint8_t eff[1366];                 // nodes of depth 0 .. 5: (4^6 - 1) / 3 = 1365, 1-based
    __shared__ int deep;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) deep = 0;
    for (int i = 1 + tid; i <= 1365; i += 256) {
        int d = 0;
        for (int64_t first = 1, cnt = 1; i >= first + cnt; first += cnt, cnt *= 4) ++d;
        eff[i] = (d < L && i - 1 < nstatus && status[i - 1] != 0) ? 1 : 0;
    }
    __syncthreads();
    // a node exists only if its parent is split: level by level (first index of depth d: (4^d + 2) / 3)
    for (int d = 1; d < 6; ++d) {
        const int first = ((1 << (2 * d)) + 2) / 3, cnt = 1 << (2 * d);
        for (int i = first + tid; i < first + cnt; i += 256) eff[i] = eff[i] & eff[(i + 2) / 4];
        __syncthreads();
    }
    // the split node of depth d with row path rp and column path cp (bit t = branch of level t + 1), or 0
    auto node_of = [&](int d, int rp, int cp) {
        int i = 1;
        for (int t = 0; t < d; ++t) {
            if (!eff[i]) return 0;
            i = 4 * i - 2 + 2 * ((rp >> t) & 1) + ((cp >> t) & 1);
        }
        return i;
    };
    auto sp = [&](int d, int rp, int cp) {
        if (d >= 6) return false;
        const int i = node_of(d, rp, cp);
        return i > 0 && eff[i] != 0;
    };
    for (int e = tid; e < 4096; e += 256) {
        const int ln = e & 63, r = e >> 6;            // layout C: lane = row, register = column
        int d = 0;
        while (sp(d, ln, r)) ++d;
        int orow = ln >> d, ocol = r >> d;
        for (int t = 0; t < d; ++t) { orow |= ((ln >> t) & 1) << (5 - t); ocol |= ((r >> t) & 1) << (5 - t); }
        const unsigned o = (unsigned)orow + 64u * (unsigned)ocol;
        tab->perm[((r >> 3) * 64 + ln) * 8 + (r & 7)] = (unsigned short)(lat_sc_addr(o) & 0x3fffu);
        if (d > 0) atomicMax(&deep, d);
    }
    // masks: wavefront q mod 4 makes mask q; level l = q / 32 + 1, class s = q % 32 (s < 2^(l-1))
    for (int q = wave; q < 2 * 192; q += 4) {
        const bool isC = q >= 192;
        const int qq = isC ? q - 192 : q, l1 = qq >> 5, s = qq & 31;           // l1 = l - 1 = depth of the node
        const int lp = lane & ((1 << l1) - 1);
        const bool bit = s < (1 << l1) && (isC ? sp(l1, lp, s) : sp(l1, s, lp));
        const unsigned long long m = __ballot(bit);
        if (lane == 0) (isC ? tab->mC : tab->mA)[qq] = m;
    }
    __syncthreads();
    if (tid == 0) tab->depth = deep;
}

template <int NS, int WPE, typename IO, bool INV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat2d64t_f64(
    const IO *__restrict__ x, IO *__restrict__ y, int last_img, unsigned in_img, WxLatW cw, const WxLat2dTree *__restrict__ tab)
{
    __shared__ __attribute__((aligned(16))) double lds[2048];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    typedef typename std::conditional<std::is_same<IO, float>::value, lat_f2v, double>::type V;
    typedef typename lat_row<V>::type ROW;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    const bool lastw = PAIR && blockIdx.x == gridDim.x - 1;
    const int img0 = PAIR ? (lastw ? last_img : (int)(blockIdx.x << 1)) : (int)blockIdx.x;
    const unsigned bimg = (unsigned)(lastw ? cw.tail_bsig : 1);
    const unsigned bofs = PAIR ? bimg * 4096u : 0xffffffffu, bofs_in = PAIR ? bimg * in_img : 0xffffffffu;
    const IO *xs = x + (int64_t)img0 * in_img;
    IO *ys = y + (int64_t)img0 * 4096;
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
    const int D = __builtin_amdgcn_readfirstlane(tab->depth);
    unsigned pw[32];
    auto load_pw = [&]() {
        const uint4 *pp = reinterpret_cast<const uint4 *>(tab->perm) + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 t = pp[64 * q];
            pw[4 * q] = t.x; pw[4 * q + 1] = t.y; pw[4 * q + 2] = t.z; pw[4 * q + 3] = t.w;
        }
    };
    if constexpr (!INV) {
        // forward: gl[1] = g, g2 = g^-2: after the shears the a-slot holds a / g, the d-slot d g
        const double g = cw.gl[1], ginv = cw.c.g2 * cw.gl[1];
        V c[64];
        {
            V a[64];
            lat_absorb<0, 0>(a, lds0, xs, lane, cw, 4096u, 0, 0, 0, bofs_in);
#define WX_Q_FWD(LV)                                                                           \
            if (LV == 1 || D >= LV) {                               /* the root is split (the caller checks): level 1 always runs */ \
                if constexpr (LV > 1) { V bb[64]; lat_t3i(c, bb, lds0, lane); lat_t2i(bb, a, lds0, lane); } \
                lat_level_cm<LV - 1, NS, false>(a, cf, tab->mA + 32 * (LV - 1), g, ginv);       \
                { V bb[64]; lat_t2(a, bb, lds0, lane); lat_t3(bb, c, lds0, lane); }             \
                lat_level_cm<LV - 1, NS, false>(c, cf, tab->mC + 32 * (LV - 1), g, ginv);       \
            }
            WX_Q_FWD(1) WX_Q_FWD(2) WX_Q_FWD(3) WX_Q_FWD(4) WX_Q_FWD(5) WX_Q_FWD(6)
#undef WX_Q_FWD
        }
        load_pw();
        lat_sync();
        lat_for<2>([&](auto Hc) {
            constexpr int h = Hc;
            lat_for<32>([&](auto Rc) {
                constexpr int r = 2 * Rc + h;                       // the first column branch = register bit 0: half h of the image
                const unsigned p = (r & 1) ? (pw[r >> 1] >> 16) : (pw[r >> 1] & 0xffffu);
                lds_wr<0>(lds0 + p, c[r]);
            });
            lat_sync();
            lat_for<2>([&](auto Gc) {
                constexpr int k0 = 8 * Gc;
                ROW v[8];
                lat_for<8>([&](auto Kc) { lat_sc_ldrow(v[Kc], lat_sc_row<h, k0 + Kc>(lds0, lane)); });
                lat_for<8>([&](auto Kc) {
                    constexpr int q = 16 * h + k0 + Kc;
                    lat_sc_gst(ys + 128 * q, 2u * lane, bofs == 0xffffffffu ? 0u : bofs, v[Kc]);
                });
            });
            lat_sync();
        });
    } else {
        // synthesis: gl[1] = 1 / g, g2 = g^2 -- the a-slot of a split node enters as a / g, the d-slot as d g
        const double ga = cw.gl[1], gd = cw.c.g2 * cw.gl[1];
        load_pw();
        V c[64];
        lat_for<2>([&](auto Hc) {
            constexpr int h = Hc;
            lat_for<2>([&](auto Gc) {
                constexpr int k0 = 8 * Gc;
                ROW v[8];
                lat_for<8>([&](auto Kc) {
                    constexpr int q = 16 * h + k0 + Kc;
                    lat_sc_gld(v[Kc], xs + 128 * q, 2u * lane, bofs_in == 0xffffffffu ? 0u : bofs_in);
                });
                lat_for<8>([&](auto Kc) { lat_sc_strow(lat_sc_row<h, k0 + Kc>(lds0, lane), v[Kc]); });
            });
            lat_sync();
            lat_for<32>([&](auto Rc) {
                constexpr int r = 2 * Rc + h;
                const unsigned p = (r & 1) ? (pw[r >> 1] >> 16) : (pw[r >> 1] & 0xffffu);
                lat_sc_rd(c[r], lds0 + p);
            });
            // the reads have landed before the image half is overwritten, and before the registers are used
            lat_for<4>([&](auto Wc) {
                constexpr int b = 16 * Wc + h;
                lat_wait8(c[b], c[b + 2], c[b + 4], c[b + 6], c[b + 8], c[b + 10], c[b + 12], c[b + 14]);
            });
            lat_sync();
        });
        V a[64];
#define WX_Q_INV(LV)                                                                           \
        if (LV == 1 || D >= LV) {                                                               \
            lat_level_cm<LV - 1, NS, true>(c, cf, tab->mC + 32 * (LV - 1), ga, gd);             \
            { V bb[64]; lat_t3i(c, bb, lds0, lane); lat_t2i(bb, a, lds0, lane); }               \
            lat_level_cm<LV - 1, NS, true>(a, cf, tab->mA + 32 * (LV - 1), ga, gd);             \
            if constexpr (LV > 1) { V bb[64]; lat_t2(a, bb, lds0, lane); lat_t3(bb, c, lds0, lane); } \
        }
        WX_Q_INV(6) WX_Q_INV(5) WX_Q_INV(4) WX_Q_INV(3) WX_Q_INV(2) WX_Q_INV(1)
#undef WX_Q_INV
        lat_emit<0, 0>(a, lds0, ys, lane, cw, 4096u, 0, 0, bofs);
    }
}

}  // namespace

// 0 = not applicable (the caller goes on to the tile kernels), 1 = launched, < 0 = error
template <typename IO, int NSMAX, bool INV>
static int wx_lattice_2d64t_launch(const IO *x, IO *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, const uint8_t *dstatus,
                                   int64_t nstatus, hipStream_t st)
{
    constexpr bool PAIR = std::is_same<IO, float>::value;
    if (L < 1 || L > 6 || filt.F < 2 || (filt.F & 1) || filt.F > 2 * NSMAX || batch < 1 || batch > 0x3fffffff || !dstatus) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 0;
    if (in_img < 4096 || (in_img & 3) || in_img > 0x3fffffff) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, L, INV, &cw.c)) return 0;
    {
        WxLat one;
        if (!wx_lattice_factor(filt, 1, false, &one)) return 0;
        const long double g = one.g0;                           // product of the cosines of one level
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= INV ? 1 / g : g; }
    }
    WxPairPlan pp;
    if (PAIR) {
        if (!wx_lat_pair_plan(batch, 0, x == y, &pp)) return 0;
    } else {
        pp.nwave = (unsigned)batch; pp.tail_sig = (int)(batch - 1); pp.tail_bsig = 0;
    }
    cw.tail_bsig = pp.tail_bsig;
    WxScratch scr(st);
    WxLat2dTree *tab = (WxLat2dTree *)scr.alloc(sizeof(WxLat2dTree));
    if (!tab) return WX_EHIP;
    hipLaunchKernelGGL(k_lat2d64t_prep, dim3(1), dim3(256), 0, st, dstatus, nstatus, L, tab);
    const WxLat2dTree *ctab = tab;
#define WX_GOQ(NSS)                                                                                                                  \
    case NSS:                                                                                                                        \
        hipLaunchKernelGGL((k_lat2d64t_f64<NSS, 2, IO, INV>), dim3(pp.nwave), dim3(64), 0, st, x, y, pp.tail_sig, (unsigned)in_img, cw, ctab); \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GOQ(1) WX_GOQ(2) WX_GOQ(4)
    default:
        if constexpr (NSMAX > 4) {
            switch (wx_lat_stages(filt.F)) {
                WX_GOQ(6) WX_GOQ(8)
            default: return 0;
            }
        } else
            return 0;
    }
#undef WX_GOQ
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice 2-D quad-tree launch (64 x 64 images)", __FILE__, __LINE__);
    return 1;
}
