// wx_lattice_tree_sc.h -- tree-driven wpt / iwpt / iwpd for 4096-sample Float64 signals, second form: EVERY level runs under
// the tree's masks and the packet order is produced by ONE permutation through LDS.
//
// Reference: Wavelets.jl's wpt / iwpt with a tree::BitVector as called by wptall / iwptall (dwt/dwt_all.jl:152-166, 210-225),
// iwpd by tree (DWT.jl:340-351: getbasiscoef, Utils.jl:101-134, then iwpt), denoise(:wpt) (Denoising.jl:527).
//
// The first form (k_lat_wpt_tree_f64 in wx_lattice_dev.h) lets the leaves of depth l < 6 leave through their own exchange
// right after level l: up to five dependent exchange chains per signal, which is what holds random trees at 0.42 / 0.36 of
// the HBM peak.  Here a leaf simply stops changing: the in-place lattice keeps coefficient (path p of d levels, position k
// in the node) at index i = k << d | p (bit t of p = branch of level t + 1), so
//   * levels on bits 0, 1 (layout A: reg = i[5:0]): the node of a butterfly is a register-index pattern -> wave-uniform
//     conditions (the root is always split);
//   * levels on bits 2 .. 5 (layout B: lane = i[1:0] << 4 | i[11:8], reg = i[7:2]): node = (lane >> 4, low register bits) -> one
//     64-bit lane mask per (level, register class), whole 16-lane rows on or off, so the halo rotations inside a row stay valid;
//   * levels on bits 6 .. 11 (layout C: lane = i[5:0]): lat_level_cm as before; for sparse trees the levels on bits 6, 7 run in
//     layout B instead, where 4 nodes share a register class instead of 64 and most classes are skipped outright;
//   masks are scalar loads handed to the hardware as exec masks (__builtin_amdgcn_inverse_ballot_w64); every level
//   normalises its own gains under the same mask, so a coefficient carries its final value whatever its depth;
//   * the output position of coefficient i with leaf depth d is  o = bitreverse_d(i[d-1:0]) << (12 - d) | i >> d: a table of
//     4096 16-bit LDS addresses per tree (k_lat_treesc_prep), 64 per lane.  The wavefront writes its registers to a
//     packet-order image of the signal in LDS (two halves of 16 KiB, ds_write_b64 under "this half" masks, XOR-swizzled so that
//     the 64 rows a register's lanes usually hit fall into different banks) and streams it out with 16-byte-per-lane stores of
//     whole KiB: no store predicate, no per-depth exchange, no in-register unshuffle.
// The inverse mirrors it: KiB loads -> LDS image -> 64 table-addressed ds_read_b64 per lane -> masked synthesis levels C, B, A.
// For iwpd by tree a 16-byte piece is read from the column of its leaf's depth (4-bit table per piece).
// SH = 1, 2 (2048- and 1024-sample signals): 2^SH signals share a wavefront, index bits [SH-1:0] are the signal number, the
// levels act on bits SH .. 11 (the first of them is the root's: unmasked), and the image in LDS is the signals one after the other.
// SH = 3 .. 6 (512 .. 64 samples, round 5; launcher wx_lattice_tree_s.h): the same -- the first layout is B (SH = 6: C, where the root's
// level is a masked level whose mask is every lane), the half of the image is the signal number's top bit = lane bit SH - 1 of layout C, a
// 128-element piece of the image is one 128-sample signal or two 64-sample signals.
#pragma once

struct WxLatTreeSc {
    unsigned short perm[64 * 64];      // [r >> 3][lane][r & 7], r = register of layout C: swizzled byte address in the 32 KiB packet-order image
    alignas(16) unsigned long long mA[2];        // level on bit 1: node (1, s) is split -> all lanes
    alignas(64) unsigned long long mB[6 * 32];   // [32 K + s]: level on bit 2 + K, register class s: lanes whose node is split
    alignas(64) unsigned long long mC[6 * 32];   // [32 K + s]: level on bit 6 + K (rows of 32 so that 8 masks are one aligned scalar load)
    unsigned dep[4 * 64];              // [w][lane]: leaf depth of piece (h, k) -- elements 2048 h + 128 k + 2 lane, +1 -- nibble 16 h + k
    unsigned anyA, anyB[6], anyC[6];
    unsigned deepB;                    // the levels on bits 6, 7 run in layout B (sparse trees: most register classes are idle there)
};

namespace {

// swizzled byte address of packet-order element o: rows of 64 elements (512 bytes); the 16-byte chunk index of a row is
// XORed with a key of the row
__host__ __device__ constexpr unsigned lat_sc_addr(unsigned o)
{
    const unsigned row = o >> 6, ch = (o >> 1) & 31u, key = (row & 31u) ^ (row >> 5);
    return row * 512u + ((ch ^ key) << 4) + (o & 1u) * 8u;
}

// 8 workgroups of 256 threads, each with its own LDS copy of the tree and an eighth of the tables: the tree goes to LDS as "exists and is split" flags (eff[heap index], made top-down level by level), after
// which every walk is a handful of LDS reads -- the first version walked the tree in global memory from one wavefront and took
// 0.11 ms per call, a tenth of the transform it prepares.
template <int SH>
__global__ __launch_bounds__(256) void k_lat_treesc_prep(const uint8_t *__restrict__ status, int64_t nstatus, int L,
                                                           WxLatTreeSc *__restrict__ tab)
{
    constexpr int SB = 12 - SH;                                // index bits of one signal
    __shared__ uint8_t eff[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid + 1; idx < (1 << SB); idx += 256) {
        const int d = 31 - __clz(idx);
        eff[idx] = (d < L && idx - 1 < nstatus && status[idx - 1] != 0) ? 1 : 0;
    }
    if (tid == 0) eff[0] = 0;
    __syncthreads();
    for (int d = 1; d < SB; ++d) {
        for (int idx = (1 << d) + tid; idx < (2 << d); idx += 256) eff[idx] = eff[idx] & eff[idx >> 1];
        __syncthreads();
    }
    // node (d, j) exists and is split
    auto sp = [&](int d, int j) { return d < SB && eff[(1 << d) + j] != 0; };
    constexpr int NB = 8;                                      // workgroups (the launch's grid)
    const int blk = blockIdx.x;
    for (int e = 512 * blk + tid; e < 512 * (blk + 1); e += 256) {
        const int ln = e & 63, r = e >> 6;
        const int i = ln | (r << 6), sig = i & ((1 << SH) - 1), p = i >> SH;
        int d = 0, j = 0;
        while (sp(d, j)) { j = (j << 1) | ((p >> d) & 1); ++d; }
        const unsigned o = ((unsigned)sig << SB) | ((unsigned)j << (SB - d)) | ((unsigned)p >> d);
        // the half of the image (o bit 11) is the first branch (SH = 0: the root is split) or the signal number's top bit:
        // index bit 0 (SH < 2) or 1 (SH = 2), a lane bit -- only the address inside the half is stored
        tab->perm[((r >> 3) * 64 + ln) * 8 + (r & 7)] = (unsigned short)(lat_sc_addr(o & 4095u) & 0x3fffu);
    }
    // lanes whose node -- the one the level on index bit b would split -- is split; the path of that node is index bits SH .. b - 1.
    // Mask q of the 2 + 192 + 192 is made by wavefront q mod 4.
    auto level_bit = [&](int b, auto bit_of) {
        const int d = b - SH;
        if (d < 0) return false;
        int j = 0;
        for (int t = 0; t < d; ++t) j |= bit_of(SH + t) << (d - 1 - t);
        return sp(d, j);
    };
    for (int q = 4 * blk + wave; q < 2 + 192 + 192; q += 4 * NB) {
        bool bit;
        if (q < 2) {
            const int s = q;
            bit = level_bit(1, [&](int) { return s; });                                      // layout A: index bit 0 = register bit 0
        } else if (q < 2 + 192) {
            const int K = (q - 2) >> 5, s = (q - 2) & 31;
            // layout B: index bits 0, 1 = lane bits 4, 5; bits 2 .. = register bits 0 ..
            bit = s < (1 << K) && level_bit(2 + K, [&](int sb) { return sb < 2 ? (lane >> (4 + sb)) & 1 : (s >> (sb - 2)) & 1; });
        } else {
            const int K = (q - 194) >> 5, s = (q - 194) & 31;
            // layout C: index bits 0 .. 5 = lane bits, bits 6 .. = register bits
            bit = s < (1 << K) && level_bit(6 + K, [&](int sb) { return sb < 6 ? (lane >> sb) & 1 : (s >> (sb - 6)) & 1; });
        }
        const unsigned long long m = __ballot(bit);
        if (lane == 0) {
            if (q < 2) tab->mA[q] = m;
            else if (q < 194) tab->mB[q - 2] = m;
            else tab->mC[q - 194] = m;
        }
    }
    for (int e = 256 * blk + tid; e < 256 * (blk + 1); e += 256) {
        const int ln = e & 63, q = e >> 6;                         // piece q = 16 h + k of lane ln: elements 128 q + 2 ln, +1
        const int pos = (128 * q + 2 * ln) & ((1 << SB) - 1);
        int d = 0;
        while (sp(d, pos >> (SB - d))) ++d;
        atomicOr(&tab->dep[64 * (q >> 3) + ln], (unsigned)d << (4 * (q & 7)));
    }
}
// second step: the "any" flags and the choice of layout for the levels on bits 6, 7 out of the masks (one wavefront)
__global__ __launch_bounds__(64) void k_lat_treesc_prep2(WxLatTreeSc *__restrict__ tab, int sh = 0)
{
    const int lane = threadIdx.x;
    if (lane == 0) {
        tab->anyA = (tab->mA[0] | tab->mA[1]) != 0;
        int nactB = 0;
        for (int K = 0; K < 6; ++K) {
            unsigned long long aB = 0, aC = 0;
            for (int s = 0; s < (1 << K); ++s) {
                aB |= tab->mB[32 * K + s];
                aC |= tab->mC[32 * K + s];
                if (K >= 4 && tab->mB[32 * K + s]) ++nactB;
            }
            tab->anyB[K] = aB != 0;
            tab->anyC[K] = aC != 0;
        }
        // bits 6, 7: in layout C a lane is a node of depth 6, so a register class is busy as soon as one of 64 nodes is split
        // there; in layout B only 4 nodes share a class (at the price of halo moves): B when at most half of its 48 classes are busy
        tab->deepB = nactB <= 24 && sh < 6;                    // 64-sample signals never see layout B
    }
}

// one packet level on register-index bit K under the lane masks mk[s] (one per register class s = low K register bits),
// H cyclic lane bits hold the rest of a sequence (lat_nbr).  The shears and the gains run under the mask; the renamings and
// halo rotations between the shears are executed by every lane: over a level they compose to the identity, and a sequence's
// lanes (a 16-lane row in layout B, the wavefront in layout A) are all in or all out.
template <int K, int H, int NS, bool INV, typename V, typename C>
__device__ __forceinline__ void lat_level_hm(V (&x)[64], const C &cf, const unsigned long long *__restrict__ mk, double ga_, double gd_)
{
    typedef typename lat_vtraits<V>::coef CF;
    const CF ga = lat_sgpr((CF)ga_), gd = lat_sgpr((CF)gd_);
    constexpr int NSEQ = 1 << K, M = 32 >> K, S = 1 << K, G = NSEQ < 8 ? NSEQ : 8;
    auto U = [](int s, int m) { return s + ((2 * m) << K); };
    lat_for<NSEQ / G>([&](auto Gc) {
    constexpr int s0 = G * Gc;
    unsigned long long mm[G];
    lat_masks<G>(mm, mk + s0);                              // one scalar load for the group (see lat_level_cm)
    lat_for<G>([&](auto Sc) {
        constexpr int s = s0 + Sc;
        const unsigned long long msk = mm[Sc];
        if (!msk) return;
        auto shift = [&](auto SHc) {
            constexpr int SHv = decltype(SHc)::value;
            if constexpr (SHv != 0) {
                V old[M];
#pragma unroll
                for (int m = 0; m < M; ++m) old[m] = x[U(s, m) + S];
                lat_for<M>([&](auto Mc) {
                    constexpr int m = Mc;
                    constexpr int g = m + SHv;
                    constexpr int d = (g >= 0) ? g / M : -((-g + M - 1) / M);
                    constexpr int src = g - d * M;
                    x[U(s, m) + S] = lat_nbr<H, d>(old[src]);
                });
            }
        };
        auto scale = [&]() {
            if (__builtin_amdgcn_inverse_ballot_w64(msk)) {
                asm volatile("");                           // a real exec region (see lat_level_cm)
#pragma unroll
                for (int m = 0; m < M; ++m) { x[U(s, m)] = lat_mul(x[U(s, m)], ga); x[U(s, m) + S] = lat_mul(x[U(s, m) + S], gd); }
            }
        };
        constexpr bool one_shot = (H != 6) || (NS - 1 <= M);
        if constexpr (!INV) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const CF pj = (CF)cf.p[j], kj = (CF)cf.kap[j];
                if (__builtin_amdgcn_inverse_ballot_w64(msk)) {
                    asm volatile("");
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        x[U(s, m)] = lat_fma(pj, x[U(s, m) + S], x[U(s, m)]);
                        x[U(s, m) + S] = lat_fma(-kj, x[U(s, m)], x[U(s, m) + S]);
                    }
                }
                if (j + 1 < NS) shift(std::integral_constant<int, 1>{});
            }
            if constexpr (one_shot) shift(std::integral_constant<int, -(NS - 1)>{});
            else {
#pragma unroll
                for (int j = 0; j + 1 < NS; ++j) shift(std::integral_constant<int, -1>{});
            }
            scale();
        } else {
            scale();
            if constexpr (one_shot) shift(std::integral_constant<int, NS - 1>{});
            else {
#pragma unroll
                for (int j = 0; j + 1 < NS; ++j) shift(std::integral_constant<int, 1>{});
            }
#pragma unroll
            for (int j = NS - 1; j >= 0; --j) {
                const CF pj = (CF)cf.p[j], kj = (CF)cf.kap[j];
                if (__builtin_amdgcn_inverse_ballot_w64(msk)) {
                    asm volatile("");
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        x[U(s, m) + S] = lat_fma(kj, x[U(s, m)], x[U(s, m) + S]);
                        x[U(s, m)] = lat_fma(-pj, x[U(s, m) + S], x[U(s, m)]);
                    }
                }
                if (j > 0) shift(std::integral_constant<int, -1>{});
            }
        }
    });
    });
}

// table-addressed reads of the inverse: into a fresh register / into a register whose other lanes keep their value
template <typename V> __device__ __forceinline__ void lat_sc_rd(V &dst, unsigned ad)
{
    asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(ad) : "memory");
}
template <typename V> __device__ __forceinline__ void lat_sc_rd_keep(V &dst, unsigned ad)
{
    asm volatile("ds_read_b64 %0, %1" : "+v"(dst) : "v"(ad) : "memory");
}
typedef lat_d2 __attribute__((address_space(3))) *lat_l2p;
__device__ __forceinline__ lat_l2p lat_sc_lp(unsigned a) { return (lat_l2p)(uintptr_t)a; }
// a 16-byte row of the LDS image = two consecutive slots <-> global memory.  V = lat_f2v: the row holds (A[p], B[p], A[p+1], B[p+1]),
// signal A's two samples go to p, signal B's to p + boff
typedef float lat_f4v __attribute__((ext_vector_type(4)));
typedef lat_f4v __attribute__((address_space(3))) *lat_l4p;
template <typename V> struct lat_row;
template <> struct lat_row<double> { typedef lat_d2 type; };
template <> struct lat_row<lat_f2v> { typedef lat_f4v type; };
__device__ __forceinline__ void lat_sc_ldrow(lat_d2 &r, unsigned a) { r = *lat_sc_lp(a); }
__device__ __forceinline__ void lat_sc_ldrow(lat_f4v &r, unsigned a) { r = *(lat_l4p)(uintptr_t)a; }
__device__ __forceinline__ void lat_sc_strow(unsigned a, lat_d2 r) { *lat_sc_lp(a) = r; }
__device__ __forceinline__ void lat_sc_strow(unsigned a, lat_f4v r) { *(lat_l4p)(uintptr_t)a = r; }
template <typename IO> __device__ __forceinline__ void lat_sc_gst(IO *base, unsigned off, unsigned, lat_d2 r) { lat_st2(lat_sbase(base) + off, r); }
__device__ __forceinline__ void lat_sc_gst(float *base, unsigned off, unsigned boff, lat_f4v r)
{
    typedef lat_f2 __attribute__((address_space(1))) *P;
    lat_f2 a, b;
    a.x = r.x; a.y = r.z;
    b.x = r.y; b.y = r.w;
    *(P)(lat_sbase(base) + off) = a;
    *(P)(lat_sbase(base + boff) + off) = b;
}
template <typename GP> __device__ __forceinline__ void lat_sc_gldp(lat_d2 &r, GP pa, GP) { r = lat_ld2(pa); }
__device__ __forceinline__ void lat_sc_gldp(lat_f4v &r, const float __attribute__((address_space(1))) *pa, const float __attribute__((address_space(1))) *pb)
{
    typedef const lat_f2 __attribute__((address_space(1))) *P;
    const lat_f2 a = *(P)pa, b = *(P)pb;
    r.x = a.x; r.y = b.x; r.z = a.y; r.w = b.y;
}
template <typename IO> __device__ __forceinline__ void lat_sc_gld(lat_d2 &r, const IO *base, unsigned off, unsigned) { r = lat_ld2(lat_sbase(base) + off); }
__device__ __forceinline__ void lat_sc_gld(lat_f4v &r, const float *base, unsigned off, unsigned boff)
{
    typedef const lat_f2 __attribute__((address_space(1))) *P;
    const lat_f2 a = *(P)(lat_sbase(base) + off), b = *(P)(lat_sbase(base + boff) + off);
    r.x = a.x; r.y = b.x; r.z = a.y; r.w = b.y;
}

// byte address of the 16 bytes lane `lane` moves in instruction k of half h (elements 2048 h + 128 k + 2 lane, +1)
template <int H, int Kk> __device__ __forceinline__ unsigned lat_sc_row(unsigned lds0, int lane)
{
    const unsigned hi = (unsigned)lane >> 5, ch = ((unsigned)lane & 31u) ^ hi ^ (unsigned)H;
    return lds0 + 512u * hi + 1024u * Kk + ((ch ^ (2u * Kk)) << 4);
}

__device__ __forceinline__ void lat_sc_ptab(unsigned (&pw)[32], const WxLatTreeSc *__restrict__ tab, int lane)
{
    const uint4 *pp = reinterpret_cast<const uint4 *>(tab->perm) + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint4 t = pp[64 * q];
        pw[4 * q] = t.x; pw[4 * q + 1] = t.y; pw[4 * q + 2] = t.z; pw[4 * q + 3] = t.w;
    }
}

// the level on index bit BIT, register bit KK of a layout with HH cyclic lane bits: the root's (BIT = SH) is unmasked
#define WX_SC_FWD(KK, HH, REG, BIT, MK, ANY)                                                   \
    if constexpr (BIT == SH) {                                                                  \
        lat_level<KK, HH, NS, false>(REG, cf);                                                  \
        _Pragma("unroll") for (int r = 0; r < 64; ++r) REG[r] = lat_mul(REG[r], ((r >> KK) & 1) ? lat_sgpr((CF)ginv) : lat_sgpr((CF)g));    \
    } else if constexpr (BIT > SH) {                                                            \
        if (ANY) lat_level_hm<KK, HH, NS, false>(REG, cf, MK, g, ginv);                         \
    }
#define WX_SC_INV(KK, HH, REG, BIT, MK, ANY)                                                   \
    if constexpr (BIT == SH) {                                                                  \
        _Pragma("unroll") for (int r = 0; r < 64; ++r) REG[r] = lat_mul(REG[r], ((r >> KK) & 1) ? lat_sgpr((CF)gd) : lat_sgpr((CF)ga));     \
        lat_level<KK, HH, NS, true>(REG, cf);                                                   \
    } else if constexpr (BIT > SH) {                                                            \
        if (ANY) lat_level_hm<KK, HH, NS, true>(REG, cf, MK, ga, gd);                           \
    }

// IO: the signal's type in memory (Float64 in the registers either way: Float32 signals are widened by the loads and rounded once by the
// stores, like the full-tree kernels k_lat_wpt_f64<.., float>)
// the forward transform from the first register layout on (layout A for 4096 / 2048 samples, B for 1024): `src(regs)` fills it -- from
// memory (lat_absorb: k_lat_wpt_treesc_f64) or from a child computed in place (wx_lattice_8k.h: k_lat_wpt_treesc8k_f64)
template <int NS, int SH, typename IO, bool FP32A, typename SRC>
__device__ __forceinline__ void lat_treesc_fwd(IO *__restrict__ ys, unsigned lds0, int lane, unsigned out_stride, unsigned boff_out, const WxLatW &cw,
                                               const WxLatTreeSc *__restrict__ tab, SRC &&src)
{
    constexpr int NQ = 32 >> SH;
    typedef typename std::conditional<FP32A, lat_f2v, double>::type V;
    typedef typename lat_vtraits<V>::coef CF;
    typedef typename lat_row<V>::type ROW;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
    // forward: gl[1] = g, g2 = g^-2: after the shears the a-slot holds a / g, the d-slot d g
    const double g = cw.gl[1], ginv = cw.c.g2 * cw.gl[1];
    V c[64];
    if constexpr (SH >= 6) {
        src(c);
    } else {
        V bb[64];
        if constexpr (SH < 2) {
            V a[64];
            src(a);
            WX_SC_FWD(0, 6, a, 0, tab->mA, true)
            WX_SC_FWD(1, 6, a, 1, tab->mA, tab->anyA)
            lat_t2(a, bb, lds0, lane);
        } else
            src(bb);
        WX_SC_FWD(0, 4, bb, 2, tab->mB + 0, tab->anyB[0])
        WX_SC_FWD(1, 4, bb, 3, tab->mB + 32, tab->anyB[1])
        WX_SC_FWD(2, 4, bb, 4, tab->mB + 64, tab->anyB[2])
        WX_SC_FWD(3, 4, bb, 5, tab->mB + 96, tab->anyB[3])
        if (tab->deepB) {
            WX_SC_FWD(4, 4, bb, 6, tab->mB + 128, tab->anyB[4])
            WX_SC_FWD(5, 4, bb, 7, tab->mB + 160, tab->anyB[5])
        }
        lat_t3(bb, c, lds0, lane);
    }
    unsigned pw[32];
    lat_sc_ptab(pw, tab, lane);                                 // arrives behind the lane-local levels
    const unsigned long long *mk = tab->mC;
    if (!tab->deepB) {
        if (tab->anyC[0]) lat_level_cm<0, NS, false>(c, cf, mk + 0, g, ginv);
        if (tab->anyC[1]) lat_level_cm<1, NS, false>(c, cf, mk + 32, g, ginv);
    }
    if (tab->anyC[2]) lat_level_cm<2, NS, false>(c, cf, mk + 64, g, ginv);
    if (tab->anyC[3]) lat_level_cm<3, NS, false>(c, cf, mk + 96, g, ginv);
    if (tab->anyC[4]) lat_level_cm<4, NS, false>(c, cf, mk + 128, g, ginv);
    if (tab->anyC[5]) lat_level_cm<5, NS, false>(c, cf, mk + 160, g, ginv);
    lat_sync();
    constexpr int HBIT = SH == 0 ? 0 : SH - 1;                  // the lane bit that is bit 11 of the output position
    lat_for<2>([&](auto Hc) {
        constexpr int h = Hc;
        if (((lane >> HBIT) & 1) == h) {
            lat_for<64>([&](auto Rc) {
                constexpr int r = Rc;
                const unsigned p = (r & 1) ? (pw[r >> 1] >> 16) : (pw[r >> 1] & 0xffffu);
                lds_wr<0>(lds0 + p, c[r]);
            });
        }
        lat_sync();
        lat_for<2>([&](auto Gc) {
            constexpr int k0 = 8 * Gc;
            ROW v[8];
            lat_for<8>([&](auto Kc) {
                constexpr int k = k0 + Kc;
                lat_sc_ldrow(v[Kc], lat_sc_row<h, k>(lds0, lane));
            });
            lat_for<8>([&](auto Kc) {
                constexpr int k = k0 + Kc, q = 16 * h + k;
                if constexpr (NQ >= 1) {
                    constexpr int sg = q / NQ, qq = q % NQ;
                    lat_sc_gst(ys + (size_t)sg * out_stride + 128 * qq, 2u * lane, boff_out, v[Kc]);
                } else                                          // 64-sample signals: the piece is signals 2 q (lanes 0 .. 31) and 2 q + 1
                    lat_sc_gst(ys + (size_t)(2 * q) * out_stride, ((unsigned)lane >> 5) * out_stride + 2u * ((unsigned)lane & 31u), boff_out, v[Kc]);
            });
        });
        lat_sync();
    });
}

template <int NS, int WPE, int SH, typename IO = double, bool FP32A = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpt_treesc_f64(
    const IO *__restrict__ x, IO *__restrict__ y, int L, int last_sig, unsigned in_stride, unsigned out_stride, WxLatW cw,
    const WxLatTreeSc *__restrict__ tab)
{
    static_assert(SH >= 0 && SH <= 6, "4096 .. 64 samples");
    __shared__ __attribute__((aligned(16))) double lds[2048];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    // FP32A (IO = float): Float32 arithmetic on PAIRS of signals (lat_f2v, wx_lattice_dev.h) -- the wavefront takes 2 x 2^SH signals with
    // the one tree of the call; the second set follows the first at 2^SH signals' distance
    typedef typename std::conditional<FP32A, lat_f2v, double>::type V;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    // pair kernels: last_sig = the tail wavefront's first signal, cw.tail_bsig = its second set's distance (wx_lat_pair_plan)
    const bool lastw = PAIR && blockIdx.x == gridDim.x - 1;
    const int sig0 = PAIR ? (lastw ? last_sig : (int)(blockIdx.x << (SH + 1))) : min((int)blockIdx.x << SH, last_sig);
    const unsigned bsig = (unsigned)(lastw ? cw.tail_bsig : (1 << SH));
    const unsigned boff_out = bsig * out_stride, boff_in = PAIR ? bsig * in_stride : 0xffffffffu;
    const IO *xs = x + (int64_t)sig0 * in_stride;                // signals in_stride / out_stride elements apart
    lat_treesc_fwd<NS, SH, IO, FP32A>(y + (int64_t)sig0 * out_stride, lds0, lane, out_stride, boff_out, cw, tab, [&](V (&regs)[64]) {
        lat_absorb<(SH < 2 ? 0 : (SH < 6 ? 2 : 6)), 16 * SH>(regs, lds0, xs, lane, cw, in_stride, 0, 0, 0, boff_in);
    });
}

// the inverse transform up to the last register layout (A for 4096 / 2048 samples, B for 1024): `sink(regs)` takes it -- to memory
// (lat_emit: k_lat_iwpt_treesc_f64) or on to the synthesis of an 8192-sample parent (wx_lattice_8k.h)
// `hook(c)` sees the complete coefficient set in the last layout (lane = index bits 5 .. 0) before the first synthesis level: the noise estimate and
// threshold of the one-pass denoise(:dwt) kernel (wx_lattice_dn.h); the plain inverse passes nothing
template <int NS, int SH, bool THR, typename IO, bool FP32A, typename SINK, typename HOOK>
__device__ __forceinline__ void lat_treesc_inv_h(const IO *__restrict__ xs, int sig0, unsigned lds0, int lane, unsigned in_stride, unsigned col_stride,
                                                 unsigned boff_in, unsigned bsig, const WxLatW &cw, const WxLatTreeSc *__restrict__ tab,
                                                 const WxThreshArg &thr, SINK &&sink, HOOK &&hook)
{
    constexpr int NQ = 32 >> SH;                               // 128-element pieces of one signal
    typedef typename std::conditional<FP32A, lat_f2v, double>::type V;
    typedef typename lat_vtraits<V>::coef CF;
    typedef typename lat_row<V>::type ROW;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
    unsigned dep[4] = {0, 0, 0, 0};
    if (col_stride) {
#pragma unroll
        for (int w = 0; w < 4; ++w) dep[w] = tab->dep[64 * w + lane];
    }
    ROW v[16];
    // pieces k0 .. k0 + NK - 1 of half h; piece q = 16 h + k is elements 128 (q mod NQ) + 2 lane, +1 of signal q / NQ
    auto fetch = [&](auto Hc, auto K0c, auto NKc) {
        constexpr int h = Hc, k0 = K0c, nk = NKc;
        lat_for<nk>([&](auto Kc) {
            constexpr int k = k0 + Kc, q = 16 * h + k;
            const unsigned co = ((dep[q >> 3] >> (4 * (q & 7))) & 15u) * col_stride;      // col_stride = 0: dense leaves
            if constexpr (NQ < 1) {
                // 64-sample signals: the piece is signals 2 q (lanes 0 .. 31) and 2 q + 1
                lat_sc_gld(v[k], xs + (size_t)(2 * q) * in_stride, ((unsigned)lane >> 5) * in_stride + 2u * ((unsigned)lane & 31u) + co, boff_in);
            } else {
                constexpr int sg = q / (NQ < 1 ? 1 : NQ), qq = q % (NQ < 1 ? 1 : NQ);
                const IO *src = xs + (size_t)sg * in_stride + 128 * qq;
                if constexpr (qq == 0 && SH <= 2) {
                    // idwt of a pyramid: positions 0 .. 63 are the samples the lane-local tail (wx_dwttail.hip) has rebuilt
                    const IO *hp = reinterpret_cast<const IO *>(thr.head);
                    // both candidate bases are formed OUTSIDE the lane-dependent choice (a scalar-register base inside a divergent branch
                    // gets merged across the arms into a lane-dependent value), the choice is made on the complete addresses
                    const IO *hb = hp ? hp + 64 * (int64_t)(sig0 + sg) : src;
                    const auto pah = lat_sbase(hb) + 2 * lane, pbh = lat_sbase(hb + (hp ? 64u * bsig : boff_in)) + 2 * lane;
                    const auto pas = lat_sbase(src) + (2 * lane + co), pbs = lat_sbase(src + boff_in) + (2 * lane + co);
                    const bool useh = hp && lane < 32;
                    lat_sc_gldp(v[k], useh ? pah : pas, useh ? pbh : pbs);
                } else
                    lat_sc_gld(v[k], src, 2u * lane + co, boff_in);
            }
        });
    };
    double tt[4] = {0, 0, 0, 0};
    if constexpr (THR) {
#pragma unroll
        for (int s4 = 0; s4 < (1 << SH); ++s4)
            tt[s4] = (double)reinterpret_cast<const IO *>(thr.t)[thr.per_signal ? sig0 + s4 : 0] * thr.scale;
    }
    auto put = [&](auto Hc, auto K0c, auto NKc) {
        constexpr int h = Hc, k0 = K0c, nk = NKc;
        if constexpr (THR && !FP32A) {
            // threshold of denoise() (Denoising.jl:527 threshold!(x, th, t) before iwpt): positions [lo, n) of every signal
            lat_for<nk>([&](auto Kc) {
                constexpr int k = k0 + Kc, q = 16 * h + k, sg = q / (NQ < 1 ? 1 : NQ), qq = q % (NQ < 1 ? 1 : NQ);
                const int pos = 128 * qq + 2 * lane;
                if (qq == 0 && thr.head && lane < 32) return;      // the tail has thresholded what it read
                if (pos >= thr.lo) v[k].x = wx_thresh<double>(v[k].x, tt[sg], thr.kind);
                if (pos + 1 >= thr.lo) v[k].y = wx_thresh<double>(v[k].y, tt[sg], thr.kind);
            });
        }
        lat_for<nk>([&](auto Kc) {
            constexpr int k = k0 + Kc;
            lat_sc_strow(lat_sc_row<h, k>(lds0, lane), v[k]);
        });
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 8> I8;
    typedef std::integral_constant<int, 16> I16;
    fetch(I0{}, I0{}, I16{});
    unsigned pw[32];
    lat_sc_ptab(pw, tab, lane);
    V c[64];
    // half 0 (the lanes with bit HBIT clear): every lane reads, the other lanes' values are replaced below; the first eight
    // pieces of half 1 travel meanwhile -- all sixteen would not fit the 256 registers next to c[] and the table
    put(I0{}, I0{}, I16{});
    lat_sync();
    fetch(I1{}, I0{}, I8{});
    lat_for<64>([&](auto Rc) {
        constexpr int r = Rc;
        const unsigned p = (r & 1) ? (pw[r >> 1] >> 16) : (pw[r >> 1] & 0xffffu);
        lat_sc_rd(c[r], lds0 + p);
    });
    lat_wait16<0>(c);
    lat_wait16<16>(c);
    lat_wait16<32>(c);
    lat_wait16<48>(c);
    lat_sync();
    put(I1{}, I0{}, I8{});
    fetch(I1{}, I8{}, I8{});
    put(I1{}, I8{}, I8{});
    lat_sync();
    constexpr int HBIT = SH == 0 ? 0 : SH - 1;
    if ((lane >> HBIT) & 1) {
        lat_for<64>([&](auto Rc) {
            constexpr int r = Rc;
            const unsigned p = (r & 1) ? (pw[r >> 1] >> 16) : (pw[r >> 1] & 0xffffu);
            lat_sc_rd_keep(c[r], lds0 + p);
        });
    }
    lat_wait16<0>(c);
    lat_wait16<16>(c);
    lat_wait16<32>(c);
    lat_wait16<48>(c);
    lat_sync();
    hook(c);
    // synthesis: gl[1] = 1 / g, g2 = g^2 -- the a-slot of a split node enters as a / g, the d-slot as d g
    const double ga = cw.gl[1], gd = cw.c.g2 * cw.gl[1];
    const unsigned long long *mk = tab->mC;
    if (tab->anyC[5]) lat_level_cm<5, NS, true>(c, cf, mk + 160, ga, gd);
    if (tab->anyC[4]) lat_level_cm<4, NS, true>(c, cf, mk + 128, ga, gd);
    if (tab->anyC[3]) lat_level_cm<3, NS, true>(c, cf, mk + 96, ga, gd);
    if (tab->anyC[2]) lat_level_cm<2, NS, true>(c, cf, mk + 64, ga, gd);
    if (!tab->deepB) {
        if (tab->anyC[1]) lat_level_cm<1, NS, true>(c, cf, mk + 32, ga, gd);
        if (tab->anyC[0]) lat_level_cm<0, NS, true>(c, cf, mk + 0, ga, gd);
    }
    if constexpr (SH >= 6) {
        sink(c);
    } else {
        V bb[64];
        lat_t3i(c, bb, lds0, lane);
        if (tab->deepB) {
            WX_SC_INV(5, 4, bb, 7, tab->mB + 160, tab->anyB[5])
            WX_SC_INV(4, 4, bb, 6, tab->mB + 128, tab->anyB[4])
        }
        WX_SC_INV(3, 4, bb, 5, tab->mB + 96, tab->anyB[3])
        WX_SC_INV(2, 4, bb, 4, tab->mB + 64, tab->anyB[2])
        WX_SC_INV(1, 4, bb, 3, tab->mB + 32, tab->anyB[1])
        WX_SC_INV(0, 4, bb, 2, tab->mB + 0, tab->anyB[0])
        if constexpr (SH >= 2) {
            sink(bb);
        } else {
            V a[64];
            lat_t2i(bb, a, lds0, lane);
            WX_SC_INV(1, 6, a, 1, tab->mA, tab->anyA)
            WX_SC_INV(0, 6, a, 0, tab->mA, true)
            sink(a);
        }
    }
}

template <int NS, int SH, bool THR, typename IO, bool FP32A, typename SINK>
__device__ __forceinline__ void lat_treesc_inv(const IO *__restrict__ xs, int sig0, unsigned lds0, int lane, unsigned in_stride, unsigned col_stride,
                                               unsigned boff_in, unsigned bsig, const WxLatW &cw, const WxLatTreeSc *__restrict__ tab,
                                               const WxThreshArg &thr, SINK &&sink)
{
    typedef typename std::conditional<FP32A, lat_f2v, double>::type V;
    lat_treesc_inv_h<NS, SH, THR, IO, FP32A>(xs, sig0, lds0, lane, in_stride, col_stride, boff_in, bsig, cw, tab, thr, sink, [](V (&)[64]) {});
}

template <int NS, int WPE, int SH, bool THR, typename IO = double, bool FP32A = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_iwpt_treesc_f64(
    const IO *__restrict__ xw, IO *__restrict__ y, int L, int last_sig, unsigned in_stride, unsigned col_stride,
    unsigned out_stride, WxLatW cw, const WxLatTreeSc *__restrict__ tab, WxThreshArg thr)
{
    static_assert(SH >= 0 && SH <= 6, "4096 .. 64 samples");
    static_assert(!(THR && SH > 2), "the threshold rides on the kernels of 1024 samples and more");
    __shared__ __attribute__((aligned(16))) double lds[2048];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    typedef typename std::conditional<FP32A, lat_f2v, double>::type V;
    static_assert(!(FP32A && THR), "the threshold of denoise() rides on the Float64-register kernels");
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    const bool lastw = PAIR && blockIdx.x == gridDim.x - 1;
    const int sig0 = PAIR ? (lastw ? last_sig : (int)(blockIdx.x << (SH + 1))) : min((int)blockIdx.x << SH, last_sig);
    const unsigned bsig = (unsigned)(lastw ? cw.tail_bsig : (1 << SH));
    const unsigned boff_in = bsig * in_stride, boff_out = PAIR ? bsig * out_stride : 0xffffffffu;
    IO *ys = y + (int64_t)sig0 * out_stride;
    lat_treesc_inv<NS, SH, THR, IO, FP32A>(xw + (int64_t)sig0 * in_stride, sig0, lds0, lane, in_stride, col_stride, boff_in, bsig, cw, tab, thr,
                                           [&](V (&regs)[64]) { lat_emit<(SH < 2 ? 0 : (SH < 6 ? 2 : 6)), 16 * SH>(regs, lds0, ys, lane, cw, out_stride, 0, 0, boff_out); });
}
#undef WX_SC_FWD
#undef WX_SC_INV

}  // namespace
