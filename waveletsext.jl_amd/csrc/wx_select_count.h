// wx_select_count.h -- exact order statistics of values held in REGISTERS, by counting: the medians of the noise estimate (noisest, Denoising.jl:214-232:
// mad!(finest details) / 0.6745 with Statistics.median! = a/2 + b/2 of the two central order statistics) for the one-pass denoise kernels
// (wx_lattice_dn.h) and the wavefront-per-signal kernel of wx_noisest_* (wx_denoise.hip: k_mad_count).
//
// A pivot p per signal, c = #{v < p} summed over the signal's lanes (a ballot and a population count per register when the signal is the whole
// wavefront, per-lane counters and a reduction over the lane group otherwise), the bracket [lo, hi) with #{v < lo} <= k < #{v < hi} halves in value
// until one element is left in it (middle of lo and hi; the order-preserving integer image of the doubles takes over when values stop separating:
// at most 64 more steps whatever the data); a pivot that makes no progress twice -- many equal values at the median: sparse or quantised signals --
// is replaced by the successor of the smallest candidate, which removes all its copies at once.  No arithmetic on the values: the result is the
// element of rank k itself, as a sort would find it.  Values are handled as doubles (Float32 data widen exactly; T only decides the arithmetic of
// the deviations and of a/2 + b/2, which the reference does in the signal's type).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

namespace {

template <int... I, typename F> __device__ __forceinline__ void dn_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <int N, typename F> __device__ __forceinline__ void dn_for(F &&f)
{
    dn_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// order-preserving image of the doubles (NaN aside): a < b <=> key(a) < key(b), -0 just below +0
__device__ __forceinline__ unsigned long long dn_key(double d)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dn_unkey(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
// the smallest double above a (finite a)
__device__ __forceinline__ double dn_next_up(double a)
{
    if (a == 0.0) return __longlong_as_double(1ll);
    return dn_unkey(dn_key(a) + 1ull);
}
// the value of lane ^ 32 beside the lane's own, as (value of the lower half's lane, value of the upper half's lane): v_permlane32_swap, no LDS
// (with ds_bpermute in its loops the 256-sample kernel spilled 19 ... 168 registers)
__device__ __forceinline__ void dn_x32(unsigned v, unsigned &lower, unsigned &upper)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    lower = r[0];
    upper = r[1];
}
__device__ __forceinline__ void dn_x32(double v, double &lower, double &upper)
{
    unsigned l0, u0, l1, u1;
    dn_x32((unsigned)__double2loint(v), l0, u0);
    dn_x32((unsigned)__double2hiint(v), l1, u1);
    lower = __hiloint2double((int)l1, (int)l0);
    upper = __hiloint2double((int)u1, (int)u0);
}
// a signal's lanes: GW of them, ST apart (ST = 1: consecutive lanes)
constexpr int dn_log2(int v) { return v <= 1 ? 0 : 1 + dn_log2(v >> 1); }
template <int O> __device__ __forceinline__ double dn_xor(double v)
{
    if constexpr (O == 32) { double a, b; dn_x32(v, a, b); return ((unsigned)__lane_id() & 32u) ? a : b; }
    else return __shfl_xor(v, O, 64);
}
template <int GW, int ST = 1> __device__ __forceinline__ double dn_gmin(double v)
{
    dn_for<dn_log2(GW)>([&](auto Ic) { constexpr int o = (ST * GW / 2) >> Ic; const double u = dn_xor<o>(v); v = u < v ? u : v; });
    return v;
}
template <int GW, int ST = 1> __device__ __forceinline__ double dn_gmax(double v)
{
    dn_for<dn_log2(GW)>([&](auto Ic) { constexpr int o = (ST * GW / 2) >> Ic; const double u = dn_xor<o>(v); v = u > v ? u : v; });
    return v;
}
template <int GW, int ST = 1> __device__ __forceinline__ int dn_gsum(int v)
{
    if constexpr (GW == 16 && ST == 1) {
        // every lane of a 16-lane row gets the row's sum: four DPP adds (quad_perm 1 0 3 2, quad_perm 2 3 0 1, row_half_mirror, row_mirror)
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    } else {
        dn_for<dn_log2(GW)>([&](auto Ic) {
            constexpr int o = (ST * GW / 2) >> Ic;
            if constexpr (o == 32) { unsigned a, b; dn_x32((unsigned)v, a, b); v = (int)(a + b); }
            else v += __shfl_xor(v, o, 64);
        });
    }
    return v;
}
// BW > 1: the signal is BW wavefronts of one workgroup (every wavefront runs the same control flow on the same block-wide sums, so the pivots agree
// and the barriers below are met by all): the per-wavefront result goes through a small LDS table
template <int BW> __device__ __forceinline__ int dn_block_sum(int v)
{
    if constexpr (BW == 1) return v;
    else {
        __shared__ int tab[BW];
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) tab[w] = v;
        __syncthreads();
        int s = 0;
#pragma unroll
        for (int j = 0; j < BW; ++j) s += tab[j];
        __syncthreads();
        return s;
    }
}
template <int BW, bool MAX> __device__ __forceinline__ double dn_block_ext(double v)
{
    if constexpr (BW == 1) return v;
    else {
        __shared__ double tab[BW];
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) tab[w] = v;
        __syncthreads();
        double s = tab[0];
#pragma unroll
        for (int j = 1; j < BW; ++j) s = MAX ? (tab[j] > s ? tab[j] : s) : (tab[j] < s ? tab[j] : s);
        __syncthreads();
        return s;
    }
}
// c += #{lanes of the group with pred}: the whole wavefront -> ballot + population count (scalar), else a per-lane counter (dn_gsum later)
template <int GW> __device__ __forceinline__ void dn_acc(int &c, bool pred)
{
    if constexpr (GW == 64) c += (int)__popcll(__builtin_amdgcn_ballot_w64(pred));
    else c += pred ? 1 : 0;
}
template <int GW, int ST = 1> __device__ __forceinline__ int dn_fin(int c)
{
    if constexpr (GW == 64) return c;
    else return dn_gsum<GW, ST>(c);
}
__device__ __forceinline__ bool dn_any(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0; }

// the detail registers of the layout the root level ran in: index bit CB of the register set, class (= signal bits held by the register
// index) q = r mod 2^CB
template <int CB, int NR, typename F> __device__ __forceinline__ void dn_each(F &&f)
{
    dn_for<NR>([&](auto Rc) {
        constexpr int r = Rc;
        if constexpr (CB < 0) f(Rc, std::integral_constant<int, 0>{});                    // every register, one class (dn_noisest in the last layout)
        else if constexpr ((r >> CB) & 1) f(Rc, std::integral_constant<int, (r & ((1 << CB) - 1))>{});
    });
}
constexpr int dn_nc(int cb) { return cb < 0 ? 1 : 1 << cb; }

// median (Statistics.median!: a/2 + b/2 of the order statistics k and k + 1, cnt even) of v = e (DEV = false) or |e - ctr| (DEV = true) per class
// and lane group; [blo, bhi): #{v < blo} = 0, #{v < bhi} = cnt
template <int CB, int GW, int ST, bool DEV, int NR = 64, typename T = double, int BW = 1>
__device__ __forceinline__ void dn_median(const double (&e)[NR], const double (&ctr)[dn_nc(CB)], const double (&blo)[dn_nc(CB)],
                                          const double (&bhi)[dn_nc(CB)], int cnt, bool act, double (&med)[dn_nc(CB)])
{
    constexpr int NC = dn_nc(CB);
    const int k = cnt / 2 - 1;
    // the deviations are formed again in every pass: hoisted out of the loops (they do not change) they are 32 / 64 more live doubles -- the
    // centre goes through an opaque copy per pass
    double cc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) cc[q] = ctr[q];
    auto opaque = [&]() {
        if constexpr (DEV) {
#pragma unroll
            for (int q = 0; q < NC; ++q) asm volatile("" : "+v"(cc[q]));
        }
    };
    auto val = [&](auto Rc, auto Qc) -> double {
        constexpr int r = Rc, q = Qc;
        if constexpr (DEV) return (double)(T)fabs((double)(T)((T)e[r] - (T)cc[q]));
        else return e[r];
    };
    // smallest v >= lo of every group
    auto min_ge = [&](const double (&lo)[NC], double (&a)[NC]) {
#pragma unroll
        for (int q = 0; q < NC; ++q) a[q] = __builtin_inf();
        opaque();
        dn_each<CB, NR>([&](auto Rc, auto Qc) {
            constexpr int q = Qc;
            const double v = val(Rc, Qc);
            const double w = v >= lo[q] ? v : __builtin_inf();
            a[q] = w < a[q] ? w : a[q];
        });
#pragma unroll
        for (int q = 0; q < NC; ++q) a[q] = dn_block_ext<BW, false>(dn_gmin<GW, ST>(a[q]));
    };
    double lo[NC], hi[NC];
    int clo[NC], chi[NC], stall[NC];
    bool done[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) { lo[q] = blo[q]; hi[q] = bhi[q]; clo[q] = 0; chi[q] = cnt; stall[q] = 0; done[q] = !act; }
    for (int it = 0; it < 192; ++it) {
        bool forced[NC], anyst = false;
#pragma unroll
        for (int q = 0; q < NC; ++q) { forced[q] = !done[q] && stall[q] >= 2; anyst = anyst || forced[q]; }
        if (dn_any(anyst)) {
            double a[NC];
            min_ge(lo, a);
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (forced[q]) { lo[q] = a[q]; stall[q] = 0; }
        }
        double p[NC];
        bool live = false, odd = false, vsp[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            p[q] = hi[q];
            vsp[q] = false;
            if (!done[q]) {
                if (chi[q] - clo[q] <= 1) done[q] = true;
                else {
                    p[q] = forced[q] ? dn_next_up(lo[q]) : lo[q] * 0.5 + hi[q] * 0.5;
                    vsp[q] = !(p[q] > lo[q] && p[q] < hi[q]) || (it >= 48 && !forced[q]);      // the value's middle does not separate (any more)
                    odd = odd || vsp[q];
                }
            }
        }
        if (dn_any(odd)) {
            // rare: the middle of the order-preserving integer images; no double between lo and hi -> order statistic k is lo itself
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (vsp[q]) {
                    const unsigned long long kl = dn_key(lo[q]), kh = dn_key(hi[q]);
                    p[q] = dn_unkey(kl + ((kh - kl) >> 1));
                    if (!(p[q] > lo[q] && p[q] < hi[q])) { done[q] = true; p[q] = hi[q]; }
                }
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) live = live || !done[q];
        if (!dn_any(live)) break;
        int c[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) c[q] = 0;
        opaque();
        dn_each<CB, NR>([&](auto Rc, auto Qc) {
            constexpr int r = Rc, q = Qc;
            dn_acc<GW>(c[q], val(Rc, Qc) < p[q]);
            // whole-wavefront signals: the count is taken every eight registers -- left alone the compiler forms all 32 / 64 lane masks first, 64 /
            // 128 scalar registers of them (66 ... 202 spilled)
            if constexpr (GW == 64 && (r & 7) == 7) asm volatile("" : "+s"(c[q]));
        });
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int cq = dn_block_sum<BW>(dn_fin<GW, ST>(c[q]));
            if (!done[q]) {
                stall[q] = (cq == clo[q] || cq == chi[q]) ? stall[q] + 1 : 0;
                if (cq <= k) { lo[q] = p[q]; clo[q] = cq; }
                else { hi[q] = p[q]; chi[q] = cq; }
            }
        }
    }
    // order statistic k = the smallest v >= lo; k + 1 = the same value when at least k + 2 elements are <= it, else the next larger one
    double a[NC], nx[NC];
    int le[NC];
    min_ge(lo, a);
#pragma unroll
    for (int q = 0; q < NC; ++q) { nx[q] = __builtin_inf(); le[q] = 0; }
    // two sweeps: in one, the compiler counts first and keeps the 64 deviations for the minimum (166 spilled registers in the 256-sample kernel)
    opaque();
    dn_each<CB, NR>([&](auto Rc, auto Qc) {
        constexpr int r = Rc, q = Qc;
        dn_acc<GW>(le[q], val(Rc, Qc) <= a[q]);
        if constexpr (GW == 64 && (r & 7) == 7) asm volatile("" : "+s"(le[q]));
    });
    opaque();
    dn_each<CB, NR>([&](auto Rc, auto Qc) {
        constexpr int q = Qc;
        const double v = val(Rc, Qc);
        const double w = v > a[q] ? v : __builtin_inf();
        nx[q] = w < nx[q] ? w : nx[q];
    });
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int leq = dn_block_sum<BW>(dn_fin<GW, ST>(le[q]));
        const double nmin = dn_block_ext<BW, false>(dn_gmin<GW, ST>(nx[q]));
        const double b = leq >= k + 2 ? a[q] : nmin;
        med[q] = (double)(T)((T)((T)a[q] / (T)2) + (T)((T)b / (T)2));
    }
}

// noise estimates of the signals whose finest details sit in the registers with bit CB set (CB < 0: in every register of the lanes with `act`):
// sig[q] for class q of this lane's group (GW lanes, ST apart)
// NR registers per lane hold the values (64 in the lattice kernels); cnt_ = number of values per signal when it is not the full (registers x lanes)
// block (k_mad_count: slots beyond it hold +Inf, which the counting never reaches below rank cnt_)
template <int CB, int GW, int ST = 1, int NR = 64, typename T = double, int BW = 1>
__device__ __forceinline__ void dn_noisest(const double (&e)[NR], double (&sig)[dn_nc(CB)], bool act = true, int cnt_ = 0)
{
    constexpr int NC = dn_nc(CB);
    const int cnt = cnt_ ? cnt_ : (CB < 0 ? NR : ((NR / 2) >> CB)) * GW * BW;
    double vmin[NC], vmax[NC], zero[NC], med[NC], dhi[NC], mad[NC];
    int bad[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) { vmin[q] = __builtin_inf(); vmax[q] = -__builtin_inf(); bad[q] = 0; zero[q] = 0.0; }
    dn_each<CB, NR>([&](auto Rc, auto Qc) {
        constexpr int r = Rc, q = Qc;
        const double v = e[r];
        vmin[q] = v < vmin[q] ? v : vmin[q];
        vmax[q] = v > vmax[q] ? v : vmax[q];
        dn_acc<GW>(bad[q], v != v);
        if constexpr (GW == 64 && (r & 7) == 7) asm volatile("" : "+s"(bad[q]));
    });
    double hi0[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        vmin[q] = dn_block_ext<BW, false>(dn_gmin<GW, ST>(vmin[q])); vmax[q] = dn_block_ext<BW, true>(dn_gmax<GW, ST>(vmax[q]));
        bad[q] = dn_block_sum<BW>(dn_fin<GW, ST>(bad[q]));
        hi0[q] = dn_next_up(vmax[q]);
    }
    dn_median<CB, GW, ST, false, NR, T, BW>(e, zero, vmin, hi0, cnt, act, med);
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const double d0 = (double)(T)fabs((double)(T)((T)vmin[q] - (T)med[q])), d1 = (double)(T)fabs((double)(T)((T)vmax[q] - (T)med[q]));
        dhi[q] = dn_next_up(d0 > d1 ? d0 : d1);
    }
    dn_median<CB, GW, ST, true, NR, T, BW>(e, med, zero, dhi, cnt, act, mad);
#pragma unroll
    for (int q = 0; q < NC; ++q) sig[q] = bad[q] ? __builtin_nan("") : (double)(T)((T)mad[q] / (T)0.6745);
}

}  // namespace
