#pragma once
// wx_lattice.hip -- full-tree 1-D wavelet packets (wpt / iwpt by level, Float64, n = 4096, L >= 6) as a lattice of
// plane rotations held in the registers of ONE wavefront per signal.
//
// Reference semantics: dwt/dwt_one_level.jl:79-107 (analysis step) and :192-223 (synthesis step), driven level by
// level by Wavelets.jl's wpt / iwpt (call sites dwt/dwt_all.jl:152-166, 210-225).
//
// Why: the direct form costs 2F multiply-adds per output pair; at F = 8, L = 10 that is 4.3e10 Float64 flops per
// 65536 x 4096 batch = the whole FP64 budget of the 60 %-of-HBM target (0.89 ms).  The 2x2 polyphase matrix of an
// orthonormal QMF is paraunitary:
//     [a]   [ Qe(w)      Qo(w)   ] [v_even]      Qe(w) = sum q[2m] w^m, Qo(w) = sum q[2m+1] w^m, w = advance one pair
//     [d] = [-Qo(1/w)    Qe(1/w) ] [v_odd ]
// and factors into F/2 rotations c_j [[1, t_j], [-t_j, 1]] separated by "advance the odd channel by one pair"
// (Vaidyanathan's lattice; factorisation on the host in long double, wx_lattice_factor): F multiply-adds per pair,
// half the direct form, and the common gain (prod c_j)^L is applied once at the end.
//
// Layout: level l acts on index bit b = l - 1 of the natural sample index p with dilation 2^b and period n (the
// a-children stay on the slots with bit b = 0, the d-children on bit b = 1), so the depth-L transform is L in-place
// stencils followed by a bit reversal of the low L index bits (Wavelets.jl's packet order).  A wavefront holds the
// 4096 samples as 64 registers per lane and changes which six index bits are register-resident four times:
//     L0 (eight full 128-byte lines per load)       ->  A: reg p[5:0]   levels 1-2,  halo = wave rotate (DPP)
//                                                   ->  B: reg p[7:2]   levels 3-6,  halo = row rotate (DPP)
//                                                   ->  C: reg p[11:6]  levels 7-12, whole sequences in registers
//                                                   ->  S: full 128-byte lines per 8 lanes for the stores
// Each exchange moves 16 registers per round through an 8.6 KiB LDS window with conflict-free ds_write_b64 /
// ds_read_b64 (maps derived and checked in tools/lattice_lds_maps.py).  No workgroup barrier anywhere: one wavefront
// = one workgroup, ordering is wave-level.  Per signal and lane: F/2 * 64 * L FMAs (2560 for db4, L = 10), ~110
// DPP moves for the halos, 256 LDS writes + 256 LDS reads, one 32 KiB read and one 32 KiB write of HBM.
//
// Rounding: the rotations reassociate the reference's tap sums; measured difference from the oracle <= 3e-15
// relative for every filter of the table at L = 12 (tools/lattice_proto.py), far inside the 1e-10 bar.
#include "wx_common.h"
#include "wx_kernels.h"
#include <cmath>
#include <cstdlib>
#include <utility>

#define WX_LAT_MAXS 10        // rotations per level = F / 2 (F <= 20)
#define WX_LAT_LDS 1104       // elements of the LDS window (max over the eight exchanges: 1102)

// Rotation j (c_j [[1, t_j], [-t_j, 1]]) is applied as two in-place shears on (u, w = sigma_j v):
//     u += p_j w,  w -= kap_j u      p_j = t_j / sigma_j,  kap_j = sigma_j t_j c_j^2,  sigma_{j+1} = sigma_j c_j^2
// (no temporary: the 128 data registers of a lane leave room for 3 wavefronts per SIMD).  After a level the a-slot
// holds a / g and the d-slot d * g (g = prod c_j), so a leaf whose path took k detail branches carries g^(2k - L):
// one multiply per element at the end (analysis) or at the start (synthesis).
struct WxLat {
    double p[WX_LAT_MAXS];
    double kap[WX_LAT_MAXS];
    double g0;                // analysis: g^L, synthesis: g^-L      (leaf with k = 0)
    double g2;                // analysis: g^-2, synthesis: g^2      (per detail branch on the path)
};

namespace {

__device__ __forceinline__ void lat_sync()
{
    // wave-level ordering of LDS traffic: the hardware executes a wavefront's DS operations in order; this only
    // stops the compiler from moving a lane's loads above another lane's stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// keeps a wave-uniform pointer in scalar registers and opaque to address reassociation, so that the accesses below use
// the "scalar base + 32-bit lane offset + immediate" form (one address VGPR instead of a 64-bit pair per access)
typedef double lat_d2 __attribute__((ext_vector_type(2)));
typedef double lat_d4 __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(1))) *lat_gc;
typedef double __attribute__((address_space(1))) *lat_gm;
// What one 64-bit register slot holds (V): a Float64 sample, or -- Float32 ARITHMETIC, round 5 -- the Float32 samples of TWO
// signals (A in .x, B in .y).  With the pair every rotation is one v_pk_fma_f32 (the issue cost of one v_fma_f64), every halo move,
// LDS exchange and address is shared by the two signals, so a wavefront does for 2 x 4096 Float32 samples exactly what it does for
// 4096 Float64 samples: same bytes, same instructions.  The reference itself rounds to Float32 at every accumulate
// (dwt/dwt_one_level.jl:97-103); until round 4 Float32 signals were transformed in Float64 registers, bound by FP64 issue at half
// the bytes.
typedef float lat_f2v __attribute__((ext_vector_type(2)));
template <typename V> struct lat_vtraits;
template <> struct lat_vtraits<double> { typedef double coef; static constexpr int pair = 0; };
template <> struct lat_vtraits<lat_f2v> { typedef float coef; static constexpr int pair = 1; };
__device__ __forceinline__ double lat_fma(double c, double a, double b) { return fma(c, a, b); }
__device__ __forceinline__ lat_f2v lat_fma(float c, lat_f2v a, lat_f2v b)
{
    const lat_f2v cc = {c, c};
    return __builtin_elementwise_fma(cc, a, b);
}
__device__ __forceinline__ double lat_mul(double a, double g) { return a * g; }
__device__ __forceinline__ lat_f2v lat_mul(lat_f2v a, float g) { const lat_f2v gg = {g, g}; return a * gg; }
template <typename V> struct lat_v2 { V x, y; };            // two consecutive samples (of each signal)
// the rotation coefficients as Float32 in SCALAR registers (v_readfirstlane of the one conversion at kernel entry).  Converting at the
// point of use -- (float)cf.p[j] inside the level functions -- left it to the compiler where the v_cvt_f32_f64 runs; in the tree-driven
// kernels the levels are exec-masked regions, and a conversion placed inside one region was reused in the next under another mask:
// wrong inverse transforms for 12+ taps at 1024 samples (found by tests/test_gpu_lattice_pairs.py before the kernels shipped).
struct WxLatF {
    float p[WX_LAT_MAXS];
    float kap[WX_LAT_MAXS];
};
template <int NS> __device__ __forceinline__ WxLatF lat_cf32(const WxLat &c)
{
    WxLatF f;
#pragma unroll
    for (int j = 0; j < WX_LAT_MAXS; ++j) {
        f.p[j] = j < NS ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)c.p[j]))) : 0.0f;
        f.kap[j] = j < NS ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)c.kap[j]))) : 0.0f;
    }
    return f;
}
template <int NS, bool PAIR> __device__ __forceinline__ typename std::conditional<PAIR, WxLatF, const WxLat &>::type lat_cfsel(const WxLat &c)
{
    if constexpr (PAIR) return lat_cf32<NS>(c);
    else return c;
}
__device__ __forceinline__ float lat_sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double lat_sgpr(double v) { return v; }
// global accesses of the kernels: 16 bytes per lane, streamed once (WX_LAT_NT: non-temporal hint)
#ifndef WX_LAT_NT
#define WX_LAT_NT 3     // measured (db4, L = 10, 65536 signals): forward 0.82 -> 0.79 ms, inverse 0.88 -> 0.80 ms
#endif
__device__ __forceinline__ lat_d2 lat_ld2(const double __attribute__((address_space(1))) *p)
{
    typedef const lat_d2 __attribute__((address_space(1))) *P;
#if WX_LAT_NT & 1
    return __builtin_nontemporal_load((P)p);
#else
    return *(P)p;
#endif
}
__device__ __forceinline__ void lat_st2(double __attribute__((address_space(1))) *p, lat_d2 v)
{
    typedef lat_d2 __attribute__((address_space(1))) *P;
#if WX_LAT_NT & 2
    __builtin_nontemporal_store(v, (P)p);
#else
    *(P)p = v;
#endif
}
// Float32 signals in memory, Float64 in the registers (k_lat_wpt_f64<.., float> / k_lat_iwpt_f64<.., float>): the same
// element offsets, 8 bytes per lane instead of 16
typedef float lat_f2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(1))) *lat_gcf;
typedef float __attribute__((address_space(1))) *lat_gmf;
__device__ __forceinline__ lat_d2 lat_ld2(const float __attribute__((address_space(1))) *p)
{
    typedef const lat_f2 __attribute__((address_space(1))) *P;
    const lat_f2 v = *(P)p;
    lat_d2 r;
    r.x = (double)v.x;
    r.y = (double)v.y;
    return r;
}
__device__ __forceinline__ void lat_st2(float __attribute__((address_space(1))) *p, lat_d2 v)
{
    typedef lat_f2 __attribute__((address_space(1))) *P;
    lat_f2 o;
    o.x = (float)v.x;
    o.y = (float)v.y;
    *(P)p = o;
}
// stores of the wpd kernel (28 of the 30 GiB it moves): WX_LAT_WPD_NT selects the hint separately
#ifndef WX_LAT_WPD_NT
#define WX_LAT_WPD_NT 0     // measured on config 2 (same box): plain stores 5.95 ms, non-temporal 6.20 ms
#endif
__device__ __forceinline__ void lat_st2w(double __attribute__((address_space(1))) *p, lat_d2 v)
{
    typedef lat_d2 __attribute__((address_space(1))) *P;
#if WX_LAT_WPD_NT
    __builtin_nontemporal_store(v, (P)p);
#else
    *(P)p = v;
#endif
}
__device__ __forceinline__ void lat_st2w(float __attribute__((address_space(1))) *p, lat_d2 v) { lat_st2(p, v); }
__device__ __forceinline__ lat_gc lat_sbase(const double *p)
{
    lat_gc g = (lat_gc)p;
    asm("" : "+s"(g));
    return g;
}
__device__ __forceinline__ lat_gm lat_sbase(double *p)
{
    lat_gm g = (lat_gm)p;
    asm("" : "+s"(g));
    return g;
}
__device__ __forceinline__ lat_gcf lat_sbase(const float *p)
{
    lat_gcf g = (lat_gcf)p;
    asm("" : "+s"(g));
    return g;
}
__device__ __forceinline__ lat_gmf lat_sbase(float *p)
{
    lat_gmf g = (lat_gmf)p;
    asm("" : "+s"(g));
    return g;
}
// two consecutive samples as register values V from memory of element type IO, at (wave-uniform pointer) + (32-bit lane offset):
// (double, double) 16 bytes per lane, (double, float) 8 bytes widened, (lat_f2v, float) 8 bytes from signal A and 8 bytes from
// signal B at base + boff, interleaved.  Both streams use the "scalar base + lane offset" address form (B's base is a scalar add).
__device__ __forceinline__ lat_v2<double> lat_ldv(const double *base, unsigned off, unsigned, double *)
{
    const lat_d2 t = lat_ld2(lat_sbase(base) + off);
    return lat_v2<double>{t.x, t.y};
}
__device__ __forceinline__ lat_v2<double> lat_ldv(const float *base, unsigned off, unsigned, double *)
{
    const lat_d2 t = lat_ld2(lat_sbase(base) + off);
    return lat_v2<double>{t.x, t.y};
}
__device__ __forceinline__ lat_v2<lat_f2v> lat_ldv(const float *base, unsigned off, unsigned boff, lat_f2v *)
{
    typedef const lat_f2 __attribute__((address_space(1))) *P;
    const lat_f2 a = *(P)(lat_sbase(base) + off), b = *(P)(lat_sbase(base + boff) + off);
    lat_v2<lat_f2v> r;
    r.x.x = a.x; r.x.y = b.x;
    r.y.x = a.y; r.y.y = b.y;
    return r;
}
__device__ __forceinline__ void lat_stv(double *base, unsigned off, unsigned, double v0, double v1, bool wpd)
{
    lat_d2 o;
    o.x = v0; o.y = v1;
    if (wpd) lat_st2w(lat_sbase(base) + off, o); else lat_st2(lat_sbase(base) + off, o);
}
__device__ __forceinline__ void lat_stv(float *base, unsigned off, unsigned, double v0, double v1, bool)
{
    lat_d2 o;
    o.x = v0; o.y = v1;
    lat_st2(lat_sbase(base) + off, o);
}
__device__ __forceinline__ void lat_stv(float *base, unsigned off, unsigned boff, lat_f2v v0, lat_f2v v1, bool)
{
    typedef lat_f2 __attribute__((address_space(1))) *P;
    lat_f2 a, b;
    a.x = v0.x; a.y = v1.x;
    b.x = v0.y; b.y = v1.y;
    *(P)(lat_sbase(base) + off) = a;
    *(P)(lat_sbase(base + boff) + off) = b;
}

// LDS traffic as explicit single ds_write_b64 / ds_read_b64 (byte address VGPR + 16-bit immediate).  The compiler would
// pair them into ds_write2_b64 / ds_read2_b64: half the read rate (MI355X_MICROARCH.md, LDS table) and, worse, pairs of
// destination registers that must be adjacent -- with the depth-dependent register order of the C layout that costs
// hundreds of copies and spills.  volatile asm statements keep their program order, the hardware executes a
// wavefront's DS operations in order, and lat_wait() is the only wait the reads need.
template <int OFF, typename V> __device__ __forceinline__ void lds_wr(unsigned addr, V v)
{
    static_assert(sizeof(V) == 8, "one 64-bit slot");
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF, typename V = double> __device__ __forceinline__ V lds_rd(unsigned addr)
{
    static_assert(sizeof(V) == 8, "one 64-bit slot");
    V v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// all outstanding DS reads have landed; the operands tie the values to the wait so that no use is scheduled above it
template <typename V> __device__ __forceinline__ void lat_wait8(V &a, V &b, V &c, V &d, V &e, V &f, V &g, V &h)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "memory");
}

template <int CTRL> __device__ __forceinline__ double lat_dpp(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    // rotations: every lane has a source, so with bound_ctrl the old value is dead and no initialising move is emitted
    const int plo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(phi, plo);
}
template <int CTRL> __device__ __forceinline__ lat_f2v lat_dpp(lat_f2v v)
{
    const int plo = __builtin_amdgcn_update_dpp(0, __float_as_int(v.x), CTRL, 0xF, 0xF, true);
    const int phi = __builtin_amdgcn_update_dpp(0, __float_as_int(v.y), CTRL, 0xF, 0xF, true);
    lat_f2v r;
    r.x = __int_as_float(plo);
    r.y = __int_as_float(phi);
    return r;
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <int... I, typename F> __device__ __forceinline__ void lat_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void lat_for(F &&f)
{
    lat_for_impl(std::make_integer_sequence<int, N>{}, f);
}
// value held by the lane that owns the chunk D places further along the same dilated sequence (D < 0: back);
// H = number of cyclic lane bits (they are the low bits of the lane id): 6 wave (|D| <= 1), 4 row of 16, 0 = the whole
// sequence is in this lane
template <int H, int D, typename V> __device__ __forceinline__ V lat_nbr(V v)
{
    if constexpr (H == 0 || D == 0) return v;
    else if constexpr (H == 6) {
        static_assert(D == 1 || D == -1, "wave rotations move one lane");
        return lat_dpp<(D > 0 ? 0x134 : 0x13C)>(v);                           // wave_rol:1 / wave_ror:1
    } else {
        static_assert(H == 4 && D > -16 && D < 16, "row rotations");
        return lat_dpp<0x120 + ((16 - D) & 15)>(v);                           // row_ror:n: lane i takes lane i - n (mod 16)
    }
}

// one packet level on register-index bit K (2^K interleaved sequences of 32 >> K pairs per lane)
template <int K, int H, int NS, bool INV, typename V, typename C> __device__ __forceinline__ void lat_level(V (&x)[64], const C &cf)
{
    typedef typename lat_vtraits<V>::coef CF;
    constexpr int NSEQ = 1 << K, M = 32 >> K, S = 1 << K;
    auto U = [](int s, int m) { return s + ((2 * m) << K); };
    // odd channel: pair m takes the value of pair m + SH of the periodic sequence (SH < 0: delay); the pairs that come
    // from another lane's chunk are one DPP move each, all independent
    auto shift = [&](auto SHc) {
        constexpr int SH = decltype(SHc)::value;
        if constexpr (SH != 0) {
#pragma unroll
            for (int s = 0; s < NSEQ; ++s) {
                V old[M];
#pragma unroll
                for (int m = 0; m < M; ++m) old[m] = x[U(s, m) + S];
                lat_for<M>([&](auto Mc) {
                    constexpr int m = Mc;
                    constexpr int g = m + SH;                                  // source pair in sequence order
                    constexpr int d = (g >= 0) ? g / M : -((-g + M - 1) / M);   // floor(g / M): chunks away
                    constexpr int src = g - d * M;
                    x[U(s, m) + S] = lat_nbr<H, d>(old[src]);
                });
            }
        }
    };
    constexpr bool one_shot = (H != 6) || (NS - 1 <= M);                       // wave rotations reach one lane only
    if constexpr (!INV) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const CF pj = (CF)cf.p[j], kj = (CF)cf.kap[j];
#pragma unroll
            for (int s = 0; s < NSEQ; ++s)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    x[U(s, m)] = lat_fma(pj, x[U(s, m) + S], x[U(s, m)]);
                    x[U(s, m) + S] = lat_fma(-kj, x[U(s, m)], x[U(s, m) + S]);
                }
            if (j + 1 < NS) shift(std::integral_constant<int, 1>{});
        }
        if constexpr (one_shot) shift(std::integral_constant<int, -(NS - 1)>{});
        else {
#pragma unroll
            for (int j = 0; j + 1 < NS; ++j) shift(std::integral_constant<int, -1>{});
        }
    } else {
        if constexpr (one_shot) shift(std::integral_constant<int, NS - 1>{});
        else {
#pragma unroll
            for (int j = 0; j + 1 < NS; ++j) shift(std::integral_constant<int, 1>{});
        }
#pragma unroll
        for (int j = NS - 1; j >= 0; --j) {
            const CF pj = (CF)cf.p[j], kj = (CF)cf.kap[j];
#pragma unroll
            for (int s = 0; s < NSEQ; ++s)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    x[U(s, m) + S] = lat_fma(kj, x[U(s, m)], x[U(s, m) + S]);
                    x[U(s, m)] = lat_fma(-pj, x[U(s, m) + S], x[U(s, m)]);
                }
            if (j > 0) shift(std::integral_constant<int, -1>{});
        }
    }
}

__device__ __forceinline__ int lat_rev6(int v) { return (int)(__builtin_bitreverse32((unsigned)v) >> 26); }

template <int B0, typename A> __device__ __forceinline__ void lat_wait16(A &x)
{
    // x[B0 .. B0+15] are the destinations of the 16 reads just issued
    lat_wait8(x[B0], x[B0 + 1], x[B0 + 2], x[B0 + 3], x[B0 + 4], x[B0 + 5], x[B0 + 6], x[B0 + 7]);
    lat_wait8(x[B0 + 8], x[B0 + 9], x[B0 + 10], x[B0 + 11], x[B0 + 12], x[B0 + 13], x[B0 + 14], x[B0 + 15]);
}

// element e of a lane's 64-sample output chunk held by C register r, depth L (6 <= L <= 12):
// e[k] = r[L-6+k] for k < 12-L, e[12-L+j] = r[L-7-j] for j <= L-7
constexpr int lat_pi(int L, int r)
{
    int e = 0;
    for (int k = 0; k < 12 - L; ++k) e |= ((r >> (L - 6 + k)) & 1) << k;
    for (int j = 0; j <= L - 7; ++j) e |= ((r >> (L - 7 - j)) & 1) << (12 - L + j);
    return e;
}
constexpr int lat_pi_inv(int L, int e)
{
    for (int r = 0; r < 64; ++r)
        if (lat_pi(L, r) == e) return r;
    return -1;
}

// f[m] = g0 * g2^(popcount(lane) + m): the gain of a leaf whose path has popcount(lane) detail branches in the lane bits
// (levels 1-6) and m in the register bits (levels 7-L)
__device__ __forceinline__ void lat_gains(double (&f)[7], int lane, const WxLat &cf)
{
    double b = cf.g0;
#pragma unroll
    for (int k = 0; k < 6; ++k) b = ((lane >> k) & 1) ? b * cf.g2 : b;
    f[0] = b;
#pragma unroll
    for (int m = 1; m < 7; ++m) f[m] = f[m - 1] * cf.g2;
}
constexpr int lat_pc(int v)
{
    int c = 0;
    for (; v; v >>= 1) c += v & 1;
    return c;
}

// C -> S exchange + stores for a compile-time depth (exchange T4 of tools/lattice_lds_maps.py)
template <int L, typename TM> __device__ __forceinline__ void lat_store_c(double (&c)[64], unsigned lds0, TM *__restrict__ ys,
                                                             int lane, const WxLat &cf)
{
    double gf[7];
    lat_gains(gf, lane, cf);
    const int ch = lat_rev6(lane);
    const int c0 = ch & 1, c1 = (ch >> 1) & 1, c2 = (ch >> 2) & 1;
    const int wrow = (c0 ^ c2) | ((ch >> 3) << 1) | (c1 << 4) | (c2 << 5);
    const unsigned wa = lds0 + 8u * 17u * wrow;
    const int q = lane >> 3;                              // chunk within the instruction's group of 8
    const int q0 = q & 1, q1 = (q >> 1) & 1, q2 = q >> 2;
    const unsigned ra = lds0 + 8u * (17u * ((q0 ^ q2) + 16 * q1 + 32 * q2) + 2u * (lane & 7));
    const unsigned yo = 64u * q + 2u * (lane & 7);       // lane part of the store address (elements)
    lat_for<4>([&](auto K) {
        constexpr int k = K;
        lat_for<16>([&](auto E) {
            constexpr int e4 = E;
            constexpr int rr = lat_pi_inv(L, 16 * k + e4);
            lds_wr<8 * e4>(wa, c[rr] * gf[lat_pc(rr & ((1 << (L - 6)) - 1))]);
        });
        lat_for<2>([&](auto HH) {
            constexpr int hh = HH;
            double v[8];
            lat_for<4>([&](auto I) {
                constexpr int i = 4 * hh + I;
                v[2 * I] = lds_rd<8 * (34 * i)>(ra);
                v[2 * I + 1] = lds_rd<8 * (34 * i + 1)>(ra);
            });
            lat_wait8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            lat_for<4>([&](auto I) {
                constexpr int i = 4 * hh + I;
                lat_d2 o;
                o.x = v[2 * I];
                o.y = v[2 * I + 1];
                lat_st2(lat_sbase(ys + 512 * i + 16 * k) + yo, o);
            });
        });
    });
}

// loads + S -> C exchange (T4i)
template <int L, typename TM> __device__ __forceinline__ void lat_load_c(double (&c)[64], unsigned lds0, const TM *__restrict__ xs,
                                                            int lane, const WxLat &cf)
{
    double gf[7];
    lat_gains(gf, lane, cf);
    const int ch = lat_rev6(lane);
    const int rrow = (((ch >> 1) ^ ch) & 1) | ((ch >> 2) << 1) | ((ch & 1) << 5);
    const unsigned ra = lds0 + 8u * 17u * rrow;
    const int lanepart = (((lane >> 4) ^ (lane >> 3)) & 1) | (((lane >> 5) & 1) << 1) | (((lane >> 3) & 1) << 5);
    const unsigned wa = lds0 + 8u * (17u * lanepart + 2u * (lane & 7));
    const unsigned xo = 64u * (lane >> 3) + 2u * (lane & 7);
    lat_d2 v[8];
    lat_for<8>([&](auto I) {
        constexpr int i = I;
        v[i] = lat_ld2(lat_sbase(xs + 512 * i) + xo);
    });
    lat_for<4>([&](auto K) {
        constexpr int k = K;
        lat_for<8>([&](auto I) {
            constexpr int i = I;
            lds_wr<8 * (68 * i)>(wa, v[i].x);
            lds_wr<8 * (68 * i + 1)>(wa, v[i].y);
        });
        // the next round's loads go out as soon as their registers are free (the asm statements are memory barriers
        // for the compiler, so this order is kept)
        if constexpr (k < 3)
            lat_for<8>([&](auto I) {
                constexpr int i = I;
                v[i] = lat_ld2(lat_sbase(xs + 512 * i + 16 * (k + 1)) + xo);
            });
        lat_for<16>([&](auto E) {
            constexpr int e4 = E;
            c[lat_pi_inv(L, 16 * k + e4)] = lds_rd<8 * e4>(ra);
        });
        lat_for<2>([&](auto G) {
            constexpr int g = G;
            lat_wait8(c[lat_pi_inv(L, 16 * k + 8 * g)], c[lat_pi_inv(L, 16 * k + 8 * g + 1)], c[lat_pi_inv(L, 16 * k + 8 * g + 2)],
                      c[lat_pi_inv(L, 16 * k + 8 * g + 3)], c[lat_pi_inv(L, 16 * k + 8 * g + 4)], c[lat_pi_inv(L, 16 * k + 8 * g + 5)],
                      c[lat_pi_inv(L, 16 * k + 8 * g + 6)], c[lat_pi_inv(L, 16 * k + 8 * g + 7)]);
        });
        lat_for<16>([&](auto E) {
            constexpr int rr = lat_pi_inv(L, 16 * k + E);
            c[rr] *= gf[lat_pc(rr & ((1 << (L - 6)) - 1))];
        });
    });
}

// T1: L0 -> A (reg p[5:0], lane p[11:6]), round f = p[5:4]; and its inverse T1i (A -> L0)
__device__ __forceinline__ void lat_t1(lat_d2 (&r)[32], double (&a)[64], unsigned lds0, int lane)
{
    const unsigned wa = lds0 + 8u * (17u * (lane >> 3) + 2u * (lane & 7)), ra = lds0 + 8u * 17u * lane;
    lat_for<4>([&](auto Fq) {
        constexpr int f = Fq;
        lat_for<8>([&](auto Hq) {
            constexpr int hi3 = Hq;
            lds_wr<8 * (136 * hi3)>(wa, r[4 * hi3 + f].x);
            lds_wr<8 * (136 * hi3 + 1)>(wa, r[4 * hi3 + f].y);
        });
        double t[16];
        lat_for<16>([&](auto M) {
            constexpr int m = M;
            t[m] = lds_rd<8 * m>(ra);
        });
        lat_wait16<0>(t);
        lat_for<16>([&](auto M) {
            constexpr int m = M;
            a[16 * f + m] = t[m];
        });
    });
}
template <typename SINK> __device__ __forceinline__ void lat_t1i(double (&a)[64], unsigned lds0, int lane, SINK &&sink)
{
    const unsigned wa = lds0 + 8u * 17u * lane, ra = lds0 + 8u * (17u * (lane >> 3) + 4u * (lane & 7));
    lat_for<4>([&](auto Fq) {
        constexpr int f = Fq;
        lat_for<16>([&](auto M) {
            constexpr int m = M;                     // m = 2 j + e
            lds_wr<8 * (4 * (m >> 1) + 2 * (m & 1))>(wa, a[16 * f + m]);
        });
        double t[16];
        lat_for<16>([&](auto M) {
            constexpr int hi3 = M / 2, e = M % 2;
            t[M] = lds_rd<8 * (136 * hi3 + 2 * e)>(ra);
        });
        lat_wait16<0>(t);
        lat_d2 o[8];
        lat_for<8>([&](auto Hq) {
            constexpr int hi3 = Hq;
            o[hi3].x = t[2 * hi3];
            o[hi3].y = t[2 * hi3 + 1];
        });
        sink(Fq, o);
    });
}

// ---------------------------------------------------------------- forward
// the transform of one 4096-sample signal from its samples in the L0 arrangement (register 4 hi3 + f of a lane = the two samples
// p, p + 1 with p[11:9] = hi3, p[8:6] = lane >> 3, p[5:4] = f, p[3:1] = lane & 7: what eight complete 128-byte lines per load give),
// shared by the kernel that loads them (k_lat_wpt_f64) and the one that computes them as a child of an 8192-sample signal
// (k_lat_wpt8k_f64)
template <int NS, typename TM>
__device__ __forceinline__ void lat_fwd_from_l0(lat_d2 (&r)[32], unsigned lds0, int lane, int L, const WxLat &cf, TM *__restrict__ ys)
{
    double a[64];
    lat_t1(r, a, lds0, lane);
    lat_level<0, 6, NS, false>(a, cf);
    lat_level<1, 6, NS, false>(a, cf);
    // T2: A -> B (reg p[7:2], lane mu = p[11:8] | p[1:0] << 4)
    double bb[64];
    {
        const int sw = lane ^ ((lane >> 5) << 1);
        const unsigned wa0 = lds0 + 8u * sw, wa1 = lds0 + 8u * (sw ^ 1);
        const int H = lane & 15, p10 = lane >> 4;
        const int lam0 = 4 * H, sg = (p10 & 1) | ((lam0 >> 5) << 1);
        unsigned ra[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) ra[h] = lds0 + 8u * (64 * p10 + ((lam0 + h) ^ sg));
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                lds_wr<8 * 64 * j>((j & 1) ? wa1 : wa0, a[16 * f + j]);
            });
            double t[16];
            lat_for<16>([&](auto Q) {
                constexpr int h = Q / 4, g = Q % 4;
                t[Q] = lds_rd<8 * 256 * g>(ra[h]);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto Q) {
                constexpr int h = Q / 4, g = Q % 4;
                bb[16 * h + 4 * f + g] = t[Q];
            });
        });
    }
    lat_level<0, 4, NS, false>(bb, cf);
    lat_level<1, 4, NS, false>(bb, cf);
    lat_level<2, 4, NS, false>(bb, cf);
    lat_level<3, 4, NS, false>(bb, cf);
    // T3: B -> C (reg p[11:6], lane nu = p[5:0])
    double c[64];
    {
        const unsigned wa = lds0 + 8u * (lane + (lane >> 5));
        const unsigned ra = lds0 + 8u * (66 * (lane >> 2) + 16 * (lane & 1) + 33 * ((lane >> 1) & 1));
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                lds_wr<8 * 66 * j>(wa, bb[16 * f + j]);
            });
            double t[16];
            lat_for<16>([&](auto Hq) {
                constexpr int H = Hq;
                t[H] = lds_rd<8 * H>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto Hq) {
                constexpr int H = Hq;
                c[4 * H + f] = t[H];
            });
        });
    }
    if (L > 6) lat_level<0, 0, NS, false>(c, cf);
    if (L > 7) lat_level<1, 0, NS, false>(c, cf);
    if (L > 8) lat_level<2, 0, NS, false>(c, cf);
    if (L > 9) lat_level<3, 0, NS, false>(c, cf);
    if (L > 10) lat_level<4, 0, NS, false>(c, cf);
    if (L > 11) lat_level<5, 0, NS, false>(c, cf);
    switch (L) {
    case 6: lat_store_c<6>(c, lds0, ys, lane, cf); break;
    case 7: lat_store_c<7>(c, lds0, ys, lane, cf); break;
    case 8: lat_store_c<8>(c, lds0, ys, lane, cf); break;
    case 9: lat_store_c<9>(c, lds0, ys, lane, cf); break;
    case 10: lat_store_c<10>(c, lds0, ys, lane, cf); break;
    case 11: lat_store_c<11>(c, lds0, ys, lane, cf); break;
    default: lat_store_c<12>(c, lds0, ys, lane, cf); break;
    }
}

template <int NS, int WPE, typename TM = double>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpt_f64(
    const TM *__restrict__ x, TM *__restrict__ y, int L, int64_t batch, WxLat cf)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    const int64_t sig = blockIdx.x;
    const TM *xs = x + sig * 4096;
    // L0: instruction (hi3 = p[11:9], f = p[5:4]) loads eight complete 128-byte lines: lane holds p[8:6] = lane >> 3,
    // p[3:1] = lane & 7, register e = p[0]
    lat_d2 r[32];
    const unsigned xo = 64u * (lane >> 3) + 2u * (lane & 7);
    lat_for<32>([&](auto Q) {
        constexpr int hi3 = Q / 4, f = Q % 4;
        r[Q] = lat_ld2(lat_sbase(xs + 512 * hi3 + 16 * f) + xo);
    });
    lat_fwd_from_l0<NS>(r, lds0, lane, L, cf, y + sig * 4096);
}

// ---------------------------------------------------------------- wpd: every level leaves through an LDS transposition
// After level l the registers hold level l of the packet table in the in-place order: the coefficient at sample index p
// belongs at position  o = bitreverse_l(p[l-1:0]) << (12 - l) | p >> l  of column l (DWT.jl:145-158: node j of depth l =
// rows [j n/2^l, (j+1) n/2^l)), scaled by g^(l - 2 popcount(p[l-1:0])).  o is a bit permutation of p, so "which register
// of which lane" -> "which byte of which 128-byte line" is static routing: p bit t sits in a register or lane bit that
// depends on the layout (A, B, C) and lands on o bit (t < l ? 11 - t : t - l).  An exchange round moves 16 registers per
// lane (two register bits that land on line-address bits are fixed per round): slot = 17 * line + position, where the six
// line-address bits of the round are ordered with the low lane bits first (conflict-free ds_write_b64 for every
// layout and level, 2-way ds_read_b64; enumerated in tools/lattice_emu.py::emit_plan and checked against the oracle's
// wpd there); the read side hands every group of 8 lanes one complete line for a 16-byte-per-lane store.
struct LatSrc { int reg; int bit; };                       // reg = 1: register-index bit, 0: lane-id bit
constexpr LatSrc lat_src(int lay, int t)
{
    if (lay == 0) return t < 6 ? LatSrc{1, t} : LatSrc{0, t - 6};
    if (lay == 2) return (t >= 2 && t < 8) ? LatSrc{1, t - 2} : (t >= 8 ? LatSrc{0, t - 8} : LatSrc{0, 4 + t});
    return t >= 6 ? LatSrc{1, t - 6} : LatSrc{0, t};
}
constexpr int lat_pbit(int lay, int reg, int bit)           // sample-index bit held by that register / lane bit
{
    for (int t = 0; t < 12; ++t)
        if (lat_src(lay, t).reg == reg && lat_src(lay, t).bit == bit) return t;
    return -1;
}
// level code lc = l + 16 sh: sh low index bits are a signal number (2^sh signals of 4096 >> sh samples interleaved in the
// registers, see k_lat_wpt_f64<.., SH>), the l levels acted on bits sh .. sh + l - 1: the signal number goes to the top of the
// address, the path bits are reversed below it, the rest is the position inside the node
// lc = 256 + l (wx_lattice_2d64.h): the 4096 slots are a 64 x 64 IMAGE (column-major: bits 0..5 the row, 6..11 the column) after l levels
// down the columns (row bits 0 .. l-1) and l levels along the rows (column bits 0 .. l-1): each half of the address is routed like a
// 64-sample signal of its own (path bits reversed on top of the position inside the node), 2 l gains on the way
// + 1024 (LAT_ORD32, set by lat_emit / lat_absorb for 4-byte elements): another choice of the round bits, see lat_round_bit
constexpr int LAT_ORD32 = 1024;
constexpr bool lat_is2d(int lc) { return (lc & 1023) >= 256 && (lc & 1023) < 512; }
// lc = 512 + l + 16 sh (row pass of an image, k_lat_rows_g_f64): the 2^sh signals of the wavefront are adjacent ROWS -- the signal
// number is the CONTIGUOUS dimension of the memory (address = o[sh-1:0] + (o >> sh) * column stride), the routed position of the
// packet order lies above it
constexpr bool lat_isT(int lc) { return (lc & 1023) >= 512; }
constexpr int lat_lv(int lc) { return lat_is2d(lc) ? 2 * (lc & 15) : (lc & 15); }
constexpr int lat_sh(int lc) { return lat_is2d(lc) ? 0 : ((lc & 255) >> 4); }
constexpr int lat_sb(int lc) { return lat_isT(lc) ? lat_sh(lc) : 12 - lat_sh(lc); }   // address = o[sb-1:0] + (o >> sb) * sstride
constexpr bool lat_on_path(int lc, int t)                  // t = -1: old code counted it; never occurs for k, i < 6
{
    return lat_is2d(lc) ? (t >= 0 && t % 6 < (lc & 15)) : (t >= lat_sh(lc) && t < lat_sh(lc) + lat_lv(lc));
}
constexpr int lat_obit(int lc, int t)
{
    if (t < 0) return 12;                                    // "no such bit" (lat_pbit / lat_round_bit = -1): above the address
    if (lat_is2d(lc)) return 6 * (t / 6) + (t % 6 < (lc & 15) ? 5 - t % 6 : t % 6 - (lc & 15));
    if (lat_isT(lc)) return t < lat_sh(lc) ? t : (t < lat_sh(lc) + lat_lv(lc) ? 11 - t + lat_sh(lc) : t - lat_lv(lc));
    return t < lat_sh(lc) ? 12 - lat_sh(lc) + t : (t < lat_sh(lc) + lat_lv(lc) ? 11 - t : t - lat_sh(lc) - lat_lv(lc));
}
constexpr int lat_reg_o(int lay, int l, int i) { return lat_obit(l, lat_pbit(lay, 1, i)); }
constexpr int lat_lane_o(int lay, int l, int k) { return lat_obit(l, lat_pbit(lay, 0, k)); }
// the j-th (j = 0, 1) register bit fixed per round: the lowest register bits that land on a line-address bit.
// LAT_ORD32 (4-byte elements: a 16-element line is HALF a 128-byte memory line): the register bits that land on the HIGHEST address bits
// instead, so that address bit 4 -- the two halves of a memory line -- is not a round bit when there is a choice: the halves are then moved
// by the same or by neighbouring instructions instead of by different rounds.  (PMC, 131072 x 4096 Float32, L = 10: the pair kernels
// fetched 1.67 x and wrote 1.21 x their bytes, profiles/r05_target_f32.md -- every line came or went as two halves a round apart.)
constexpr int lat_round_bit(int lay, int l, int j)
{
    if (l & LAT_ORD32) {
        int b0 = -1, b1 = -1;                                // b0: the largest o, b1: the second largest
        for (int i = 0; i < 6; ++i) {
            const int o = lat_reg_o(lay, l, i);
            if (o < 4) continue;
            if (b0 < 0 || o > lat_reg_o(lay, l, b0)) { b1 = b0; b0 = i; }
            else if (b1 < 0 || o > lat_reg_o(lay, l, b1)) b1 = i;
        }
        const int lo = b0 < b1 ? b0 : b1, hi = b0 < b1 ? b1 : b0;
        return j == 0 ? lo : hi;
    }
    int c = 0;
    for (int i = 0; i < 6; ++i)
        if (lat_reg_o(lay, l, i) >= 4) { if (c == j) return i; ++c; }
    return -1;
}
constexpr bool lat_is_round_bit(int lay, int l, int i) { return i == lat_round_bit(lay, l, 0) || i == lat_round_bit(lay, l, 1); }
constexpr int lat_vbit(int lay, int l, int j)               // j-th of the four register bits that vary inside a round
{
    int c = 0;
    for (int i = 0; i < 6; ++i)
        if (!lat_is_round_bit(lay, l, i)) { if (c == j) return i; ++c; }
    return -1;
}
// the q-th line-address bit of a round: {is_reg, source bit, o bit}; low lane bits first, then lane bits 4, 5, then registers
struct LatLine { int reg; int bit; int ob; };
constexpr LatLine lat_line(int lay, int l, int q)
{
    int c = 0;
    for (int k = 0; k < 6; ++k)
        if (lat_lane_o(lay, l, k) >= 4) { if (c == q) return LatLine{0, k, lat_lane_o(lay, l, k)}; ++c; }
    for (int j = 0; j < 4; ++j) {
        const int i = lat_vbit(lay, l, j);
        if (lat_reg_o(lay, l, i) >= 4) { if (c == q) return LatLine{1, i, lat_reg_o(lay, l, i)}; ++c; }
    }
    return LatLine{-1, -1, -1};
}
constexpr int lat_emit_reg(int lay, int l, int rho, int v)  // register of round rho, v = 0..15
{
    int r = 0;
    for (int j = 0; j < 4; ++j) r |= ((v >> j) & 1) << lat_vbit(lay, l, j);
    for (int j = 0; j < 2; ++j) r |= ((rho >> j) & 1) << lat_round_bit(lay, l, j);
    return r;
}
constexpr int lat_emit_slot_reg(int lay, int l, int r)      // register part of the LDS slot (elements)
{
    int hi = 0, pos = 0;
    for (int i = 0; i < 6; ++i)
        if (lat_reg_o(lay, l, i) < 4) pos |= ((r >> i) & 1) << lat_reg_o(lay, l, i);
    for (int q = 0; q < 6; ++q)
        if (lat_line(lay, l, q).reg == 1) hi |= ((r >> lat_line(lay, l, q).bit) & 1) << q;
    return 17 * hi + pos;
}
constexpr int lat_emit_pc_reg(int lay, int l, int r)        // detail branches on the path held in register bits
{
    int c = 0;
    for (int i = 0; i < 6; ++i)
        if (lat_on_path(l, lat_pbit(lay, 1, i))) c += (r >> i) & 1;
    return c;
}
constexpr int lat_emit_o_round(int lay, int l, int rho)
{
    int o = 0;
    for (int j = 0; j < 2; ++j) o |= ((rho >> j) & 1) << lat_reg_o(lay, l, lat_round_bit(lay, l, j));
    return o;
}
constexpr int lat_emit_o_instr(int lay, int l, int i)       // line-address bits 3..5 of the round come from the store index
{
    int o = 0;
    for (int q = 3; q < 6; ++q) o |= ((i >> (q - 3)) & 1) << lat_line(lay, l, q).ob;
    return o;
}

struct WxLatW {
    WxLat c;
    double gl[13];            // g^l
    int tail_bsig;            // pair kernels (lat_f2v): signals between the two signal sets of the LAST wavefront (see lat_pair_sig)
};
// Pair kernels: wavefront w takes the signal sets [w per, w per + half) and [w per + half, (w + 1) per), per = 2 half = 2^(SH+1).  The
// last wavefront of a batch that is not a multiple of per starts at `tail_sig` and its second set follows `tail_bsig` signals after the
// first: a remainder r >= half gives tail_sig = batch - r, tail_bsig = r - half (the sets overlap INSIDE the wavefront, whose loads all
// precede its stores: valid in place, for SH = 0 the lone last signal is simply both halves of the pair); r < half re-does the last per
// signals (tail_sig = batch - per: out of place only).  lat_pair_plan fills the launch; false = not applicable.
struct WxPairPlan { unsigned nwave; int tail_sig; int tail_bsig; };
static inline bool wx_lat_pair_plan(int64_t batch, int SH, bool in_place, WxPairPlan *pp)
{
    const int64_t half = (int64_t)1 << SH, per = 2 * half;
    if (batch < half || batch > 0x7fffffff) return false;
    int64_t r = batch % per;
    if (r == 0) r = per;
    if (batch < per) {                                       // one wavefront: the second set ends with the batch
        pp->nwave = 1; pp->tail_sig = 0; pp->tail_bsig = (int)(batch - half);
        return true;
    }
    pp->nwave = (unsigned)((batch + per - 1) / per);
    if (r >= half) { pp->tail_sig = (int)(batch - r); pp->tail_bsig = (int)(r - half); return true; }
    if (in_place) return false;
    pp->tail_sig = (int)(batch - per); pp->tail_bsig = (int)half;
    return true;
}

// PRED (tree-driven transforms, k_lat_wpt_tree_f64): only the lines of the nodes that are LEAVES at this depth are stored --
// bit 8 rho + i of `word` says whether this lane's 16 bytes of store instruction i of round rho belong to one (a table made
// by k_lat_tree_prep with the same routing functions), `anyw` is the OR of the words over the lanes (wave-uniform: rounds and
// half-rounds without a leaf line are skipped, exchange included).
template <int LAY, int LVL, bool PRED = false, typename IO = double, typename V = double>
__device__ __forceinline__ void lat_emit(V (&x)[64], unsigned lds0, IO *__restrict__ ycol, int lane, const WxLatW &cw,
                                         unsigned sstride = 4096u >> lat_sh(LVL), unsigned word = 0, unsigned anyw = 0,
                                         unsigned bofs = 0xffffffffu)
{
    constexpr int LV = (sizeof(IO) == 4 && !PRED) ? (LVL | LAT_ORD32) : LVL;   // 4-byte elements: the round bits of lat_round_bit's second rule
    typedef typename lat_vtraits<V>::coef CF;
    // V = lat_f2v: the second signal set of the wavefront follows the first at 2^sh signals' distance (bofs elements when given)
    const unsigned boff = bofs != 0xffffffffu ? bofs : (sstride << lat_sh(LV));
    // sstride: elements between the columns of consecutive signals of the wavefront (interleaved kernels; the signal
    // number is the top lat_sh(LV) bits of the routed address)
    constexpr int SB = lat_sb(LV);
    // lane parts: line-address bits, in-line position bits, detail branches of the path
    int hi_lane = 0, pos_lane = 0;
    double b = cw.gl[lat_lv(LV)];
    lat_for<6>([&](auto Kc) {
        constexpr int k = Kc;
        constexpr int ob = lat_lane_o(LAY, LV, k);
        if constexpr (ob < 4) pos_lane |= ((lane >> k) & 1) << ob;
        if constexpr (lat_on_path(LV, lat_pbit(LAY, 0, k))) b = ((lane >> k) & 1) ? b * cw.c.g2 : b;
    });
    lat_for<6>([&](auto Qc) {
        constexpr int q = Qc;
        constexpr LatLine ln = lat_line(LAY, LV, q);
        if constexpr (ln.reg == 0) hi_lane |= ((lane >> ln.bit) & 1) << q;
    });
    double gfd[7];
    gfd[0] = b;
#pragma unroll
    for (int m = 1; m < 7; ++m) gfd[m] = gfd[m - 1] * cw.c.g2;
    CF gf[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) gf[m] = (CF)gfd[m];
    const unsigned wa = lds0 + 8u * (unsigned)(17 * hi_lane + pos_lane);
    // read side: store instruction i of a round covers lines 8 i + (lane >> 3), a lane takes elements 2 (lane & 7), +1
    const int qq = lane >> 3;
    int o_lane = 2 * (lane & 7);
    lat_for<3>([&](auto Qc) {
        constexpr int q = Qc;
        constexpr int ob = lat_line(LAY, LV, q).ob;           // forced constant evaluation: left to the optimiser the
        o_lane |= ((qq >> q) & 1) << ob;                          // bit-map loops are not always folded
    });
    const unsigned ra = lds0 + 8u * (unsigned)(17 * qq + 2 * (lane & 7));
    const unsigned yo = (unsigned)(o_lane & ((1 << SB) - 1)) + (unsigned)(o_lane >> SB) * sstride;
    lat_for<4>([&](auto Rc) {
        constexpr int rho = Rc;
        if (PRED && ((anyw >> (8 * rho)) & 0xffu) == 0) return;
        lat_for<16>([&](auto Vc) {
            constexpr int r = lat_emit_reg(LAY, LV, rho, Vc);
            constexpr int pc = lat_emit_pc_reg(LAY, LV, r);
            lds_wr<8 * lat_emit_slot_reg(LAY, LV, r)>(wa, lat_mul(x[r], gf[pc]));
        });
        lat_for<2>([&](auto HH) {
            constexpr int hh = HH;
            if (PRED && ((anyw >> (8 * rho + 4 * hh)) & 0xfu) == 0) return;
            V v[8];
            lat_for<4>([&](auto I) {
                constexpr int i = 4 * hh + I;
                v[2 * I] = lds_rd<8 * (17 * 8 * i), V>(ra);
                v[2 * I + 1] = lds_rd<8 * (17 * 8 * i + 1), V>(ra);
            });
            lat_wait8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            lat_for<4>([&](auto I) {
                constexpr int i = 4 * hh + I;
                constexpr int oc = lat_emit_o_round(LAY, LV, rho) + lat_emit_o_instr(LAY, LV, i);
                if constexpr (PRED) {
                    if ((word >> (8 * rho + i)) & 1u)
                        lat_stv(ycol + (oc & ((1 << SB) - 1)) + (size_t)(oc >> SB) * sstride, yo, boff, v[2 * I], v[2 * I + 1], false);
                } else
                    lat_stv(ycol + (oc & ((1 << SB) - 1)) + (size_t)(oc >> SB) * sstride, yo, boff, v[2 * I], v[2 * I + 1], true);
            });
        });
    });
}

// the mirror of lat_emit: a column in the order of lat_obit(LVL, .) -> the registers of layout LAY, scaled by the path gains
// (the two LDS address patterns swap roles: 16-byte loads of complete lines are written where lat_emit reads, the registers
// are read where lat_emit writes)
__device__ __forceinline__ void lat_st1(double __attribute__((address_space(1))) *p, double v) { *p = v; }
// PRED (k_lat_iwpt_tree_f64): only the lines of this depth's leaves are loaded (`word`, `anyw` as in lat_emit), everything
// else arrives as zeros, and the arrivals are ADDED to what the deeper levels have synthesised (see below).
// threshold of denoise() riding on the absorbed leaves (Denoising.jl:527 threshold!(x, th, t) before iwpt): positions
// [lo, n) of every signal, threshold tt[s] for signal s of the wavefront; kind < 0: none
struct LatThr {
    double tt[4];
    int kind, lo;
};

// the lines of rounds 0 and 1 of lat_absorb<LAY, LVL, true, true>, issued early (k_lat_iwpt_tree_f64 asks for them before it
// runs the synthesis level above, so that their latency hides behind that level's arithmetic; inside the absorb the loads
// run two rounds ahead of the exchange: one round ahead left every round waiting for memory)
template <int LAY, int LVL>
__device__ __forceinline__ unsigned lat_absorb_xo(int lane, unsigned sstride)
{
    constexpr int SB = lat_sb(LVL);
    const int qq = lane >> 3;
    int o_lane = 2 * (lane & 7);
    lat_for<3>([&](auto Qc) {
        constexpr int q = Qc;
        constexpr int ob = lat_line(LAY, LVL, q).ob;
        o_lane |= ((qq >> q) & 1) << ob;
    });
    return (unsigned)(o_lane & ((1 << SB) - 1)) + (unsigned)(o_lane >> SB) * sstride;
}
// dep / cstride (iwpd by tree, all depths >= 6 - SH in one absorb): the piece of store index 8 RN + i sits in the column of
// its leaf's depth -- nibble (8 RN + i) of the lane's four `dep` words -- cstride elements per column
template <int LAY, int LVL, int RN, typename IO = double, typename V = double>
__device__ __forceinline__ void lat_absorb_fetch(lat_v2<V> (&v)[16], const IO *__restrict__ xcol, unsigned xo, unsigned sstride, unsigned word,
                                                 const unsigned *dep = nullptr, unsigned cstride = 0)
{
    constexpr int SB = lat_sb(LVL);
    lat_for<8>([&](auto I) {
        constexpr int i = I;
        constexpr int oc = lat_emit_o_round(LAY, LVL, RN) + lat_emit_o_instr(LAY, LVL, i);
        constexpr int idx = 8 * RN + i;
        lat_v2<V> &d = v[8 * (RN & 1) + i];
        d.x = d.y = V{};
        const unsigned co = dep ? ((dep[idx >> 3] >> (4 * (idx & 7))) & 15u) * cstride : 0u;
        if ((word >> idx) & 1u) d = lat_ldv(xcol + (oc & ((1 << SB) - 1)) + (size_t)(oc >> SB) * sstride, xo + co, sstride << lat_sh(LVL), (V *)nullptr);
    });
}
template <int LAY, int LVL, typename IO = double, typename V = double>
__device__ __forceinline__ void lat_absorb_fetch01(lat_v2<V> (&v)[16], const IO *__restrict__ xcol, int lane, unsigned sstride, unsigned word,
                                                   const unsigned *dep = nullptr, unsigned cstride = 0)
{
    const unsigned xo = lat_absorb_xo<LAY, LVL>(lane, sstride);
    lat_absorb_fetch<LAY, LVL, 0>(v, xcol, xo, sstride, word, dep, cstride);
    lat_absorb_fetch<LAY, LVL, 1>(v, xcol, xo, sstride, word, dep, cstride);
}

template <int LAY, int LVL, bool PRED = false, bool PRE = false, typename IO = double, typename V = double>
__device__ __forceinline__ void lat_absorb(V (&x)[64], unsigned lds0, const IO *__restrict__ xcol, int lane, const WxLatW &cw,
                                           unsigned sstride, unsigned word, unsigned anyw, lat_v2<V> (&v)[16],
                                           const unsigned *dep = nullptr, unsigned cstride = 0, const LatThr *th = nullptr,
                                           unsigned bofs = 0xffffffffu)
{
    constexpr int LV = (sizeof(IO) == 4 && !PRED) ? (LVL | LAT_ORD32) : LVL;   // as in lat_emit
    typedef typename lat_vtraits<V>::coef CF;
    const unsigned boff = bofs != 0xffffffffu ? bofs : (sstride << lat_sh(LV));   // V = lat_f2v: where the second signal set starts
    static_assert(PRE == PRED, "prefetched lines come with the predicated form");
    constexpr int SB = lat_sb(LV);
    int hi_lane = 0, pos_lane = 0;
    double b = cw.gl[lat_lv(LV)];
    lat_for<6>([&](auto Kc) {
        constexpr int k = Kc;
        constexpr int ob = lat_lane_o(LAY, LV, k);
        if constexpr (ob < 4) pos_lane |= ((lane >> k) & 1) << ob;
        if constexpr (lat_on_path(LV, lat_pbit(LAY, 0, k))) b = ((lane >> k) & 1) ? b * cw.c.g2 : b;
    });
    lat_for<6>([&](auto Qc) {
        constexpr int q = Qc;
        constexpr LatLine ln = lat_line(LAY, LV, q);
        if constexpr (ln.reg == 0) hi_lane |= ((lane >> ln.bit) & 1) << q;
    });
    double gfd[7];
    gfd[0] = b;
#pragma unroll
    for (int m = 1; m < 7; ++m) gfd[m] = gfd[m - 1] * cw.c.g2;
    CF gf[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) gf[m] = (CF)gfd[m];
    const unsigned rda = lds0 + 8u * (unsigned)(17 * hi_lane + pos_lane);
    const int qq = lane >> 3;
    int o_lane = 2 * (lane & 7);
    lat_for<3>([&](auto Qc) {
        constexpr int q = Qc;
        constexpr int ob = lat_line(LAY, LV, q).ob;
        o_lane |= ((qq >> q) & 1) << ob;
    });
    const unsigned wra = lds0 + 8u * (unsigned)(17 * qq + 2 * (lane & 7));
    const unsigned xo = (unsigned)(o_lane & ((1 << SB) - 1)) + (unsigned)(o_lane >> SB) * sstride;
    auto fetch = [&](auto Rn) {
        constexpr int rn = Rn;
        if constexpr (PRED) lat_absorb_fetch<LAY, LV, rn>(v, xcol, xo, sstride, word, dep, cstride);
        else
            lat_for<8>([&](auto I) {
                constexpr int i = I;
                constexpr int oc = lat_emit_o_round(LAY, LV, rn) + lat_emit_o_instr(LAY, LV, i);
                v[8 * (rn & 1) + i] = lat_ldv(xcol + (oc & ((1 << SB) - 1)) + (size_t)(oc >> SB) * sstride, xo, boff, (V *)nullptr);
            });
    };
    if constexpr (!PRE) fetch(std::integral_constant<int, 0>{});
    lat_for<4>([&](auto Rc) {
        constexpr int rho = Rc;
        const bool live = !PRED || ((anyw >> (8 * rho)) & 0xffu) != 0;       // wave-uniform
        if (live)
            lat_for<8>([&](auto I) {
                constexpr int i = I;
                lat_v2<V> &d = v[8 * (rho & 1) + i];
                if constexpr (PRED && std::is_same<V, double>::value) {
                    if (th && th->kind >= 0) {
                        constexpr int oc = lat_emit_o_round(LAY, LV, rho) + lat_emit_o_instr(LAY, LV, i);
                        const int o = o_lane | oc, pos = o & ((1 << SB) - 1), sg = o >> SB;
                        const double tt = sg == 0 ? th->tt[0] : (sg == 1 ? th->tt[1] : (sg == 2 ? th->tt[2] : th->tt[3]));
                        if ((word >> (8 * rho + i)) & 1u) {
                            if (pos >= th->lo) d.x = wx_thresh<double>(d.x, tt, th->kind);
                            if (pos + 1 >= th->lo) d.y = wx_thresh<double>(d.y, tt, th->kind);
                        }
                    }
                }
                lds_wr<8 * (17 * 8 * i)>(wra, d.x);
                lds_wr<8 * (17 * 8 * i + 1)>(wra, d.y);
            });
        // the next lines travel while this round is exchanged: one round ahead (plain), two rounds ahead (tree-driven)
        if constexpr (!PRE && rho < 3) fetch(std::integral_constant<int, rho + 1>{});
        if constexpr (PRE && rho < 2) fetch(std::integral_constant<int, rho + 2>{});
        if (!live) return;
        V t[16];
        lat_for<16>([&](auto Vc) {
            constexpr int r = lat_emit_reg(LAY, LV, rho, Vc);
            t[Vc] = lds_rd<8 * lat_emit_slot_reg(LAY, LV, r), V>(rda);
        });
        lat_wait16<0>(t);
        lat_for<16>([&](auto Vc) {
            constexpr int r = lat_emit_reg(LAY, LV, rho, Vc);
            constexpr int pc = lat_emit_pc_reg(LAY, LV, r);
            // PRED: the transform is linear and a register that holds a leaf of this depth holds an exact zero so far (its
            // descendants do not exist: zeros in, zeros out), while every loaded value that is not a leaf's is a zero too
            // (whole 16-byte pieces are predicated): adding is selecting
            if constexpr (PRED) x[r] = lat_fma(gf[pc], t[Vc], x[r]);
            else x[r] = lat_mul(t[Vc], gf[pc]);
        });
    });
}

template <int LAY, int LVL, bool PRED = false, typename IO = double, typename V = double>
__device__ __forceinline__ void lat_absorb(V (&x)[64], unsigned lds0, const IO *__restrict__ xcol, int lane, const WxLatW &cw,
                                           unsigned sstride = 4096u >> lat_sh(LVL), unsigned word = 0, unsigned anyw = 0,
                                           unsigned long long rmask = 0, unsigned bofs = 0xffffffffu)
{
    (void)rmask;
    lat_v2<V> v[16];
    if constexpr (PRED) {
        lat_absorb_fetch01<LAY, LVL>(v, xcol, lane, sstride, word);
        lat_absorb<LAY, LVL, true, true, IO>(x, lds0, xcol, lane, cw, sstride, word, anyw, v);
    } else
        lat_absorb<LAY, LVL, false, false, IO>(x, lds0, xcol, lane, cw, sstride, word, anyw, v, nullptr, 0, nullptr, bofs);
}

// the three layout changes of the forward direction, shared by wpt and wpd
template <typename V> __device__ __forceinline__ void lat_t2(V (&a)[64], V (&bb)[64], unsigned lds0, int lane)
{
    const int sw = lane ^ ((lane >> 5) << 1);
    const unsigned wa0 = lds0 + 8u * sw, wa1 = lds0 + 8u * (sw ^ 1);
    const int H = lane & 15, p10 = lane >> 4;
    const int lam0 = 4 * H, sg = (p10 & 1) | ((lam0 >> 5) << 1);
    unsigned ra[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) ra[h] = lds0 + 8u * (64 * p10 + ((lam0 + h) ^ sg));
    lat_for<4>([&](auto Fq) {
        constexpr int f = Fq;
        lat_for<16>([&](auto Jq) {
            constexpr int j = Jq;
            lds_wr<8 * 64 * j>((j & 1) ? wa1 : wa0, a[16 * f + j]);
        });
        V t[16];
        lat_for<16>([&](auto Q) {
            constexpr int h = Q / 4, g = Q % 4;
            t[Q] = lds_rd<8 * 256 * g, V>(ra[h]);
        });
        lat_wait16<0>(t);
        lat_for<16>([&](auto Q) {
            constexpr int h = Q / 4, g = Q % 4;
            bb[16 * h + 4 * f + g] = t[Q];
        });
    });
}
template <typename V> __device__ __forceinline__ void lat_t3(V (&bb)[64], V (&c)[64], unsigned lds0, int lane)
{
    const unsigned wa = lds0 + 8u * (lane + (lane >> 5));
    const unsigned ra = lds0 + 8u * (66 * (lane >> 2) + 16 * (lane & 1) + 33 * ((lane >> 1) & 1));
    lat_for<4>([&](auto Fq) {
        constexpr int f = Fq;
        lat_for<16>([&](auto Jq) {
            constexpr int j = Jq;
            lds_wr<8 * 66 * j>(wa, bb[16 * f + j]);
        });
        V t[16];
        lat_for<16>([&](auto Hq) {
            constexpr int H = Hq;
            t[H] = lds_rd<8 * H, V>(ra);
        });
        lat_wait16<0>(t);
        lat_for<16>([&](auto Hq) {
            constexpr int H = Hq;
            c[4 * H + f] = t[H];
        });
    });
}

// wpd!(y, x, wt, L) DWT.jl:131-161 / wpdall dwt/dwt_all.jl:260-282 for 4096-sample Float64 signals: y is (4096, L+1, batch)
template <int NS, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpd_f64(
    const double *__restrict__ x, double *__restrict__ y, int L, int64_t batch, WxLatW cw)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    const int64_t sig = blockIdx.x;
    const double *xs = x + sig * 4096;
    double *ys = y + sig * 4096 * (int64_t)(L + 1);
    const WxLat &cf = cw.c;
    double a[64];
    {
        lat_d2 r[32];
        const unsigned xo = 64u * (lane >> 3) + 2u * (lane & 7);
        lat_for<32>([&](auto Q) {
            constexpr int hi3 = Q / 4, f = Q % 4;
            r[Q] = lat_ld2(lat_sbase(xs + 512 * hi3 + 16 * f) + xo);
        });
        // column 0 of the table is the signal (DWT.jl:145)
        lat_for<32>([&](auto Q) {
            constexpr int hi3 = Q / 4, f = Q % 4;
            lat_st2w(lat_sbase(ys + 512 * hi3 + 16 * f) + xo, r[Q]);
        });
        const unsigned wa = lds0 + 8u * (17u * (lane >> 3) + 2u * (lane & 7)), ra = lds0 + 8u * 17u * lane;
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<8>([&](auto Hq) {
                constexpr int hi3 = Hq;
                lds_wr<8 * (136 * hi3)>(wa, r[4 * hi3 + f].x);
                lds_wr<8 * (136 * hi3 + 1)>(wa, r[4 * hi3 + f].y);
            });
            double t[16];
            lat_for<16>([&](auto M) {
                constexpr int m = M;
                t[m] = lds_rd<8 * m>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto M) {
                constexpr int m = M;
                a[16 * f + m] = t[m];
            });
        });
    }
    lat_level<0, 6, NS, false>(a, cf);
    lat_emit<0, 1>(a, lds0, ys + 4096 * 1, lane, cw);
    if (L < 2) return;
    lat_level<1, 6, NS, false>(a, cf);
    lat_emit<0, 2>(a, lds0, ys + 4096 * 2, lane, cw);
    if (L < 3) return;
    double bb[64];
    lat_t2(a, bb, lds0, lane);
    lat_level<0, 4, NS, false>(bb, cf);
    lat_emit<2, 3>(bb, lds0, ys + 4096 * 3, lane, cw);
    if (L < 4) return;
    lat_level<1, 4, NS, false>(bb, cf);
    lat_emit<2, 4>(bb, lds0, ys + 4096 * 4, lane, cw);
    if (L < 5) return;
    lat_level<2, 4, NS, false>(bb, cf);
    lat_emit<2, 5>(bb, lds0, ys + 4096 * 5, lane, cw);
    if (L < 6) return;
    lat_level<3, 4, NS, false>(bb, cf);
    lat_emit<2, 6>(bb, lds0, ys + 4096 * 6, lane, cw);
    if (L < 7) return;
    double c[64];
    lat_t3(bb, c, lds0, lane);
    lat_level<0, 0, NS, false>(c, cf);
    lat_emit<6, 7>(c, lds0, ys + 4096 * 7, lane, cw);
    if (L < 8) return;
    lat_level<1, 0, NS, false>(c, cf);
    lat_emit<6, 8>(c, lds0, ys + 4096 * 8, lane, cw);
    if (L < 9) return;
    lat_level<2, 0, NS, false>(c, cf);
    lat_emit<6, 9>(c, lds0, ys + 4096 * 9, lane, cw);
    if (L < 10) return;
    lat_level<3, 0, NS, false>(c, cf);
    lat_emit<6, 10>(c, lds0, ys + 4096 * 10, lane, cw);
    if (L < 11) return;
    lat_level<4, 0, NS, false>(c, cf);
    lat_emit<6, 11>(c, lds0, ys + 4096 * 11, lane, cw);
    if (L < 12) return;
    lat_level<5, 0, NS, false>(c, cf);
    lat_emit<6, 12>(c, lds0, ys + 4096 * 12, lane, cw);
}

// ---------------------------------------------------------------- shorter signals: 2^SH of them interleaved in one wavefront
// 4096 >> SH samples per signal (2048, 1024): sample i of signal s sits at register-file index p = (i << SH) | s, i.e. the
// low SH index bits are a signal number that no level touches and level l acts on bit SH + l - 1 -- the same rotations,
// exchanges and halos as the 4096-sample kernel from its level SH + 1 on (the halo wraps at p + 4096 = i + 4096 >> SH of the
// same signal).  The 2^SH signals are adjacent in memory, so the wavefront still owns one contiguous 32 KiB block: it is read
// with 8-byte loads (a lane's two registers belong to different signals) and written through the static bit routing of
// lat_emit with the bit map of lat_obit(l + 16 SH, .): signal number on top, path bits reversed below it.
__device__ __forceinline__ double lat_ld1(const double __attribute__((address_space(1))) *p) { return *p; }
template <int SH> constexpr int lat_rotr(int p) { return (p >> SH) | ((p & ((1 << SH) - 1)) << (12 - SH)); }

template <int NS, int WPE, int SH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpt_sh_f64(
    const double *__restrict__ x, double *__restrict__ y, int L, int last_sig, WxLatW cw)
{
    static_assert(SH == 1 || SH == 2, "two or four signals per wavefront");
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    // the last wavefront of a batch that is not a multiple of 2^SH re-does the last 2^SH signals (out of place only)
    const int sig0 = min((int)blockIdx.x << SH, last_sig);
    const int64_t off = (int64_t)sig0 * (4096 >> SH);
    const double *xs = x + off;
    double *ys = y + off;
    const WxLat &cf = cw.c;
    double a[64];
    {
        lat_d2 r[32];
        const int lp = 64 * (lane >> 3) + 2 * (lane & 7);
        const unsigned xo = (unsigned)((lp >> SH) | ((lp & ((1 << SH) - 1)) << (12 - SH)));
        lat_for<32>([&](auto Q) {
            constexpr int hi3 = Q / 4, f = Q % 4;
            r[Q].x = lat_ld1(lat_sbase(xs + lat_rotr<SH>(512 * hi3 + 16 * f)) + xo);
            r[Q].y = lat_ld1(lat_sbase(xs + lat_rotr<SH>(512 * hi3 + 16 * f + 1)) + xo);
        });
        const unsigned wa = lds0 + 8u * (17u * (lane >> 3) + 2u * (lane & 7)), ra = lds0 + 8u * 17u * lane;
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<8>([&](auto Hq) {
                constexpr int hi3 = Hq;
                lds_wr<8 * (136 * hi3)>(wa, r[4 * hi3 + f].x);
                lds_wr<8 * (136 * hi3 + 1)>(wa, r[4 * hi3 + f].y);
            });
            double t[16];
            lat_for<16>([&](auto M) {
                constexpr int m = M;
                t[m] = lds_rd<8 * m>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto M) {
                constexpr int m = M;
                a[16 * f + m] = t[m];
            });
        });
    }
    if constexpr (SH < 2) lat_level<1, 6, NS, false>(a, cf);
    double bb[64];
    lat_t2(a, bb, lds0, lane);
    lat_level<0, 4, NS, false>(bb, cf);
    lat_level<1, 4, NS, false>(bb, cf);
    lat_level<2, 4, NS, false>(bb, cf);
    lat_level<3, 4, NS, false>(bb, cf);
    double c[64];
    lat_t3(bb, c, lds0, lane);
    const int Le = L + SH;                                  // highest index bit + 1 that a level acts on: 6 .. 12
    if (Le > 6) lat_level<0, 0, NS, false>(c, cf);
    if (Le > 7) lat_level<1, 0, NS, false>(c, cf);
    if (Le > 8) lat_level<2, 0, NS, false>(c, cf);
    if (Le > 9) lat_level<3, 0, NS, false>(c, cf);
    if (Le > 10) lat_level<4, 0, NS, false>(c, cf);
    if (Le > 11) lat_level<5, 0, NS, false>(c, cf);
    switch (Le) {
    case 6: lat_emit<6, 6 - SH + 16 * SH>(c, lds0, ys, lane, cw); break;
    case 7: lat_emit<6, 7 - SH + 16 * SH>(c, lds0, ys, lane, cw); break;
    case 8: lat_emit<6, 8 - SH + 16 * SH>(c, lds0, ys, lane, cw); break;
    case 9: lat_emit<6, 9 - SH + 16 * SH>(c, lds0, ys, lane, cw); break;
    case 10: lat_emit<6, 10 - SH + 16 * SH>(c, lds0, ys, lane, cw); break;
    case 11: lat_emit<6, 11 - SH + 16 * SH>(c, lds0, ys, lane, cw); break;
    default: lat_emit<6, 12 - SH + 16 * SH>(c, lds0, ys, lane, cw); break;
    }
}

// ---------------------------------------------------------------- inverse
// the inverse transform of one 4096-sample signal up to the L0 arrangement of its samples (see lat_fwd_from_l0): `sink(f, o)` receives,
// round by round (f = p[5:4]), the eight 16-byte pieces o[hi3] = samples p, p + 1 with p[11:9] = hi3, p[8:6] = lane >> 3, p[3:1] = lane & 7.
// k_lat_iwpt_f64 stores them (eight complete lines per instruction); k_lat_iwpt8k_f64 keeps them as one child of an 8192-sample signal.
template <int NS, typename TM, typename SINK>
__device__ __forceinline__ void lat_inv_to_l0(const TM *__restrict__ xs, unsigned lds0, int lane, int L, const WxLat &cf, SINK &&sink)
{
    double c[64];
    switch (L) {
    case 6: lat_load_c<6>(c, lds0, xs, lane, cf); break;
    case 7: lat_load_c<7>(c, lds0, xs, lane, cf); break;
    case 8: lat_load_c<8>(c, lds0, xs, lane, cf); break;
    case 9: lat_load_c<9>(c, lds0, xs, lane, cf); break;
    case 10: lat_load_c<10>(c, lds0, xs, lane, cf); break;
    case 11: lat_load_c<11>(c, lds0, xs, lane, cf); break;
    default: lat_load_c<12>(c, lds0, xs, lane, cf); break;
    }
    if (L > 11) lat_level<5, 0, NS, true>(c, cf);
    if (L > 10) lat_level<4, 0, NS, true>(c, cf);
    if (L > 9) lat_level<3, 0, NS, true>(c, cf);
    if (L > 8) lat_level<2, 0, NS, true>(c, cf);
    if (L > 7) lat_level<1, 0, NS, true>(c, cf);
    if (L > 6) lat_level<0, 0, NS, true>(c, cf);
    // T3i: C -> B
    double bb[64];
    {
        const unsigned wa = lds0 + 8u * (34 * (lane >> 1) + (lane & 1));
        const int H = lane & 15, p0 = (lane >> 4) & 1, p1 = lane >> 5;
        const unsigned ra = lds0 + 8u * (34 * p1 + 2 * H + p0);
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<16>([&](auto Hq) {
                constexpr int Hh = Hq;
                lds_wr<8 * 2 * Hh>(wa, c[4 * Hh + f]);
            });
            double t[16];
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                t[j] = lds_rd<8 * 68 * j>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                bb[16 * f + j] = t[j];
            });
        });
    }
    lat_level<3, 4, NS, true>(bb, cf);
    lat_level<2, 4, NS, true>(bb, cf);
    lat_level<1, 4, NS, true>(bb, cf);
    lat_level<0, 4, NS, true>(bb, cf);
    // T2i: B -> A
    double a[64];
    {
        const int H = lane & 15, p10 = lane >> 4;
        unsigned wa[4];
#pragma unroll
        for (int h = 0; h < 4; ++h)
            wa[h] = lds0 + 8u * (((h ^ (H >> 2)) | ((H & 3) << 2) | (((H >> 2) & 1) << 4) | ((H >> 3) << 5)) + 64 * p10);
        const int Ha = lane >> 2, ha = lane & 3;
        const unsigned ra = lds0 + 8u * ((ha ^ (Ha >> 2)) | ((Ha & 3) << 2) | (((Ha >> 2) & 1) << 4) | ((Ha >> 3) << 5));
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<16>([&](auto Q) {
                constexpr int h = Q / 4, g = Q % 4;
                lds_wr<8 * 256 * g>(wa[h], bb[16 * h + 4 * f + g]);
            });
            double t[16];
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                t[j] = lds_rd<8 * 64 * j>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                a[16 * f + j] = t[j];
            });
        });
    }
    lat_level<1, 6, NS, true>(a, cf);
    lat_level<0, 6, NS, true>(a, cf);
    lat_t1i(a, lds0, lane, sink);
}

template <int NS, int WPE, typename TM = double>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_iwpt_f64(
    const TM *__restrict__ xw, TM *__restrict__ y, int L, int64_t batch, int64_t in_stride, WxLat cf)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    const int64_t sig = blockIdx.x;
    const unsigned yo = 64u * (lane >> 3) + 2u * (lane & 7);
    TM *ys = y + sig * 4096;
    lat_inv_to_l0<NS>(xw + sig * in_stride, lds0, lane, L, cf, [&](auto Fq, lat_d2 (&o)[8]) {
        constexpr int f = decltype(Fq)::value;
        lat_for<8>([&](auto Hq) {
            constexpr int hi3 = Hq;
            lat_st2(lat_sbase(ys + 512 * hi3 + 16 * f) + yo, o[hi3]);
        });
    });
}

// wpd of 2^SH interleaved signals: y is (n, L+1, batch), n = 4096 >> SH; every level leaves through lat_emit with the
// signal number routed to the table of its signal (sstride = n (L+1))
template <int NS, int WPE, int SH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpd_sh_f64(
    const double *__restrict__ x, double *__restrict__ y, int L, int last_sig, WxLatW cw)
{
    static_assert(SH == 1 || SH == 2, "two or four signals per wavefront");
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    constexpr int N = 4096 >> SH;
    const int sig0 = min((int)blockIdx.x << SH, last_sig);
    const double *xs = x + (int64_t)sig0 * N;
    const unsigned ts = (unsigned)(N * (L + 1));                          // table stride between signals
    double *ys = y + (int64_t)sig0 * ts;
    const WxLat &cf = cw.c;
    double a[64];
    {
        lat_d2 r[32];
        const int lp = 64 * (lane >> 3) + 2 * (lane & 7);
        const unsigned xo = (unsigned)((lp >> SH) | ((lp & ((1 << SH) - 1)) << (12 - SH)));
        lat_for<32>([&](auto Q) {
            constexpr int hi3 = Q / 4, f = Q % 4;
            r[Q].x = lat_ld1(lat_sbase(xs + lat_rotr<SH>(512 * hi3 + 16 * f)) + xo);
            r[Q].y = lat_ld1(lat_sbase(xs + lat_rotr<SH>(512 * hi3 + 16 * f + 1)) + xo);
        });
        const unsigned wa = lds0 + 8u * (17u * (lane >> 3) + 2u * (lane & 7)), ra = lds0 + 8u * 17u * lane;
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<8>([&](auto Hq) {
                constexpr int hi3 = Hq;
                lds_wr<8 * (136 * hi3)>(wa, r[4 * hi3 + f].x);
                lds_wr<8 * (136 * hi3 + 1)>(wa, r[4 * hi3 + f].y);
            });
            double t[16];
            lat_for<16>([&](auto M) {
                constexpr int m = M;
                t[m] = lds_rd<8 * m>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto M) {
                constexpr int m = M;
                a[16 * f + m] = t[m];
            });
        });
    }
    // column 0 of every table is the signal (DWT.jl:145): the plain rotation of the index
    lat_emit<0, 16 * SH>(a, lds0, ys, lane, cw, ts);
    // level l acts on bit SH + l - 1: layout A holds bits 0, 1, layout B bits 2..5, layout C bits 6..11
#define WX_LVL_A(K)                                                                               \
    if constexpr (K >= SH) {                                                                      \
        constexpr int l = K - SH + 1;                                                             \
        lat_level<K, 6, NS, false>(a, cf);                                                        \
        lat_emit<0, l + 16 * SH>(a, lds0, ys + N * l, lane, cw, ts);                              \
        if (L <= l) return;                                                                       \
    }
    WX_LVL_A(1)
#undef WX_LVL_A
    double bb[64];
    lat_t2(a, bb, lds0, lane);
#define WX_LVL_B(K)                                                                               \
    {                                                                                             \
        constexpr int l = K + 2 - SH + 1;                                                         \
        lat_level<K, 4, NS, false>(bb, cf);                                                       \
        lat_emit<2, l + 16 * SH>(bb, lds0, ys + N * l, lane, cw, ts);                             \
        if (L <= l) return;                                                                       \
    }
    WX_LVL_B(0) WX_LVL_B(1) WX_LVL_B(2) WX_LVL_B(3)
#undef WX_LVL_B
    double c[64];
    lat_t3(bb, c, lds0, lane);
#define WX_LVL_C(K)                                                                               \
    {                                                                                             \
        constexpr int l = K + 6 - SH + 1;                                                         \
        lat_level<K, 0, NS, false>(c, cf);                                                        \
        lat_emit<6, l + 16 * SH>(c, lds0, ys + N * l, lane, cw, ts);                              \
        if (L <= l) return;                                                                       \
    }
    WX_LVL_C(0) WX_LVL_C(1) WX_LVL_C(2) WX_LVL_C(3) WX_LVL_C(4) WX_LVL_C(5)
#undef WX_LVL_C
}

// inverse of k_lat_wpt_sh_f64: 2^SH signals of 4096 >> SH samples per wavefront, leaves dense and adjacent in memory
template <int NS, int WPE, int SH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_iwpt_sh_f64(
    const double *__restrict__ xw, double *__restrict__ y, int L, int last_sig, unsigned in_stride, WxLatW cw)
{
    static_assert(SH == 1 || SH == 2, "two or four signals per wavefront");
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    const int sig0 = min((int)blockIdx.x << SH, last_sig);
    const double *xs = xw + (int64_t)sig0 * in_stride;        // leaves of signal s at xs + s in_stride (a packet table's last column)
    double *ys = y + (int64_t)sig0 * (4096 >> SH);
    const WxLat &cf = cw.c;
    const int Le = L + SH;
    double c[64];
    switch (Le) {
    case 6: lat_absorb<6, 6 - SH + 16 * SH>(c, lds0, xs, lane, cw, in_stride); break;
    case 7: lat_absorb<6, 7 - SH + 16 * SH>(c, lds0, xs, lane, cw, in_stride); break;
    case 8: lat_absorb<6, 8 - SH + 16 * SH>(c, lds0, xs, lane, cw, in_stride); break;
    case 9: lat_absorb<6, 9 - SH + 16 * SH>(c, lds0, xs, lane, cw, in_stride); break;
    case 10: lat_absorb<6, 10 - SH + 16 * SH>(c, lds0, xs, lane, cw, in_stride); break;
    case 11: lat_absorb<6, 11 - SH + 16 * SH>(c, lds0, xs, lane, cw, in_stride); break;
    default: lat_absorb<6, 12 - SH + 16 * SH>(c, lds0, xs, lane, cw, in_stride); break;
    }
    if (Le > 11) lat_level<5, 0, NS, true>(c, cf);
    if (Le > 10) lat_level<4, 0, NS, true>(c, cf);
    if (Le > 9) lat_level<3, 0, NS, true>(c, cf);
    if (Le > 8) lat_level<2, 0, NS, true>(c, cf);
    if (Le > 7) lat_level<1, 0, NS, true>(c, cf);
    if (Le > 6) lat_level<0, 0, NS, true>(c, cf);
    // T3i: C -> B
    double bb[64];
    {
        const unsigned wa = lds0 + 8u * (34 * (lane >> 1) + (lane & 1));
        const int H = lane & 15, p0 = (lane >> 4) & 1, p1 = lane >> 5;
        const unsigned ra = lds0 + 8u * (34 * p1 + 2 * H + p0);
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<16>([&](auto Hq) {
                constexpr int Hh = Hq;
                lds_wr<8 * 2 * Hh>(wa, c[4 * Hh + f]);
            });
            double t[16];
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                t[j] = lds_rd<8 * 68 * j>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                bb[16 * f + j] = t[j];
            });
        });
    }
    lat_level<3, 4, NS, true>(bb, cf);
    lat_level<2, 4, NS, true>(bb, cf);
    lat_level<1, 4, NS, true>(bb, cf);
    lat_level<0, 4, NS, true>(bb, cf);
    // T2i: B -> A
    double a[64];
    {
        const int H = lane & 15, p10 = lane >> 4;
        unsigned wa[4];
#pragma unroll
        for (int h = 0; h < 4; ++h)
            wa[h] = lds0 + 8u * (((h ^ (H >> 2)) | ((H & 3) << 2) | (((H >> 2) & 1) << 4) | ((H >> 3) << 5)) + 64 * p10);
        const int Ha = lane >> 2, ha = lane & 3;
        const unsigned ra = lds0 + 8u * ((ha ^ (Ha >> 2)) | ((Ha & 3) << 2) | (((Ha >> 2) & 1) << 4) | ((Ha >> 3) << 5));
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            lat_for<16>([&](auto Q) {
                constexpr int h = Q / 4, g = Q % 4;
                lds_wr<8 * 256 * g>(wa[h], bb[16 * h + 4 * f + g]);
            });
            double t[16];
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                t[j] = lds_rd<8 * 64 * j>(ra);
            });
            lat_wait16<0>(t);
            lat_for<16>([&](auto Jq) {
                constexpr int j = Jq;
                a[16 * f + j] = t[j];
            });
        });
    }
    if constexpr (SH < 2) lat_level<1, 6, NS, true>(a, cf);
    // layout A -> memory through the static bit routing (complete lines): address = the index rotated right by SH
    lat_emit<0, 16 * SH>(a, lds0, ys, lane, cw);
}

// ---------------------------------------------------------------- short signals, general form: 2^SH signals per wavefront
// SH = 3 .. 6 (512, 256, 128, 64 samples; the same code is valid for SH = 1, 2).  Both ends are the bit routing of
// lat_absorb / lat_emit (complete lines), and the data enters in the first layout that has a level to run: bits 0, 1 live
// in layout A, bits 2..5 in B, bits 6..11 in C, so for SH >= 2 the exchange A -> B disappears and for SH = 6 (64 signals of
// 64 samples) the whole transform happens in layout C between one absorb and one emit.
template <int SH> constexpr int lat_lay0() { return SH < 2 ? 0 : (SH < 6 ? 2 : 6); }

// T3i: C -> B and T2i: B -> A (the exchanges of k_lat_iwpt_f64)
template <typename V> __device__ __forceinline__ void lat_t3i(V (&c)[64], V (&bb)[64], unsigned lds0, int lane)
{
    const unsigned wa = lds0 + 8u * (34 * (lane >> 1) + (lane & 1));
    const int H = lane & 15, p0 = (lane >> 4) & 1, p1 = lane >> 5;
    const unsigned ra = lds0 + 8u * (34 * p1 + 2 * H + p0);
    lat_for<4>([&](auto Fq) {
        constexpr int f = Fq;
        lat_for<16>([&](auto Hq) {
            constexpr int Hh = Hq;
            lds_wr<8 * 2 * Hh>(wa, c[4 * Hh + f]);
        });
        V t[16];
        lat_for<16>([&](auto Jq) {
            constexpr int j = Jq;
            t[j] = lds_rd<8 * 68 * j, V>(ra);
        });
        lat_wait16<0>(t);
        lat_for<16>([&](auto Jq) {
            constexpr int j = Jq;
            bb[16 * f + j] = t[j];
        });
    });
}
template <typename V> __device__ __forceinline__ void lat_t2i(V (&bb)[64], V (&a)[64], unsigned lds0, int lane)
{
    const int H = lane & 15, p10 = lane >> 4;
    unsigned wa[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
        wa[h] = lds0 + 8u * (((h ^ (H >> 2)) | ((H & 3) << 2) | (((H >> 2) & 1) << 4) | ((H >> 3) << 5)) + 64 * p10);
    const int Ha = lane >> 2, ha = lane & 3;
    const unsigned ra = lds0 + 8u * ((ha ^ (Ha >> 2)) | ((Ha & 3) << 2) | (((Ha >> 2) & 1) << 4) | ((Ha >> 3) << 5));
    lat_for<4>([&](auto Fq) {
        constexpr int f = Fq;
        lat_for<16>([&](auto Q) {
            constexpr int h = Q / 4, g = Q % 4;
            lds_wr<8 * 256 * g>(wa[h], bb[16 * h + 4 * f + g]);
        });
        V t[16];
        lat_for<16>([&](auto Jq) {
            constexpr int j = Jq;
            t[j] = lds_rd<8 * 64 * j, V>(ra);
        });
        lat_wait16<0>(t);
        lat_for<16>([&](auto Jq) {
            constexpr int j = Jq;
            a[16 * f + j] = t[j];
        });
    });
}

// forward wpt, leaves only: Le = L + SH in 6 .. 12 (the last level runs in layout C)
// (IO = float: Float32 signals -- the loads widen, the stores round once, as in k_lat_wpt_f64<NS, WPE, float>)
// (F32A = false with IO = float: Float32 in memory, Float64 in the registers, one signal set per wavefront -- the round-4 form, kept for
// the 64- and 128-sample kernels where the pair form measured slower: profiles/r05_floor.txt)
// the body of the interleaved forward kernels: the wavefront's 2^SH signals (two sets for V = lat_f2v) at xs -> ys.  TC = 0: signals one
// after the other, 4096 >> SH elements apart (ss_in = ss_out); TC = 512: adjacent ROWS of an image, ss = its column stride (lat_isT)
template <int NS, int SH, int TC, typename IO, typename V, typename CFT>
__device__ __forceinline__ void lat_g_fwd(const IO *__restrict__ xs, IO *__restrict__ ys, int L, unsigned lds0, int lane, const WxLatW &cw,
                                          const CFT &cf, unsigned ss_in, unsigned ss_out, unsigned bofs_in, unsigned bofs)
{
    V c[64];
    if constexpr (SH < 2) {
        V a[64], bb[64];
        lat_absorb<0, TC + 16 * SH>(a, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in);
        if constexpr (SH < 1) lat_level<0, 6, NS, false>(a, cf);
        lat_level<1, 6, NS, false>(a, cf);
        lat_t2(a, bb, lds0, lane);
        lat_level<0, 4, NS, false>(bb, cf);
        lat_level<1, 4, NS, false>(bb, cf);
        lat_level<2, 4, NS, false>(bb, cf);
        lat_level<3, 4, NS, false>(bb, cf);
        lat_t3(bb, c, lds0, lane);
    } else if constexpr (SH < 6) {
        V bb[64];
        lat_absorb<2, TC + 16 * SH>(bb, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in);
        if constexpr (SH <= 2) lat_level<0, 4, NS, false>(bb, cf);
        if constexpr (SH <= 3) lat_level<1, 4, NS, false>(bb, cf);
        if constexpr (SH <= 4) lat_level<2, 4, NS, false>(bb, cf);
        lat_level<3, 4, NS, false>(bb, cf);
        lat_t3(bb, c, lds0, lane);
    } else {
        lat_absorb<6, TC + 16 * SH>(c, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in);
    }
    const int Le = L + SH;
    if (Le > 6) lat_level<0, 0, NS, false>(c, cf);
    if (Le > 7) lat_level<1, 0, NS, false>(c, cf);
    if (Le > 8) lat_level<2, 0, NS, false>(c, cf);
    if (Le > 9) lat_level<3, 0, NS, false>(c, cf);
    if (Le > 10) lat_level<4, 0, NS, false>(c, cf);
    if (Le > 11) lat_level<5, 0, NS, false>(c, cf);
    switch (Le) {
    case 6: if constexpr (SH < 6) lat_emit<6, TC + 6 - SH + 16 * SH>(c, lds0, ys, lane, cw, ss_out, 0, 0, bofs); break;
    case 7: lat_emit<6, TC + 7 - SH + 16 * SH>(c, lds0, ys, lane, cw, ss_out, 0, 0, bofs); break;
    case 8: lat_emit<6, TC + 8 - SH + 16 * SH>(c, lds0, ys, lane, cw, ss_out, 0, 0, bofs); break;
    case 9: lat_emit<6, TC + 9 - SH + 16 * SH>(c, lds0, ys, lane, cw, ss_out, 0, 0, bofs); break;
    case 10: lat_emit<6, TC + 10 - SH + 16 * SH>(c, lds0, ys, lane, cw, ss_out, 0, 0, bofs); break;
    case 11: lat_emit<6, TC + 11 - SH + 16 * SH>(c, lds0, ys, lane, cw, ss_out, 0, 0, bofs); break;
    default: lat_emit<6, TC + 12 - SH + 16 * SH>(c, lds0, ys, lane, cw, ss_out, 0, 0, bofs); break;
    }
}

template <int NS, int WPE, int SH, typename IO = double, bool F32A = true>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpt_g_f64(
    const IO *__restrict__ x, IO *__restrict__ y, int L, int last_sig, WxLatW cw)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    // IO = float: Float32 arithmetic on pairs of signals (lat_f2v) -- the wavefront takes 2 x 2^SH signals, the second set follows the
    // first in memory; last_sig is then batch - 2 x 2^SH
    typedef typename std::conditional<std::is_same<IO, float>::value && F32A, lat_f2v, double>::type V;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    // pair kernels: last_sig is the tail wavefront's first signal, cw.tail_bsig its second set's distance (wx_lat_pair_plan)
    const bool lastw = PAIR && blockIdx.x == gridDim.x - 1;
    const int sig0 = PAIR ? (lastw ? last_sig : (int)(blockIdx.x << (SH + 1))) : min((int)blockIdx.x << SH, last_sig);
    const unsigned bofs = PAIR ? (unsigned)(lastw ? cw.tail_bsig : (1 << SH)) * (4096u >> SH) : 0xffffffffu;
    const int64_t off = (int64_t)sig0 * (4096 >> SH);
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
    lat_g_fwd<NS, SH, 0, IO, V>(x + off, y + off, L, lds0, lane, cw, cf, 4096u >> SH, 4096u >> SH, bofs, bofs);
}

// inverse wpt: leaves of signal s at xw + s in_stride (dense array or the last column of packet tables)
template <int NS, int SH, int TC, typename IO, typename V, typename CFT>
__device__ __forceinline__ void lat_g_inv(const IO *__restrict__ xs, IO *__restrict__ ys, int L, unsigned lds0, int lane, const WxLatW &cw,
                                          const CFT &cf, unsigned ss_in, unsigned ss_out, unsigned bofs_in, unsigned bofs)
{
    const int Le = L + SH;
    V c[64];
    switch (Le) {
    case 6: if constexpr (SH < 6) lat_absorb<6, TC + 6 - SH + 16 * SH>(c, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in); break;
    case 7: lat_absorb<6, TC + 7 - SH + 16 * SH>(c, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in); break;
    case 8: lat_absorb<6, TC + 8 - SH + 16 * SH>(c, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in); break;
    case 9: lat_absorb<6, TC + 9 - SH + 16 * SH>(c, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in); break;
    case 10: lat_absorb<6, TC + 10 - SH + 16 * SH>(c, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in); break;
    case 11: lat_absorb<6, TC + 11 - SH + 16 * SH>(c, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in); break;
    default: lat_absorb<6, TC + 12 - SH + 16 * SH>(c, lds0, xs, lane, cw, ss_in, 0, 0, 0, bofs_in); break;
    }
    if (Le > 11) lat_level<5, 0, NS, true>(c, cf);
    if (Le > 10) lat_level<4, 0, NS, true>(c, cf);
    if (Le > 9) lat_level<3, 0, NS, true>(c, cf);
    if (Le > 8) lat_level<2, 0, NS, true>(c, cf);
    if (Le > 7) lat_level<1, 0, NS, true>(c, cf);
    if (Le > 6) lat_level<0, 0, NS, true>(c, cf);
    if constexpr (SH >= 6) {
        lat_emit<6, TC + 16 * SH>(c, lds0, ys, lane, cw, ss_out, 0, 0, bofs);
    } else {
        V bb[64];
        lat_t3i(c, bb, lds0, lane);
        lat_level<3, 4, NS, true>(bb, cf);
        if constexpr (SH <= 4) lat_level<2, 4, NS, true>(bb, cf);
        if constexpr (SH <= 3) lat_level<1, 4, NS, true>(bb, cf);
        if constexpr (SH <= 2) lat_level<0, 4, NS, true>(bb, cf);
        if constexpr (SH >= 2) {
            lat_emit<2, TC + 16 * SH>(bb, lds0, ys, lane, cw, ss_out, 0, 0, bofs);
        } else {
            V a[64];
            lat_t2i(bb, a, lds0, lane);
            lat_level<1, 6, NS, true>(a, cf);
            if constexpr (SH < 1) lat_level<0, 6, NS, true>(a, cf);
            lat_emit<0, TC + 16 * SH>(a, lds0, ys, lane, cw, ss_out, 0, 0, bofs);
        }
    }
}

template <int NS, int WPE, int SH, typename IO = double, bool F32A = true>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_iwpt_g_f64(
    const IO *__restrict__ xw, IO *__restrict__ y, int L, int last_sig, unsigned in_stride, WxLatW cw)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    typedef typename std::conditional<std::is_same<IO, float>::value && F32A, lat_f2v, double>::type V;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    const bool lastw = PAIR && blockIdx.x == gridDim.x - 1;
    const int sig0 = PAIR ? (lastw ? last_sig : (int)(blockIdx.x << (SH + 1))) : min((int)blockIdx.x << SH, last_sig);
    const unsigned bsig = (unsigned)(lastw ? cw.tail_bsig : (1 << SH));
    const unsigned bofs_in = PAIR ? bsig * in_stride : 0xffffffffu, bofs = PAIR ? bsig * (4096u >> SH) : 0xffffffffu;
    const IO *xs = xw + (int64_t)sig0 * in_stride;
    IO *ys = y + (int64_t)sig0 * (4096 >> SH);
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
    lat_g_inv<NS, SH, 0, IO, V>(xs, ys, L, lds0, lane, cw, cf, in_stride, 4096u >> SH, bofs_in, bofs);
}

// wpd: y is (n, L+1, batch), every level leaves through lat_emit into the table of its signal (L >= 1, L + SH <= 12)
// (IO = float: Float32 signals and table, Float64 registers -- the loads widen, every emission rounds once)
// (F32A with IO = float: Float32 arithmetic on pairs of signals, lat_f2v -- the wavefront takes 2 x 2^SH signals; last_sig / cw.tail_bsig as in
// k_lat_wpt_g_f64)
template <int NS, int WPE, int SH, typename IO = double, bool F32A = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpd_g_f64(
    const IO *__restrict__ x, IO *__restrict__ y, int L, int last_sig, WxLatW cw)
{
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    constexpr int N = 4096 >> SH;
    typedef typename std::conditional<std::is_same<IO, float>::value && F32A, lat_f2v, double>::type V;
    constexpr bool PAIR = lat_vtraits<V>::pair != 0;
    const bool lastw = PAIR && blockIdx.x == gridDim.x - 1;
    const int sig0 = PAIR ? (lastw ? last_sig : (int)(blockIdx.x << (SH + 1))) : min((int)blockIdx.x << SH, last_sig);
    const IO *xs = x + (int64_t)sig0 * N;
    const unsigned ts = (unsigned)(N * (L + 1));
    IO *ys = y + (int64_t)sig0 * ts;
    const unsigned bsig = (unsigned)(lastw ? cw.tail_bsig : (1 << SH));
    const unsigned bofs_in = PAIR ? bsig * (unsigned)N : 0xffffffffu, bofs = PAIR ? bsig * ts : 0xffffffffu;
    const typename std::conditional<PAIR, WxLatF, const WxLat &>::type cf = lat_cfsel<NS, PAIR>(cw.c);
#define WX_LVL(LAY, KK, HH, REG, BIT)                                                           \
    if constexpr (BIT >= SH) {                                                                  \
        constexpr int l = BIT - SH + 1;                                                         \
        lat_level<KK, HH, NS, false>(REG, cf);                                                  \
        lat_emit<LAY, l + 16 * SH>(REG, lds0, ys + N * l, lane, cw, ts, 0, 0, bofs);            \
        if (L <= l) return;                                                                     \
    }
    V c[64];
    if constexpr (SH < 2) {
        V a[64], bb[64];
        lat_absorb<0, 16 * SH>(a, lds0, xs, lane, cw, 4096u >> SH, 0, 0, 0, bofs_in);
        lat_emit<0, 16 * SH>(a, lds0, ys, lane, cw, ts, 0, 0, bofs);       // column 0 = the signal
        WX_LVL(0, 0, 6, a, 0) WX_LVL(0, 1, 6, a, 1)
        lat_t2(a, bb, lds0, lane);
        WX_LVL(2, 0, 4, bb, 2) WX_LVL(2, 1, 4, bb, 3) WX_LVL(2, 2, 4, bb, 4) WX_LVL(2, 3, 4, bb, 5)
        lat_t3(bb, c, lds0, lane);
    } else if constexpr (SH < 6) {
        V bb[64];
        lat_absorb<2, 16 * SH>(bb, lds0, xs, lane, cw, 4096u >> SH, 0, 0, 0, bofs_in);
        lat_emit<2, 16 * SH>(bb, lds0, ys, lane, cw, ts, 0, 0, bofs);
        WX_LVL(2, 0, 4, bb, 2) WX_LVL(2, 1, 4, bb, 3) WX_LVL(2, 2, 4, bb, 4) WX_LVL(2, 3, 4, bb, 5)
        lat_t3(bb, c, lds0, lane);
    } else {
        lat_absorb<6, 16 * SH>(c, lds0, xs, lane, cw, 4096u >> SH, 0, 0, 0, bofs_in);
        lat_emit<6, 16 * SH>(c, lds0, ys, lane, cw, ts, 0, 0, bofs);
    }
    WX_LVL(6, 0, 0, c, 6) WX_LVL(6, 1, 0, c, 7) WX_LVL(6, 2, 0, c, 8) WX_LVL(6, 3, 0, c, 9) WX_LVL(6, 4, 0, c, 10) WX_LVL(6, 5, 0, c, 11)
#undef WX_LVL
}


// ---------------------------------------------------------------- tree-driven wpt / iwpt / iwpd (wx_lattice_tree.h)
// wpt(x, wt, tree) is the set of leaves of the tree picked out of the packet table (Wavelets.jl's wpt along a tree;
// Utils.jl:101-134 getbasiscoef; call sites DWT.jl:340-351, dwt/dwt_all.jl:152-166, 210-225, LDB.jl:303, 409,
// Denoising.jl:527).  The lattice computes every node of every level anyway, in registers, so the tree only decides
// WHERE a coefficient leaves (forward) or enters (inverse): after level l the static routing of lat_emit writes the lines of
// the nodes that are leaves at depth l -- and only those -- to their place in the output; the inverse takes the leaves of
// depth l in through lat_absorb before it runs synthesis level l on top of what the deeper levels have rebuilt.  Nodes
// below a leaf are computed and never stored (forward) or are zeros (inverse).  The predicates are tables of a tree,
// made once per call by k_lat_tree_prep with the same routing functions the kernels are compiled from.
//
// Level code LVL = l + 16 SH as everywhere in this file: 2^SH signals of N = 4096 >> SH samples per wavefront.
__device__ __forceinline__ bool lat_tree_leaf(const uint8_t *status, int64_t nstatus, int l, int j)
{
    const int idx = (1 << l) + j;                                   // heap index (1-based) of node j of depth l
    for (int d = 1; d <= l; ++d) {
        const int anc = idx >> d;
        if (anc - 1 >= nstatus || !status[anc - 1]) return false;     // an ancestor is a leaf: the node does not exist
    }
    return idx - 1 >= nstatus || !status[idx - 1];
}
constexpr int lat_tree_lay(int bit) { return bit < 2 ? 0 : (bit < 6 ? 2 : 6); }      // layout in which index bit `bit` is transformed

// ---------------------------------------------------------------- deep leaves in one exchange
// The leaves of depth < 6 - SH leave / enter level by level as described above.  In layout C (register p[11:6], lane p[5:0])
// every level is lane-local and a lane owns the 64 output positions of "its" node of depth 6 - SH whatever the subtree below
// looks like.  So for the leaves of depth >= 6 - SH:
//  * the levels of layout C run under LANE MASKS (one 64-bit mask per level and sequence: bit = "this lane's node is split";
//    the mask is the exec operand, __builtin_amdgcn_inverse_ballot_w64): a leaf stops changing where the tree says, nodes
//    that do not exist are never touched.  The register renamings between the shears stay unconditional (they add up to
//    the identity over a level); each level normalises its own gains (a g, d / g) under the same mask;
//  * afterwards an in-register conditional UNSHUFFLE (stage j deinterleaves the blocks of 64 >> j registers whose node is
//    split, v_cndmask on the same masks) turns the in-place dilated order into the packet order of the lane's chunk, and
//    ONE lat_emit of depth 6 - SH stores all of them: one exchange instead of one per populated depth.
// The inverse mirrors it: one lat_absorb of the existing chunks (for iwpd every 16-byte piece comes out of the column of its
// leaf's depth), the conditional shuffle, the masked synthesis levels.
struct WxLatTreeTab {
    unsigned words[13 * 64];           // [l][lane]: leaf pieces of depth l per store instruction (lat_emit / lat_absorb, PRED)
    unsigned any[16];                  // [l]: OR over the lanes
    unsigned long long masks[64];      // [(1 << K) - 1 + s]: lanes whose node of depth 6 - SH + K, sequence s, is split
    unsigned wordsF[64];               // [lane]: pieces of existing nodes of depth 6 - SH (the final exchange)
    unsigned wdepth[4 * 64];           // [w][lane]: leaf depth of piece 8 w + k in nibble k (iwpd: the column it is read from)
    unsigned anyF;
    unsigned stage_any[6];             // some node of depth 6 - SH + j is split
};

__device__ __forceinline__ bool lat_tree_exists(const uint8_t *status, int64_t nstatus, int l, int j)
{
    const int idx = (1 << l) + j;
    for (int d = 1; d <= l; ++d) {
        const int anc = idx >> d;
        if (anc - 1 >= nstatus || !status[anc - 1]) return false;
    }
    return true;
}
__device__ __forceinline__ bool lat_tree_split(const uint8_t *status, int64_t nstatus, int l, int j)
{
    const int idx = (1 << l) + j;
    return lat_tree_exists(status, nstatus, l, j) && idx - 1 < nstatus && status[idx - 1];
}

// blocks 0 .. 11: level l = block + 1 of words / any; block 12: the deep tables
template <int SH>
__global__ __launch_bounds__(64) void k_lat_tree_prep(const uint8_t *__restrict__ status, int64_t nstatus, int L, int Lcut,
                                                        WxLatTreeTab *__restrict__ tab)
{
    const int lane = threadIdx.x;
    constexpr int SB = 12 - SH, L6 = 6 - SH;
    if (blockIdx.x < 12) {
        const int l = blockIdx.x + 1;
        unsigned w = 0;
        if (l <= L && l <= SB && l <= Lcut) {
            const int lay = lat_tree_lay(SH + l - 1), lc = l + 16 * SH;
            const int qq = lane >> 3;
            int o_lane = 2 * (lane & 7);
            for (int q = 0; q < 3; ++q) o_lane |= ((qq >> q) & 1) << lat_line(lay, lc, q).ob;
            for (int rho = 0; rho < 4; ++rho)
                for (int i = 0; i < 8; ++i) {
                    const int o = o_lane | lat_emit_o_round(lay, lc, rho) | lat_emit_o_instr(lay, lc, i);
                    const int pos = o & ((1 << SB) - 1);
                    if (lat_tree_leaf(status, nstatus, l, pos >> (SB - l))) w |= 1u << (8 * rho + i);
                }
        }
        tab->words[64 * l + lane] = w;
        unsigned a = w;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) a |= __shfl_xor(a, d, 64);
        if (lane == 0) tab->any[l] = a;
        return;
    }
    // ---- deep tables
    for (int K = 0; K < 6; ++K) {
        unsigned long long acc = 0;
        for (int s = 0; s < (1 << K); ++s) {
            const int d = L6 + K;                                  // depth of the node that level d + 1 would split
            int j = 0;
            for (int t = 0; t < d; ++t) {
                const int bit = t < L6 ? (lane >> (SH + t)) & 1 : (s >> (t - L6)) & 1;
                j |= bit << (d - 1 - t);
            }
            const bool sp = d < L && d < SB && lat_tree_split(status, nstatus, d, j);
            const unsigned long long m = __ballot(sp);
            if (lane == 0) tab->masks[(1 << K) - 1 + s] = m;
            acc |= m;
        }
        if (lane == 0) tab->stage_any[K] = acc != 0;
    }
    unsigned w = 0, dep[4] = {0, 0, 0, 0};
    if (L >= L6) {
        const int lay = 6, lc = L6 + 16 * SH;
        const int qq = lane >> 3;
        int o_lane = 2 * (lane & 7);
        for (int q = 0; q < 3; ++q) o_lane |= ((qq >> q) & 1) << lat_line(lay, lc, q).ob;
        for (int rho = 0; rho < 4; ++rho)
            for (int i = 0; i < 8; ++i) {
                const int o = o_lane | lat_emit_o_round(lay, lc, rho) | lat_emit_o_instr(lay, lc, i);
                const int pos = o & ((1 << SB) - 1);
                if (!lat_tree_exists(status, nstatus, L6, pos >> (SB - L6))) continue;
                int d = L6;
                while (d < SB && lat_tree_split(status, nstatus, d, pos >> (SB - d))) ++d;
                const int idx = 8 * rho + i;
                w |= 1u << idx;
                dep[idx >> 3] |= (unsigned)d << (4 * (idx & 7));
            }
    }
    tab->wordsF[lane] = w;
    for (int k = 0; k < 4; ++k) tab->wdepth[64 * k + lane] = dep[k];
    unsigned a = w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a |= __shfl_xor(a, d, 64);
    if (lane == 0) tab->anyF = a;
}

// one packet level of layout C (register-index bit K, whole sequences in a lane) under the lane masks mk[s]; ga / gd scale
// the a- and d-slots of the split nodes: after the shears (analysis) or before them (synthesis).
// The plain level (lat_level) advances the odd channel by one pair after every rotation -- a renaming when the sequence sits in
// one lane.  Under a mask a renaming would become moves, so here rotation j pairs u[m] with w[(m + j) mod M] instead (after j
// advances that IS the partner, and the final - (NS - 1) puts every w back where it started): no renaming, and the whole level
// of a sequence is ONE exec region.  The empty volatile asm keeps it a region: if-converted, the compiler computes the level for
// every lane and selects per register (two v_cndmask per coefficient and level, a third of the level's instructions).
// the masks of G consecutive sequences in ONE scalar load, pinned in SGPRs: read one by one where they are tested, every
// sequence starts with its own scalar-cache round trip (s_load_dwordx2 + s_waitcnt lgkmcnt(0): 127 per signal and direction)
typedef unsigned long long lat_u64x2 __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned long long lat_u64x4 __attribute__((ext_vector_type(4), aligned(8)));
typedef unsigned long long lat_u64x8 __attribute__((ext_vector_type(8), aligned(8)));
template <int G> __device__ __forceinline__ void lat_masks(unsigned long long (&mm)[G], const unsigned long long *__restrict__ mk)
{
    static_assert(G == 1 || G == 2 || G == 4 || G == 8, "groups of 1, 2, 4, 8 masks");
    if constexpr (G == 1) {
        mm[0] = mk[0];
        asm volatile("" : "+s"(mm[0]));
    } else if constexpr (G == 2) {
        const lat_u64x2 v = *reinterpret_cast<const lat_u64x2 *>(mk);
        mm[0] = v[0]; mm[1] = v[1];
        asm volatile("" : "+s"(mm[0]), "+s"(mm[1]));
    } else if constexpr (G == 4) {
        const lat_u64x4 v = *reinterpret_cast<const lat_u64x4 *>(mk);
#pragma unroll
        for (int j = 0; j < 4; ++j) mm[j] = v[j];
        asm volatile("" : "+s"(mm[0]), "+s"(mm[1]), "+s"(mm[2]), "+s"(mm[3]));
    } else {
        const lat_u64x8 v = *reinterpret_cast<const lat_u64x8 *>(mk);
#pragma unroll
        for (int j = 0; j < 8; ++j) mm[j] = v[j];
        asm volatile("" : "+s"(mm[0]), "+s"(mm[1]), "+s"(mm[2]), "+s"(mm[3]), "+s"(mm[4]), "+s"(mm[5]), "+s"(mm[6]), "+s"(mm[7]));
    }
}

template <int K, int NS, bool INV, typename V, typename C>
__device__ __forceinline__ void lat_level_cm(V (&x)[64], const C &cf, const unsigned long long *__restrict__ mk, double ga_, double gd_)
{
    typedef typename lat_vtraits<V>::coef CF;
    const CF ga = lat_sgpr((CF)ga_), gd = lat_sgpr((CF)gd_);
    constexpr int NSEQ = 1 << K, M = 32 >> K, S = 1 << K, G = NSEQ < 8 ? NSEQ : 8;
    auto U = [](int s, int m) { return s + ((2 * m) << K); };
    lat_for<NSEQ / G>([&](auto Gc) {
    constexpr int s0 = G * Gc;
    unsigned long long mm[G];
    lat_masks<G>(mm, mk + s0);
    lat_for<G>([&](auto Sc) {
        constexpr int s = s0 + Sc;
        const unsigned long long msk = mm[Sc];              // wave-uniform
        if (!msk) return;
        if (__builtin_amdgcn_inverse_ballot_w64(msk)) {
            // K >= 4 (sequences of 2 and 1 pairs): left to the compiler, which computes every lane and selects -- the rotations
            // of a sequence are one dependent chain, and only the select form lets it interleave the 16 / 32 sequences
            if constexpr (K < 4) asm volatile("");
            if constexpr (!INV) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const CF pj = (CF)cf.p[j], kj = (CF)cf.kap[j];
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const int u = U(s, m), w = U(s, (m + j) % M) + S;
                        x[u] = lat_fma(pj, x[w], x[u]);
                        x[w] = lat_fma(-kj, x[u], x[w]);
                    }
                }
#pragma unroll
                for (int m = 0; m < M; ++m) { x[U(s, m)] = lat_mul(x[U(s, m)], ga); x[U(s, m) + S] = lat_mul(x[U(s, m) + S], gd); }
            } else {
#pragma unroll
                for (int m = 0; m < M; ++m) { x[U(s, m)] = lat_mul(x[U(s, m)], ga); x[U(s, m) + S] = lat_mul(x[U(s, m) + S], gd); }
#pragma unroll
                for (int j = NS - 1; j >= 0; --j) {
                    const CF pj = (CF)cf.p[j], kj = (CF)cf.kap[j];
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const int u = U(s, m), w = U(s, (m + j) % M) + S;
                        x[w] = lat_fma(kj, x[u], x[w]);
                        x[u] = lat_fma(-pj, x[w], x[u]);
                    }
                }
            }
        }
    });
    });
}

constexpr int lat_brev(int v, int bits)
{
    int r = 0;
    for (int k = 0; k < bits; ++k) r |= ((v >> k) & 1) << (bits - 1 - k);
    return r;
}
// stage J of the conditional unshuffle (INV: shuffle): block b of 64 >> J registers (its top J index bits are the path below
// the chunk's node, first branch on top) is deinterleaved where the node of that path is split
template <int J, bool INV>
__device__ __forceinline__ void lat_tree_stage(double (&y)[64], const unsigned long long *__restrict__ mk)
{
    constexpr int NB = 1 << J, B = 64 >> J;
    lat_for<NB>([&](auto Bc) {
        constexpr int b = Bc;
        const unsigned long long m = mk[lat_brev(b, J)];
        if (m) {
            const bool c = __builtin_amdgcn_inverse_ballot_w64(m);
            double t[B];
#pragma unroll
            for (int k = 0; k < B; ++k) t[k] = y[b * B + k];
            lat_for<B>([&](auto Kc) {
                constexpr int k = Kc;
                constexpr int o = INV ? ((k & 1) * (B / 2) + (k >> 1)) : (2 * (k % (B / 2)) + k / (B / 2));
                if constexpr (o != k) y[b * B + k] = c ? t[o] : t[k];
            });
        }
    });
}

template <int NS, int WPE, int SH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpt_tree_f64(
    const double *__restrict__ x, double *__restrict__ y, int L, int last_sig, WxLatW cw, const WxLatTreeTab *__restrict__ tab)
{
    static_assert(SH >= 0 && SH <= 2, "4096, 2048 or 1024 samples");
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    constexpr int N = 4096 >> SH, L6 = 6 - SH;
    const int sig0 = min((int)blockIdx.x << SH, last_sig);
    const double *xs = x + (int64_t)sig0 * N;
    double *ys = y + (int64_t)sig0 * N;
    const WxLat &cf = cw.c;
    unsigned wd[6];
#pragma unroll
    for (int l = 1; l < L6; ++l) wd[l] = tab->words[64 * l + lane];
    // levels above the chunk depth: every node, leaves of depth l leave after level l (as in k_lat_wpt_tree_f64)
#define WX_LVL(LAY, KK, HH, REG, BIT)                                                           \
    if constexpr (BIT >= SH) {                                                                  \
        constexpr int l = BIT - SH + 1;                                                         \
        lat_level<KK, HH, NS, false>(REG, cf);                                                  \
        if constexpr (l < L6) {                                                                 \
            const unsigned aw = (unsigned)__builtin_amdgcn_readfirstlane((int)tab->any[l]);     \
            if (aw) lat_emit<LAY, l + 16 * SH, true>(REG, lds0, ys, lane, cw, (unsigned)N, wd[l], aw); \
            if (L <= l) return;                                                                 \
        }                                                                                       \
    }
    double c[64];
    if constexpr (SH < 2) {
        double a[64], bb[64];
        lat_absorb<0, 16 * SH>(a, lds0, xs, lane, cw);
        WX_LVL(0, 0, 6, a, 0) WX_LVL(0, 1, 6, a, 1)
        lat_t2(a, bb, lds0, lane);
        WX_LVL(2, 0, 4, bb, 2) WX_LVL(2, 1, 4, bb, 3) WX_LVL(2, 2, 4, bb, 4) WX_LVL(2, 3, 4, bb, 5)
        lat_t3(bb, c, lds0, lane);
    } else {
        double bb[64];
        lat_absorb<2, 16 * SH>(bb, lds0, xs, lane, cw);
        WX_LVL(2, 0, 4, bb, 2) WX_LVL(2, 1, 4, bb, 3) WX_LVL(2, 2, 4, bb, 4) WX_LVL(2, 3, 4, bb, 5)
        lat_t3(bb, c, lds0, lane);
    }
#undef WX_LVL
    // levels below the chunk depth: lane-masked, normalised per level (forward: gl[1] = g, g2 = g^-2)
    const double g = cw.gl[1], ginv = cw.c.g2 * cw.gl[1];
    const unsigned long long *mk = tab->masks;
    if (L > L6 + 0 && tab->stage_any[0]) lat_level_cm<0, NS, false>(c, cf, mk + 0, g, ginv);
    if (L > L6 + 1 && tab->stage_any[1]) lat_level_cm<1, NS, false>(c, cf, mk + 1, g, ginv);
    if (L > L6 + 2 && tab->stage_any[2]) lat_level_cm<2, NS, false>(c, cf, mk + 3, g, ginv);
    if (L > L6 + 3 && tab->stage_any[3]) lat_level_cm<3, NS, false>(c, cf, mk + 7, g, ginv);
    if (L > L6 + 4 && tab->stage_any[4]) lat_level_cm<4, NS, false>(c, cf, mk + 15, g, ginv);
    if (L > L6 + 5 && tab->stage_any[5]) lat_level_cm<5, NS, false>(c, cf, mk + 31, g, ginv);
    // in-place dilated order -> packet order of the lane's chunk
    if (tab->stage_any[0]) lat_tree_stage<0, false>(c, mk + 0);
    if (tab->stage_any[1]) lat_tree_stage<1, false>(c, mk + 1);
    if (tab->stage_any[2]) lat_tree_stage<2, false>(c, mk + 3);
    if (tab->stage_any[3]) lat_tree_stage<3, false>(c, mk + 7);
    if (tab->stage_any[4]) lat_tree_stage<4, false>(c, mk + 15);
    const unsigned awF = (unsigned)__builtin_amdgcn_readfirstlane((int)tab->anyF);
    if (awF) lat_emit<6, L6 + 16 * SH, true>(c, lds0, ys, lane, cw, (unsigned)N, tab->wordsF[lane], awF);
}

template <int NS, int WPE, int SH, bool THR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_iwpt_tree_f64(
    const double *__restrict__ xw, double *__restrict__ y, int L, int last_sig, unsigned in_stride, unsigned col_stride, WxLatW cw,
    const WxLatTreeTab *__restrict__ tab, WxThreshArg thr)
{
    static_assert(SH >= 0 && SH <= 2, "4096, 2048 or 1024 samples");
    __shared__ double lds[WX_LAT_LDS];
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int lane = threadIdx.x;
    constexpr int N = 4096 >> SH, L6 = 6 - SH;
    const int sig0 = min((int)blockIdx.x << SH, last_sig);
    const double *xs = xw + (int64_t)sig0 * in_stride;
    double *ys = y + (int64_t)sig0 * N;
    const WxLat &cf = cw.c;
    // THR: the threshold of denoise() on the absorbed leaves (its own instantiation: the plain inverse has no register to spare)
    LatThr lt;
    const LatThr *th = nullptr;
    if constexpr (THR) {
        lt.kind = thr.kind;
        lt.lo = thr.lo;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            lt.tt[k] = k < (1 << SH) ? reinterpret_cast<const double *>(thr.t)[thr.per_signal ? sig0 + k : 0] * thr.scale : 0.0;
        th = &lt;
    }
    lat_v2<double> pv[16];
    double c[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) c[r] = 0.0;
    const unsigned awF = (unsigned)__builtin_amdgcn_readfirstlane((int)tab->anyF);
    if (awF) {
        // every leaf of depth >= 6 - SH in one absorb; out of a packet table each piece sits in the column of its depth
        const unsigned wF = tab->wordsF[lane];
        unsigned dep[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) dep[k] = tab->wdepth[64 * k + lane];
        const unsigned *dp = col_stride ? dep : nullptr;
        lat_absorb_fetch01<6, L6 + 16 * SH>(pv, xs, lane, in_stride, wF, dp, col_stride);
        lat_absorb<6, L6 + 16 * SH, true, true>(c, lds0, xs, lane, cw, in_stride, wF, awF, pv, dp, col_stride, th);
        const unsigned long long *mk = tab->masks;
        if (tab->stage_any[4]) lat_tree_stage<4, true>(c, mk + 15);
        if (tab->stage_any[3]) lat_tree_stage<3, true>(c, mk + 7);
        if (tab->stage_any[2]) lat_tree_stage<2, true>(c, mk + 3);
        if (tab->stage_any[1]) lat_tree_stage<1, true>(c, mk + 1);
        if (tab->stage_any[0]) lat_tree_stage<0, true>(c, mk + 0);
        // synthesis: gl[1] = 1 / g, g2 = g^2 -- the a-slot of a split node enters as a / g, the d-slot as d g
        const double ga = cw.gl[1], gd = cw.c.g2 * cw.gl[1];
        if (L > L6 + 5 && tab->stage_any[5]) lat_level_cm<5, NS, true>(c, cf, mk + 31, ga, gd);
        if (L > L6 + 4 && tab->stage_any[4]) lat_level_cm<4, NS, true>(c, cf, mk + 15, ga, gd);
        if (L > L6 + 3 && tab->stage_any[3]) lat_level_cm<3, NS, true>(c, cf, mk + 7, ga, gd);
        if (L > L6 + 2 && tab->stage_any[2]) lat_level_cm<2, NS, true>(c, cf, mk + 3, ga, gd);
        if (L > L6 + 1 && tab->stage_any[1]) lat_level_cm<1, NS, true>(c, cf, mk + 1, ga, gd);
        if (L > L6 + 0 && tab->stage_any[0]) lat_level_cm<0, NS, true>(c, cf, mk + 0, ga, gd);
    }
    // the levels above the chunk depth: as in k_lat_iwpt_tree_f64 (depth 6 - SH itself has been taken in already)
#define WX_ILVL(LAY, KK, HH, REG, BIT)                                                          \
    if constexpr (BIT >= SH) {                                                                  \
        constexpr int l = BIT - SH + 1;                                                         \
        if (L >= l) {                                                                           \
            if constexpr (l < L6) {                                                             \
                const unsigned aw = (unsigned)__builtin_amdgcn_readfirstlane((int)tab->any[l]); \
                if (aw) {                                                                       \
                    const unsigned wl = tab->words[64 * l + lane];                              \
                    if (L == l) lat_absorb_fetch01<LAY, l + 16 * SH>(pv, xs + (size_t)l * col_stride, lane, in_stride, wl); \
                    lat_absorb<LAY, l + 16 * SH, true, true>(REG, lds0, xs + (size_t)l * col_stride, lane, cw, in_stride, wl, aw, pv, \
                                                             nullptr, 0u, th);                  \
                }                                                                               \
            }                                                                                   \
            if constexpr (l > 1 && l <= L6) {                                                   \
                if (__builtin_amdgcn_readfirstlane((int)tab->any[l - 1]))                       \
                    lat_absorb_fetch01<lat_tree_lay(BIT - 1), l - 1 + 16 * SH>(pv, xs + (size_t)(l - 1) * col_stride, lane, in_stride, \
                                                                               tab->words[64 * (l - 1) + lane]); \
            }                                                                                   \
            lat_level<KK, HH, NS, true>(REG, cf);                                               \
        }                                                                                       \
    }
    double bb[64];
    lat_t3i(c, bb, lds0, lane);
    WX_ILVL(2, 3, 4, bb, 5) WX_ILVL(2, 2, 4, bb, 4) WX_ILVL(2, 1, 4, bb, 3) WX_ILVL(2, 0, 4, bb, 2)
    if constexpr (SH >= 2) {
        lat_emit<2, 16 * SH>(bb, lds0, ys, lane, cw);
    } else {
        double a[64];
        lat_t2i(bb, a, lds0, lane);
        WX_ILVL(0, 1, 6, a, 1) WX_ILVL(0, 0, 6, a, 0)
        lat_emit<0, 16 * SH>(a, lds0, ys, lane, cw);
    }
#undef WX_ILVL
}

}  // namespace

bool wx_lattice_factor(const WxFilt &filt, int L, bool inverse, WxLat *out);      // wx_lattice.hip
