#pragma once
// wx_lattice_8k.h -- full-tree wpt / iwpt of 8192-sample Float64 signals in ONE pass over the data.
//
// Until round 5 an 8192-sample signal took a tiled pass for its first level (HBM: read n, write n) and then the 4096-sample lattice kernel
// on the two children (read n, write n): 0.34 of the HBM peak.  Here a workgroup of TWO wavefronts takes a signal, wavefront c its child c
// (0 = approximation, 1 = detail): it computes the child's 4096 samples in the direct form of the reference's step
// (dwt/dwt_one_level.jl:79-107: a[i] = sum_j q[j] v[2i + j], d[i] = sum_j (-1)^j q[j] v[2i + 1 - j], periodic) straight into the L0
// register arrangement of the lattice kernel (lat_fwd_from_l0) -- the parent is staged through the wavefront's own 8.6 KiB LDS window in
// eight chunks of 1024 samples (+ F - 2 of halo), both children read it (the second read hits L1 / L2: same workgroup) -- and goes on with
// the child's L - 1 lattice levels.  HBM sees the signal once each way; the price is the direct-form level: F multiply-adds per child
// sample against the lattice's F / 2 (+ 8 % vector work at db4, L = 13) and 160 LDS reads.
// Inverse: each wavefront runs the lattice synthesis of its child to the L0 arrangement, the two exchange chunks of 512 child samples
// through a shared LDS buffer and each writes half of the parent chunk (dwt_one_level.jl:192-223:
// v[2k] = sum_m q[2m] a[k-m] - q[2m+1] d[k+m], v[2k+1] = sum_m q[2m+1] a[k-m] + q[2m] d[k+m]).
#include "wx_lattice_dev.h"

namespace {

typedef lat_d2 __attribute__((address_space(3))) *lat8_l2p;
// 16-byte LDS accesses as volatile asm, like the exchanges of wx_lattice_dev.h: left to the compiler, the window reads of a whole chunk were
// hoisted above the arithmetic and spilled (88 scratch loads per wavefront; the first build of the forward kernel ran at 1.16 ms per GiB)
template <int OFF> __device__ __forceinline__ lat_d2 lat8_rd128(unsigned addr)
{
    lat_d2 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ void lat8_wr128(unsigned addr, lat_d2 v)
{
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lat8_landed(lat_d2 &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory"); }
// the parent is read by both children of the workgroup: a plain (temporal) load, so that the second read hits L2
__device__ __forceinline__ lat_d2 lat8_ld2(const double __attribute__((address_space(1))) *p)
{
    typedef const lat_d2 __attribute__((address_space(1))) *P;
    return *(P)p;
}

// child c of signal `xs` (8192 samples) into the L0 registers
template <int F>
__device__ __forceinline__ void lat8k_child_l0(lat_d2 (&r)[32], const double *__restrict__ xs, unsigned lds0, int lane, int child,
                                               const WxFilt &filt)
{
    // taps over the window W[t] = parent[2 i + t] (a) / parent[2 i + 2 - F + t] (d), t = 0 .. F-1: a: q[t]; d: (-1)^(F-1-t) q[F-1-t]
    double tp[F];
#pragma unroll
    for (int t = 0; t < F; ++t) {
        const double ta = filt.q[t], td = ((F - 1 - t) & 1) ? -filt.q[F - 1 - t] : filt.q[F - 1 - t];
        tp[t] = child ? td : ta;                                     // `child` is wave-uniform (scalar select)
    }
    constexpr int HALO = F - 2;                                       // parent samples beyond the chunk (to the right for a, left for d)
    // local index 0 of the staged window <-> parent (1024 hi3 - (child ? HALO : 0)); the chunk's own samples start at local H0
    const int H0 = child ? HALO : 0;
    const unsigned wst = lds0 + 8u * (unsigned)(H0 + 2 * lane);       // the chunk's 16-byte pieces: + 1024 k bytes
    const unsigned hst = lds0 + 8u * (unsigned)((child ? 0 : 1024) + 2 * lane);   // the halo's pieces (lanes < HALO / 2)
    const unsigned rd0 = lds0 + 8u * (unsigned)(128 * (lane >> 3) + 4 * (lane & 7));  // window of the lane's pair at f = 0
    lat_d2 nx[8], nh;
    auto fetch = [&](int hi3) {
        const double *cp = xs + 1024 * hi3;
#pragma unroll
        for (int k = 0; k < 8; ++k) nx[k] = lat8_ld2(lat_sbase(cp + 128 * k) + 2u * lane);
        if (HALO > 0) {
            // a: parent 1024 (hi3 + 1) + 2 lane; d: parent 1024 hi3 - HALO + 2 lane  (mod 8192)
            const int hp = (child ? 1024 * hi3 - HALO : 1024 * (hi3 + 1)) & 8191;
            nh.x = nh.y = 0.0;
            if (2 * lane < HALO) nh = lat8_ld2(lat_sbase(xs + hp) + 2u * lane);
        }
    };
    fetch(0);
    lat_for<8>([&](auto H3) {
        constexpr int hi3 = H3;
        lat_sync();                                                   // the previous chunk's window reads are done
        lat_for<8>([&](auto Kc) { lat8_wr128<1024 * Kc>(wst, nx[Kc]); });
        if (HALO > 0 && 2 * lane < HALO) lat8_wr128<0>(hst, nh);
        if (hi3 < 7) fetch(hi3 + 1);
        lat_sync();
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            // pair (c, c + 1), c = 16 f + 64 (lane >> 3) + 2 (lane & 7): windows W[0 .. F-1] and W[2 .. F+1] from local 2 c
            lat_d2 w[(F + 2) / 2];
            lat_for<(F + 2) / 2>([&](auto Tc) { w[Tc] = lat8_rd128<8 * (32 * f + 2 * Tc)>(rd0); });
            // all of this pair's reads have landed (tie every piece to the wait)
            lat_for<(F + 2) / 2>([&](auto Tc) { lat8_landed(w[Tc]); });
            double o0 = 0.0, o1 = 0.0;
#pragma unroll
            for (int t = 0; t < F; ++t) {
                const double a0 = (t & 1) ? w[t >> 1].y : w[t >> 1].x;
                const double a1 = (t & 1) ? w[(t >> 1) + 1].y : w[(t >> 1) + 1].x;
                o0 = fma(tp[t], a0, o0);
                o1 = fma(tp[t], a1, o1);
            }
            // the two sums are due HERE: the next pair's (volatile) reads may not start before them, or the compiler defers all the
            // arithmetic of the chunk loop to its end and spills every window it has read
            asm volatile("" : "+v"(o0), "+v"(o1));
            r[4 * hi3 + f].x = o0;
            r[4 * hi3 + f].y = o1;
        });
    });
    lat_sync();
}

template <int NS, int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpt8k_f64(
    const double *__restrict__ x, double *__restrict__ y, int L1 /* levels of the children: L - 1 in 6 .. 12 */, int64_t batch, WxLat cf,
    WxFilt filt)
{
    __shared__ double lds2[2][WX_LAT_LDS];
    const int child = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds2[child];
    const int64_t sig = blockIdx.x;
    lat_d2 r[32];
    lat8k_child_l0<2 * NS>(r, x + sig * 8192, lds0, lane, child, filt);
    lat_fwd_from_l0<NS>(r, lds0, lane, L1, cf, y + sig * 8192 + 4096 * child);
}

// the last synthesis level of an 8192-sample signal from its two children in the L0 arrangement (one per wavefront of the workgroup):
// chunks of 512 child samples through the shared buffer xch, each wavefront writes half of the parent chunk (idwt_step!, direct form)
template <int F>
__device__ __forceinline__ void lat8k_synth(lat_d2 (&o)[32], double (&xch)[2][2][512 + 16], int child, int lane, double *__restrict__ ys,
                                            const WxFilt &filt)
{
    constexpr int HB = F / 2 - 1;                                     // halo of a child chunk: a to the left, d to the right
    double q0[F];
#pragma unroll
    for (int t = 0; t < F; ++t) q0[t] = filt.q[t];
    const int g = lane >> 3, j = lane & 7;
    lat_for<8>([&](auto H3) {
        constexpr int hi3 = H3;
        double *mine = xch[hi3 & 1][child] + (child ? 0 : HB);        // local child index cl at mine[cl]
        lat_for<4>([&](auto Fq) {
            constexpr int f = Fq;
            double *p = mine + 16 * f + 64 * g + 2 * j;
            p[0] = o[4 * hi3 + f].x;
            p[1] = o[4 * hi3 + f].y;
        });
        // halos: a needs the last HB samples of the previous chunk (f = 3, g = 7, cl = 512 - HB .. 511), d the first HB of the next
        // (f = 0, g = 0, cl = 0 .. HB - 1)
        if (HB > 0 && !child && g == 7) {
            constexpr int h = (hi3 + 7) & 7;
            const int cl = 16 * 3 + 64 * 7 + 2 * j;                   // 496 + 2 j, + 1
            if (cl >= 512 - HB) mine[cl - 512] = o[4 * h + 3].x;
            if (cl + 1 >= 512 - HB) mine[cl + 1 - 512] = o[4 * h + 3].y;
        }
        if (HB > 0 && child && g == 0) {
            constexpr int h = (hi3 + 1) & 7;
            const int cl = 2 * j;
            if (cl < HB) mine[512 + cl] = o[4 * h + 0].x;
            if (cl + 1 < HB) mine[512 + cl + 1] = o[4 * h + 0].y;
        }
        __syncthreads();
        const double *ab = xch[hi3 & 1][0] + HB, *db = xch[hi3 & 1][1];
        // wavefront w writes the pairs k = 256 w + 64 t + lane of the chunk: 16 bytes per lane, 1 KiB per instruction
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = 256 * child + 64 * t + lane;
            double v0 = 0.0, v1 = 0.0;
#pragma unroll
            for (int m = 0; m < F / 2; ++m) {
                const double am = ab[k - m], dm = db[k + m];
                v0 = fma(q0[2 * m], am, v0);
                v0 = fma(-q0[2 * m + 1], dm, v0);
                v1 = fma(q0[2 * m + 1], am, v1);
                v1 = fma(q0[2 * m], dm, v1);
            }
            lat_d2 ov;
            ov.x = v0; ov.y = v1;
            lat_st2(lat_sbase(ys + 1024 * hi3 + 512 * child + 128 * t) + 2u * lane, ov);
        }
        // the next chunk goes to the other buffer; the one after next waits for this chunk's readers at its own barrier
    });
}

// ---- inverse ----
template <int NS, int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_iwpt8k_f64(
    const double *__restrict__ xw, double *__restrict__ y, int L1, int64_t batch, int64_t in_stride, WxLat cf, WxFilt filt)
{
    __shared__ double lds2[2][WX_LAT_LDS];
    __shared__ __attribute__((aligned(16))) double xch[2][2][512 + 16];   // [chunk parity][child][HB + 512 (a) | 512 + HB (d)]
    const int child = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds2[child];
    const int64_t sig = blockIdx.x;
    lat_d2 o[32];
    lat_inv_to_l0<NS>(xw + sig * in_stride + 4096 * child, lds0, lane, L1, cf, [&](auto Fq, lat_d2 (&oo)[8]) {
        constexpr int f = decltype(Fq)::value;
        lat_for<8>([&](auto Hq) { o[4 * Hq + f] = oo[Hq]; });
    });
    lat8k_synth<2 * NS>(o, xch, child, lane, y + sig * 8192, filt);
}

}  // namespace
