#pragma once
// wx_lattice2d.h -- 2-D full-tree packets of 512 x 512 Float32 images (depth 6; HB = 1: 256 x 256 at depth 5, HB = 2:
// 1024 x 1024 at depth 7, see L2G below) as two applications of ONE kernel: a lattice transform down 32 contiguous columns in
// registers whose result is stored TRANSPOSED.  The text below describes the 512 x 512 geometry.
//
// Reference semantics: 2-D wpt / iwpt by level (DWT.jl:500-548, 662-710 over dwt/dwt_one_level.jl:319-354, 401-436); a
// full tree of depth L is separable: the 1-D packet transform down every column, then along every row (§4.5 of DESIGN.md).
// The round-1 path runs the row pass out of LDS strips (LDS-issue bound, 2.5 TB/s) and the column pass one wavefront per
// column (latency bound).  Here both passes are the same contiguous-signal kernel: it reads the columns of the image
// (contiguous), transforms them with the rotation lattice of wx_lattice.hip -- v_pk_fma_f32 on pairs of adjacent columns,
// 16 columns = 4096 Float32 pairs per wavefront, the same register layouts A and B and the same intra-wavefront LDS
// exchanges as the Float64 kernel -- and writes Z[j + 512 o(i)]: position o(i) of the packet order becomes the column,
// the column index j the contiguous dimension.  The second application transforms the former rows and restores the
// orientation.  A workgroup of W = 4 wavefronts covers 64 columns = a 256-byte run of every row of the transposed image;
// the transposition goes through LDS in four rounds of 128 rows (two register bits fixed per round).
//
// What bounds a pass is its memory access shape, not arithmetic (tools/dbg/l2d_pattern.hip times the shapes without any
// arithmetic on 4096 images: the round-2 shape, 128-byte runs per row, 1.67 ms; 256-byte runs with the non-temporal hint
// 1.51 ms; a plain copy 1.45 ms; the kernel without its global accesses 0.72 ms).  Hence (round 3): W = 4 instead of 2,
// non-temporal loads and stores, two wavefronts per SIMD without the 13-22 spilled registers of the three-wavefront build
// (their scratch traffic was the 1.17 x of round 2's PMC bytes), and the intermediate image -- which nobody outside the
// library sees -- in a BLOCKED layout (l2_src_off) that makes the first pass's stores and the second pass's loads
// contiguous: only the last store of a transform is a strided one.  3.39 -> 2.95 ms per direction on config 4.  Every even
// filter length up to 20 taps (round 2: 4 and 8) -- db8 3.5 / 3.3 ms (the levels start to show), Haar / coif2 as db4.
//
// Halo: the columns of a wavefront are independent periodic sequences of 512 samples: in layout A (reg i[5:0]) the
// neighbouring chunk is the next of 8 lanes (cyclic: two DPP moves and a select), in layout B (reg i[7:2]) the other of 2
// lanes (quad_perm).  Depth 6 needs levels on index bits 0..5 only: A takes 0-1, B takes 2-5.
#include "wx_common.h"
#include "wx_kernels.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <utility>

bool wx_lattice_coeffs(const WxFilt &filt, int L, bool inverse, double *p, double *kap, double *g0, double *g2);
const void *wx_const_upload(const void *host, size_t bytes, hipStream_t st, bool have_stream);   // wx_host.hip: content-keyed constant cache

#define WX_L2_MAXS 10
#define WX_L2_WIN 1104        // 8-byte slots of one wavefront's exchange window
#ifndef WX_L2D_G
#define WX_L2D_G 16            // LDS reads in flight per wait of an exchange (8 or 16)
#endif
#ifndef WX_L2D_W
#define WX_L2D_W 4             // wavefronts per workgroup = 16 W columns = a 64 W byte run of every transposed row
#endif
#ifndef WX_L2D_NT
#define WX_L2D_NT 1            // non-temporal hint on the loads and stores (every byte is touched once)
#endif
#ifndef WX_L2D_WPE
#define WX_L2D_WPE 2           // wavefronts per SIMD the kernels are built for (3: 168 registers, 13-22 of them spilled)
#endif

struct WxLat2 {
    float p[WX_L2_MAXS];
    float kap[WX_L2_MAXS];
    float g0, g2;
};
// The levels below the lattice's (nodes of 8 samples: up to three more) as ONE orthogonal 8 x 8 matrix per node: m[r][c] = coefficient r
// (wpt layout of a full tree of depth e on 8 samples) from sample c, periodic inside the node like dwt_step! on a node of 8, 4, 2
// samples (dwt/dwt_one_level.jl:94-105).  The forward pass applies it in its store phase, the inverse pass its transpose in the load phase.
struct WxL2M {
    float m[8][8];
    int e;                                // 0: the lattice's depth exactly, no matrix
};

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1))) *l2_gc;
typedef float __attribute__((address_space(1))) *l2_gm;

__device__ __forceinline__ l2_gc l2_sbase(const float *p)
{
    l2_gc g = (l2_gc)p;
    asm("" : "+s"(g));
    return g;
}
__device__ __forceinline__ l2_gm l2_sbase(float *p)
{
    l2_gm g = (l2_gm)p;
    asm("" : "+s"(g));
    return g;
}
__device__ __forceinline__ f4 l2_ld(l2_gc p)
{
    const f4 __attribute__((address_space(1))) *q = (const f4 __attribute__((address_space(1))) *)p;
    return WX_L2D_NT ? __builtin_nontemporal_load(q) : *q;
}
__device__ __forceinline__ void l2_st(l2_gm p, f4 v)
{
    f4 __attribute__((address_space(1))) *q = (f4 __attribute__((address_space(1))) *)p;
    if (WX_L2D_NT) __builtin_nontemporal_store(v, q);
    else *q = v;
}
// Memory flavour MF of a pass.  0: non-temporal global accesses on both sides (one launch per pass).  The fused kernel (k_lat2d_fused_f32)
// hands the intermediate image from pass to pass through a ring that is rewritten in place every few hundred microseconds by workgroups
// on any XCD, so the ring side is accessed at agent scope or wider -- 1 (first pass): image in non-temporal, ring out through sc1 buffer
// stores (written through to the memory side, no dirty line stays in the writer's L2); 2 (second pass): ring in through sc0 sc1 buffer
// loads (never served by this CU's L1; the XCD L2s are kept coherent for such stores: tools/dbg/ring_stale.hip), image out non-temporal.
// Buffer instructions because the builtins carry the cache policy AND stay visible to the compiler's wait-count / hazard passes (an
// inline-asm sc1 store followed by a write of its data registers lost 3.6 % of the words: profiles/r06_cfg4_fused.md).
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t l2_rsrc(const float *p)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7fffffff, 0x00020000);
}
template <int MF> __device__ __forceinline__ f4 l2_ldm(const float *img, __amdgpu_buffer_rsrc_t rs, int soff, unsigned lo)
{
    if constexpr (MF == 2) return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, 4u * lo, 4 * soff, 17));
    else return l2_ld(l2_sbase(img + soff) + lo);
}
// The store's scalar offset is folded into the vector offset: with a REGISTER soffset the compiler assumes that a VALU write of the data
// registers right behind a 16-byte buffer store is safe (GCNHazardRecognizer: "this hazard only exists if the instruction is not using a
// register in the soffset field") and schedules one there; on gfx950 that write reached the store -- one store instruction of the
// 256 x 256 forward kernel wrote garbage in every run, others under load only (profiles/r06_cfg4_fused.md).  With soffset 0 the
// recogniser inserts the wait state itself.
template <int MF> __device__ __forceinline__ void l2_stm(float *img, __amdgpu_buffer_rsrc_t rs, int soff, unsigned lo, f4 v)
{
    if constexpr (MF == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, 4u * (lo + (unsigned)soff), 0, 16);
    else l2_st(l2_sbase(img + soff) + lo, v);
}
// The intermediate image between the two passes is the library's own: instead of the plain transposed image
// Z[j + 512 o] the first pass writes it in blocks of 16 W source columns, Zb[(j / 16 W) * 512 * 16 W + o * 16 W + j % 16 W],
// so that a workgroup's 512 runs of 64 W bytes are one contiguous 32 W KiB block (BS), and the second pass reads its
// columns o out of those blocks (BL): sample j of column o at (j / 64) * WX_L2D_BLK + 64 o + j % 64.
// HB = 1 (256 x 256 images, depth 5): a register column of 512 samples is the same column of TWO consecutive images --
// index bit 8 selects the image, the levels and their periodic halos stay inside a half (the cyclic neighbours of layout A
// are the 4 lanes of a quad, layout B holds whole sequences), everything else is the 512-row kernel.
template <int HB> struct L2G {
    static constexpr int R = HB == 2 ? 1024 : (512 >> HB);  // rows = columns of an image (HB = 2: 1024, see below)
    static constexpr int RB = HB == 2 ? 10 : 9 - HB;        // bits of a row index
    static constexpr int LD = HB == 2 ? 7 : 6 - HB;         // levels
    static constexpr int IB = HB == 1 ? 1 : 0;              // image-select bits of a register column
    static constexpr int NPW = HB == 2 ? 4 : 8;             // column pairs per wavefront
    static constexpr int BW = 2 * NPW * WX_L2D_W;           // columns per workgroup = block width of the intermediate
    static constexpr int BLK = R * BW;                      // elements of one block of the intermediate
    static constexpr int IMG = R * R;
    static constexpr int HA = HB == 2 ? 4 : (HB == 1 ? 2 : 3);   // halo of layout A: 16 lanes of a row / a quad / 8 lanes
    static constexpr int HBQ = HB == 2 ? 2 : (HB == 1 ? 0 : 1);  // halo of layout B: a quad / none / lane ^ 1
};
// HB = 2 (1024 x 1024 images, depth 7): a wavefront holds 8 columns of 1024 rows instead of 16 of 512 -- the same 64 x 64
// float2 registers with index bit 9 where the top column-pair bit was: layout A lane = i[9:6] | cp << 4 (the 16 chunks of a
// column are the 16 lanes of a DPP row), layout B lane = (i[9:8] | cp << 2) | i[1:0] << 4 (the 4 chunks are a quad), one
// more level on register bit 4 of layout B.  The exchanges T1, T2 do not change (they never look at what the lane bits mean).
// element offset of sample smp9 (9 bits: image-select bits above the row bits) of column col
template <bool BL, int HB> __device__ __forceinline__ int64_t l2_src_off(int col, int smp9)
{
    typedef L2G<HB> G;
    const int im = smp9 >> G::RB, smp = smp9 & (G::R - 1);
    if constexpr (BL) return (int64_t)im * G::IMG + (int64_t)(smp / G::BW) * G::BLK + (int64_t)col * G::BW + (smp % G::BW);
    else return (int64_t)im * G::IMG + (int64_t)col * G::R + smp;
}
// lane part of a load address: column half h, samples 64 i6 + 32 i5 + 4 sub
template <bool BL, int HB> __device__ __forceinline__ unsigned l2_lane_off(int h, int i6, int i5, int sub)
{
    typedef L2G<HB> G;
    if constexpr (!BL) return (unsigned)G::R * h + 64u * i6 + 32u * i5 + 4u * sub;
    else if constexpr (G::BW == 64) return 64u * h + (unsigned)G::BLK * i6 + 32u * i5 + 4u * sub;
    else return 32u * h + (unsigned)G::BLK * (2 * i6 + i5) + 4u * sub;          // blocks of 32 columns
}
template <int... I, typename F> __device__ __forceinline__ void l2_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void l2_for(F &&f)
{
    l2_for_impl(std::make_integer_sequence<int, N>{}, f);
}
template <int OFF> __device__ __forceinline__ void l2_wr32(unsigned addr, float v)
{
    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void l2_wr64(unsigned addr, f2 v)
{
    const double d = __builtin_bit_cast(double, v);
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(d), "n"(OFF) : "memory");
}
// the value of a pending LDS read stays an untouched 64-bit register pair until the wait: turning it into a float2 first
// lets the compiler move its halves (it does not know the data has not landed) -- only l2_wait16 hands out float2
template <int OFF> __device__ __forceinline__ double l2_rd64(unsigned addr)
{
    double d;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
    return d;
}
__device__ __forceinline__ void l2_wait8(double &a, double &b, double &c, double &d, double &e, double &f, double &g, double &h)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "memory");
}
__device__ __forceinline__ void l2_waitn(double (&x)[16])
{
    l2_wait8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
    l2_wait8(x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15]);
}
__device__ __forceinline__ void l2_waitn(double (&x)[8]) { l2_wait8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]); }
// keeps a coefficient pair in vector registers (see the note at l2_level)
__device__ __forceinline__ void l2_vgpr(f2 &v)
{
    double d = __builtin_bit_cast(double, v);
    asm volatile("" : "+v"(d));
    v = __builtin_bit_cast(f2, d);
}
// workgroup barrier after inline-asm LDS stores: the compiler does not count them, so the wait that makes them visible to
// the other wavefront is explicit (without it the transposed store raced: one image in a few hundred came out wrong)
__device__ __forceinline__ void l2_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    __syncthreads();
}
template <int CTRL> __device__ __forceinline__ int l2_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// value held by the lane that owns the chunk D places further along the same column; HALO 3: the chunks of a column are
// the 8 lanes that share lane >> 3 (cyclic); HALO 1: the 2 lanes lane, lane ^ 1
template <int HALO, int D> __device__ __forceinline__ f2 l2_nbr(f2 v, bool edge_hi, bool edge_lo)
{
    if constexpr (D == 0 || HALO == 0) return v;
    else {
        // the element travels as one 64-bit register pair (two 32-bit DPP moves), exactly like the Float64 kernel's halo:
        // per-component code on the float2 was merged by the compiler into one move for both halves
        const double dv = __builtin_bit_cast(double, v);
        const int lo = __double2loint(dv), hi = __double2hiint(dv);
        int rlo, rhi;
        if constexpr (HALO == 1) {
            if constexpr ((D & 1) == 0) return v;
            rlo = l2_dpp<0xB1>(lo);                                   // quad_perm [1,0,3,2]
            rhi = l2_dpp<0xB1>(hi);
        } else if constexpr (HALO == 2) {
            // the four chunks of a sequence are the lanes of a quad (cyclic): lane q takes lane (q + D) mod 4
            constexpr int E = ((D % 4) + 4) % 4;
            if constexpr (E == 0) return v;
            constexpr int SEL = ((0 + E) & 3) | (((1 + E) & 3) << 2) | (((2 + E) & 3) << 4) | (((3 + E) & 3) << 6);
            rlo = l2_dpp<SEL>(lo);
            rhi = l2_dpp<SEL>(hi);
        } else if constexpr (HALO == 4) {
            // the sixteen chunks of a column are the lanes of a DPP row (cyclic): row_ror:n, lane i takes lane i - n (mod 16)
            static_assert(D > -16 && D < 16, "row rotations");
            rlo = l2_dpp<0x120 + ((16 - D) & 15)>(lo);
            rhi = l2_dpp<0x120 + ((16 - D) & 15)>(hi);
        } else {
            static_assert(D == 1 || D == -1, "layout A moves one chunk");
            if constexpr (D > 0) {
                // lanes 0..6 of a group of 8 take lane + 1 (row_shl:1), lane 7 takes lane - 7 (row_shr:7)
                const int alo = l2_dpp<0x101>(lo), ahi = l2_dpp<0x101>(hi);
                const int blo = l2_dpp<0x117>(lo), bhi = l2_dpp<0x117>(hi);
                rlo = edge_hi ? blo : alo;
                rhi = edge_hi ? bhi : ahi;
            } else {
                const int alo = l2_dpp<0x111>(lo), ahi = l2_dpp<0x111>(hi);
                const int blo = l2_dpp<0x107>(lo), bhi = l2_dpp<0x107>(hi);
                rlo = edge_lo ? blo : alo;
                rhi = edge_lo ? bhi : ahi;
            }
        }
        return __builtin_bit_cast(f2, __hiloint2double(rhi, rlo));
    }
}

// one packet level on register-index bit K (see lat_level of wx_lattice.hip); both halves of an element (two adjacent
// columns) take the same rotation: v_pk_fma_f32
template <int K, int HALO, int NS, bool INV, typename CF> __device__ __forceinline__ void l2_level(f2 (&x)[64], const CF &cf, int lane)
{
    constexpr int NSEQ = 1 << K, M = 32 >> K, S = 1 << K;
    auto U = [](int s, int m) { return s + ((2 * m) << K); };
    const bool edge_hi = (lane & 7) == 7, edge_lo = (lane & 7) == 0;
    auto shift = [&](auto SHc) {
        constexpr int SH = decltype(SHc)::value;
        if constexpr (SH != 0) {
#pragma unroll
            for (int s = 0; s < NSEQ; ++s) {
                f2 old[M];
#pragma unroll
                for (int m = 0; m < M; ++m) old[m] = x[U(s, m) + S];
                l2_for<M>([&](auto Mc) {
                    constexpr int m = Mc;
                    constexpr int g = m + SH;
                    constexpr int d = (g >= 0) ? g / M : -((-g + M - 1) / M);
                    constexpr int src = g - d * M;
                    x[U(s, m) + S] = l2_nbr<HALO, d>(old[src], edge_hi, edge_lo);
                });
            }
        }
    };
    constexpr bool one_shot = (HALO != 3) || (NS - 1 <= M);
    if constexpr (!INV) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            f2 pj = {cf.p[j], cf.p[j]}, kj = {-cf.kap[j], -cf.kap[j]};
            l2_vgpr(pj);
            l2_vgpr(kj);
#pragma unroll
            for (int s = 0; s < NSEQ; ++s)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    x[U(s, m)] = __builtin_elementwise_fma(pj, x[U(s, m) + S], x[U(s, m)]);
                    x[U(s, m) + S] = __builtin_elementwise_fma(kj, x[U(s, m)], x[U(s, m) + S]);
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
            f2 pj = {-cf.p[j], -cf.p[j]}, kj = {cf.kap[j], cf.kap[j]};
            l2_vgpr(pj);
            l2_vgpr(kj);
#pragma unroll
            for (int s = 0; s < NSEQ; ++s)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    x[U(s, m) + S] = __builtin_elementwise_fma(kj, x[U(s, m)], x[U(s, m) + S]);
                    x[U(s, m)] = __builtin_elementwise_fma(pj, x[U(s, m) + S], x[U(s, m)]);
                }
            if (j > 0) shift(std::integral_constant<int, -1>{});
        }
    }
}

// A (reg i[5:0], lane i[8:6] | cp << 3)  ->  B (reg i[7:2], lane (i8 | cp << 1) | i[1:0] << 4): exchange T2 of
// tools/lattice_lds_maps.py (the lane numbers are those of the Float64 kernel with p[11:9] = cp)
__device__ __forceinline__ void l2_t2(f2 (&a)[64], f2 (&bb)[64], unsigned lds0, int lane)
{
    const int sw = lane ^ ((lane >> 5) << 1);
    const unsigned wa0 = lds0 + 8u * sw, wa1 = lds0 + 8u * (sw ^ 1);
    const int H = lane & 15, p10 = lane >> 4;
    const int lam0 = 4 * H, sg = (p10 & 1) | ((lam0 >> 5) << 1);
    unsigned ra[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) ra[h] = lds0 + 8u * (64 * p10 + ((lam0 + h) ^ sg));
    l2_for<4>([&](auto Fq) {
        constexpr int f = Fq;
        l2_for<16>([&](auto Jq) {
            constexpr int j = Jq;
            l2_wr64<8 * 64 * j>((j & 1) ? wa1 : wa0, a[16 * f + j]);
        });
        l2_for<16 / WX_L2D_G>([&](auto Hq) {
            constexpr int q0 = WX_L2D_G * Hq;
            double t[WX_L2D_G];
            l2_for<WX_L2D_G>([&](auto Q) {
                constexpr int h = (q0 + Q) / 4, g = (q0 + Q) % 4;
                t[Q] = l2_rd64<8 * 256 * g>(ra[h]);
            });
            l2_waitn(t);
            l2_for<WX_L2D_G>([&](auto Q) {
                constexpr int h = (q0 + Q) / 4, g = (q0 + Q) % 4;
                bb[16 * h + 4 * f + g] = __builtin_bit_cast(f2, t[Q]);
            });
        });
    });
}

// forward: src image (column j = 512 contiguous samples at src + 512 j) -> dst image transposed and in packet order:
// dst[j + 512 o(i)], o(i) = bitreverse6(i[5:0]) << 3 | i[8:6].  grid (16, images), 128 threads.
// simg / dimg: the unit's images (HB = 1: two consecutive ones) in the source and the destination; bx: the workgroup's column block
// ME = false: compiled without the 8 x 8 matrix of the levels below the lattice's depth (the fused kernel: see k_lat2d_fused_f32)
template <int NS, bool BL, bool BS, int HB, int MF, bool ME = true, typename CF, typename MM>
__device__ __forceinline__ void l2_fwd_body(const float *__restrict__ simg, float *__restrict__ dimg, int bx, const CF &cf, const MM &mm,
                                            unsigned ldsb, int tid)
{
    typedef L2G<HB> G;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = ldsb + 8u * WX_L2_WIN * wave;
    const __amdgpu_buffer_rsrc_t rs = l2_rsrc(MF == 2 ? simg : dimg);      // the ring side of a fused pass
    const int j0 = 2 * G::NPW * (WX_L2D_W * bx + wave);
    f2 a[64];
    {
        // loads: instruction Q = (cp, i8, i7) [HB = 2: (cp, i9, i8, i7)] covers 8 complete lines: lane = sub | h << 3 | i5 << 4 |
        // i6 << 5 holds samples i = 128 (Q mod QS) + 64 i6 + 32 i5 + 4 sub + {0..3} of column 2 cp + h
        constexpr int QS = 32 / G::NPW;                   // instructions per column pair
        const int sub = lane & 7, h = (lane >> 3) & 1, i5 = (lane >> 4) & 1, i6 = lane >> 5;
        const unsigned lo = l2_lane_off<BL, HB>(h, i6, i5, sub);
        f4 r[32];
        l2_for<32>([&](auto Q) {
            constexpr int cp = Q / QS, sq = Q % QS;
            r[Q] = l2_ldm<MF>(simg, rs, (int)l2_src_off<BL, HB>(j0 + 2 * cp, 128 * sq), lo);
        });
        // T1: the two columns of a pair meet in one 8-byte slot (two ds_write_b32): slot = 17 lam + m, lam = i[8:6] | cp << 3,
        // m = i[5:2]; round rho = i[1:0]
        const unsigned wa = lds0 + 4u * (34u * i6 + 2u * sub + 16u * i5 + h), ra = lds0 + 8u * 17u * lane;
        l2_for<4>([&](auto Rq) {
            constexpr int rho = Rq;
            l2_for<32>([&](auto Q) {
                l2_wr32<4 * 34 * 2 * Q>(wa, r[Q][rho]);      // lam = i6 | Q << 1: the lane of layout A
            });
            l2_for<16 / WX_L2D_G>([&](auto Hq) {
                constexpr int m0 = WX_L2D_G * Hq;
                double t[WX_L2D_G];
                l2_for<WX_L2D_G>([&](auto M) {
                    constexpr int m = m0 + M;
                    t[M] = l2_rd64<8 * m>(ra);
                });
                l2_waitn(t);
                l2_for<WX_L2D_G>([&](auto M) {
                    constexpr int m = m0 + M;
                    a[4 * m + rho] = __builtin_bit_cast(f2, t[M]);
                });
            });
        });
    }
    constexpr int HA = G::HA, HBB = G::HBQ;               // halos of layouts A and B
    l2_level<0, HA, NS, false>(a, cf, lane);
    l2_level<1, HA, NS, false>(a, cf, lane);
    f2 bb[64];
    l2_t2(a, bb, lds0, lane);
    l2_level<0, HBB, NS, false>(bb, cf, lane);
    l2_level<1, HBB, NS, false>(bb, cf, lane);
    l2_level<2, HBB, NS, false>(bb, cf, lane);
    if constexpr (G::LD >= 6) l2_level<3, HBB, NS, false>(bb, cf, lane);
    if constexpr (G::LD >= 7) l2_level<4, HBB, NS, false>(bb, cf, lane);
    // gains: a leaf whose path took k detail branches carries g^(2k - LD); path bits i[1:0] sit in the lane, i[LD-1:2] in the
    // register index
    float gf[6];
    {
        float b = cf.g0;
        b = (lane & 16) ? b * cf.g2 : b;
        b = (lane & 32) ? b * cf.g2 : b;
        gf[0] = b;
#pragma unroll
        for (int m = 1; m < 6; ++m) gf[m] = gf[m - 1] * cf.g2;
    }
    if constexpr (HB == 2) {
        // 1024 rows: round rho fixes (i2, i3) -> 256 of the 1024 rows; a row of the workgroup is 4 W column pairs = 32 W bytes.
        // row of the round = i8 | i9 << 1 | i7 << 2 | i6 << 3 | i5 << 4 | i4 << 5 | i1 << 6 | i0 << 7, slot = 16 row + (pair ^ 4 i9)
        constexpr int RS = 4 * WX_L2D_W, LPR = 2 * WX_L2D_W, RPI = 64 * WX_L2D_W / LPR;
        const int i8 = lane & 1, i9 = (lane >> 1) & 1, cp = (lane >> 2) & 3, i0 = (lane >> 4) & 1, i1 = lane >> 5;
        const unsigned wa = ldsb + 8u * ((unsigned)RS * (i8 | (i9 << 1) | (i1 << 6) | (i0 << 7)) + (unsigned)((4 * wave + cp) ^ (4 * i9)));
        l2_barrier();
        l2_for<4>([&](auto Rq) {
            constexpr int rho = Rq;
            l2_for<16>([&](auto Vq) {
                constexpr int v = Vq;                     // v bits: i4, i5, i6, i7
                constexpr int r = rho + 4 * v;
                constexpr int rowreg = (((v >> 3) & 1) << 2) | (((v >> 2) & 1) << 3) | (((v >> 1) & 1) << 4) | ((v & 1) << 5);
                constexpr int pc = (rho & 1) + (rho >> 1) + (v & 1) + ((v >> 1) & 1) + ((v >> 2) & 1);
                f2 val = bb[r];
                val.x *= gf[pc];
                val.y *= gf[pc];
                l2_wr64<8 * RS * rowreg>(wa, val);
            });
            l2_barrier();
            auto row_o = [&](int rr) {                    // o = i0 i1 i2 i3 i4 i5 i6 | i9 i8 i7 (bit 9 .. bit 0)
                return ((rr >> 2) & 1) | ((rr & 1) << 1) | (((rr >> 1) & 1) << 2) | (((rr >> 3) & 1) << 3) | (((rr >> 4) & 1) << 4) |
                       (((rr >> 5) & 1) << 5) | ((rho >> 1) << 6) | ((rho & 1) << 7) | (((rr >> 6) & 1) << 8) | ((rr >> 7) << 9);
            };
            auto row_ld = [&](int rr, int u) {
                const int s9 = (rr >> 1) & 1;
                return *(const f4 __attribute__((address_space(3))) *)(uintptr_t)(ldsb + 8u * ((unsigned)RS * rr + (unsigned)((2 * u) ^ (4 * s9))));
            };
            if (ME && mm.e) {
                // the levels below depth 7: a thread takes the 8 rows of a node (o bits 2..0 = rr bits 1, 0, 2) for its column pairs
                const int g = tid / LPR, u = tid % LPR;
                f4 in[8], out[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) in[e] = row_ld(8 * g + (4 * (e & 1) + ((e >> 1) & 1) + 2 * (e >> 2)), u);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    f4 acc = mm.m[r][0] * in[0];
#pragma unroll
                    for (int c = 1; c < 8; ++c) acc += mm.m[r][c] * in[c];
                    out[r] = acc;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = row_o(8 * g + (4 * (e & 1) + ((e >> 1) & 1) + 2 * (e >> 2)));
                    l2_stm<MF>(dimg, rs, (BS ? G::BLK : G::BW) * bx, (unsigned)((BS ? G::BW : G::R) * o + 4 * u), out[e]);
                }
            } else {
            l2_for<256 / RPI>([&](auto Kq) {
                constexpr int k = Kq;
                const int rr = RPI * k + tid / LPR, u = tid % LPR;
                const f4 val = row_ld(rr, u);
                const int o = row_o(rr);
                l2_stm<MF>(dimg, rs, (BS ? G::BLK : G::BW) * bx, (unsigned)((BS ? G::BW : G::R) * o + 4 * u), val);
            });
            }
            l2_barrier();
        });
        return;
    }
    // transposed store: round rho fixes (i2, i3) = register bits 0, 1 -> 128 of the 512 rows o(i); a row of the
    // workgroup is 8 W column pairs = 64 W bytes.  slot = 8 W row + (pair ^ 8 i8)
    constexpr int RS = 8 * WX_L2D_W, LPR = 4 * WX_L2D_W, RPI = 16;           // slots per row, lanes per row, rows per read
    const int i8 = lane & 1, cp = (lane >> 1) & 7, i0 = (lane >> 4) & 1, i1 = lane >> 5;
    const unsigned wa = ldsb + 8u * ((unsigned)RS * ((i8 << 2) | (i1 << 5) | (i0 << 6)) + (unsigned)((8 * wave + cp) ^ (8 * i8)));
    l2_barrier();                                      // the exchange windows are reused as the row buffer
    l2_for<4>([&](auto Rq) {
        constexpr int rho = Rq;
        l2_for<16>([&](auto Vq) {
            constexpr int v = Vq;                         // v bits: i4, i5, i6, i7
            constexpr int r = rho + 4 * v;
            constexpr int rowreg = ((v & 1) << 4) | (((v >> 1) & 1) << 3) | (((v >> 3) & 1) << 1) | ((v >> 2) & 1);
            constexpr int pc = (rho & 1) + (rho >> 1) + (v & 1) + (HB == 0 ? ((v >> 1) & 1) : 0);
            f2 val = bb[r];
            val.x *= gf[pc];
            val.y *= gf[pc];
            l2_wr64<8 * RS * rowreg>(wa, val);
        });
        l2_barrier();
        // row bits: i6 = rr0, i7 = rr1, i8 = rr2, i5 = rr3, i4 = rr4, i1 = rr5, i0 = rr6, i2 = rho0, i3 = rho1;
        // o = bitreverse(i[LD-1:0]) above i[RB-1:LD]; HB = 1: i8 selects the image
        auto row_o = [&](int rr, int &im) {
            im = 0;
            if constexpr (HB == 0) return (rr & 31) | ((rho >> 1) << 5) | ((rho & 1) << 6) | ((rr >> 5) << 7);
            else {
                im = (rr >> 2) & 1;
                return ((rr >> 3) & 1) | ((rr & 3) << 1) | (((rr >> 4) & 1) << 3) | ((rho >> 1) << 4) | ((rho & 1) << 5) | (((rr >> 5) & 1) << 6) |
                       ((rr >> 6) << 7);
            }
        };
        auto row_ld = [&](int rr, int u) {
            const int o2 = (rr >> 2) & 1;
            return *(const f4 __attribute__((address_space(3))) *)(uintptr_t)(ldsb + 8u * ((unsigned)RS * rr + (unsigned)((2 * u) ^ (8 * o2))));
        };
        if (ME && mm.e) {
            // the levels below the lattice's depth: a thread takes the 8 rows of a node for its column pairs.  HB = 0: the node's
            // samples are rows 8 g .. 8 g + 7 in order; HB = 1: o bits 2..0 = rr bits 1, 0, 3, the image bit rr2 belongs to the group
            const int g = tid / LPR, u = tid % LPR;
            auto node_row = [&](int e) {
                if constexpr (HB == 0) return 8 * g + e;
                else return ((e >> 1) & 1) | ((e >> 2) << 1) | ((g & 1) << 2) | ((e & 1) << 3) | ((g >> 1) << 4);
            };
            f4 in[8], out[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) in[e] = row_ld(node_row(e), u);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                f4 acc = mm.m[r][0] * in[0];
#pragma unroll
                for (int c = 1; c < 8; ++c) acc += mm.m[r][c] * in[c];
                out[r] = acc;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int im;
                const int o = row_o(node_row(e), im);
                l2_stm<MF>(dimg, rs, (BS ? G::BLK : 16 * WX_L2D_W) * bx, (unsigned)(im * G::IMG + (BS ? 16 * WX_L2D_W : G::R) * o + 4 * u), out[e]);
            }
        } else {
        l2_for<128 / RPI>([&](auto Kq) {
            constexpr int k = Kq;
            const int rr = RPI * k + tid / LPR, u = tid % LPR;
            const f4 val = row_ld(rr, u);
            int im;
            const int o = row_o(rr, im);
            l2_stm<MF>(dimg, rs, (BS ? G::BLK : 16 * WX_L2D_W) * bx, (unsigned)(im * G::IMG + (BS ? 16 * WX_L2D_W : G::R) * o + 4 * u), val);
        });
        }
        l2_barrier();
    });
}

// B -> A (exchange T2i of tools/lattice_lds_maps.py)
__device__ __forceinline__ void l2_t2i(f2 (&bb)[64], f2 (&a)[64], unsigned lds0, int lane)
{
    const int H = lane & 15, p10 = lane >> 4;
    unsigned wa[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
        wa[h] = lds0 + 8u * (((h ^ (H >> 2)) | ((H & 3) << 2) | (((H >> 2) & 1) << 4) | ((H >> 3) << 5)) + 64 * p10);
    const int Ha = lane >> 2, ha = lane & 3;
    const unsigned ra = lds0 + 8u * ((ha ^ (Ha >> 2)) | ((Ha & 3) << 2) | (((Ha >> 2) & 1) << 4) | ((Ha >> 3) << 5));
    l2_for<4>([&](auto Fq) {
        constexpr int f = Fq;
        l2_for<16>([&](auto Q) {
            constexpr int h = Q / 4, g = Q % 4;
            l2_wr64<8 * 256 * g>(wa[h], bb[16 * h + 4 * f + g]);
        });
        l2_for<16 / WX_L2D_G>([&](auto Hq) {
            constexpr int j0 = WX_L2D_G * Hq;
            double t[WX_L2D_G];
            l2_for<WX_L2D_G>([&](auto Jq) {
                constexpr int j = j0 + Jq;
                t[Jq] = l2_rd64<8 * 64 * j>(ra);
            });
            l2_waitn(t);
            l2_for<WX_L2D_G>([&](auto Jq) {
                constexpr int j = j0 + Jq;
                a[16 * f + j] = __builtin_bit_cast(f2, t[Jq]);
            });
        });
    });
}

// inverse: src image (column j = 512 contiguous packet coefficients, position o(i) = bitreverse6(i[5:0]) << 3 | i[8:6]) ->
// dst image transposed, natural order: dst[j + 512 i].  grid (16, images), 128 threads.
template <int NS, bool BL, bool BS, int HB, int MF, bool ME = true, typename CF, typename MM>
__device__ __forceinline__ void l2_inv_body(const float *__restrict__ simg, float *__restrict__ dimg, int bx, const CF &cf, const MM &mm,
                                            unsigned ldsb, int tid)
{
    typedef L2G<HB> G;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = ldsb + 8u * WX_L2_WIN * wave;
    const __amdgpu_buffer_rsrc_t rs = l2_rsrc(MF == 2 ? simg : dimg);
    const int j0 = 2 * G::NPW * (WX_L2D_W * bx + wave);
    f2 bb[64];
    {
        // loads as in the forward kernel, in packet order: lane = sub | h << 3 | o5 << 4 | o6 << 5, instruction (cp, o7, o8);
        // memory index bits m8 .. m0 = (o8, o7, o6, o5, sub[2], sub[1], sub[0], component[1], component[0])
        //   HB = 0:  m8 .. m0 = i0 i1 i2 i3 i4 i5 i8 i7 i6        (packet order of depth 6 over 512 rows)
        //   HB = 1:  m8 .. m0 = i8 i0 i1 i2 i3 i4 i7 i6 i5        (image select, packet order of depth 5 over 256 rows)
        const int sub = lane & 7, h = (lane >> 3) & 1, o5 = (lane >> 4) & 1, o6 = lane >> 5;
        constexpr int QS = 32 / G::NPW;
        const unsigned lo = l2_lane_off<BL, HB>(h, o6, o5, sub);
        f4 r[32];
        l2_for<32>([&](auto Q) {
            constexpr int cp = Q / QS, sq = Q % QS;
            r[Q] = l2_ldm<MF>(simg, rs, (int)l2_src_off<BL, HB>(j0 + 2 * cp, 128 * sq), lo);
        });
        if (ME && mm.e) {
            // undo the levels below the lattice's depth first: the 8 coefficients of a node are 8 consecutive floats of the column -- this
            // lane's vector and its neighbour's (lane ^ 1: sub bit 0 is memory index bit 2); x = M^T y
            const bool up = lane & 1;
            // per lane: column k of the two 4 x 4 blocks that act on its own vector and on the neighbour's (vector operations: the
            // kernel has about 6000 SIMD cycles per wavefront in all, a scalar form of this product doubled its time)
            f4 Ac[4], Bc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    Ac[k][c] = up ? mm.m[k + 4][c + 4] : mm.m[k][c];
                    Bc[k][c] = up ? mm.m[k][c + 4] : mm.m[k + 4][c];
                }
            l2_for<32>([&](auto Q) {
                const f4 mine = r[Q];
                f4 oth;
                oth.x = __shfl_xor(mine.x, 1); oth.y = __shfl_xor(mine.y, 1); oth.z = __shfl_xor(mine.z, 1); oth.w = __shfl_xor(mine.w, 1);
                f4 res = Ac[0] * mine.x + Bc[0] * oth.x;
                res += Ac[1] * mine.y + Bc[1] * oth.y;
                res += Ac[2] * mine.z + Bc[2] * oth.z;
                res += Ac[3] * mine.w + Bc[3] * oth.w;
                r[Q] = res;
            });
        }
        // into layout B (reg i[7:2], lane mu = i8 | cp << 1 | i0 << 4 | i1 << 5): a round (the component of the loaded
        // vectors) fixes two register bits, the lane reads its 16 other registers out of row mu: slot = 17 mu + w
        //   HB = 0: round = (i6, i7) = register bits 4, 5;  w = i2 | i3 << 1 | i4 << 2 | i5 << 3 = register bits 0 .. 3
        //   HB = 1: round = (i5, i6) = register bits 3, 4;  w = i2 | i3 << 1 | i4 << 2 | i7 << 3 = register bits 0, 1, 2, 5
        // gains: the coefficient of a leaf whose path took k detail branches enters as coef * g^(2k - LD)
        float gf[6];
        {
            float b = cf.g0;
            b = (lane & 16) ? b * cf.g2 : b;
            b = (lane & 32) ? b * cf.g2 : b;
            gf[0] = b;
#pragma unroll
            for (int m = 1; m < 6; ++m) gf[m] = gf[m - 1] * cf.g2;
        }
        if constexpr (HB == 2) {
            //   HB = 2:  m9 .. m0 = i0 i1 i2 i3 i4 i5 i6 i9 i8 i7 (packet order of depth 7 over 1024 rows); lane mu = i8 | i9 << 1 |
            //   cp << 2 | i0 << 4 | i1 << 5.  The component (round) carries (i7, i8): i8 is a LANE bit of the target, so in a round
            //   the 32 lanes with that i8 read 32 registers each (i7 fixed) out of row mu >> 1: slot = 33 (mu >> 1) + w,
            //   w = i2 | i3 << 1 | i4 << 2 | i5 << 3 | i6 << 4
            const unsigned wa = lds0 + 4u * (2u * (33u * (sub & 1) + 2u * o6 + 4u * o5 + 8u * (sub >> 2) + 16u * ((sub >> 1) & 1)) + h);
            const unsigned ra = lds0 + 8u * 33u * (lane >> 1);
            l2_for<4>([&](auto Rq) {
                constexpr int rho = Rq;                   // rho bit 0 = i7, bit 1 = i8
                l2_for<32>([&](auto Q) {
                    constexpr int cp = Q / 8, i0 = (Q >> 2) & 1, i1 = (Q >> 1) & 1, i2 = Q & 1;
                    l2_wr32<4 * 2 * (33 * (2 * cp + 8 * i0 + 16 * i1) + i2)>(wa, r[Q][rho]);
                });
                if ((lane & 1) == (rho >> 1)) {
                    l2_for<2>([&](auto Hq) {
                        constexpr int w0 = 16 * Hq;
                        double t[16];
                        l2_for<16>([&](auto V) {
                            constexpr int w = w0 + V;
                            t[V] = l2_rd64<8 * w>(ra);
                        });
                        l2_waitn(t);
                        l2_for<16>([&](auto V) {
                            constexpr int w = w0 + V;
                            constexpr int pc = (w & 1) + ((w >> 1) & 1) + ((w >> 2) & 1) + ((w >> 3) & 1) + ((w >> 4) & 1);
                            f2 e = __builtin_bit_cast(f2, t[V]);
                            e.x *= gf[pc];
                            e.y *= gf[pc];
                            bb[w | ((rho & 1) << 5)] = e;
                        });
                    });
                }
            });
        } else {
        unsigned wa;
        if constexpr (HB == 0) {
            const int i8 = sub & 1, i5 = (sub >> 1) & 1, i4 = sub >> 2;
            wa = lds0 + 4u * (34u * i8 + 2u * (o6 + 2u * o5 + 4u * i4 + 8u * i5) + h);
        } else {
            const int i7 = sub & 1, i4 = (sub >> 1) & 1, i3 = sub >> 2;
            wa = lds0 + 4u * (34u * 32u * o6 + 2u * (o5 + 2u * i3 + 4u * i4 + 8u * i7) + h);
        }
        const unsigned ra = lds0 + 8u * 17u * lane;
        l2_for<4>([&](auto Rq) {
            constexpr int rho = Rq;
            l2_for<32>([&](auto Q) {
                constexpr int cp = Q >> 2, q0 = Q & 1, q1 = (Q >> 1) & 1;                 // (o7, o8)
                // HB = 0: o7 = i1 (mu bit 5), o8 = i0 (mu bit 4);  HB = 1: o7 = i0 (mu bit 4), o8 = i8 (mu bit 0)
                constexpr int mu_i = HB == 0 ? (2 * cp + 16 * q1 + 32 * q0) : (q1 + 2 * cp + 16 * q0);
                l2_wr32<4 * 34 * mu_i>(wa, r[Q][rho]);
            });
            l2_for<16 / WX_L2D_G>([&](auto Hq) {
                constexpr int v0 = WX_L2D_G * Hq;
                double t[WX_L2D_G];
                l2_for<WX_L2D_G>([&](auto V) {
                    constexpr int v = v0 + V;
                    t[V] = l2_rd64<8 * v>(ra);
                });
                l2_waitn(t);
                l2_for<WX_L2D_G>([&](auto V) {
                    constexpr int v = v0 + V;
                    f2 e = __builtin_bit_cast(f2, t[V]);
                    if constexpr (HB == 0) {
                        constexpr int pc = (v & 1) + ((v >> 1) & 1) + ((v >> 2) & 1) + ((v >> 3) & 1);
                        e.x *= gf[pc];
                        e.y *= gf[pc];
                        bb[v + 16 * rho] = e;             // register index i[7:2] = v | i6 << 4 | i7 << 5
                    } else {
                        constexpr int pc = (v & 1) + ((v >> 1) & 1) + ((v >> 2) & 1);
                        e.x *= gf[pc];
                        e.y *= gf[pc];
                        bb[(v & 7) | ((rho & 1) << 3) | ((rho >> 1) << 4) | ((v >> 3) << 5)] = e;
                    }
                });
            });
        });
        }
    }
    constexpr int HA = G::HA, HBB = G::HBQ;
    if constexpr (G::LD >= 7) l2_level<4, HBB, NS, true>(bb, cf, lane);
    if constexpr (G::LD >= 6) l2_level<3, HBB, NS, true>(bb, cf, lane);
    l2_level<2, HBB, NS, true>(bb, cf, lane);
    l2_level<1, HBB, NS, true>(bb, cf, lane);
    l2_level<0, HBB, NS, true>(bb, cf, lane);
    f2 a[64];
    l2_t2i(bb, a, lds0, lane);
    l2_level<1, HA, NS, true>(a, cf, lane);
    l2_level<0, HA, NS, true>(a, cf, lane);
    if constexpr (HB == 2) {
        // 1024 rows, natural order: layout A (reg i[5:0], lane i[9:6] | cp << 4); round rho = i[5:4]; row of the round
        // rr = i[9:6] | i[3:0] << 4 (the 16 lanes of a write land in 16 consecutive rows of 18 slots: no bank is hit twice)
        constexpr int RS = 18, LPR = 2 * WX_L2D_W, RPI = 64 * WX_L2D_W / LPR;
        const int i96 = lane & 15, cp = lane >> 4;
        const unsigned wa = ldsb + 8u * ((unsigned)RS * (unsigned)i96 + (unsigned)(4 * wave + cp));
        l2_barrier();
        l2_for<4>([&](auto Rq) {
            constexpr int rho = Rq;
            l2_for<16>([&](auto Vq) {
                constexpr int v = Vq;                     // i[3:0]
                l2_wr64<8 * RS * 16 * v>(wa, a[v + 16 * rho]);
            });
            l2_barrier();
            l2_for<256 / RPI>([&](auto Kq) {
                constexpr int k = Kq;
                const int rr = RPI * k + tid / LPR, u = tid % LPR;
                const f4 val = *(const f4 __attribute__((address_space(3))) *)(uintptr_t)(ldsb + 8u * ((unsigned)RS * rr + (unsigned)(2 * u)));
                const int i = (rr >> 4) | (rho << 4) | ((rr & 15) << 6);
                l2_stm<MF>(dimg, rs, (BS ? G::BLK : G::BW) * bx, (unsigned)((BS ? G::BW : G::R) * i + 4 * u), val);
            });
            l2_barrier();
        });
        return;
    }
    // transposed store, natural row order: layout A (reg i[5:0], lane i[8:6] | cp << 3); round rho = i[5:4];
    // row of the round rr = i[3:0] | i[8:6] << 4, slot = 8 W rr + (pair ^ i[8:6] << 1)
    constexpr int RS = 8 * WX_L2D_W, LPR = 4 * WX_L2D_W, RPI = 16;
    const int i86 = lane & 7, cp = lane >> 3;
    const unsigned wa = ldsb + 8u * ((unsigned)RS * (unsigned)(i86 << 4) + (unsigned)((8 * wave + cp) ^ (i86 << 1)));
    l2_barrier();
    l2_for<4>([&](auto Rq) {
        constexpr int rho = Rq;
        l2_for<16>([&](auto Vq) {
            constexpr int v = Vq;                         // i[3:0]
            l2_wr64<8 * RS * v>(wa, a[v + 16 * rho]);
        });
        l2_barrier();
        l2_for<128 / RPI>([&](auto Kq) {
            constexpr int k = Kq;
            const int rr = RPI * k + tid / LPR, u = tid % LPR;
            const int s86 = rr >> 4;
            const f4 val = *(const f4 __attribute__((address_space(3))) *)(uintptr_t)(ldsb + 8u * ((unsigned)RS * rr + (unsigned)((2 * u) ^ (s86 << 1))));
            const int i9 = (rr & 15) | (rho << 4) | (s86 << 6), im = i9 >> G::RB, i = i9 & (G::R - 1);
            l2_stm<MF>(dimg, rs, (BS ? G::BLK : 16 * WX_L2D_W) * bx, (unsigned)(im * G::IMG + (BS ? 16 * WX_L2D_W : G::R) * i + 4 * u), val);
        });
        l2_barrier();
    });
}

// one launch per pass: grid (column blocks, units)
template <int NS, bool BL, bool BS, int HB>
__global__ __launch_bounds__(64 * WX_L2D_W) __attribute__((amdgpu_waves_per_eu(WX_L2D_WPE, WX_L2D_WPE))) void k_lat2d_colT_f32(
    const float *__restrict__ src, float *__restrict__ dst, int last_img, WxLat2 cf, WxL2M mm)
{
    typedef L2G<HB> G;
    __shared__ double lds[WX_L2D_W * WX_L2_WIN];
    const unsigned ldsb = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    // HB = 1: a workgroup takes two consecutive images; the last workgroup of an odd batch re-does the last two
    const int64_t img0 = min((int)blockIdx.y << G::IB, last_img);
    l2_fwd_body<NS, BL, BS, HB, 0>(src + img0 * G::IMG, dst + img0 * G::IMG, blockIdx.x, cf, mm, ldsb, threadIdx.x);
}
template <int NS, bool BL, bool BS, int HB>
__global__ __launch_bounds__(64 * WX_L2D_W) __attribute__((amdgpu_waves_per_eu(WX_L2D_WPE, WX_L2D_WPE))) void k_lat2d_icolT_f32(
    const float *__restrict__ src, float *__restrict__ dst, int last_img, WxLat2 cf, WxL2M mm)
{
    typedef L2G<HB> G;
    __shared__ double lds[HB == 2 ? (256 * 18 > WX_L2D_W * WX_L2_WIN ? 256 * 18 : WX_L2D_W * WX_L2_WIN) : WX_L2D_W * WX_L2_WIN];
    const unsigned ldsb = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int64_t img0 = min((int)blockIdx.y << G::IB, last_img);
    l2_inv_body<NS, BL, BS, HB, 0>(src + img0 * G::IMG, dst + img0 * G::IMG, blockIdx.x, cf, mm, ldsb, threadIdx.x);
}

// ---- both passes in ONE persistent launch, the intermediate image in a ring that stays in the Infinity Cache (round 6) -----------------
// Two launches move every image four times through HBM (read, write the intermediate, read it, write): PMC traffic 2.0 x the
// algorithmic bytes, each pass at 0.72 of the HBM peak, the transform at 0.35 (profiles/r05_cfg4.md).  Here the workgroups of one launch
// draw tickets; ticket t of step s is a first-pass task of group s (even t) or a second-pass task of group s - D (odd t), a group
// being G units (a unit = one 512 x 512 / 1024 x 1024 image or two 256 x 256 images) = TPG tasks of 128 KiB.  The intermediate of group
// g lives in slot g mod K of a ring of K G units, so a line of the ring is written, read D steps later and overwritten K steps later --
// often enough to stay in the 256 MiB Infinity Cache, which serves it at more than the HBM rate while HBM moves the images only
// (tools/dbg/ring_probe.hip: the access shapes alone 3.12 ms in two launches, 2.25 ms through a 128 MiB ring).  Per slot two cumulative
// counters: p1[slot] counts finished first-pass tasks, p2[slot] finished second-pass tasks; tenant r = g / K of a slot reads after
// p1 >= (r + 1) TPG and writes after p2 >= r TPG (the previous tenant is consumed).  Tickets are drawn in order and a task only
// waits for tasks with smaller tickets, which are running or done: no deadlock whatever the residency.  Hand-off protocol: every storing
// wave waits for vmcnt(0), workgroup barrier, one lane adds to the counter (agent scope); the consumer's lane polls with an sc1 load,
// workgroup barrier, then the flavour-2 loads (MI355X_MICROARCH.md, inter-workgroup visibility, third row of the table).
struct WxL2Fuse {
    unsigned *ctl;            // [0] ticket, [32 (1 + 2 s)] p1 of slot s, [32 (2 + 2 s)] p2 of slot s: zeroed before the launch
    int units, G, D, K, NG;
};
// The coefficients come through a pointer into constant memory that is laundered once per task: as by-value kernel arguments (87
// scalar registers loaded at entry) they stayed live around the whole ticket loop with both bodies inlined -- 220 scalar registers
// spilled into vector lanes, 110 vector registers into scratch.  So do the lane constants: everything derived from the thread index is
// recomputed per task instead of being kept across it.
struct WxL2Consts {
    WxLat2 cf;
    WxL2M mm;
};
typedef const WxL2Consts __attribute__((address_space(4))) *l2_cst;
#define WX_L2F_MAXK 24
// ME: with / without the matrix of the deeper levels -- inside the ticket loop its 64 entries cost the 256 x 256 forward kernels 89 ... 194
// spilled registers whether a launch uses them or not, so a transform at the lattice's own depth (config 4) gets a kernel without them
template <int NS, bool INV, int HB, bool ME>
__global__ __launch_bounds__(64 * WX_L2D_W) __attribute__((amdgpu_waves_per_eu(WX_L2D_WPE, WX_L2D_WPE))) void k_lat2d_fused_f32(
    const float *__restrict__ src, float *__restrict__ ring, float *__restrict__ dst, int last_img, WxL2Fuse fz, const WxL2Consts *__restrict__ cst)
{
    typedef L2G<HB> G;
    __shared__ double lds[(INV && HB == 2) ? (256 * 18 > WX_L2D_W * WX_L2_WIN ? 256 * 18 : WX_L2D_W * WX_L2_WIN) : WX_L2D_W * WX_L2_WIN];
    __shared__ unsigned s_t, s_r;
    const unsigned ldsb = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds;
    const int tid0 = threadIdx.x;
    constexpr int TPU = G::R / G::BW;                     // tasks (column blocks) per unit
    const int TPG = fz.G * TPU;
    const unsigned total = (unsigned)(fz.NG + fz.D) * 2u * (unsigned)TPG;
    // what ticket t is: pass, group, task of the group; the counter its task waits for and the value it waits for (0 = nothing)
    auto decode = [&](unsigned t, int &second, int &g, int &k, bool &in_range) {
        const int step = t / (2u * TPG), r = t % (2u * TPG);
        second = r & 1;
        k = r >> 1;
        g = second ? step - fz.D : step;
        in_range = g >= 0 && g < fz.NG;
    };
    auto dependency = [&](int second, int g, const unsigned *&c) -> unsigned {
        const int slot = g % fz.K, tenant = g / fz.K;
        c = &fz.ctl[32 * ((second ? 1 : 2) + 2 * slot)];
        return second ? (unsigned)(tenant + 1) * TPG : (unsigned)tenant * TPG;
    };
    // The per-task bookkeeping stays off the critical path: the NEXT ticket is drawn when a task starts (its value is looked at only
    // behind the arithmetic), the counter the next task depends on is loaded by one lane behind the last store of
    // the current task (hook; with the arithmetic's registers still live it cost 30 ... 80 spilled registers) and compared at its end -- a
    // task whose dependency was seen satisfied there starts loading at once.  (Drawing
    // the ticket, polling and three barriers at the start of every task: 2.78 ms on config 4; this form: see profiles/r06_cfg4_fused.md.)
    if (tid0 == 0) {
        s_t = __hip_atomic_fetch_add(&fz.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_r = 0;
    }
    __syncthreads();
    unsigned t = __builtin_amdgcn_readfirstlane(s_t), ready = 0;
    while (t < total) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        l2_cst cp = (l2_cst)(uintptr_t)cst;
        asm volatile("" : "+s"(cp));
        unsigned nt = 0, pv = 0, need_n = 0;
        if (tid == 0) nt = __hip_atomic_fetch_add(&fz.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int second, g, k;
        bool in_range;
        decode(t, second, g, k, in_range);
        const int u = g * fz.G + k / TPU, bx = k % TPU;
        const bool work = in_range && u < fz.units;
        const int slot = in_range ? g % fz.K : 0;
        if (work) {
            float *zimg = ring + ((size_t)slot * fz.G + k / TPU) * ((size_t)G::IMG << G::IB);
            if (!ready) {
                const unsigned *c;
                const unsigned need = dependency(second, g, c);
                if (need) {
                    if (tid == 0)
                        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
                    __syncthreads();
                }
            }
            auto hook = [&]() {
                if (tid == 0) {
                    const unsigned ntu = __builtin_amdgcn_readfirstlane(nt);
                    if (ntu < total) {
                        int s2, g2, k2;
                        bool ir2;
                        decode(ntu, s2, g2, k2, ir2);
                        if (ir2) {
                            const unsigned *c;
                            need_n = dependency(s2, g2, c);
                            if (need_n) pv = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            };
            const int64_t img0 = min(u << G::IB, last_img);
            // the two bodies are laundered apart as well: with common subexpressions shared across the branch the 256 x 256 forward
            // kernels spilled 89 ... 194 registers
            if (!second) {
                asm volatile("" : "+v"(tid), "+s"(cp));
                if constexpr (INV) l2_inv_body<NS, false, true, HB, 1, ME>(src + img0 * G::IMG, zimg, bx, cp->cf, cp->mm, ldsb, tid);
                else l2_fwd_body<NS, false, true, HB, 1, ME>(src + img0 * G::IMG, zimg, bx, cp->cf, cp->mm, ldsb, tid);
            } else {
                asm volatile("" : "+v"(tid), "+s"(cp));
                if constexpr (INV) l2_inv_body<NS, true, false, HB, 2, ME>(zimg, dst + img0 * G::IMG, bx, cp->cf, cp->mm, ldsb, tid);
                else l2_fwd_body<NS, true, false, HB, 2, ME>(zimg, dst + img0 * G::IMG, bx, cp->cf, cp->mm, ldsb, tid);
            }
            // the next task's dependency: its counter is loaded behind the last stores of this task, whose acknowledgements are waited for anyway
            hook();
            // every wave waits for its own accesses (the ring stores are acknowledged, the ring loads have landed), then one lane signals
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (tid == 0) {
            if (in_range)                                      // the tasks of a ragged last group beyond the batch count too
                __hip_atomic_fetch_add(&fz.ctl[32 * ((second ? 2 : 1) + 2 * slot)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_t = nt;
            s_r = (work && pv >= need_n) ? 1u : 0u;            // need_n = 0: nothing to wait for, or nothing looked at (then the start polls)
        }
        __syncthreads();
        t = __builtin_amdgcn_readfirstlane(s_t);
        ready = __builtin_amdgcn_readfirstlane(s_r);
        __syncthreads();
    }
}

}  // namespace

// coefficients of the lattice and the matrix of the levels below its depth; false = this filter / depth is not ours
template <int HB> static bool l2_prepare(const WxFilt &filt, int L, bool inverse, WxLat2 &cf, WxL2M &mm)
{
    constexpr int LD = L2G<HB>::LD;
    double p[WX_L2_MAXS], kap[WX_L2_MAXS], g0, g2;
    if (!wx_lattice_coeffs(filt, LD, inverse, p, kap, &g0, &g2)) return false;
    for (int j = 0; j < WX_L2_MAXS; ++j) { cf.p[j] = (float)p[j]; cf.kap[j] = (float)kap[j]; }
    cf.g0 = (float)g0;
    cf.g2 = (float)g2;
    // depths beyond the lattice's: e = L - LD levels on the 8-sample nodes as one matrix (columns = images of the unit vectors)
    mm.e = L - LD;
    if (mm.e < 0 || mm.e > 3) return false;
    for (int c = 0; c < 8; ++c) {
        double v[8], w[8];
        for (int i = 0; i < 8; ++i) v[i] = i == c ? 1.0 : 0.0;
        for (int l = 0; l < mm.e; ++l) {
            const int np = 8 >> l, h = np >> 1;
            for (int j = 0; j < (1 << l); ++j) {
                const double *x = v + j * np;
                double *y = w + j * np;
                for (int i = 0; i < h; ++i) {
                    double a = 0.0, d = 0.0;
                    for (int k = 0; k < filt.F; ++k) {
                        a += filt.q[k] * x[(2 * i + k) % np];
                        d += ((k & 1) ? -filt.q[k] : filt.q[k]) * x[(((2 * i + 1 - k) % np) + np) % np];
                    }
                    y[i] = a;
                    y[h + i] = d;
                }
            }
            for (int i = 0; i < 8; ++i) v[i] = w[i];
        }
        for (int r = 0; r < 8; ++r) mm.m[r][c] = (float)v[r];
    }
    return true;
}

// geometry of the fused launch for `batch` images: groups of 8 MiB (64 tasks), second pass 12 groups behind the first (1536 tickets: three
// times the tasks in flight on 256 CUs -- the dependency of a task is looked at while the task before it computes), ring of 24 groups =
// 192 MiB (profiles/r06_cfg4_fused.md: 2.69 ms; 8 / 16 groups 2.75 ms; less than 64 MiB behind: 3.3 ... 4.1 ms); a batch of at most 24
// groups is its own ring (no slot is reused).  WX_L2F_G / _D / _K (knobs) override.
template <int HB> static void l2_fuse_plan(int64_t batch, WxL2Fuse &fz)
{
    typedef L2G<HB> G;
    static const int eg = wx_getenv("WX_L2F_G") ? atoi(wx_getenv("WX_L2F_G")) : 0, ed = wx_getenv("WX_L2F_D") ? atoi(wx_getenv("WX_L2F_D")) : 0,
                     ek = wx_getenv("WX_L2F_K") ? atoi(wx_getenv("WX_L2F_K")) : 0;
    const int64_t per = HB == 1 ? 2 : 1;
    fz.units = (int)((batch + per - 1) / per);
    fz.G = eg > 0 ? eg : 64 / (G::R / G::BW);
    fz.D = ed > 0 ? ed : 12;
    fz.K = ek > fz.D ? ek : 2 * fz.D;
    if (fz.K > WX_L2F_MAXK) fz.K = WX_L2F_MAXK;
    if (fz.D >= fz.K) fz.D = fz.K - 1;
    fz.NG = (fz.units + fz.G - 1) / fz.G;
    if (fz.NG <= fz.K) fz.K = fz.NG;
}
// elements of the ring the fused launch needs for `batch` images of geometry HB
template <int HB> static int64_t wx_lattice2d_ring_elems_t(int64_t batch)
{
    WxL2Fuse fz;
    l2_fuse_plan<HB>(batch, fz);
    const int64_t slots = (int64_t)fz.K * fz.G;           // a batch that is its own ring (K = NG) needs its units only
    return (slots < fz.units ? slots : (int64_t)fz.units) * ((int64_t)L2G<HB>::IMG << L2G<HB>::IB);
}

// both passes of a transform in one persistent launch (k_lat2d_fused_f32): 0 = not applicable, 1 = launched, < 0 = error.  ring: at least
// wx_lattice2d_ring_elems elements; ctl: WX_L2F_CTL_BYTES bytes, both the caller's scratch on the same stream
#define WX_L2F_CTL_BYTES (4 * 32 * (2 + 2 * WX_L2F_MAXK))
template <int HB>
static int wx_lattice2d_fused_launch(const float *src, float *dst, float *ring, unsigned *ctl, int64_t m, int L, int64_t batch, const WxFilt &filt,
                                     bool inverse, hipStream_t st)
{
    WxL2Consts hc;
    memset(&hc, 0, sizeof hc);                                // the upload is cached by content: no stray padding bytes
    if (!l2_prepare<HB>(filt, L, inverse, hc.cf, hc.mm)) return 0;
    const int64_t per = HB == 1 ? 2 : 1, units = (batch + per - 1) / per;
    if (batch < per || units > 65535 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)ring & 15)) return 0;
    if ((batch & (per - 1)) && src == dst) return 0;          // the last unit re-does an image: out of place only
    WxL2Fuse fz;
    l2_fuse_plan<HB>(batch, fz);
    fz.ctl = ctl;
    static std::atomic<int> cus_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    int cus = cus_of[dev & 63].load(std::memory_order_relaxed);
    if (!cus) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; else (void)hipGetLastError();
        if (cus <= 0) cus = 256;
        cus_of[dev & 63].store(cus, std::memory_order_relaxed);
    }
    const int64_t tasks = 2 * (int64_t)fz.NG * fz.G * (L2G<HB>::R / L2G<HB>::BW);
    int64_t nwg = (int64_t)cus * (WX_L2D_WPE * 4 / WX_L2D_W);   // what is resident at once
    if (nwg > tasks) nwg = tasks;
    const WxL2Consts *dc = (const WxL2Consts *)wx_const_upload(&hc, sizeof hc, st, true);
    if (!dc) return WX_EHIP;
    const hipError_t em = hipMemsetAsync(ctl, 0, WX_L2F_CTL_BYTES, st);
    if (em != hipSuccess) return wx_set_hip_error(em, "lattice2d fused: counters", __FILE__, __LINE__);
    const dim3 grid((unsigned)nwg), wg(64 * WX_L2D_W);
    const int last_img = (int)(batch - per);
#define WX_GO2F(NSS)                                                                                                       \
    case NSS:                                                                                                              \
        if (inverse) {                                                                                                     \
            if (hc.mm.e) hipLaunchKernelGGL((k_lat2d_fused_f32<NSS, true, HB, true>), grid, wg, 0, st, src, ring, dst, last_img, fz, dc);   \
            else hipLaunchKernelGGL((k_lat2d_fused_f32<NSS, true, HB, false>), grid, wg, 0, st, src, ring, dst, last_img, fz, dc);          \
        } else if (hc.mm.e) {                                                                                              \
            if constexpr (HB == 1) return 0;          /* spills: the two launches are faster */                           \
            else hipLaunchKernelGGL((k_lat2d_fused_f32<NSS, false, HB, true>), grid, wg, 0, st, src, ring, dst, last_img, fz, dc);          \
        } else if (HB == 1 && NSS == 10) {                                                                                 \
            return 0;                                 /* 18 / 20 taps on 256 x 256 forward: 57 spilled registers */        \
        } else hipLaunchKernelGGL((k_lat2d_fused_f32<NSS, false, HB, false>), grid, wg, 0, st, src, ring, dst, last_img, fz, dc);           \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GO2F(1) WX_GO2F(2) WX_GO2F(4) WX_GO2F(6) WX_GO2F(8) WX_GO2F(10)
    default: return 0;
    }
#undef WX_GO2F
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice2d fused launch", __FILE__, __LINE__);
    return 1;
}

// launch of one transposing pass for geometry HB (0: 512 x 512, 1: 256 x 256, 2: 1024 x 1024): one translation unit per geometry
// (wx_lattice2d.hip, wx_lattice2d_256.hip, wx_lattice2d_1024.hip) so that the 60 kernels of each compile in parallel
template <int HB>
static int wx_lattice2d_launch(const float *src, float *dst, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse, int pass, hipStream_t st)
{
    static const bool blocked = WX_L2D_W == 4 && !(wx_getenv("WX_L2D_BLOCKED") && atoi(wx_getenv("WX_L2D_BLOCKED")) == 0);
    WxLat2 cf;
    WxL2M mm;
    if (!l2_prepare<HB>(filt, L, inverse, cf, mm)) return 0;
    const int64_t per = HB == 1 ? 2 : 1, units = (batch + per - 1) / per;
    if (batch < per || units > 65535 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return 0;
    if ((batch & (per - 1)) && src == dst) return 0;          // the last workgroup re-does images: out of place only
    // Since round 6 a transform is ONE launch of k_lat2d_fused_f32; what is still built of the one-launch-per-pass form: the blocked pair
    // for the forward transform of 256 x 256 images (at full depth its fused kernel spills: the launcher above declines), and the pair of
    // db4 in every geometry and direction so that the fused launch can be timed against it on any box (WX_L2D_FUSED=0).  Anything else:
    // 0 = not ours, the caller takes the strip kernels.  (The unblocked variant -- pass 0, WX_L2D_BLOCKED=0 -- is gone.)
    if (!blocked || (pass != 1 && pass != 2)) return 0;
    const bool bl = pass == 2;
    const int nsq = wx_lat_stages(filt.F);
    if (!((HB == 1 && !inverse) || nsq == 4)) return 0;
    const int cols_wg = (HB == 2 ? 8 : 16) * WX_L2D_W;
    const dim3 grid((unsigned)(m / cols_wg), (unsigned)units), wg(64 * WX_L2D_W);
    const int last_img = (int)(batch - per);
#define WX_GO2K(K, NSS)                                                                                                  \
    do {                                                                                                                 \
        if (bl) hipLaunchKernelGGL((K<NSS, true, false, HB>), grid, wg, 0, st, src, dst, last_img, cf, mm);                  \
        else hipLaunchKernelGGL((K<NSS, false, true, HB>), grid, wg, 0, st, src, dst, last_img, cf, mm);                     \
    } while (0)
#define WX_GO2(NSS)                                                                                                      \
    case NSS:                                                                                                            \
        if (inverse) {                                                                                                   \
            if constexpr (NSS == 4) WX_GO2K(k_lat2d_icolT_f32, NSS);                                                     \
        } else if constexpr (HB == 1 || NSS == 4) WX_GO2K(k_lat2d_colT_f32, NSS);                                        \
        break;
    switch (nsq) {
        WX_GO2(1) WX_GO2(2) WX_GO2(4) WX_GO2(6) WX_GO2(8) WX_GO2(10)
    default: return 0;
    }
#undef WX_GO2
#undef WX_GO2K
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice2d launch", __FILE__, __LINE__);
    return 1;
}
