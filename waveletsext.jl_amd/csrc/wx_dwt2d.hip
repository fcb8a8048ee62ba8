// wx_dwt2d.hip -- batched 2-D decimated wavelet-packet kernels (quad trees) for gfx950.
//
// Reference semantics (paths relative to /root/reference/src/mod):
//   dwt_step! 2-D   dwt/dwt_one_level.jl:319-354  1-D step down every column (dim 1) into temp
//                   (low rows on top), then along every row (dim 2):  w1 = top-left (low,low),
//                   w2 = top-right (low dim1, high dim2), w3 = bottom-left, w4 = bottom-right
//   idwt_step! 2-D  dwt/dwt_one_level.jl:401-436  rows first, then columns
//   wpd! 2-D        DWT.jl:164-209   slice d+1 holds all 4^d nodes of depth d
//   wpt!/iwpt! 2-D  DWT.jl:500-548, 662-710 (quad tree in heap order, children 4i-2..4i+1)
//   iwpd! 2-D       DWT.jl:354-401
// Quad-tree geometry (Utils.jl:465-542, utils_tree.jl:57-99): the node at depth d whose path has
// child codes c_1..c_d (0 TL, 1 TR, 2 BL, 3 BR) covers row block sum (c_t>>1) 2^(d-t) and column
// block sum (c_t&1) 2^(d-t); its heap index is (4^d-1)/3 + 1 + morton(rowblock, colblock).
//
// Images are column-major (m rows contiguous).  Consecutive lanes always walk dim 1 so every
// global access is coalesced; a level is two passes (columns, rows) through a scratch image.
#include "wx_common.h"
#include <type_traits>
#include "wx_kernels.h"
#include "wx_host.h"
#include <cstdlib>

static __device__ __forceinline__ int64_t wx_quad_heap(int d, int j, int k)
{
    int64_t start = 1, mort = 0;
    for (int t = d - 1; t >= 0; --t) {
        mort = (mort << 2) | ((int64_t)((j >> t) & 1) << 1) | (int64_t)((k >> t) & 1);
        start = 4 * start - 2;
    }
    return start + mort;
}

static __device__ __forceinline__ bool wx_quad_active(const uint8_t *status, int64_t nstatus, int d, int j, int k)
{
    if (!status) return true;
    const int64_t h = wx_quad_heap(d, j, k);
    return h <= nstatus && status[h - 1];
}

// pass along dim 1 (columns).  INVERSE = false: analysis (src node -> [low; high] rows);
// INVERSE = true: synthesis ([low; high] rows -> node).  One thread per output pair.
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void k_dwt2d_dim1(const T *__restrict__ src, T *__restrict__ dst,
                                                    int64_t src_img, int64_t dst_img, int m, int n, int d,
                                                    int64_t batch, WxFilt filt, const uint8_t *__restrict__ status,
                                                    int64_t nstatus, int copy_inactive,
                                                    const int *__restrict__ act, int nact)
{
    const int mp = m >> d, np = n >> d, h = mp >> 1, mh = m >> 1;
    // act = (block-row, block-column) of the nodes this level decomposes: the grid then covers those blocks only
    const int64_t per_img = act ? (int64_t)nact * np * h : (int64_t)n * mh;
    const int64_t total = (int64_t)batch * per_img;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        int j, t, c;
        int64_t b;
        if (act) {
            b = g / per_img;
            const int64_t li = g - b * per_img;
            const int a = (int)(li / ((int64_t)np * h));
            const int lj = (int)(li - (int64_t)a * np * h);
            t = lj % h;
            j = act[2 * a];
            c = act[2 * a + 1] * np + lj / h;
        } else {
            const int i = (int)(g % mh);
            const int64_t g2 = g / mh;
            c = (int)(g2 % n);
            b = g2 / n;
            j = i / h; t = i - j * h;
        }
        const T *v = src + b * src_img + (int64_t)c * m + (int64_t)j * mp;
        T *o = dst + b * dst_img + (int64_t)c * m + (int64_t)j * mp;
        if (!act && !wx_quad_active(status, nstatus, d, j, c / np)) {
            if (copy_inactive) { o[2 * t] = v[2 * t]; o[2 * t + 1] = v[2 * t + 1]; }
            continue;                                   // in-place levels leave undecomposed blocks alone
        }
        if (!INVERSE) {
            double a = 0.0, dd = 0.0;
            int k1 = 2 * t, k2 = 2 * t + 1;
            if (k1 >= mp) k1 -= mp;
            if (k2 >= mp) k2 -= mp;
            for (int k = 0; k < filt.F; ++k) {
                a = fma(filt.q[k], (double)v[k1], a);
                dd = fma((k & 1) ? -filt.q[k] : filt.q[k], (double)v[k2], dd);
                k1 = k1 + 1 == mp ? 0 : k1 + 1;
                k2 = k2 == 0 ? mp - 1 : k2 - 1;
            }
            o[t] = (T)a;
            o[h + t] = (T)dd;
        } else {
            double v0 = 0.0, v1 = 0.0;
            int k1 = t, k2 = t;
            for (int mm = 0; mm < filt.F / 2; ++mm) {
                const double av = (double)v[k1], dv = (double)v[h + k2];
                v0 = fma(filt.q[2 * mm], av, v0);
                v0 = fma(-filt.q[2 * mm + 1], dv, v0);
                v1 = fma(filt.q[2 * mm + 1], av, v1);
                v1 = fma(filt.q[2 * mm], dv, v1);
                k1 = k1 == 0 ? h - 1 : k1 - 1;
                k2 = k2 + 1 == h ? 0 : k2 + 1;
            }
            o[2 * t] = (T)v0;
            o[2 * t + 1] = (T)v1;
        }
    }
}

// pass along dim 2 (rows): lanes run over rows r, each thread owns one output pair of one row
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void k_dwt2d_dim2(const T *__restrict__ src, T *__restrict__ dst,
                                                    int64_t src_img, int64_t dst_img, int m, int n, int d,
                                                    int64_t batch, WxFilt filt, const uint8_t *__restrict__ status,
                                                    int64_t nstatus, int copy_inactive,
                                                    const int *__restrict__ act, int nact)
{
    const int mp = m >> d, np = n >> d, h = np >> 1, nh = n >> 1;
    const int64_t per_img = act ? (int64_t)nact * mp * h : (int64_t)nh * m;
    const int64_t total = (int64_t)batch * per_img;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        int r, k, t;
        int64_t b;
        if (act) {
            b = g / per_img;
            const int64_t li = g - b * per_img;
            const int a = (int)(li / ((int64_t)mp * h));
            const int lj = (int)(li - (int64_t)a * mp * h);
            r = act[2 * a] * mp + lj % mp;
            t = lj / mp;
            k = act[2 * a + 1];
        } else {
            r = (int)(g % m);
            const int64_t g2 = g / m;
            const int jj = (int)(g2 % nh);
            b = g2 / nh;
            k = jj / h; t = jj - k * h;
        }
        const T *v = src + b * src_img + r + (int64_t)k * np * m;   // column 0 of the node block
        T *o = dst + b * dst_img + r + (int64_t)k * np * m;
        if (!act && !wx_quad_active(status, nstatus, d, r / mp, k)) {
            if (copy_inactive) {
                o[(int64_t)(2 * t) * m] = v[(int64_t)(2 * t) * m];
                o[(int64_t)(2 * t + 1) * m] = v[(int64_t)(2 * t + 1) * m];
            }
            continue;
        }
        if (!INVERSE) {
            double a = 0.0, dd = 0.0;
            int k1 = 2 * t, k2 = 2 * t + 1;
            if (k1 >= np) k1 -= np;
            if (k2 >= np) k2 -= np;
            for (int kk = 0; kk < filt.F; ++kk) {
                a = fma(filt.q[kk], (double)v[(int64_t)k1 * m], a);
                dd = fma((kk & 1) ? -filt.q[kk] : filt.q[kk], (double)v[(int64_t)k2 * m], dd);
                k1 = k1 + 1 == np ? 0 : k1 + 1;
                k2 = k2 == 0 ? np - 1 : k2 - 1;
            }
            o[(int64_t)t * m] = (T)a;
            o[(int64_t)(h + t) * m] = (T)dd;
        } else {
            double v0 = 0.0, v1 = 0.0;
            int k1 = t, k2 = t;
            for (int mm = 0; mm < filt.F / 2; ++mm) {
                const double av = (double)v[(int64_t)k1 * m], dv = (double)v[(int64_t)(h + k2) * m];
                v0 = fma(filt.q[2 * mm], av, v0);
                v0 = fma(-filt.q[2 * mm + 1], dv, v0);
                v1 = fma(filt.q[2 * mm + 1], av, v1);
                v1 = fma(filt.q[2 * mm], dv, v1);
                k1 = k1 == 0 ? h - 1 : k1 - 1;
                k2 = k2 + 1 == h ? 0 : k2 + 1;
            }
            o[(int64_t)(2 * t) * m] = (T)v0;
            o[(int64_t)(2 * t + 1) * m] = (T)v1;
        }
    }
}

// getbasiscoef 2-D (Utils.jl:127-130): out[r,c] = Xw[r, c, depth of the leaf that owns (r,c)]
template <typename T>
__global__ __launch_bounds__(256) void k_gather_leaves2d(const T *__restrict__ Xw, T *__restrict__ out, int m,
                                                         int n, int k, int64_t batch,
                                                         const int *__restrict__ colmap, int nblk, int blk_r,
                                                         int blk_c)
{
    const int64_t mn = (int64_t)m * n, total = batch * mn;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = g / mn;
        const int64_t e = g - b * mn;
        const int r = (int)(e % m), c = (int)(e / m);
        const int dsl = colmap[(r / blk_r) * nblk + (c / blk_c)];
        out[g] = Xw[(b * k + dsl) * mn + e];
    }
}

// ------------------------------------------------------------------------------------------
// Full-tree fast path.  The column steps C_d (dim 1) and the row steps R_d (dim 2) act on different
// axes of a product structure, so C_0 R_0 C_1 R_1 ... = (C_{L-1}..C_0)(R_{L-1}..R_0): a full 2-D
// packet transform of depth L is the batched 1-D transform of depth L down every column (the fused
// 1-D kernels, columns are contiguous) followed by the 1-D transform along every row.  The row
// transform runs on strips of R rows staged in LDS as [column][row-in-strip]; lanes run over the
// rows of the strip, so every LDS access of a wave is a run of consecutive words whatever the tap,
// and the HBM side moves R contiguous elements per column.  4 image passes in total instead of 4
// per level.  (Rounding differs from the per-level order by O(eps); inside the 1e-5 / 1e-10 budget.)
// ------------------------------------------------------------------------------------------
template <typename T, int V> struct alignas(sizeof(T) * V) WxRowVec { T e[V]; };

// acc += q * w on the V rows of a vector (the compiler pairs Float32 rows into v_pk_fma_f32 by itself; a hardware-vector member type
// was tried in round 4 and made the in-place variants spill: 1024-column Float32 rows 1.15 -> 1.8 ms)
template <typename T, int V> __device__ __forceinline__ void wx_vfma(WxRowVec<T, V> &acc, T q, const WxRowVec<T, V> &w)
{
#pragma unroll
    for (int e = 0; e < V; ++e) acc.e[e] = fma(q, w.e[e], acc.e[e]);
}

// NLV packet levels of a block of 16 samples held in registers (the block is a whole node of 16 samples, so every periodic wrap
// stays inside it; all indices are compile-time after unrolling).  Same sums in the same tap order as the per-level code of
// k_rows_fused below (dwt_step! / idwt_step!, dwt/dwt_one_level.jl:94-105, 207-221).
template <typename T, int F, int NLV, bool INVERSE>
__device__ __forceinline__ void wx_reg_levels16(T (&x)[16], const T (&q)[F])
{
#pragma unroll
    for (int s = 0; s < NLV; ++s) {
        const int l = INVERSE ? NLV - 1 - s : s;
        const int M = 16 >> l, HM = M >> 1;
#pragma unroll
        for (int j = 0; j < (1 << l); ++j) {
            const int b = j * M;
            T r[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) r[e] = e < M ? x[b + (e < M ? e : 0)] : (T)0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < HM) {
                    if (!INVERSE) {
                        T a = 0, d = 0;
#pragma unroll
                        for (int k = 0; k < F; ++k) {
                            a = fma(q[k], r[(2 * i + k) % M], a);
                            d = fma((k & 1) ? -q[k] : q[k], r[((2 * i + 1 - k) % M + M) % M], d);
                        }
                        x[b + i] = a;
                        x[b + HM + i] = d;
                    } else {
                        T v0 = 0, v1 = 0;
#pragma unroll
                        for (int m = 0; m < F / 2; ++m) {
                            const T av = r[((i - m) % HM + HM) % HM], dv = r[HM + (i + m) % HM];
                            v0 = fma(q[2 * m], av, v0);
                            v0 = fma(-q[2 * m + 1], dv, v0);
                            v1 = fma(q[2 * m + 1], av, v1);
                            v1 = fma(q[2 * m], dv, v1);
                        }
                        x[b + 2 * i] = v0;
                        x[b + 2 * i + 1] = v1;
                    }
                }
            }
        }
    }
}

// V = rows per lane (16-byte LDS / HBM accesses when V * sizeof(T) = 16): the filter work per LDS
// instruction grows V-fold, which is what bounds this kernel (LDS instruction issue, not bytes).
// (Tried and dropped: the four register levels as one 16 x 16 matrix from the kernel arguments -- half the multiply-adds, but 256
// coefficients do not fit the scalar registers and their reloads per block stalled the loop: 256-column Float64 rows 1.07 -> 1.64 ms.)
// REGL: the levels on nodes of at most 16 columns run in registers (below); its own instantiation, because the extra code costs the
// variants that do not use it registers (Float64 rows of 64 columns: 1.04 -> 1.22 ms with the code merely present).
// (Tried in round 4 and removed: four output pairs per item -- 2.5 instead of 4 LDS reads per output -- was slower everywhere,
// 256-column Float32 rows 1.13 -> 1.33 ms: the kernel is bound by the latency of its phases, not by LDS bytes or FMA issue.)
template <typename T, int F, bool INVERSE, int V, int KI = 0, bool REGL = false>
__global__ __launch_bounds__(1024) void k_rows_fused(const T *__restrict__ src, T *__restrict__ dst,
                                                     int64_t src_img, int64_t dst_img, int m, int log2n, int L,
                                                     int64_t nimg, WxFilt filt, int log2R, int S, int xcd, int regl)
{
    // KI == 0: two LDS images, a level reads one and writes the other.  KI > 0: one LDS image, every lane keeps the
    // results of its (at most KI) items in registers across a barrier and writes them back in place -- half the
    // LDS, so two workgroups share a CU and one loads / stores its strip while the other computes.
    constexpr bool INPLACE = KI > 0;
    constexpr int KM = INPLACE ? KI : 1;
    typedef WxRowVec<T, V> TV;
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    const int n = 1 << log2n;
    const int R = 1 << log2R;                         // rows per strip
    const int RV = R / V, log2RV = log2R - (V == 4 ? 2 : (V == 2 ? 1 : 0));
    const int SV = S / V;                             // LDS column pitch in vectors
    TV *cur = reinterpret_cast<TV *>(wx_smem);
    TV *nxt = INPLACE ? cur : cur + (size_t)n * SV;
    const int strips_per_img = (m + R - 1) >> log2R;
    const int64_t nstrips = nimg * strips_per_img;
    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];
    const int r = threadIdx.x & (RV - 1);            // row group of the strip owned by this lane
    const int g0 = threadIdx.x >> log2RV;            // first item (column pair group) of this lane
    const int gstep = blockDim.x >> log2RV;

    // one item = two output pairs of one node: 2F-tap window (forward) / F/2+1 pairs of children (inverse)
    auto compute = [&](const TV *a, int lnp, int it, TV (&res)[4]) {
        const int np = 1 << lnp, h = np >> 1;
        const int j = it >> (lnp - 2), t = it & ((h >> 1) - 1);
        const TV *v = a + (size_t)(j << lnp) * SV;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < V; ++e) res[u].e[e] = 0;
        if (!INVERSE) {
            // a[i] needs v[2i .. 2i+F-1] and d[i'] needs v[2i'+2-F .. 2i'+1]: the window of a[2t], a[2t+1] -- v[4t .. 4t+F+1] -- is also the
            // window of d[2t+F/2-1], d[2t+F/2], so an item produces those four from F + 2 LDS reads (2F with d[2t], d[2t+1]; the taps
            // and their order per output are unchanged: same bits)
            TV w[F + 2];
#pragma unroll
            for (int k = 0; k < F + 2; ++k) w[k] = v[((4 * t + k) & (np - 1)) * SV];
#pragma unroll
            for (int k = 0; k < F; ++k) {
                const T qd = (k & 1) ? -q[k] : q[k];
                wx_vfma<T, V>(res[0], q[k], w[k]);
                wx_vfma<T, V>(res[1], q[k], w[2 + k]);
                wx_vfma<T, V>(res[2], qd, w[F - 1 - k]);
                wx_vfma<T, V>(res[3], qd, w[F + 1 - k]);
            }
        } else {
            // parent samples 4t..4t+3 (k = 2t, 2t+1) from a[k-m], d[k+m]
            constexpr int HF = F / 2;
            TV aw[HF + 1], dw[HF + 1];
#pragma unroll
            for (int k = 0; k < HF + 1; ++k) {
                aw[k] = v[((2 * t + 1 - HF + k) & (h - 1)) * SV];        // a[2t+1-HF .. 2t+1]
                dw[k] = v[(h + ((2 * t + k) & (h - 1))) * SV];           // d[2t .. 2t+HF]
            }
#pragma unroll
            for (int mm = 0; mm < HF; ++mm) {
                wx_vfma<T, V>(res[0], q[2 * mm], aw[HF - 1 - mm]);
                wx_vfma<T, V>(res[0], -q[2 * mm + 1], dw[mm]);
                wx_vfma<T, V>(res[1], q[2 * mm + 1], aw[HF - 1 - mm]);
                wx_vfma<T, V>(res[1], q[2 * mm], dw[mm]);
                wx_vfma<T, V>(res[2], q[2 * mm], aw[HF - mm]);
                wx_vfma<T, V>(res[2], -q[2 * mm + 1], dw[1 + mm]);
                wx_vfma<T, V>(res[3], q[2 * mm + 1], aw[HF - mm]);
                wx_vfma<T, V>(res[3], q[2 * mm], dw[1 + mm]);
            }
        }
    };
    auto store = [&](TV *b, int lnp, int it, const TV (&res)[4]) {
        const int np = 1 << lnp, h = np >> 1;
        const int j = it >> (lnp - 2), t = it & ((h >> 1) - 1);
        TV *o = b + (size_t)(j << lnp) * SV;
        if (!INVERSE) {
            o[(2 * t) * SV] = res[0]; o[(2 * t + 1) * SV] = res[1];
            o[(h + ((2 * t + F / 2 - 1) & (h - 1))) * SV] = res[2]; o[(h + ((2 * t + F / 2) & (h - 1))) * SV] = res[3];
        } else {
            o[(4 * t) * SV] = res[0]; o[(4 * t + 1) * SV] = res[1];
            o[(4 * t + 2) * SV] = res[2]; o[(4 * t + 3) * SV] = res[3];
        }
    };

    // regl: the levels on nodes of at most 16 samples (depths log2n - 4 ... L - 1) run in registers -- a lane takes whole blocks of
    // 16 columns of its rows through all of them between one LDS read and one LDS write, in place (wx_reg_levels16); a level
    // through LDS costs 2F / 4 reads and a write per sample.  Full-depth rows of 256 columns: 8 LDS levels -> 4 + 1.
    const int dreg = log2n - 4;
    const int nreg = (REGL && regl && log2n >= 4 && L > dreg) ? L - dreg : 0;
    const int Llds = L - nreg;
    auto reg_phase = [&](TV *a, auto nlv_c) {
        constexpr int NLV = decltype(nlv_c)::value;
        for (int blk = g0; blk < (n >> 4); blk += gstep) {
            // one row at a time (element accesses: with the lanes 16 bytes apart they cost the LDS what the vector accesses
            // cost, and 16 samples + their results are all a lane holds)
            T *p = reinterpret_cast<T *>(a + (size_t)(blk << 4) * SV);
#pragma unroll 1
            for (int c = 0; c < V; ++c) {
                T x[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) x[e] = p[e * SV * V + c];
                wx_reg_levels16<T, F, NLV, INVERSE>(x, q);
#pragma unroll
                for (int e = 0; e < 16; ++e) p[e * SV * V + c] = x[e];
            }
        }
    };
    auto reg_levels = [&](TV *a) {
        switch (nreg) {
        case 1: reg_phase(a, std::integral_constant<int, 1>{}); break;
        case 2: reg_phase(a, std::integral_constant<int, 2>{}); break;
        case 3: reg_phase(a, std::integral_constant<int, 3>{}); break;
        default: reg_phase(a, std::integral_constant<int, 4>{}); break;
        }
        __syncthreads();
    };
    // xcd: workgroups are dealt to the 8 XCDs in turn, and neighbouring strips share 128-byte lines (a strip's run per column is
    // R elements): every XCD takes a contiguous eighth of the strips, so that the neighbours meet in one L2
    const int64_t per_x = (nstrips + 7) >> 3;
    for (int64_t s0 = blockIdx.x; s0 < (xcd ? per_x * 8 : nstrips); s0 += gridDim.x) {
        const int64_t sidx = xcd ? (s0 & 7) * per_x + (s0 >> 3) : s0;
        if (sidx >= nstrips) continue;                  // (uniform in the workgroup, no barrier skipped: the body is whole)
        const int64_t img = sidx / strips_per_img;
        const int r0 = (int)(sidx - img * strips_per_img) << log2R;
        const T *sp = src + img * src_img + r0 + r * V;
        T *dp = dst + img * dst_img + r0 + r * V;
        TV *a = cur + r, *b = nxt + r;
        const bool row_ok = r0 + r * V < m;          // m is a multiple of V (checked by the launcher)
        if (row_ok) {
            // four columns of a lane in flight together (a plain loop waits for each load before the next)
            int c = g0;
            for (; c + 3 * gstep < n; c += 4 * gstep) {
                const TV t0 = *reinterpret_cast<const TV *>(sp + (int64_t)c * m);
                const TV t1 = *reinterpret_cast<const TV *>(sp + (int64_t)(c + gstep) * m);
                const TV t2 = *reinterpret_cast<const TV *>(sp + (int64_t)(c + 2 * gstep) * m);
                const TV t3 = *reinterpret_cast<const TV *>(sp + (int64_t)(c + 3 * gstep) * m);
                a[c * SV] = t0; a[(c + gstep) * SV] = t1; a[(c + 2 * gstep) * SV] = t2; a[(c + 3 * gstep) * SV] = t3;
            }
            for (; c < n; c += gstep) a[c * SV] = *reinterpret_cast<const TV *>(sp + (int64_t)c * m);
        }
        __syncthreads();
        if constexpr (REGL && INVERSE) { if (nreg) reg_levels(a); }
        for (int s = 0; s < Llds; ++s) {
            const int d = INVERSE ? Llds - 1 - s : s;
            const int lnp = log2n - d;               // log2(node length)
            const int np = 1 << lnp, h = np >> 1;
            if (h >= 2) {
                if (INPLACE) {
                    TV res[KM][4];
#pragma unroll
                    for (int ki = 0; ki < KM; ++ki) {
                        const int it = g0 + ki * gstep;
                        if (it < (n >> 2)) compute(a, lnp, it, res[ki]);
                    }
                    __syncthreads();
#pragma unroll
                    for (int ki = 0; ki < KM; ++ki) {
                        const int it = g0 + ki * gstep;
                        if (it < (n >> 2)) store(b, lnp, it, res[ki]);
                    }
                } else {
                    for (int it = g0; it < (n >> 2); it += gstep) {
                        TV res[4];
                        compute(a, lnp, it, res);
                        store(b, lnp, it, res);
                    }
                }
            } else {
                // nodes of two samples: a = v0 sum(q even) + v1 sum(q odd), ... (wrapped taps); an item only
                // touches its own pair, so this step is in place as it stands
                for (int j = g0; j < (n >> 1); j += gstep) {
                    const TV x0 = a[(2 * j) * SV], x1 = a[(2 * j + 1) * SV];
                    TV y0, y1;
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        T s0 = 0, s1 = 0;
                        if (!INVERSE) {
#pragma unroll
                            for (int k = 0; k < F; ++k) {
                                s0 = fma(q[k], (k & 1) ? x1.e[e] : x0.e[e], s0);
                                s1 = fma((k & 1) ? -q[k] : q[k], (k & 1) ? x0.e[e] : x1.e[e], s1);
                            }
                        } else {
#pragma unroll
                            for (int mm = 0; mm < F / 2; ++mm) {
                                s0 = fma(q[2 * mm], x0.e[e], s0); s0 = fma(-q[2 * mm + 1], x1.e[e], s0);
                                s1 = fma(q[2 * mm + 1], x0.e[e], s1); s1 = fma(q[2 * mm], x1.e[e], s1);
                            }
                        }
                        y0.e[e] = s0; y1.e[e] = s1;
                    }
                    b[(2 * j) * SV] = y0; b[(2 * j + 1) * SV] = y1;
                }
            }
            __syncthreads();
            if (!INPLACE) { TV *tmp = a; a = b; b = tmp; }
        }
        if constexpr (REGL && !INVERSE) { if (nreg) reg_levels(a); }
        if (row_ok)
            for (int c = g0; c < n; c += gstep) *reinterpret_cast<TV *>(dp + (int64_t)c * m) = a[c * SV];
        __syncthreads();
    }
}

// strip height R (rows of an image a workgroup takes through all levels of the row pass) and LDS column pitch S >= R.  Wide images
// get lower strips, so that the two LDS images of a strip (2 n S elements) still fit: until round 4 the geometry was fixed and images
// wider than 256 (Float64) / 512 (Float32) columns fell off the fast path onto two naive launches per level (full-depth trees of
// 512 x 512 Float64 images: 21 ms per GiB, 1 % of the HBM peak).
template <typename T> static void wx_rows_geometry(int64_t n, int &R, int &S)
{
    // Strips of 256 bytes per column (64 Float32 / 32 Float64 rows) where the two LDS images fit, halved until they do; low strips get a
    // quarter of padding on the pitch when that still fits (bank conflicts between the columns of a window).  Measured in round 4
    // (tools/dbg/rows_sweep.sh, full depth, ms per GiB fwd / inv): Float64 256 columns (16, 24) 1.00 / 0.94 -> (32, 32) 0.83 / 0.79; Float32
    // 64 columns (32, 32) 1.02 / 0.94 -> (64, 64) 0.83 / 0.77; Float32 1024 columns (16, 16) 1.67 / 1.14 -> (16, 20) 1.34 / 1.18.
    constexpr int VW = 16 / (int)sizeof(T);
    R = 256 / (int)sizeof(T);
    S = R;
    // tuning knobs (strip height, a power of two, and LDS column pitch >= R)
    const char *er = wx_getenv("WX_ROWS_R"), *es = wx_getenv("WX_ROWS_S");
    if (er && es) {
        const int r = atoi(er), s2 = atoi(es);
        if (r >= 4 && r <= 64 && (r & (r - 1)) == 0 && s2 >= r && s2 <= 128) { R = r; S = s2; }
    }
    while (R > VW && (size_t)2 * n * S * sizeof(T) > 160 * 1024) {
        R >>= 1;
        S = R;
    }
    if (!(er && es) && R <= 16 && R >= 8 && (size_t)2 * n * (R + R / 4) * sizeof(T) <= 160 * 1024 && (R + R / 4) % VW == 0)
        S = R + R / 4;
}

template <typename T> bool wx_wpt2d_fast_ok(int64_t m, int64_t n, int F)
{
    int R, S;
    wx_rows_geometry<T>(n, R, S);
    const bool pow2 = n >= 2 && (n & (n - 1)) == 0;
    return pow2 && wx_fused1d_ok<T>(m, F) && (size_t)2 * n * S * sizeof(T) <= 160 * 1024 && n * m < ((int64_t)1 << 31);
}

template <typename T, bool INVERSE>
static int wx_launch_rows(const T *src, T *dst, int64_t src_img, int64_t dst_img, int64_t m, int64_t n, int L,
                          int64_t batch, const WxFilt &filt, hipStream_t st)
{
    // images of 128, 256, 512 columns: the rows of a wavefront as interleaved signals of the lattice kernels (wx_lattice_rows.h)
    static const bool latrows = !(wx_getenv("WX_LATROWS") && atoi(wx_getenv("WX_LATROWS")) == 0);
    if (latrows && batch > 0) {
        const int r = wx_lattice_rows(INVERSE, src, dst, src_img, dst_img, m, n, L, batch, filt, st);
        if (r) return r < 0 ? r : WX_OK;
    }
    int R, S;
    wx_rows_geometry<T>(n, R, S);
    size_t lds = (size_t)2 * n * S * sizeof(T);
    // 16-byte row vectors when the geometry and the pointers allow it
    constexpr int VW = 16 / (int)sizeof(T);
    const bool vec = m % VW == 0 && R % VW == 0 && S % VW == 0 && src_img % VW == 0 && dst_img % VW == 0 &&
                     ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) && filt.F <= 12 &&   // 2F-tap window of vectors in registers
                     !wx_getenv("WX_ROWS_SCALAR");
    void (*kern)(const T *, T *, int64_t, int64_t, int, int, int, int64_t, WxFilt, int, int, int, int) = nullptr;
    static const int xcd_env = wx_getenv("WX_ROWS_XCD") ? atoi(wx_getenv("WX_ROWS_XCD")) : 1;
    static const int regl_env = wx_getenv("WX_ROWS_REGL") ? atoi(wx_getenv("WX_ROWS_REGL")) : 1;
    // in place (one LDS image, two workgroups of 1024 lanes per CU) when a lane has at most two items per level
    static const int inplace_env = wx_getenv("WX_ROWS_INPLACE") ? atoi(wx_getenv("WX_ROWS_INPLACE")) : 1;
    // (two workgroups of 512 lanes: the same 16 wavefronts per CU as one workgroup of 1024 with two LDS images,
    // but their load / compute / store phases interleave)
    const int nt_ip = 512;
    const int lanes_per_col = vec ? R / VW : R;
    const int items_per_lane = (int)((n / 4 + (nt_ip / lanes_per_col) - 1) / (nt_ip / lanes_per_col));
    const bool inplace = inplace_env && vec && filt.F <= 8 && lanes_per_col <= nt_ip && items_per_lane <= 2 &&
                         (size_t)n * S * sizeof(T) <= 80 * 1024 && (size_t)n * S * sizeof(T) > 40 * 1024;
    if (inplace) lds = (size_t)n * S * sizeof(T);
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    const int nt = inplace ? nt_ip : (per_cu >= 4 ? 256 : (per_cu >= 2 ? 512 : 1024));
    // measured (db4, full depth, 1 GiB): rows of 256 Float64 columns 1.11 -> 0.97 ms, 1024 Float32 columns 1.89 -> 1.65 ms; short rows lose
    // (64 Float32 columns 1.02 -> 1.34 ms, 64 / 128 Float64 columns 1.04 -> 1.39 ms)
    const bool regl = regl_env && vec && filt.F <= 8 && ((sizeof(T) == 8 && n >= 256) || n >= 1024);
    switch (filt.F) {
#define WX_CASE(FF) case FF: kern = vec ? k_rows_fused<T, FF, INVERSE, VW> : k_rows_fused<T, FF, INVERSE, 1>; break;
        WX_CASE(10) WX_CASE(12) WX_CASE(14) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
#define WX_CASE(FF) case FF:                                                                                                                  \
        if (regl) kern = inplace ? (items_per_lane <= 1 ? k_rows_fused<T, FF, INVERSE, VW, 1, true> : k_rows_fused<T, FF, INVERSE, VW, 2, true>) \
                                 : k_rows_fused<T, FF, INVERSE, VW, 0, true>;                                                                 \
        else kern = inplace ? (items_per_lane <= 1 ? k_rows_fused<T, FF, INVERSE, VW, 1> : k_rows_fused<T, FF, INVERSE, VW, 2>)              \
                            : (vec ? k_rows_fused<T, FF, INVERSE, VW> : k_rows_fused<T, FF, INVERSE, 1>);                                    \
        break;
        WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8)
#undef WX_CASE
    default: return wx_set_error(WX_EUNSUPPORTED, "no fused row kernel for this filter length");
    }
    if (lds > 64 * 1024)
        WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    const int64_t nstrips = batch * ((m + R - 1) / R);
    int64_t grid = (int64_t)256 * per_cu;
    if (grid > nstrips) grid = nstrips;
    int log2R = 0;
    while ((1 << (log2R + 1)) <= R) ++log2R;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nt), lds, st, src, dst, src_img, dst_img, (int)m, log2n, L, batch,
                       filt, log2R, S, xcd_env, regl ? 1 : 0);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

// full tree of depth L on (m, n, batch): tmp holds m*n*batch elements
template <typename T>
int wx_dev_wpt2d_fast(const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const WxFilt &filt, T *tmp,
                      bool inverse, int64_t in_img, hipStream_t st)
{
    if (batch == 0 || m * n == 0) return WX_OK;
    const int64_t mn = m * n;
    int rc;
    // Sub-batches (WX_2D_SUB = images per sub-batch, default 0 = the whole batch per pass): the two passes of S images
    // run back to back and hand the intermediate image over in a ring of 2 S images of `tmp`, small enough to stay in
    // the 256 MiB Infinity Cache between its write and its read.  Measured on config 4 (4096 images 512 x 512 Float32):
    // no gain at any S (forward 6.9 ms whole batch, 6.8 ms at S = 128, 7.7 ms at S = 32) -- the two passes are bound by
    // latency and LDS issue, not by HBM bandwidth, so halving the HBM traffic does not show; kept as a knob.
    // Measured again with the lattice column kernels (each pass at ~61 % of HBM peak): step 6.79 ms whole batch, 7.18 ms at
    // S = 96, 7.74 ms at S = 128, 8.7 ms at S = 64, 11.3 ms at S = 32 -- short launches lose more in ramp-up and tail than
    // the cache saves.
    static const int64_t sub_env = wx_getenv("WX_2D_SUB") ? atoll(wx_getenv("WX_2D_SUB")) : 0;
    int64_t S = sub_env;
    if (S <= 0 || S >= batch) S = batch;
    if (!inverse && in_img != mn) return wx_set_error(WX_EUNSUPPORTED, "fast 2-D forward needs a dense input");
    if constexpr (sizeof(T) == 4) {
        // 512 x 512 Float32, depth 6: the transposing lattice kernel applied twice (wx_lattice2d.hip)
        if (wx_lattice2d_ok(m, n, L, filt, sizeof(T)) && in_img == mn && !(inverse && wx_getenv("WX_LATTICE2D_NOINV"))) {
            // round 6: both passes in one persistent launch, the intermediate in a ring of <= 128 MiB of `tmp` that stays in the Infinity
            // Cache (k_lat2d_fused_f32).  A launcher that declines (alignment, in-place odd batches of 256 x 256 images) leaves the two
            // launches below.
            if (wx_lattice2d_fused_on()) {                       // (`tmp` holds max(mn batch, wx_lattice2d_ring_elems) elements: wx_api_2d.hip)
                WxScratch fscr(st);
                unsigned *ctl = (unsigned *)fscr.alloc(wx_lattice2d_ctl_bytes());
                if (!ctl) return WX_EHIP;
                // a launch takes at most 65535 units (images, pairs of 256 x 256 images): longer batches in pieces, one after the other
                const int64_t piece = 65534;
                bool all = true;
                for (int64_t b0 = 0; b0 < batch && all; b0 += piece) {
                    const int64_t nb = batch - b0 < piece ? batch - b0 : piece;
                    const int rf = wx_lattice2d_fused_f32((const float *)x + b0 * mn, (float *)y + b0 * mn, (float *)tmp, ctl, m, L, nb, filt, inverse, st);
                    if (rf < 0) return rf;
                    if (rf != 1) {
                        if (b0) return wx_set_error(WX_EHIP, "lattice2d: the fused launch took a first piece of the batch and not the next");
                        all = false;
                    }
                }
                if (all) return WX_OK;
            }
            // one launch per pass (what is still built of it: wx_lattice2d_launch), the whole batch through `tmp`
            const int r1 = wx_lattice2d_colT_f32((const float *)x, (float *)tmp, m, L, batch, filt, inverse, 1, st);
            if (r1 < 0) return r1;
            if (r1 == 1) {
                const int r2 = wx_lattice2d_colT_f32((const float *)tmp, (float *)y, m, L, batch, filt, inverse, 2, st);
                if (r2 != 1) return r2 < 0 ? r2 : wx_set_error(WX_EHIP, "lattice2d: second pass not built for the first one's filter");
                return WX_OK;
            }
        }
    }
    for (int64_t b0 = 0, k = 0; b0 < batch; b0 += S, ++k) {
        const int64_t nb = (batch - b0 < S) ? batch - b0 : S;
        T *ring = (S == batch) ? tmp : tmp + (k & 1) * S * mn;
        if (!inverse) {
            // columns: the images' columns are m-sample signals, contiguous: (m, n*nb)
            if ((rc = wx_dev_wpt1d<T>(x + b0 * mn, ring, m, L, n * nb, filt, nullptr, 0, nullptr, st, 0))) return rc;
            if ((rc = wx_launch_rows<T, false>(ring, y + b0 * mn, mn, mn, m, n, L, nb, filt, st))) return rc;
        } else {
            if ((rc = wx_launch_rows<T, true>(x + b0 * in_img, ring, in_img, mn, m, n, L, nb, filt, st))) return rc;
            if ((rc = wx_dev_iwpt1d<T>(ring, y + b0 * mn, m, L, n * nb, filt, nullptr, 0, nullptr, 0, m, nullptr, nullptr, st, 0)))
                return rc;
        }
    }
    return WX_OK;
}

// Quotients by a divisor that is uniform for the launch but not known to the compiler (staged tile sides): one division per thread
// for the multiplier, a multiply-high per quotient.  Exact for n * d < 2^32, d >= 2.  (The staging loops of the tile kernels divided
// twice per element: 1300 vector instructions per thread and tile around 80 multiply-adds -- the Float32 inverse level ran its
// vector ALUs flat out on index arithmetic, profiles PMC of round 4.)
__device__ __forceinline__ unsigned wx_magic(unsigned d) { return 0xFFFFFFFFu / d + 1u; }
__device__ __forceinline__ int wx_mdiv(int n, unsigned magic) { return (int)__umulhi((unsigned)n, magic); }

// ---- one packet level in one pass: both dimensions of a tile through LDS ---------------------------------
// The two-pass level above moves every image four times (read, write, read, write).  Here a workgroup stages
// a CR x CC tile of the source slice (plus F-2 halo samples on every side when the node is larger than the tile;
// smaller nodes lie whole inside the tile and wrap in LDS), filters down the columns into a second LDS array
// ([low; high] halves) and along the rows out of it, and stores the four subbands where dwt_step! puts them
// (w1 top-left ... w4 bottom-right of the node, dwt/dwt_one_level.jl:319-354): one read and one write per
// level.  Every lane keeps a sliding window for OPT = 4 adjacent output pairs in registers (2F + 4 LDS reads for
// 8F multiply-adds); lanes run across LDS columns of odd pitch in the column pass and down the rows in the row
// pass, so LDS traffic is conflict-free and the global stores of a wavefront are 32-row runs.
// Needs dyadic node sides >= 8 and m % CR == n % CC == 0.
template <typename T, int F, int CR, int CC>
__global__ __launch_bounds__(256) void k_dwt2d_level_tile(const T *__restrict__ src, T *__restrict__ dst,
                                                          int64_t src_img, int64_t dst_img, int m, int n, int d,
                                                          WxFilt filt, T *__restrict__ dst_int, int64_t int_img,
                                                          const uint8_t *__restrict__ status, int64_t nstatus,
                                                          const int *__restrict__ act)
{
    // Tree-driven levels (status != nullptr): only the nodes the tree decomposes are transformed; a child that is
    // decomposed further goes to dst_int (the scratch image the next level reads), a leaf straight to dst.  act
    // (nodes at least as large as a tile) lists the decomposed nodes so that the grid covers nothing else.
    constexpr int OPT = 4, H = F - 2, W = 2 * F + 2 * OPT - 4;
    constexpr int PIN = (CR + 2 * H) | 1;                // pitch of a staged column (odd)
    constexpr int PT = CR | 1;                           // pitch of a column of the intermediate
    extern __shared__ __attribute__((aligned(16))) char wx_smem2[];
    T *in = reinterpret_cast<T *>(wx_smem2);
    T *tmp = in + (CC + 2 * H) * PIN;
    const int tid = threadIdx.x;
    const int mp = m >> d, np = n >> d;
    const bool bigR = mp > CR, bigC = np > CC;
    const int HR = bigR ? H : 0, HC = bigC ? H : 0;
    const int NR = CR + 2 * HR, NC = CC + 2 * HC;
    const unsigned mNR = wx_magic((unsigned)NR);
    const int lmp = 31 - __clz(mp), lnp = 31 - __clz(np);
    int R0, C0;
    if (act) {
        const int tpr = mp / CR, tpn = tpr * (np / CC);
        const int a = (int)blockIdx.x / tpn, t = (int)blockIdx.x - a * tpn;
        R0 = act[2 * a] * mp + (t % tpr) * CR;
        C0 = act[2 * a + 1] * np + (t / tpr) * CC;
    } else {
        const int tiles_r = m / CR;
        R0 = (int)(blockIdx.x % tiles_r) * CR;
        C0 = (int)(blockIdx.x / tiles_r) * CC;
    }
    if (status && !act) {
        // several small nodes per tile: nothing to do unless the tree decomposes one of them
        const int nnr = bigR ? 1 : CR >> lmp, nnc = bigC ? 1 : CC >> lnp;
        int any = 0;
        for (int e = tid; e < nnr * nnc; e += 256)
            any |= wx_quad_active(status, nstatus, d, (R0 >> lmp) + e % nnr, (C0 >> lnp) + e / nnr) ? 1 : 0;
        if (!__syncthreads_or(any)) return;
    }
    const int nbR = R0 & ~(mp - 1), nbC = C0 & ~(np - 1);          // node of the tile (big nodes)
    const T *simg = src + (int64_t)blockIdx.y * src_img;
    T *dimg = dst + (int64_t)blockIdx.y * dst_img;
    T *iimg = dst_int ? dst_int + (int64_t)blockIdx.y * int_img : dimg;
    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];

    // explicit batches of independent loads (normally the whole tile in one batch): every load of a batch is in
    // flight before the first LDS store.  Measured on 512 x 512 x 512 Float32, L = 6: 3.8 ms with a dependent
    // index walk, 3.0 ms with a plain loop, 2.45 ms with batches of 8, 2.27 ms with the whole tile in flight.
    constexpr int NBMAX = ((CR + 2 * H) * (CC + 2 * H) + 255) / 256;
    constexpr int NB = NBMAX < 24 ? NBMAX : 24;
    for (int e0 = tid; e0 < NR * NC; e0 += NB * 256) {
        T v[NB];
        int at[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = e0 + u * 256;
            at[u] = -1;
            if (e < NR * NC) {
                const int lc = wx_mdiv(e, mNR), lr = e - lc * NR;
                const int gr = bigR ? nbR + ((R0 - nbR + lr - H) & (mp - 1)) : R0 + lr;
                const int gc = bigC ? nbC + ((C0 - nbC + lc - H) & (np - 1)) : C0 + lc;
                v[u] = simg[(int64_t)gc * m + gr];
                at[u] = lc * PIN + lr;
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) if (at[u] >= 0) in[at[u]] = v[u];
    }
    __syncthreads();
    // columns: item = (staged column, group of OPT row pairs)
    const int dgi = 256 / NC, dli = 256 - dgi * NC;
    int ig = tid / NC, lc = tid - ig * NC;
    for (int e = tid; e < NC * (CR / 2 / OPT); e += 256, lc += dli, ig += dgi) {
        if (lc >= NC) { lc -= NC; ++ig; }
        const int i0 = ig * OPT;
        const T *col = in + lc * PIN;
        T w[W];
        if (bigR) {
#pragma unroll
            for (int k = 0; k < W; ++k) w[k] = col[2 * i0 + k];
        } else {
            const int nb = (2 * i0) & ~(mp - 1), p = 2 * i0 - nb - H;
#pragma unroll
            for (int k = 0; k < W; ++k) w[k] = col[nb + ((p + k) & (mp - 1))];
        }
        T *to = tmp + lc * PT;
#pragma unroll
        for (int s2 = 0; s2 < OPT; ++s2) {
            T a = 0, dd = 0;
#pragma unroll
            for (int t = 0; t < F; ++t) {
                a = fma(q[t], w[2 * s2 + H + t], a);
                dd = fma((t & 1) ? -q[t] : q[t], w[2 * s2 + 1 + H - t], dd);
            }
            to[i0 + s2] = a;
            to[CR / 2 + i0 + s2] = dd;
        }
    }
    __syncthreads();
    // rows: item = (row of the intermediate, group of OPT column pairs)
    for (int e = tid; e < CR * (CC / 2 / OPT); e += 256) {
        const int jg = e / CR, tr = e - jg * CR;
        const int j0 = jg * OPT;
        T w[W];
        if (bigC) {
#pragma unroll
            for (int k = 0; k < W; ++k) w[k] = tmp[(2 * j0 + k) * PT + tr];
        } else {
            const int nb = (2 * j0) & ~(np - 1), p = 2 * j0 - nb - H;
#pragma unroll
            for (int k = 0; k < W; ++k) w[k] = tmp[(nb + ((p + k) & (np - 1))) * PT + tr];
        }
        const int hi = tr >= CR / 2, i = tr - hi * (CR / 2);
        const int rs = R0 + 2 * i, nr = rs & ~(mp - 1);
        const int grow = nr + hi * (mp >> 1) + ((rs - nr) >> 1);
        T *olo = dimg, *ohi = dimg;                      // destinations of the low / high column subbands
        if (status) {
            const int64_t h = wx_quad_heap(d, rs >> lmp, (C0 + 2 * j0) >> lnp);     // all OPT pairs lie in one node
            if (!(h <= nstatus && status[h - 1])) continue;
            const int64_t c0 = 4 * h - 2 + 2 * hi;
            if (c0 <= nstatus && status[c0 - 1]) olo = iimg;
            if (c0 + 1 <= nstatus && status[c0]) ohi = iimg;
        }
#pragma unroll
        for (int s2 = 0; s2 < OPT; ++s2) {
            T a = 0, dd = 0;
#pragma unroll
            for (int t = 0; t < F; ++t) {
                a = fma(q[t], w[2 * s2 + H + t], a);
                dd = fma((t & 1) ? -q[t] : q[t], w[2 * s2 + 1 + H - t], dd);
            }
            const int cs = C0 + 2 * (j0 + s2), nc = cs & ~(np - 1);
            const int gcol = nc + ((cs - nc) >> 1);
            olo[(int64_t)gcol * m + grow] = a;
            ohi[(int64_t)(gcol + (np >> 1)) * m + grow] = dd;
        }
    }
}

// ---- the same level with persistent workgroups: the next tile's loads fly while this tile is filtered -----------------
// k_dwt2d_level_tile issues a tile's loads, waits for them, filters, stores and ends: with three workgroups per CU (43-50 KiB of
// LDS each) nothing hides the load latency (a level of a GiB of 1024 x 1024 Float32 images: 0.8 ms, a quarter of the HBM peak).
// Here a workgroup walks tiles (tile index fastest, then image) and holds the NEXT tile's samples in registers while it works on
// the current one in LDS.  For the modes whose tiles all do work (no tree, or the active-node list `act`) and tiles + halo of at
// most NB x 256 samples.
template <typename T, int F, int CR, int CC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_dwt2d_level_tile_p(const T *__restrict__ src, T *__restrict__ dst,
                                                            int64_t src_img, int64_t dst_img, int m, int n, int d,
                                                            WxFilt filt, T *__restrict__ dst_int, int64_t int_img,
                                                            const uint8_t *__restrict__ status, int64_t nstatus,
                                                            const int *__restrict__ act, unsigned tiles, int64_t total)
{
    constexpr int OPT = 4, H = F - 2, W = 2 * F + 2 * OPT - 4;
    constexpr int PIN = (CR + 2 * H) | 1;
    constexpr int PT = CR | 1;
    constexpr int NB = ((CR + 2 * H) * (CC + 2 * H) + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char wx_smem2[];
    T *in = reinterpret_cast<T *>(wx_smem2);
    T *tmp = in + (CC + 2 * H) * PIN;
    const int tid = threadIdx.x;
    const int mp = m >> d, np = n >> d;
    const bool bigR = mp > CR, bigC = np > CC;
    const int HR = bigR ? H : 0, HC = bigC ? H : 0;
    const int NR = CR + 2 * HR, NC = CC + 2 * HC;
    const unsigned mNR = wx_magic((unsigned)NR);
    const int lmp = 31 - __clz(mp), lnp = 31 - __clz(np);
    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];
    auto coords = [&](int64_t t, int &R0, int &C0, int64_t &img) {
        img = t / tiles;
        const int tt = (int)(t - img * tiles);
        if (act) {
            const int tpr = mp / CR, tpn = tpr * (np / CC);
            const int a = tt / tpn, w = tt - a * tpn;
            R0 = act[2 * a] * mp + (w % tpr) * CR;
            C0 = act[2 * a + 1] * np + (w / tpr) * CC;
        } else {
            const int tiles_r = m / CR;
            R0 = (tt % tiles_r) * CR;
            C0 = (tt / tiles_r) * CC;
        }
    };
    auto fetch = [&](int64_t t, T (&v)[NB]) {
        int R0, C0; int64_t img;
        coords(t, R0, C0, img);
        const int nbR = R0 & ~(mp - 1), nbC = C0 & ~(np - 1);
        const T *simg = src + img * src_img;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + u * 256;
            if (e < NR * NC) {
                const int lc = wx_mdiv(e, mNR), lr = e - lc * NR;
                const int gr = bigR ? nbR + ((R0 - nbR + lr - H) & (mp - 1)) : R0 + lr;
                const int gc = bigC ? nbC + ((C0 - nbC + lc - H) & (np - 1)) : C0 + lc;
                v[u] = simg[(int64_t)gc * m + gr];
            }
        }
    };
    T v[NB];
    int64_t t = blockIdx.x;
    if (t < total) fetch(t, v);
    for (; t < total; t += gridDim.x) {
        int R0, C0; int64_t img;
        coords(t, R0, C0, img);
        T *dimg = dst + img * dst_img;
        T *iimg = dst_int ? dst_int + img * int_img : dimg;
        // (the LDS address of staged element e is recomputed per tile: 3 waves per SIMD need <= 168 registers, and a table of
        // NB addresses beside the NB prefetched samples does not fit)
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + u * 256;
            if (e < NR * NC) { const int lc = wx_mdiv(e, mNR), lr = e - lc * NR; in[lc * PIN + lr] = v[u]; }
        }
        __syncthreads();
        if (t + gridDim.x < total) fetch(t + gridDim.x, v);          // in flight during both passes
        // columns: item = (staged column, group of OPT row pairs)
        const int dgi = 256 / NC, dli = 256 - dgi * NC;
        int ig = tid / NC, lc = tid - ig * NC;
        for (int e = tid; e < NC * (CR / 2 / OPT); e += 256, lc += dli, ig += dgi) {
            if (lc >= NC) { lc -= NC; ++ig; }
            const int i0 = ig * OPT;
            const T *col = in + lc * PIN;
            T w[W];
            if (bigR) {
#pragma unroll
                for (int k = 0; k < W; ++k) w[k] = col[2 * i0 + k];
            } else {
                const int nb = (2 * i0) & ~(mp - 1), p = 2 * i0 - nb - H;
#pragma unroll
                for (int k = 0; k < W; ++k) w[k] = col[nb + ((p + k) & (mp - 1))];
            }
            T *to = tmp + lc * PT;
#pragma unroll
            for (int s2 = 0; s2 < OPT; ++s2) {
                T a = 0, dd = 0;
#pragma unroll
                for (int k = 0; k < F; ++k) {
                    a = fma(q[k], w[2 * s2 + H + k], a);
                    dd = fma((k & 1) ? -q[k] : q[k], w[2 * s2 + 1 + H - k], dd);
                }
                to[i0 + s2] = a;
                to[CR / 2 + i0 + s2] = dd;
            }
        }
        __syncthreads();
        // rows: item = (row of the intermediate, group of OPT column pairs)
        for (int e = tid; e < CR * (CC / 2 / OPT); e += 256) {
            const int jg = e / CR, tr = e - jg * CR;
            const int j0 = jg * OPT;
            T w[W];
            if (bigC) {
#pragma unroll
                for (int k = 0; k < W; ++k) w[k] = tmp[(2 * j0 + k) * PT + tr];
            } else {
                const int nb = (2 * j0) & ~(np - 1), p = 2 * j0 - nb - H;
#pragma unroll
                for (int k = 0; k < W; ++k) w[k] = tmp[(nb + ((p + k) & (np - 1))) * PT + tr];
            }
            const int hi = tr >= CR / 2, i = tr - hi * (CR / 2);
            const int rs = R0 + 2 * i, nr = rs & ~(mp - 1);
            const int grow = nr + hi * (mp >> 1) + ((rs - nr) >> 1);
            T *olo = dimg, *ohi = dimg;
            if (status) {
                const int64_t h = wx_quad_heap(d, rs >> lmp, (C0 + 2 * j0) >> lnp);
                if (!(h <= nstatus && status[h - 1])) continue;
                const int64_t c0 = 4 * h - 2 + 2 * hi;
                if (c0 <= nstatus && status[c0 - 1]) olo = iimg;
                if (c0 + 1 <= nstatus && status[c0]) ohi = iimg;
            }
#pragma unroll
            for (int s2 = 0; s2 < OPT; ++s2) {
                T a = 0, dd = 0;
#pragma unroll
                for (int k = 0; k < F; ++k) {
                    a = fma(q[k], w[2 * s2 + H + k], a);
                    dd = fma((k & 1) ? -q[k] : q[k], w[2 * s2 + 1 + H - k], dd);
                }
                const int cs = C0 + 2 * (j0 + s2), nc = cs & ~(np - 1);
                const int gcol = nc + ((cs - nc) >> 1);
                olo[(int64_t)gcol * m + grow] = a;
                ohi[(int64_t)(gcol + (np >> 1)) * m + grow] = dd;
            }
        }
        // the next tile's samples overwrite `in` only after this barrier; `tmp` is rewritten after the barrier that follows them
        __syncthreads();
    }
}

struct WxTileTree {                       // tree-driven level: see k_dwt2d_level_tile
    void *dst_int = nullptr;
    int64_t int_img = 0;
    const uint8_t *status = nullptr;
    int64_t nstatus = 0;
    const int *act = nullptr;             // used when the nodes are at least as large as a tile
    int nact = 0;
};

template <typename T> static constexpr int wx_tile_cc() { return sizeof(T) == 4 ? 64 : 32; }
static constexpr int WX_ITILE_CC = 32;                   // columns of a tile of the inverse level kernels (both types)
static constexpr int WX_TILE_CR = 64;

template <typename T, int F, int CR, int CC>
static bool wx_launch_level_tile_F(const T *src, T *dst, int64_t src_img, int64_t dst_img, int m, int n, int d,
                                   int64_t batch, const WxFilt &filt, hipStream_t st, const WxTileTree &tt)
{
    constexpr int H = F - 2;
    const size_t lds = sizeof(T) * ((size_t)(CC + 2 * H) * ((CR + 2 * H) | 1) + (size_t)(CC + 2 * H) * (CR | 1));
    auto kern = k_dwt2d_level_tile<T, F, CR, CC>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    const int mp = m >> d, np = n >> d;
    const bool by_node = tt.status && tt.act && mp >= CR && np >= CC;
    const unsigned tiles = by_node ? (unsigned)(tt.nact * (mp / CR) * (np / CC)) : (unsigned)((m / CR) * (n / CC));
    if (tiles == 0) return true;
    T *di = (T *)tt.dst_int;
    // persistent workgroups with the next tile prefetched (k_dwt2d_level_tile_p): every tile does work (no tree, or the node list)
    static const int persist = wx_getenv("WX_TILE_PERSIST") ? atoi(wx_getenv("WX_TILE_PERSIST")) : 1;
    constexpr int NBP = ((CR + 2 * H) * (CC + 2 * H) + 255) / 256;
    if (persist && NBP <= 24 && F <= 10 && (!tt.status || by_node)) {          // (longer filters spill at three wavefronts per SIMD)
        auto kp = k_dwt2d_level_tile_p<T, F, CR, CC>;
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        const int64_t total = (int64_t)tiles * batch;
        int per_cu = (int)((160 * 1024) / lds);
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        static const int wgs_env = wx_getenv("WX_TILE_WGS") ? atoi(wx_getenv("WX_TILE_WGS")) : 0;
        int64_t grid = (int64_t)256 * (wgs_env > 0 ? wgs_env : per_cu);
        if (grid > total) grid = total;
        hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(256), lds, st, src, dst, src_img, dst_img, m, n, d, filt, di, tt.int_img, tt.status,
                           tt.nstatus, by_node ? tt.act : (const int *)nullptr, tiles, total);
        return true;
    }
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const unsigned bc = (unsigned)(batch - b0 < 65535 ? batch - b0 : 65535);
        hipLaunchKernelGGL(kern, dim3(tiles, bc), dim3(256), lds, st, src + b0 * src_img, dst + b0 * dst_img, src_img,
                           dst_img, m, n, d, filt, di ? di + b0 * tt.int_img : (T *)nullptr, tt.int_img, tt.status,
                           tt.nstatus, by_node ? tt.act : (const int *)nullptr);
    }
    return true;
}

// true when the level ran as one tile pass; false = not applicable (the caller takes the two-pass level)
template <typename T> static bool wx_level_tile_ok(int m, int n, int d, int F)
{
    static const bool off = wx_getenv("WX_LEVEL2D_TILE") && atoi(wx_getenv("WX_LEVEL2D_TILE")) == 0;
    if (off) return false;
    const int mp = m >> d, np = n >> d;
    if ((m & (m - 1)) || (n & (n - 1)) || mp < 8 || np < 8) return false;
    if (m % WX_TILE_CR || n % wx_tile_cc<T>()) return false;
    return F == 2 || F == 4 || F == 6 || F == 8 || F == 10 || F == 12 || F == 14 || F == 16 || F == 18 || F == 20;
}

template <typename T>
static bool wx_launch_level_tile(const T *src, T *dst, int64_t src_img, int64_t dst_img, int m, int n, int d,
                                 int64_t batch, const WxFilt &filt, hipStream_t st, const WxTileTree &tt = WxTileTree())
{
    if (!wx_level_tile_ok<T>(m, n, d, filt.F)) return false;
    constexpr int CR = WX_TILE_CR, CC = wx_tile_cc<T>();
    switch (filt.F) {
#define WX_CASE(FF) case FF: return wx_launch_level_tile_F<T, FF, CR, CC>(src, dst, src_img, dst_img, m, n, d, batch, filt, st, tt);
        WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(14) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
    }
    return false;
}

// ---- tree-driven inverse level in one pass ---------------------------------------------------------------
// The mirror of k_dwt2d_level_tile for idwt_step! 2-D (dwt/dwt_one_level.jl:401-436: along the rows first, then
// down the columns).  A workgroup rebuilds a CR x CC tile of every decomposed node of depth d: the four child
// tiles (F/2 - 1 halo pairs on both sides for nodes larger than the tile, whole nodes wrapping in LDS otherwise)
// are staged from the image that holds them -- a child that was itself rebuilt one level earlier comes from the
// scratch image src_int, a leaf from the caller's coefficients src_leaf -- synthesised along dim 2 into a second
// LDS array ([low; high] rows), along dim 1 into an LDS copy of the output tile (lanes across columns of odd
// pitch: conflict-free) and stored with lanes down the rows.  One read of the children and one write of the
// parent per level instead of two of each.
template <typename T, int F, int CR, int CC>
__global__ __launch_bounds__(256) void k_idwt2d_level_tile(const T *__restrict__ src_leaf, int64_t leaf_img,
                                                           const T *__restrict__ src_int, int64_t int_img,
                                                           T *__restrict__ dst, int64_t dst_img, int m, int n, int d,
                                                           WxFilt filt, const uint8_t *__restrict__ status,
                                                           int64_t nstatus, const int *__restrict__ act)
{
    constexpr int OPT = 4, HF = F / 2, G = HF - 1, WN = G + OPT;
    constexpr int HRm = CR / 2 + 2 * G, HCm = CC / 2 + 2 * G;
    constexpr int PCH = HRm | 1;                         // pitch of a staged child column
    constexpr int PTP = (2 * HRm) | 1;                   // pitch of a column of the intermediate ([low; high] rows)
    constexpr int POUT = CR | 1;                         // pitch of a column of the output tile (aliases the children)
    static_assert(CC * POUT <= 4 * HCm * PCH, "output tile must fit the staged children");
    extern __shared__ __attribute__((aligned(16))) char wx_smem3[];
    T *ch = reinterpret_cast<T *>(wx_smem3);
    T *tmp = ch + 4 * HCm * PCH;
    uint8_t *fl = reinterpret_cast<uint8_t *>(tmp + CC * PTP);       // per (node of the tile, child): 0 skip, 1 leaf, 2 scratch
    T *out = ch;
    const int tid = threadIdx.x;
    const int mp = m >> d, np = n >> d, hr = mp >> 1, hc = np >> 1;
    const int lmp = 31 - __clz(mp), lnp = 31 - __clz(np);
    const bool bigR = mp > CR, bigC = np > CC;
    const int GR = bigR ? G : 0, GC = bigC ? G : 0;
    const int NRc = CR / 2 + 2 * GR, NCc = CC / 2 + 2 * GC;          // staged rows / columns of a child
    const unsigned mNRc = wx_magic((unsigned)NRc), m2NRc = wx_magic((unsigned)(2 * NRc));
    int R0, C0;
    if (act) {
        const int tpr = mp / CR, tpn = tpr * (np / CC);
        const int a = (int)blockIdx.x / tpn, t = (int)blockIdx.x - a * tpn;
        R0 = act[2 * a] * mp + (t % tpr) * CR;
        C0 = act[2 * a + 1] * np + (t / tpr) * CC;
    } else {
        const int tiles_r = m / CR;
        R0 = (int)(blockIdx.x % tiles_r) * CR;
        C0 = (int)(blockIdx.x / tiles_r) * CC;
    }
    const int nnr = bigR ? 1 : CR >> lmp, nnc = bigC ? 1 : CC >> lnp;
    int any = 0;
    for (int e = tid; e < nnr * nnc * 4; e += 256) {
        const int c = e & 3, nd = e >> 2;
        const int64_t h = wx_quad_heap(d, (R0 >> lmp) + nd % nnr, (C0 >> lnp) + nd / nnr);
        uint8_t f = 0;
        if (h <= nstatus && status[h - 1]) {
            const int64_t hch = 4 * h - 2 + c;
            f = (hch <= nstatus && status[hch - 1]) ? 2 : 1;
        }
        fl[e] = f;
        any |= f;
    }
    if (!__syncthreads_or(any)) return;
    const int nbR = R0 & ~(mp - 1), nbC = C0 & ~(np - 1);
    const T *limg = src_leaf + (int64_t)blockIdx.y * leaf_img;
    const T *iimg = src_int ? src_int + (int64_t)blockIdx.y * int_img : limg;
    T *dimg = dst + (int64_t)blockIdx.y * dst_img;
    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];

    // stage the four children: every load in flight before the first LDS store (see k_dwt2d_level_tile).  A thread keeps ONE staged
    // row (tid & 63; rows beyond the NRc <= 64 staged ones idle) and walks the columns four apart, so the row part of the address is
    // made once per child and a column costs a mask and an add -- an element index decoded by two divisions per element made this
    // loop 500 of the kernel's 950 vector instructions per thread (round 4, PMC: the vector ALUs were the bound of the Float32 level).
    {
        static_assert(HRm <= 64, "one staged row per lane of a wavefront");
        constexpr int ITER = (HCm + 3) / 4;
        const int lr = tid & 63, lcg = tid >> 6;
        const bool rok = lr < NRc;
        T v[4][ITER];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int grow, jn = 0;
            if (bigR) grow = nbR + (c >> 1) * hr + ((((R0 - nbR) >> 1) + lr - G) & (hr - 1));
            else { jn = lr >> (lmp - 1); grow = R0 + jn * mp + (c >> 1) * hr + (lr & (hr - 1)); }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int lc = lcg + 4 * it;
                v[c][it] = (T)0;
                if (rok && lc < NCc) {
                    int gcol, kn = 0;
                    if (bigC) gcol = nbC + (c & 1) * hc + ((((C0 - nbC) >> 1) + lc - G) & (hc - 1));
                    else { kn = lc >> (lnp - 1); gcol = C0 + kn * np + (c & 1) * hc + (lc & (hc - 1)); }
                    const uint8_t f = fl[(kn * nnr + jn) * 4 + c];
                    if (f) v[c][it] = (f == 2 ? iimg : limg)[(int64_t)gcol * m + grow];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int lc = lcg + 4 * it;
                if (rok && lc < NCc) ch[(c * HCm + lc) * PCH + lr] = v[c][it];
            }
    }
    __syncthreads();
    // dim 2: item = (row half, staged child row, group of OPT column pairs); lanes down the rows
    for (int e = tid; e < 2 * NRc * (CC / 2 / OPT); e += 256) {
        const int g = wx_mdiv(e, m2NRc), r2 = e - g * (2 * NRc);
        const int rh = wx_mdiv(r2, mNRc), lr = r2 - rh * NRc;
        const T *ca = ch + (rh * 2) * HCm * PCH + lr, *cd = ch + (rh * 2 + 1) * HCm * PCH + lr;
        T aw[WN], dw[WN];
        if (bigC) {
#pragma unroll
            for (int t = 0; t < WN; ++t) { aw[t] = ca[(g * OPT + t) * PCH]; dw[t] = cd[(g * OPT + t + G) * PCH]; }
        } else {
            const int base = (g * OPT) & ~(hc - 1), p = g * OPT - base;
#pragma unroll
            for (int t = 0; t < WN; ++t) {
                aw[t] = ca[(base + ((p - G + t) & (hc - 1))) * PCH];
                dw[t] = cd[(base + ((p + t) & (hc - 1))) * PCH];
            }
        }
        T *to = tmp + rh * HRm + lr;
#pragma unroll
        for (int s2 = 0; s2 < OPT; ++s2) {
            T v0 = 0, v1 = 0;
#pragma unroll
            for (int mm = 0; mm < HF; ++mm) {
                v0 = fma(q[2 * mm], aw[s2 + G - mm], v0); v0 = fma(-q[2 * mm + 1], dw[s2 + mm], v0);
                v1 = fma(q[2 * mm + 1], aw[s2 + G - mm], v1); v1 = fma(q[2 * mm], dw[s2 + mm], v1);
            }
            to[(2 * (g * OPT + s2)) * PTP] = v0;
            to[(2 * (g * OPT + s2) + 1) * PTP] = v1;
        }
    }
    __syncthreads();
    // dim 1: item = (column, group of OPT row pairs); lanes across the columns
    for (int e = tid; e < CC * (CR / 2 / OPT); e += 256) {
        const int g = e / CC, c = e - g * CC;
        const T *lo = tmp + c * PTP, *hi = lo + HRm;
        T aw[WN], dw[WN];
        if (bigR) {
#pragma unroll
            for (int t = 0; t < WN; ++t) { aw[t] = lo[g * OPT + t]; dw[t] = hi[g * OPT + t + G]; }
        } else {
            const int base = (g * OPT) & ~(hr - 1), p = g * OPT - base;
#pragma unroll
            for (int t = 0; t < WN; ++t) { aw[t] = lo[base + ((p - G + t) & (hr - 1))]; dw[t] = hi[base + ((p + t) & (hr - 1))]; }
        }
        T *to = out + c * POUT;
#pragma unroll
        for (int s2 = 0; s2 < OPT; ++s2) {
            T v0 = 0, v1 = 0;
#pragma unroll
            for (int mm = 0; mm < HF; ++mm) {
                v0 = fma(q[2 * mm], aw[s2 + G - mm], v0); v0 = fma(-q[2 * mm + 1], dw[s2 + mm], v0);
                v1 = fma(q[2 * mm + 1], aw[s2 + G - mm], v1); v1 = fma(q[2 * mm], dw[s2 + mm], v1);
            }
            to[2 * (g * OPT + s2)] = v0;
            to[2 * (g * OPT + s2) + 1] = v1;
        }
    }
    __syncthreads();
    for (int e = tid; e < CR * CC; e += 256) {
        const int c = e / CR, r = e - c * CR;
        const int jn = bigR ? 0 : r >> lmp, kn = bigC ? 0 : c >> lnp;
        if (fl[(kn * nnr + jn) * 4]) dimg[(int64_t)(C0 + c) * m + R0 + r] = out[c * POUT + r];
    }
}

// ---- the inverse level with persistent workgroups and the next tile's children prefetched (see k_dwt2d_level_tile_p) -------
// For the node-list mode (nodes at least as large as a tile: one node per tile, every tile active): the flags of the four children
// come straight from the status bytes of the tile's node (wave-uniform), no flag table in LDS.
template <typename T, int F, int CR, int CC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_idwt2d_level_tile_p(const T *__restrict__ src_leaf, int64_t leaf_img,
                                                             const T *__restrict__ src_int, int64_t int_img,
                                                             T *__restrict__ dst, int64_t dst_img, int m, int n, int d,
                                                             WxFilt filt, const uint8_t *__restrict__ status,
                                                             int64_t nstatus, const int *__restrict__ act, unsigned tiles, int64_t total)
{
    constexpr int OPT = 4, HF = F / 2, G = HF - 1, WN = G + OPT;
    constexpr int HRm = CR / 2 + 2 * G, HCm = CC / 2 + 2 * G;
    constexpr int PCH = HRm | 1;
    constexpr int PTP = (2 * HRm) | 1;
    constexpr int POUT = CR | 1;
    constexpr int NB = (4 * HRm * HCm + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char wx_smem3[];
    T *ch = reinterpret_cast<T *>(wx_smem3);
    T *tmp = ch + 4 * HCm * PCH;
    T *out = ch;
    const int tid = threadIdx.x;
    const int mp = m >> d, np = n >> d, hr = mp >> 1, hc = np >> 1;
    const int lmp = 31 - __clz(mp), lnp = 31 - __clz(np);
    const bool bigR = mp > CR, bigC = np > CC;              // mp >= CR and np >= CC here: a tile lies in one node
    const int GR = bigR ? G : 0, GC = bigC ? G : 0;
    const int NRc = CR / 2 + 2 * GR, NCc = CC / 2 + 2 * GC;
    const int per = NRc * NCc, tot = 4 * per;
    const unsigned mNRc = wx_magic((unsigned)NRc), mper = wx_magic((unsigned)per), m2NRc = wx_magic((unsigned)(2 * NRc));
    T q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (T)filt.q[k];
    auto coords = [&](int64_t t, int &R0, int &C0, int64_t &img) {
        img = t / tiles;
        const int tt = (int)(t - img * tiles);
        const int tpr = mp / CR, tpn = tpr * (np / CC);
        const int a = tt / tpn, w = tt - a * tpn;
        R0 = act[2 * a] * mp + (w % tpr) * CR;
        C0 = act[2 * a + 1] * np + (w / tpr) * CC;
    };
    auto fetch = [&](int64_t t, T (&v)[NB]) {
        int R0, C0; int64_t img;
        coords(t, R0, C0, img);
        const int nbR = R0 & ~(mp - 1), nbC = C0 & ~(np - 1);
        const T *limg = src_leaf + img * leaf_img;
        const T *iimg = src_int ? src_int + img * int_img : limg;
        const int64_t h = wx_quad_heap(d, R0 >> lmp, C0 >> lnp);
        const T *from[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t hch = 4 * h - 2 + c;
            from[c] = (hch <= nstatus && status[hch - 1]) ? iimg : limg;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + u * 256;
            if (e < tot) {
                const int c = wx_mdiv(e, mper), r2 = e - c * per;
                const int lc = wx_mdiv(r2, mNRc), lr = r2 - lc * NRc;
                const int grow = bigR ? nbR + (c >> 1) * hr + ((((R0 - nbR) >> 1) + lr - G) & (hr - 1)) : R0 + (c >> 1) * hr + lr;
                const int gcol = bigC ? nbC + (c & 1) * hc + ((((C0 - nbC) >> 1) + lc - G) & (hc - 1)) : C0 + (c & 1) * hc + lc;
                const T *fp = c == 0 ? from[0] : (c == 1 ? from[1] : (c == 2 ? from[2] : from[3]));
                v[u] = fp[(int64_t)gcol * m + grow];
            }
        }
    };
    T v[NB];
    int64_t t = blockIdx.x;
    if (t < total) fetch(t, v);
    for (; t < total; t += gridDim.x) {
        int R0, C0; int64_t img;
        coords(t, R0, C0, img);
        T *dimg = dst + img * dst_img;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + u * 256;
            if (e < tot) {
                const int c = wx_mdiv(e, mper), r2 = e - c * per;
                const int lc = wx_mdiv(r2, mNRc), lr = r2 - lc * NRc;
                ch[(c * HCm + lc) * PCH + lr] = v[u];
            }
        }
        __syncthreads();
        if (t + gridDim.x < total) fetch(t + gridDim.x, v);
        // dim 2: item = (row half, staged child row, group of OPT column pairs); lanes down the rows
        for (int e = tid; e < 2 * NRc * (CC / 2 / OPT); e += 256) {
            const int g = wx_mdiv(e, m2NRc), r2 = e - g * (2 * NRc);
            const int rh = wx_mdiv(r2, mNRc), lr = r2 - rh * NRc;
            const T *ca = ch + (rh * 2) * HCm * PCH + lr, *cd = ch + (rh * 2 + 1) * HCm * PCH + lr;
            T aw[WN], dw[WN];
            if (bigC) {
#pragma unroll
                for (int k = 0; k < WN; ++k) { aw[k] = ca[(g * OPT + k) * PCH]; dw[k] = cd[(g * OPT + k + G) * PCH]; }
            } else {
                const int base = (g * OPT) & ~(hc - 1), p = g * OPT - base;
#pragma unroll
                for (int k = 0; k < WN; ++k) {
                    aw[k] = ca[(base + ((p - G + k) & (hc - 1))) * PCH];
                    dw[k] = cd[(base + ((p + k) & (hc - 1))) * PCH];
                }
            }
            T *to = tmp + rh * HRm + lr;
#pragma unroll
            for (int s2 = 0; s2 < OPT; ++s2) {
                T v0 = 0, v1 = 0;
#pragma unroll
                for (int mm = 0; mm < HF; ++mm) {
                    v0 = fma(q[2 * mm], aw[s2 + G - mm], v0); v0 = fma(-q[2 * mm + 1], dw[s2 + mm], v0);
                    v1 = fma(q[2 * mm + 1], aw[s2 + G - mm], v1); v1 = fma(q[2 * mm], dw[s2 + mm], v1);
                }
                to[(2 * (g * OPT + s2)) * PTP] = v0;
                to[(2 * (g * OPT + s2) + 1) * PTP] = v1;
            }
        }
        __syncthreads();
        // dim 1: item = (column, group of OPT row pairs); lanes across the columns
        for (int e = tid; e < CC * (CR / 2 / OPT); e += 256) {
            const int g = e / CC, c = e - g * CC;
            const T *lo = tmp + c * PTP, *hi = lo + HRm;
            T aw[WN], dw[WN];
            if (bigR) {
#pragma unroll
                for (int k = 0; k < WN; ++k) { aw[k] = lo[g * OPT + k]; dw[k] = hi[g * OPT + k + G]; }
            } else {
                const int base = (g * OPT) & ~(hr - 1), p = g * OPT - base;
#pragma unroll
                for (int k = 0; k < WN; ++k) { aw[k] = lo[base + ((p - G + k) & (hr - 1))]; dw[k] = hi[base + ((p + k) & (hr - 1))]; }
            }
            T *to = out + c * POUT;
#pragma unroll
            for (int s2 = 0; s2 < OPT; ++s2) {
                T v0 = 0, v1 = 0;
#pragma unroll
                for (int mm = 0; mm < HF; ++mm) {
                    v0 = fma(q[2 * mm], aw[s2 + G - mm], v0); v0 = fma(-q[2 * mm + 1], dw[s2 + mm], v0);
                    v1 = fma(q[2 * mm + 1], aw[s2 + G - mm], v1); v1 = fma(q[2 * mm], dw[s2 + mm], v1);
                }
                to[2 * (g * OPT + s2)] = v0;
                to[2 * (g * OPT + s2) + 1] = v1;
            }
        }
        __syncthreads();
        for (int e = tid; e < CR * CC; e += 256) {
            const int c = e / CR, r = e - c * CR;
            dimg[(int64_t)(C0 + c) * m + R0 + r] = out[c * POUT + r];
        }
        __syncthreads();
    }
}

template <typename T, int F, int CR, int CC>
static bool wx_launch_ilevel_tile_F(const T *src_leaf, int64_t leaf_img, const T *src_int, T *dst, int64_t dst_img, int m,
                                    int n, int d, int64_t batch, const WxFilt &filt, hipStream_t st, const WxTileTree &tt)
{
    constexpr int G = F / 2 - 1, HRm = CR / 2 + 2 * G, HCm = CC / 2 + 2 * G;
    const size_t lds = sizeof(T) * ((size_t)4 * HCm * (HRm | 1) + (size_t)CC * ((2 * HRm) | 1)) + 256;
    auto kern = k_idwt2d_level_tile<T, F, CR, CC>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    const int mp = m >> d, np = n >> d;
    const bool by_node = tt.act && mp >= CR && np >= CC;
    const unsigned tiles = by_node ? (unsigned)(tt.nact * (mp / CR) * (np / CC)) : (unsigned)((m / CR) * (n / CC));
    if (tiles == 0) return true;
    static const int persist = wx_getenv("WX_TILE_PERSIST") ? atoi(wx_getenv("WX_TILE_PERSIST")) : 1;
    constexpr int NBP = (4 * HRm * HCm + 255) / 256;
    if (persist && NBP <= 24 && F <= 8 && sizeof(T) == 8 && by_node) {          // (Float32: no gain, 1.32 vs 1.34 ms: instruction-bound)     // (Float32 at 8 taps spills 124 bytes per lane at three wavefronts per SIMD: 1.5 -> 1.8 ms)
        auto kp = k_idwt2d_level_tile_p<T, F, CR, CC>;
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        const int64_t total = (int64_t)tiles * batch;
        int per_cu = (int)((160 * 1024) / lds);
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        static const int wgs_env = wx_getenv("WX_TILE_WGS") ? atoi(wx_getenv("WX_TILE_WGS")) : 0;
        int64_t grid = (int64_t)256 * (wgs_env > 0 ? wgs_env : per_cu);
        if (grid > total) grid = total;
        hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(256), lds, st, src_leaf, leaf_img, src_int, tt.int_img, dst, dst_img, m, n, d, filt,
                           tt.status, tt.nstatus, tt.act, tiles, total);
        return true;
    }
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const unsigned bc = (unsigned)(batch - b0 < 65535 ? batch - b0 : 65535);
        hipLaunchKernelGGL(kern, dim3(tiles, bc), dim3(256), lds, st, src_leaf + b0 * leaf_img, leaf_img,
                           src_int ? src_int + b0 * tt.int_img : (const T *)nullptr, tt.int_img, dst + b0 * dst_img, dst_img,
                           m, n, d, filt, tt.status, tt.nstatus, by_node ? tt.act : (const int *)nullptr);
    }
    return true;
}

template <typename T>
static bool wx_launch_ilevel_tile(const T *src_leaf, int64_t leaf_img, const T *src_int, T *dst, int64_t dst_img, int m,
                                  int n, int d, int64_t batch, const WxFilt &filt, hipStream_t st, const WxTileTree &tt)
{
    if (!wx_level_tile_ok<T>(m, n, d, filt.F) || !tt.status) return false;
    // (Float32: 32 columns per tile like Float64 -- half the prefetched samples per lane, so the persistent form fits three wavefronts
    // per SIMD without spilling)
    constexpr int CR = WX_TILE_CR, CC = WX_ITILE_CC;
    switch (filt.F) {
#define WX_CASE(FF) case FF: return wx_launch_ilevel_tile_F<T, FF, CR, CC>(src_leaf, leaf_img, src_int, dst, dst_img, m, n, d, batch, filt, st, tt);
        WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(14) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
    }
    return false;
}

static int wx_grid2(int64_t total)
{
    int64_t g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}

template <typename T, bool INVERSE>
static void wx_launch_level2d(const T *src, T *tmp, T *dst, int64_t src_img, int64_t dst_img, int m, int n, int d,
                              int64_t batch, const WxFilt &filt, const uint8_t *status, int64_t nstatus,
                              hipStream_t st, int copy_inactive = 1, const int *act = nullptr, int nact = 0)
{
    const int64_t mn = (int64_t)m * n;
    const int g = wx_grid2(act ? batch * nact * ((int64_t)(m >> d) * (n >> d)) / 2 : batch * mn / 2);
    if (!INVERSE) {
        hipLaunchKernelGGL((k_dwt2d_dim1<T, false>), dim3(g), dim3(256), 0, st, src, tmp, src_img, mn, m, n, d, batch,
                           filt, status, nstatus, copy_inactive, act, nact);
        hipLaunchKernelGGL((k_dwt2d_dim2<T, false>), dim3(g), dim3(256), 0, st, (const T *)tmp, dst, mn, dst_img, m, n,
                           d, batch, filt, status, nstatus, copy_inactive, act, nact);
    } else {
        hipLaunchKernelGGL((k_dwt2d_dim2<T, true>), dim3(g), dim3(256), 0, st, src, tmp, src_img, mn, m, n, d, batch,
                           filt, status, nstatus, copy_inactive, act, nact);
        hipLaunchKernelGGL((k_dwt2d_dim1<T, true>), dim3(g), dim3(256), 0, st, (const T *)tmp, dst, mn, dst_img, m, n,
                           d, batch, filt, status, nstatus, copy_inactive, act, nact);
    }
}

// wpd 2-D: y is (m, n, L+1, batch); tmp holds m*n*batch elements
template <typename T>
int wx_dev_wpd2d(const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const WxFilt &filt, T *tmp,
                 hipStream_t st)
{
    if (batch == 0 || m * n == 0) return WX_OK;
    const int64_t mn = m * n, yimg = mn * (L + 1);
    WX_HIP_CHECK(hipMemcpy2DAsync(y, yimg * sizeof(T), x, mn * sizeof(T), mn * sizeof(T), batch,
                                  hipMemcpyDeviceToDevice, st));
    for (int d = 0; d < L; ++d) {
        if (wx_launch_level_tile<T>(y + d * mn, y + (d + 1) * mn, yimg, yimg, (int)m, (int)n, d, batch, filt, st)) continue;
        wx_launch_level2d<T, false>(y + d * mn, tmp, y + (d + 1) * mn, yimg, yimg, (int)m, (int)n, d, batch, filt,
                                    nullptr, 0, st);
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

// copies the blocks (jr, jc) of depth d listed in `act` -- (m >> d) x (n >> d) elements each -- of every image from src to dst
template <typename T>
__global__ __launch_bounds__(256) void k_copy_blocks2d(const T *__restrict__ src, T *__restrict__ dst, int64_t src_img, int64_t dst_img, int m,
                                                       int d, int64_t batch, const int *__restrict__ act, int nact, int n)
{
    const int mp = m >> d, np = n >> d;
    const int64_t per = (int64_t)mp * np, total = per * nact * batch;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = g / (per * nact);
        int64_t r = g - b * per * nact;
        const int a = (int)(r / per);
        r -= (int64_t)a * per;
        const int c = (int)(r / mp), i = (int)(r - (int64_t)c * mp);
        const int64_t e = (int64_t)(act[2 * a + 1] * np + c) * m + act[2 * a] * mp + i;
        dst[b * dst_img + e] = src[b * src_img + e];
    }
}

// wpt / iwpt 2-D on (m, n, batch); tmp and pong hold m*n*batch elements each (pong only if L > 1)
template <typename T>
int wx_dev_wpt2d(const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const WxFilt &filt,
                 const uint8_t *status, int64_t nstatus, T *tmp, T *pong, bool inverse, int64_t in_img,
                 hipStream_t st, const uint8_t *htree)
{
    if (batch == 0 || m * n == 0) return WX_OK;
    const int64_t mn = m * n;
    if (L == 0) {
        WX_HIP_CHECK(hipMemcpy2DAsync(y, mn * sizeof(T), x, in_img * sizeof(T), mn * sizeof(T), batch,
                                      hipMemcpyDeviceToDevice, st));
        return WX_OK;
    }
    if (status) {
        // Tree-driven: a level rewrites exactly the blocks of the nodes it decomposes (children live where the
        // parent was), so all levels run in place on y through tmp and skip every other block -- the work is
        // proportional to the decomposed area (a dwt tree touches 1 + 1/4 + 1/16 + ... images, not L).
        // (block-row, block-column) of the decomposed nodes of every depth, from the host copy of the tree
        std::vector<const int *> dact((size_t)L, nullptr);
        std::vector<int> nact((size_t)L, 0);
        if (htree) {
            for (int d = 0; d < L; ++d) {
                std::vector<int> lst;
                int64_t start = 1;
                for (int t = 0; t < d; ++t) start = 4 * start - 2;
                const int64_t cnt = (int64_t)1 << (2 * d);
                for (int64_t mo = 0; mo < cnt; ++mo) {
                    const int64_t hp = start + mo;
                    if (!(hp <= nstatus && htree[hp - 1])) continue;
                    int jr = 0, jc = 0;
                    for (int t = 0; t < d; ++t) { jr |= (int)((mo >> (2 * t + 1)) & 1) << t; jc |= (int)((mo >> (2 * t)) & 1) << t; }
                    lst.push_back(jr); lst.push_back(jc);
                }
                nact[(size_t)d] = (int)(lst.size() / 2);
                if (!lst.empty()) {
                    dact[(size_t)d] = (const int *)wx_const_upload(lst.data(), lst.size() * sizeof(int), st, true);
                    if (!dact[(size_t)d]) return WX_EHIP;
                }
            }
        }
        auto level = [&](const T *src, int64_t simg, int d, bool inv, int copy) {
            if (htree && nact[(size_t)d] == 0) return;                     // nothing decomposed at this depth
            if (inv) wx_launch_level2d<T, true>(src, tmp, y, simg, mn, (int)m, (int)n, d, batch, filt, status, nstatus, st, copy,
                                                htree ? dact[(size_t)d] : nullptr, nact[(size_t)d]);
            else wx_launch_level2d<T, false>(src, tmp, y, simg, mn, (int)m, (int)n, d, batch, filt, status, nstatus, st, copy,
                                             htree ? dact[(size_t)d] : nullptr, nact[(size_t)d]);
        };
        // levels whose nodes are at least 8 x 8 run as one tile pass each; a tree that goes deeper (a pyramid of full depth ends at
        // 1 x 1) continues on the small nodes with the two-pass level, block by block: until round 4 such a tree took the two-pass
        // level for EVERY level, the root's included -- four trips of the whole image (2-D idwtall: 8 % of the HBM peak)
        // (a tile pass walks the whole image unless its nodes are at least a tile large: it pays while the nodes are, or while the
        // decomposed nodes of the level cover at least half of the image)
        int Lt = 0;
        while (Lt < L && wx_level_tile_ok<T>((int)m, (int)n, Lt, filt.F) &&
               (((m >> Lt) >= WX_TILE_CR && (n >> Lt) >= wx_tile_cc<T>()) ||
                2 * (int64_t)nact[(size_t)Lt] * (m >> Lt) * (n >> Lt) >= mn))
            ++Lt;
        auto level2 = [&](const T *src, int64_t simg, T *dstb, T *interm, int d, bool inv) {
            if (nact[(size_t)d] == 0) return;
            if (inv) wx_launch_level2d<T, true>(src, interm, dstb, simg, mn, (int)m, (int)n, d, batch, filt, status, nstatus, st, 0,
                                                dact[(size_t)d], nact[(size_t)d]);
            else wx_launch_level2d<T, false>(src, interm, dstb, simg, mn, (int)m, (int)n, d, batch, filt, status, nstatus, st, 0,
                                             dact[(size_t)d], nact[(size_t)d]);
        };
        if (!inverse && htree && (L == 1 || pong) && Lt >= 1) {
            // one pass per level: a level reads its nodes from the scratch image of its parity (the root from x),
            // sends the children that are decomposed further to the other scratch image and the leaves to y
            T *sc[2] = {tmp, pong};
            for (int d = 0; d < Lt; ++d) {
                if (nact[(size_t)d] == 0) continue;
                WxTileTree tt;
                tt.dst_int = sc[(d + 1) & 1]; tt.int_img = mn;
                tt.status = status; tt.nstatus = nstatus;
                tt.act = dact[(size_t)d]; tt.nact = nact[(size_t)d];
                if (!wx_launch_level_tile<T>(d ? (const T *)sc[d & 1] : x, y, d ? mn : in_img, mn, (int)m, (int)n, d, batch,
                                             filt, st, tt))
                    return wx_set_error(WX_EHIP, "2-D tile level failed to launch");
            }
            // the nodes of depth Lt that are decomposed further sit in sc[Lt & 1]: the first small level takes them from there
            // into y (the other scratch image is its intermediate), the rest run in place on y
            for (int d = Lt; d < L; ++d) level2(d == Lt ? (const T *)sc[Lt & 1] : (const T *)y, mn, y, sc[(Lt + 1) & 1], d, false);
            WX_HIP_CHECK(hipGetLastError());
            return WX_OK;
        }
        if (inverse && htree && (L == 1 || pong) && Lt >= 1) {
            // the mirror image: a level takes the children that were rebuilt one level earlier from the scratch
            // image of their parity and the leaves from x, and writes the parents into the other scratch image (the
            // root into y)
            T *sc[2] = {tmp, pong};
            if (Lt < L && nact[(size_t)Lt]) {
                // the small nodes first, in place in sc[Lt & 1]: the blocks of depth Lt that are decomposed come over from x
                T *B = sc[Lt & 1];
                const int64_t tot = (int64_t)(m >> Lt) * (n >> Lt) * nact[(size_t)Lt] * batch;
                hipLaunchKernelGGL(k_copy_blocks2d<T>, dim3(wx_grid2(tot)), dim3(256), 0, st, x, B, in_img, mn, (int)m, Lt, batch,
                                   dact[(size_t)Lt], nact[(size_t)Lt], (int)n);
                for (int d = L - 1; d >= Lt; --d) level2(B, mn, B, sc[(Lt + 1) & 1], d, true);
            }
            for (int d = Lt - 1; d >= 0; --d) {
                if (nact[(size_t)d] == 0) continue;
                WxTileTree tt;
                tt.int_img = mn;
                tt.status = status; tt.nstatus = nstatus;
                tt.act = dact[(size_t)d]; tt.nact = nact[(size_t)d];
                if (!wx_launch_ilevel_tile<T>(x, in_img, (const T *)sc[(d + 1) & 1], d ? sc[d & 1] : y, mn, (int)m, (int)n, d,
                                              batch, filt, st, tt))
                    return wx_set_error(WX_EHIP, "2-D inverse tile level failed to launch");
            }
            WX_HIP_CHECK(hipGetLastError());
            return WX_OK;
        }
        if (!inverse) {
            level(x, in_img, 0, false, 1);                                 // the root: every block is written
            for (int d = 1; d < L; ++d) level(y, mn, d, false, 0);
        } else {
            WX_HIP_CHECK(hipMemcpy2DAsync(y, mn * sizeof(T), x, in_img * sizeof(T), mn * sizeof(T), batch,
                                          hipMemcpyDeviceToDevice, st));
            for (int d = L - 1; d >= 0; --d) level(y, mn, d, true, 0);
        }
        WX_HIP_CHECK(hipGetLastError());
        return WX_OK;
    }
    const T *src = x;
    int64_t src_img = in_img;
    for (int s = 0; s < L; ++s) {
        const int d = inverse ? L - 1 - s : s;
        T *dst = ((L - 1 - s) & 1) ? pong : y;                 // last step lands in y
        if (inverse)
            wx_launch_level2d<T, true>(src, tmp, dst, src_img, mn, (int)m, (int)n, d, batch, filt, status, nstatus, st);
        else
            wx_launch_level2d<T, false>(src, tmp, dst, src_img, mn, (int)m, (int)n, d, batch, filt, status, nstatus, st);
        src = dst;
        src_img = mn;
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

template <typename T>
int wx_dev_gather_leaves2d(const T *Xw, T *out, int64_t m, int64_t n, int k, int64_t batch, const int *colmap,
                           int nblk, hipStream_t st)
{
    if (batch == 0 || m * n == 0) return WX_OK;
    hipLaunchKernelGGL(k_gather_leaves2d<T>, dim3(wx_grid2(batch * m * n)), dim3(256), 0, st, Xw, out, (int)m, (int)n,
                       k, batch, colmap, nblk, (int)(m / nblk), (int)(n / nblk));
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

#define WX_INST(T)                                                                                            \
    template int wx_dev_wpd2d<T>(const T *, T *, int64_t, int64_t, int, int64_t, const WxFilt &, T *, hipStream_t); \
    template int wx_dev_wpt2d<T>(const T *, T *, int64_t, int64_t, int, int64_t, const WxFilt &, const uint8_t *,  \
                                 int64_t, T *, T *, bool, int64_t, hipStream_t, const uint8_t *);             \
    template int wx_dev_gather_leaves2d<T>(const T *, T *, int64_t, int64_t, int, int64_t, const int *, int, hipStream_t); \
    template bool wx_wpt2d_fast_ok<T>(int64_t, int64_t, int);                                                  \
    template int wx_dev_wpt2d_fast<T>(const T *, T *, int64_t, int64_t, int, int64_t, const WxFilt &, T *, bool, int64_t, hipStream_t);
WX_INST(double)
WX_INST(float)
