// wx_jbb.hip -- joint best basis (JBB) reductions for gfx950.
//
// Reference (paths relative to /root/reference/src/mod):
//   tree_costs(X::Array{T,3}, ::JBB)  bestbasis/bestbasis_tree.jl:150-180
//       EX = sum(X, dims=3)/N, EX2 = sum(X.^2, dims=3)/N, sigma = sqrt(EX2 - EX^2) per (coef, column)
//       redundant: cost[i] = coefcost(sigma[:, i]) / 2^depth(i); otherwise one cost per (level, node)
//   coefcost(x, LoglpCost(p)) = p * sum(log.(abs.(x)));  coefcost(x, NormCost(p)) = norm(x, p)^p
//       bestbasis/bestbasis_costs.jl:127-132
// The moments kernel walks the signal axis sequentially per element, i.e. in the same order as
// Julia's sum(X, dims=3); partial moments of batch shards are plain sums, so multi-GPU needs one
// all-reduce(sum) of [sum | sumsq].
#include "wx_common.h"
#include "wx_kernels.h"

// sum[e] (+)= sum_b X[e, b];  sumsq[e] (+)= sum_b X[e, b]^2      (e in [0, nk), b in chunk)
// x*x rounded on its own, as the reference's X.^2 (bestbasis_tree.jl:154): folding the square into the
// running sum (one rounding) would give a batch of identical signals a variance of +-1 ulp instead of the exact
// 0 (sigma = 0, cost -Inf) of the reference.  This file is compiled with -ffp-contract=on (see the Makefile):
// products are only fused inside one expression or through an explicit fma(), never across statements
// (-ffp-contract=fast fuses in the backend whatever the source says).
template <typename T> static __device__ __forceinline__ T wx_sq_unfused(T v)
{
    const T sq = v * v;
    return sq;
}

template <typename T>
__global__ __launch_bounds__(256) void k_jbb_moments(const T *__restrict__ X, T *__restrict__ sum,
                                                     T *__restrict__ sumsq, int64_t nk, int64_t batch,
                                                     int accumulate, int64_t chunk, int64_t out_stride)
{
    // blockIdx.y selects a chunk of the signal axis; chunk c writes its partial at c*out_stride
    const int64_t c = blockIdx.y;
    const int64_t b0 = c * chunk;
    int64_t b1 = b0 + chunk; if (b1 > batch) b1 = batch;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nk; e += (int64_t)gridDim.x * blockDim.x) {
        T s = 0, q = 0;
        if (accumulate && c == 0) { s = sum[e]; q = sumsq[e]; }
        const T *p = X + e;
        // eight signals' loads in flight, summed in the same order as one by one
        int64_t b = b0;
        for (; b + 8 <= b1; b += 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(b + u) * nk];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s = (T)(s + v[u]);
                const T vq = wx_sq_unfused<T>(v[u]);
                q = (T)(q + vq);
            }
        }
        for (; b < b1; ++b) {
            const T v = p[b * nk];
            s = (T)(s + v);
            const T vq = wx_sq_unfused<T>(v);
            q = (T)(q + vq);
        }
        sum[c * out_stride + e] = s;
        sumsq[c * out_stride + e] = q;
    }
}

// sequential combine of chunk partials into chunk 0 (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void k_jbb_combine(T *__restrict__ sum, T *__restrict__ sumsq, int64_t nk,
                                                     int nchunks, int64_t stride)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nk; e += (int64_t)gridDim.x * blockDim.x) {
        T s = sum[e], q = sumsq[e];
        for (int c = 1; c < nchunks; ++c) { s = (T)(s + sum[c * stride + e]); q = (T)(q + sumsq[c * stride + e]); }
        sum[e] = s;
        sumsq[e] = q;
    }
}

// one block per cost entry; block-wide tree reduction
// E[x^2] - E[x]^2 exactly as the reference rounds it (separate multiply and subtract): with a fused
// multiply-add a batch of identical signals would give a tiny negative variance (NaN sigma, the reference's
// @assert all(sigma .>= 0)) instead of the exact 0 the reference gets.
template <typename T> static __device__ __forceinline__ T wx_var_unfused(T ex2, T ex)
{
    const T sq = ex * ex;                 // separate statements: not contracted under -ffp-contract=on
    return ex2 - sq;
}

template <typename T>
__global__ __launch_bounds__(256) void k_jbb_costs(const T *__restrict__ sum, const T *__restrict__ sumsq,
                                                   int64_t Ntot, int n, int k, int redundant, int cost_kind,
                                                   double p, T *__restrict__ costs)
{
    __shared__ double red[256];
    const int idx = blockIdx.x;                   // 0-based cost index
    int col, off, len, depth;
    if (redundant) {
        col = idx; off = 0; len = n;
        depth = 0; for (int t = idx + 1; t > 1; t >>= 1) ++depth;
    } else {
        depth = 0; for (int t = idx + 1; t > 1; t >>= 1) ++depth;   // heap order == (lvl, node) order
        const int node = idx + 1 - (1 << depth);
        col = depth; len = n >> depth; off = node * len;
    }
    double acc = 0.0;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const int64_t e = (int64_t)col * n + off + i;
        const T ex = (T)(sum[e] / (T)Ntot), ex2 = (T)(sumsq[e] / (T)Ntot);
        const T var = wx_var_unfused<T>(ex2, ex);
        const T sg = (T)sqrt((double)var);                           // NaN if cancellation made var < 0
        if (cost_kind == 0) acc += (double)(T)log((double)(T)fabs((double)sg));
        else acc += pow(fabs((double)sg), p);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double c = cost_kind == 0 ? p * red[0] : red[0];
        if (redundant) c /= (double)((int64_t)1 << depth);
        costs[idx] = (T)c;
    }
}

// tree_costs(X::Array{T,4}, ::JBB) bestbasis/bestbasis_tree.jl:182-207: one block per quad node.
// redundant: cost[i] = coefcost(sigma[:,:,i]) / 4^depth(i); otherwise the node's (rows, cols) block of
// slice depth+1 (Utils.jl:465-542 geometry through the morton code of the heap index)
template <typename T>
__global__ __launch_bounds__(256) void k_jbb_costs2d(const T *__restrict__ sum, const T *__restrict__ sumsq,
                                                     int64_t Ntot, int m, int n, int k, int redundant, int cost_kind,
                                                     double p, T *__restrict__ costs)
{
    __shared__ double red[256];
    const int64_t idx = blockIdx.x;               // 0-based heap index
    int depth = 0;
    { int64_t t = 3 * (idx + 1) - 2; while (t >= 4) { t >>= 2; ++depth; } }
    int64_t start = 1;
    for (int t = 0; t < depth; ++t) start = 4 * start - 2;
    const int64_t mort = idx + 1 - start;
    int jr = 0, jc = 0;
    for (int t = 0; t < depth; ++t) { jr |= (int)((mort >> (2 * t + 1)) & 1) << t; jc |= (int)((mort >> (2 * t)) & 1) << t; }
    int r0, c0, nr, ncl;
    int64_t slice;
    if (redundant) { slice = idx; r0 = 0; c0 = 0; nr = m; ncl = n; }
    else { slice = depth; nr = m >> depth; ncl = n >> depth; r0 = jr * nr; c0 = jc * ncl; }
    double acc = 0.0;
    const int cnt = nr * ncl;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int r = r0 + i % nr, c = c0 + i / nr;
        const int64_t e = slice * (int64_t)m * n + (int64_t)c * m + r;
        const T ex = (T)(sum[e] / (T)Ntot), ex2 = (T)(sumsq[e] / (T)Ntot);
        const T sg = (T)sqrt((double)wx_var_unfused<T>(ex2, ex));
        if (cost_kind == 0) acc += (double)(T)log((double)(T)fabs((double)sg));
        else acc += pow(fabs((double)sg), p);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double c = cost_kind == 0 ? p * red[0] : red[0];
        if (redundant) c /= (double)((int64_t)1 << (2 * depth));
        costs[idx] = (T)c;
    }
}

template <typename T>
int wx_dev_jbb_costs2d(const T *sum, const T *sumsq, int64_t Ntot, int64_t m, int64_t n, int64_t k, int redundant,
                       int cost_kind, double p, T *costs, hipStream_t st)
{
    const int64_t ncost = redundant ? k : ((((int64_t)1 << (2 * k)) - 1) / 3);
    if (ncost == 0) return WX_OK;
    hipLaunchKernelGGL(k_jbb_costs2d<T>, dim3((unsigned)ncost), dim3(256), 0, st, sum, sumsq, Ntot, (int)m, (int)n, (int)k,
                       redundant, cost_kind, p, costs);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template int wx_dev_jbb_costs2d<double>(const double *, const double *, int64_t, int64_t, int64_t, int64_t, int, int, double, double *, hipStream_t);
template int wx_dev_jbb_costs2d<float>(const float *, const float *, int64_t, int64_t, int64_t, int64_t, int, int, double, float *, hipStream_t);

// ------------------------------------------------------------------------------------------
// acwpd + JBB moments without the packet table (BASELINE config 5).
// Below depth D0 the dilated autocorrelation steps (stride 2^d >= 2^D0) never mix samples of
// different residue classes mod 2^D0, so the subtree under (node q of depth D0, class r) is an
// independent undecimated packet decomposition of a signal of n' = n / 2^D0 samples.  One workgroup
// owns one (q, r): it walks the signals in order, G at a time, keeps the G sub-signals' levels in
// LDS and accumulates sum / sum-of-squares of every (node, sample) of its subtree in registers --
// sequentially over the signals, i.e. in the order of Julia's sum(X, dims=3).  HBM sees only the
// shallow top table (depth <= D0) instead of the full (n, 2^(L+1)-1) table per signal.
// ------------------------------------------------------------------------------------------
struct WxFoldTap { double B; int off; int pad; };

constexpr int WX_AC_PLANE = 768;      // positions per signal-pair plane: the two largest levels, 512 + 256

template <int LP, int G, int NL>
__global__ __launch_bounds__(256) void k_acwpd_subtree_moments(const double *__restrict__ top, double *__restrict__ sum,
                                                               double *__restrict__ sumsq, int log2n, int D0,
                                                               int ncols_top, int64_t batch, WxAcFilt ac,
                                                               int accumulate)
{
    static_assert(G == 4, "the LDS layout holds two planes of signal pairs");
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    constexpr int NT = 256;
    const int n = 1 << log2n;
    const int lnp = log2n - D0;                   // log2 of the sub-signal length n'
    const int np = 1 << lnp;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, and the 2^D0 residue classes of one
    // node read the same cache lines of the top table (8 bytes out of every 2^D0 * 8), so consecutive logical
    // ids (same node) are mapped to the same XCD / L2 instead of being spread over all eight
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int q = bid >> D0;                      // node of depth D0
    const int r = bid & ((1 << D0) - 1);          // residue class
    const int tid = threadIdx.x;
    // LDS: plane 0 holds signals (0,1) of the group as one double2 per position, plane 1 signals (2,3) at the
    // fixed distance WX_AC_PLANE, so a thread reads the 4 signals of a position with one address (16-byte
    // reads, the second through the immediate offset) and consecutive items read consecutive 16 bytes:
    // conflict free.  Within a plane, even levels live at [0, szA), odd levels at [szA, szA + szB).
    int szA = np, szB = 0;
    for (int j = 1; j < LP; ++j) { const int c = np << j; if (j & 1) { if (c > szB) szB = c; } else if (c > szA) szA = c; }
    double2 *bufA = reinterpret_cast<double2 *>(wx_smem);
    double2 *bufB = bufA + szA;
    // Tap table per level as LDS broadcasts (the tap loop stays rolled: unrolling it lets the scheduler
    // hoist every window load and spill).  At level j the sub-signal splits into classes of M = n' >> j
    // samples and only odd lags are non-zero, so when M/2 <= 2 NL the +-lags alias onto the M/2 odd
    // residues mod M and the periodised filter has M/2 taps (sums of the b_l that alias) instead of 2 NL.
    WxFoldTap *ft = reinterpret_cast<WxFoldTap *>(bufA + 2 * WX_AC_PLANE);
    int *fT = reinterpret_cast<int *>(ft + LP * 2 * NL);
    if (tid < LP * 2 * NL) {
        const int j = tid / (2 * NL), k = tid - j * (2 * NL);
        const int M = np >> j;
        double B = 0.0;
        int rho = 0;
        if (M / 2 <= 2 * NL) {
            rho = 2 * k + 1;
            if (k < M / 2)
                for (int l = 0; l < NL; ++l) {
                    const int lag = (2 * l + 1) % M;
                    if (lag == rho) B += ac.b[2 * l];
                    if ((M - lag) % M == rho) B += ac.b[2 * l];
                }
            if (k == 0) fT[j] = M / 2;
        } else {
            if (k < NL) { rho = 2 * k + 1; B = ac.b[2 * k]; }
            else { rho = M - (2 * (k - NL) + 1); B = ac.b[2 * (k - NL)]; }
            if (k == 0) fT[j] = 2 * NL;
        }
        ft[tid].B = B;
        ft[tid].off = (rho << j) & (np - 1);
        ft[tid].pad = 0;
    }
    const double c1 = ac.c1;
    const int64_t colq = ((int64_t)1 << D0) - 1 + q;                // heap column (0-based) of the subtree root
    const int64_t sig_stride = (int64_t)n * ncols_top;

    // thread tid owns items tid and tid + 256 (the deepest level may have 512) of every level and sums its
    // signals in order, group after group: exactly the order of Julia's sum(X, dims=3)
    double acc[LP + 1][4];                                           // [LP] = second item of the deepest level
#pragma unroll
    for (int j = 0; j <= LP; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[j][k] = 0.0;

    // sub-signal fetch: element e = (g, i); missing signals of the last group are zeros, which add
    // exact zeros to every moment below
    const int fe_g = tid >> lnp, fe_i = tid & (np - 1);
    const bool fetcher = tid < G * np;                               // G * np <= 256 (see wx_acwpd_fused_depth)
    const double *fsrc = top + colq * n + r + ((int64_t)fe_i << D0);
    double *fdst = reinterpret_cast<double *>(bufA + (fe_g >> 1) * WX_AC_PLANE + fe_i) + (fe_g & 1);
    double pre = (fetcher && fe_g < batch) ? fsrc[fe_g * sig_stride] : 0.0;
    for (int64_t sig0 = 0; sig0 < batch; sig0 += G) {
        if (fetcher) *fdst = pre;
        __syncthreads();
        {   // next group's samples travel while this group is processed
            const int64_t gn = sig0 + G + fe_g;
            pre = (fetcher && gn < batch) ? fsrc[gn * sig_stride] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < LP; ++j) {
            const double2 *cur = (j & 1) ? bufB : bufA;
            double2 *nxt = (j & 1) ? bufA : bufB;
            const int cnt = np << j;                                  // items of this level
            const WxFoldTap *tp = ft + j * 2 * NL;
            const int T = __builtin_amdgcn_readfirstlane(fT[j]);
#pragma unroll
            for (int h = 0; h < (j == LP - 1 ? 2 : 1); ++h) {         // only the deepest level can have 512 items
                const int item = tid + h * NT;
                const int aj = j + h;
                if (item < cnt) {
                    const int p = item >> lnp, i = item & (np - 1);
                    const double2 *v = cur + ((size_t)p << lnp);
                    double S0 = 0.0, S1 = 0.0, S2 = 0.0, S3 = 0.0;
#pragma unroll 1
                    for (int l = 0; l < T; ++l) {
                        const double bt = tp[l].B;
                        const double2 *xv = v + ((i + tp[l].off) & (np - 1));
                        const double2 x01 = xv[0], x23 = xv[WX_AC_PLANE];
                        S0 = fma(bt, x01.x, S0); S1 = fma(bt, x01.y, S1);
                        S2 = fma(bt, x23.x, S2); S3 = fma(bt, x23.y, S3);
                    }
                    const double2 c01 = v[i], c23 = v[i + WX_AC_PLANE];
                    double a0 = acc[aj][0], a1 = acc[aj][1], a2 = acc[aj][2], a3 = acc[aj][3];
                    double lo[4], hi[4];
                    const double xc[4] = {c01.x, c01.y, c23.x, c23.y};
                    const double SS[4] = {S0, S1, S2, S3};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        lo[c] = fma(c1, xc[c], SS[c]); hi[c] = fma(c1, xc[c], -SS[c]);   // v/sqrt2 +- S (transform: tolerance)
                        const double lq = wx_sq_unfused<double>(lo[c]), hq = wx_sq_unfused<double>(hi[c]);
                        a0 += lo[c]; a1 += lq;
                        a2 += hi[c]; a3 += hq;
                    }
                    acc[aj][0] = a0; acc[aj][1] = a1; acc[aj][2] = a2; acc[aj][3] = a3;
                    if (j + 1 < LP) {
                        double2 *olo = nxt + (((size_t)(2 * p)) << lnp) + i;
                        double2 *ohi = olo + np;
                        olo[0] = make_double2(lo[0], lo[1]); olo[WX_AC_PLANE] = make_double2(lo[2], lo[3]);
                        ohi[0] = make_double2(hi[0], hi[1]); ohi[WX_AC_PLANE] = make_double2(hi[2], hi[3]);
                    }
                }
            }
            __syncthreads();
        }
    }
    // add to the heap columns of the subtree
    const int64_t H = ((int64_t)1 << D0) + q;                         // 1-based heap index of the subtree root
#pragma unroll
    for (int j = 0; j < LP; ++j) {
#pragma unroll
        for (int h = 0; h < (j == LP - 1 ? 2 : 1); ++h) {
            const int item = tid + h * NT;
            const int aj = j + h;
            if (item < (np << j)) {
                const int p = item >> lnp, i = item & (np - 1);
                const int64_t hl = (H << (j + 1)) + 2 * p;           // low child, 1-based
                const int64_t row = r + ((int64_t)i << D0);
                const int64_t el = (hl - 1) * n + row, eh = hl * n + row;
                if (accumulate) {
                    sum[el] += acc[aj][0]; sumsq[el] += acc[aj][1]; sum[eh] += acc[aj][2]; sumsq[eh] += acc[aj][3];
                } else {
                    sum[el] = acc[aj][0]; sumsq[el] = acc[aj][1]; sum[eh] = acc[aj][2]; sumsq[eh] = acc[aj][3];
                }
            }
        }
    }
}

// D0 for which the deepest subtree level has at most 512 (parent, sample) items; < 0 if the fused
// path does not apply
int wx_acwpd_fused_depth(int64_t n, int L, int F)
{
    const int NL = F / 2;
    if (NL < 1) return -1;
    if (n < 2 || (n & (n - 1))) return -1;
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    for (int D0 = 0; D0 < L; ++D0) {
        const int LP = L - D0;
        if (LP > 6 || D0 > log2n) continue;
        const int64_t cnt = (n >> D0) << (LP - 1);
        if (cnt <= 512 && (n >> D0) >= 1 && (n >> D0) * 4 <= 256) {   // 4 signals x n' fetched by 256 threads
            // the LDS kernel of this file is instantiated for F/2 in {1..6, 8, 9, 10}; the matrix-pipe kernel
            // (wx_acsubtree.hip) periodises any filter
            if (NL <= 10 && NL != 7) return D0;
            return wx_acwpd_mfma_ok(n, L, D0) ? D0 : -1;
        }
    }
    return -1;
}

int wx_dev_acwpd_subtree_moments(const double *top, double *sum, double *sumsq, int64_t n, int L, int D0,
                                 int64_t batch, const WxAcFilt &ac, int accumulate, hipStream_t st)
{
    constexpr int G = 4;
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    const int LP = L - D0;
    const int np = (int)(n >> D0);
    int szA = np, szB = 0;
    for (int j = 1; j < LP; ++j) { const int c = np << j; if (j & 1) { if (c > szB) szB = c; } else if (c > szA) szA = c; }
    size_t lds = ((size_t)2 * 768 * 2 + 2 * 6 * 2 * 10 + 8) * sizeof(double);   // two signal-pair planes + tap table
    const int ncols_top = (1 << (D0 + 1)) - 1;
    const unsigned grid = 1u << (2 * D0);
    typedef void (*kern_t)(const double *, double *, double *, int, int, int, int64_t, WxAcFilt, int);
    kern_t kern = nullptr;
    const int NL = ac.F / 2;
#define WX_LP(NLV) \
    switch (LP) { case 1: kern = k_acwpd_subtree_moments<1, G, NLV>; break; case 2: kern = k_acwpd_subtree_moments<2, G, NLV>; break; \
                  case 3: kern = k_acwpd_subtree_moments<3, G, NLV>; break; case 4: kern = k_acwpd_subtree_moments<4, G, NLV>; break; \
                  case 5: kern = k_acwpd_subtree_moments<5, G, NLV>; break; case 6: kern = k_acwpd_subtree_moments<6, G, NLV>; break; }
    switch (NL) {
    case 1: WX_LP(1) break; case 2: WX_LP(2) break; case 3: WX_LP(3) break; case 4: WX_LP(4) break; case 5: WX_LP(5) break;
    case 6: WX_LP(6) break; case 8: WX_LP(8) break; case 9: WX_LP(9) break; case 10: WX_LP(10) break;
    }
#undef WX_LP
    if (!kern) return wx_set_error(WX_EUNSUPPORTED, "fused acwpd moments: unsupported filter length / subtree depth");
    if (lds > 64 * 1024)
        WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, top, sum, sumsq, log2n, D0, ncols_top, batch, ac, accumulate);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

template <typename T>
int wx_dev_jbb_moments(const T *X, T *sum, T *sumsq, int64_t nk, int64_t batch, int accumulate, T *scratch,
                       int nchunks, hipStream_t st)
{
    if (nk == 0) return WX_OK;
    int64_t gx = (nk + 255) / 256;
    if (gx > 4096) gx = 4096;
    if (nchunks <= 1 || !scratch) {
        hipLaunchKernelGGL(k_jbb_moments<T>, dim3((unsigned)gx, 1), dim3(256), 0, st, X, sum, sumsq, nk, batch,
                           accumulate, batch > 0 ? batch : 1, (int64_t)0);
    } else {
        // chunk 0 lands in (sum, sumsq) directly; chunks 1.. in scratch laid out [2][nchunks-1][nk]
        // -> express as one strided layout: partial c of `sum` at psum + c*nk
        T *psum = scratch, *psq = scratch + (int64_t)nchunks * nk;
        const int64_t chunk = (batch + nchunks - 1) / nchunks;
        if (accumulate) {
            WX_HIP_CHECK(hipMemcpyAsync(psum, sum, sizeof(T) * nk, hipMemcpyDeviceToDevice, st));
            WX_HIP_CHECK(hipMemcpyAsync(psq, sumsq, sizeof(T) * nk, hipMemcpyDeviceToDevice, st));
        }
        hipLaunchKernelGGL(k_jbb_moments<T>, dim3((unsigned)gx, (unsigned)nchunks), dim3(256), 0, st, X, psum, psq,
                           nk, batch, accumulate, chunk, nk);
        hipLaunchKernelGGL(k_jbb_combine<T>, dim3((unsigned)gx), dim3(256), 0, st, psum, psq, nk, nchunks, nk);
        WX_HIP_CHECK(hipMemcpyAsync(sum, psum, sizeof(T) * nk, hipMemcpyDeviceToDevice, st));
        WX_HIP_CHECK(hipMemcpyAsync(sumsq, psq, sizeof(T) * nk, hipMemcpyDeviceToDevice, st));
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

template <typename T>
int wx_dev_jbb_costs(const T *sum, const T *sumsq, int64_t Ntot, int64_t n, int64_t k, int redundant,
                     int cost_kind, double p, T *costs, hipStream_t st)
{
    const int64_t ncost = redundant ? k : (((int64_t)1 << k) - 1);
    if (ncost == 0) return WX_OK;
    hipLaunchKernelGGL(k_jbb_costs<T>, dim3((unsigned)ncost), dim3(256), 0, st, sum, sumsq, Ntot, (int)n, (int)k,
                       redundant, cost_kind, p, costs);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

#define WX_INST(T)                                                                                        \
    template int wx_dev_jbb_moments<T>(const T *, T *, T *, int64_t, int64_t, int, T *, int, hipStream_t); \
    template int wx_dev_jbb_costs<T>(const T *, const T *, int64_t, int64_t, int64_t, int, int, double, T *, hipStream_t);
WX_INST(double)
WX_INST(float)
