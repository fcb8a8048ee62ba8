// wx_acsubtree.hip -- acwpd + JBB moments below depth D0 (BASELINE config 5), one wavefront per subtree.
//
// Reference (paths relative to /root/reference/src/mod):
//   acwpd(x, wt, L)                    ACWT.jl:733-759   (autocorrelation packet table, heap order, undecimated)
//   one level                          acwt/acwt_one_level.jl: lo = x/sqrt2 + S, hi = x/sqrt2 - S, S = sum over the odd lags
//   tree_costs(X::Array{T,3}, ::JBB)   bestbasis/bestbasis_tree.jl:150-180   (EX = sum(X, dims=3)/N, EX2 = sum(X.^2, dims=3)/N)
//
// The work: below depth D0 the dilated steps never mix residue classes mod 2^D0, so the subtree under (node q of depth
// D0, class r) is an independent undecimated packet decomposition of the n' = n / 2^D0 = 32 samples x0[i] = top[q][r + 2^D0 i]
// with 5 levels: 2 + 4 + 8 + 16 + 32 nodes x 32 samples = 1984 coefficients per signal, each needing sum and sum of
// squares over the signals IN ORDER (the order of Julia's sum(X, dims=3)).  k_acwpd_subtree_moments (wx_jbb.hip) gives a
// 256-thread workgroup to every (q, r) and keeps every level in LDS: 5.8e8 16-byte LDS wave-instructions per launch,
// bound by LDS issue (10 ms per 2048 signals).  Here ONE WAVEFRONT owns a (q, r):
//   * the 62 accumulators of a lane (31 coefficients x {sum, sumsq}) stay in registers for the whole batch;
//   * levels 0..2 (16, 8, 4 periodised taps) are matrix products on the FP64 matrix pipe: a data tile is 16 rows
//     (samples of one parity) x 16 columns (8 signals x 2 parities), held as the C/D operand layout of
//     v_mfma_f64_16x16x4_f64 (row = (lane >> 4) + 4 reg, column = lane & 15); the periodised filter of a level is a
//     16 x 16 circulant A with A[a][b] = tap((b - a) mod 16) (zero where the level's dilation skips), and because a
//     tile's register t of lane (g, col) is exactly the B operand element k = 4 t + g, the output tile of one level
//     feeds the next level's products with no data movement at all -- the tap loop, its LDS reads and the cross-lane
//     traffic are gone.  Level 0 couples the two parities: its input tile is the sequence advanced by one sample;
//   * levels 3 and 4 (2 and 1 periodised taps) never leave a lane: lane (p3, c3) owns the four samples i = c3 + 8 m of the
//     depth-3 node p3, and the dilations 8 and 16 only move m;
//   * the only LDS traffic is the transposition from "signal = column" to "coefficient = lane": every coefficient of
//     levels 1..3 is written once and read once (448 x 8 bytes each way per signal instead of ~36 KB), 16 KiB per wave,
//     16-byte reads of signal pairs, XOR-swizzled so that reads and writes are bank-conflict free; no barrier anywhere
//     (the window is private to the wavefront and LDS operations of a wavefront complete in order).
// Sums run over the signals in order (8 per block, blocks in order); x*x is rounded on its own (wx_sq_unfused: this file
// is built with -ffp-contract=on), so a batch of identical signals keeps the reference's exact sigma = 0.
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"

typedef double wx_d4 __attribute__((ext_vector_type(4)));
// a wave-uniform pointer kept in scalar registers: loads take the "scalar base + 32-bit lane byte offset" form
typedef const char __attribute__((address_space(1))) *acs_gc;
static __device__ __forceinline__ acs_gc acs_sbase(const double *p)
{
    acs_gc g = (acs_gc)p;
    asm("" : "+s"(g));
    return g;
}
static __device__ __forceinline__ double acs_ld(acs_gc base, unsigned boff)
{
    return *(const double __attribute__((address_space(1))) *)(base + boff);
}

static __device__ __forceinline__ double acs_sq(double v)
{
    const double sq = v * v;
    return sq;
}

// folded (periodised) taps of the five subtree levels: [0,16) level 0, [16,24) level 1, [24,28) level 2, [28,30) level 3,
// [30] level 4, [31] = c1 = 1/sqrt2
constexpr int ACS_TAB = 32;

// LDS window: 256 slots of 8 doubles (the 8 signals of one coefficient); the four 16-byte chunks (signal pairs) of a slot
// sit at chunk position (pair ^ (slot >> 2)) & 3, which makes the 64-lane writes (8 signals x 2 slots per 16 lanes) and
// the 16-byte reads (one slot per lane) bank-conflict free.  Every slot used below is (compile-time constant) + (a lane
// term < 8 for writes, the lane itself for reads), so the swizzle is a lane term XOR a constant: writes need two base
// registers (constant's bit 1 clear / set), reads four (one per pair), everything else is an immediate offset.
static __device__ __forceinline__ int acs_wbase(int w, int s, int flip)
{
    return w * 8 + (((((s >> 1) ^ (w >> 2)) & 3) ^ (flip ? 2 : 0)) << 1) + (s & 1);
}
static __device__ __forceinline__ int acs_rbase(int lane, int c)
{
    return lane * 8 + (((c ^ (lane >> 2)) & 3) << 1);
}

// SUMS = false (round 5): only the sums of squares.  The first moments of EVERY node are the transform of the sum signal
// (sum_b acwpd(x_b) = acwpd(sum_b x_b): api_acwpd_jbb_moments runs one extra signal through the plain acwpd kernels), which removes one
// of the three vector instructions per coefficient and signal -- 248 of the ~1120 this kernel issues per 8-signal block; it is bound
// by FP64 issue (profiles/r03_cfg5.md), so the time follows.
template <int WPE, bool SUMS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_acwpd_subtree_mfma(const double *__restrict__ top, double *__restrict__ sum, double *__restrict__ sumsq,
                          const double *__restrict__ tab, int D0, int ncols_top, int64_t batch, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) double acs_lds[];          // 256 slots x 8 signals
    const int lane = threadIdx.x;
    int bid = blockIdx.x;
    // XCD-aware order: the 2^D0 classes of a node read the same lines of the top table -> same XCD / L2
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int q = bid >> D0, r = bid & ((1 << D0) - 1);
    const int64_t n = (int64_t)32 << D0;
    const int s = lane & 7, par = (lane >> 3) & 1, g = lane >> 4;

    // A operands: lane (a = lane & 15, g), step t -> A[a][b = g + 4 t] = tap table of the level at d = (b - a) & 15.  The
    // three 16-entry tables (zeros where the level's dilation skips) sit behind the window in LDS and are re-read at
    // every use: 24 registers fewer to carry through the register-bound lane-local phase.
    double *const atab = acs_lds + 256 * 8;
    if (lane < 48) {
        const int j = lane >> 4, d = lane & 15;
        atab[lane] = j == 0 ? tab[d] : j == 1 ? ((d & 1) ? tab[16 + (d >> 1)] : 0.0) : (((d & 3) == 2) ? tab[24 + (d >> 2)] : 0.0);
    }
    const int d0 = (g - (lane & 15)) & 15;
#define ACS_A(j, t) atab[16 * (j) + ((d0 + 4 * (t)) & 15)]
    const double B30 = tab[28], B31 = tab[29], B40 = tab[30], c1 = tab[31];

    // owner side: lane = (p3, c3)
    const int p3 = lane >> 3, c3 = lane & 7;
    double a1s = 0, a1q = 0, a2s[2] = {0, 0}, a2q[2] = {0, 0}, a3s[4], a3q[4], a4s[2][4], a4q[2][4], a5s[4][4], a5q[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        a3s[m] = a3q[m] = 0;
#pragma unroll
        for (int b = 0; b < 2; ++b) a4s[b][m] = a4q[b][m] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) a5s[b][m] = a5q[b][m] = 0;
    }

    // loads: plain tile register t <-> sample i = 2 (g + 4 t) + par; advanced tile <-> (i + 1) & 31; a block's eight
    // signals are addressed as (uniform base of the block) + (32-bit lane offset)
    const int64_t colq = ((int64_t)1 << D0) - 1 + q;
    const int64_t sig_stride = n * ncols_top;
    const double *src = top + colq * n + r;
    unsigned offP[4], offS[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = 2 * (g + 4 * t) + par;
        offP[t] = 8u * ((unsigned)(s * sig_stride) + (unsigned)(i << D0));               // bytes; < 2^32: 8 signals of the top table
        offS[t] = 8u * ((unsigned)(s * sig_stride) + (unsigned)(((i + 1) & 31) << D0));
    }
    double P[4], S[4];
    {
        const bool ok = s < batch;
        const acs_gc p = acs_sbase(src);
#pragma unroll
        for (int t = 0; t < 4; ++t) { P[t] = ok ? acs_ld(p, offP[t]) : 0.0; S[t] = ok ? acs_ld(p, offS[t]) : 0.0; }
    }
    const int w = 2 * g + par;
    const int wb0 = acs_wbase(w, s, 0), wb1 = acs_wbase(w, s, 1);
    int rb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) rb[c] = acs_rbase(lane, c);
#define ACS_W(slotc) acs_lds[((((slotc) >> 2) & 2) ? wb1 : wb0) + (slotc) * 8]      /* slot = slotc + w, slotc multiple of 8 */
#define ACS_R(slotc, c) (*reinterpret_cast<const double2 *>(&acs_lds[rb[c] + (slotc) * 8]))   /* slot = slotc + lane */

    // LDS slots of the coefficients this lane WRITES (tile row a = g + 4 t, i = 2 a + par), per level
    //   level 1 (node p1): owner lane = (4 p1 + (i >> 3)) * 8 + (i & 7), slot = owner
    //   level 2 (node p2): owner = (2 p2 + (i >> 4)) * 8 + (i & 7), e = (i >> 3) & 1, slot = 64 + 64 e + owner
    //   level 3 (node p3): owner = 8 p3 + (i & 7), m = i >> 3, slot = 64 m + owner
    for (int64_t sig0 = 0; sig0 < batch; sig0 += 8) {
        const wx_d4 X0 = {P[0], P[1], P[2], P[3]}, XS = {S[0], S[1], S[2], S[3]};
        // ---- level 0 -> the two depth-1 nodes
        wx_d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ACS_A(0, t), XS[t], acc, 0, 0, 0);
        wx_d4 L1[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) { L1[0][t] = fma(c1, X0[t], acc[t]); L1[1][t] = fma(c1, X0[t], -acc[t]); }
#pragma unroll
        for (int p1 = 0; p1 < 2; ++p1)
#pragma unroll
            for (int t = 0; t < 4; ++t) ACS_W((4 * p1 + t) * 8) = L1[p1][t];               // i >> 3 = t, i & 7 = w
        // ---- level 1 -> the four depth-2 nodes
        wx_d4 L2[4];
#pragma unroll
        for (int p1 = 0; p1 < 2; ++p1) {
            wx_d4 a = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) a = __builtin_amdgcn_mfma_f64_16x16x4f64(ACS_A(1, t), L1[p1][t], a, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) { L2[2 * p1][t] = fma(c1, L1[p1][t], a[t]); L2[2 * p1 + 1][t] = fma(c1, L1[p1][t], -a[t]); }
        }
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2)
#pragma unroll
            for (int t = 0; t < 4; ++t) ACS_W(64 + 64 * (t & 1) + (2 * p2 + (t >> 1)) * 8) = L2[p2][t];
        // ---- level 2 -> the eight depth-3 nodes: four independent product chains, interleaved; while the matrix pipe works
        // the owners of the depth-1 and depth-2 coefficients add the 8 signals in order
        wx_d4 a3[2];
        auto level2_pair = [&](int pp) {
            a3[0] = a3[1] = wx_d4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double At = ACS_A(2, t);
#pragma unroll
                for (int k = 0; k < 2; ++k) a3[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(At, L2[2 * pp + k][t], a3[k], 0, 0, 0);
            }
        };
        level2_pair(0);
        {
            double2 v1[4], v2[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v1[c] = ACS_R(0, c);
#pragma unroll
                for (int e = 0; e < 2; ++e) v2[e][c] = ACS_R(64 + 64 * e, c);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (SUMS) a1s += v1[c].x;
                a1q += acs_sq(v1[c].x);
#pragma unroll
                for (int e = 0; e < 2; ++e) { if (SUMS) a2s[e] += v2[e][c].x; a2q[e] += acs_sq(v2[e][c].x); }
                if (SUMS) a1s += v1[c].y;
                a1q += acs_sq(v1[c].y);
#pragma unroll
                for (int e = 0; e < 2; ++e) { if (SUMS) a2s[e] += v2[e][c].y; a2q[e] += acs_sq(v2[e][c].y); }
            }
        }
        // the depth-3 slots overlay the two regions just read: same wavefront, LDS operations complete in order
        auto level3_write = [&](int pp) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int p2 = 2 * pp + k;
                    ACS_W(64 * t + (2 * p2) * 8) = fma(c1, L2[p2][t], a3[k][t]);
                    ACS_W(64 * t + (2 * p2 + 1) * 8) = fma(c1, L2[p2][t], -a3[k][t]);
                }
        };
        level3_write(0);
        level2_pair(1);
        level3_write(1);
        // next block's samples travel during the lane-local phase (most of the block's time)
        if (sig0 + 16 <= batch) {
            const acs_gc p = acs_sbase(src + (sig0 + 8) * sig_stride);
#pragma unroll
            for (int t = 0; t < 4; ++t) { P[t] = acs_ld(p, offP[t]); S[t] = acs_ld(p, offS[t]); }
        } else {
            const bool ok = sig0 + 8 + s < batch;
            const acs_gc p = acs_sbase(src + (sig0 + 8) * sig_stride);
#pragma unroll
            for (int t = 0; t < 4; ++t) { P[t] = ok ? acs_ld(p, offP[t]) : 0.0; S[t] = ok ? acs_ld(p, offS[t]) : 0.0; }
        }
        // ---- levels 3 and 4 inside the lane: x[m] = depth-3 node p3 at i = c3 + 8 m
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double2 xv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) xv[m] = ACS_R(64 * m, c);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double x[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) x[m] = h ? xv[m].y : xv[m].x;
                double y[2][4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (SUMS) a3s[m] += x[m];
                    a3q[m] += acs_sq(x[m]);
                    const double S3 = fma(B31, x[(m + 3) & 3], B30 * x[(m + 1) & 3]);
                    y[0][m] = fma(c1, x[m], S3); y[1][m] = fma(c1, x[m], -S3);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (SUMS) a4s[b][m] += y[b][m];
                        a4q[b][m] += acs_sq(y[b][m]);
                        const double S4 = B40 * y[b][(m + 2) & 3];
                        const double lo = fma(c1, y[b][m], S4), hi = fma(c1, y[b][m], -S4);
                        if (SUMS) { a5s[2 * b][m] += lo; a5s[2 * b + 1][m] += hi; }
                        a5q[2 * b][m] += acs_sq(lo);
                        a5q[2 * b + 1][m] += acs_sq(hi);
                    }
            }
        }
    }

    // ---- add to the heap columns of the subtree: node pk of relative level k is heap (1-based) (H << k) + pk
    const int64_t H = ((int64_t)1 << D0) + q;
    auto put = [&](int k, int pk, int i, double vs, double vq) {
        const int64_t e = (((H << k) + pk) - 1) * n + r + ((int64_t)i << D0);
        if (accumulate) { if (SUMS) sum[e] += vs; sumsq[e] += vq; }
        else { if (SUMS) sum[e] = vs; sumsq[e] = vq; }
    };
    put(1, p3 >> 2, c3 + 8 * (p3 & 3), a1s, a1q);
#pragma unroll
    for (int e = 0; e < 2; ++e) put(2, p3 >> 1, c3 + 8 * (2 * (p3 & 1) + e), a2s[e], a2q[e]);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        put(3, p3, c3 + 8 * m, a3s[m], a3q[m]);
#pragma unroll
        for (int b = 0; b < 2; ++b) put(4, 2 * p3 + b, c3 + 8 * m, a4s[b][m], a4q[b][m]);
#pragma unroll
        for (int b = 0; b < 4; ++b) put(5, 4 * p3 + b, c3 + 8 * m, a5s[b][m], a5q[b][m]);
    }
}

// the MFMA subtree kernel takes the full-depth case: n' = n / 2^D0 = 32 samples, 5 levels below D0
bool wx_acwpd_mfma_ok(int64_t n, int L, int D0)
{
    static const bool off = wx_getenv("WX_ACWPD_MFMA") && atoi(wx_getenv("WX_ACWPD_MFMA")) == 0;
    if (off || D0 < 0 || D0 > 12 || (n >> D0) != 32 || L - D0 != 5) return false;
    // a block's eight signals are addressed with 32-bit byte offsets from a uniform base
    const int64_t sig_stride = n * ((((int64_t)1) << (D0 + 1)) - 1);
    return 8 * 8 * sig_stride < ((int64_t)1 << 32);
}

// ---- the linear path of the first moments ------------------------------------------------------------------------------------
// part[g][i] = sum of x[i, b] over the signals b of group g, in order; then out[i] = sum over g in order.  (One thread per coefficient
// walking all signals would be 2048 dependent loads: the groups are the parallelism; the association is fixed, so the result is
// deterministic, and differs from one sequential sum by rounding only: bestbasis_tree.jl:153.)
__global__ __launch_bounds__(256) void k_sum_signals_part(const double *__restrict__ x, int n, int64_t batch, int groups, double *__restrict__ part)
{
    const int i = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (i >= n) return;
    const int64_t per = (batch + groups - 1) / groups;
    const int64_t b0 = g * per, b1 = b0 + per < batch ? b0 + per : batch;
    double a = 0.0;
    int64_t b = b0;
    for (; b + 4 <= b1; b += 4) {
        const double v0 = x[b * n + i], v1 = x[(b + 1) * n + i], v2 = x[(b + 2) * n + i], v3 = x[(b + 3) * n + i];
        a = __dadd_rn(__dadd_rn(__dadd_rn(__dadd_rn(a, v0), v1), v2), v3);
    }
    for (; b < b1; ++b) a = __dadd_rn(a, x[b * n + i]);
    part[(int64_t)g * n + i] = a;
}
__global__ __launch_bounds__(256) void k_sum_signals_comb(const double *__restrict__ part, int n, int groups, double *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    for (int g = 0; g < groups; ++g) a = __dadd_rn(a, part[(int64_t)g * n + i]);
    out[i] = a;
}
__global__ __launch_bounds__(256) void k_add_to(double *__restrict__ dst, const double *__restrict__ src, int64_t count)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (int64_t)gridDim.x * 256) dst[e] = __dadd_rn(dst[e], src[e]);
}
int wx_dev_sum_signals(const double *x, int64_t n, int64_t batch, double *out, double *part, int groups, hipStream_t st)
{
    if (n >= ((int64_t)1 << 31)) return wx_set_error(WX_EUNSUPPORTED, "sum of signals: n >= 2^31");
    const unsigned gx = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_sum_signals_part, dim3(gx, (unsigned)groups), dim3(256), 0, st, x, (int)n, batch, groups, part);
    hipLaunchKernelGGL(k_sum_signals_comb, dim3(gx), dim3(256), 0, st, (const double *)part, (int)n, groups, out);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
int wx_dev_add_to(double *dst, const double *src, int64_t count, hipStream_t st)
{
    int64_t g = (count + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(k_add_to, dim3((unsigned)(g < 1 ? 1 : g)), dim3(256), 0, st, dst, src, count);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

int wx_dev_acwpd_subtree_mfma(const double *top, double *sum, double *sumsq, int64_t n, int L, int D0, int64_t batch,
                              const WxAcFilt &ac, int accumulate, hipStream_t st, bool sums)
{
    if (!wx_acwpd_mfma_ok(n, L, D0)) return wx_set_error(WX_EUNSUPPORTED, "acwpd subtree (matrix pipe): n / 2^D0 must be 32 with 5 levels");
    // periodised taps: at subtree level j the sub-signal splits into classes of M = 32 >> j samples and only odd lags
    // are non-zero, so the +-lags alias onto the M/2 odd residues mod M (wx_jbb.hip folds the same way when M/2 <= F)
    double tabh[ACS_TAB];
    const int NL = ac.F / 2;
    int pos = 0;
    for (int j = 0; j < 5; ++j) {
        const int M = 32 >> j;
        for (int k = 0; k < M / 2; ++k) {
            const int rho = 2 * k + 1;
            double B = 0.0;
            for (int l = 0; l < NL; ++l) {
                const int lag = (2 * l + 1) % M;
                if (lag == rho) B += ac.b[2 * l];
                if ((M - lag) % M == rho) B += ac.b[2 * l];
            }
            tabh[pos++] = B;
        }
    }
    tabh[31] = ac.c1;
    const double *tab = (const double *)wx_const_upload(tabh, sizeof tabh, st, true);
    if (!tab) return WX_EHIP;
    static const int wpe = wx_getenv("WX_ACWPD_MFMA_WPE") ? atoi(wx_getenv("WX_ACWPD_MFMA_WPE")) : 2;
    const unsigned grid = 1u << (2 * D0);
    const int ncols_top = (1 << (D0 + 1)) - 1;
    const size_t lds = (256 * 8 + 48) * sizeof(double);
    if (wpe == 1 && sums)
        hipLaunchKernelGGL((k_acwpd_subtree_mfma<1, true>), dim3(grid), dim3(64), lds, st, top, sum, sumsq, tab, D0, ncols_top, batch, accumulate);
    else if (wpe == 1)
        hipLaunchKernelGGL((k_acwpd_subtree_mfma<1, false>), dim3(grid), dim3(64), lds, st, top, sum, sumsq, tab, D0, ncols_top, batch, accumulate);
    else if (sums)
        hipLaunchKernelGGL((k_acwpd_subtree_mfma<2, true>), dim3(grid), dim3(64), lds, st, top, sum, sumsq, tab, D0, ncols_top, batch, accumulate);
    else
        hipLaunchKernelGGL((k_acwpd_subtree_mfma<2, false>), dim3(grid), dim3(64), lds, st, top, sum, sumsq, tab, D0, ncols_top, batch, accumulate);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
