// wx_swt1d.hip -- batched 1-D redundant (undecimated) transforms for gfx950: the stationary
// family (sdwt / swpt / swpd and inverses) and the autocorrelation family (acdwt / acwpt / acwpd
// and inverses).
//
// Reference semantics (paths relative to /root/reference/src/mod), 0-based, s = 2^d:
//   sdwt_step!   swt/swt_one_level.jl:99-127   a[i] = sum_j q[j] v[(i+(j-1)s) mod n]
//                                              d[i] = sum_j (-1)^j q[j] v[(i-j s) mod n]
//   isdwt_step!  swt/swt_one_level.jl:279-318  shift (sv, sw): the parent samples of residue class
//                sv (mod s) are an ordinary idwt_step of the child samples of class sw (mod 2s);
//                stored one parent-sample earlier when sw == sv.   :257-277 average: both child
//                classes sw = sv and sw = sv + s for every sv, then / 2.
//   acdwt_step!  acwt/acwt_one_level.jl:101-128  w1 = v/sqrt2 + S, w2 = v/sqrt2 - S,
//                S[k] = sum_{l odd} a_l/(2 sqrt2) (v[k-ls] + v[k+ls])   (acwt_utils.jl:7-72; the
//                even-lag autocorrelations vanish by QMF orthogonality)
//   iacdwt_step! acwt/acwt_one_level.jl:217-224  v = (w1 + w2)/sqrt2
// Containers: SWT.jl:109-130 (sdwt (n,L+1) = [s_L d_L .. d_1]), :439-472 (swpt (n,2^L), children
// overwrite the parent column), :840-868 (swpd (n,2^(L+1)-1), heap order); ACWT.jl:109-129,
// 427-460, 733-759 use the same three layouts.
//
// Forward: one workgroup owns one (signal, parent node): the parent column is staged in LDS
// (which also resolves the reference's parent/child column aliasing), both children are written
// coalesced.  For a fixed tap all lanes read consecutive LDS words, so the dilation stride never
// causes bank conflicts.
#include "wx_common.h"
#include "wx_kernels.h"
#include "wx_host.h"
#include <cstdlib>
#include <map>
#include <vector>

int wx_force_generic();
static int wx_force_generic_swt() { return wx_force_generic(); }

enum { WX_LAYOUT_DWT = 0, WX_LAYOUT_WPT = 1, WX_LAYOUT_WPD = 2 };

// smallest dilation at which the fused sdwt / isdwt kernels slide register windows over a residue class (below it: one
// LDS read per tap)
static int wx_sdwt_window_min()
{
    static int v = -1;
    if (v < 0) { const char *e = wx_getenv("WX_SDWT_WIN_S"); v = (e && atoi(e) >= 1) ? atoi(e) : 0; }
    return v;
}
// default: every level for filters of four taps or more (db4, n = 4096, L = 6: sdwt 1.48 -> 1.02 ms, isdwt 2.40 -> 1.85 ms --
// the 4-way LDS bank conflicts of the narrow classes cost less than the tap-by-tap loop); Haar has nothing to slide over
static int wx_sdwt_window_min_for(int F) { const int e = wx_sdwt_window_min(); return e ? e : (F >= 4 ? 1 : 16); }

static __device__ __forceinline__ void wx_fwd_cols(int layout, int L, int d, int b, int &pcol, int &lcol, int &hcol)
{
    if (layout == WX_LAYOUT_DWT) { pcol = L - d; lcol = L - d - 1; hcol = L - d; }
    else if (layout == WX_LAYOUT_WPT) { const int w = 1 << (L - d); pcol = b * w; lcol = pcol; hcol = pcol + (w >> 1); }
    else { pcol = (1 << d) - 1 + b; lcol = (1 << (d + 1)) - 1 + 2 * b; hcol = lcol + 1; }
}

// ------------------------------------------------------------------------------------------
// forward level: blockIdx.x = node, blockIdx.y = signal (grid-strided)
// ------------------------------------------------------------------------------------------
template <typename T, bool AC>
__global__ __launch_bounds__(1024) void k_swt_fwd_level(const T *__restrict__ x, T *__restrict__ xw, int n,
                                                       int ncols, int64_t batch, int L, int d, int layout,
                                                       WxFilt filt, WxAcFilt ac)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *v = reinterpret_cast<T *>(wx_smem);
    const int b = blockIdx.x;
    int pcol, lcol, hcol;
    wx_fwd_cols(layout, L, d, b, pcol, lcol, hcol);
    const int s = (1 << d) % n;                      // dilation (s < n whenever L <= maxtransformlevels(n))
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        T *base = xw + sig * (int64_t)n * ncols;
        const T *src = (d == 0) ? x + sig * (int64_t)n : base + (int64_t)pcol * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const T t = src[i];
            v[i] = t;
            if (d == 0 && layout == WX_LAYOUT_WPD) base[i] = t;      // root column of the packet table
        }
        __syncthreads();
        T *lo = base + (int64_t)lcol * n, *hi = base + (int64_t)hcol * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (!AC) {
                double a = 0.0, dd = 0.0;
                int k1 = i - s; if (k1 < 0) k1 += n;                 // (i + (0-1)s) mod n
                int k2 = i;
                for (int j = 0; j < filt.F; ++j) {
                    a = fma(filt.q[j], (double)v[k1], a);
                    dd = fma((j & 1) ? -filt.q[j] : filt.q[j], (double)v[k2], dd);
                    k1 += s; if (k1 >= n) k1 -= n;
                    k2 -= s; if (k2 < 0) k2 += n;
                }
                lo[i] = (T)a;
                hi[i] = (T)dd;
            } else {
                double S = 0.0;
                int km = i, kp = i;
                const int s2 = (2 * s) % n;
                km -= s; if (km < 0) km += n;
                kp += s; if (kp >= n) kp -= n;
                for (int l = 1; l < ac.F; l += 2) {                  // odd lags only
                    S = fma(ac.b[l - 1], (double)v[km] + (double)v[kp], S);
                    km -= s2; if (km < 0) km += n;
                    kp += s2; if (kp >= n) kp -= n;
                }
                const double c = ac.c1 * (double)v[i];
                lo[i] = (T)(c + S);
                hi[i] = (T)(c - S);
            }
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------
// swpd / acwpd (heap-ordered table), two levels per pass: the parent column and its two children stay in LDS, so the table
// is read once per six written columns instead of once per two (swt/swt_one_level.jl:99-127, acwt/acwt_one_level.jl:101-128
// as driven by SWT.jl:840-902 / ACWT.jl:733-759).  blockIdx.x = node of depth d, blockIdx.y = signal (grid-strided).
// ------------------------------------------------------------------------------------------
template <typename T, bool AC>
__global__ __launch_bounds__(1024) void k_swpd_fwd_two(const T *__restrict__ x, T *__restrict__ xw, int n, int ncols,
                                                      int64_t batch, int d, WxFilt filt, WxAcFilt ac)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *v = reinterpret_cast<T *>(wx_smem), *c0 = v + n, *c1 = c0 + n;
    const int b = blockIdx.x;
    const int pcol = (1 << d) - 1 + b, ccol = (1 << (d + 1)) - 1 + 2 * b, gcol = (1 << (d + 2)) - 1 + 4 * b;
    // one analysis step at dilation s: src (LDS) -> lo, hi (global) and, if keep, -> klo, khi (LDS)
    auto step = [&](const T *src, int s, T *lo, T *hi, T *klo, T *khi) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            double a, dd;
            if (!AC) {
                a = 0.0; dd = 0.0;
                int k1 = i - s; if (k1 < 0) k1 += n;
                int k2 = i;
                for (int j = 0; j < filt.F; ++j) {
                    a = fma(filt.q[j], (double)src[k1], a);
                    dd = fma((j & 1) ? -filt.q[j] : filt.q[j], (double)src[k2], dd);
                    k1 += s; if (k1 >= n) k1 -= n;
                    k2 -= s; if (k2 < 0) k2 += n;
                }
            } else {
                double S = 0.0;
                int km = i, kp = i;
                const int s2 = (2 * s) % n;
                km -= s; if (km < 0) km += n;
                kp += s; if (kp >= n) kp -= n;
                for (int l = 1; l < ac.F; l += 2) {                  // odd lags only
                    S = fma(ac.b[l - 1], (double)src[km] + (double)src[kp], S);
                    km -= s2; if (km < 0) km += n;
                    kp += s2; if (kp >= n) kp -= n;
                }
                const double c = ac.c1 * (double)src[i];
                a = c + S;
                dd = c - S;
            }
            lo[i] = (T)a;
            hi[i] = (T)dd;
            if (klo) { klo[i] = (T)a; khi[i] = (T)dd; }
        }
    };
    const int s = (1 << d) % n, s1 = (2 << d) % n;
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        T *base = xw + sig * (int64_t)n * ncols;
        const T *src = (d == 0) ? x + sig * (int64_t)n : base + (int64_t)pcol * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const T t = src[i];
            v[i] = t;
            if (d == 0) base[i] = t;                                 // root column of the packet table
        }
        __syncthreads();
        step(v, s, base + (int64_t)ccol * n, base + (int64_t)(ccol + 1) * n, c0, c1);
        __syncthreads();
        step(c0, s1, base + (int64_t)gcol * n, base + (int64_t)(gcol + 1) * n, nullptr, nullptr);
        step(c1, s1, base + (int64_t)(gcol + 2) * n, base + (int64_t)(gcol + 3) * n, nullptr, nullptr);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// acwpd top table WITH the JBB moments of its columns (BASELINE config 5).  The table of depths 0 .. D0 was
// written by three two-level passes, read back by the moment kernel (4.3 GB per 2048 signals) and read again, depth D0 only, by the
// subtree kernel.  Here a workgroup keeps its parent node and walks `its` signals in order (blockIdx.y, grid-strided), so the sums
// over the signal axis of everything it produces -- the root, two children, four grandchildren -- accumulate in registers: a thread
// owns NP positions of every column.  The columns nobody reads again (odd depths, the root) are not written at all; partial sums
// per workgroup go to a scratch array and k_acwpd_top_combine adds them in the fixed order of blockIdx.y (deterministic; the
// association differs from one sequential sum over the signals by rounding only: bestbasis/bestbasis_tree.jl:153-154).
// x * x is rounded on its own like the moment kernel (wx_sq_unfused, wx_jbb.hip): no fused multiply-add into the sum.
// ------------------------------------------------------------------------------------------
// SUMS = false (round 5): only the sums of squares -- the first moments of every column come from the transform of the SUM of the signals
// (the transform is linear: sum_b acwpd(x_b) = acwpd(sum_b x_b), api_acwpd_jbb_moments), one add less per coefficient and signal
template <int NP, bool SUMS>
__global__ __launch_bounds__(1024) void k_acwpd_top_two_mom(const double *__restrict__ x, double *__restrict__ xw, int n, int ncols,
                                                            int64_t batch, int d, int last, WxAcFilt ac, double *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    double *v = reinterpret_cast<double *>(wx_smem), *c0 = v + n, *c1 = c0 + n;
    const int b = blockIdx.x;
    const int pcol = (1 << d) - 1 + b, gcol = (1 << (d + 2)) - 1 + 4 * b;
    const int s = (1 << d) % n, s1 = (2 << d) % n;
    // moments: slot 0 = root (d == 0 only), 1, 2 = children, 3 .. 6 = grandchildren
    double ms[7][NP], mq[7][NP];
#pragma unroll
    for (int c = 0; c < 7; ++c)
#pragma unroll
        for (int p = 0; p < NP; ++p) ms[c][p] = mq[c][p] = 0.0;
    auto add = [&](double &sm, double &sq, double val) {
        if (SUMS) sm = __dadd_rn(sm, val);
        sq = __dadd_rn(sq, __dmul_rn(val, val));
    };
    // one autocorrelation step at dilation st (acwt/acwt_one_level.jl:101-128): lo = c + S, hi = c - S
    const int nm = n - 1;                                    // n is a power of two: the periodic wrap is a mask
    auto step = [&](const double *src, int st, int i, double &lo, double &hi) {
        double S = 0.0;
        int lag = st;                                         // wave-uniform: the index arithmetic per tap is two adds and two ands
        for (int l = 1; l < ac.F; l += 2) {                  // odd lags only
            S = fma(ac.b[l - 1], src[(i - lag) & nm] + src[(i + lag) & nm], S);
            lag += 2 * st;
        }
        const double c = ac.c1 * src[i];
        lo = c + S;
        hi = c - S;
    };
    // the parent column of the next signal travels while this one is computed
    double nxt[NP];
    auto fetch = [&](int64_t sig) {
        const double *src = (d == 0) ? x + sig * (int64_t)n : xw + sig * (int64_t)n * ncols + (int64_t)pcol * n;
#pragma unroll
        for (int p = 0; p < NP; ++p) nxt[p] = src[threadIdx.x + p * 1024];
    };
    if ((int64_t)blockIdx.y < batch) fetch(blockIdx.y);
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        double *base = xw + sig * (int64_t)n * ncols;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = threadIdx.x + p * 1024;
            const double t = nxt[p];
            v[i] = t;
            if (d == 0) add(ms[0][p], mq[0][p], t);
        }
        if (sig + gridDim.y < batch) fetch(sig + gridDim.y);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = threadIdx.x + p * 1024;
            double lo, hi;
            step(v, s, i, lo, hi);
            c0[i] = lo; c1[i] = hi;
            add(ms[1][p], mq[1][p], lo);
            add(ms[2][p], mq[2][p], hi);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = threadIdx.x + p * 1024;
            double g0, g1, g2, g3;
            step(c0, s1, i, g0, g1);
            step(c1, s1, i, g2, g3);
            // the grandchildren are the next pass's parents (or, after the last pass, the subtree kernel's input): they are written
            base[(int64_t)gcol * n + i] = g0;
            base[(int64_t)(gcol + 1) * n + i] = g1;
            base[(int64_t)(gcol + 2) * n + i] = g2;
            base[(int64_t)(gcol + 3) * n + i] = g3;
            add(ms[3][p], mq[3][p], g0);
            add(ms[4][p], mq[4][p], g1);
            add(ms[5][p], mq[5][p], g2);
            add(ms[6][p], mq[6][p], g3);
        }
        __syncthreads();
    }
    (void)last;
    // partial sums of this workgroup: part[((y * nodes + b) * 7 + slot) * n + i], sums first, squares behind them
    const int64_t nodes = gridDim.x;
    const int64_t slab = ((int64_t)blockIdx.y * nodes + b) * 7;
    const int64_t half = (int64_t)gridDim.y * nodes * 7 * n;
#pragma unroll
    for (int c = 0; c < 7; ++c)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = threadIdx.x + p * 1024;
            if (SUMS) part[(slab + c) * n + i] = ms[c][p];
            part[half + (slab + c) * n + i] = mq[c][p];
        }
}

// sum[col, i] (+)= sum over y of the partials, y ascending; col of (node b, slot): root 0, children 2^(d+1)-1 + 2b + {0,1},
// grandchildren 2^(d+2)-1 + 4b + {0..3}
struct WxTopComb {                       // the passes of one chunk: one combine launch for all of them (blockIdx.y = pass)
    int npass;
    int d[8], nodes[8], gy[8];
    int64_t off[8];                      // element offset of the pass's partials
};
__global__ __launch_bounds__(256) void k_acwpd_top_combine(const double *__restrict__ part_all, int n, WxTopComb tc, int acc,
                                                           double *__restrict__ sum, double *__restrict__ sumsq, int sums)
{
    const int ps = blockIdx.y;
    const int d = tc.d[ps], nodes = tc.nodes[ps], gy = tc.gy[ps];
    const double *part = part_all + tc.off[ps];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;               // over nodes * 7 * n
    if (e >= (int64_t)nodes * 7 * n) return;
    const int i = (int)(e % n);
    const int64_t t = e / n;
    const int slot = (int)(t % 7), b = (int)(t / 7);
    if (slot == 0 && d != 0) return;
    const int col = slot == 0 ? 0 : (slot <= 2 ? (1 << (d + 1)) - 1 + 2 * b + (slot - 1) : (1 << (d + 2)) - 1 + 4 * b + (slot - 3));
    const int64_t half = (int64_t)gy * nodes * 7 * n;
    double a = 0.0, q = 0.0;
    for (int y = 0; y < gy; ++y) {
        const int64_t o = (((int64_t)y * nodes + b) * 7 + slot) * n + i;
        if (sums) a = __dadd_rn(a, part[o]);
        q = __dadd_rn(q, part[half + o]);
    }
    const int64_t dst = (int64_t)col * n + i;
    if (sums) sum[dst] = acc ? __dadd_rn(sum[dst], a) : a;
    sumsq[dst] = acc ? __dadd_rn(sumsq[dst], q) : q;
}

// the top table of depths 0 .. D0 (only the even depths >= 2 are written: tab is (n, 2^(D0+1)-1, batch) like acwpd's) and the
// moments of ALL its columns; 1 = done, 0 = not applicable (the caller takes the three plain passes + the moment kernel), < 0 = error
bool wx_acwpd_top_moments_ok(int64_t n, int D0)
{
    static const bool off = wx_getenv("WX_ACWPD_TOPMOM") && atoi(wx_getenv("WX_ACWPD_TOPMOM")) == 0;
    if (off || wx_force_generic_swt() || D0 < 2 || (D0 & 1) || D0 > 14) return false;
    return n == 1024 || n == 2048 || n == 4096;
}
int wx_dev_acwpd_top_moments(const double *x, double *tab, int64_t n, int D0, int64_t batch, const WxAcFilt &ac, double *sum, double *sumsq,
                             int acc, hipStream_t st, bool sums)
{
    if (!wx_acwpd_top_moments_ok(n, D0) || batch < 1) return 0;
    const int NP = (int)(n / 1024);
    const int ncols = (1 << (D0 + 1)) - 1;
    const size_t lds = (size_t)3 * n * sizeof(double);
    WxScratch scr(st);
    // per pass nodes * gy workgroups, gy = 128, 64, 32, ... signal groups (the last pass has the most nodes); every pass keeps its
    // own partials and ONE combine launch adds them all at the end (one launch per pass until late in round 4: 3 x 29 us per chunk)
    WxTopComb tc = {};
    int64_t pelems = 0;
    for (int d = 0; d < D0; d += 2) {
        if (tc.npass >= 8) return 0;
        const int nodes = 1 << d;
        int64_t gy = 128 >> (d / 2); if (gy < 32) gy = 32; if (gy > batch) gy = batch;    // the partials' count: the combine walks them in order
        tc.d[tc.npass] = d; tc.nodes[tc.npass] = nodes; tc.gy[tc.npass] = (int)gy; tc.off[tc.npass] = pelems;
        pelems += (int64_t)2 * gy * nodes * 7 * n;
        ++tc.npass;
    }
    double *part = (double *)scr.alloc((size_t)pelems * sizeof(double));
    if (!part) return WX_EHIP;
    int64_t totmax = 0;
    for (int ps = 0; ps < tc.npass; ++ps) {
        const int d = tc.d[ps], nodes = tc.nodes[ps];
        const int64_t gy = tc.gy[ps];
#define WX_TM(NPP)                                                                                                                     \
        {                                                                                                                              \
            auto k = sums ? k_acwpd_top_two_mom<NPP, true> : k_acwpd_top_two_mom<NPP, false>;                                           \
            if (lds > 64 * 1024) WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(k, dim3(nodes, (unsigned)gy), dim3(1024), lds, st, x, tab, (int)n, ncols, batch, d, d + 2 >= D0 ? 1 : 0, ac, part + tc.off[ps]); \
        }
        if (NP == 1) WX_TM(1) else if (NP == 2) WX_TM(2) else WX_TM(4)
#undef WX_TM
        const int64_t tot = (int64_t)nodes * 7 * n;
        if (tot > totmax) totmax = tot;
    }
    hipLaunchKernelGGL(k_acwpd_top_combine, dim3((unsigned)((totmax + 255) / 256), (unsigned)tc.npass), dim3(256), 0, st, (const double *)part,
                       (int)n, tc, acc, sum, sumsq, sums ? 1 : 0);
    WX_HIP_CHECK(hipGetLastError());
    return 1;
}

// ------------------------------------------------------------------------------------------
// sdwt / acdwt with every level in one kernel: the running approximation stays in LDS (ping-pong), each
// level writes only its detail column, the last approximation goes to column 0.  HBM sees the signal once
// and every output column once ((L+2) n per signal instead of 3 L n one level at a time).
// ------------------------------------------------------------------------------------------
// FT > 0 (orthogonal filters): for dilations s >= 16 a lane owns R = 4 consecutive samples of one residue class
// mod s, so the F taps of both branches slide over a window of R + 2F - 3 class samples held in registers
// (22 LDS reads instead of 64 for F = 8); lanes run over consecutive residues, i.e. consecutive LDS words.
template <typename T, bool AC, int FT>
__global__ __launch_bounds__(1024) void k_sdwt_fused(const T *__restrict__ x, T *__restrict__ xw, int n, int64_t batch,
                                                    int L, WxFilt filt, WxAcFilt ac, int wmin)
{
    constexpr int R = 4;
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *a = reinterpret_cast<T *>(wx_smem);
    T *b = a + n;
    for (int64_t sig = blockIdx.x; sig < batch; sig += gridDim.x) {
        T *base = xw + sig * (int64_t)n * (L + 1);
        const T *src = x + sig * (int64_t)n;
        wx_stage<T>(a, src, n);
        __syncthreads();
        T *v = a, *w = b;
        for (int d = 0; d < L; ++d) {
            const int s = (1 << d) % n;
            T *hi = base + (int64_t)(L - d) * n;
            const int M = s > 0 ? n / s : 0;
            if (!AC && FT > 0 && s >= wmin && M >= R && M % R == 0 && (int64_t)M * s == n) {
                constexpr int FW = FT > 0 ? FT : 2;
                constexpr int WN = R + 2 * FW - 3;
                for (int qi = threadIdx.x; qi < n / R; qi += blockDim.x) {
                    const int c = qi % s, u0 = (qi / s) * R;
                    double W[WN];
                    int u = (u0 - (FW - 1)) % M; if (u < 0) u += M;
#pragma unroll
                    for (int k = 0; k < WN; ++k) { W[k] = (double)v[c + u * s]; u = u + 1 == M ? 0 : u + 1; }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        double lo = 0.0, dd = 0.0;
#pragma unroll
                        for (int j = 0; j < FW; ++j) {
                            lo = fma(filt.q[j], W[FW - 2 + r + j], lo);
                            dd = fma((j & 1) ? -filt.q[j] : filt.q[j], W[FW - 1 + r - j], dd);
                        }
                        const int i = c + (u0 + r) * s;
                        w[i] = (T)lo;
                        hi[i] = (T)dd;
                    }
                }
            } else
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                if (!AC) {
                    double lo = 0.0, dd = 0.0;
                    int k1 = i - s; if (k1 < 0) k1 += n;
                    int k2 = i;
                    for (int j = 0; j < filt.F; ++j) {
                        lo = fma(filt.q[j], (double)v[k1], lo);
                        dd = fma((j & 1) ? -filt.q[j] : filt.q[j], (double)v[k2], dd);
                        k1 += s; if (k1 >= n) k1 -= n;
                        k2 -= s; if (k2 < 0) k2 += n;
                    }
                    w[i] = (T)lo;
                    hi[i] = (T)dd;
                } else {
                    double S = 0.0;
                    int km = i, kp = i;
                    const int s2 = (2 * s) % n;
                    km -= s; if (km < 0) km += n;
                    kp += s; if (kp >= n) kp -= n;
                    for (int l = 1; l < ac.F; l += 2) {
                        S = fma(ac.b[l - 1], (double)v[km] + (double)v[kp], S);
                        km -= s2; if (km < 0) km += n;
                        kp += s2; if (kp >= n) kp -= n;
                    }
                    const double c = ac.c1 * (double)v[i];
                    w[i] = (T)(c + S);
                    hi[i] = (T)(c - S);
                }
            }
            __syncthreads();
            T *t = v; v = w; w = t;
        }
        for (int i = threadIdx.x; i < n; i += blockDim.x) base[i] = v[i];
        __syncthreads();
    }
}

// Average-based isdwt, every level in one kernel.  The mean of the two shift variants of a synthesis step is
// the shift-invariant adjoint filter (see k_swt_inv_multi):
//   r_d[p] = 1/2 sum_j q[j] r_{d+1}[p + (1-j) s] + (-1)^j q[j] w_d[p + j s],   s = 2^d,
// r_L = column 0, w_d = detail column L-d.  r ping-pongs in LDS, the detail column of the level is staged next
// to it; HBM sees every input column once and the signal once.
template <typename T, int FT, bool PIPE>
__global__ __launch_bounds__(1024) void k_isdwt_avg_fused(const T *__restrict__ xw, T *__restrict__ x, int n,
                                                         int64_t batch, int L, WxFilt filt, int wmin)
{
    constexpr int R = 4;
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *a = reinterpret_cast<T *>(wx_smem);
    T *b = a + n;
    T *wd = b + n;
    // PIPE (the host sets it when a CU holds a single workgroup and n <= PF blockDim.x): the detail column of the next level
    // is fetched into registers while the current level is computed -- a load issued only after the level's barrier left
    // HBM idle for its whole latency, L + 1 times per signal (now once, for the first two columns).  With several
    // workgroups per CU the other workgroups cover the latency and the registers are better spent on occupancy.
    constexpr int PF = PIPE ? 32 / sizeof(T) : 1;
    T pre[PF];
    const int NT = blockDim.x;
    for (int64_t sig = blockIdx.x; sig < batch; sig += gridDim.x) {
        const T *base = xw + sig * (int64_t)n * (L + 1);
        if (PIPE) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int i = threadIdx.x + k * NT;
                pre[k] = i < n ? base[n + i] : (T)0;
            }
        }
        wx_stage<T>(a, base, n);
        T *r = a, *rn = b;
        for (int d = L - 1; d >= 0; --d) {
            if (PIPE) {
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const int i = threadIdx.x + k * NT;
                    if (i < n) wd[i] = pre[k];
                }
                if (d > 0) {
                    const T *wnext = base + (int64_t)(L - d + 1) * n;
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        const int i = threadIdx.x + k * NT;
                        pre[k] = i < n ? wnext[i] : (T)0;
                    }
                }
            } else {
                wx_stage<T>(wd, base + (int64_t)(L - d) * n, n);
            }
            __syncthreads();
            const int s = (1 << d) % n;
            const int M = s > 0 ? n / s : 0;
            if (FT > 0 && s >= wmin && M >= R && M % R == 0 && (int64_t)M * s == n) {
                // R samples of one residue class per lane: r_c[u + 1 - j] and w_c[u + j] slide over R + F - 1 values each
                constexpr int FW = FT > 0 ? FT : 2;
                constexpr int WN = R + FW - 1;
                for (int qi = threadIdx.x; qi < n / R; qi += blockDim.x) {
                    const int c = qi % s, u0 = (qi / s) * R;
                    double Wr[WN], Ww[WN];
                    int ur = (u0 + 1 - (FW - 1)) % M; if (ur < 0) ur += M;      // r_c[u0 + 2 - F .. u0 + R]
                    int uw = u0;                                               // w_c[u0 .. u0 + R + F - 2]
#pragma unroll
                    for (int k = 0; k < WN; ++k) {
                        Wr[k] = (double)r[c + ur * s]; ur = ur + 1 == M ? 0 : ur + 1;
                        Ww[k] = (double)wd[c + uw * s]; uw = uw + 1 == M ? 0 : uw + 1;
                    }
#pragma unroll
                    for (int t = 0; t < R; ++t) {
                        double acc = 0.0;
#pragma unroll
                        for (int j = 0; j < FW; ++j) {
                            acc = fma(filt.q[j], Wr[t + (FW - 1) - j], acc);     // r_c[u0 + t + 1 - j]
                            acc = fma((j & 1) ? -filt.q[j] : filt.q[j], Ww[t + j], acc);
                        }
                        rn[c + (u0 + t) * s] = (T)(0.5 * acc);
                    }
                }
            } else
            for (int p = threadIdx.x; p < n; p += blockDim.x) {
                double acc = 0.0;
                int k1 = p + s; if (k1 >= n) k1 -= n;          // p + (1 - 0) s
                int k2 = p;                                    // p + 0 s
                for (int j = 0; j < filt.F; ++j) {
                    acc = fma(filt.q[j], (double)r[k1], acc);
                    acc = fma((j & 1) ? -filt.q[j] : filt.q[j], (double)wd[k2], acc);
                    k1 -= s; if (k1 < 0) k1 += n;
                    k2 += s; if (k2 >= n) k2 -= n;
                }
                rn[p] = (T)(0.5 * acc);
            }
            __syncthreads();
            T *t = r; r = rn; rn = t;
        }
        T *dst = x + sig * (int64_t)n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = r[i];
        __syncthreads();
    }
}

// element `idx` of an array whose base is the same for the whole wavefront: the base stays in scalar registers and the lane
// part is one 32-bit register (a generic pointer per load costs a 64-bit register pair each -- the unrolled tap loads of
// the fused kernels below spilled hundreds of registers that way)
template <typename T> static __device__ __forceinline__ T wx_uld(const T *base, unsigned idx)
{
    typedef const T __attribute__((address_space(1))) *P;
    P g = (P)base;
    asm("" : "+s"(g));
    return g[idx];
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load and store
// (vmcnt(0)), which puts the memory latency of a level's detail column in front of every barrier of the fused kernels below
static __device__ __forceinline__ void wx_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// Signals longer than the LDS of a CU (n sizeof(T) > 160 KiB; the reference has no length limit: swt/swt_one_level.jl:99-127,
// acwt/acwt_one_level.jl:101-128): one level per launch straight from global memory.  Thread i of a tile reads the taps
// v[(i + k s) mod n] -- consecutive lanes, consecutive addresses for every tap, the re-reads are served by L1 / L2.
// grid (tiles of the column, node, signal-strided).
// ------------------------------------------------------------------------------------------
template <typename T, bool AC>
__global__ __launch_bounds__(256) void k_swt_fwd_level_g(const T *__restrict__ x, T *__restrict__ xw, int n, int ncols,
                                                        int64_t batch, int L, int d, int layout, WxFilt filt, WxAcFilt ac,
                                                        const T *__restrict__ alt_in, T *__restrict__ alt_out, int alt_nc)
{
    const int b = blockIdx.y;
    int pcol, lcol, hcol;
    wx_fwd_cols(layout, L, d, b, pcol, lcol, hcol);
    const int s = (int)(((int64_t)1 << d) % n);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int64_t sig = blockIdx.z; sig < batch; sig += gridDim.z) {
        T *base = xw + sig * (int64_t)n * ncols;
        // sdwt / acdwt and swpt / acwpt write a child over the column their parent is read from (the one-CU kernels stage the
        // parent in LDS first): here the nodes a later level reads travel through two scratch arrays as well -- alt_in
        // (n, nodes of this depth, batch), alt_out (n, alt_nc x nodes, batch): the low child only (sdwt) or both (swpt)
        const int nodes = gridDim.y;
        const T *v = (d == 0) ? x + sig * (int64_t)n : (alt_in ? alt_in + (sig * nodes + b) * (int64_t)n : base + (int64_t)pcol * n);
        if (d == 0 && layout == WX_LAYOUT_WPD) base[i] = v[i];      // root column of the packet table
        double a, dd;
        if (!AC) {
            a = 0.0; dd = 0.0;
            int k1 = i - s; if (k1 < 0) k1 += n;
            int k2 = i;
            for (int j = 0; j < filt.F; ++j) {
                a = fma(filt.q[j], (double)v[k1], a);
                dd = fma((j & 1) ? -filt.q[j] : filt.q[j], (double)v[k2], dd);
                k1 += s; if (k1 >= n) k1 -= n;
                k2 -= s; if (k2 < 0) k2 += n;
            }
        } else {
            double S = 0.0;
            int km = i, kp = i;
            const int s2 = (int)((2 * (int64_t)s) % n);
            km -= s; if (km < 0) km += n;
            kp += s; if (kp >= n) kp -= n;
            for (int l = 1; l < ac.F; l += 2) {
                S = fma(ac.b[l - 1], (double)v[km] + (double)v[kp], S);
                km -= s2; if (km < 0) km += n;
                kp += s2; if (kp >= n) kp -= n;
            }
            const double c = ac.c1 * (double)v[i];
            a = c + S;
            dd = c - S;
        }
        base[(int64_t)lcol * n + i] = (T)a;
        base[(int64_t)hcol * n + i] = (T)dd;
        if (alt_out) {
            T *o = alt_out + ((sig * nodes + b) * alt_nc) * (int64_t)n;
            o[i] = (T)a;
            if (alt_nc > 1) o[n + i] = (T)dd;
        }
    }
}

// ------------------------------------------------------------------------------------------
// sdwt / acdwt, every level in one kernel, ONE column of LDS (k_sdwt_fused needs two): a thread keeps the NPT approximation
// samples it has computed in registers across the barrier and writes them over the parent afterwards.  For signals whose
// column fills most of a CU's LDS (n = 16384 Float64: 128 KiB, config 3's length).  n == NPT * blockDim.x, a power of two.
// The inverse (below) runs its taps OUTSIDE the thread's NPT outputs: NPT independent global loads per tap are in flight
// together and nothing but the NPT accumulators lives across a tap (a tap loop inside the output loop made every tap a load
// followed by its use, 32 round trips per level; unrolling both spilled 40 .. 850 registers).
// ------------------------------------------------------------------------------------------
template <typename T, bool AC, int NPT, int FT>
__global__ __launch_bounds__(1024) void k_sdwt_fused_ip(const T *__restrict__ x, T *__restrict__ xw, int n, int64_t batch,
                                                       int L, WxFilt filt, WxAcFilt ac)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *v = reinterpret_cast<T *>(wx_smem);
    const int NT = blockDim.x;
    for (int64_t sig = blockIdx.x; sig < batch; sig += gridDim.x) {
        T *base = xw + sig * (int64_t)n * (L + 1);
        wx_stage<T>(v, x + sig * (int64_t)n, n);
        wx_lds_barrier();
        for (int d = 0; d < L; ++d) {
            const int s = (1 << d) % n;
            T *hi = base + (int64_t)(L - d) * n;
            T lo_reg[NPT];
            // (forward: the taps are LDS reads and the detail goes out as stores nobody waits for, so one output at a time
            // with its taps unrolled is the fastest form measured -- 2.70 ms on config 3 against 3.30 ms with the taps outside)
#pragma unroll
            for (int it = 0; it < NPT; ++it) {
                const int i = threadIdx.x + it * NT;
                if (i >= n) continue;
                if (!AC && FT > 0) {
                    double lo = 0.0, dd = 0.0;
                    int k1 = i - s; if (k1 < 0) k1 += n;
                    int k2 = i;
#pragma unroll
                    for (int j = 0; j < FT; ++j) {
                        lo = fma(filt.q[j], (double)v[k1], lo);
                        dd = fma((j & 1) ? -filt.q[j] : filt.q[j], (double)v[k2], dd);
                        k1 += s; if (k1 >= n) k1 -= n;
                        k2 -= s; if (k2 < 0) k2 += n;
                    }
                    lo_reg[it] = (T)lo;
                    hi[i] = (T)dd;
                } else if (!AC) {
                    double lo = 0.0, dd = 0.0;
                    int k1 = i - s; if (k1 < 0) k1 += n;
                    int k2 = i;
                    for (int j = 0; j < filt.F; ++j) {
                        lo = fma(filt.q[j], (double)v[k1], lo);
                        dd = fma((j & 1) ? -filt.q[j] : filt.q[j], (double)v[k2], dd);
                        k1 += s; if (k1 >= n) k1 -= n;
                        k2 -= s; if (k2 < 0) k2 += n;
                    }
                    lo_reg[it] = (T)lo;
                    hi[i] = (T)dd;
                } else {
                    double S = 0.0;
                    int km = i, kp = i;
                    const int s2 = (2 * s) % n;
                    km -= s; if (km < 0) km += n;
                    kp += s; if (kp >= n) kp -= n;
                    for (int l = 1; l < ac.F; l += 2) {
                        S = fma(ac.b[l - 1], (double)v[km] + (double)v[kp], S);
                        km -= s2; if (km < 0) km += n;
                        kp += s2; if (kp >= n) kp -= n;
                    }
                    const double c = ac.c1 * (double)v[i];
                    lo_reg[it] = (T)(c + S);
                    hi[i] = (T)(c - S);
                }
            }
            wx_lds_barrier();
            if (d + 1 < L) {
#pragma unroll
                for (int it = 0; it < NPT; ++it) {
                    const int i = threadIdx.x + it * NT;
                    if (i < n) v[i] = lo_reg[it];
                }
                wx_lds_barrier();
            } else {
#pragma unroll
                for (int it = 0; it < NPT; ++it) {
                    const int i = threadIdx.x + it * NT;
                    if (i < n) base[i] = lo_reg[it];                  // the last approximation is column 0
                }
            }
        }
    }
}

// Average-based isdwt, every level in one kernel, ONE column of LDS: the running reconstruction r lives in LDS and is
// overwritten in place (outputs in registers across the barrier), the detail column of a level is read straight from
// global memory (F reads per sample, consecutive lanes on consecutive addresses).  See k_isdwt_avg_fused for the step:
//   r_d[p] = 1/2 sum_j q[j] r_{d+1}[p + (1 - j) s] + (-1)^j q[j] w_d[p + j s]
template <typename T, int NPT>
__global__ __launch_bounds__(1024) void k_isdwt_avg_fused_ip(const T *__restrict__ xw, T *__restrict__ x, int n, int64_t batch,
                                                            int L, WxFilt filt)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *r = reinterpret_cast<T *>(wx_smem);
    const int NT = blockDim.x, msk = n - 1, t0 = threadIdx.x;
    for (int64_t sig = blockIdx.x; sig < batch; sig += gridDim.x) {
        const T *base = xw + sig * (int64_t)n * (L + 1);
        wx_stage<T>(r, base, n);
        wx_lds_barrier();
        for (int d = L - 1; d >= 0; --d) {
            const T *wd = base + (int64_t)(L - d) * n;
            const int s = (1 << d) & msk;
            double acc[NPT];
#pragma unroll
            for (int it = 0; it < NPT; ++it) acc[it] = 0.0;
            for (int j = 0; j < filt.F; ++j) {
                const double qa = filt.q[j], qd = (j & 1) ? -qa : qa;
                const int o1 = (1 - j) * s, o2 = j * s;
                T w[NPT];
#pragma unroll
                for (int it = 0; it < NPT; ++it) w[it] = wx_uld<T>(wd, (unsigned)((t0 + it * NT + o2) & msk));
#pragma unroll
                for (int it = 0; it < NPT; ++it) {
                    const int p = t0 + it * NT;
                    acc[it] = fma(qa, (double)r[(p + o1) & msk], acc[it]);
                    acc[it] = fma(qd, (double)w[it], acc[it]);
                }
            }
            wx_lds_barrier();
            if (d > 0) {
#pragma unroll
                for (int it = 0; it < NPT; ++it) r[t0 + it * NT] = (T)(0.5 * acc[it]);
                wx_lds_barrier();
            } else {
                T *dst = x + sig * (int64_t)n;
#pragma unroll
                for (int it = 0; it < NPT; ++it) dst[t0 + it * NT] = (T)(0.5 * acc[it]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K levels per pass (swpt only): the 2^K descendants of depth d+K are computed straight from the
// LDS-resident parent with composite taps (products of the K per-level taps, merged per offset on
// the host).  The intermediate levels never touch HBM: a pass reads 2^d*n and writes 2^(d+K)*n.
//   table layout: coef[t * 2^K + c] multiplies v[(i + shift[t]) mod n], c = descendant (natural
//   order), t = index into the union of offsets
// ------------------------------------------------------------------------------------------
template <typename T, int NC>
__global__ __launch_bounds__(1024) void k_swt_fwd_multi(const T *__restrict__ x, T *__restrict__ xw, int n,
                                                        int ncols, int64_t batch, int L, int d, int K,
                                                        const double *__restrict__ coef,
                                                        const int *__restrict__ shift, int U)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *v = reinterpret_cast<T *>(wx_smem);
    const int b = blockIdx.x;
    const int wp = 1 << (L - d);                   // column pitch of depth-d nodes
    const int wc = 1 << (L - d - K);               // column pitch of the descendants
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        T *base = xw + sig * (int64_t)n * ncols;
        const T *src = (d == 0) ? x + sig * (int64_t)n : base + (int64_t)(b * wp) * n;
        wx_stage<T>(v, src, n);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            double acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = 0.0;
            // taps in groups of four: the table entries of a group (wave-uniform, scalar loads) and its four LDS reads
            // are issued together instead of one wait per tap; same order of the multiply-adds as one tap at a time
            int t = 0;
            for (; t + 4 <= U; t += 4) {
                double vv[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    int k = i + shift[t + tt]; if (k >= n) k -= n;
                    vv[tt] = (double)v[k];
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] = fma(coef[(t + tt) * NC + c], vv[tt], acc[c]);
            }
            for (; t < U; ++t) {
                int k = i + shift[t]; if (k >= n) k -= n;
                const double vv = (double)v[k];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = fma(coef[t * NC + c], vv, acc[c]);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) base[(int64_t)((b * NC + c) * wc) * n + i] = (T)acc[c];
        }
        __syncthreads();
    }
}


// Sliding taps of one descendant: a thread owns OPT consecutive rows (class samples) u0 .. u0 + OPT - 1 of one residue, the W
// = U + OPT - 1 class samples from row k on are read once each and feed the OPT running sums; cp[w - j] is the tap table of
// the descendant, zero padded by OPT - 1 on both sides.  Window samples go in groups of four: the table entries of a group
// (wave-uniform, scalar loads) and its four LDS reads are issued together, not one wait per sample.  vb = tile + residue.
template <typename T, int OPT>
static __device__ __forceinline__ void wx_slide_taps(const T *vb, int lgR, int k, int nu, const double *__restrict__ cp, int W,
                                                     double (&acc)[OPT])
{
    int w = 0;
    for (; w + 4 <= W; w += 4) {
        double val[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            val[tt] = (double)vb[k << lgR];
            k = (k + 1 == nu) ? 0 : k + 1;
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int j = 0; j < OPT; ++j) acc[j] = fma(cp[w + tt - j], val[tt], acc[j]);
    }
    for (; w < W; ++w) {
        const double val = (double)vb[k << lgR];
        k = (k + 1 == nu) ? 0 : k + 1;
#pragma unroll
        for (int j = 0; j < OPT; ++j) acc[j] = fma(cp[w - j], val, acc[j]);
    }
}

// Residue-class variant for deep passes (s = 2^d large): descendant samples of class r (mod s)
// depend only on the parent's samples of class r, so a workgroup stages an (n/s) x R tile of the
// parent (runs of R consecutive samples every s) instead of the whole column: small LDS footprint,
// several workgroups per CU.  The composite taps of one descendant are U contiguous offsets (in units of s) from
// ustart[c] on (reduced mod n/s): each descendant slides over its own window (wx_slide_taps), so no multiply-add is spent
// on the offsets that only the other descendants use, and the address arithmetic is per window sample, not per tap.
// coefp[c][OPT - 1 + t] = tap t of descendant c, zero padded by OPT - 1 on both sides (pitch U + 2 (OPT - 1)).
// blockDim.x = (n/s) R / OPT exactly; (n/s) is a multiple of OPT.
template <typename T, int NC, int OPT>
__global__ __launch_bounds__(512) void k_swt_fwd_multi_rc(const T *__restrict__ x, T *__restrict__ xw, int n,
                                                          int ncols, int64_t batch, int L, int d, int K, int R,
                                                          const double *__restrict__ coefp,
                                                          const int *__restrict__ ustart, int U)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *v = reinterpret_cast<T *>(wx_smem);
    const int s = 1 << d;
    const int nu = n >> d;
    const int nblk = s / R;
    const int b = blockIdx.x / nblk;
    const int r0 = (blockIdx.x - b * nblk) * R;
    const int lgR = __ffs(R) - 1;
    const int NT = blockDim.x;
    const int wp = 1 << (L - d);
    const int wc = 1 << (L - d - K);
    const int r = threadIdx.x & (R - 1);
    const int u0 = (threadIdx.x >> lgR) * OPT;
    const int UP = U + 2 * (OPT - 1);
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        T *base = xw + sig * (int64_t)n * ncols;
        const T *src = ((d == 0) ? x + sig * (int64_t)n : base + (int64_t)(b * wp) * n) + r0;
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int o = threadIdx.x + j * NT;
            v[o] = src[(o & (R - 1)) + (int64_t)(o >> lgR) * s];
        }
        __syncthreads();
        T *dp = base + r0 + r + (int64_t)u0 * s;
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            double acc[OPT];
#pragma unroll
            for (int j = 0; j < OPT; ++j) acc[j] = 0.0;
            int k = u0 + ustart[c]; if (k >= nu) k -= nu;
            wx_slide_taps<T, OPT>(v + r, lgR, k, nu, coefp + c * UP + (OPT - 1), U + OPT - 1, acc);
            T *dc = dp + (int64_t)((b * NC + c) * wc) * n;
#pragma unroll
            for (int j = 0; j < OPT; ++j) dc[(int64_t)j * s] = (T)acc[j];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// SWT inverse level.  Children/parents are addressed through a small descriptor so that the same
// kernel serves isdwt / iswpt / iswpd (tree-driven).  One thread per parent sample it owns.
//   sm_mode 0: average based (all n positions); 1: shift based (only class sv, n/s positions)
// ------------------------------------------------------------------------------------------
struct WxInvDesc {
    const void *in;        // caller's coefficient array (read only), column stride n, signal stride n*ncols
    void *cur;             // scratch holding computed nodes of depth d+1 (column = node index in level)
    void *out;             // destination for depth d (scratch, or the final x when d == 0)
    int64_t cur_cols;      // columns per signal in cur
    int64_t out_cols;      // columns per signal in out
    int ncols;             // columns per signal in `in`
    int layout, L, d;
    const uint8_t *tree;   // heap-ordered tree bytes (WPD layout only; nullptr = full tree of depth L)
    int64_t ntree;
};

template <typename T>
static __device__ __forceinline__ const T *wx_inv_child(const WxInvDesc &D, int64_t sig, int n, int b, int which,
                                                        bool &valid)
{
    // child `which` (0 low / 1 high) of node b at depth d; returns its column base for signal sig
    const T *in = reinterpret_cast<const T *>(D.in) + sig * (int64_t)n * D.ncols;
    const T *cur = reinterpret_cast<const T *>(D.cur) + sig * (int64_t)n * D.cur_cols;
    valid = true;
    if (D.layout == WX_LAYOUT_DWT) {
        // w1 = running reconstruction (cur column 0, or in[:,0] at the first step), w2 = in[:, L-d]
        if (which == 0) return (D.d == D.L - 1) ? in : cur;
        return in + (int64_t)(D.L - D.d) * n;
    }
    const int cb = 2 * b + which;                        // child index within depth d+1
    if (D.layout == WX_LAYOUT_WPT)
        return (D.d == D.L - 1) ? in + (int64_t)cb * n : cur + (int64_t)cb * n;
    const int64_t heap = ((int64_t)1 << (D.d + 1)) + cb; // 1-based heap index of the child
    const bool computed = D.tree ? (heap <= D.ntree && D.tree[heap - 1]) : (D.d + 1 < D.L);
    if (computed) return cur + (int64_t)cb * n;
    if (heap > D.ncols) { valid = false; return in; }
    return in + (heap - 1) * (int64_t)n;
}

template <typename T>
__global__ __launch_bounds__(256) void k_swt_inv_level(WxInvDesc D, int n, int64_t batch, int sm_mode, int sv,
                                                       int sw, WxFilt filt)
{
    const int d = D.d;
    const int s = 1 << d;                // parent stride
    const int np = n >> d;               // parent samples per residue class
    const int nc = np >> 1;              // child samples per residue class
    const int nodes = (D.layout == WX_LAYOUT_DWT) ? 1 : (1 << d);
    const int per_node = sm_mode ? np : n;
    const int64_t total = (int64_t)batch * nodes * per_node;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(g % per_node);
        const int64_t t2 = g / per_node;
        const int b = (int)(t2 % nodes);
        const int64_t sig = t2 / nodes;
        if (D.layout == WX_LAYOUT_WPD && D.tree) {
            const int64_t heap = ((int64_t)1 << d) + b;
            if (!(heap <= D.ntree && D.tree[heap - 1])) continue;        // node has no children: nothing to do
        }
        bool ok1, ok2;
        const T *w1 = wx_inv_child<T>(D, sig, n, b, 0, ok1);
        const T *w2 = wx_inv_child<T>(D, sig, n, b, 1, ok2);
        T *out = reinterpret_cast<T *>(D.out) + sig * (int64_t)n * D.out_cols +
                 (int64_t)((D.layout == WX_LAYOUT_DWT || d == 0) ? 0 : b) * n;
        int p, cls;
        if (sm_mode) { cls = sv; p = -1; } else { p = r; cls = p & (s - 1); }
        double acc = 0.0;
        const int nvar = sm_mode ? 1 : 2;
        int pos_out = 0;
        for (int var = 0; var < nvar; ++var) {
            // variant A: sw == sv (parent sample t0 stored at class position t0-1); B: sw == sv + s
            const bool shifted = sm_mode ? (sw != sv) : (var == 1);
            const int swc = shifted ? cls + s : cls;
            int t0;
            if (sm_mode) {
                t0 = r;                                   // parent sample index inside the class
                const int u = shifted ? t0 : (t0 == 0 ? np - 1 : t0 - 1);
                pos_out = cls + u * s;
            } else {
                const int u = p >> d;
                t0 = shifted ? u : (u + 1 == np ? 0 : u + 1);
                pos_out = p;
            }
            const int tau = t0 >> 1;
            const bool odd = t0 & 1;
            int k1 = tau, k2 = tau;
            double v = 0.0;
            for (int m = 0; m < filt.F / 2; ++m) {
                const double a = (double)w1[swc + (int64_t)k1 * 2 * s];
                const double c = (double)w2[swc + (int64_t)k2 * 2 * s];
                if (!odd) { v = fma(filt.q[2 * m], a, v); v = fma(-filt.q[2 * m + 1], c, v); }
                else { v = fma(filt.q[2 * m + 1], a, v); v = fma(filt.q[2 * m], c, v); }
                k1 = k1 == 0 ? nc - 1 : k1 - 1;
                k2 = k2 + 1 == nc ? 0 : k2 + 1;
            }
            acc += v;
        }
        out[pos_out] = (T)(sm_mode ? acc : acc * 0.5);
    }
}

// The same level, average based, out of an LDS tile (round 6: VERDICT r5 item 6 (i)).  k_swt_inv_level gives every output sample its own 2 F
// global loads -- the vector cache serves them, but it is the kernel's bound: the top levels of a long column, which no fused pass takes
// (they need whole classes of a node in LDS), ran at a quarter of the HBM rate each, and `iswpt` of 16384-sample signals at depth 4 was four
// such launches: 0.08 of the roofline.  Here a workgroup stages P consecutive positions of both children of a node plus the halo
// H = s (F - 1) the dilated taps reach (positions are periodic in n), and every output reads its 2 F taps from LDS: with
// t0 = u + 1 (variant A) resp. u (variant B), u = p >> d, the first tap of either child sits at q0 = p (t0 odd) or p + s (t0 even) and the
// taps walk 2 s apart, downwards in the low child and upwards in the high one -- the index arithmetic of k_swt_inv_level without its wraps.
// Same products, same order of additions as that kernel.
template <typename T, int HF>
__global__ __launch_bounds__(256) void k_swt_inv_level_tile(WxInvDesc D, int n, int64_t batch, int P, int H, WxFilt filt)
{
    extern __shared__ __attribute__((aligned(16))) char wx_tile_smem[];
    T *t1 = reinterpret_cast<T *>(wx_tile_smem), *t2 = t1 + (P + 2 * H);
    const int d = D.d, s = 1 << d;
    const int nodes = (D.layout == WX_LAYOUT_DWT) ? 1 : (1 << d);
    const int tiles = n / P;
    const int64_t units = (int64_t)batch * nodes * tiles;
    for (int64_t unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int tl = (int)(unit % tiles);
        const int64_t t2i = unit / tiles;
        const int b = (int)(t2i % nodes);
        const int64_t sig = t2i / nodes;
        if (D.layout == WX_LAYOUT_WPD && D.tree) {
            const int64_t heap = ((int64_t)1 << d) + b;
            if (!(heap <= D.ntree && D.tree[heap - 1])) continue;        // node has no children: nothing to do (uniform for the workgroup)
        }
        bool ok1, ok2;
        const T *w1 = wx_inv_child<T>(D, sig, n, b, 0, ok1);
        const T *w2 = wx_inv_child<T>(D, sig, n, b, 1, ok2);
        T *out = reinterpret_cast<T *>(D.out) + sig * (int64_t)n * D.out_cols + (int64_t)((D.layout == WX_LAYOUT_DWT || d == 0) ? 0 : b) * n;
        const int p0 = tl * P;
        __syncthreads();                                          // the previous unit's reads are done
        {
            // eight loads of either child in flight per lane (one at a time left the staging bound by the load latency)
            const int tot = P + 2 * H;
            int i = threadIdx.x;
            for (; i + 3 * 256 < tot; i += 4 * 256) {
                T a[4], c[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int q = p0 - H + i + 256 * e;
                    q = q < 0 ? q + n : (q >= n ? q - n : q);     // H <= n
                    a[e] = w1[q];
                    c[e] = w2[q];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { t1[i + 256 * e] = a[e]; t2[i + 256 * e] = c[e]; }
            }
            for (; i < tot; i += 256) {
                int q = p0 - H + i;
                q = q < 0 ? q + n : (q >= n ? q - n : q);
                t1[i] = w1[q];
                t2[i] = w2[q];
            }
        }
        __syncthreads();
        const int np = n >> d;
        for (int i = threadIdx.x; i < P; i += 256) {
            const int p = p0 + i, u = p >> d;
            double acc = 0.0;
#pragma unroll
            for (int var = 0; var < 2; ++var) {
                const int t0 = var ? u : (u + 1 == np ? 0 : u + 1);
                const bool odd = t0 & 1;
                int k1 = H + i + (odd ? 0 : s), k2 = k1;
                double v = 0.0;
                T av[HF], cv[HF];
#pragma unroll
                for (int m = 0; m < HF; ++m) { av[m] = t1[k1 - 2 * s * m]; cv[m] = t2[k2 + 2 * s * m]; }
#pragma unroll
                for (int m = 0; m < HF; ++m) {
                    const double a = (double)av[m], c = (double)cv[m];
                    if (!odd) { v = fma(filt.q[2 * m], a, v); v = fma(-filt.q[2 * m + 1], c, v); }
                    else { v = fma(filt.q[2 * m + 1], a, v); v = fma(filt.q[2 * m], c, v); }
                }
                acc += v;
            }
            out[p] = (T)(acc * 0.5);
        }
    }
}


// ------------------------------------------------------------------------------------------
// Fused average-based iswpt pass: depth d+K -> d in one kernel.  The average of the two shift
// variants of a stationary synthesis step is the shift-invariant filter
//   parent[p] = 1/2 * sum_j q[j] lo[p + (1-j) s] + (-1)^j q[j] hi[p + j s],      s = 2^d
// (the adjoint of the analysis step), so K steps compose into one set of taps per descendant,
// all at multiples of s: a workgroup that owns R residue classes mod s of one node needs only
// those classes of the 2^K descendants.  LDS tile v[c][u][r] = desc_c[r0 + r + u s].
// ------------------------------------------------------------------------------------------
template <typename T, int NC, int OPT>
__global__ __launch_bounds__(512) void k_swt_inv_multi(const T *__restrict__ src, int64_t src_cols,
                                                       T *__restrict__ dst, int64_t dst_cols, int n, int64_t batch,
                                                       int d, int R, const double *__restrict__ coefp,
                                                       const int *__restrict__ ustart, int U)
{
    // One descendant column tile at a time through a double-buffered LDS tile, accumulators in
    // registers; the next column's tile is fetched into registers while the current one is used.
    // A thread owns OPT consecutive rows u0..u0+OPT-1 of one residue r, so the U contiguous taps of a descendant
    // slide over a window of OPT+U-1 LDS values (not OPT*U, wx_slide_taps): coefp[c][w - j + OPT-1] is the tap
    // table of descendant c zero-padded by OPT-1 on both sides, ustart[c] = its first tap offset reduced mod nu.
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *v = reinterpret_cast<T *>(wx_smem);
    const int s = 1 << d;
    const int nu = n >> d;
    const int nblk = s / R;
    const int b = blockIdx.x / nblk;
    const int r0 = (blockIdx.x - b * nblk) * R;
    const int tile = nu * R;
    const int lgR = __ffs(R) - 1;
    const int r = threadIdx.x & (R - 1);
    const int u0 = (threadIdx.x >> lgR) * OPT;
    const int UP = U + 2 * (OPT - 1);
    const int64_t g0 = r + (int64_t)u0 * s;
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        const T *sp = src + (sig * src_cols + (int64_t)b * NC) * n + r0 + g0;
        T pre[OPT];
        double acc[OPT];
#pragma unroll
        for (int j = 0; j < OPT; ++j) { acc[j] = 0.0; v[((u0 + j) << lgR) + r] = sp[(int64_t)j * s]; }
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            const T *vb = v + (c & 1) * tile + r;
            if (c + 1 < NC) {
#pragma unroll
                for (int j = 0; j < OPT; ++j) pre[j] = sp[(int64_t)(c + 1) * n + (int64_t)j * s];
            }
            int k = u0 + ustart[c]; if (k >= nu) k -= nu;
            wx_slide_taps<T, OPT>(vb, lgR, k, nu, coefp + c * UP + (OPT - 1), U + OPT - 1, acc);
            if (c + 1 < NC) {
                T *vn = v + ((c + 1) & 1) * tile;
#pragma unroll
                for (int j = 0; j < OPT; ++j) vn[((u0 + j) << lgR) + r] = pre[j];
            }
            __syncthreads();
        }
        T *dp = dst + (sig * dst_cols + b) * n + r0 + g0;
#pragma unroll
        for (int j = 0; j < OPT; ++j) dp[(int64_t)j * s] = (T)acc[j];
    }
}

// ------------------------------------------------------------------------------------------
// ACWT inverses: pure pairwise sums (acwt_one_level.jl:217-224), bit-compatible evaluation order
// ------------------------------------------------------------------------------------------
// iacdwt! ACWT.jl:287-304: x = xw[:,0]; for d = L-1..0: x = (x + xw[:, L-d]) / sqrt2
template <typename T>
__global__ __launch_bounds__(256) void k_iacdwt(const T *__restrict__ xw, T *__restrict__ x, int n, int L,
                                                int64_t batch)
{
    const double sqrt2 = 1.4142135623730951;
    const int64_t total = (int64_t)batch * n;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t sig = g / n;
        const int i = (int)(g - sig * n);
        const T *c = xw + sig * (int64_t)n * (L + 1) + i;
        T v = c[0];
        for (int d = L - 1; d >= 0; --d) v = (T)(((double)v + (double)c[(int64_t)(L - d) * n]) / sqrt2);
        x[g] = v;
    }
}

// iacwpt! ACWT.jl:581-610: binary tree of (low + high)/sqrt2 over the 2^L leaf columns
template <typename T>
__global__ __launch_bounds__(256) void k_iacwpt(const T *__restrict__ xw, T *__restrict__ x, int n, int L,
                                                int64_t batch)
{
    const double sqrt2 = 1.4142135623730951;
    const int64_t total = (int64_t)batch * n;
    const int m = 1 << L;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t sig = g / n;
        const int i = (int)(g - sig * n);
        const T *c = xw + sig * (int64_t)n * m + i;
        T st[32];
        for (int b = 0; b < m; ++b) {
            T val = c[(int64_t)b * n];
            int k = b, lvl = 0;
            while (k & 1) { val = (T)(((double)st[lvl] + (double)val) / sqrt2); k >>= 1; ++lvl; }
            st[lvl] = val;
        }
        x[g] = st[L];
    }
}

// iacwpd! ACWT.jl:944-968: value(node) = leaf column, or (value(2i) + value(2i+1))/sqrt2 when
// tree[i]; iterative post-order walk, identical for every thread (uniform control flow)
template <typename T>
__global__ __launch_bounds__(256) void k_iacwpd(const T *__restrict__ xw, T *__restrict__ x, int n, int ncols,
                                                int64_t batch, const uint8_t *__restrict__ tree, int64_t ntree,
                                                int Lfull)
{
    const double sqrt2 = 1.4142135623730951;
    const int64_t total = (int64_t)batch * n;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t sig = g / n;
        const int i = (int)(g - sig * n);
        const T *c = xw + sig * (int64_t)n * ncols + i;
        T val[34];
        int64_t node = 1;
        int depth = 0;
        // state machine: descend left while the node has children; at a leaf load it, then climb
        // while we are a right child, combining with the stored left value
        while (true) {
            const bool haskids = tree ? (node <= ntree && tree[node - 1]) : (depth < Lfull);
            if (haskids) { node = 2 * node; ++depth; continue; }
            T v = c[(node - 1) * (int64_t)n];
            while (node > 1 && (node & 1)) {                     // right child: combine with left sibling
                v = (T)(((double)val[depth] + (double)v) / sqrt2);
                node >>= 1; --depth;
            }
            if (node == 1) { x[g] = v; break; }
            val[depth] = v;                                       // left child done: go to the sibling
            node = node + 1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
static int64_t wx_swtfwd_lds_bytes()
{
    static int64_t v = -1;
    if (v < 0) { const char *e = wx_getenv("WX_SWTFWD_LDS_KIB"); v = (e && atoi(e) > 0 && atoi(e) <= 64 ? atoi(e) : 32) * 1024; }
    return v;
}
static int wx_grid1(int64_t total)
{
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

// composite taps of K consecutive stationary levels starting at dilation s (units of s):
// level k (0-based) has taps lo: offset (j-1)*2^k, hi: offset -j*2^k
static void wx_swt_composite(const WxFilt &f, int K, std::vector<double> &coef, std::vector<int> &offs)
{
    const int NC = 1 << K;
    std::vector<std::map<int, double>> filt(NC);
    for (int c = 0; c < NC; ++c) {
        std::map<int, double> cur;
        cur[0] = 1.0;
        for (int k = 0; k < K; ++k) {
            const int hi = (c >> (K - 1 - k)) & 1;              // first level's choice is the MSB
            std::map<int, double> nxt;
            for (auto &e : cur)
                for (int j = 0; j < f.F; ++j) {
                    const int o = hi ? -j * (1 << k) : (j - 1) * (1 << k);
                    const double q = hi ? ((j & 1) ? -f.q[j] : f.q[j]) : f.q[j];
                    nxt[e.first + o] += e.second * q;
                }
            cur.swap(nxt);
        }
        filt[c] = cur;
    }
    std::map<int, int> uni;
    for (auto &m : filt) for (auto &e : m) uni[e.first] = 0;
    int U = 0;
    offs.clear();
    for (auto &e : uni) { e.second = U++; offs.push_back(e.first); }
    coef.assign((size_t)NC * U, 0.0);
    for (int c = 0; c < NC; ++c) for (auto &e : filt[c]) coef[(size_t)c * U + uni[e.first]] = e.second;
}

template <typename T>
int wx_dev_swt_fwd(const T *x, T *xw, int64_t n, int L, int layout, int64_t batch, const WxFilt &filt,
                   const WxAcFilt *ac, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    const size_t lds = (size_t)n * sizeof(T);
    const int ncols = layout == WX_LAYOUT_DWT ? L + 1 : (layout == WX_LAYOUT_WPT ? (1 << L) : (1 << (L + 1)) - 1);
    WxAcFilt acz;
    if (ac) acz = *ac; else { acz.F = 0; acz.c1 = 0; }
    if (lds > 160 * 1024) {
        // longer than a CU's LDS: one level per launch from global memory (any length below 2^30)
        if (n >= ((int64_t)1 << 30)) return wx_set_error(WX_EUNSUPPORTED, "redundant transforms: signal length >= 2^30");
        auto kg = ac ? k_swt_fwd_level_g<T, true> : k_swt_fwd_level_g<T, false>;
        const unsigned tiles = (unsigned)((n + 255) / 256);
        if (L > 16) return wx_set_error(WX_EUNSUPPORTED, "redundant transforms of long signals: more than 65535 nodes per level");
        WxScratch scr(st);
        T *alt[2] = {nullptr, nullptr};
        const bool inplace_layout = layout != WX_LAYOUT_WPD;       // DWT, WPT: a child replaces its parent's column
        const int alt_nc = layout == WX_LAYOUT_WPT ? 2 : 1;
        if (inplace_layout && L >= 2) {
            // the widest array a level reads: (n, 2^(L-1), batch) for swpt, (n, 1, batch) for sdwt
            const size_t cols = layout == WX_LAYOUT_WPT ? ((size_t)1 << (L - 1)) : 1;
            alt[0] = (T *)scr.alloc(sizeof(T) * n * cols * batch);
            alt[1] = L >= 3 ? (T *)scr.alloc(sizeof(T) * n * cols * batch) : nullptr;
            if (!alt[0] || (L >= 3 && !alt[1])) return WX_EHIP;
        }
        for (int d = 0; d < L; ++d) {
            const int nodes = layout == WX_LAYOUT_DWT ? 1 : (1 << d);
            const T *ain = (inplace_layout && d > 0) ? alt[(d - 1) & 1] : nullptr;
            T *aout = (inplace_layout && d + 1 < L) ? alt[d & 1] : nullptr;
            hipLaunchKernelGGL(kg, dim3(tiles, (unsigned)nodes, (unsigned)(batch > 1024 ? 1024 : batch)), dim3(256), 0, st, x, xw,
                               (int)n, ncols, batch, L, d, layout, filt, acz, ain, aout, alt_nc);
        }
        WX_HIP_CHECK(hipGetLastError());
        return WX_OK;
    }
    auto kern = ac ? k_swt_fwd_level<T, true> : k_swt_fwd_level<T, false>;
    if (lds > 64 * 1024)
        WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nt = n >= 8192 ? 1024 : (n >= 2048 ? 512 : 256);
    if (layout == WX_LAYOUT_DWT && L >= 2 && 2 * lds <= 160 * 1024 && !wx_force_generic_swt()) {
        // sdwt / acdwt: every level in one kernel, approximation resident in LDS
        typedef void (*KF)(const T *, T *, int, int64_t, int, WxFilt, WxAcFilt, int);
        KF kf = ac ? k_sdwt_fused<T, true, 0> : k_sdwt_fused<T, false, 0>;
        if (!ac) switch (filt.F) {
#define WX_CASE(FF) case FF: kf = k_sdwt_fused<T, false, FF>; break;
            WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(14) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
            default: break;                              // other lengths: runtime tap loop
        }
        if (2 * lds > 64 * 1024)
            WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds)));
        int per_cu = (int)((160 * 1024) / (2 * lds));
        if (per_cu > 8) per_cu = 8;
        int64_t grid = (int64_t)256 * per_cu;
        if (grid > batch) grid = batch;
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(nt), 2 * lds, st, x, xw, (int)n, batch, L, filt, acz, wx_sdwt_window_min_for(filt.F));
        WX_HIP_CHECK(hipGetLastError());
        return WX_OK;
    }
    if (layout == WX_LAYOUT_DWT && L >= 2 && n == 1024 * (int64_t)(128 / sizeof(T)) && !wx_force_generic_swt() &&
        !(wx_getenv("WX_SDWT_INPLACE") && atoi(wx_getenv("WX_SDWT_INPLACE")) == 0)) {
        // the column fills more than half of a CU's LDS: in-place fused kernel, one workgroup of 1024 threads per CU
        constexpr int NPT = 128 / sizeof(T);
        typedef void (*KFI)(const T *, T *, int, int64_t, int, WxFilt, WxAcFilt);
        KFI ki = ac ? k_sdwt_fused_ip<T, true, NPT, 0> : k_sdwt_fused_ip<T, false, NPT, 0>;
        if (!ac) switch (filt.F) {
#define WX_CASE(FF) case FF: ki = k_sdwt_fused_ip<T, false, NPT, FF>; break;
            WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(14) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
            default: break;
        }
        WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ki), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int64_t grid = 256;
        if (grid > batch) grid = batch;
        hipLaunchKernelGGL(ki, dim3((unsigned)grid), dim3(1024), lds, st, x, xw, (int)n, batch, L, filt, acz);
        WX_HIP_CHECK(hipGetLastError());
        return WX_OK;
    }
    // swpt: fuse K levels per pass (K = 3 for very short filters, else 2); the tables live in a
    // stream-ordered scratch buffer
    // with the sliding windows of k_swt_fwd_multi_rc the two-level passes win for every filter length of the library
    // (coif6 / db10, n = 16384, L = 12: 20.3 -> 13.4 ms against single levels)
    static const int kf_maxf = wx_getenv("WX_SWTFWD_KF_MAXF") ? atoi(wx_getenv("WX_SWTFWD_KF_MAXF")) : 20;
    static const int k3_maxf = wx_getenv("WX_SWTFWD_K3_MAXF") ? atoi(wx_getenv("WX_SWTFWD_K3_MAXF")) : 4;
    const int KF = (!ac && layout == WX_LAYOUT_WPT && !wx_force_generic_swt()) ? (filt.F <= k3_maxf ? 3 : (filt.F <= kf_maxf ? 2 : 1)) : 1;
    double *dcoef = nullptr;
    int *dshift = nullptr;
    int d = 0;
    // Haar, deep trees: the last six levels are one register pass (wx_haarswt.hip); the passes here stop at depth L - 6
    const bool haar6 = !ac && layout == WX_LAYOUT_WPT && !wx_force_generic_swt() && wx_haar_swpt6_ok(n, L, filt, sizeof(T));
    // any filter, deep trees: the levels from depth log2(n) - 4 on are a lane-local register pass (wx_swtdeep.hip)
    const int deepLP = (!haar6 && (layout == WX_LAYOUT_WPT || layout == WX_LAYOUT_WPD) && !wx_force_generic_swt())
                           ? wx_swpt_deep_levels(n, L, ac ? ac->F : filt.F, ac != nullptr, sizeof(T)) : 0;
    const int dstop = haar6 ? L - 6 : L - deepLP;
    while (d < dstop) {
        const int K = (KF > 1 && dstop - d >= 2) ? (dstop - d >= KF ? KF : dstop - d) : 1;
        int64_t gy = batch;
        if (gy > 65535) gy = 65535;
        // swpd / acwpd: two levels per pass while three columns fit the LDS (WX_SWPD_TWO=0: one level per pass)
        static const bool two_off = wx_getenv("WX_SWPD_TWO") && atoi(wx_getenv("WX_SWPD_TWO")) == 0;
        if (K == 1 && layout == WX_LAYOUT_WPD && dstop - d >= 2 && 3 * lds <= 160 * 1024 && !two_off && !wx_force_generic_swt()) {
            auto k2 = ac ? k_swpd_fwd_two<T, true> : k_swpd_fwd_two<T, false>;
            if (3 * lds > 64 * 1024)
                WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)(3 * lds)));
            hipLaunchKernelGGL(k2, dim3(1 << d, (unsigned)gy), dim3(nt), 3 * lds, st, x, xw, (int)n, ncols, batch, d, filt, acz);
            d += 2;
            continue;
        }
        if (K == 1) {
            const int nodes = layout == WX_LAYOUT_DWT ? 1 : (1 << d);
            // long signals occupy most of a CU's LDS (one workgroup per CU): give that workgroup 16 waves
            hipLaunchKernelGGL(kern, dim3(nodes, (unsigned)gy), dim3(nt), lds, st, x, xw, (int)n, ncols, batch, L, d,
                               layout, filt, acz);
            d += 1;
            continue;
        }
        std::vector<double> coef;
        std::vector<int> offs;
        wx_swt_composite(filt, K, coef, offs);
        const int U = (int)offs.size();
        std::vector<int> shift(U);
        const int64_t sdil = (int64_t)1 << d;
        for (int t = 0; t < U; ++t) {
            int64_t sh = ((int64_t)offs[t] * sdil) % n;
            if (sh < 0) sh += n;
            shift[t] = (int)sh;
        }
        // residue-class tiles when runs of >= 64 bytes fit the per-workgroup tile budget (and 512 threads x 8 rows)
        int Rrc = 0, OPTrc = 0;
        {
            const int64_t nu = n >> d;
            int64_t r = (int64_t)(wx_swtfwd_lds_bytes() / sizeof(T)) / nu, rp = 1;
            while (rp * 2 <= r) rp *= 2;
            if (r < 1) rp = 0;
            if (rp > sdil) rp = sdil;
            if (rp > 128) rp = 128;
            while (rp > 1 && nu * rp > 4096) rp >>= 1;
            if (rp * (int64_t)sizeof(T) >= 64 && nu * rp <= 4096) {
                // rows per thread: four when the tile gives at least a wavefront of threads, at most 512 threads
                int o = 4;
                while (o > 1 && (nu % o != 0 || nu * rp / o < 64)) o >>= 1;
                while (o < 8 && nu * rp / o > 512 && nu % (2 * o) == 0) o <<= 1;
                if (nu % o == 0 && nu * rp / o <= 512) { Rrc = (int)rp; OPTrc = o; }
            }
        }
        if (Rrc) {
            // per descendant: the contiguous run of offsets it uses, padded for the sliding window
            const int64_t nu = n >> d;
            const int NCc = 1 << K;
            for (int t = 1; t < U; ++t)
                if (offs[t] != offs[0] + t) return wx_set_error(WX_EHIP, "swpt: composite taps are not contiguous");
            std::vector<int> first(NCc, 0), len(NCc, 1);
            int Uc = 1;
            for (int c = 0; c < NCc; ++c) {
                int f = -1, l = -1;
                for (int t = 0; t < U; ++t) if (coef[(size_t)c * U + t] != 0.0) { if (f < 0) f = t; l = t; }
                if (f < 0) { f = 0; l = 0; }
                first[c] = f; len[c] = l - f + 1;
                if (len[c] > Uc) Uc = len[c];
            }
            const int UP = Uc + 2 * (OPTrc - 1);
            std::vector<double> pad((size_t)NCc * UP, 0.0);
            std::vector<int> ust(NCc);
            for (int c = 0; c < NCc; ++c) {
                for (int t = 0; t < len[c]; ++t) pad[(size_t)c * UP + (OPTrc - 1) + t] = coef[(size_t)c * U + first[c] + t];
                int64_t o = (int64_t)offs[first[c]] % nu;
                if (o < 0) o += nu;
                ust[c] = (int)o;
            }
            dcoef = (double *)wx_const_upload(pad.data(), pad.size() * sizeof(double), st, true);
            dshift = (int *)wx_const_upload(ust.data(), ust.size() * sizeof(int), st, true);
            if (!dcoef || !dshift) return WX_EHIP;
            const int64_t tile = nu * Rrc;
            const int NT = (int)(tile / OPTrc);
            typedef void (*KM)(const T *, T *, int, int, int64_t, int, int, int, int, const double *, const int *, int);
            KM kr = nullptr;
            if (K == 2) kr = OPTrc == 8 ? k_swt_fwd_multi_rc<T, 4, 8> : OPTrc == 4 ? k_swt_fwd_multi_rc<T, 4, 4> : OPTrc == 2 ? k_swt_fwd_multi_rc<T, 4, 2> : k_swt_fwd_multi_rc<T, 4, 1>;
            else kr = OPTrc == 8 ? k_swt_fwd_multi_rc<T, 8, 8> : OPTrc == 4 ? k_swt_fwd_multi_rc<T, 8, 4> : OPTrc == 2 ? k_swt_fwd_multi_rc<T, 8, 2> : k_swt_fwd_multi_rc<T, 8, 1>;
            hipLaunchKernelGGL(kr, dim3((unsigned)(((int64_t)1 << d) * (sdil / Rrc)), (unsigned)gy), dim3(NT),
                               (size_t)tile * sizeof(T), st, x, xw, (int)n, ncols, batch, L, d, K, Rrc,
                               (const double *)dcoef, (const int *)dshift, Uc);
            d += K;
            continue;
        }
        {   // the whole-column kernel reads the table tap-major: the 2^K coefficients of one offset are one contiguous scalar load
            const int NCc = 1 << K;
            std::vector<double> ct(coef.size());
            for (int c = 0; c < NCc; ++c) for (int t = 0; t < U; ++t) ct[(size_t)t * NCc + c] = coef[(size_t)c * U + t];
            coef.swap(ct);
        }
        dcoef = (double *)wx_const_upload(coef.data(), coef.size() * sizeof(double), st, true);
        dshift = (int *)wx_const_upload(shift.data(), shift.size() * sizeof(int), st, true);
        if (!dcoef || !dshift) return WX_EHIP;
        auto km = K == 2 ? k_swt_fwd_multi<T, 4> : k_swt_fwd_multi<T, 8>;
        if (lds > 64 * 1024)
            WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(km),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(km, dim3(1 << d, (unsigned)gy), dim3(nt), lds, st, x, xw, (int)n, ncols, batch, L, d, K,
                           (const double *)dcoef, (const int *)dshift, U);
        d += K;
    }
    WX_HIP_CHECK(hipGetLastError());
    if constexpr (sizeof(T) == 8) {
        if (haar6) return wx_haar_swpt6_fwd((double *)xw, n, L, batch, filt, st);
        if (deepLP) {
            if (dstop == 0)                                              // the root column of the table is the signal
                WX_HIP_CHECK(hipMemcpy2DAsync(xw, sizeof(T) * n * ncols, x, sizeof(T) * n, sizeof(T) * n, batch, hipMemcpyDeviceToDevice, st));
            return wx_swpt_deep_fwd((double *)xw, n, L, batch, filt, ac, layout == WX_LAYOUT_WPD, st);
        }
    }
    return WX_OK;
}

static int64_t wx_swtinv_lds_bytes()
{
    static int64_t v = -1;
    if (v < 0) { const char *e = wx_getenv("WX_SWTINV_LDS_KIB"); v = (e && atoi(e) > 0 && atoi(e) <= 64 ? atoi(e) : 32) * 1024; }
    return v;
}
static int wx_swtinv_threads()
{
    static int v = -1;
    if (v < 0) { const char *e = wx_getenv("WX_SWTINV_NT"); v = (e && (atoi(e) == 64 || atoi(e) == 128 || atoi(e) == 256)) ? atoi(e) : 512; }
    return v;
}
// inverse schedule (see WxSwtInvPlan): fused two-level (three for F <= 4) passes for the average-based full iswpt while the
// 2^K descendant tiles of >= 64-byte runs fit in 128 KiB of LDS, single levels otherwise
void wx_swt_inv_plan(int layout, int L, int F, int64_t sm, int64_t n, size_t esz, bool has_tree, WxSwtInvPlan *P,
                     bool haar6)
{
    P->npass = 0;
    P->need_cols[0] = P->need_cols[1] = 0;
    const bool fuse = layout == WX_LAYOUT_WPT && sm < 0 && !has_tree && F <= 20 && !wx_force_generic_swt();
    const int64_t budget = wx_swtinv_lds_bytes();
    int d = L, pp = 0;
    while (d > 0) {
        int K = 1, R = 0, OPT = 1;
        if (fuse && haar6 && d == L) { K = wx_haar_iswpt_levels(); R = 64; OPT = 0; }   // register pass of wx_haarswt.hip (OPT = 0 marks it)
        if (K == 1 && d == L && layout == WX_LAYOUT_WPT && sm < 0 && !has_tree && !haar6 && !wx_force_generic_swt()) {
            const int lp = wx_swpt_deep_levels(n, L, F, false, esz);   // lane-local pass of wx_swtdeep.hip (OPT = -1 marks it)
            if (lp) { K = lp; R = 64; OPT = -1; }
        }
        for (int Kt = (F <= 4 ? 3 : 2); fuse && Kt >= 2 && K == 1; --Kt) {
            if (d < Kt) continue;
            const int dp = d - Kt;
            const int64_t s = (int64_t)1 << dp, nu = n >> dp;
            int64_t r = (int64_t)(budget / esz) / nu;                  // one column tile = nu * R elements
            int64_t rp = 1; while (rp * 2 <= r) rp *= 2;
            if (r < 1) rp = 0;
            if (rp > s) rp = s;
            if (rp > 128) rp = 128;
            // rows per thread: the largest of 8/4/2/1 dividing nu that keeps >= 64 threads; at most 512 threads
            int o = 1;
            for (;; rp >>= 1) {
                // runs shorter than 64 bytes only when the tile is the whole column (every residue: contiguous)
                if (rp < 1 || (rp * (int64_t)esz < 64 && rp != s)) { rp = 0; break; }
                o = 1;
                for (int oo = 8; oo >= 2; oo >>= 1) if (nu % oo == 0 && nu * rp / oo >= 64) { o = oo; break; }
                while (nu * rp / o > 512 && o < 8 && nu % (o * 2) == 0) o *= 2;
                if (nu * rp / o <= 512) break;
            }
            if (rp) { K = Kt; R = (int)rp; OPT = o; }
        }
        const int i = P->npass++;
        P->from[i] = d; P->to[i] = d - K; P->R[i] = R; P->OPT[i] = OPT;
        if (d - K == 0) P->buf[i] = -1;
        else {
            P->buf[i] = pp;
            const int64_t cols = layout == WX_LAYOUT_DWT ? 1 : ((int64_t)1 << (d - K));
            if (cols > P->need_cols[pp]) P->need_cols[pp] = cols;
            pp ^= 1;
        }
        d -= K;
    }
}

// scratch: two level buffers sized by the plan; sm < 0 = average based
template <typename T>
int wx_dev_swt_inv(const T *xw, T *x, int64_t n, int L, int layout, int ncols, int64_t batch, int64_t sm,
                   const uint8_t *dtree, int64_t ntree, const WxFilt &filt, const WxSwtInvPlan &plan, T *scratch0,
                   T *scratch1, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    if (L == 0) {
        WX_HIP_CHECK(hipMemcpy2DAsync(x, n * sizeof(T), xw, (size_t)n * ncols * sizeof(T), n * sizeof(T), batch,
                                      hipMemcpyDeviceToDevice, st));
        return WX_OK;
    }
    if (layout == WX_LAYOUT_DWT && sm < 0 && L >= 2 && (size_t)3 * n * sizeof(T) <= 160 * 1024 && !wx_force_generic_swt()) {
        // average-based isdwt: every level in one kernel
        const size_t lds3 = (size_t)3 * n * sizeof(T);
        typedef void (*KI)(const T *, T *, int, int64_t, int, WxFilt, int);
        int per_cu = (int)((160 * 1024) / lds3);
        if (per_cu > 8) per_cu = 8;
        const int nt = n >= 4096 ? 1024 : (n >= 1024 ? 512 : 256);
        const bool pipe = per_cu == 1 && n <= (int64_t)(32 / sizeof(T)) * nt;
        KI ki = pipe ? k_isdwt_avg_fused<T, 0, true> : k_isdwt_avg_fused<T, 0, false>;
        switch (filt.F) {
#define WX_CASE(FF) case FF: ki = pipe ? k_isdwt_avg_fused<T, FF, true> : k_isdwt_avg_fused<T, FF, false>; break;
            WX_CASE(2) WX_CASE(4) WX_CASE(6) WX_CASE(8) WX_CASE(10) WX_CASE(12) WX_CASE(14) WX_CASE(16) WX_CASE(18) WX_CASE(20)
#undef WX_CASE
            default: break;
        }
        if (lds3 > 64 * 1024)
            WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ki),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        int64_t grid = (int64_t)256 * per_cu;
        if (grid > batch) grid = batch;
        hipLaunchKernelGGL(ki, dim3((unsigned)grid), dim3(nt), lds3, st, xw, x, (int)n, batch, L, filt, wx_sdwt_window_min_for(filt.F));
        WX_HIP_CHECK(hipGetLastError());
        return WX_OK;
    }
    if (layout == WX_LAYOUT_DWT && sm < 0 && L >= 2 && n == 1024 * (int64_t)(128 / sizeof(T)) &&
        !wx_force_generic_swt() && !(wx_getenv("WX_SDWT_INPLACE") && atoi(wx_getenv("WX_SDWT_INPLACE")) == 0)) {
        // the three columns of k_isdwt_avg_fused do not fit: reconstruction in place in one column of LDS, details from global
        constexpr int NPT = 128 / sizeof(T);
        auto ki = k_isdwt_avg_fused_ip<T, NPT>;
        const size_t lds1 = (size_t)n * sizeof(T);
        WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ki), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        int64_t grid = 256;
        if (grid > batch) grid = batch;
        hipLaunchKernelGGL(ki, dim3((unsigned)grid), dim3(1024), lds1, st, xw, x, (int)n, batch, L, filt);
        WX_HIP_CHECK(hipGetLastError());
        return WX_OK;
    }
    // main2depthshift (Utils.jl:297-305)
    int64_t sd[64];
    sd[0] = 0;
    if (sm >= 0) { int64_t acc = 0; for (int d = 0; d < L; ++d) { acc += ((sm >> d) & 1) << d; sd[d + 1] = acc; } }
    T *bufs[2] = {scratch0, scratch1};
    double *dcoef[16] = {nullptr};                       // padded tap tables per (K, rows-per-thread)
    int Utab[16] = {0};
    std::vector<int> omin[16];                           // per descendant: smallest offset of its adjoint taps
    T *prev = nullptr;                                   // buffer holding depth plan.from[i] (nullptr = xw)
    int64_t prev_cols = 0;
    for (int i = 0; i < plan.npass; ++i) {
        const int d = plan.to[i];
        const int K = plan.from[i] - d;
        const int nodes_d = layout == WX_LAYOUT_DWT ? 1 : (1 << d);
        T *outp = plan.buf[i] < 0 ? x : bufs[plan.buf[i]];
        const int64_t out_cols = plan.buf[i] < 0 ? 1 : nodes_d;
        if (plan.OPT[i] == -1) {
            if constexpr (sizeof(T) == 8) {
                const int rcd = wx_swpt_deep_inv((const double *)(prev ? prev : xw), prev ? prev_cols : ncols, (double *)outp, out_cols, n,
                                                 plan.from[i], K, batch, filt, st);
                if (rcd) return rcd;
            } else {
                return wx_set_error(WX_EHIP, "iswpt: the lane-local pass is Float64 only");
            }
        } else if (K >= 5 && plan.OPT[i] == 0) {
            if constexpr (sizeof(T) == 8) {
                const T *srcp6 = prev ? prev : xw;
                const int rc6 = wx_haar_iswpt6((const double *)srcp6, prev ? prev_cols : ncols, (double *)outp, out_cols, n,
                                               plan.from[i], batch, filt, st);
                if (rc6) return rc6;
            } else {
                return wx_set_error(WX_EHIP, "iswpt: the Haar register pass is Float64 only");
            }
        } else if (K >= 2) {
            const int R = plan.R[i];
            const int64_t nu = n >> d;
            const int64_t tile = nu * R;
            const int OPT = plan.OPT[i];
            const int NT = (int)(tile / OPT);
            const int slot = K * 4 + (OPT == 8 ? 3 : OPT == 4 ? 2 : OPT == 2 ? 1 : 0);
            if (!dcoef[slot]) {
                // adjoint of the forward composite: negated offsets (so reversed tap order), (1/2)^K gain,
                // zero padded by OPT-1 on both sides for the sliding window
                std::vector<double> coef;
                std::vector<int> offs;
                wx_swt_composite(filt, K, coef, offs);
                const int U = (int)offs.size();
                for (int t = 1; t < U; ++t)
                    if (offs[t] != offs[0] + t) return wx_set_error(WX_EHIP, "iswpt: composite taps are not contiguous");
                // every descendant uses its own contiguous run of the offsets: the window slides over that run only
                const int NCk = 1 << K;
                std::vector<int> first(NCk, 0), len(NCk, 1);
                int Uc = 1;
                for (int c = 0; c < NCk; ++c) {
                    int f = -1, l = -1;
                    for (int t = 0; t < U; ++t) if (coef[(size_t)c * U + (U - 1 - t)] != 0.0) { if (f < 0) f = t; l = t; }
                    if (f < 0) { f = 0; l = 0; }
                    first[c] = f; len[c] = l - f + 1;
                    if (len[c] > Uc) Uc = len[c];
                }
                const int UP = Uc + 2 * (OPT - 1);
                std::vector<double> pad((size_t)NCk * UP, 0.0);
                omin[slot].assign(NCk, 0);
                for (int c = 0; c < NCk; ++c) {
                    for (int t = 0; t < len[c]; ++t)
                        pad[(size_t)c * UP + (OPT - 1) + t] = coef[(size_t)c * U + (U - 1 - (first[c] + t))] * (K == 2 ? 0.25 : 0.125);
                    omin[slot][c] = -offs[U - 1 - first[c]];
                }
                dcoef[slot] = (double *)wx_const_upload(pad.data(), pad.size() * sizeof(double), st, true);
                if (!dcoef[slot]) return WX_EHIP;
                Utab[slot] = Uc;
            }
            std::vector<int> ust(omin[slot].size());
            for (size_t c = 0; c < ust.size(); ++c) {
                int64_t o = omin[slot][c] % nu;
                if (o < 0) o += nu;
                ust[c] = (int)o;
            }
            const int *dust = (const int *)wx_const_upload(ust.data(), ust.size() * sizeof(int), st, true);
            if (!dust) return WX_EHIP;
            const size_t lds = (size_t)2 * tile * sizeof(T);
            typedef void (*KM)(const T *, int64_t, T *, int64_t, int, int64_t, int, int, const double *, const int *, int);
            KM km = nullptr;
            if (K == 2) km = OPT == 8 ? k_swt_inv_multi<T, 4, 8> : OPT == 4 ? k_swt_inv_multi<T, 4, 4> : OPT == 2 ? k_swt_inv_multi<T, 4, 2> : k_swt_inv_multi<T, 4, 1>;
            else km = OPT == 8 ? k_swt_inv_multi<T, 8, 8> : OPT == 4 ? k_swt_inv_multi<T, 8, 4> : OPT == 2 ? k_swt_inv_multi<T, 8, 2> : k_swt_inv_multi<T, 8, 1>;
            if (lds > 64 * 1024)
                WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(km),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int64_t gy = batch;
            if (gy > 65535) gy = 65535;
            const T *srcp = prev ? prev : xw;
            const int64_t src_cols = prev ? prev_cols : ncols;
            hipLaunchKernelGGL(km, dim3((unsigned)(nodes_d * (((int64_t)1 << d) / R)), (unsigned)gy), dim3(NT), lds, st,
                               srcp, src_cols, outp, out_cols, (int)n, batch, d, R, (const double *)dcoef[slot],
                               dust, Utab[slot]);
        } else {
            WxInvDesc D;
            D.in = xw; D.ncols = ncols; D.layout = layout; D.L = L; D.d = d; D.tree = dtree; D.ntree = ntree;
            // the first pass reads the caller's table; wx_inv_child picks `in` when d == L-1, and for the
            // wpt layout after a fused pass the source is always a scratch buffer
            D.cur = prev; D.cur_cols = prev_cols;
            D.out = outp; D.out_cols = out_cols;
            const int sm_mode = sm >= 0 ? 1 : 0;
            const int64_t per_node = sm_mode ? (n >> d) : n;
            const int64_t total = batch * nodes_d * per_node;
            // average based, long columns: out of an LDS tile (k_swt_inv_level_tile) while the halo s (F - 1) of the dilated taps stays small
            // against the tile -- the top levels, which is where the per-sample kernel is slow
            static const bool tile_off = wx_getenv("WX_SWTINV_TILE") && atoi(wx_getenv("WX_SWTINV_TILE")) == 0;
            const int64_t halo = ((int64_t)1 << d) * (filt.F - 1);
            const int Ptile = 2048;
            if (!tile_off && sm_mode == 0 && n >= 2 * Ptile && n % Ptile == 0 && halo <= 512 && !wx_force_generic_swt()) {
                const size_t ldst = (size_t)2 * (Ptile + 2 * halo) * sizeof(T);
                const int64_t units = batch * nodes_d * (n / Ptile);
                const int64_t gt = units < 256 * 8 ? units : 256 * 8;
                typedef void (*KT)(WxInvDesc, int, int64_t, int, int, WxFilt);
                KT kt = nullptr;
                switch (filt.F / 2) {
#define WX_KT(h) case h: kt = k_swt_inv_level_tile<T, h>; break;
                    WX_KT(1) WX_KT(2) WX_KT(3) WX_KT(4) WX_KT(5) WX_KT(6) WX_KT(7) WX_KT(8) WX_KT(9) WX_KT(10)
#undef WX_KT
                }
                if (kt) {
                    if (ldst > 64 * 1024)
                        WX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst));
                    hipLaunchKernelGGL(kt, dim3((unsigned)gt), dim3(256), ldst, st, D, (int)n, batch, Ptile, (int)halo, filt);
                } else
                    hipLaunchKernelGGL(k_swt_inv_level<T>, dim3(wx_grid1(total)), dim3(256), 0, st, D, (int)n, batch, sm_mode, 0, 0, filt);
            } else
            hipLaunchKernelGGL(k_swt_inv_level<T>, dim3(wx_grid1(total)), dim3(256), 0, st, D, (int)n, batch, sm_mode,
                               sm >= 0 ? (int)sd[d] : 0, sm >= 0 ? (int)sd[d + 1] : 0, filt);
        }
        prev = plan.buf[i] < 0 ? nullptr : outp;
        prev_cols = out_cols;
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

template <typename T>
int wx_dev_iacdwt(const T *xw, T *x, int64_t n, int L, int64_t batch, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    hipLaunchKernelGGL(k_iacdwt<T>, dim3(wx_grid1(batch * n)), dim3(256), 0, st, xw, x, (int)n, L, batch);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template <typename T>
int wx_dev_iacwpt(const T *xw, T *x, int64_t n, int L, int64_t batch, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    hipLaunchKernelGGL(k_iacwpt<T>, dim3(wx_grid1(batch * n)), dim3(256), 0, st, xw, x, (int)n, L, batch);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template <typename T>
int wx_dev_iacwpd(const T *xw, T *x, int64_t n, int ncols, int64_t batch, const uint8_t *dtree, int64_t ntree,
                  int Lfull, hipStream_t st)
{
    if (batch == 0 || n == 0) return WX_OK;
    hipLaunchKernelGGL(k_iacwpd<T>, dim3(wx_grid1(batch * n)), dim3(256), 0, st, xw, x, (int)n, ncols, batch, dtree,
                       ntree, Lfull);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

#define WX_INST(T)                                                                                              \
    template int wx_dev_swt_fwd<T>(const T *, T *, int64_t, int, int, int64_t, const WxFilt &, const WxAcFilt *, \
                                   hipStream_t);                                                                \
    template int wx_dev_swt_inv<T>(const T *, T *, int64_t, int, int, int, int64_t, int64_t, const uint8_t *,    \
                                   int64_t, const WxFilt &, const WxSwtInvPlan &, T *, T *, hipStream_t);        \
    template int wx_dev_iacdwt<T>(const T *, T *, int64_t, int, int64_t, hipStream_t);                          \
    template int wx_dev_iacwpt<T>(const T *, T *, int64_t, int, int64_t, hipStream_t);                          \
    template int wx_dev_iacwpd<T>(const T *, T *, int64_t, int, int64_t, const uint8_t *, int64_t, int, hipStream_t);
WX_INST(double)
WX_INST(float)
