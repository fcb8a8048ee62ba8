// wx_denoise.hip -- denoising core on the device: SURVEY section 8(f) row 1 (threshold between the forward and
// the inverse transform; both transforms are the batch entry points of the other translation units).
//   noisest(x, redundant, tree)      Denoising.jl:214-232  = Wavelets.Threshold.mad!(dr) / 0.6745
//   threshold!(x, TH, t)             Wavelets.jl Threshold (HardTH / SoftTH / SemiSoftTH / SteinTH; SemiSoftTH is the
//                                    piecewise-linear rule with the upper knee at 2t), applied as
//                                    in denoise(), Denoising.jl:483-599, to the rows / columns it selects
// Wavelets.jl is not vendored in the reference tree: mad! and the four threshold loops are restated from its
// published source (parity unpinned, like the filter tables).
#include "../../include/waveletsext_hip.h"     // the definitions below must match the public prototypes
#include "wx_common.h"
#include "wx_host.h"
#include "wx_kernels.h"
#include "wx_select_count.h"
#include <vector>

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);

namespace {

// out of place: Y = X with the selected rows / columns thresholded (one read + one write of the table instead
// of a copy followed by an in-place pass); colflag[c] != 0 selects column c (nullptr = all)
template <typename T>
__global__ __launch_bounds__(256) void k_threshold_copy(const T *__restrict__ X, T *__restrict__ Y, int n, int k,
                                                        int64_t batch, int th_kind, const T *__restrict__ t,
                                                        int per_signal, int row_lo, const uint8_t *__restrict__ colflag,
                                                        double scale = 1.0)
{
    const int64_t per_sig = (int64_t)n * k;
    const int64_t total = per_sig * batch;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t sig = g / per_sig;
        const int64_t rem = g - sig * per_sig;
        const int c = (int)(rem / n), r = (int)(rem - (int64_t)c * n);
        const T v = X[g];
        const bool sel = r >= row_lo && (!colflag || colflag[c]);
        Y[g] = sel ? wx_thresh<T>(v, (T)((double)(per_signal ? t[sig] : t[0]) * scale), th_kind) : v;
    }
}

// X (n, k, batch): rows [row_lo, n) of the selected columns of every signal, in place
template <typename T>
__global__ __launch_bounds__(256) void k_threshold(T *__restrict__ X, int n, int k, int64_t batch, int th_kind,
                                                   const T *__restrict__ t, int per_signal, int row_lo,
                                                   const int *__restrict__ cols, int ncols)
{
    const int rows = n - row_lo;
    const int64_t per_sig = (int64_t)rows * ncols;
    const int64_t total = per_sig * batch;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t sig = g / per_sig;
        const int64_t rem = g - sig * per_sig;
        const int c = (int)(rem / rows), r = (int)(rem - (int64_t)c * rows) + row_lo;
        T *p = X + (sig * k + (cols ? cols[c] : c)) * (int64_t)n + r;
        const T v = *p, tt = per_signal ? t[sig] : t[0];
        const T out = wx_thresh<T>(v, tt, th_kind);
        if (out != v || th_kind != 0) *p = out;
    }
}

// ---- exact order statistics in LDS (256 threads) -------------------------------------------------
// Value bucketing instead of a sort: with the exact min / max of the candidates, bucket(x) = floor((x - lo) * scale)
// is monotone, so the k-th smallest lies in the bucket where the running count crosses k.  One histogram pass
// (1024 buckets: coefficients spread over them, so the LDS atomics do not pile up on one word) leaves a handful
// of candidates, which are ranked directly.  Degenerate data (everything in one bucket) narrows [lo, hi] to that
// bucket's own min / max and repeats; equal candidates end the search.  No arithmetic on the result: exact.
constexpr int WX_NB = 1024;       // buckets
constexpr int WX_LST = 256;       // candidates ranked directly

template <typename T> __device__ __forceinline__ T wx_wave_min(T v)
{
    for (int o = 32; o > 0; o >>= 1) { const T u = __shfl_xor(v, o, 64); v = u < v ? u : v; }
    return v;
}
template <typename T> __device__ __forceinline__ T wx_wave_max(T v)
{
    for (int o = 32; o > 0; o >>= 1) { const T u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
}
struct WxSelScratch {
    unsigned int hist[WX_NB];
    double lst[WX_LST];
    double red[16];
    int ired[8];
};

// block-wide min and max of f(i) over i in [0, cnt) restricted by pred; every thread gets the result
template <typename T, typename P>
__device__ void wx_block_minmax(const T *v, int cnt, P pred, WxSelScratch *S, T &mn, T &mx)
{
    T a = (T)INFINITY, b = (T)-INFINITY;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) { const T x = v[i]; if (pred(x)) { a = x < a ? x : a; b = x > b ? x : b; } }
    a = wx_wave_min(a); b = wx_wave_max(b);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { S->red[w] = (double)a; S->red[4 + w] = (double)b; }
    __syncthreads();
    double m0 = S->red[0], m1 = S->red[4];
    for (int j = 1; j < (int)(blockDim.x >> 6); ++j) { m0 = S->red[j] < m0 ? S->red[j] : m0; m1 = S->red[4 + j] > m1 ? S->red[4 + j] : m1; }
    mn = (T)m0; mx = (T)m1;
}

// k-th smallest (0-based) of v[0..cnt); all threads return the same value
template <typename T>
__device__ T wx_select_kth(const T *v, int cnt, int k, WxSelScratch *S)
{
    T lo, hi;
    wx_block_minmax(v, cnt, [](T) { return true; }, S, lo, hi);
    int kk = k;                                       // rank among the candidates lo <= x <= hi
    for (;;) {
        if (!(lo < hi)) return lo;                    // all candidates equal (or NaN-degenerate)
        const double scale = (double)WX_NB / ((double)hi - (double)lo);
        auto bucket = [&](T x) { int b = (int)(((double)x - (double)lo) * scale); return b < WX_NB ? b : WX_NB - 1; };
        for (int i = threadIdx.x; i < WX_NB; i += blockDim.x) S->hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const T x = v[i];
            if (x >= lo && x <= hi) atomicAdd(&S->hist[bucket(x)], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) S->ired[2] = -1;
        __syncthreads();
        if (threadIdx.x < 64) {
            // lane l owns buckets 16l .. 16l+15: wave-wide scan of the lane sums, then the crossing bucket
            unsigned sum = 0;
            for (int j = 0; j < WX_NB / 64; ++j) sum += S->hist[(WX_NB / 64) * threadIdx.x + j];
            unsigned incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o, 64); if ((int)threadIdx.x >= o) incl += up; }
            const unsigned excl = incl - sum;
            if ((unsigned)kk >= excl && (unsigned)kk < incl) {
                unsigned c = excl;
                int b = (WX_NB / 64) * threadIdx.x;
                const int bend = b + WX_NB / 64 - 1;
                while (b < bend && (unsigned)kk >= c + S->hist[b]) { c += S->hist[b]; ++b; }
                S->ired[0] = b; S->ired[1] = (int)c; S->ired[2] = (int)S->hist[b];
            }
        }
        __syncthreads();
        const int bsel = S->ired[0], before = S->ired[1], cb = S->ired[2];
        if (cb < 0) return (T)NAN;                     // rank beyond the comparable elements (NaNs in the data)
        if (cb <= WX_LST) {
            // gather the bucket's candidates and rank them (ties by slot): the (kk - before)-th is the answer
            if (threadIdx.x == 0) S->ired[3] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
                const T x = v[i];
                if (x >= lo && x <= hi && bucket(x) == bsel) S->lst[atomicAdd(&S->ired[3], 1)] = (double)x;
            }
            __syncthreads();
            const int want = kk - before;
            if ((int)threadIdx.x < cb) {
                const double x = S->lst[threadIdx.x];
                int rank = 0;
                for (int j = 0; j < cb; ++j) { const double y = S->lst[j]; rank += (y < x) || (y == x && j < (int)threadIdx.x); }
                if (rank == want) S->red[8] = x;
            }
            __syncthreads();
            const T r = (T)S->red[8];
            __syncthreads();
            return r;
        }
        // too many candidates in one bucket: narrow to that bucket's own range and repeat
        T nlo, nhi;
        const T clo = lo, chi = hi;
        wx_block_minmax(v, cnt, [&](T x) { return x >= clo && x <= chi && bucket(x) == bsel; }, S, nlo, nhi);
        lo = nlo; hi = nhi; kk -= before;
    }
}

// median as Statistics.median!: middle element, or middle(a, b) = a/2 + b/2 of the two middle ones
template <typename T> __device__ T wx_median_lds(const T *v, int cnt, WxSelScratch *S)
{
    const T a = wx_select_kth<T>(v, cnt, (cnt - 1) / 2, S);
    if (cnt & 1) return a;
    // the next order statistic: a again if at least (cnt/2 + 1) elements are <= a, else the smallest element > a
    int le = 0;
    T nx = (T)INFINITY;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) { const T x = v[i]; le += x <= a; if (x > a && x < nx) nx = x; }
    nx = wx_wave_min(nx);
    for (int o = 32; o > 0; o >>= 1) le += __shfl_xor(le, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { S->red[w] = (double)nx; S->ired[4 + w] = le; }
    __syncthreads();
    double m = S->red[0];
    int tot = S->ired[4];
    for (int j = 1; j < (int)(blockDim.x >> 6); ++j) { m = S->red[j] < m ? S->red[j] : m; tot += S->ired[4 + j]; }
    __syncthreads();
    const T b = tot >= cnt / 2 + 1 ? a : (T)m;
    return (T)((T)(a / (T)2) + (T)(b / (T)2));
}

// one workgroup per signal: median, absolute deviations, median again -- exact order statistics, so the result
// equals the reference's partial sorts bit for bit
template <typename T>
__global__ __launch_bounds__(256) void k_mad(const T *__restrict__ X, int64_t sig_stride, int64_t off, int cnt,
                                             T *__restrict__ sigma)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    T *v = reinterpret_cast<T *>(wx_smem);
    __shared__ WxSelScratch S;
    const T *x = X + (int64_t)blockIdx.x * sig_stride + off;
    wx_stage<T>(v, x, cnt);
    __syncthreads();
    // a NaN among the values: Statistics.median gives NaN; the selection's comparisons would just lose it
    int nan = 0;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) nan |= v[i] != v[i];
    nan = __syncthreads_or(nan);
    const T m = wx_median_lds<T>(v, cnt, &S);
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) v[i] = (T)fabs((double)(T)(v[i] - m));
    __syncthreads();
    const T r = wx_median_lds<T>(v, cnt, &S);
    if (threadIdx.x == 0) sigma[blockIdx.x] = nan ? (T)__builtin_nan("") : (T)(r / (T)0.6745);
}

// short detail ranges (at most 512 coefficients: signals up to 1024 samples at the finest level): ONE WAVEFRONT per signal, four per
// workgroup.  The values sit in a private LDS row; every lane ranks its elements by counting (x_j < x_i, ties by index -- broadcast reads),
// so the ranks are a permutation and the order statistics k and k + 1 are simply the elements of those ranks: exact, like the selection of
// k_mad.  (One workgroup per signal took 27.9 ms per GiB of 64-sample signals: 2 M workgroups for 32 values each.)
template <typename T, int E>
__global__ __launch_bounds__(256) void k_mad_wave(const T *__restrict__ X, int64_t sig_stride, int64_t off, int cnt, int64_t batch,
                                                  T *__restrict__ sigma)
{
    __shared__ T rows[4][512 + 2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t sig = (int64_t)blockIdx.x * 4 + wave;
    if (sig >= batch) return;
    T *v = rows[wave];
    const T *x = X + sig * sig_stride + off;
    T e[E];                                                  // E = ceil(cnt / 64) elements per lane
#pragma unroll
    for (int u = 0; u < E; ++u) { const int i = lane + 64 * u; e[u] = i < cnt ? x[i] : (T)0; }
    bool nan = false;
#pragma unroll
    for (int u = 0; u < E; ++u) nan = nan || e[u] != e[u];
    const bool any_nan = __builtin_amdgcn_ballot_w64(nan) != 0;          // every comparison with a NaN is false: its rank would be 0, the others' ranks ignore it
    const int k0 = (cnt - 1) / 2;
    T med = (T)0;
    for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int u = 0; u < E; ++u) { const int i = lane + 64 * u; if (i < cnt) v[i] = e[u]; }
        // a NaN among the values makes every comparison false: the ranks are no permutation then and ranks k0, k0 + 1 may not occur --
        // the two slots start as NaN, so the estimate is NaN like Statistics.median's instead of a stale LDS word
        if (lane == 0) { v[512] = (T)__builtin_nan(""); v[513] = (T)__builtin_nan(""); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int rk[E];
#pragma unroll
        for (int u = 0; u < E; ++u) rk[u] = 0;
        for (int j = 0; j < cnt; ++j) {
            const T vj = v[j];                                   // the same address for every lane: a broadcast read
#pragma unroll
            for (int u = 0; u < E; ++u) { const int i = lane + 64 * u; rk[u] += (vj < e[u]) || (vj == e[u] && j < i); }
        }
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const int i = lane + 64 * u;
            if (i < cnt && rk[u] == k0) v[512] = e[u];
            if (i < cnt && rk[u] == k0 + 1) v[513] = e[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const T a = v[512];
        const T b = (cnt & 1) ? a : v[513];
        med = (cnt & 1) ? a : (T)((T)(a / (T)2) + (T)(b / (T)2));   // Statistics.median!: middle(a, b) = a/2 + b/2
        __builtin_amdgcn_wave_barrier();
        if (round == 0) {
#pragma unroll
            for (int u = 0; u < E; ++u) e[u] = (T)fabs((double)(T)(e[u] - med));
        }
    }
    if (lane == 0) sigma[sig] = any_nan ? (T)__builtin_nan("") : (T)(med / (T)0.6745);
}

// 65 ... 512 coefficients (signals of 256 ... 1024 samples at the finest level; built up to 2048), round 6: ONE WAVEFRONT SORTS the values in its registers.
// Element i of the P = 64 E slots (E a power of two; slots beyond cnt hold +Inf) sits in lane i / E, register i % E.  A bitonic sorting
// network in its "every comparison ascending" form (the first stage of the merge of 2 m elements pairs i with i ^ (2 m - 1), the later
// stages i with i ^ j): a partner inside the lane is a register pair (v_min / v_max, direction known at compile time), a partner in another
// lane comes through the cross-lane network (__shfl: ds_bpermute, no memory) and the lane keeps the minimum or the maximum by the side it is
// on.  The median is the element(s) of rank k0 (and k0 + 1) -- one v_readlane each; the absolute deviations of a SORTED sequence fall to the
// median and rise again, i.e. they are bitonic, so the second median needs one merge (log2 P stages), not a second sort.  Exact like the
// selection it replaces: same order statistics, same middle(a, b) = a/2 + b/2 (Statistics.median!), same rounded subtraction.  NaN anywhere
// gives NaN (min / max would drop it).  (k_mad: one 256-thread workgroup per signal -- 2 M workgroups for 256 values each at n = 512: 3.7 ms
// per GiB of signals; k_mad_wave's counting is quadratic: 2.1 ms at n = 256.  This kernel: see profiles/r06_floor_misc.txt.)
template <typename T> __device__ __forceinline__ T mad_shfl(T v, int src_lane)
{
    if constexpr (sizeof(T) == 8) {
        const double d = (double)v;
        const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(d)), hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(d));
        return (T)__hiloint2double(hi, lo);
    } else {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, (float)v)));
    }
}
template <typename T> __device__ __forceinline__ T mad_min(T a, T b) { return a < b ? a : b; }      // no NaN reaches the network
template <typename T> __device__ __forceinline__ T mad_max(T a, T b) { return a < b ? b : a; }
// one stage: partner of slot i is i ^ X (X = 2 m - 1 for the first stage of a merge, a single bit j afterwards)
template <typename T, int E, int X> __device__ __forceinline__ void mad_stage(T (&r)[E], int lane)
{
    if constexpr (X < E) {
        // inside the lane: registers u and u ^ X, the lower index keeps the minimum
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const int w = u ^ X;
            if (w > u) { const T a = r[u], b = r[w]; r[u] = mad_min(a, b); r[w] = mad_max(a, b); }
        }
    } else {
        // across lanes: lane ^ (X / E) holds the partner in register u ^ (X % E)  (X % E is 0 or E - 1)
        constexpr int XL = X / E, XR = X % E;
        const int pl = lane ^ XL;
        const bool low = lane < pl;
        T o[E];
#pragma unroll
        for (int u = 0; u < E; ++u) o[u] = mad_shfl<T>(r[u ^ XR], pl);
#pragma unroll
        for (int u = 0; u < E; ++u) { const T a = r[u], b = o[u]; r[u] = low ? mad_min(a, b) : mad_max(a, b); }
    }
}
// merge of sorted (or bitonic) runs into runs of M slots: FIRST = the runs are two sorted halves (mirror pairing), else bitonic (plain pairing)
template <typename T, int E, int M, bool FIRST> __device__ __forceinline__ void mad_merge(T (&r)[E], int lane)
{
    if constexpr (M >= 2) {
        mad_stage<T, E, FIRST ? M - 1 : M / 2>(r, lane);
        mad_merge<T, E, M / 2, false>(r, lane);
    }
}
template <typename T, int E, int M> __device__ __forceinline__ void mad_sort(T (&r)[E], int lane)
{
    if constexpr (M >= 2) {
        mad_sort<T, E, M / 2>(r, lane);
        mad_merge<T, E, M, true>(r, lane);
    }
}
// the element of slot idx (the same register index for every lane; the lane part may differ from group to group)
template <typename T, int E> __device__ __forceinline__ T mad_pick(const T (&r)[E], int idx)
{
    const int u = idx % E, l = idx / E;
    T v = r[0];
#pragma unroll
    for (int q = 1; q < E; ++q) v = u == q ? r[q] : v;
    return mad_shfl<T>(v, l);
}
// PL = lanes per signal: 64 (E registers each), or 32 / 16 / 8 / 4 with E = 1 -- 2 ... 16 SHORT signals side by side in one wavefront (every
// partner of the network is lane ^ something below PL, so the groups never meet): 64-sample signals, 32 details each, took one wavefront per
// signal in k_mad_wave (1.3 ms per GiB of signals: 16 M wavefronts)
template <typename T, int E, int PL = 64>
__global__ __launch_bounds__(256) void k_mad_sort(const T *__restrict__ X, int64_t sig_stride, int64_t off, int cnt, int64_t batch,
                                                  T *__restrict__ sigma)
{
    static_assert(PL == 64 || E == 1, "several signals per wavefront: one slot per lane");
    constexpr int P = PL * E, SPW = 64 / PL;
    const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63, lane = wl & (PL - 1), sub = wl / PL;
    int64_t sig = ((int64_t)blockIdx.x * 4 + wave) * SPW + sub;
    const bool live = sig < batch;
    if (SPW == 1 && !live) return;
    if (!live) sig = batch - 1;                                       // the lanes of a group beyond the batch keep the wavefront's network company
    const T *x = X + sig * sig_stride + off;
    const T inf = (T)__builtin_inf();
    T r[E];
    bool nan = false;
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int i = lane * E + u;
        r[u] = i < cnt ? x[i] : inf;
        nan = nan || r[u] != r[u];
    }
    // NaN anywhere in the SIGNAL: the ballot restricted to the signal's lanes
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(nan);
    const bool any_nan = PL == 64 ? bal != 0 : ((bal >> (sub * PL)) & ((1ull << (PL & 63)) - 1ull)) != 0;
    mad_sort<T, E, P>(r, wl);
    const int k0 = (cnt - 1) / 2;
    const int base = sub * PL * E;                                    // slot 0 of this lane's signal
    const T a = mad_pick<T, E>(r, base + k0), b = mad_pick<T, E>(r, base + k0 + ((cnt & 1) ? 0 : 1));
    const T med = (cnt & 1) ? a : (T)((T)(a / (T)2) + (T)(b / (T)2));
    // |v - med| of the sorted values: falls, then rises (the slots beyond cnt stay +Inf at the end): one bitonic merge sorts it
#pragma unroll
    for (int u = 0; u < E; ++u) r[u] = (T)fabs((double)(T)(r[u] - med));
    mad_merge<T, E, P, false>(r, wl);
    const T a2 = mad_pick<T, E>(r, base + k0), b2 = mad_pick<T, E>(r, base + k0 + ((cnt & 1) ? 0 : 1));
    const T mad = (cnt & 1) ? a2 : (T)((T)(a2 / (T)2) + (T)(b2 / (T)2));
    if (lane == 0 && live) sigma[sig] = any_nan ? (T)__builtin_nan("") : (T)(mad / (T)0.6745);
}

// 256 ... 4096 coefficients (signals of 512 ... 8192 samples at the finest level), round 6: ONE WAVEFRONT per signal keeps the values in NR = cnt / 64
// registers per lane and finds the two medians by COUNTING against a pivot (wx_select_count.h: ballot + population count per register, the
// bracket halved in value) -- no LDS, no sort.  Per GiB of signals in denoiseall(:dwt), Float64: 256 / 512 / 1024 / 2048 / 4096 coefficients
// 0.63 / 0.43 / 0.33 / 0.30 / 0.46 ms (k_mad_sort: 0.69 / 0.71; the workgroup selection k_mad: 0.84 ms at 2048).
template <typename T, int NR>
__global__ __launch_bounds__(256) void k_mad_count(const T *__restrict__ X, int64_t sig_stride, int64_t off, int64_t batch, T *__restrict__ sigma)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t sig = (int64_t)blockIdx.x * 4 + wave;
    if (sig >= batch) return;
    const T *x = X + sig * sig_stride + off;
    double e[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) e[u] = (double)x[lane + 64 * u];
    double sg[1];
    dn_noisest<-1, 64, 1, NR, T>(e, sg);
    if (lane == 0) sigma[sig] = (T)sg[0];
}

// 32 ... 1024 coefficients: FOUR signals per wavefront, each a row of 16 lanes with NR = cnt / 16 registers per lane (per-lane counters, four DPP
// adds per pass) -- the pivot arithmetic of a pass, which outweighs the counting at these sizes, is shared by the four signals.  Per GiB of signals
// in denoiseall(:dwt), Float64: 0.47 / 0.37 / 0.32 ms at 128 / 256 / 512 coefficients (one signal per wavefront: 0.63 / 0.43 at 256 / 512; sort: 0.67 at 128).
template <typename T, int NR>
__global__ __launch_bounds__(256) void k_mad_count_rows(const T *__restrict__ X, int64_t sig_stride, int64_t off, int64_t batch, T *__restrict__ sigma)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, li = lane & 15;
    const int64_t sig0 = ((int64_t)blockIdx.x * 4 + wave) * 4;
    if (sig0 >= batch) return;
    int64_t sig = sig0 + row;
    const bool live = sig < batch;
    if (!live) sig = batch - 1;                                      // the rows beyond the batch keep the wavefront's loops company
    const T *x = X + sig * sig_stride + off;
    double e[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) e[u] = (double)x[li + 16 * u];
    double sg[1];
    dn_noisest<-1, 16, 1, NR, T>(e, sg);
    if (li == 0 && live) sigma[sig] = (T)sg[0];
}

// 8192 ... 32768 coefficients (signals of 16384 ... 65536 samples at the finest level): the same counting with a WORKGROUP of BW wavefronts per
// signal -- every wavefront counts its NR registers per lane, the block sums go through a BW-entry LDS table (two barriers per pass); every
// wavefront runs the same pivot arithmetic on the same sums.  Per GiB of signals in denoiseall(:dwt): 8192 / 16384 / 32768 coefficients took the
// LDS histogram selection k_mad 0.73 / 1.23 ms and k_mad_g (values in global scratch) 1.69 ms; this kernel: profiles/r06_denoise_onepass.md section 5.
template <typename T, int NR, int BW>
__global__ __launch_bounds__(64 * BW) void k_mad_count_wg(const T *__restrict__ X, int64_t sig_stride, int64_t off, T *__restrict__ sigma)
{
    const int64_t sig = blockIdx.x;
    const T *x = X + sig * sig_stride + off;
    double e[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) e[u] = (double)x[threadIdx.x + 64 * BW * u];
    double sg[1];
    dn_noisest<-1, 64, 1, NR, T, BW>(e, sg);
    if (threadIdx.x == 0) sigma[sig] = (T)sg[0];
}

// the same on detail ranges that do not fit a CU's LDS (signals of more than 32768 Float64 / 65536 Float32 samples' worth of details):
// the copy that the second median overwrites lives in a global scratch row instead; the selection (wx_select_kth: counting passes
// over the values) reads it through L2
template <typename T>
__global__ __launch_bounds__(256) void k_mad_g(const T *__restrict__ X, int64_t sig_stride, int64_t off, int cnt, T *__restrict__ work,
                                               T *__restrict__ sigma)
{
    __shared__ WxSelScratch S;
    const T *x = X + (int64_t)blockIdx.x * sig_stride + off;
    T *v = work + (int64_t)blockIdx.x * cnt;
    int nan = 0;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) nan |= x[i] != x[i];
    nan = __syncthreads_or(nan);
    const T m = wx_median_lds<T>(x, cnt, &S);
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) v[i] = (T)fabs((double)(T)(x[i] - m));
    __threadfence_block();
    __syncthreads();
    const T r = wx_median_lds<T>(v, cnt, &S);
    if (threadIdx.x == 0) sigma[blockIdx.x] = nan ? (T)__builtin_nan("") : (T)(r / (T)0.6745);
}

}  // namespace

// Y = X with rows [row_lo, n) thresholded (k = 1): used by the thresholding inverse when its kernel cannot take the
// threshold in its own load stage
template <typename T>
int wx_dev_threshold_copy(const T *X, T *Y, int64_t n, int64_t batch, int th_kind, const T *t, int per_signal, int64_t row_lo,
                          double scale, hipStream_t st)
{
    const int64_t total = n * batch;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(k_threshold_copy<T>, dim3((unsigned)grid), dim3(256), 0, st, X, Y, (int)n, 1, batch, th_kind, t, per_signal,
                       (int)row_lo, (const uint8_t *)nullptr, scale);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}
template int wx_dev_threshold_copy<double>(const double *, double *, int64_t, int64_t, int, const double *, int, int64_t, double, hipStream_t);
template int wx_dev_threshold_copy<float>(const float *, float *, int64_t, int64_t, int, const float *, int, int64_t, double, hipStream_t);

namespace {

int need_device()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

template <typename T>
int api_noisest(const T *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, T *sigma, void *stream)
{
    WX_REQUIRE(n >= 1 && k >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    WX_REQUIRE(wx_isdyadic(n), WX_EASSERT, "@assert isdyadic(size(x,1)) (Denoising.jl:218)");
    WX_REQUIRE(0 <= row_lo && row_lo < n && 0 <= col && col < k, WX_EBOUNDS, "detail range outside the array");
    const int64_t cnt = n - row_lo;
    WX_REQUIRE(cnt < ((int64_t)1 << 30), WX_EUNSUPPORTED, "noisest: 2^30 detail coefficients or more per signal");
    int rc;
    if ((rc = need_device())) return rc;
    if (batch == 0) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxIO io(st);
    const T *dX = (const T *)io.in(X, sizeof(T) * n * k * batch);
    T *ds = (T *)io.out(sigma, sizeof(T) * batch);
    if (!dX || !ds) return io.finish(WX_EHIP);
    static const int mad_count_wg = wx_getenv("WX_MAD_COUNT_WG") ? atoi(wx_getenv("WX_MAD_COUNT_WG")) : 1;
    if (mad_count_wg && (cnt == 8192 || cnt == 16384 || cnt == 32768) && batch <= 0x7ffffff0) {
        const dim3 g((unsigned)batch);
        if (cnt == 8192) hipLaunchKernelGGL((k_mad_count_wg<T, 32, 4>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, ds);
        else if (cnt == 16384) hipLaunchKernelGGL((k_mad_count_wg<T, 64, 4>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, ds);
        else hipLaunchKernelGGL((k_mad_count_wg<T, 64, 8>), g, dim3(512), 0, st, dX, n * k, col * n + row_lo, ds);
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "noisest kernel failed to launch"));
        return io.finish(WX_OK);
    }
    const size_t lds = (size_t)cnt * sizeof(T);
    if (lds > 128 * 1024) {
        // long signals (the reference has no limit: Denoising.jl:214-232): absolute deviations in a global scratch row per signal,
        // at most 1 GiB of them at a time
        WxScratch scr(st);
        int64_t per = ((int64_t)1 << 30) / (int64_t)lds;
        if (per < 1) per = 1;
        if (per > batch) per = batch;
        T *work = (T *)scr.alloc((size_t)per * lds);
        if (!work) return io.finish(WX_EHIP);
        for (int64_t b0 = 0; b0 < batch; b0 += per) {
            const int64_t nb = batch - b0 < per ? batch - b0 : per;
            hipLaunchKernelGGL(k_mad_g<T>, dim3((unsigned)nb), dim3(256), 0, st, dX + b0 * n * k, n * k, col * n + row_lo, (int)cnt, work, ds + b0);
        }
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "noisest kernel failed to launch"));
        return io.finish(WX_OK);
    }
    // 65 ... 2048 coefficients: one wavefront sorts the values in its registers (k_mad_sort); WX_MAD_SORT_MAX (knob) lowers the limit
    // (measured per GiB of signals, denoiseall: n = 256 / 512 / 1024 -- 128 / 256 / 512 coefficients -- 3.13 / 4.86 / 3.24 -> 1.83 / 1.88 / 1.83 ms;
    // 1024 / 2048 coefficients lose to the workgroup selection: 3.5 against 2.3, 4.3 against 1.8 ms -- the cross-lane stages grow with E)
    static const int mad_sort_max = wx_getenv("WX_MAD_SORT_MAX") ? atoi(wx_getenv("WX_MAD_SORT_MAX")) : 256;
    // 128 ... 512 coefficients (knobs WX_MAD_ROWS_MIN / WX_MAD_ROWS_MAX; built for 32 ... 1024), a power of two: four signals per wavefront
    // (k_mad_count_rows).  Per GiB of Float64 signals: 32 / 64 coefficients 1.03 / 0.67 ms against the sorting kernel's 0.55 / 0.69; 128 / 256 / 512
    // 0.47 / 0.37 / 0.32 against 0.67 (sort) / 0.63 / 0.43 (one signal per wavefront); 1024 0.33 against 0.32
    static const int mad_rows_max = wx_getenv("WX_MAD_ROWS_MAX") ? atoi(wx_getenv("WX_MAD_ROWS_MAX")) : 512;
    static const int mad_rows_min = wx_getenv("WX_MAD_ROWS_MIN") ? atoi(wx_getenv("WX_MAD_ROWS_MIN")) : 128;
    if (cnt >= 32 && cnt >= mad_rows_min && cnt <= 1024 && cnt <= mad_rows_max && (cnt & (cnt - 1)) == 0 && batch <= 0x7ffffff0) {
        const dim3 g((unsigned)((batch + 15) / 16));
#define WX_MR(NRR) hipLaunchKernelGGL((k_mad_count_rows<T, NRR>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, batch, ds)
        switch (cnt) {
        case 32: WX_MR(2); break;
        case 64: WX_MR(4); break;
        case 128: WX_MR(8); break;
        case 256: WX_MR(16); break;
        case 512: WX_MR(32); break;
        default: WX_MR(64); break;
        }
#undef WX_MR
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "noisest kernel failed to launch"));
        return io.finish(WX_OK);
    }
    // 256 ... 4096 coefficients (a power of two): one wavefront per signal counts against pivots (k_mad_count); WX_MAD_COUNT_MIN (knob) moves the lower limit
    static const int mad_count_min = wx_getenv("WX_MAD_COUNT_MIN") ? atoi(wx_getenv("WX_MAD_COUNT_MIN")) : 256;
    if (cnt >= 256 && cnt >= mad_count_min && cnt <= 4096 && (cnt & (cnt - 1)) == 0 && batch <= 0x7ffffff0) {
        const dim3 g((unsigned)((batch + 3) / 4));
#define WX_MC(NRR) hipLaunchKernelGGL((k_mad_count<T, NRR>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, batch, ds)
        switch (cnt) {
        case 256: WX_MC(4); break;
        case 512: WX_MC(8); break;
        case 1024: WX_MC(16); break;
        case 2048: WX_MC(32); break;
        default: WX_MC(64); break;
        }
#undef WX_MC
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "noisest kernel failed to launch"));
        return io.finish(WX_OK);
    }
    if (cnt >= 3 && cnt <= 2048 && cnt <= mad_sort_max && batch <= 0x7ffffff0) {
        dim3 g((unsigned)((batch + 3) / 4));
#define WX_MS(EE) hipLaunchKernelGGL((k_mad_sort<T, EE>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, (int)cnt, batch, ds)
#define WX_MSS(PLL) do { g = dim3((unsigned)((batch + 4 * (64 / PLL) - 1) / (4 * (64 / PLL)))); hipLaunchKernelGGL((k_mad_sort<T, 1, PLL>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, (int)cnt, batch, ds); } while (0)
        if (cnt <= 4) WX_MSS(4);
        else if (cnt <= 8) WX_MSS(8);
        else if (cnt <= 16) WX_MSS(16);
        else if (cnt <= 32) WX_MSS(32);
        else if (cnt <= 64) WX_MS(1);
        else if (cnt <= 128) WX_MS(2);
        else if (cnt <= 256) WX_MS(4);
        else if (cnt <= 512) WX_MS(8);
        else if (cnt <= 1024) WX_MS(16);
        else WX_MS(32);
#undef WX_MS
#undef WX_MSS
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "noisest kernel failed to launch"));
        return io.finish(WX_OK);
    }
    static const int mad_wave_max = wx_getenv("WX_MAD_WAVE_MAX") ? atoi(wx_getenv("WX_MAD_WAVE_MAX")) : 128;   // 512 coefficients: 8.7 against 3.2 ms (the counting is quadratic)
    if (cnt <= 512 && cnt <= mad_wave_max && batch <= 0x7ffffff0) {
        const dim3 g((unsigned)((batch + 3) / 4));
        if (cnt <= 64) hipLaunchKernelGGL((k_mad_wave<T, 1>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, (int)cnt, batch, ds);
        else if (cnt <= 128) hipLaunchKernelGGL((k_mad_wave<T, 2>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, (int)cnt, batch, ds);
        else if (cnt <= 256) hipLaunchKernelGGL((k_mad_wave<T, 4>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, (int)cnt, batch, ds);
        else hipLaunchKernelGGL((k_mad_wave<T, 8>), g, dim3(256), 0, st, dX, n * k, col * n + row_lo, (int)cnt, batch, ds);
        if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "noisest kernel failed to launch"));
        return io.finish(WX_OK);
    }
    if (lds > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mad<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "hipFuncSetAttribute(LDS)"));
    }
    hipLaunchKernelGGL(k_mad<T>, dim3((unsigned)batch), dim3(256), lds, st, dX, n * k, col * n + row_lo, (int)cnt, ds);
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "noisest kernel failed to launch"));
    return io.finish(WX_OK);
}

template <typename T>
int api_threshold(const T *X, T *Y, int64_t n, int64_t k, int64_t batch, int th_kind, const T *t, int64_t nt,
                  int64_t row_lo, const uint8_t *colmask, void *stream)
{
    WX_REQUIRE(X != nullptr && Y != nullptr, WX_EARG, "NULL array");
    WX_REQUIRE(n >= 1 && k >= 1 && batch >= 0, WX_EARG, "bad dimensions");
    WX_REQUIRE(th_kind >= 0 && th_kind <= 3, WX_EARG, "th_kind: 0 HardTH, 1 SoftTH, 2 SemiSoftTH, 3 SteinTH");
    WX_REQUIRE(t != nullptr && (nt == 1 || nt == batch), WX_EARG, "one threshold, or one per signal");
    WX_REQUIRE(0 <= row_lo && row_lo <= n, WX_EBOUNDS, "row range outside the array");
    WX_REQUIRE(n < ((int64_t)1 << 31) && k < ((int64_t)1 << 31), WX_EUNSUPPORTED, "array too large");
    int rc;
    if ((rc = need_device())) return rc;
    const bool inplace = (const T *)Y == X;
    std::vector<int> cols;
    if (colmask) for (int64_t c = 0; c < k; ++c) if (colmask[c]) cols.push_back((int)c);
    const int ncols = colmask ? (int)cols.size() : (int)k;
    if (batch == 0) return WX_OK;
    if (inplace && (ncols == 0 || row_lo == n)) return WX_OK;
    hipStream_t st = wx_stream(stream);
    WxScratch scr(st);
    WxIO io(st);
    const T *dt = (const T *)io.in(t, sizeof(T) * nt);
    const int per_signal = nt == batch ? 1 : 0;
    if (inplace) {
        T *dX = (T *)io.in(X, sizeof(T) * n * k * batch);
        for (auto &it : io.items) if (it.user == X) it.copy_out = true;
        if (!dX || !dt) return io.finish(WX_EHIP);
        const int *dcols = nullptr;
        if (colmask) {
            dcols = (const int *)scr.upload(cols.data(), cols.size() * sizeof(int));
            if (!dcols) return io.finish(WX_EHIP);
        }
        const int64_t total = (n - row_lo) * ncols * batch;
        int64_t grid = (total + 255) / 256;
        if (grid > 256 * 32) grid = 256 * 32;
        hipLaunchKernelGGL(k_threshold<T>, dim3((unsigned)grid), dim3(256), 0, st, dX, (int)n, (int)k, batch, th_kind, dt,
                           per_signal, (int)row_lo, dcols, ncols);
    } else {
        const T *dX = (const T *)io.in(X, sizeof(T) * n * k * batch);
        T *dY = (T *)io.out(Y, sizeof(T) * n * k * batch);
        if (!dX || !dY || !dt) return io.finish(WX_EHIP);
        const uint8_t *dflag = nullptr;
        if (colmask) {
            dflag = (const uint8_t *)scr.upload(colmask, (size_t)k);
            if (!dflag) return io.finish(WX_EHIP);
        }
        const int64_t total = n * k * batch;
        int64_t grid = (total + 255) / 256;
        if (grid > 256 * 32) grid = 256 * 32;
        hipLaunchKernelGGL(k_threshold_copy<T>, dim3((unsigned)grid), dim3(256), 0, st, dX, dY, (int)n, (int)k, batch, th_kind,
                           dt, per_signal, (int)row_lo, dflag);
    }
    if (hipGetLastError() != hipSuccess) return io.finish(wx_set_error(WX_EHIP, "threshold kernel failed to launch"));
    return io.finish(WX_OK);
}

}  // namespace

extern "C" {
int wx_noisest_f64(const double *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, double *sigma, void *stream)
{ return api_noisest<double>(X, n, k, batch, row_lo, col, sigma, stream); }
int wx_noisest_f32(const float *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, float *sigma, void *stream)
{ return api_noisest<float>(X, n, k, batch, row_lo, col, sigma, stream); }
int wx_threshold_f64(const double *X, double *Y, int64_t n, int64_t k, int64_t batch, int th_kind, const double *t, int64_t nt,
                     int64_t row_lo, const uint8_t *colmask, void *stream)
{ return api_threshold<double>(X, Y, n, k, batch, th_kind, t, nt, row_lo, colmask, stream); }
int wx_threshold_f32(const float *X, float *Y, int64_t n, int64_t k, int64_t batch, int th_kind, const float *t, int64_t nt,
                     int64_t row_lo, const uint8_t *colmask, void *stream)
{ return api_threshold<float>(X, Y, n, k, batch, th_kind, t, nt, row_lo, colmask, stream); }
}
