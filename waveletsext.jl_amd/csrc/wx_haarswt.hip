// wx_haarswt.hip -- the last six levels of the stationary Haar packet transform (swpt / average-based iswpt, Float64) as
// sliding Walsh-Hadamard transforms in registers.
//
// Reference semantics: swt/swt_one_level.jl:99-127 (sdwt_step!) and :257-318 (isdwt_step!, average of the two shifts),
// driven level by level by swpt! / iswpt! (SWT.jl:439-472, 613-712).  With the two-tap filter q0 = q1 = c a stationary
// level at dilation s is a = c (v[i-s] + v[i]), d = c (v[i] - v[i-s]): no multiplies beyond the gain, and the 2^K
// descendants of a node K levels down are, at every position i,
//     leaf_path[i] = c^K sum_{k < 2^K} (-1)^{popcount(k & path)} v[i - k s]          (path bit j = choice at level j)
// -- the Walsh-Hadamard transform of the window (v[i], v[i-s], ..., v[i-(2^K-1)s]).  Positions of one residue class
// mod s form an independent periodic sequence, so a LANE walks along one class with the window in registers: one new
// sample per step, a 64-point transform (first stage into 32 temporaries, then two 32-point transforms in place),
// 64 stores.  The 64 lanes of a wavefront take 64 consecutive classes: every load and every store of the wavefront is
// one contiguous 512-byte run.  K = 6: the pass over levels L-5 .. L reads 1/64 of what it writes, so the intermediate
// depth costs 3 % extra traffic (the generic fused passes take K = 3: 28 %) and nothing goes through LDS.
//
// The inverse is the adjoint walk (average-based synthesis = adjoint / 2 per level): at every step the 64 leaf values of
// a position are transformed and added to a shift register of 64 running sums; the sum that has seen its 64
// contributions leaves the register.  The sums that wrap around the end of the class are parked in the output column
// during the first 63 steps and merged during the last 63.
//
// Applies to Float64, q[0] == q[1], the wpt layout, L >= 12 (dilation of the pass >= 64), average-based inverse.
#include "wx_common.h"
#include "wx_kernels.h"
#include <cstdlib>

namespace {

// wave-uniform base of a node's 64 columns kept in scalar registers + one 32-bit element offset per access (64 bases
// would not fit the scalar registers, 64 address pairs not the vector registers); 64 n < 2^32 elements is checked by
// the launcher
typedef const double __attribute__((address_space(1))) *hs_gc;
typedef double __attribute__((address_space(1))) *hs_gm;
__device__ __forceinline__ hs_gc hs_sbase(const double *p)
{
    hs_gc g = (hs_gc)p;
    asm("" : "+s"(g));
    return g;
}
__device__ __forceinline__ hs_gm hs_sbase(double *p)
{
    hs_gm g = (hs_gm)p;
    asm("" : "+s"(g));
    return g;
}

// running element offset of the next column: kept opaque so that the 64 offsets of a step are 64 adds in sequence and
// not 64 loop-invariant registers
__device__ __forceinline__ unsigned hs_next(unsigned off, unsigned step)
{
    off += step;
    asm volatile("" : "+v"(off));
    return off;
}

__device__ __forceinline__ int hs_rev5(int v) { return (int)(__builtin_bitreverse32((unsigned)v) >> 27); }
constexpr int hs_rev5c(int v) { return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4); }
constexpr int hs_rev6c(int v) { return (hs_rev5c(v & 31) << 1) | (v >> 5); }

// in-place 32-point Walsh-Hadamard transform, natural (Hadamard) order: index bit j <-> butterfly distance 2^j;
// the element with the bit set gets (low - high)
__device__ __forceinline__ void hs_wht32(double (&t)[32])
{
#pragma unroll
    for (int j = 0; j < 5; ++j) {
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (!(i & (1 << j))) {
                const double a = t[i], b = t[i | (1 << j)];
                t[i] = a + b;
                t[i | (1 << j)] = a - b;
            }
    }
}

// forward: node columns of depth d0 = L - 6 (column node * 64 of the wpt layout) -> their 64 leaf columns, in place.
// One wavefront per (signal, node, block of 64 residue classes); U steps per loop iteration (the window shifts by U).
template <int U>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_haar_swpt6_fwd(
    double *__restrict__ xw, int n, int ncols, int64_t batch, int d0, double gain)
{
    const int s = 1 << d0, nsteps = n >> d0, groups = s >> 6;
    const int node = blockIdx.x / groups, g = blockIdx.x - node * groups;
    const int lane = threadIdx.x;
    const int64_t r = 64 * g + lane;
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        double *colu = xw + (sig * ncols + (int64_t)node * 64) * n;        // parent column = column of leaf 0 (uniform)
        const hs_gm col = hs_sbase(colu);
        const unsigned ro = (unsigned)r;
        // H[i] = gain * v[m0 - 63 + i]: the window of step m0 + j is H[63 + j - k], k = 0..63
        double H[63 + U];
#pragma unroll
        for (int i = 0; i < 63; ++i) H[i] = gain * col[ro + (unsigned)(nsteps - 63 + i) * (unsigned)s];
        double nxt[U];
#pragma unroll
        for (int j = 0; j < U; ++j) nxt[j] = col[ro + (unsigned)j * (unsigned)s];
        for (int m0 = 0; m0 < nsteps; m0 += U) {
#pragma unroll
            for (int j = 0; j < U; ++j) H[63 + j] = gain * nxt[j];
            if (m0 + U < nsteps) {
#pragma unroll
                for (int j = 0; j < U; ++j) nxt[j] = col[ro + (unsigned)(m0 + U + j) * (unsigned)s];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const unsigned po = ro + (unsigned)(m0 + j) * (unsigned)s;
                double t[32];
                // window index bit 5 (level L) first: sums -> even leaves, differences -> odd leaves
#pragma unroll
                for (int k = 0; k < 32; ++k) t[k] = H[63 + j - k] + H[63 + j - k - 32];
                hs_wht32(t);
                unsigned off = po;                                    // even leaves 0, 2, ..., 62
#pragma unroll
                for (int c2 = 0; c2 < 32; ++c2) { col[off] = t[hs_rev5c(c2)]; off = hs_next(off, 2u * (unsigned)n); }
#pragma unroll
                for (int k = 0; k < 32; ++k) t[k] = H[63 + j - k] - H[63 + j - k - 32];
                hs_wht32(t);
                __builtin_amdgcn_sched_barrier(0);                    // keep the two halves' temporaries apart
                off = po + (unsigned)n;                               // odd leaves 1, 3, ..., 63
#pragma unroll
                for (int c2 = 0; c2 < 32; ++c2) { col[off] = t[hs_rev5c(c2)]; off = hs_next(off, 2u * (unsigned)n); }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 63; ++i) H[i] = H[i + U];
        }
    }
}

// in-place 2^LG-point Walsh-Hadamard transform (same convention as hs_wht32)
template <int LG> __device__ __forceinline__ void hs_wht(double (&t)[1 << LG])
{
#pragma unroll
    for (int j = 0; j < LG; ++j) {
#pragma unroll
        for (int i = 0; i < (1 << LG); ++i)
            if (!(i & (1 << j))) {
                const double a = t[i], b = t[i | (1 << j)];
                t[i] = a + b;
                t[i | (1 << j)] = a - b;
            }
    }
}
constexpr int hs_revc(int v, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

// inverse: the 2^K leaf columns of every node of depth d0 = L - K (src, wpt layout, column node * 2^K + c) -> the node's
// column in dst (dst_cols columns per signal, column = node).  gain = (c / 2)^K.  K = 5: 16 + 16 transform values and
// 32 running sums per lane leave room for three wavefronts per SIMD (K = 6 needs 258 registers: one wavefront, and the
// 64 loads of a step are then fully exposed -- measured 16 ms per 32 GiB against 8.5 ms for the LDS passes).
// ORD (round 6, experiment behind WX_HAAR_ISWT_ORDER): 0 = consecutive blocks are the position groups of one node (the two wavefronts that
// interleave 512-byte pieces of the same 32 columns run side by side), 1 = consecutive blocks are consecutive nodes (the groups of a node
// are 2^d0 blocks apart); WPE = wavefronts per SIMD the kernel is built for (0 = the compiler's choice: three)
template <int K, int U, int ORD = 0, int WPE = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE ? WPE : 1, WPE ? WPE : 8)))
void k_haar_iswpt(const double *__restrict__ src, int64_t src_cols,
                                                    double *__restrict__ dst, int64_t dst_cols, int n, int64_t batch,
                                                    int d0, double gain)
{
    constexpr int NC = 1 << K, HALF = NC / 2;
    const int s = 1 << d0, nsteps = n >> d0, groups = s >> 6;
    const int node = ORD ? (int)(blockIdx.x & (unsigned)(s - 1)) : (int)(blockIdx.x / groups);
    const int g = ORD ? (int)(blockIdx.x >> d0) : (int)(blockIdx.x - node * groups);
    const int lane = threadIdx.x;
    const int64_t r = 64 * g + lane;
    for (int64_t sig = blockIdx.y; sig < batch; sig += gridDim.y) {
        const hs_gc lf = hs_sbase(src + (sig * src_cols + (int64_t)node * NC) * n);
        const hs_gm out = hs_sbase(dst + (sig * dst_cols + node) * n);
        const unsigned ro = (unsigned)r;
        // R[i]: running sum of position m0 + U - 1 - i at the start of the iteration that handles steps m0 .. m0+U-1
        double R[NC - 1 + U];
#pragma unroll
        for (int i = 0; i < NC - 1 + U; ++i) R[i] = 0.0;
        for (int m0 = 0; m0 < nsteps; m0 += U) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const unsigned po = ro + (unsigned)(m0 + j) * (unsigned)s;
                double a[HALF], b[HALF];
                // leaf index bit K-1 (level d0 + 1) <-> window bit 0: a = even window positions, b = odd
                unsigned off = po;
#pragma unroll
                for (int c = 0; c < HALF; ++c) { a[c] = lf[off]; off = hs_next(off, (unsigned)n); }
#pragma unroll
                for (int c = 0; c < HALF; ++c) {
                    const double hi = lf[off];
                    off = hs_next(off, (unsigned)n);
                    b[c] = a[c] - hi;
                    a[c] = a[c] + hi;
                }
                // remaining leaf bits K-2..0 <-> window bits 1..K-1: a transform whose index bit i is leaf bit i
                hs_wht<K - 1>(a);
                hs_wht<K - 1>(b);
                // transformed index z (leaf bits 0..K-2) -> window offset k = 2 * rev(z) (+ 1 for b); the step's position
                // is U - 1 - j places from the head of R
#pragma unroll
                for (int z = 0; z < HALF; ++z) {
                    R[U - 1 - j + 2 * hs_revc(z, K - 1)] += a[z];
                    R[U - 1 - j + 2 * hs_revc(z, K - 1) + 1] += b[z];
                }
            }
            // the U oldest sums are complete -- or, during the first NC - 1 steps, are the part of a sum near the end of
            // the class that wraps around: parked in the output column
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int pos = m0 + j - (NC - 1);                     // position of R[NC - 2 + U - j]
                const double v = R[NC - 2 + U - j];
                if (pos >= 0) out[ro + (unsigned)pos * (unsigned)s] = v * gain;
                else out[ro + (unsigned)(pos + nsteps) * (unsigned)s] = v;
            }
#pragma unroll
            for (int i = NC - 2 + U; i >= U; --i) R[i] = R[i - U];
#pragma unroll
            for (int i = 0; i < U; ++i) R[i] = 0.0;
        }
        // positions nsteps - (NC-1) .. nsteps - 1 (R[U], R[U+1], ... after the last shift): add the parked parts
#pragma unroll
        for (int i = 0; i < NC - 1; ++i) {
            const unsigned pos = ro + (unsigned)(nsteps - 1 - i) * (unsigned)s;
            out[pos] = (R[U + i] + out[pos]) * gain;
        }
    }
}

}  // namespace

static bool hs_enabled()
{
    static const bool off = wx_getenv("WX_HAAR_SWT6") && atoi(wx_getenv("WX_HAAR_SWT6")) == 0;
    return !off;
}
bool wx_haar_swpt6_ok(int64_t n, int L, const WxFilt &filt, size_t esz)
{
    return hs_enabled() && esz == 8 && filt.F == 2 && filt.q[0] == filt.q[1] && L >= 12 && L <= 30 && (n >> L) >= 1 &&
           ((n >> (L - 6)) % 4) == 0 && (n >> (L - 6)) >= 64;
}

// levels L-5 .. L of swpt from the node columns of depth L - 6, in place in the (n, 2^L) table
int wx_haar_swpt6_fwd(double *xw, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    const int d0 = L - 6;
    double gain = 1.0;
    for (int i = 0; i < 6; ++i) gain *= filt.q[0];
    const int64_t blocks = ((int64_t)1 << d0) * (((int64_t)1 << d0) >> 6);
    int64_t gy = batch > 65535 ? 65535 : batch;
    // steps per loop iteration: 1 measured best (7.0 ms per 32 GiB; 8.1 at 2, 11.8 at 4: the unrolled bodies spill)
    static const int U = wx_getenv("WX_HAAR_SWT6_U") ? atoi(wx_getenv("WX_HAAR_SWT6_U")) : 1;
    if (U == 1)
        hipLaunchKernelGGL(k_haar_swpt6_fwd<1>, dim3((unsigned)blocks, (unsigned)gy), dim3(64), 0, st, xw, (int)n, 1 << L, batch, d0, gain);
    else if (U == 2)
        hipLaunchKernelGGL(k_haar_swpt6_fwd<2>, dim3((unsigned)blocks, (unsigned)gy), dim3(64), 0, st, xw, (int)n, 1 << L, batch, d0, gain);
    else
        hipLaunchKernelGGL(k_haar_swpt6_fwd<4>, dim3((unsigned)blocks, (unsigned)gy), dim3(64), 0, st, xw, (int)n, 1 << L, batch, d0, gain);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "haar swpt6 launch", __FILE__, __LINE__);
    return WX_OK;
}

// depth L -> L - K of the average-based iswpt (K = wx_haar_iswpt_levels()): src (n, src_cols) holds the leaves of the
// wpt layout, dst gets the 2^(L-K) node columns
int wx_haar_iswpt_levels()
{
    static const int k = wx_getenv("WX_HAAR_ISWT_K") ? atoi(wx_getenv("WX_HAAR_ISWT_K")) : 5;
    return k == 6 ? 6 : 5;
}
int wx_haar_iswpt6(const double *src, int64_t src_cols, double *dst, int64_t dst_cols, int64_t n, int L, int64_t batch,
                   const WxFilt &filt, hipStream_t st)
{
    const int K = wx_haar_iswpt_levels();
    const int d0 = L - K;
    double gain = 1.0;
    for (int i = 0; i < K; ++i) gain *= 0.5 * filt.q[0];
    const int64_t blocks = ((int64_t)1 << d0) * (((int64_t)1 << d0) >> 6);
    int64_t gy = batch > 65535 ? 65535 : batch;
    if (K == 6)
        hipLaunchKernelGGL((k_haar_iswpt<6, 1>), dim3((unsigned)blocks, (unsigned)gy), dim3(64), 0, st, src, src_cols, dst,
                           dst_cols, (int)n, batch, d0, gain);
    else
    {
        // profiles/r06_cfg3_inverse.txt: six consecutive processes each -- order 1 is 1-2 % faster in every placement of the table (6.88 / 7.47 /
        // 8.12 ms against 7.03 / 7.58 / 8.19), residency 2 the same as 3, 4 (spills) 18 ms; the spread itself follows the process sequence
        // whatever the order or the residency: it is the physical placement of the 32 GiB table (profiles/r05_cfg3_inverse.md)
        static const int ord = wx_getenv("WX_HAAR_ISWT_ORDER") ? atoi(wx_getenv("WX_HAAR_ISWT_ORDER")) : 1;
        static const int wpe = wx_getenv("WX_HAAR_ISWT_WPE") ? atoi(wx_getenv("WX_HAAR_ISWT_WPE")) : 0;
#define WX_HI(O, W) hipLaunchKernelGGL((k_haar_iswpt<5, 1, O, W>), dim3((unsigned)blocks, (unsigned)gy), dim3(64), 0, st, src, src_cols, dst, dst_cols, (int)n, batch, d0, gain)
        if (ord == 1 && wpe == 2) WX_HI(1, 2);
        else if (ord == 1 && wpe == 4) WX_HI(1, 4);
        else if (ord == 1) WX_HI(1, 0);
        else if (wpe == 2) WX_HI(0, 2);
        else if (wpe == 4) WX_HI(0, 4);
        else WX_HI(0, 0);
#undef WX_HI
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "haar iswpt launch", __FILE__, __LINE__);
    return WX_OK;
}
