// wx_smalltree.hip -- wpt / iwpt along ANY tree (pyramids and full trees included) of SHORT signals, 16 ... 512 samples.
//
// The fused LDS kernels of wx_dwt1d.hip give a signal a workgroup of at least 64 threads; a 64-sample signal has 32 pairs at its
// first level and one at its last, so most lanes idle through a chain of barriers (round 3: 4-10 % of the HBM peak for 64- and
// 128-sample signals with a tree, every Float32 transform of such signals included).  Here a signal gets n / 8 lanes -- eight
// 64-sample signals share a wavefront, a workgroup of four wavefronts takes 512 x 4 samples per step -- and every lane owns four
// pair slots of every level: slot q of depth d is pair q mod (m / 2) of node q div (m / 2), m = n >> d.  All signals follow the
// same tree, so the level loop is uniform; whether a slot's node is decomposed (status byte, like k_fwd1d_level) only selects
// between the filter and a two-element copy, and both buffers keep the wpt layout (node j of depth d at [j m, (j + 1) m),
// Utils.jl:101-134), so the last buffer is the output.  Arithmetic and tap order: dwt_step! / idwt_step!
// (dwt/dwt_one_level.jl:94-105, 207-221), Float64 accumulation, one rounding to the signal's type per level.
#include "wx_common.h"
#include "wx_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int ST_NT = 256;                 // 4 wavefronts, each with its own 512-sample slab
constexpr int ST_SLAB = 512;               // samples per wavefront and step (64 lanes x 8)

template <typename T, int F, bool INVERSE>
__global__ __launch_bounds__(ST_NT) void k_small_tree(const T *__restrict__ x, T *__restrict__ y, int log2n, int L, int64_t nslabs,
                                                      int64_t total, const uint8_t *__restrict__ status, int64_t nstatus, WxFilt filt)
{
    typedef typename WxVec2<T>::type V2;
    typedef typename std::conditional<sizeof(T) == 8, double, float>::type A;     // Float32 signals accumulate in Float32 (1e-7 per level)
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    const int n = 1 << log2n;
    const int sstride = n + 2;                                // words per signal: no two signals of a slab start in the same bank
    const int nsig = ST_SLAB >> log2n;                        // signals per slab
    const int slabw = nsig * sstride + 8;                     // words per slab buffer
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T *lds = reinterpret_cast<T *>(wx_smem) + (size_t)wave * 2 * slabw;      // this wavefront's two buffers: words [0, slabw) and [slabw, 2 slabw)
    uint8_t *sst = reinterpret_cast<uint8_t *>(reinterpret_cast<T *>(wx_smem) + (size_t)(ST_NT / 64) * 2 * slabw);
    // the tree: one byte per node of depth < L (heap order), NULL = the full tree of depth L
    const int nnodes = (1 << L) - 1;
    for (int i = threadIdx.x; i < nnodes; i += ST_NT) sst[i] = status ? (i < nstatus ? status[i] : 0) : 1;
    const int G = n >> 3;                                     // lanes per signal
    const int s = lane / G, g = lane - s * G;                 // this lane's signal of the slab, its place in the group
    const int sb = s * sstride;
    A q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (A)filt.q[k];
    __syncthreads();
    const int64_t per_step = (int64_t)gridDim.x * (ST_NT / 64);
    const int64_t nsteps = (nslabs + per_step - 1) / per_step;
    for (int64_t it = 0; it < nsteps; ++it) {
        const int64_t slab = (it * gridDim.x + blockIdx.x) * (ST_NT / 64) + wave;
        const bool live = slab < nslabs;                      // every wavefront of a workgroup runs the same number of steps
        const int64_t e0 = slab * ST_SLAB;
        // load: 4 x 2 samples per lane, consecutive lanes on consecutive addresses
        if (live) {
            V2 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = t * 128 + lane * 2;
                if (e0 + e < total) v[t] = *reinterpret_cast<const V2 *>(x + e0 + e);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = t * 128 + lane * 2;
                if (e0 + e < total) {
                    const int a = (e >> log2n) * sstride + (e & (n - 1));
                    lds[a] = v[t].x;
                    lds[a + 1] = v[t].y;
                }
            }
        }
        __syncthreads();
        int co = 0, no = slabw;                               // integer offsets keep the accesses LDS instructions
        const bool sig_ok = live && (e0 + (int64_t)(s + 1) * n <= total);
        for (int lv = 0; lv < L; ++lv) {
            const int d = INVERSE ? L - 1 - lv : lv;
            const int lm = log2n - d, m = 1 << lm, hm = m >> 1;
            if (sig_ok) {
                // (two adjacent pairs sharing one window were measured: slower -- more registers, lanes four samples apart)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int qq = t * G + g;
                    const int j = qq >> (lm - 1), i = qq & (hm - 1);
                    const int base = sb + (j << lm);
                    const bool split = sst[(1 << d) - 1 + j] != 0;
                    if (!split) {
                        lds[no + base + 2 * i] = lds[co + base + 2 * i];
                        lds[no + base + 2 * i + 1] = lds[co + base + 2 * i + 1];
                        continue;
                    }
                    if (!INVERSE) {
                        // a[i] reads v[2i .. 2i+F-1]; d[i'] reads v[2i'+2-F .. 2i'+1], the same window for i' = i + F/2 - 1 (mod m/2): one
                        // window of F samples gives both (2F - 2 reads with d[i]; the taps and their order per output are unchanged)
                        A w[F];
#pragma unroll
                        for (int e = 0; e < F; ++e) w[e] = (A)lds[co + base + ((2 * i + e) & (m - 1))];
                        A a = 0, dd = 0;
#pragma unroll
                        for (int k = 0; k < F; ++k) {
                            a = fma(q[k], w[k], a);
                            dd = fma((k & 1) ? -q[k] : q[k], w[F - 1 - k], dd);
                        }
                        lds[no + base + i] = (T)a;
                        lds[no + base + hm + ((i + F / 2 - 1) & (hm - 1))] = (T)dd;
                    } else {
                        A v0 = 0, v1 = 0;
#pragma unroll
                        for (int mm = 0; mm < F / 2; ++mm) {
                            const A av = (A)lds[co + base + ((i - mm) & (hm - 1))];
                            const A dv = (A)lds[co + base + hm + ((i + mm) & (hm - 1))];
                            v0 = fma(q[2 * mm], av, v0);
                            v0 = fma(-q[2 * mm + 1], dv, v0);
                            v1 = fma(q[2 * mm + 1], av, v1);
                            v1 = fma(q[2 * mm], dv, v1);
                        }
                        lds[no + base + 2 * i] = (T)v0;
                        lds[no + base + 2 * i + 1] = (T)v1;
                    }
                }
            }
            __syncthreads();
            const int tmp = co; co = no; no = tmp;
        }
        if (live) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = t * 128 + lane * 2;
                if (e0 + e < total) {
                    const int a = co + (e >> log2n) * sstride + (e & (n - 1));
                    V2 v;
                    v.x = lds[a];
                    v.y = lds[a + 1];
                    *reinterpret_cast<V2 *>(y + e0 + e) = v;
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int F>
int launch_small(bool inverse, const T *x, T *y, int64_t n, int L, int64_t batch, const WxFilt &filt, const uint8_t *status, int64_t nstatus,
                 hipStream_t st)
{
    const int log2n = 63 - __builtin_clzll((unsigned long long)n);
    const int64_t total = n * batch;
    const int64_t nslabs = (total + ST_SLAB - 1) / ST_SLAB;
    const int nsig = ST_SLAB >> log2n;
    const size_t slabw = (size_t)nsig * (n + 2) + 8;
    const size_t lds = sizeof(T) * (ST_NT / 64) * 2 * slabw + 512 + 64;
    int64_t wgs = (nslabs + (ST_NT / 64) - 1) / (ST_NT / 64);
    const int64_t cap = (int64_t)256 * 8;
    if (wgs > cap) wgs = cap;
    if (inverse)
        hipLaunchKernelGGL((k_small_tree<T, F, true>), dim3((unsigned)wgs), dim3(ST_NT), lds, st, x, y, log2n, L, nslabs, total, status, nstatus, filt);
    else
        hipLaunchKernelGGL((k_small_tree<T, F, false>), dim3((unsigned)wgs), dim3(ST_NT), lds, st, x, y, log2n, L, nslabs, total, status, nstatus, filt);
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

}  // namespace

template <typename T> bool wx_small_tree_ok(int64_t n, int F)
{
    static const bool off = wx_getenv("WX_SMALLTREE") && atoi(wx_getenv("WX_SMALLTREE")) == 0;
    return !off && n >= 16 && n <= 512 && (n & (n - 1)) == 0 && F >= 2 && F <= 20 && (F & 1) == 0;
}
template bool wx_small_tree_ok<double>(int64_t, int);
template bool wx_small_tree_ok<float>(int64_t, int);

// Which short signals leave the fused LDS kernels (profiles/r04_floor.txt): any tree of signals up to 256 samples (one lane per signal
// where the signal fits a lane's registers -- wx_lanetree.h: up to 64 samples, 128 for Float32 --, n / 8 lanes per signal here
// otherwise; the fused kernels run them at 6-20 %, 512-sample trees at 22-31 %: those stay); pyramids up to 128 samples (longer ones run
// at 40 % with the lane-local tail); Float32 full trees up to 128 samples (Float64 ones have the interleaved lattice kernels at 65 %).
template <typename T> bool wx_small_tree_wanted(int64_t n, int F, bool has_tree, bool pyramid)
{
    if (!wx_small_tree_ok<T>(n, F)) return false;
    static const int maxn = wx_getenv("WX_SMALLTREE_MAXN") ? atoi(wx_getenv("WX_SMALLTREE_MAXN")) : 256;
    if (has_tree) return pyramid ? n <= 128 : n <= maxn;
    return sizeof(T) == 4 && n <= 128;
}
template bool wx_small_tree_wanted<double>(int64_t, int, bool, bool);
template bool wx_small_tree_wanted<float>(int64_t, int, bool, bool);

// x, y: (n, batch) dense; status: device bytes, one per node of depth < L in heap order ("exists and is decomposed"), NULL = full tree
template <typename T>
int wx_dev_small_tree(bool inverse, const T *x, T *y, int64_t n, int L, int64_t batch, const WxFilt &filt, const uint8_t *status,
                      int64_t nstatus, hipStream_t st)
{
    if (batch == 0) return WX_OK;
    if (!wx_small_tree_ok<T>(n, filt.F) || L < 1 || ((int64_t)1 << L) > n) return wx_set_error(WX_EUNSUPPORTED, "small-signal tree kernel: shape");
    switch (filt.F) {
#define WX_ST(FF) case FF: return launch_small<T, FF>(inverse, x, y, n, L, batch, filt, status, nstatus, st);
        WX_ST(2) WX_ST(4) WX_ST(6) WX_ST(8) WX_ST(10) WX_ST(12) WX_ST(14) WX_ST(16) WX_ST(18) WX_ST(20)
#undef WX_ST
    }
    return wx_set_error(WX_EUNSUPPORTED, "small-signal tree kernel: filter length");
}
template int wx_dev_small_tree<double>(bool, const double *, double *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t, hipStream_t);
template int wx_dev_small_tree<float>(bool, const float *, float *, int64_t, int, int64_t, const WxFilt &, const uint8_t *, int64_t, hipStream_t);
