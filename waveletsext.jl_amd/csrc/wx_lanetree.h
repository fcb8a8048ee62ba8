// wx_lanetree.h -- wpt / iwpt along any tree of VERY short signals (16 ... 64 samples; 128 for Float32): one LANE per signal.
//
// All signals of a batch follow the same tree, so when a lane owns a whole signal the walk over the tree is uniform: which node
// comes next, where it starts and how the periodic wrap folds its taps are the same for the 64 lanes of a wavefront -- scalar or
// compile-time, not per-lane arithmetic.  A wavefront transposes 64 consecutive signals into LDS (signal stride n + 1 words: the
// lanes' rows start in different banks); a decomposed node of m samples is read into m registers (LDS offsets are immediates), its
// m / 2 pairs are computed with the wrap resolved at compile time, and the m results go back IN PLACE -- the node is in registers,
// nothing it still needs is overwritten, and a node that is not decomposed simply stays where it is (wpt layout, Utils.jl:101-134).
// Per pair: 2 F multiply-adds and two LDS accesses, no index arithmetic (the n / 8-lanes-per-signal kernel of wx_smalltree.hip
// spends 40 vector instructions per sample and level on addresses: 9-14 % of the HBM peak for Float32 signals of 64 samples).
// Arithmetic and tap order: dwt_step! / idwt_step! (dwt/dwt_one_level.jl:94-105, 207-221); Float64 signals accumulate in Float64,
// Float32 signals in Float32.
#pragma once
#include "wx_common.h"
#include <type_traits>

struct WxLaneTree {
    unsigned bits[8];       // bit h-1: heap node h (depth < L) is decomposed
};

template <typename T, int F, int M, bool INVERSE, typename A>
__device__ __forceinline__ void wx_lane_node(T *sig, const A (&q)[F])
{
    A r[M];
#pragma unroll
    for (int e = 0; e < M; ++e) r[e] = (A)sig[e];
    constexpr int HM = M / 2;
    if (!INVERSE) {
#pragma unroll
        for (int i = 0; i < HM; ++i) {
            A a = 0, d = 0;
#pragma unroll
            for (int k = 0; k < F; ++k) {
                a = fma(q[k], r[(2 * i + k) % M], a);
                d = fma((k & 1) ? -q[k] : q[k], r[((2 * i + 1 - k) % M + M) % M], d);
            }
            sig[i] = (T)a;
            sig[HM + i] = (T)d;
        }
    } else {
#pragma unroll
        for (int i = 0; i < HM; ++i) {
            A v0 = 0, v1 = 0;
#pragma unroll
            for (int m = 0; m < F / 2; ++m) {
                const A av = r[((i - m) % HM + HM) % HM], dv = r[HM + (i + m) % HM];
                v0 = fma(q[2 * m], av, v0);
                v0 = fma(-q[2 * m + 1], dv, v0);
                v1 = fma(q[2 * m + 1], av, v1);
                v1 = fma(q[2 * m], dv, v1);
            }
            sig[2 * i] = (T)v0;
            sig[2 * i + 1] = (T)v1;
        }
    }
}

template <typename T, int F, int LOG2N, bool INVERSE>
__global__ __launch_bounds__(256) void k_lane_tree(const T *__restrict__ x, T *__restrict__ y, int L, int64_t batch, WxLaneTree tree, WxFilt filt)
{
    typedef typename WxVec2<T>::type V2;
    typedef typename std::conditional<sizeof(T) == 8, double, float>::type A;
    extern __shared__ __attribute__((aligned(16))) char wx_smem[];
    constexpr int N = 1 << LOG2N, STRIDE = N + 1, SLAB = 64 * N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T *lds = reinterpret_cast<T *>(wx_smem) + (size_t)wave * (64 * STRIDE + 8);
    T *mine = lds + lane * STRIDE;
    A q[F];
#pragma unroll
    for (int k = 0; k < F; ++k) q[k] = (A)filt.q[k];
    const int64_t total = batch * N, nslabs = (batch + 63) / 64;
    const int nw = blockDim.x >> 6;                           // wavefronts per workgroup: as many slabs as the LDS holds, at most 4
    const int64_t per_step = (int64_t)gridDim.x * nw;
    const int64_t nsteps = (nslabs + per_step - 1) / per_step;
    for (int64_t it = 0; it < nsteps; ++it) {
        const int64_t slab = (it * gridDim.x + blockIdx.x) * nw + wave;
        const int64_t e0 = slab * SLAB;
        if (slab < nslabs) {
            // 64 consecutive signals, 16 bytes (Float64) / 8 bytes (Float32) per lane and load, 8 loads in flight
#pragma unroll
            for (int c0 = 0; c0 < SLAB / 128; c0 += 8) {
                V2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = (c0 + u) * 128 + lane * 2;
                    if (c0 + u < SLAB / 128 && e0 + e < total) v[u] = *reinterpret_cast<const V2 *>(x + e0 + e);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = (c0 + u) * 128 + lane * 2;
                    if (c0 + u < SLAB / 128 && e0 + e < total) {
                        T *p = lds + (e >> LOG2N) * STRIDE + (e & (N - 1));
                        p[0] = v[u].x;
                        p[1] = v[u].y;
                    }
                }
            }
        }
        __syncthreads();
        if (slab < nslabs && slab * 64 + lane < batch) {
            auto level = [&](auto dc) {
                constexpr int D = decltype(dc)::value;
                if (D >= L) return;
                constexpr int M = N >> D;
                for (int j = 0; j < (1 << D); ++j) {
                    const int h = (1 << D) - 1 + j;                       // heap index - 1: the same for every lane
                    if ((tree.bits[h >> 5] >> (h & 31)) & 1u) wx_lane_node<T, F, M, INVERSE, A>(mine + j * M, q);
                }
            };
            if (!INVERSE) {
                level(std::integral_constant<int, 0>{});
                level(std::integral_constant<int, 1>{});
                level(std::integral_constant<int, 2>{});
                level(std::integral_constant<int, 3>{});
                if constexpr (LOG2N >= 5) level(std::integral_constant<int, 4>{});
                if constexpr (LOG2N >= 6) level(std::integral_constant<int, 5>{});
                if constexpr (LOG2N >= 7) level(std::integral_constant<int, 6>{});
                if constexpr (LOG2N >= 8) level(std::integral_constant<int, 7>{});
            } else {
                if constexpr (LOG2N >= 8) level(std::integral_constant<int, 7>{});
                if constexpr (LOG2N >= 7) level(std::integral_constant<int, 6>{});
                if constexpr (LOG2N >= 6) level(std::integral_constant<int, 5>{});
                if constexpr (LOG2N >= 5) level(std::integral_constant<int, 4>{});
                level(std::integral_constant<int, 3>{});
                level(std::integral_constant<int, 2>{});
                level(std::integral_constant<int, 1>{});
                level(std::integral_constant<int, 0>{});
            }
        }
        __syncthreads();
        if (slab < nslabs) {
#pragma unroll 4
            for (int c = 0; c < SLAB / 128; ++c) {
                const int e = c * 128 + lane * 2;
                if (e0 + e < total) {
                    const T *p = lds + (e >> LOG2N) * STRIDE + (e & (N - 1));
                    V2 v;
                    v.x = p[0];
                    v.y = p[1];
                    *reinterpret_cast<V2 *>(y + e0 + e) = v;
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int F, int LOG2N>
static int wx_lane_launch(bool inverse, const T *x, T *y, int L, int64_t batch, const WxLaneTree &tree, const WxFilt &filt, hipStream_t st)
{
    constexpr int N = 1 << LOG2N;
    const size_t per_wave = sizeof(T) * (64 * (N + 1) + 8);
    const int nw = per_wave * 4 <= 150 * 1024 ? 4 : (per_wave * 2 <= 150 * 1024 ? 2 : 1);
    const size_t lds = per_wave * nw;
    const int64_t nslabs = (batch + 63) / 64;
    int64_t wgs = (nslabs + nw - 1) / nw;
    const int64_t cap = (int64_t)256 * 8;
    if (wgs > cap) wgs = cap;
    if (inverse) {
        auto k = k_lane_tree<T, F, LOG2N, true>;
        if (lds > 64 * 1024) WX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k, dim3((unsigned)wgs), dim3(64 * nw), lds, st, x, y, L, batch, tree, filt);
    } else {
        auto k = k_lane_tree<T, F, LOG2N, false>;
        if (lds > 64 * 1024) WX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k, dim3((unsigned)wgs), dim3(64 * nw), lds, st, x, y, L, batch, tree, filt);
    }
    WX_HIP_CHECK(hipGetLastError());
    return WX_OK;
}

template <typename T, int F>
static int wx_lane_dispatch(bool inverse, const T *x, T *y, int64_t n, int L, int64_t batch, const WxLaneTree &tree, const WxFilt &filt,
                            hipStream_t st)
{
    switch (n) {
    case 16: return wx_lane_launch<T, F, 4>(inverse, x, y, L, batch, tree, filt, st);
    case 32: return wx_lane_launch<T, F, 5>(inverse, x, y, L, batch, tree, filt, st);
    case 64: return wx_lane_launch<T, F, 6>(inverse, x, y, L, batch, tree, filt, st);
    case 128:
        if constexpr (sizeof(T) == 4) return wx_lane_launch<T, F, 7>(inverse, x, y, L, batch, tree, filt, st);
        break;
    // (256 Float32 samples, the root in 256 registers at one wavefront per SIMD, was measured: 11 % for the full tree against the 15 % of
    // the fused LDS kernel and the 16-24 % of the n / 8-lanes-per-signal kernel, which keep those signals)
    }
    return wx_set_error(WX_EUNSUPPORTED, "lane-per-signal tree kernel: length");
}
