// wx_haar.hip -- Haar wavelet packets as Walsh-Hadamard transforms (full tree, 1-D, Float64).
//
// With the two-tap filter q0 = q1 a packet level has no halo: a = q0 (v[2i] + v[2i+1]), d = q0 (v[2i+1] - v[2i])
// (dwt/dwt_one_level.jl:79-107 with F = 2), so the depth-L transform of a block of 2^L consecutive samples is an
// in-place Walsh-Hadamard transform of that block, level l acting on index bit l-1, followed by a bit reversal of
// the packet index (the level-1 choice is the most significant bit of the node number, Wavelets.jl's packet
// order).  Nothing here needs an LDS round trip per level: a lane loads 16 samples (index bits 0,1 and 8,9 in
// registers, bits 2..7 across the 64 lanes of a wavefront, one wavefront per block of 1024 samples), does four
// levels in registers and six across lanes, and ONE pass through LDS turns the bit-reversed packet order into
// coalesced stores.  One read and one write of the signal, ~600 VALU instructions and 32 LDS accesses per lane.
// The generic fused kernel (wx_dwt1d.hip) makes an LDS round trip and a barrier per level and reaches 45 % of the
// HBM peak on this transform; this is the route to the 60 % north star for the Haar filter.
#include "wx_common.h"
#include "wx_kernels.h"
#include <cmath>
#include <cstdlib>

namespace {

// value of the lane whose id differs in bit m: DPP quad permutes / row rotate for bits 0, 1, 3 (VALU only), the LDS
// crossbar (ds_bpermute) for the others
template <int M> __device__ __forceinline__ double haar_partner(double v)
{
    if constexpr (M == 0 || M == 1 || M == 3) {
        constexpr int ctrl = M == 0 ? 0xB1 : (M == 1 ? 0x4E : 0x128);   // quad_perm [1,0,3,2], [2,3,0,1], row_ror:8
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const int plo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, false);
        const int phi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, false);
        return __hiloint2double(phi, plo);
    } else {
        return __shfl_xor(v, 1 << M, 64);
    }
}

// cross-lane butterfly on lane bit M: forward: the lane whose bit is 0 keeps x0 + x1, the other x1 - x0;
// inverse (a on bit 0, d on bit 1): x0 = a - d, x1 = a + d
template <int M, bool INV> __device__ __forceinline__ double haar_xlane(double v, bool hi)
{
    const double p = haar_partner<M>(v);
    if (!INV) return hi ? v - p : v + p;
    return hi ? p + v : v - p;
}

// LDS position of output element o: the low four index bits are XORed with bits 6..9, so that the 16 lanes of a
// write group (whose outputs differ in exactly those bits) land on 16 different banks; a bijection inside every
// aligned block of 1024 elements
__device__ __forceinline__ int haar_sw(int o) { return o ^ ((o >> 6) & 15); }

template <int NT>
__global__ __launch_bounds__(NT) void k_haar_wpt_f64(const double *__restrict__ x, double *__restrict__ y, int log2n,
                                                     int L, int64_t batch, double scale)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem6[];
    double *lds = reinterpret_cast<double *>(wx_smem6);
    const int n = 1 << log2n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        const double *xs = x + b * n + wave * 1024;
        double r[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double4 v = reinterpret_cast<const double4 *>(xs + k * 256)[lane];
            r[k][0] = v.x; r[k][1] = v.y; r[k][2] = v.z; r[k][3] = v.w;
        }
        // index bit of level l is l - 1: bits 0,1 = c (registers), 2..7 = lane, 8,9 = k (registers)
        if (L >= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double a0 = r[k][0] + r[k][1], d0 = r[k][1] - r[k][0];
                const double a1 = r[k][2] + r[k][3], d1 = r[k][3] - r[k][2];
                r[k][0] = a0; r[k][1] = d0; r[k][2] = a1; r[k][3] = d1;
            }
        }
        if (L >= 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double a0 = r[k][0] + r[k][2], d0 = r[k][2] - r[k][0];
                const double a1 = r[k][1] + r[k][3], d1 = r[k][3] - r[k][1];
                r[k][0] = a0; r[k][2] = d0; r[k][1] = a1; r[k][3] = d1;
            }
        }
#define WX_XL(M)                                                                          \
        if (L >= 3 + M) {                                                                 \
            const bool hi = (lane >> M) & 1;                                              \
            _Pragma("unroll") for (int k = 0; k < 4; ++k)                                 \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) r[k][c] = haar_xlane<M, false>(r[k][c], hi); \
        }
        WX_XL(0) WX_XL(1) WX_XL(2) WX_XL(3) WX_XL(4) WX_XL(5)
#undef WX_XL
        if (L >= 9) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double a0 = r[0][c] + r[1][c], d0 = r[1][c] - r[0][c];
                const double a1 = r[2][c] + r[3][c], d1 = r[3][c] - r[2][c];
                r[0][c] = a0; r[1][c] = d0; r[2][c] = a1; r[3][c] = d1;
            }
        }
        if (L >= 10) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double a0 = r[0][c] + r[2][c], d0 = r[2][c] - r[0][c];
                const double a1 = r[1][c] + r[3][c], d1 = r[3][c] - r[1][c];
                r[0][c] = a0; r[2][c] = d0; r[1][c] = a1; r[3][c] = d1;
            }
        }
        // slot e of the signal now holds packet f = e mod 2^L (bit l-1 = choice of level l), time index e >> L;
        // it belongs at node j = bitreverse_L(f), position j * (n >> L) + (e >> L)
        const int S1 = (1 << L) - 1, tl = log2n - L;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = wave * 1024 + k * 256 + lane * 4 + c;
                const int j = (int)(__builtin_bitreverse32((unsigned)(e & S1)) >> (32 - L));
                lds[haar_sw((j << tl) + (e >> L))] = r[k][c] * scale;
            }
        __syncthreads();
        double *ys = y + b * n;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int o = q * NT + tid;
            ys[o] = lds[haar_sw(o)];
        }
        __syncthreads();
    }
}

// the read-side counterpart of haar_sw for the inverse: its scattered accesses are 32-lane ds_read_b64 groups whose
// positions differ in bits 5..9
__device__ __forceinline__ int haar_sw_inv(int o) { return o ^ ((o >> 5) & 31); }

// inverse: leaves in packet order -> one pass through LDS to the bit-reversed slots -> inverse butterflies
// (a, d) -> (a - d, a + d) on every index bit below L -> coalesced 32-byte stores
template <int NT>
__global__ __launch_bounds__(NT) void k_haar_iwpt_f64(const double *__restrict__ xw, double *__restrict__ y, int log2n,
                                                      int L, int64_t batch, double scale)
{
    extern __shared__ __attribute__((aligned(16))) char wx_smem7[];
    double *lds = reinterpret_cast<double *>(wx_smem7);
    const int n = 1 << log2n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        const double *xs = xw + b * n;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int o = q * NT + tid;
            lds[haar_sw_inv(o)] = xs[o];
        }
        __syncthreads();
        double r[4][4];
        const int S1 = (1 << L) - 1, tl = log2n - L;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = wave * 1024 + k * 256 + lane * 4 + c;
                const int j = (int)(__builtin_bitreverse32((unsigned)(e & S1)) >> (32 - L));
                r[k][c] = lds[haar_sw_inv((j << tl) + (e >> L))];
            }
        if (L >= 10) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double x0 = r[0][c] - r[2][c], x1 = r[0][c] + r[2][c];
                const double z0 = r[1][c] - r[3][c], z1 = r[1][c] + r[3][c];
                r[0][c] = x0; r[2][c] = x1; r[1][c] = z0; r[3][c] = z1;
            }
        }
        if (L >= 9) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double x0 = r[0][c] - r[1][c], x1 = r[0][c] + r[1][c];
                const double z0 = r[2][c] - r[3][c], z1 = r[2][c] + r[3][c];
                r[0][c] = x0; r[1][c] = x1; r[2][c] = z0; r[3][c] = z1;
            }
        }
#define WX_XL(M)                                                                          \
        if (L >= 3 + M) {                                                                 \
            const bool hi = (lane >> M) & 1;                                              \
            _Pragma("unroll") for (int k = 0; k < 4; ++k)                                 \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) r[k][c] = haar_xlane<M, true>(r[k][c], hi); \
        }
        WX_XL(5) WX_XL(4) WX_XL(3) WX_XL(2) WX_XL(1) WX_XL(0)
#undef WX_XL
        if (L >= 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double x0 = r[k][0] - r[k][2], x1 = r[k][0] + r[k][2];
                const double z0 = r[k][1] - r[k][3], z1 = r[k][1] + r[k][3];
                r[k][0] = x0; r[k][2] = x1; r[k][1] = z0; r[k][3] = z1;
            }
        }
        if (L >= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double x0 = r[k][0] - r[k][1], x1 = r[k][0] + r[k][1];
                const double z0 = r[k][2] - r[k][3], z1 = r[k][2] + r[k][3];
                r[k][0] = x0; r[k][1] = x1; r[k][2] = z0; r[k][3] = z1;
            }
        }
        double *ys = y + b * n + wave * 1024;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double4 v;
            v.x = r[k][0] * scale; v.y = r[k][1] * scale; v.z = r[k][2] * scale; v.w = r[k][3] * scale;
            reinterpret_cast<double4 *>(ys + k * 256)[lane] = v;
        }
        __syncthreads();
    }
}

}  // namespace

// 1 = the transform was launched here, 0 = not applicable (the caller takes the general kernels; decided before
// anything is queued), < 0 = HIP failure reported through wx_set_hip_error
static int wx_haar_launch(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt,
                          hipStream_t st)
{
    static const bool off = wx_getenv("WX_HAAR_WHT") && atoi(wx_getenv("WX_HAAR_WHT")) == 0;
    if (off || filt.F != 2 || filt.q[0] != filt.q[1]) return 0;
    if ((n & (n - 1)) || n < 1024 || n > 8192 || L < 1 || L > 10) return 0;
    int log2n = 0;
    while (((int64_t)1 << (log2n + 1)) <= n) ++log2n;
    // the butterflies add unscaled values and the gain q0^L is applied once (the reference rounds q0 (a +- b) at every
    // level: results differ from it by a few ulp, inside the 1e-10 bar).  q0^2 is exactly 1/2 for the Haar
    // filter up to its own rounding, so pairs of levels contribute an exact power of two.
    double scale = (L & 1) ? filt.q[0] : 1.0;
    {
        const double q2 = filt.q[0] * filt.q[0];
        const double half = (fabs(q2 - 0.5) < 1e-15) ? 0.5 : q2;
        for (int l = 0; l < L / 2; ++l) scale *= half;
    }
    const size_t lds = (size_t)n * sizeof(double);
    const int nt = (int)(n / 16);
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu * nt > 2048) per_cu = 2048 / nt;
    int64_t grid = (int64_t)256 * per_cu;
    if (grid > batch) grid = batch;
#define WX_GO(NTT)                                                                                              \
    {                                                                                                           \
        auto kern = inverse ? k_haar_iwpt_f64<NTT> : k_haar_wpt_f64<NTT>;                                       \
        if (lds > 64 * 1024) {                                                                                  \
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                     \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
            if (ea != hipSuccess) return wx_set_hip_error(ea, "hipFuncSetAttribute", __FILE__, __LINE__);       \
        }                                                                                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTT), lds, st, x, y, log2n, L, batch, scale);        \
    }
    switch (nt) {
    case 64: WX_GO(64) break;
    case 128: WX_GO(128) break;
    case 256: WX_GO(256) break;
    case 512: WX_GO(512) break;
    default: return 0;
    }
#undef WX_GO
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "haar wpt launch", __FILE__, __LINE__);
    return 1;
}

int wx_haar_wpt_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_haar_launch(false, x, y, n, L, batch, filt, st);
}
int wx_haar_iwpt_f64(const double *xw, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st)
{
    return wx_haar_launch(true, xw, y, n, L, batch, filt, st);
}
