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
        for (int64_t b = b0; b < b1; ++b) {
            const T v = p[b * nk];
            s = (T)(s + v);
            q = (T)(q + (T)(v * v));
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
        const T var = (T)(ex2 - (T)(ex * ex));
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
