// wx_bbcost.h -- the entropy cost term and the workgroup sum shared by the per-signal best basis (wx_bb.hip) and
// the shift-invariant packet decomposition (wx_siwt.hip).  bestbasis/bestbasis_costs.jl:104-124.
#pragma once
#include "wx_common.h"

namespace {

template <typename T> __device__ __forceinline__ double bb_term(T x, T nrm, int cost_kind)
{
    // coefcost(x::T, et, nrm): s = (x/nrm)^2 in T; Shannon -s log s, log-energy -log s, -0 when s == 0
    const T r = (T)(x / nrm);
    const T s = (T)(r * r);
    if (s == (T)0) return -0.0;
    const T lg = (T)log((double)s);
    return cost_kind == 0 ? (double)(T)(-(T)(s * lg)) : (double)(T)(-lg);
}

__device__ __forceinline__ double bb_block_sum(double v, double *red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

}  // namespace
