// wx_bbcost.h -- the entropy cost term and the workgroup sum shared by the per-signal best basis (wx_bb.hip) and
// the shift-invariant packet decomposition (wx_siwt.hip).  bestbasis/bestbasis_costs.jl:104-124.
#pragma once
#include "wx_common.h"

namespace {

// log of a positive finite double in ~35 instructions instead of the library's ~100 (the entropy costs are
// bound by it): s = m 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh z, z = (m-1)/(m+1), |z| <= 0.172, ten
// terms of the odd series; e ln 2 in two parts.  <= 2 ulp from glibc's log over 5e7 samples (all binades,
// around 1, tiny squares); denormals, zero, inf and NaN take the library.
__device__ __forceinline__ double wx_log_pos(double s)
{
    if (!(s >= 2.2250738585072014e-308) || !(s < __builtin_inf())) return log(s);
    int e;
    double m = frexp(s, &e);                       // [0.5, 1)
    if (m < 0.70710678118654752) { m += m; e -= 1; }
    // (m - 1) / (m + 1), divisor in [1.7, 2.42]: reciprocal estimate + two Newton steps + one correction of the
    // quotient, without the scaling / fix-up instructions of the general division
    const double dv = m + 1.0, nu = m - 1.0;
    double rc = __builtin_amdgcn_rcp(dv);
    rc = fma(fma(-dv, rc, 1.0), rc, rc);
    rc = fma(fma(-dv, rc, 1.0), rc, rc);
    double z = nu * rc;
    z = fma(fma(-dv, z, nu), rc, z);
    const double w = z * z;
    // Horner with the coefficients as scalar operands of v_fma_f64: written as plain fma() the compiler picks
    // v_fmac_f64, whose addend is the destination, and spends a v_mov_b64 per step to put the constant there
    double p = 1.0 / 19.0;
#define WX_HORNER(c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(w), "s"((double)(c)))
    WX_HORNER(1.0 / 17.0);
    WX_HORNER(1.0 / 15.0);
    WX_HORNER(1.0 / 13.0);
    WX_HORNER(1.0 / 11.0);
    WX_HORNER(1.0 / 9.0);
    WX_HORNER(1.0 / 7.0);
    WX_HORNER(1.0 / 5.0);
    WX_HORNER(1.0 / 3.0);
#undef WX_HORNER
    const double zz = z + z;
    const double lm = fma(zz * w, p, zz);
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
}

// x / nrm for many x and one nrm: rn = RN(1 / nrm) is formed once per signal, then q = RN(x rn),
// q' = RN(q + RN(x - q nrm) rn) -- the residual is exact (fma), so q' is the correctly rounded quotient (Markstein)
// except for a divisor whose significand is all ones; three full-rate instructions instead of the division
// sequence with its reciprocal.
template <typename T> struct WxNorm {
    T nrm, rn;
    __device__ __forceinline__ explicit WxNorm(T n) : nrm(n), rn((T)1 / n) {}
    __device__ __forceinline__ T div(T x) const
    {
        const T q = x * rn;
        return fma(fma(-q, nrm, x), rn, q);
    }
};

template <typename T> __device__ __forceinline__ double bb_term(T x, const WxNorm<T> &nr, int cost_kind)
{
    // coefcost(x::T, et, nrm): s = (x/nrm)^2 in T; Shannon -s log s, log-energy -log s, -0 when s == 0
    const T r = nr.div(x);
    const T s = (T)(r * r);
    if (s == (T)0) return -0.0;
    const T lg = (T)wx_log_pos((double)s);
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
