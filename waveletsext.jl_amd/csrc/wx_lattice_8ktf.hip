// wx_lattice_8ktf.hip -- wpt of 8192-sample Float64 signals ALONG A TREE in one pass (the forward counterpart of wx_lattice_8kt.h): the root is
// split; wavefront c of a workgroup computes child c of the first level in the direct form of dwt_step! straight into the lattice's load
// arrangement (lat8k_child_l0, wx_lattice_8k.h) and runs the masked tree kernel of the child's 4096-sample subtree on it (lat_treesc_fwd with T1
// as the source instead of lat_absorb), or stores the child as it is when it is a leaf.  Reference: Wavelets.jl's wpt with a tree::BitVector as
// called by wptall (dwt/dwt_all.jl:152-166).
// Register budget: the child's 64 registers are alive while the tree kernel's first layout is filled -- at two wavefronts per SIMD (256
// registers) the allocator spilled 105-230 of them and the kernel lost to the tiled pass + tree kernel; built for ONE wavefront per SIMD (512
// registers, no spills) it does not.
#include "wx_lattice_8kt.h"
bool wx_lattice_factor(const WxFilt &filt, int L, bool inverse, WxLat *out);

namespace {

template <int NS, int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_wpt_treesc8k_f64(
    const double *__restrict__ x, double *__restrict__ y, int64_t batch, WxLatW cw, const WxLatTreeSc *__restrict__ tab0,
    const WxLatTreeSc *__restrict__ tab1, WxFilt filt)
{
    __shared__ __attribute__((aligned(16))) double lds2[2][2048];
    const int child = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds2[child];
    const int64_t sig = blockIdx.x;
    const WxLatTreeSc *tab = child ? tab1 : tab0;
    double *ys = y + sig * 8192 + 4096 * child;
    lat_d2 r[32];
    lat8k_child_l0<2 * NS>(r, x + sig * 8192, lds0, lane, child, filt);
    if (!tab) {
        // a leaf: the child in natural order (the L0 arrangement is 16-byte pieces of it)
        const unsigned xo = 64u * (lane >> 3) + 2u * (lane & 7);
        lat_for<32>([&](auto Q) {
            constexpr int hi3 = Q / 4, f = Q % 4;
            lat_st2(lat_sbase(ys + 512 * hi3 + 16 * f) + xo, r[Q]);
        });
    } else {
        lat_treesc_fwd<NS, 0, double, false>(ys, lds0, lane, 4096u, 0u, cw, tab, [&](double (&a)[64]) { lat_t1(r, a, lds0, lane); });
    }
}

}  // namespace

// x, y: (8192, batch) dense, x != y.  Arguments as wx_lattice_tree8k_inv_f64.  0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_tree8k_fwd_f64(const double *x, double *y, int64_t batch, const WxFilt &filt, const uint8_t *dstatus0, int depth0,
                              const uint8_t *dstatus1, int depth1, hipStream_t st)
{
    static const bool off = (wx_getenv("WX_LATTICE_8K") && atoi(wx_getenv("WX_LATTICE_8K")) == 0) ||
                            (wx_getenv("WX_LATTICE_8KTF") && atoi(wx_getenv("WX_LATTICE_8KTF")) == 0);
    static const int wpe = wx_getenv("WX_LATTICE_8KTF_WPE") ? atoi(wx_getenv("WX_LATTICE_8KTF_WPE")) : 1;
    if (off || filt.F < 2 || filt.F > 16 || (filt.F & 1) || batch <= 0 || batch > 0x7fffffff || x == y || depth0 > 12 || depth1 > 12) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) return 0;
    WxLatW cw;
    if (!wx_lattice_factor(filt, 1, false, &cw.c)) return 0;
    {
        WxLat one;
        if (!wx_lattice_factor(filt, 1, false, &one)) return 0;
        const long double g = one.g0;                           // product of the cosines of one level
        long double acc = 1;
        for (int l = 0; l <= 12; ++l) { cw.gl[l] = (double)acc; acc *= g; }
    }
    cw.tail_bsig = 0;
    WxScratch scr(st);
    const WxLatTreeSc *t0 = nullptr, *t1 = nullptr;
    int rc = wx_lat8k_child_tab(dstatus0, depth0, scr, st, &t0);
    if (rc == WX_OK) rc = wx_lat8k_child_tab(dstatus1, depth1, scr, st, &t1);
    if (rc != WX_OK) return rc;
#define WX_GO8F(NSS)                                                                                                      \
    case NSS:                                                                                                             \
        if (wpe == 2) hipLaunchKernelGGL((k_lat_wpt_treesc8k_f64<NSS, 2>), dim3((unsigned)batch), dim3(128), 0, st, x, y, batch, cw, t0, t1, filt); \
        else hipLaunchKernelGGL((k_lat_wpt_treesc8k_f64<NSS, 1>), dim3((unsigned)batch), dim3(128), 0, st, x, y, batch, cw, t0, t1, filt); \
        break;
    switch (wx_lat_stages(filt.F)) {
        WX_GO8F(1) WX_GO8F(2) WX_GO8F(3) WX_GO8F(4)
    default: return 0;
    }
#undef WX_GO8F
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wx_set_hip_error(e, "lattice tree launch (8192 samples, forward)", __FILE__, __LINE__);
    return 1;
}
