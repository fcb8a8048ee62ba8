#pragma once
// wx_lattice_8kt.h -- iwpt of 8192-sample Float64 signals ALONG A TREE in one pass (wx_lattice_8k.h + wx_lattice_tree_sc.h): the root is
// split (otherwise there is nothing to do); wavefront c of a workgroup rebuilds child c -- the masked tree kernel of its 4096-sample subtree
// (lat_treesc_inv with T1i as the sink instead of lat_emit), or a plain load in the L0 arrangement when the child is a leaf -- and the two
// synthesise the parent through the shared exchange of wx_lattice_8k.h.  Replaces, for n = 8192, one tree launch per split child + the tiled
// top pass of wx_dev_wpt_long_tree (3-4 n samples of traffic): random trees 0.69 -> 0.56 ms per GiB (0.39 -> 0.48 of the HBM peak).
#include "wx_lattice_8k.h"
#include "wx_host.h"
#include "wx_lattice_tree_sc.h"

namespace {

// (The forward counterpart -- lat8k_child_l0 feeding lat_treesc_fwd through T1 -- was built and is correct, but the register allocator
// spills the 64 registers of the child while it is being computed (105-230 spilled registers at two wavefronts per SIMD, although the
// front-end alone needs 110 and the tree kernel 198 at other times): 0.81 ms per GiB against 0.68 for the tiled pass + tree kernel.
// Forward trees on 8192 samples keep that path.)

template <int NS, int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lat_iwpt_treesc8k_f64(
    const double *__restrict__ xw, double *__restrict__ y, int64_t batch, WxLatW cw, const WxLatTreeSc *__restrict__ tab0,
    const WxLatTreeSc *__restrict__ tab1, WxFilt filt)
{
    __shared__ __attribute__((aligned(16))) double lds2[2][2048];
    __shared__ __attribute__((aligned(16))) double xch[2][2][512 + 16];
    const int child = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)(double __attribute__((address_space(3))) *)lds2[child];
    const int64_t sig = blockIdx.x;
    const double *xs = xw + sig * 8192 + 4096 * child;
    const WxLatTreeSc *tab = child ? tab1 : tab0;
    lat_d2 o[32];
    if (!tab) {
        const unsigned xo = 64u * (lane >> 3) + 2u * (lane & 7);
        lat_for<32>([&](auto Q) {
            constexpr int hi3 = Q / 4, f = Q % 4;
            o[Q] = lat_ld2(lat_sbase(xs + 512 * hi3 + 16 * f) + xo);
        });
    } else {
        const WxThreshArg none{nullptr, 0, 0, 0, 1.0};
        lat_treesc_inv<NS, 0, false, double, false>(xs, (int)sig, lds0, lane, 4096u, 0u, 0u, 1u, cw, tab, none, [&](double (&a)[64]) {
            lat_t1i(a, lds0, lane, [&](auto Fq, lat_d2 (&oo)[8]) {
                constexpr int f = decltype(Fq)::value;
                lat_for<8>([&](auto Hq) { o[4 * Hq + f] = oo[Hq]; });
            });
        });
    }
    lat8k_synth<2 * NS>(o, xch, child, lane, y + sig * 8192, filt);
}

// tables of one child's subtree (status array of 4095 bytes from the child as root, depth L); nullptr when the child is a leaf
template <int DUMMY = 0>
int wx_lat8k_child_tab(const uint8_t *dstatus, int depth, WxScratch &scr, hipStream_t st, const WxLatTreeSc **out)
{
    *out = nullptr;
    if (!dstatus || depth < 1) return WX_OK;
    WxLatTreeSc *tsc = (WxLatTreeSc *)scr.alloc(sizeof(WxLatTreeSc));
    if (!tsc) return WX_EHIP;
    if (hipMemsetAsync(tsc->dep, 0, sizeof(tsc->dep), st) != hipSuccess) return wx_set_error(WX_EHIP, "lattice tree tables");
    hipLaunchKernelGGL((k_lat_treesc_prep<0>), dim3(8), dim3(256), 0, st, dstatus, (int64_t)4095, depth, tsc);
    hipLaunchKernelGGL(k_lat_treesc_prep2, dim3(1), dim3(64), 0, st, tsc);
    *out = tsc;
    return WX_OK;
}

}  // namespace
