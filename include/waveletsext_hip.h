/*
 * waveletsext_hip.h -- C ABI of libwaveletsext_hip.so, the MI355X (gfx950) drop-in for the
 * batched wavelet-packet hot path of WaveletsExt.jl.
 *
 * The reference has no FFI: its boundary is Julia multiple dispatch.  Each entry point below
 * names the reference method(s) it replaces (paths relative to /root/reference/src/mod); the
 * Julia `ccall` shim that keeps those signatures is in INTEGRATION.md and
 * waveletsext.jl_amd/julia/WaveletsExtHIP.jl.
 *
 * Conventions
 *  - Arrays are dense, column-major, exactly as Julia lays them out; the LAST dimension is the
 *    batch (`*all` drivers: dwt/dwt_all.jl, swt/swt_all.jl, acwt/acwt_all.jl).  batch = 1 gives
 *    the single-signal methods.
 *  - Data pointers may be host or device memory (detected with hipPointerGetAttributes).
 *    Device pointers: the call is asynchronous on `stream`.  Host pointers: the library stages
 *    H2D / D2H itself and returns after the result is in the caller's buffer.
 *  - `qmf` (length F, even, <= 64) is ALWAYS a host pointer to WT.qmf(wt) (Float64); the library
 *    derives the (g, h) pair of WT.makereverseqmfpair(wt, true) and the autocorrelation filters
 *    itself.  `tree` is a host pointer, one byte per node in heap order (a Julia BitVector
 *    converted with Vector{UInt8}); tree == NULL selects the full tree of depth L.
 *    Trees and other small tables are uploaded once per distinct content with a blocking copy and then cached
 *    on the device (released by wx_shutdown), so repeated calls with the same tree stay asynchronous.
 *  - `stream` is a hipStream_t (NULL = default stream).
 *  - Return value: WX_OK, or a negative status.  WX_EASSERT / WX_EARG / WX_EBOUNDS mean the
 *    reference would have thrown AssertionError / ArgumentError / BoundsError for these
 *    arguments; wx_last_error() gives the message.  Nothing throws across the boundary.
 *  - In/out buffers must not alias unless stated.  The library is re-entrant (callable from several host
 *    threads on their own streams); its only state is the per-thread last-error string, the cached scratch
 *    pool, the cache of small constant tables (both per device) and, for host-array calls, one pinned
 *    staging ring with its copy threads -- all released by wx_shutdown().  examples/roundtrip.c drives it from plain C.
 *  - Host arrays of 64 MiB and more: inputs travel through the pinned ring (host threads fill one buffer while the
 *    DMA engine drains the other), result arrays receive madvise(MADV_HUGEPAGE) before their first touch so that a
 *    freshly allocated array faults in 2 MiB steps (the advice persists on the address range: wx_set_host_hugepages).
 *  - Device arrays should start on a 32-byte boundary (what hipMalloc, CuArray-style allocators and whole columns of dyadic length
 *    give); a device pointer that does not -- a view shifted by a few elements -- is taken through an aligned scratch copy on the
 *    call's stream: correct, two device-to-device copies slower.
 *  - Dispatch does not depend on the environment: the library's WX_* tuning knobs (profiles/NOTES.md, DESIGN.md
 *    section 10) are read only when the process also sets WX_KNOBS=1.  The parity suite's dispatch override lives in
 *    csrc/wx_debug.h, outside this header.
 */
#ifndef WAVELETSEXT_HIP_H
#define WAVELETSEXT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WX_OK 0
#define WX_EASSERT (-1)
#define WX_EARG (-2)
#define WX_EBOUNDS (-3)
#define WX_EHIP (-10)
#define WX_EUNSUPPORTED (-11)

/* library */
int wx_version(void);                       /* 10000*major + 100*minor + patch */
const char *wx_last_error(void);            /* message of the last failing call on this thread */
int wx_device_count(void);                  /* number of visible HIP devices (0 if none) */
/* what this binary was built from: "libwaveletsext_hip <version> src=<digest of the kernel sources> git=<commit>[+dirty]
 * arch=<gfx target> hip=<HIP version> clang=<compiler> flags=[...]" (static string; bench.py prints it in its line) */
const char *wx_build_info(void);
/* releases the library's only state, the cached stream-ordered scratch of the current device */
int wx_shutdown(void);
/* host-array results of 64 MiB and more are advised MADV_HUGEPAGE before their first touch (on by default: 17 -> 48 GB/s of host bytes).
 * The advice is a property of the caller's ADDRESS RANGE, not of the call: it stays after the call returns and after the array is freed
 * and its range reused, it splits the mapping it falls into, and khugepaged may collapse pages there later.  on = 0 switches it off for
 * the process (results that are also inputs of the same call are never advised); returns the previous setting. */
int wx_set_host_hugepages(int on);

/* ------------------------------------------------------------------------------------------
 * 1-D decimated wavelet packets
 * ------------------------------------------------------------------------------------------ */

/* wpd!(y, x, wt, L) DWT.jl:131-161 and wpdall(x, wt, L) dwt/dwt_all.jl:260-282.
 * x: (n, batch); y: (n, L+1, batch).  Requires 0 <= L <= maxtransformlevels(n). */
int wx_wpd1d_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_wpd1d_f32(const float *x, float *y, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);

/* wpt!(y, x, wt, L | tree) -- Wavelets.jl 1-D method as called by wptall, dwt/dwt_all.jl:152-166.
 * x, y: (n, batch).  n must be dyadic (Wavelets.jl maketree). */
int wx_wpt1d_f64(const double *x, double *y, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                 const double *qmf, int F, void *stream);
int wx_wpt1d_f32(const float *x, float *y, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                 const double *qmf, int F, void *stream);

/* iwpt!(xhat, xw, wt, L | tree) -- Wavelets.jl 1-D method as called by iwptall,
 * dwt/dwt_all.jl:210-225.  xw, xhat: (n, batch). */
int wx_iwpt1d_f64(const double *xw, double *xhat, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream);
int wx_iwpt1d_f32(const float *xw, float *xhat, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream);

/* iwpd!(xhat, xw, wt, L | tree) DWT.jl:322-351 and iwpdall dwt/dwt_all.jl:324-342.
 * xw: (n, k, batch) packet table with k = levels + 1 columns; xhat: (n, batch). */
int wx_iwpd1d_f64(const double *xw, double *xhat, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream);
int wx_iwpd1d_f32(const float *xw, float *xhat, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream);

/* getbasiscoef(Xw, tree) Utils.jl:101-134 / getbasiscoefall(Xw, tree::BitVector) Utils.jl:169-197,
 * 1-D signals.  Xw: (n, k, batch); out: (n, batch). */
int wx_getbasiscoef1d_f64(const double *Xw, double *out, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                          int64_t batch, void *stream);
int wx_getbasiscoef1d_f32(const float *Xw, float *out, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                          int64_t batch, void *stream);


/* ------------------------------------------------------------------------------------------
 * 2-D decimated wavelet packets (quad trees, heap order, children 4i-2..4i+1 =
 * top-left, top-right, bottom-left, bottom-right; utils/utils_tree.jl:57-75)
 * Images are (m, n) column-major; batch is the last dimension.
 * ------------------------------------------------------------------------------------------ */
/* wpd!(y, x, wt, L) DWT.jl:164-209 / wpdall dwt/dwt_all.jl:260-282.  x (m,n,batch) -> y (m,n,L+1,batch) */
int wx_wpd2d_f64(const double *x, double *y, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_wpd2d_f32(const float *x, float *y, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
/* wpt!(y, x, wt, L | tree) DWT.jl:493-548 / wptall dwt/dwt_all.jl:152-166 */
int wx_wpt2d_f64(const double *x, double *y, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                 const double *qmf, int F, void *stream);
int wx_wpt2d_f32(const float *x, float *y, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree, int64_t batch,
                 const double *qmf, int F, void *stream);
/* iwpt!(xhat, xw, wt, L | tree) DWT.jl:655-710 / iwptall dwt/dwt_all.jl:210-225 */
int wx_iwpt2d_f64(const double *xw, double *xhat, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream);
int wx_iwpt2d_f32(const float *xw, float *xhat, int64_t m, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream);
/* iwpd!(xhat, xw, wt, L | tree) DWT.jl:331-337,354-401 / iwpdall; xw (m,n,k,batch) */
int wx_iwpd2d_f64(const double *xw, double *xhat, int64_t m, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream);
int wx_iwpd2d_f32(const float *xw, float *xhat, int64_t m, int64_t n, int k, int L, const uint8_t *tree, int64_t ntree,
                  int64_t batch, const double *qmf, int F, void *stream);

/* ------------------------------------------------------------------------------------------
 * 1-D stationary (undecimated) transforms -- SWT.jl, swt/swt_all.jl
 * `sm` is the shift of the shift-based inverse (SWT.jl:259-284, 613-646, 1063-1093);
 * sm < 0 selects the average-based inverse (SWT.jl:313-330, 685-712, 1137-1160).
 * ------------------------------------------------------------------------------------------ */

/* sdwt!(xw, x, wt, L) SWT.jl:109-130 / sdwtall swt_all.jl:33.  x (n,batch) -> xw (n, L+1, batch) = [s_L d_L .. d_1] */
int wx_sdwt1d_f64(const double *x, double *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_sdwt1d_f32(const float *x, float *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
/* isdwt!(x, xw, wt[, sm]) SWT.jl:259-330 / isdwtall swt_all.jl:89,107 */
int wx_isdwt1d_f64(const double *xw, double *x, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
int wx_isdwt1d_f32(const float *xw, float *x, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
/* swpt!(xw, x, wt, L) SWT.jl:439-472 / swptall swt_all.jl:156-176.  xw (n, 2^L, batch), natural order */
int wx_swpt1d_f64(const double *x, double *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_swpt1d_f32(const float *x, float *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
/* iswpt!(x, xw, wt[, sm]) SWT.jl:613-712 / iswptall swt_all.jl:212,230; xw has 2^L columns */
int wx_iswpt1d_f64(const double *xw, double *x, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
int wx_iswpt1d_f32(const float *xw, float *x, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
/* swpd!(xw, x, wt, L) SWT.jl:840-868 / swpdall swt_all.jl:279-299.  xw (n, 2^(L+1)-1, batch), heap order */
int wx_swpd1d_f64(const double *x, double *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_swpd1d_f32(const float *x, float *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
/* iswpd!(x, xw, wt, L | tree[, sm]) SWT.jl:1035-1160 / iswpdall swt_all.jl:343-392; xw has ncols columns */
int wx_iswpd1d_f64(const double *xw, double *x, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                   int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
int wx_iswpd1d_f32(const float *xw, float *x, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                   int64_t sm, int64_t batch, const double *qmf, int F, void *stream);

/* ------------------------------------------------------------------------------------------
 * 1-D autocorrelation wavelet transforms -- ACWT.jl, acwt/acwt_all.jl (Float64 only: the
 * reference's acdwt_step! requires eltype(filter) == eltype(data), acwt_one_level.jl:101-106)
 * ------------------------------------------------------------------------------------------ */
/* acdwt! ACWT.jl:109-129 / acdwtall acwt_all.jl:33 */
int wx_acdwt1d_f64(const double *x, double *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
/* iacdwt! ACWT.jl:287-304 / iacdwtall */
int wx_iacdwt1d_f64(const double *xw, double *x, int64_t n, int L, int64_t batch, void *stream);
/* acwpt! ACWT.jl:427-460 / acwptall acwt_all.jl:136 */
int wx_acwpt1d_f64(const double *x, double *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
/* iacwpt! ACWT.jl:581-610 / iacwptall */
int wx_iacwpt1d_f64(const double *xw, double *x, int64_t n, int L, int64_t batch, void *stream);
/* acwpd! ACWT.jl:733-759 / acwpdall acwt_all.jl:239 */
int wx_acwpd1d_f64(const double *x, double *xw, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
/* iacwpd!(x, xw, L | tree) ACWT.jl:917-968 / iacwpdall acwt_all.jl:300-333 */
int wx_iacwpd1d_f64(const double *xw, double *x, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                    int64_t batch, void *stream);

/* ------------------------------------------------------------------------------------------
 * Joint best basis (JBB) -- bestbasis/bestbasis_tree.jl:150-180, BestBasis.jl:59-83,194-201
 * ------------------------------------------------------------------------------------------ */
/* sum[e] = sum_b X[e,b], sumsq[e] = sum_b X[e,b]^2 over the last (signal) axis; X is (nk, batch)
 * with nk = n*k the flattened (coef, column) index.  accumulate != 0 adds to the existing
 * contents (batch shards / RCCL all-reduce partials). */
int wx_jbb_moments_f64(const double *X, double *sum, double *sumsq, int64_t nk, int64_t batch, int accumulate, void *stream);
int wx_jbb_moments_f32(const float *X, float *sum, float *sumsq, int64_t nk, int64_t batch, int accumulate, void *stream);
/* tree_costs(X, JBB(cost, redundant)) from the moments of Ntot signals.  cost_kind 0 = LoglpCost(p),
 * 1 = NormCost(p).  costs has k entries (redundant) or 2^k - 1 (wpd table with k columns). */
int wx_jbb_costs_f64(const double *sum, const double *sumsq, int64_t Ntot, int64_t n, int64_t k, int redundant,
                     int cost_kind, double p, double *costs, void *stream);
int wx_jbb_costs_f32(const float *sum, const float *sumsq, int64_t Ntot, int64_t n, int64_t k, int redundant,
                     int cost_kind, double p, float *costs, void *stream);
/* bestbasis_treeselection(costs, n, :min | :max) BestBasis.jl:59-83.  HOST pointers only; costs
 * (k entries) is mutated like the reference; tree receives n-1 bytes. */
int wx_treeselect_f64(double *costs, int64_t k, int64_t n, int type_max, uint8_t *tree);
int wx_treeselect_f32(float *costs, int64_t k, int64_t n, int type_max, uint8_t *tree);
/* The same selection, and *min_rel_gap = min over the split decisions taken (nodes still in the tree when visited,
 * BestBasis.jl:70-76) of |cc - pc| / |pc|: the margin by which the closest `cc < pc` was decided.  The reference prunes on
 * a strict Float64 comparison of log-sums, so a tree is reproducible across summation orders only while this margin is
 * far above the costs' rounding error (~1e-13); +inf when no decision was taken (L = 0). */
int wx_treeselect_gap_f64(double *costs, int64_t k, int64_t n, int type_max, uint8_t *tree, double *min_rel_gap);
int wx_treeselect_gap_f32(float *costs, int64_t k, int64_t n, int type_max, uint8_t *tree, double *min_rel_gap);
/* acwpdall + the JBB moments of its output without materialising the (n, 2^(L+1)-1, batch)
 * table (BASELINE config 5): sum / sumsq are (n, 2^(L+1)-1). */
int wx_acwpd_jbb_moments_f64(const double *x, double *sum, double *sumsq, int64_t n, int L, int64_t batch,
                             const double *qmf, int F, int accumulate, void *stream);

/* ------------------------------------------------------------------------------------------
 * 2-D redundant transforms.  Images (m, n) column-major, batch last.  Slices per image:
 * sdwt/acdwt 3L+1 (SWT.jl:132-158), swpt/acwpt 4^L natural order (SWT.jl:474-513), swpd/acwpd
 * (4^(L+1)-1)/3 in quad-heap order (SWT.jl:870-902).  sm as in the 1-D entry points.
 * ------------------------------------------------------------------------------------------ */
int wx_sdwt2d_f64(const double *x, double *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_sdwt2d_f32(const float *x, float *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_swpt2d_f64(const double *x, double *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_swpt2d_f32(const float *x, float *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_swpd2d_f64(const double *x, double *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_swpd2d_f32(const float *x, float *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
/* isdwt! 2-D SWT.jl:286-358; iswpt! 2-D :648-758; iswpd! 2-D :1095-1199 */
int wx_isdwt2d_f64(const double *xw, double *x, int64_t m, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
int wx_isdwt2d_f32(const float *xw, float *x, int64_t m, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
int wx_iswpt2d_f64(const double *xw, double *x, int64_t m, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
int wx_iswpt2d_f32(const float *xw, float *x, int64_t m, int64_t n, int L, int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
int wx_iswpd2d_f64(const double *xw, double *x, int64_t m, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                   int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
int wx_iswpd2d_f32(const float *xw, float *x, int64_t m, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                   int64_t sm, int64_t batch, const double *qmf, int F, void *stream);
/* acdwt!/acwpt!/acwpd! 2-D ACWT.jl:131-157, 462-501, 761-793 and inverses :306-329, 612-648, 970-1000 (Float64) */
int wx_acdwt2d_f64(const double *x, double *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_acwpt2d_f64(const double *x, double *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_acwpd2d_f64(const double *x, double *xw, int64_t m, int64_t n, int L, int64_t batch, const double *qmf, int F, void *stream);
int wx_iacdwt2d_f64(const double *xw, double *x, int64_t m, int64_t n, int L, int64_t batch, void *stream);
int wx_iacwpt2d_f64(const double *xw, double *x, int64_t m, int64_t n, int L, int64_t batch, void *stream);
int wx_iacwpd2d_f64(const double *xw, double *x, int64_t m, int64_t n, int64_t ncols, int L, const uint8_t *tree, int64_t ntree,
                    int64_t batch, void *stream);

/* tree_costs(X::Array{T,4}, ::JBB) bestbasis_tree.jl:182-207 from moments (wx_jbb_moments_* with nk = m*n*k);
 * costs has k entries (redundant) or (4^k - 1)/3 (wpd table with k slices) */
int wx_jbb_costs2d_f64(const double *sum, const double *sumsq, int64_t Ntot, int64_t m, int64_t n, int64_t k, int redundant,
                       int cost_kind, double p, double *costs, void *stream);
int wx_jbb_costs2d_f32(const float *sum, const float *sumsq, int64_t Ntot, int64_t m, int64_t n, int64_t k, int redundant,
                       int cost_kind, double p, float *costs, void *stream);
/* bestbasis_treeselection(costs, n, m, type) BestBasis.jl:85-110 (HOST pointers; tree gets gettreelength(m,n) bytes) */
int wx_treeselect2d_f64(double *costs, int64_t k, int64_t m, int64_t n, int type_max, uint8_t *tree);
int wx_treeselect2d_f32(float *costs, int64_t k, int64_t m, int64_t n, int type_max, uint8_t *tree);
/* getbasiscoef / getbasiscoefall for 2-D signals Utils.jl:127-130, 192-195.  Xw (m,n,k,batch) -> out (m,n,batch) */
int wx_getbasiscoef2d_f64(const double *Xw, double *out, int64_t m, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                          int64_t batch, void *stream);
int wx_getbasiscoef2d_f32(const float *Xw, float *out, int64_t m, int64_t n, int k, const uint8_t *tree, int64_t ntree,
                          int64_t batch, void *stream);
/* getbasiscoefall(Xw, tree::BitArray{2}) Utils.jl:199-225: ONE tree per signal -- what bestbasistreeall (BestBasis.jl:253-262)
 * returns.  trees: HOST pointer, (ntree, batch) bytes, column b = the tree of signal b (a Julia BitMatrix converted with
 * Matrix{UInt8}); every tree is checked like the reference does (:209), one launch gathers all signals.
 * 1-D: Xw (n, k, batch) -> out (n, batch); 2-D: Xw (m, n, k, batch) -> out (m, n, batch). */
int wx_getbasiscoef1d_trees_f64(const double *Xw, double *out, int64_t n, int k, const uint8_t *trees, int64_t ntree, int64_t batch,
                                void *stream);
int wx_getbasiscoef1d_trees_f32(const float *Xw, float *out, int64_t n, int k, const uint8_t *trees, int64_t ntree, int64_t batch,
                                void *stream);
int wx_getbasiscoef2d_trees_f64(const double *Xw, double *out, int64_t m, int64_t n, int k, const uint8_t *trees, int64_t ntree,
                                int64_t batch, void *stream);
int wx_getbasiscoef2d_trees_f32(const float *Xw, float *out, int64_t m, int64_t n, int k, const uint8_t *trees, int64_t ntree,
                                int64_t batch, void *stream);


/* ------------------------------------------------------------------------------------------
 * Standard (per-signal) best basis, BB -- SURVEY 8(f) row 3, widened after the 8(a) rows.
 * tree_costs(X, BB(cost, redundant)) bestbasis/bestbasis_tree.jl:210-258 with coefcost(x, cost, nrm)
 * bestbasis/bestbasis_costs.jl:104-125: cost_kind 0 = ShannonEntropyCost, 1 = LogEnergyEntropyCost;
 * nrm = norm of the signal (first column / slice); the non-redundant 2-D branch normalises every block by
 * its own norm like the reference (:252).  X (n, k, batch) -> costs (ncost, batch), ncost = k (redundant)
 * or 2^k - 1; 2-D X (m, n, k, batch), ncost = k or (4^k - 1)/3.
 * wx_treeselect_batch_*: bestbasis_treeselection (BestBasis.jl:59-110) for every signal at once, i.e. the
 * loop of bestbasistreeall(X, BB()) (BestBasis.jl:253-262): costs (ncost, batch) are mutated like the
 * reference, trees (n - 1 | gettreelength(m, n), batch) one byte per node; n = 0 selects the 1-D (binary)
 * tree of an m-sample signal.  Pointers may be host or device.
 * ------------------------------------------------------------------------------------------ */
int wx_bb_costs_f64(const double *X, double *costs, int64_t n, int64_t k, int64_t batch, int redundant, int cost_kind,
                    void *stream);
int wx_bb_costs_f32(const float *X, float *costs, int64_t n, int64_t k, int64_t batch, int redundant, int cost_kind,
                    void *stream);
int wx_bb_costs2d_f64(const double *X, double *costs, int64_t m, int64_t n, int64_t k, int64_t batch, int redundant,
                      int cost_kind, void *stream);
int wx_bb_costs2d_f32(const float *X, float *costs, int64_t m, int64_t n, int64_t k, int64_t batch, int redundant,
                      int cost_kind, void *stream);
int wx_treeselect_batch_f64(double *costs, int64_t ncost, int64_t m, int64_t n, int type_max, int64_t batch,
                            uint8_t *trees, void *stream);
int wx_treeselect_batch_f32(float *costs, int64_t ncost, int64_t m, int64_t n, int type_max, int64_t batch,
                            uint8_t *trees, void *stream);

/* ------------------------------------------------------------------------------------------
 * Denoising core -- SURVEY 8(f) row 1: the two device steps between a forward and an inverse batch transform
 * (denoise / denoiseall, Denoising.jl:483-712).  X (n, k, batch) is any coefficient container of this header:
 * k = 1 for dwt / wpt leaves, L+1 for sdwt / acdwt, 2^(L+1)-1 for swpd / acwpd.
 * wx_noisest_*: noisest(x, redundant, tree) Denoising.jl:214-232 = Wavelets.Threshold.mad!(dr)/0.6745 for every
 * signal, dr = rows [row_lo, n) of column `col` (the caller resolves finestdetailrange, Utils.jl:416-436);
 * exact order statistics (radix select in LDS; over a global-memory copy for more than 128 KiB of details), sigma has `batch` entries.
 * wx_threshold_*: Y = Wavelets.Threshold.threshold(X, TH, t) on rows [row_lo, n) of the columns with
 * colmask != 0 (NULL = all), everything else copied; Y == X thresholds in place (threshold!) and touches only the
 * selected elements: th_kind 0 HardTH, 1 SoftTH, 2 SemiSoftTH, 3 SteinTH; t holds nt = 1 or `batch`
 * thresholds (sigma_i * dnt.t).  Wavelets.jl is not vendored: mad! and the threshold loops are restated from
 * its source.  Pointers may be host or device.
 * ------------------------------------------------------------------------------------------ */
int wx_noisest_f64(const double *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, double *sigma,
                   void *stream);
int wx_noisest_f32(const float *X, int64_t n, int64_t k, int64_t batch, int64_t row_lo, int64_t col, float *sigma,
                   void *stream);
int wx_threshold_f64(const double *X, double *Y, int64_t n, int64_t k, int64_t batch, int th_kind, const double *t,
                     int64_t nt, int64_t row_lo, const uint8_t *colmask, void *stream);
int wx_threshold_f32(const float *X, float *Y, int64_t n, int64_t k, int64_t batch, int th_kind, const float *t,
                     int64_t nt, int64_t row_lo, const uint8_t *colmask, void *stream);

/* Order statistics over the signal axis for Local Discriminant Basis (X: (nk, N) column-major, cls[i] in [0, nc) the class
 * of signal i, as wx_class_mean_*):
 * wx_class_median_mad_*: med[e, c] = median, mad[e, c] = mad(normalize = false) of X[e, class c] -- the two tables of
 *     discriminant_power(coefs, y, RobustFishersClassSeparability()), ldb/ldb_measures.jl:481-519;
 * wx_emd_measure_*:      D[e] = sum over the class pairs of the earth mover's distance between the signatures
 *     (X[e, class a], weight 1/N_a) and (X[e, class b], weight 1/N_b): discriminant_measure(energy_map(Xw, y,
 *     Signatures(:equal)), EarthMoverDistance()), ldb/ldb_energymap.jl:186-238, ldb/ldb_measures.jl:185-201, 254-360.
 * Any number of signals and classes: the padded signals of one coefficient are sorted in a 128 KiB LDS window while they fit (sum
 * over classes of nextpow2(N_c) <= 16384 Float64 / 32768 Float32 values) and in a global-memory window beyond that.  Pointers may
 * be host or device; cls is a host array. */
int wx_class_median_mad_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *med, double *mad,
                            void *stream);
int wx_class_median_mad_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, float *med, float *mad,
                            void *stream);
int wx_emd_measure_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *D, void *stream);
int wx_emd_measure_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, float *D, void *stream);

/* Average-shifted-histogram densities over the signal axis (same X / cls convention as above).  `ash`, `xy`, `pdf` belong to
 * AverageShiftedHistograms.jl (Project.toml compat "0.8, 0.9"), which is not in the reference tree: its published
 * algorithm is restated (csrc/wx_ldbstat.hip), parity unpinned.
 * wx_pdf_energy_map_*:        Gamma (nk, pdf_len, nc) Float64 = energy_map(Xw, y, ProbabilityDensity()),
 *     ldb/ldb_energymap.jl:143-184; pdf_len = (nbins + 1) mbins, nbins = ceil((30 N)^(1/5)), mbins = ceil(100 / nbins);
 * wx_signature_weights_*:     W (nk, N) = the :pdf weights of energy_map(Xw, y, Signatures(:pdf)), ldb_energymap.jl:216-232;
 * wx_emd_measure_weighted_*:  D[e] = sum over the class pairs of the earth mover's distance between the signatures
 *     (X[e, class], W[e, class]), ldb/ldb_measures.jl:185-201, 254-360. */
int wx_pdf_energy_map_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *Gamma, void *stream);
int wx_pdf_energy_map_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *Gamma, void *stream);
int wx_signature_weights_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *W, void *stream);
int wx_signature_weights_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, float *W, void *stream);
int wx_emd_measure_weighted_f64(const double *X, const double *W, int64_t nk, int64_t N, const int32_t *cls, int nc,
                                double *D, void *stream);
int wx_emd_measure_weighted_f32(const float *X, const float *W, int64_t nk, int64_t N, const int32_t *cls, int nc,
                                float *D, void *stream);

/* 3-D discrete wavelet transform of a batch of cubes: dwtall / idwtall on 4-D arrays (dwt/dwt_all.jl:39-54, 95-110 over
 * Wavelets.jl's 3-D dwt! / idwt!, which is not vendored: the separable pyramid -- one analysis step along dimension 1,
 * 2, 3 of the low-pass sub-cube per level -- is restated from its published source).  x, y: (n1, n2, n3, batch)
 * column-major, n1 == n2 == n3 dyadic, 0 <= L <= maxtransformlevels(n1); y may be x.  Pointers may be host or device. */
int wx_dwt3d_f64(const double *x, double *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf,
                 int F, void *stream);
int wx_dwt3d_f32(const float *x, float *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf,
                 int F, void *stream);
int wx_idwt3d_f64(const double *x, double *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf,
                  int F, void *stream);
int wx_idwt3d_f32(const float *x, float *y, int64_t n1, int64_t n2, int64_t n3, int L, int64_t batch, const double *qmf,
                  int F, void *stream);

/* denoiseall(x, :sig, wt; L, dnt, estnoise = noisest, smooth) Denoising.jl:651-712 (denoise of one signal, Denoising.jl:483-599: batch = 1) for the
 * VisuShrink family: xhat[:, i] = idwt(threshold(dwt(x[:, i], wt, L), th_kind, sigma_i * t), wt, L) with sigma_i = noisest(dwt(x[:, i])) =
 * mad(finest detail coefficients) / 0.6745 (Denoising.jl:214-232), t = dnt.t (VisuShrink: sqrt(2 log n)), th_kind as wx_threshold_*.
 * undersmooth != 0 (smooth = :undersmooth) leaves the coarsest scaling coefficients alone.  sigma (optional, host or device, batch values)
 * receives the noise estimates.  Float64 signals of 1024 ... 4096 samples with filters of up to 8 taps and the Hard / Soft / SemiSoft rules
 * take ONE pass (signal in, denoised signal out: the coefficients never leave the registers); every other case runs the three steps
 * wx_wpt1d_* -> wx_noisest_* -> wx_iwpt1d_thresh_* on stream-ordered scratch, so the result does not depend on which applies. */
int wx_denoiseall_sig_f64(const double *x, double *xhat, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind,
                          double t, int undersmooth, double *sigma, void *stream);
int wx_denoiseall_sig_f32(const float *x, float *xhat, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind,
                          double t, int undersmooth, float *sigma, void *stream);

/* denoiseall(xw, :dwt, wt; L, dnt, estnoise = noisest, smooth) Denoising.jl:651-712: as wx_denoiseall_sig_* with the coefficients xw = dwtall(x, wt, L)
 * as the input -- sigma_i = noisest(xw[:, i]), xhat[:, i] = idwt(threshold(xw[:, i], ...), wt, L).  One pass (coefficients in, signals out) under the
 * conditions of wx_denoiseall_sig_*, else wx_noisest_* -> wx_iwpt1d_thresh_*.  xw is not modified. */
int wx_denoiseall_dwt_f64(const double *xw, double *xhat, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind,
                          double t, int undersmooth, double *sigma, void *stream);
int wx_denoiseall_dwt_f32(const float *xw, float *xhat, int64_t n, int L, int64_t batch, const double *qmf, int F, int th_kind,
                          double t, int undersmooth, float *sigma, void *stream);

/* iwpt / idwt with the thresholding step of denoise() applied while the coefficients are loaded (Denoising.jl:510-533:
 * threshold(x, dnt.th, sigma * dnt.t) followed by idwt / iwpt): one pass over the coefficient array instead of two.
 * Arguments as wx_iwpt1d_* plus those of wx_threshold_* (k = 1): rows [row_lo, n) of signal i are thresholded with
 * scale * t[0] (nt = 1) or scale * t[i] (nt = batch) -- t may be the device vector wx_noisest_* just wrote and scale =
 * dnt.t, so that the noise estimates never visit the host.  xw is not modified. */
int wx_iwpt1d_thresh_f64(const double *xw, double *xhat, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                         int64_t batch, const double *qmf, int F, int th_kind, const double *t, int64_t nt,
                         int64_t row_lo, double scale, void *stream);
int wx_iwpt1d_thresh_f32(const float *xw, float *xhat, int64_t n, int L, const uint8_t *tree, int64_t ntree,
                         int64_t batch, const double *qmf, int F, int th_kind, const float *t, int64_t nt,
                         int64_t row_lo, double scale, void *stream);

/* Threshold selection of SureShrink and RelErrorShrink for every signal of the batch (they are what
 * denoiseall(...; estnoise = relerrorthreshold) and the SureShrink(xw, redundant, tree) constructor evaluate):
 * wx_surethreshold_*:     surethreshold(coef, redundant, tree)             Denoising.jl:146-166
 * wx_relerrorthreshold_*: relerrorthreshold(coef, redundant, tree, elbows) Denoising.jl:285-327 (orth2relerror
 *                         :344-349, findelbow :367-381; the plot of makeplot = true is not part of the path)
 * over all n rows of the columns with colmask != 0 (NULL = all k columns: the caller resolves getleaf(tree) for
 * swpd / acwpd tables); t has `batch` entries.  One workgroup per signal: bitonic sort of the magnitudes, workgroup
 * scan, first-index argmin / argmax.  Pointers may be host or device. */
int wx_surethreshold_f64(const double *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, double *t,
                         void *stream);
int wx_surethreshold_f32(const float *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, float *t,
                         void *stream);
int wx_relerrorthreshold_f64(const double *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, int elbows,
                             double *t, void *stream);
int wx_relerrorthreshold_f32(const float *X, int64_t n, int64_t k, int64_t batch, const uint8_t *colmask, int elbows,
                             float *t, void *stream);

/* ------------------------------------------------------------------------------------------
 * Local Discriminant Basis, the batch-sized steps -- SURVEY 8(f) row 2.
 * wx_energy_map_*: energy_map(Xw, y, TimeFrequency()) ldb/ldb_energymap.jl:109-141.  Xw is a packet table
 * (n, k, N) or (m, n, k, N) passed flat: nk = elements per signal, nroot = elements of the root column/slice
 * (n or m*n), cls[i] in [0, nc) = index of signal i's class in unique(y) order.  Gamma (nk, nc):
 * Gamma[e, c] = sum_{i in c} Xw[e, i]^2 / sum_{i in c} norm(x_i)^2.  norm_sum (nc entries, may be NULL) returns the
 * denominators: a batch sharded over GPUs combines its shards as sum_r Gamma_r * norm_sum_r / sum_r norm_sum_r
 * (two all-reduces of small arrays); with norm_sum given a class may be empty on this shard (its Gamma is NaN).
 * wx_class_mean_* / wx_class_var_*: per-class mean and variance (two passes, n-1 denominator, like Julia's
 * mean / var over the signal axis) of X (nk, N) for FishersClassSeparability, ldb/ldb_measures.jl:441-479.
 * The rest of fitdec! (LDB.jl:186-251: discriminant measure on the small class maps, node costs with top_k,
 * wx_treeselect_* with type max, ordering) is host logic on small arrays; the feature gather of transform
 * (LDB.jl:300-305) is wx_wpt*_ / wx_getbasiscoef* plus an index selection.  Pointers may be host or device,
 * cls is a host array.
 * ------------------------------------------------------------------------------------------ */
int wx_energy_map_f64(const double *Xw, int64_t nk, int64_t nroot, int64_t N, const int32_t *cls, int nc, double *Gamma,
                      double *norm_sum, void *stream);
int wx_energy_map_f32(const float *Xw, int64_t nk, int64_t nroot, int64_t N, const int32_t *cls, int nc, float *Gamma,
                      float *norm_sum, void *stream);
int wx_class_mean_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, double *mean, void *stream);
int wx_class_mean_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, float *mean, void *stream);
int wx_class_var_f64(const double *X, int64_t nk, int64_t N, const int32_t *cls, int nc, const double *mean, double *var,
                     void *stream);
int wx_class_var_f32(const float *X, int64_t nk, int64_t N, const int32_t *cls, int nc, const float *mean, float *var,
                     void *stream);

/* ------------------------------------------------------------------------------------------
 * Shift-invariant wavelet packet decomposition (Cohen, Raz & Malah) -- SURVEY 8(f) row 4.
 * The reference's ShiftInvariantWaveletTransformObject (siwt/siwt_utls.jl:75-90, a Dict of node objects per
 * signal) is a flat table here: W is (n, NS, batch), NS = wx_siwt_ncols(L, d) = sum_j 2^min(j,d).  Depth j owns
 * the 2^min(j,d) columns that start at sum_{i<j} 2^min(i,d); column `slot` of depth j is TransformShift
 * t = slot << max(0, j-d) (the only shifts siwpd_subtree! creates, SIWT.jl:104-131), and the node (j, i, t) is
 * rows [i*(n>>j), (i+1)*(n>>j)) of that column.  Costs / status use one entry per node: index
 * sum_{i<j} 2^i 2^min(i,d) + (slot << j) + i, NN = wx_siwt_nnodes(L, d) per signal.
 * wx_siwpd_*          siwpd(x, wt, L, d) SIWT.jl:57-69 for every signal of x (n, batch); costs (NN, batch, may be
 *                     NULL) = Nodes[index].Cost: coefcost(Value, ShannonEntropyCost(), norm(x)) (siwt_utls.jl:118-126).
 * wx_siwt_bestbasis_* bestbasistree!(siwtObj) siwt/siwt_bestbasis.jl:28-102 per signal: costs are updated in place
 *                     (Nodes[index].Cost, MinCost = costs[0]); status (NN, batch): 0 = node deleted, 1 = leaf of the
 *                     best tree, 2 = kept with its non-shifted children, 3 = kept with its shifted children.
 * wx_isiwpd_*         isiwpd(siwtObj) SIWT.jl:166-229: merges children into parents bottom-up along `status`,
 *                     overwriting the parents' rows of W like the reference overwrites Nodes[index].Value; xh (n, batch).
 *                     literal = 0: the inverse step is told `shifted` exactly when the children came from the shifted
 *                     step (the reading under which the reference's test isiwpd(siwtObj) ~ signal holds); literal = 1:
 *                     the flag as siwt/siwt_one_level.jl:126 spells it (true for the NON-shifted children).
 * Errors: WX_EASSERT for 0 <= L <= maxtransformlevels(n), 1 <= d <= L (SIWT.jl:62-63).
 * ------------------------------------------------------------------------------------------ */
int64_t wx_siwt_ncols(int L, int d);
int64_t wx_siwt_nnodes(int L, int d);
int wx_siwpd_f64(const double *x, double *W, double *costs, int64_t n, int L, int d, int64_t batch, const double *qmf,
                 int F, void *stream);
int wx_siwpd_f32(const float *x, float *W, float *costs, int64_t n, int L, int d, int64_t batch, const double *qmf,
                 int F, void *stream);
int wx_siwt_bestbasis_f64(double *costs, uint8_t *status, int L, int d, int64_t batch, void *stream);
int wx_siwt_bestbasis_f32(float *costs, uint8_t *status, int L, int d, int64_t batch, void *stream);
int wx_isiwpd_f64(double *W, const uint8_t *status, double *xh, int64_t n, int L, int d, int64_t batch, const double *qmf,
                  int F, int literal, void *stream);
int wx_isiwpd_f32(float *W, const uint8_t *status, float *xh, int64_t n, int L, int d, int64_t batch, const double *qmf,
                  int F, int literal, void *stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU exchange (one process per GPU, RCCL over xGMI; bound lazily, single-GPU callers never
 * load RCCL).  Transforms shard over the batch (last) dimension with no collective: the loops
 * dwt/dwt_all.jl:277-279, swt/swt_all.jl:171-173, acwt/acwt_all.jl:254-256 are independent per
 * signal.  The only exchange steps are
 *   C1 all-gather of the reconstructed output shards (iwpdall / iwptall ... results), and
 *   C2 all-reduce (sum) of the JBB moments [sum | sumsq] before wx_jbb_costs_* -- the means over
 *      all signals of bestbasis/bestbasis_tree.jl:153-154.
 * id128: 128-byte RCCL unique id made by rank 0 (wx_comm_unique_id) and broadcast by the launcher
 * (MPI.jl / Distributed.jl / torch.distributed).  Buffers are device pointers; `count` elements per
 * rank (equal on every rank; wx_allgatherv_out_* takes ragged shards), recv holds nranks*count; asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------ */
int wx_comm_unique_id(void *id128);
int wx_comm_init(int nranks, int rank, const void *id128, void **comm);
int wx_comm_destroy(void *comm);
int wx_allgather_out_f64(const double *send, double *recv, int64_t count, void *comm, void *stream);
int wx_allgather_out_f32(const float *send, float *recv, int64_t count, void *comm, void *stream);
/* C1 for RAGGED shards (B mod nranks != 0; the reference's drivers take any batch, dwt/dwt_all.jl:277-279): counts[r]
 * (host array, nranks entries, identical on every rank) = elements rank r contributes -- signal length x its share of the
 * batch; they land at element offset counts[0] + ... + counts[r-1] of recv on every rank.  One grouped point-to-point
 * exchange (ncclSend / ncclRecv), no padding; `send` may already be recv + this rank's offset (in place). */
int wx_allgatherv_out_f64(const double *send, double *recv, const int64_t *counts, int nranks, void *comm, void *stream);
int wx_allgatherv_out_f32(const float *send, float *recv, const int64_t *counts, int nranks, void *comm, void *stream);
int wx_allreduce_moments_f64(double *buf, int64_t count, void *comm, void *stream);
int wx_allreduce_moments_f32(float *buf, int64_t count, void *comm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WAVELETSEXT_HIP_H */
