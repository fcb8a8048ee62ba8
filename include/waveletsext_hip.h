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
 *    Tree-driven calls synchronise `stream` once while uploading the tree.
 *  - `stream` is a hipStream_t (NULL = default stream).
 *  - Return value: WX_OK, or a negative status.  WX_EASSERT / WX_EARG / WX_EBOUNDS mean the
 *    reference would have thrown AssertionError / ArgumentError / BoundsError for these
 *    arguments; wx_last_error() gives the message.  Nothing throws across the boundary.
 *  - In/out buffers must not alias unless stated.  The library is re-entrant; the only global
 *    state is the per-thread last-error string.
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
/* test hook: nonzero forces the one-level-per-launch kernels instead of the fused LDS kernels */
void wx_set_force_generic(int on);

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

#ifdef __cplusplus
}
#endif
#endif /* WAVELETSEXT_HIP_H */
