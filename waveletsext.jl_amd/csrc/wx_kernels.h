// wx_kernels.h -- device-pointer launchers (all pointers are device memory, calls are
// asynchronous on `st`).  The C ABI in wx_api.hip wraps these.
#pragma once
#include "wx_common.h"

template <typename T> bool wx_fused1d_ok(int64_t n, int F);
int wx_lattice_f32(bool inverse, const float *x, float *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt,
                   hipStream_t st);                                       // wx_lattice_f32.hip
int wx_swpt_deep_levels(int64_t n, int L, int F, bool ac, size_t esz);      // wx_swtdeep.hip
int wx_swpt_deep_fwd(double *xw, int64_t n, int L, int64_t batch, const WxFilt &filt, const WxAcFilt *ac, bool wpd, hipStream_t st);
int wx_swpt_deep_inv(const double *src, int64_t src_cols, double *dst, int64_t dst_cols, int64_t n, int L, int LP, int64_t batch,
                     const WxFilt &filt, hipStream_t st);
bool wx_lattice_applicable_f64(const WxFilt &filt);     // wx_lattice.hip: the filter has a lattice instantiation and factorises
int wx_skip_register_kernels();   // test hook (wx_debug_set_dispatch(2), wx_debug.h): skip the Haar / lattice register kernels

// ---- short signals (16 .. 512 samples) along any tree: n / 8 lanes per signal, uniform level loop (wx_smalltree.hip) ----
template <typename T> bool wx_small_tree_ok(int64_t n, int F);
// which trees of which lengths take it (status != NULL: a tree; pyramid: that tree is a pyramid)
template <typename T> bool wx_small_tree_wanted(int64_t n, int F, bool has_tree, bool pyramid);
template <typename T>
int wx_dev_small_tree(bool inverse, const T *x, T *y, int64_t n, int L, int64_t batch, const WxFilt &filt, const uint8_t *status,
                      int64_t nstatus, hipStream_t st);

// ---- very short signals, one lane per signal (wx_lanetree.h): 16 .. 64 samples, 128 for Float32; the tree as a bit mask ----
struct WxLaneTree;
int wx_lane_tree_f64(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, const WxLaneTree &tree, const WxFilt &filt,
                     hipStream_t st);
int wx_lane_tree_f32(bool inverse, const float *x, float *y, int64_t n, int L, int64_t batch, const WxLaneTree &tree, const WxFilt &filt,
                     hipStream_t st);

// ---- 1-D decimated (wx_dwt1d.hip) ----
template <typename T>
int wx_dev_wpd1d(const T *x, T *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st,
                 int force_generic);
template <typename T>
int wx_dev_wpt1d(const T *x, T *y, int64_t n, int L, int64_t batch, const WxFilt &filt,
                 const uint8_t *status, int64_t nstatus, T *scratch, hipStream_t st, int force_generic);
template <typename T>
int wx_dev_iwpt1d(const T *xw, T *xh, int64_t n, int L, int64_t batch, const WxFilt &filt,
                  const uint8_t *status, int64_t nstatus, const int *colmap, int log2blk, int64_t in_stride,
                  T *scratch, T *scratch2, hipStream_t st, int force_generic);
// dwt / idwt (the pyramid of depth Lp) of long signals, n = 4096 << dl, dl = 2 .. 4: one tiled pass per top level on the
// approximation branch, then the 4096-sample pyramid on the tree-driven lattice kernels (wx_dwt1d.hip); scratch: n * batch
template <typename T> bool wx_dwt_long_ok(int64_t n, const WxFilt &filt);
template <typename T>
int wx_dev_dwt_long(const T *x, T *y, int64_t n, int Lp, int64_t batch, const WxFilt &filt, const uint8_t *status, int64_t nstatus,
                    T *scratch, hipStream_t st);
template <typename T>
int wx_dev_idwt_long(const T *xw, T *y, int64_t n, int Lp, int64_t batch, const WxFilt &filt, const uint8_t *status, int64_t nstatus,
                     const WxThreshArg &thr, T *scratch, hipStream_t st);
// wpt / iwpt along any tree (host copy `htree`) of long Float64 signals the lattice takes at 4096 samples: the split nodes of the
// top levels one tiled pass per level, one tree-driven lattice launch per 4096-sample node that is split further
template <typename T> bool wx_wpt_long_tree_ok(int64_t n, const WxFilt &filt);
template <typename T>
int wx_dev_wpt_long_tree(const T *x, T *y, int64_t n, int Lp, int64_t batch, const WxFilt &filt, const uint8_t *htree, int64_t ntree,
                         T *scratch, bool inverse, hipStream_t st);
template <typename T>
int wx_dev_getbasiscoef1d(const T *Xw, T *out, int64_t n, int k, int64_t batch, const int *colmap, int blk,
                          hipStream_t st);

// ---- 1-D redundant: SWT / ACWT (wx_swt1d.hip); layout 0 = dwt (n,L+1), 1 = wpt (n,2^L), 2 = wpd heap ----
template <typename T>
int wx_dev_swt_fwd(const T *x, T *xw, int64_t n, int L, int layout, int64_t batch, const WxFilt &filt,
                   const WxAcFilt *ac, hipStream_t st);
// inverse schedule: pass i reconstructs depth to[i] from depth from[i] (from - to = 1, or 2 for a fused
// average-based iswpt pass working on R[i] residue classes per workgroup, OPT[i] rows per thread); its output lives in scratch
// buffer buf[i] (-1 = the caller's x).  need_cols[] = columns per signal each scratch buffer must hold.
struct WxSwtInvPlan {
    int npass;
    int from[32], to[32], buf[32], R[32], OPT[32];
    int64_t need_cols[2];
};
void wx_swt_inv_plan(int layout, int L, int F, int64_t sm, int64_t n, size_t esz, bool has_tree, WxSwtInvPlan *P,
                     bool haar6 = false);
// last six levels of the Haar swpt / average-based iswpt as sliding Walsh-Hadamard transforms (wx_haarswt.hip)
bool wx_haar_swpt6_ok(int64_t n, int L, const WxFilt &filt, size_t esz);
int wx_haar_swpt6_fwd(double *xw, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
int wx_haar_iswpt_levels();
int wx_haar_iswpt6(const double *src, int64_t src_cols, double *dst, int64_t dst_cols, int64_t n, int L, int64_t batch,
                   const WxFilt &filt, hipStream_t st);
template <typename T>
int wx_dev_swt_inv(const T *xw, T *x, int64_t n, int L, int layout, int ncols, int64_t batch, int64_t sm,
                   const uint8_t *dtree, int64_t ntree, const WxFilt &filt, const WxSwtInvPlan &plan, T *scratch0,
                   T *scratch1, hipStream_t st);
template <typename T> int wx_dev_iacdwt(const T *xw, T *x, int64_t n, int L, int64_t batch, hipStream_t st);
template <typename T> int wx_dev_iacwpt(const T *xw, T *x, int64_t n, int L, int64_t batch, hipStream_t st);
template <typename T>
int wx_dev_iacwpd(const T *xw, T *x, int64_t n, int ncols, int64_t batch, const uint8_t *dtree, int64_t ntree,
                  int Lfull, hipStream_t st);

// acwpd top table (depths 0 .. D0) + the JBB moments of its columns in the passes that produce them (wx_swt1d.hip): 1 = done, 0 = n/a
bool wx_acwpd_top_moments_ok(int64_t n, int D0);
// sums = false: only sumsq is produced (the first moments come from the transform of the sum signal, see api_acwpd_jbb_moments)
int wx_dev_acwpd_top_moments(const double *x, double *tab, int64_t n, int D0, int64_t batch, const WxAcFilt &ac, double *sum, double *sumsq,
                             int acc, hipStream_t st, bool sums = true);

// ---- JBB (wx_jbb.hip) ----
template <typename T>
int wx_dev_jbb_moments(const T *X, T *sum, T *sumsq, int64_t nk, int64_t batch, int accumulate, T *scratch,
                       int nchunks, hipStream_t st);
template <typename T>
int wx_dev_jbb_costs(const T *sum, const T *sumsq, int64_t Ntot, int64_t n, int64_t k, int redundant,
                     int cost_kind, double p, T *costs, hipStream_t st);

// ---- 2-D decimated (wx_dwt2d.hip) ----
template <typename T>
int wx_dev_wpd2d(const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const WxFilt &filt, T *tmp,
                 hipStream_t st);
template <typename T>
int wx_dev_wpt2d(const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const WxFilt &filt,
                 const uint8_t *status, int64_t nstatus, T *tmp, T *pong, bool inverse, int64_t in_img,
                 hipStream_t st, const uint8_t *htree = nullptr);
template <typename T>
int wx_dev_gather_leaves2d(const T *Xw, T *out, int64_t m, int64_t n, int k, int64_t batch, const int *colmap,
                           int nblk, hipStream_t st);
template <typename T> bool wx_wpt2d_fast_ok(int64_t m, int64_t n, int F);
template <typename T>
int wx_dev_wpt2d_fast(const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const WxFilt &filt, T *tmp,
                      bool inverse, int64_t in_img, hipStream_t st);

// ---- fused acwpd + JBB moments (wx_jbb.hip) ----
int wx_acwpd_fused_depth(int64_t n, int L, int F);
template <typename T>
int wx_dev_threshold_copy(const T *X, T *Y, int64_t n, int64_t batch, int th_kind, const T *t, int per_signal, int64_t row_lo,
                          double scale, hipStream_t st);                                // wx_denoise.hip
template <typename T> bool wx_iwpt1d_thresh_fusable(int64_t n, int F, const uint8_t *status);
template <typename T>
int wx_dev_iwpt1d_thresh(const T *xw, T *xh, int64_t n, int L, int64_t batch, const WxFilt &filt, const uint8_t *status,
                         int64_t nstatus, const WxThreshArg &thr, hipStream_t st);
bool wx_acwpd_mfma_ok(int64_t n, int L, int D0);                    // wx_acsubtree.hip: one wavefront per subtree, matrix pipe
int wx_dev_acwpd_subtree_mfma(const double *top, double *sum, double *sumsq, int64_t n, int L, int D0, int64_t batch,
                              const WxAcFilt &ac, int accumulate, hipStream_t st, bool sums = true);
// sum of the signals of a batch, in signal order inside each of a fixed number of groups, groups added in order (deterministic), and
// dst += src: the linear path of the first moments (wx_acsubtree.hip)
int wx_dev_sum_signals(const double *x, int64_t n, int64_t batch, double *out, double *part, int groups, hipStream_t st);
int wx_dev_add_to(double *dst, const double *src, int64_t count, hipStream_t st);
int wx_dev_acwpd_subtree_moments(const double *top, double *sum, double *sumsq, int64_t n, int L, int D0,
                                 int64_t batch, const WxAcFilt &ac, int accumulate, hipStream_t st);

// ---- 2-D redundant: SWT / ACWT (wx_swt2d.hip); layout 0 = dwt (m,n,3L+1), 1 = wpt (m,n,4^L), 2 = wpd quad heap ----
template <typename T>
int wx_dev_red2d_fwd(const T *x, T *xw, int64_t m, int64_t n, int L, int layout, int64_t batch, const WxFilt &filt,
                     const WxAcFilt *ac, T *tmp, hipStream_t st);
template <typename T>
int wx_dev_red2d_inv(const T *xw, T *x, int64_t m, int64_t n, int L, int layout, int64_t ncols, int64_t batch, int64_t sm,
                     bool ac, const uint8_t *dtree, int64_t ntree, const WxFilt &filt, T *s0, T *s1, T *tmp,
                     hipStream_t st);
template <typename T>
int wx_dev_jbb_costs2d(const T *sum, const T *sumsq, int64_t Ntot, int64_t m, int64_t n, int64_t k, int redundant,
                       int cost_kind, double p, T *costs, hipStream_t st);

// Haar packets as Walsh-Hadamard transforms (wx_haar.hip); 0 = not applicable, 1 = launched, < 0 = error
int wx_haar_wpt_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
int wx_haar_iwpt_f64(const double *xw, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);

// full-tree Float64 packets as a lattice of plane rotations in the registers of one wavefront per signal
// (wx_lattice.hip); 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_wpd_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
// tree-driven wpt / iwpt / iwpd on the lattice (wx_lattice_tree.h): 0 = not applicable, 1 = launched, < 0 = error.
// in_stride / out_stride: elements between consecutive signals of x / y (out_stride = 0: n)
int wx_lattice_tree_f64(bool inverse, const double *x, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                        int64_t col_stride, const WxFilt &filt, const uint8_t *dstatus, int64_t nstatus, hipStream_t st,
                        const WxThreshArg *thr = nullptr, int64_t out_stride = 0);
bool wx_lattice_tree_applicable_f64(int64_t n, const WxFilt &filt);
// the same for Float32 signals (dense leaves; wx_lattice_tree32.h)
int wx_lattice_tree_f32(bool inverse, const float *x, float *y, int64_t n, int L, int64_t batch, int64_t in_stride, const WxFilt &filt,
                        const uint8_t *dstatus, int64_t nstatus, hipStream_t st, const WxThreshArg *thr = nullptr, int64_t out_stride = 0);
bool wx_lattice_tree_applicable_f32(int64_t n, const WxFilt &filt);
int wx_lattice_wpt_f64(const double *x, double *y, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
// 8192-sample signals along a tree, inverse, in one pass (wx_lattice_8kt.h); child c is a leaf when dstatus_c is NULL
int wx_lattice_tree8k_inv_f64(const double *x, double *y, int64_t batch, const WxFilt &filt, const uint8_t *dstatus0, int depth0,
                              const uint8_t *dstatus1, int depth1, hipStream_t st);
int wx_lattice_iwpt_f64(const double *xw, double *y, int64_t n, int L, int64_t batch, int64_t in_stride,
                        const WxFilt &filt, hipStream_t st);

// 512 x 512 Float32 images, depth 6: one transposing lattice pass (wx_lattice2d.hip); 0 = not applicable, 1 = launched
bool wx_lattice2d_ok(int64_t m, int64_t n, int L, const WxFilt &filt, size_t esz);
int wx_lattice2d_colT_f32(const float *src, float *dst, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse, int pass, hipStream_t st);
// both passes in one persistent launch through a cache-resident ring (round 6): 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice2d_fused_f32(const float *src, float *dst, float *ring, unsigned *ctl, int64_t m, int L, int64_t batch, const WxFilt &filt, bool inverse,
                           hipStream_t st);
int64_t wx_lattice2d_ring_elems(int64_t m, int64_t batch);
size_t wx_lattice2d_ctl_bytes();
bool wx_lattice2d_fused_on();
// row pass of a 2-D full tree on the lattice kernels (wx_lattice_rows.h, dispatch in wx_lattice_rows.hip): L levels along the rows of
// (m, n) images; 0 = not applicable, 1 = launched, < 0 = error
int wx_lattice_rows(bool inverse, const double *x, double *y, int64_t in_img, int64_t out_img, int64_t m, int64_t n, int L, int64_t batch,
                    const WxFilt &filt, hipStream_t st);
int wx_lattice_rows(bool inverse, const float *x, float *y, int64_t in_img, int64_t out_img, int64_t m, int64_t n, int L, int64_t batch,
                    const WxFilt &filt, hipStream_t st);
// 64 x 64 images, full quad tree of depth 1 .. 6 in ONE pass, an image per wavefront (wx_lattice_2d64.h); 0 = not applicable, 1 = launched.
// in_img: elements between the images of the inverse's input (a slice of a packet table)
int wx_lattice_2d64_fwd_f64(const double *x, double *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
int wx_lattice_2d64_fwd_f32(const float *x, float *y, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
int wx_lattice_2d64_inv_f64(const double *x, double *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, hipStream_t st);
int wx_lattice_2d64_inv_f32(const float *x, float *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, hipStream_t st);
static inline int wx_lattice_2d64(bool inverse, const double *x, double *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, hipStream_t st)
{
    if (inverse) return wx_lattice_2d64_inv_f64(x, y, L, batch, in_img, filt, st);
    return in_img == 4096 ? wx_lattice_2d64_fwd_f64(x, y, L, batch, filt, st) : 0;
}
static inline int wx_lattice_2d64(bool inverse, const float *x, float *y, int L, int64_t batch, int64_t in_img, const WxFilt &filt, hipStream_t st)
{
    if (inverse) return wx_lattice_2d64_inv_f32(x, y, L, batch, in_img, filt, st);
    return in_img == 4096 ? wx_lattice_2d64_fwd_f32(x, y, L, batch, filt, st) : 0;
}

// the deep levels of the pyramid in the registers of a lane (wx_dwttail.hip)
int wx_dwt_tail_levels(int64_t n, int L, int F, size_t esz);
template <typename T> int wx_dwt_tail(T *y, int64_t n, int Lt, int64_t batch, const WxFilt &filt, hipStream_t st);
template <typename T> int wx_idwt_tail(const T *xw, T *head, int64_t n, int Lt, int64_t batch, const WxFilt &filt, const WxThreshArg &thr,
                                        hipStream_t st);
int wx_dwt2d_tail_levels(int64_t m, int64_t n, int L, int F, size_t esz);
// pyramids of small images, a whole image per workgroup in LDS (wx_pyr2d.hip)
template <typename T> bool wx_pyr2d_small_ok(int64_t m, int64_t n, int L, int F);
template <typename T> int wx_dev_pyr2d_small(bool inverse, const T *x, T *y, int64_t m, int64_t n, int L, int64_t batch, const WxFilt &filt, hipStream_t st);
template <typename T> int wx_dwt2d_tail(T *y, int64_t m, int Lt, int64_t batch, const WxFilt &filt, hipStream_t st);
