// wx_kernels.h -- device-pointer launchers (all pointers are device memory, calls are
// asynchronous on `st`).  The C ABI in wx_api.hip wraps these.
#pragma once
#include "wx_common.h"

template <typename T> bool wx_fused1d_ok(int64_t n, int F);

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
template <typename T>
int wx_dev_getbasiscoef1d(const T *Xw, T *out, int64_t n, int k, int64_t batch, const int *colmap, int blk,
                          hipStream_t st);
