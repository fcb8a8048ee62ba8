// wx_host.h -- host-side plumbing shared by the C-ABI translation units: pointer
// classification (host vs device), staging of host buffers, filter packing, tree checks.
#pragma once
#include "wx_common.h"
#include <vector>

hipStream_t wx_stream(void *s);

// true if p is device-accessible memory owned by HIP (device or managed allocation)
bool wx_is_device_ptr(const void *p);

int wx_pack_filter(const double *qmf, int F, WxFilt *out);

// integer helpers (same semantics as the reference / Wavelets.jl; see util.py for citations)
int wx_maxtransformlevels(int64_t n);
bool wx_isdyadic(int64_t n);
int wx_getdepth_binary(int64_t idx);
int wx_getdepth_quad(int64_t idx);
int64_t wx_gettreelength2d(int64_t m, int64_t n);
bool wx_isvalidtree1d(int64_t n, const uint8_t *tree, int64_t ntree);
bool wx_isvalidtree2d(int64_t m, int64_t n, const uint8_t *tree, int64_t ntree);
// depth of the deepest decomposed node + 1 (0 if the root is not decomposed)
int wx_tree_depth1d(const uint8_t *tree, int64_t ntree);
int wx_tree_depth2d(const uint8_t *tree, int64_t ntree);
// leaf depth of each block of n>>Leff positions (getbasiscoef traversal, Utils.jl:117-131)
void wx_leaf_colmap1d(const uint8_t *tree, int64_t ntree, int Leff, std::vector<int> &col);

// Small immutable tables (composite taps, ...) live in a per-device cache keyed by their content: uploaded
// once with a blocking copy, reused by every later call, released by wx_shutdown().  nullptr on failure.
const void *wx_const_upload(const void *host, size_t bytes);                                  // waits on the host
const void *wx_const_upload(const void *host, size_t bytes, hipStream_t st, bool have_stream);  // ordered on st

// Stream-ordered device scratch that frees itself on the same stream.
struct WxScratch {
    hipStream_t st;
    std::vector<void *> ptrs;
    explicit WxScratch(hipStream_t s) : st(s) {}
    ~WxScratch();
    void *alloc(size_t bytes);                      // nullptr on failure (error already recorded)
    // upload a small host array; synchronises the stream so the host copy may be released
    void *upload(const void *host, size_t bytes);
};

// Presents caller buffers (host or device) as device pointers for the duration of one call.
struct WxIO {
    hipStream_t st;
    struct Item { void *user; void *dev; size_t bytes; bool staged; bool copy_out; bool realigned; };
    std::vector<Item> items;
    bool any_staged = false;
    bool any_realigned = false;                     // a device output that does not start on a 32-byte boundary goes through an aligned copy
    int err = 0;                                    // WX_EARG when a NULL pointer was passed for a non-empty array
    explicit WxIO(hipStream_t s) : st(s) {}
    ~WxIO();
    const void *in(const void *p, size_t bytes);    // staged H2D if p is host memory; aligned device copy if p is a misaligned device pointer
    void *out(void *p, size_t bytes);               // staged, copied back by finish()
    int finish(int rc);                             // D2H copies + sync when anything was staged
};
