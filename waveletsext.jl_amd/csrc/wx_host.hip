// wx_host.hip -- host-side plumbing of libwaveletsext_hip.so (no kernels here).
#include "wx_host.h"
#include <mutex>
#include <stdio.h>
#include <string.h>
#include <string>

static thread_local std::string g_wx_err;

int wx_set_hip_error(hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_wx_err = buf;
    (void)hipGetLastError();
    return WX_EHIP;
}
int wx_set_error(int code, const char *msg)
{
    g_wx_err = msg ? msg : "";
    return code;
}
const char *wx_err_cstr() { return g_wx_err.c_str(); }

hipStream_t wx_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

bool wx_is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// device memory must belong to the current device: the kernels are launched there (one process per GPU is the
// intended deployment; a caller driving several GPUs from one process selects the device before each call)
static bool wx_on_other_device(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice && a.device != cur;
}

int wx_pack_filter(const double *qmf, int F, WxFilt *out)
{
    if (!qmf) return wx_set_error(WX_EARG, "qmf is NULL");
    if (F < 2 || (F & 1) || F > WX_MAXF) return wx_set_error(WX_EARG, "filter length must be even, 2..64");
    memset(out, 0, sizeof *out);
    for (int i = 0; i < F; ++i) out->q[i] = qmf[i];
    out->F = F;
    return WX_OK;
}

int wx_maxtransformlevels(int64_t n)
{
    if (n < 2) return 0;
    int tl = 0;
    while ((n & 1) == 0) { n >>= 1; ++tl; }
    return tl;
}
bool wx_isdyadic(int64_t n) { return n >= 1 && (n & (n - 1)) == 0; }
int wx_getdepth_binary(int64_t idx) { int d = 0; while (idx > 1) { idx >>= 1; ++d; } return d; }
int wx_getdepth_quad(int64_t idx)
{
    int64_t t = 3 * idx - 2; int d = 0;
    while (t >= 4) { t >>= 2; ++d; }
    return d;
}
int64_t wx_gettreelength2d(int64_t m, int64_t n)
{
    const int L = wx_maxtransformlevels(m < n ? m : n);
    return (((int64_t)1 << (2 * L)) - 1) / 3;
}
bool wx_isvalidtree1d(int64_t n, const uint8_t *tree, int64_t ntree)
{
    if (ntree != n - 1) return false;
    if (ntree == 0) return true;
    if (!tree) return false;
    for (int64_t i = 1; 2 * i + 1 <= ntree; ++i)
        if (!tree[i - 1] && (tree[2 * i - 1] || tree[2 * i])) return false;
    return true;
}
bool wx_isvalidtree2d(int64_t m, int64_t n, const uint8_t *tree, int64_t ntree)
{
    if (wx_gettreelength2d(m, n) != ntree) return false;
    if (ntree == 0) return true;
    if (!tree) return false;
    const int L0 = ntree > 0 ? wx_getdepth_quad(ntree) : 0;
    const int64_t ns = (((int64_t)1 << (2 * L0)) - 1) / 3;
    for (int64_t i = 1; i <= ns; ++i) {
        const bool haschild = tree[4 * i - 3] || tree[4 * i - 2] || tree[4 * i - 1] || tree[4 * i];
        if (!tree[i - 1] && haschild) return false;
    }
    return true;
}
int wx_tree_depth1d(const uint8_t *tree, int64_t ntree)
{
    int L = 0;
    for (int64_t i = ntree; i >= 1; --i)
        if (tree[i - 1]) { L = wx_getdepth_binary(i) + 1; break; }
    return L;
}
int wx_tree_depth2d(const uint8_t *tree, int64_t ntree)
{
    int L = 0;
    for (int64_t i = ntree; i >= 1; --i)
        if (tree[i - 1]) { L = wx_getdepth_quad(i) + 1; break; }
    return L;
}
static void wx_colmap_rec(const uint8_t *tree, int64_t ntree, int64_t node, int d, int64_t j, int Leff,
                          std::vector<int> &col)
{
    if (node <= ntree && tree[node - 1]) {
        wx_colmap_rec(tree, ntree, 2 * node, d + 1, 2 * j, Leff, col);
        wx_colmap_rec(tree, ntree, 2 * node + 1, d + 1, 2 * j + 1, Leff, col);
    } else {
        const int64_t w = (int64_t)1 << (Leff - d);
        for (int64_t k = j * w; k < (j + 1) * w; ++k) col[(size_t)k] = d;
    }
}
void wx_leaf_colmap1d(const uint8_t *tree, int64_t ntree, int Leff, std::vector<int> &col)
{
    col.assign((size_t)1 << Leff, 0);
    wx_colmap_rec(tree, ntree, 1, 0, 0, Leff, col);
}

// ---- scratch ------------------------------------------------------------------------------
WxScratch::~WxScratch()
{
    for (void *p : ptrs)
        if (p && hipFreeAsync(p, st) != hipSuccess) (void)hipGetLastError();
}
// keep freed scratch cached in the device's default memory pool (the default release threshold of 0
// would hand it back to the driver at every synchronisation and make each call re-map gigabytes)
static void wx_retain_pool()
{
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev]) return;
    done[dev] = true;
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess) { (void)hipGetLastError(); return; }
    uint64_t thr = UINT64_MAX;
    if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr) != hipSuccess) (void)hipGetLastError();
}

// ---- per-device cache of small constant tables -------------------------------------------------
// Keyed on (device, 64-bit hash, size) and verified against a pinned host copy of the content, which is also the
// source of the asynchronous upload: no std::string copy per call, no NULL-stream hipMemcpy (that would serialise
// against every blocking stream of the host program).  An entry carries the event of its upload; a later user on
// another stream waits for that event on the device, a caller without a stream waits for it on the host.  Bounded at
// 1024 entries: a flush retires the whole generation, tagged with its device, and frees the generation retired by the
// previous flush under that device (so a table handed out during a running API call outlives it by a full generation).
#include <map>
#include <tuple>
namespace {
struct WxConstEnt {
    void *dev;
    void *pin;
    size_t bytes;
    hipEvent_t ev;
    bool ready;
    int device;
};
std::mutex g_const_mu;
std::multimap<std::tuple<int, uint64_t, size_t>, WxConstEnt> g_const;
std::vector<WxConstEnt> g_const_old;                                // previous generation, freed at the next flush
uint64_t wx_fnv1a(const void *p, size_t n)
{
    const unsigned char *c = static_cast<const unsigned char *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}
void wx_const_release(WxConstEnt &e)
{
    if (e.ev && hipEventDestroy(e.ev) != hipSuccess) (void)hipGetLastError();
    if (e.dev && hipFree(e.dev) != hipSuccess) (void)hipGetLastError();
    if (e.pin && hipHostFree(e.pin) != hipSuccess) (void)hipGetLastError();
    e.ev = nullptr; e.dev = e.pin = nullptr;
}
void wx_const_free_retired(int only_dev)
{
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
    std::vector<WxConstEnt> keep;
    for (auto &e : g_const_old) {
        if (only_dev >= 0 && e.device != only_dev) { keep.push_back(e); continue; }
        if (hipSetDevice(e.device) != hipSuccess) { (void)hipGetLastError(); keep.push_back(e); continue; }
        if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
        wx_const_release(e);
    }
    g_const_old.swap(keep);
    if (hipSetDevice(cur) != hipSuccess) (void)hipGetLastError();
}
}
const void *wx_const_upload(const void *host, size_t bytes, hipStream_t st, bool have_stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const auto key = std::make_tuple(dev, wx_fnv1a(host, bytes), bytes);
    std::lock_guard<std::mutex> lk(g_const_mu);
    auto range = g_const.equal_range(key);
    for (auto it = range.first; it != range.second; ++it) {
        WxConstEnt &e = it->second;
        if (bytes && memcmp(e.pin, host, bytes) != 0) continue;       // hash collision
        if (!e.ready) {
            hipError_t q = hipEventQuery(e.ev);
            if (q == hipSuccess) e.ready = true;
            else {
                (void)hipGetLastError();
                q = have_stream ? hipStreamWaitEvent(st, e.ev, 0) : hipEventSynchronize(e.ev);
                if (q != hipSuccess) { wx_set_hip_error(q, "wait for a constant table", __FILE__, __LINE__); return nullptr; }
                if (!have_stream) e.ready = true;
            }
        }
        return e.dev;
    }
    if (g_const.size() >= 1024) {
        wx_const_free_retired(-1);
        for (auto &kv : g_const) g_const_old.push_back(kv.second);
        g_const.clear();
    }
    WxConstEnt e = {nullptr, nullptr, bytes, nullptr, false, dev};
    hipError_t rc = hipMalloc(&e.dev, bytes ? bytes : 16);
    if (rc == hipSuccess) rc = hipHostMalloc(&e.pin, bytes ? bytes : 16, hipHostMallocDefault);
    if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e.ev, hipEventDisableTiming);
    if (rc == hipSuccess) {
        if (bytes) memcpy(e.pin, host, bytes);
        static thread_local hipStream_t own[64] = {nullptr};          // uploads of callers that pass no stream
        hipStream_t up = st;
        if (!have_stream) {
            if (dev < 64 && !own[dev]) rc = hipStreamCreateWithFlags(&own[dev], hipStreamNonBlocking);
            up = dev < 64 ? own[dev] : nullptr;
        }
        if (rc == hipSuccess && bytes) rc = hipMemcpyAsync(e.dev, e.pin, bytes, hipMemcpyHostToDevice, up);
        if (rc == hipSuccess) rc = hipEventRecord(e.ev, up);
        if (rc == hipSuccess && !have_stream) { rc = hipEventSynchronize(e.ev); e.ready = true; }
    }
    if (rc != hipSuccess) {
        wx_set_hip_error(rc, "upload of a constant table", __FILE__, __LINE__);
        wx_const_release(e);
        return nullptr;
    }
    g_const.emplace(key, e);
    return e.dev;
}
const void *wx_const_upload(const void *host, size_t bytes) { return wx_const_upload(host, bytes, nullptr, false); }

void wx_release_host_staging();
// hand the cached scratch and constant tables of the current device back to the driver (the only state
// the library owns)
extern "C" int wx_shutdown(void)
{
    int dev = 0, n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void)hipGetLastError(); return WX_OK; }
    WX_HIP_CHECK(hipGetDevice(&dev));
    WX_HIP_CHECK(hipDeviceSynchronize());
    {
        std::lock_guard<std::mutex> lk(g_const_mu);
        for (auto it = g_const.begin(); it != g_const.end();) {
            if (std::get<0>(it->first) == dev) { wx_const_release(it->second); it = g_const.erase(it); }
            else ++it;
        }
        wx_const_free_retired(dev);
    }
    wx_release_host_staging();
    hipMemPool_t pool;
    WX_HIP_CHECK(hipDeviceGetDefaultMemPool(&pool, dev));
    WX_HIP_CHECK(hipMemPoolTrimTo(pool, 0));
    return WX_OK;
}

void *WxScratch::alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    wx_retain_pool();
    hipError_t e = hipMallocAsync(&p, bytes, st);
    if (e != hipSuccess) { wx_set_hip_error(e, "hipMallocAsync(scratch)", __FILE__, __LINE__); return nullptr; }
    ptrs.push_back(p);
    return p;
}
void *WxScratch::upload(const void *host, size_t bytes)
{
    // trees and column maps are small and repeat from call to call: the content-keyed cache makes the
    // upload a lookup (no allocation, no stream synchronisation)
    if (bytes > 0 && bytes <= 64 * 1024) return const_cast<void *>(wx_const_upload(host, bytes, st, true));
    void *p = alloc(bytes);
    if (!p) return nullptr;
    hipError_t e = hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { wx_set_hip_error(e, "upload(tree)", __FILE__, __LINE__); return nullptr; }
    return p;
}

// ---- staged IO ----------------------------------------------------------------------------
#include <chrono>
static bool wx_host_trace()
{
    static const bool on = wx_getenv("WX_HOST_TRACE") && atoi(wx_getenv("WX_HOST_TRACE")) != 0;
    return on;
}
static double wx_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static hipError_t wx_h2d_staged(void *dev, const void *user, size_t bytes, hipStream_t st);
static void wx_advise_hugepages(void *user, size_t bytes);
WxIO::~WxIO()
{
    for (auto &it : items) {
        if (!it.dev) continue;
        if (it.staged && hipFree(it.dev) != hipSuccess) (void)hipGetLastError();
        if (it.realigned && hipFreeAsync(it.dev, st) != hipSuccess) (void)hipGetLastError();
    }
}
// Device arrays that do not start on a 32-byte boundary (a Julia view such as `@view x[2:end]`, a sub-array of a batch: the reference takes
// any view, dwt/dwt_all.jl:277) pass through an aligned scratch copy on the call's stream: the register kernels address whole 16- / 32-byte
// lines per lane and their launchers decline anything else -- until round 6 some call sites turned such a refusal into WX_EHIP
// (the lattice kernel "did not take" its subtree) instead of a slower path.  Two device-to-device copies for the rare caller, every kernel for everybody.
static bool wx_dev_unaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 31) != 0; }
const void *WxIO::in(const void *p, size_t bytes)
{
    if (bytes != 0 && p == nullptr) { err = wx_set_error(WX_EARG, "NULL data pointer for a non-empty array"); return nullptr; }
    if (bytes != 0 && wx_on_other_device(p)) { err = wx_set_error(WX_EARG, "array lives on another device than the current one"); return nullptr; }
    if (bytes == 0) return p;
    if (wx_is_device_ptr(p)) {
        if (!wx_dev_unaligned(p)) return p;
        void *a = nullptr;
        hipError_t ea = hipMallocAsync(&a, bytes, st);
        if (ea != hipSuccess) { wx_set_hip_error(ea, "hipMallocAsync(realign in)", __FILE__, __LINE__); return nullptr; }
        items.push_back({const_cast<void *>(p), a, bytes, false, false, true});
        ea = hipMemcpyAsync(a, p, bytes, hipMemcpyDeviceToDevice, st);
        if (ea != hipSuccess) { wx_set_hip_error(ea, "hipMemcpyAsync(realign in)", __FILE__, __LINE__); return nullptr; }
        return a;
    }
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) { wx_set_hip_error(e, "hipMalloc(stage in)", __FILE__, __LINE__); return nullptr; }
    items.push_back({const_cast<void *>(p), d, bytes, true, false, false});
    any_staged = true;
    const double t0 = wx_host_trace() ? wx_now_ms() : 0.0;
    e = wx_h2d_staged(d, p, bytes, st);
    if (wx_host_trace()) {
        (void)hipStreamSynchronize(st);
        fprintf(stderr, "wx host: H2D %.1f MiB in %.2f ms (%.1f GB/s)\n", bytes / 1048576.0, wx_now_ms() - t0, bytes / (wx_now_ms() - t0) / 1e6);
    }
    if (e != hipSuccess) { wx_set_hip_error(e, "hipMemcpyAsync(H2D)", __FILE__, __LINE__); return nullptr; }
    return d;
}
void *WxIO::out(void *p, size_t bytes)
{
    if (bytes != 0 && p == nullptr) { err = wx_set_error(WX_EARG, "NULL data pointer for a non-empty array"); return nullptr; }
    if (bytes != 0 && wx_on_other_device(p)) { err = wx_set_error(WX_EARG, "array lives on another device than the current one"); return nullptr; }
    if (bytes == 0) return p;
    if (wx_is_device_ptr(p)) {
        if (!wx_dev_unaligned(p)) return p;
        void *a = nullptr;
        hipError_t ea = hipMallocAsync(&a, bytes, st);
        if (ea != hipSuccess) { wx_set_hip_error(ea, "hipMallocAsync(realign out)", __FILE__, __LINE__); return nullptr; }
        items.push_back({p, a, bytes, false, true, true});
        any_realigned = true;
        return a;
    }
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) { wx_set_hip_error(e, "hipMalloc(stage out)", __FILE__, __LINE__); return nullptr; }
    bool also_input = false;                       // an in-place result has been faulted in by its owner already
    for (const auto &it : items) also_input = also_input || it.user == p;
    items.push_back({p, d, bytes, true, true, false});
    any_staged = true;
    if (!also_input) wx_advise_hugepages(p, bytes);
    return d;
}
// ---- device -> pageable host memory at PCIe speed ------------------------------------------------------------
// A D2H copy straight into a freshly allocated pageable array is bound by the host's first-touch page faults, taken
// one at a time by the copying thread (measured: 10.5 GB/s for the 3.5 GiB packet table of 8192 x 4096 signals).  Here
// the table comes over in 32 MiB chunks through two pinned buffers (DMA at the link rate) while a small pool of host
// threads copies the previous chunk into the caller's array -- each thread faults its own slice, so the page faults
// run in parallel and hide behind the next chunk's DMA.  The pool and the pinned ring are created on first use and
// released by wx_shutdown().  This replaces the copy-per-signal loop of the reference's batch drivers
// (dwt/dwt_all.jl:277-279) on the host side of the drop-in.
#include <condition_variable>
#include <thread>
#include <atomic>
namespace {
class WxHostPool {
  public:
    ~WxHostPool() { stop(); }
    void parallel_memcpy(char *dst, const char *src, size_t n)
    {
        start();
        const size_t slice = 1 << 20;
        std::unique_lock<std::mutex> lk(mu);
        jd = dst; js = src; jn = n; jslice = slice;
        jnext = 0; jtotal = (n + slice - 1) / slice; jdone = 0;
        ++generation;
        cv.notify_all();
        lk.unlock();
        work();                                                        // the caller copies too
        lk.lock();
        cv_done.wait(lk, [&] { return jdone == jtotal; });
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        for (auto &t : th) if (t.joinable()) t.join();
        th.clear();
        started = false;
        quit = false;
    }

  private:
    void start()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (started) return;
        started = true;
        int n = wx_getenv("WX_HOST_THREADS") ? atoi(wx_getenv("WX_HOST_THREADS")) : 16;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw) n = hw;
        if (n < 1) n = 1;
        for (int i = 0; i + 1 < n; ++i) th.emplace_back([this] { loop(); });
    }
    void work()
    {
        for (;;) {
            size_t i;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (jnext >= jtotal) return;
                i = jnext++;
            }
            const size_t off = i * jslice, len = (off + jslice <= jn) ? jslice : jn - off;
            memcpy(jd + off, js + off, len);
            std::lock_guard<std::mutex> lk(mu);
            if (++jdone == jtotal) cv_done.notify_all();
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
            }
            work();
        }
    }
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> th;
    bool started = false, quit = false;
    uint64_t generation = 0;
    char *jd = nullptr;
    const char *js = nullptr;
    size_t jn = 0, jslice = 0, jnext = 0, jtotal = 0, jdone = 0;
};
// The two pinned buffers are portable (any device may copy into them); the events that mark a slot as filled belong to
// the device of the call's stream, so they are created per call (an event of another device cannot be recorded on the
// stream: with a process-global pair every staged copy on a second device failed).
struct WxPinRing {
    static constexpr size_t CH = (size_t)32 << 20;
    void *buf[2] = {nullptr, nullptr};
    bool ok = false;
    bool init()
    {
        if (ok) return true;
        for (int i = 0; i < 2; ++i)
            if (hipHostMalloc(&buf[i], CH, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); release(); return false; }
        ok = true;
        return true;
    }
    void release()
    {
        for (int i = 0; i < 2; ++i) {
            if (buf[i] && hipHostFree(buf[i]) != hipSuccess) (void)hipGetLastError();
            buf[i] = nullptr;
        }
        ok = false;
    }
};
std::mutex g_stage_mu;                  // one staged copy at a time (one ring)
WxPinRing g_ring;
WxHostPool g_pool;
}
void wx_release_host_staging()
{
    std::lock_guard<std::mutex> lk(g_stage_mu);
    g_ring.release();
    g_pool.stop();
}
// D2H of a large array into pageable memory through the pinned ring; returns hipSuccess or the first error
static hipError_t wx_d2h_staged(void *user, const void *dev, size_t bytes, hipStream_t st)
{
    static const bool off = wx_getenv("WX_HOST_STAGING") && atoi(wx_getenv("WX_HOST_STAGING")) == 0;
    hipPointerAttribute_t at;
    const bool pinned_user = hipPointerGetAttributes(&at, user) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    std::unique_lock<std::mutex> lk(g_stage_mu, std::try_to_lock);
    if (off || pinned_user || bytes < ((size_t)8 << 20) || !lk.owns_lock() || !g_ring.init())
        return hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, st);
    const size_t CH = WxPinRing::CH;
    const size_t nch = (bytes + CH - 1) / CH;
    auto len = [&](size_t k) { return (k + 1) * CH <= bytes ? CH : bytes - k * CH; };
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (ev[0]) (void)hipEventDestroy(ev[0]);
            return hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, st);
        }
    size_t done = 0;                                                   // chunks that have reached the user's array
    hipError_t e = hipMemcpyAsync(g_ring.buf[0], dev, len(0), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(ev[0], st);
    for (size_t k = 0; k < nch && e == hipSuccess; ++k) {
        const int s = (int)(k & 1);
        if (k + 1 < nch) {                                             // the other slot was drained in the previous round
            e = hipMemcpyAsync(g_ring.buf[s ^ 1], (const char *)dev + (k + 1) * CH, len(k + 1), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipEventRecord(ev[s ^ 1], st);
            if (e != hipSuccess) break;
        }
        e = hipEventSynchronize(ev[s]);
        if (e != hipSuccess) break;
        g_pool.parallel_memcpy((char *)user + k * CH, (const char *)g_ring.buf[s], len(k));
        done = k + 1;
    }
    if (e != hipSuccess) {
        // nothing may still be writing into the ring when the lock is released; the rest goes the plain way
        (void)hipGetLastError();
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        e = done < nch ? hipMemcpyAsync((char *)user + done * CH, (const char *)dev + done * CH, bytes - done * CH, hipMemcpyDeviceToHost, st)
                       : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    for (int i = 0; i < 2; ++i) (void)hipEventDestroy(ev[i]);
    return e;
}

// A freshly allocated result array (Julia's Array{T}(undef, ...), numpy.empty) has no pages yet: every 4 KiB page the copy
// threads touch is one fault -- measured on the GPU box (tools/dbg/hostpath_probe.hip, profiles/r05_hostpath_probe.txt) 14.6 GB/s
// with 16 threads, which is what bounded `wpdall` of host arrays at 15.8 GB/s in round 4.  With transparent huge pages in
// `madvise` mode (this image) the same first touch runs at 275 GB/s once the range carries MADV_HUGEPAGE: one fault per 2 MiB.
// The advice changes nothing else about the caller's memory; ranges under 64 MiB and failures are ignored.
#include <sys/mman.h>
static std::atomic<int> g_host_hugepages{1};
// process-wide switch for the advice (on by default; wx_set_host_hugepages(0) before the first host-array call turns it off): the advice
// STAYS on the caller's address range after the call -- and after the caller frees it and the allocator reuses the range --, splits the
// mapping it falls into, and lets khugepaged collapse pages there later.  A host that minds says so here.
extern "C" int wx_set_host_hugepages(int on)
{
    return g_host_hugepages.exchange(on ? 1 : 0);
}
static void wx_advise_hugepages(void *user, size_t bytes)
{
    static const bool off = wx_getenv("WX_HOST_HUGEPAGES") && atoi(wx_getenv("WX_HOST_HUGEPAGES")) == 0;
    if (off || !g_host_hugepages.load(std::memory_order_relaxed) || bytes < ((size_t)64 << 20)) return;
    const uintptr_t a = ((uintptr_t)user + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1);
    const uintptr_t b = ((uintptr_t)user + bytes) & ~(uintptr_t)(((size_t)2 << 20) - 1);
    if (b > a) (void)madvise((void *)a, (size_t)(b - a), MADV_HUGEPAGE);
}

// H2D of a large pageable array through the pinned ring: the host threads copy chunk k + 1 into one pinned buffer while the
// DMA engine sends chunk k from the other (the runtime's own path for pageable memory reached 28 GB/s on the GPU box, the ring
// is bound by the link: 57 GB/s).  Ordered on `st` like a plain hipMemcpyAsync; returns once the last chunk is queued AND the
// ring is free again (the buffers belong to the next staged copy).
static hipError_t wx_h2d_staged(void *dev, const void *user, size_t bytes, hipStream_t st)
{
    static const bool off = wx_getenv("WX_HOST_STAGING") && atoi(wx_getenv("WX_HOST_STAGING")) == 0;
    hipPointerAttribute_t at;
    const bool pinned_user = hipPointerGetAttributes(&at, user) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    std::unique_lock<std::mutex> lk(g_stage_mu, std::try_to_lock);
    if (off || pinned_user || bytes < ((size_t)64 << 20) || !lk.owns_lock() || !g_ring.init())
        return hipMemcpyAsync(dev, user, bytes, hipMemcpyHostToDevice, st);
    const size_t CH = WxPinRing::CH;
    const size_t nch = (bytes + CH - 1) / CH;
    auto len = [&](size_t k) { return (k + 1) * CH <= bytes ? CH : bytes - k * CH; };
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (ev[0]) (void)hipEventDestroy(ev[0]);
            return hipMemcpyAsync(dev, user, bytes, hipMemcpyHostToDevice, st);
        }
    hipError_t e = hipSuccess;
    size_t sent = 0;
    for (size_t k = 0; k < nch && e == hipSuccess; ++k) {
        const int s = (int)(k & 1);
        if (k >= 2) e = hipEventSynchronize(ev[s]);                    // the DMA out of this slot (chunk k - 2) has finished
        if (e != hipSuccess) break;
        g_pool.parallel_memcpy((char *)g_ring.buf[s], (const char *)user + k * CH, len(k));
        e = hipMemcpyAsync((char *)dev + k * CH, g_ring.buf[s], len(k), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(ev[s], st);
        if (e == hipSuccess) sent = k + 1;
    }
    // the ring must be idle before the lock goes: wait for the (up to) two chunks still in flight
    for (int i = 0; i < 2; ++i) {
        const hipError_t w = hipEventSynchronize(ev[i]);
        if (w != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(st); (void)hipGetLastError(); }
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = sent < nch ? hipMemcpyAsync((char *)dev + sent * CH, (const char *)user + sent * CH, bytes - sent * CH, hipMemcpyHostToDevice, st)
                       : hipSuccess;
    }
    for (int i = 0; i < 2; ++i) (void)hipEventDestroy(ev[i]);
    return e;
}

int WxIO::finish(int rc)
{
    if (err != WX_OK) rc = err;                    // an argument error outranks the caller's generic code
    if (rc == WX_OK)                               // results of realigned device outputs (and in / out arrays whose item a caller marked
        for (auto &it : items)                     // copy_out) go back on the stream, nothing waits
            if (it.realigned && it.copy_out) {
                const hipError_t e = hipMemcpyAsync(it.user, it.dev, it.bytes, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) rc = wx_set_hip_error(e, "hipMemcpyAsync(realign out)", __FILE__, __LINE__);
            }
    if (!any_staged) return rc;
    if (rc == WX_OK) {
        for (auto &it : items)
            if (it.copy_out && it.staged) {
                double t0 = 0.0;
                if (wx_host_trace()) { (void)hipStreamSynchronize(st); t0 = wx_now_ms(); }
                hipError_t e = wx_d2h_staged(it.user, it.dev, it.bytes, st);
                if (wx_host_trace()) {
                    (void)hipStreamSynchronize(st);
                    fprintf(stderr, "wx host: D2H %.1f MiB in %.2f ms (%.1f GB/s)\n", it.bytes / 1048576.0, wx_now_ms() - t0, it.bytes / (wx_now_ms() - t0) / 1e6);
                }
                if (e != hipSuccess) rc = wx_set_hip_error(e, "hipMemcpyAsync(D2H)", __FILE__, __LINE__);
            }
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess && rc == WX_OK) rc = wx_set_hip_error(e, "hipStreamSynchronize", __FILE__, __LINE__);
    return rc;
}
