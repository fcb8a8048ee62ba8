// Multi-GPU exchange steps of the hot path behind the C ABI: one process per GPU,
//   C1  all-gather of the reconstructed output shards     (dwt_all.jl:277-279 batch loop, sharded)
//   C2  all-reduce (sum) of the JBB moments [sum | sumsq] (bestbasis_tree.jl:153-154 means over signals)
// RCCL is bound lazily (dlopen) so that single-GPU callers never load it.  Transforms themselves need
// no collective.  The Python mirror uses torch.distributed for the same two steps; these entry points
// are for hosts without it (the Julia shim: MPI.jl / Distributed.jl broadcasts the 128-byte id).
#include "../../include/waveletsext_hip.h"     // the definitions below must match the public prototypes
#include "wx_common.h"
#include "wx_host.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <mutex>

#define WX_REQUIRE(cond, code, msg) \
    do { if (!(cond)) return wx_set_error(code, msg); } while (0)

extern "C" int wx_device_count(void);
static int wx_need_device2()
{
    if (wx_device_count() < 1) return wx_set_error(WX_EHIP, "no HIP device visible: the MI355X kernels cannot run");
    return WX_OK;
}

namespace {
struct WxNcclId { char internal[128]; };                  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_getid)(WxNcclId *);
typedef int (*fn_init)(void **, int, WxNcclId, int);
typedef int (*fn_destroy)(void *);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_errstr)(int);
typedef int (*fn_send)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_group)(void);
typedef int (*fn_count)(void *, int *);
struct Rccl {
    void *h = nullptr;
    fn_getid getid = nullptr;
    fn_init init = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_errstr errstr = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_group gstart = nullptr, gend = nullptr;
    fn_count count = nullptr, userrank = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.h) return WX_OK;
    // WX_RCCL_LIB is a deployment path, not a tuning knob: read with or without WX_KNOBS=1
    const char *names[] = {getenv("WX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return wx_set_error(WX_EUNSUPPORTED, "RCCL not found (set WX_RCCL_LIB to librccl.so)");
    Rccl r;
    r.h = h;
    r.getid = (fn_getid)dlsym(h, "ncclGetUniqueId");
    r.init = (fn_init)dlsym(h, "ncclCommInitRank");
    r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    r.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    r.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
    r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    r.send = (fn_send)dlsym(h, "ncclSend");
    r.recv = (fn_recv)dlsym(h, "ncclRecv");
    r.gstart = (fn_group)dlsym(h, "ncclGroupStart");
    r.gend = (fn_group)dlsym(h, "ncclGroupEnd");
    r.count = (fn_count)dlsym(h, "ncclCommCount");
    r.userrank = (fn_count)dlsym(h, "ncclCommUserRank");
    if (!r.getid || !r.init || !r.destroy || !r.allgather || !r.allreduce)
        return wx_set_error(WX_EUNSUPPORTED, "RCCL library lacks the collective entry points");
    g_rccl = r;
    return WX_OK;
}

int nccl_fail(int rc, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, g_rccl.errstr ? g_rccl.errstr(rc) : "RCCL error");
    return wx_set_error(WX_EHIP, buf);
}
enum { kNcclSum = 0, kNcclFloat32 = 7, kNcclFloat64 = 8 };   // rccl.h:448,466-467
}  // namespace

extern "C" {

int wx_comm_unique_id(void *id128)
{
    WX_REQUIRE(id128 != nullptr, WX_EARG, "id128 is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    WxNcclId id;
    const int nr = g_rccl.getid(&id);
    if (nr) return nccl_fail(nr, "ncclGetUniqueId");
    memcpy(id128, &id, sizeof id);
    return WX_OK;
}

int wx_comm_init(int nranks, int rank, const void *id128, void **comm)
{
    WX_REQUIRE(comm != nullptr && id128 != nullptr, WX_EARG, "comm / id128 is NULL");
    WX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, WX_EARG, "need 0 <= rank < nranks");
    int rc = wx_need_device2();
    if (rc) return rc;
    if ((rc = load_rccl())) return rc;
    WxNcclId id;
    memcpy(&id, id128, sizeof id);
    void *c = nullptr;
    const int nr = g_rccl.init(&c, nranks, id, rank);
    if (nr) return nccl_fail(nr, "ncclCommInitRank");
    *comm = c;
    return WX_OK;
}

int wx_comm_destroy(void *comm)
{
    if (!comm) return WX_OK;
    int rc = load_rccl();
    if (rc) return rc;
    const int nr = g_rccl.destroy(comm);
    if (nr) return nccl_fail(nr, "ncclCommDestroy");
    return WX_OK;
}

static int allgather_impl(const void *send, void *recv, int64_t count, int dt, void *comm, void *stream)
{
    WX_REQUIRE(comm != nullptr, WX_EARG, "comm is NULL");
    WX_REQUIRE(count >= 0, WX_EARG, "negative count");
    if (count == 0) return WX_OK;
    WX_REQUIRE(send != nullptr && recv != nullptr, WX_EARG, "NULL buffer");
    WX_REQUIRE(wx_is_device_ptr(send) && wx_is_device_ptr(recv), WX_EARG, "collectives take device pointers");
    int rc = load_rccl();
    if (rc) return rc;
    const int nr = g_rccl.allgather(send, recv, (size_t)count, dt, comm, wx_stream(stream));
    if (nr) return nccl_fail(nr, "ncclAllGather");
    return WX_OK;
}

// C1 with RAGGED shards (the reference's `*all` drivers take any batch, dwt/dwt_all.jl:277-279: B mod nranks != 0 is the normal
// case): rank r contributes counts[r] elements, which land at element offset counts[0] + ... + counts[r-1] of every rank's recv.
// One grouped point-to-point exchange -- every piece travels once over its own xGMI link and lands in place, no padding and no
// staging copy (the schedule of distributed.OverlappedAllGather); the own piece is a device copy unless send already points
// into recv at its offset.
static int allgatherv_impl(const void *send, void *recv, const int64_t *counts, int nranks, int dt, size_t esz, void *comm, void *stream)
{
    WX_REQUIRE(comm != nullptr, WX_EARG, "comm is NULL");
    WX_REQUIRE(counts != nullptr && nranks >= 1, WX_EARG, "counts is NULL / nranks < 1");
    int rc = load_rccl();
    if (rc) return rc;
    WX_REQUIRE(g_rccl.send && g_rccl.recv && g_rccl.gstart && g_rccl.gend && g_rccl.count && g_rccl.userrank, WX_EUNSUPPORTED,
               "RCCL library lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
    int cn = 0, rank = -1, nr;
    if ((nr = g_rccl.count(comm, &cn))) return nccl_fail(nr, "ncclCommCount");
    if ((nr = g_rccl.userrank(comm, &rank))) return nccl_fail(nr, "ncclCommUserRank");
    WX_REQUIRE(cn == nranks, WX_EARG, "counts must have one entry per rank of the communicator");
    int64_t total = 0, off = 0;
    for (int r = 0; r < nranks; ++r) {
        WX_REQUIRE(counts[r] >= 0, WX_EARG, "negative count");
        if (r < rank) off += counts[r];
        total += counts[r];
    }
    if (total == 0) return WX_OK;
    WX_REQUIRE(recv != nullptr && wx_is_device_ptr(recv), WX_EARG, "collectives take device pointers");
    const int64_t mine = counts[rank];
    if (mine) WX_REQUIRE(send != nullptr && wx_is_device_ptr(send), WX_EARG, "collectives take device pointers");
    hipStream_t st = wx_stream(stream);
    char *own = (char *)recv + (size_t)off * esz;
    if (mine && (const void *)own != send) {
        const hipError_t e = hipMemcpyAsync(own, send, (size_t)mine * esz, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return wx_set_hip_error(e, "allgatherv: own piece", __FILE__, __LINE__);
    }
    if (nranks == 1) return WX_OK;
    if ((nr = g_rccl.gstart())) return nccl_fail(nr, "ncclGroupStart");
    int64_t o = 0;
    int bad = 0;
    for (int r = 0; r < nranks && !bad; ++r) {
        if (r != rank) {
            if (mine) bad = g_rccl.send(send, (size_t)mine, dt, r, comm, st);
            if (!bad && counts[r]) bad = g_rccl.recv((char *)recv + (size_t)o * esz, (size_t)counts[r], dt, r, comm, st);
        }
        o += counts[r];
    }
    nr = g_rccl.gend();
    if (bad) return nccl_fail(bad, "ncclSend / ncclRecv");
    if (nr) return nccl_fail(nr, "ncclGroupEnd");
    return WX_OK;
}

static int allreduce_impl(void *buf, int64_t count, int dt, void *comm, void *stream)
{
    WX_REQUIRE(comm != nullptr, WX_EARG, "comm is NULL");
    WX_REQUIRE(count >= 0, WX_EARG, "negative count");
    if (count == 0) return WX_OK;
    WX_REQUIRE(buf != nullptr, WX_EARG, "NULL buffer");
    WX_REQUIRE(wx_is_device_ptr(buf), WX_EARG, "collectives take device pointers");
    int rc = load_rccl();
    if (rc) return rc;
    const int nr = g_rccl.allreduce(buf, buf, (size_t)count, dt, kNcclSum, comm, wx_stream(stream));
    if (nr) return nccl_fail(nr, "ncclAllReduce");
    return WX_OK;
}

int wx_allgather_out_f64(const double *send, double *recv, int64_t count, void *comm, void *stream)
{ return allgather_impl(send, recv, count, kNcclFloat64, comm, stream); }
int wx_allgather_out_f32(const float *send, float *recv, int64_t count, void *comm, void *stream)
{ return allgather_impl(send, recv, count, kNcclFloat32, comm, stream); }
int wx_allgatherv_out_f64(const double *send, double *recv, const int64_t *counts, int nranks, void *comm, void *stream)
{ return allgatherv_impl(send, recv, counts, nranks, kNcclFloat64, sizeof(double), comm, stream); }
int wx_allgatherv_out_f32(const float *send, float *recv, const int64_t *counts, int nranks, void *comm, void *stream)
{ return allgatherv_impl(send, recv, counts, nranks, kNcclFloat32, sizeof(float), comm, stream); }
int wx_allreduce_moments_f64(double *buf, int64_t count, void *comm, void *stream)
{ return allreduce_impl(buf, count, kNcclFloat64, comm, stream); }
int wx_allreduce_moments_f32(float *buf, int64_t count, void *comm, void *stream)
{ return allreduce_impl(buf, count, kNcclFloat32, comm, stream); }

}  // extern "C"
