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
struct Rccl {
    void *h = nullptr;
    fn_getid getid = nullptr;
    fn_init init = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_errstr errstr = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.h) return WX_OK;
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
int wx_allreduce_moments_f64(double *buf, int64_t count, void *comm, void *stream)
{ return allreduce_impl(buf, count, kNcclFloat64, comm, stream); }
int wx_allreduce_moments_f32(float *buf, int64_t count, void *comm, void *stream)
{ return allreduce_impl(buf, count, kNcclFloat32, comm, stream); }

}  // extern "C"
