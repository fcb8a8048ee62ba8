/* stub_rccl.c -- a RECORDING stand-in for librccl.so (test infrastructure, tests/test_gpu_comm_stub.py): the library's wx_comm_* entry
 * points load RCCL through dlopen (csrc/wx_comm.hip: WX_RCCL_LIB names the file), and on this pool no box has two GPUs, so the
 * point-to-point schedule of wx_allgatherv_out_* -- counts, element offsets, peers, the order inside the group -- has never met a
 * communicator of more than one rank.  This file gives it one: ncclCommInitRank remembers (nranks, rank), ncclSend / ncclRecv move no
 * data and write down what they were asked to do; the test reads the log and checks it against the contract of include/waveletsext_hip.h.
 * Signatures: rccl.h of ROCm 7 (ncclResult_t = int, ncclComm_t = pointer, ncclUniqueId = 128 bytes by value). */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct { int nranks, rank; } StubComm;
typedef struct { int kind; const void *ptr; size_t count; int dtype, peer; const void *comm; void *stream; int group_depth; } StubCall;

static StubCall g_log[4096];
static int g_n = 0, g_depth = 0, g_groups = 0;

static void put(int kind, const void *p, size_t count, int dt, int peer, const void *comm, void *st)
{
    if (g_n < 4096) { StubCall c = {kind, p, count, dt, peer, comm, st, g_depth}; g_log[g_n++] = c; }
}

int ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5a, sizeof *id); return 0; }
int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank)
{
    (void)id;
    StubComm *c = (StubComm *)malloc(sizeof *c);
    c->nranks = nranks; c->rank = rank;
    *comm = c;
    return 0;
}
int ncclCommDestroy(void *comm) { free(comm); return 0; }
int ncclCommCount(void *comm, int *n) { *n = ((StubComm *)comm)->nranks; return 0; }
int ncclCommUserRank(void *comm, int *r) { *r = ((StubComm *)comm)->rank; return 0; }
const char *ncclGetErrorString(int e) { (void)e; return "stub"; }
int ncclGroupStart(void) { ++g_depth; ++g_groups; return 0; }
int ncclGroupEnd(void) { --g_depth; return 0; }
int ncclSend(const void *buf, size_t count, int dt, int peer, void *comm, void *st) { put(1, buf, count, dt, peer, comm, st); return 0; }
int ncclRecv(void *buf, size_t count, int dt, int peer, void *comm, void *st) { put(2, buf, count, dt, peer, comm, st); return 0; }
int ncclAllGather(const void *s, void *r, size_t count, int dt, void *comm, void *st) { put(3, s, count, dt, -1, comm, st); put(4, r, count, dt, -1, comm, st); return 0; }
int ncclAllReduce(const void *s, void *r, size_t count, int dt, int op, void *comm, void *st) { put(5, s, count, dt, op, comm, st); put(6, r, count, dt, op, comm, st); return 0; }

/* the test's side */
int stub_ncalls(void) { return g_n; }
int stub_ngroups(void) { return g_groups; }
int stub_depth(void) { return g_depth; }
void stub_reset(void) { g_n = 0; g_groups = 0; }
int stub_call(int i, int *kind, const void **ptr, size_t *count, int *dtype, int *peer, int *group_depth)
{
    if (i < 0 || i >= g_n) return -1;
    *kind = g_log[i].kind; *ptr = g_log[i].ptr; *count = g_log[i].count; *dtype = g_log[i].dtype; *peer = g_log[i].peer;
    *group_depth = g_log[i].group_depth;
    return 0;
}
