// comm.hip — the one real exchange step of the path (anchor k-means all-reduces its int64 sums),
// issued by the library itself on the launch stream: dg_comm_init / dg_comm_allreduce_i64 / dg_comm_destroy (§8(b)).
// RCCL is bound at run time (dlopen), preferring the copy the process has already loaded (torch links librccl.so), so
// libdetgeom.so has no link-time dependency on it and every other entry point works where RCCL is absent.
#include "common.hpp"
#include <dlfcn.h>
#include <mutex>
#include <string.h>
#include <rccl/rccl.h>   // types and constants only

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

void rccl_bind()
{
    static const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;       // the copy already in the process (torch's)
    if (!h)
        for (const char *nm : names)
            if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so not found (%s)", dlerror()); return; }
#define DG_BIND(field, sym)                                                                                  \
    do {                                                                                                     \
        *(void **)(&g_rccl.field) = dlsym(h, sym);                                                           \
        if (!g_rccl.field) { snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so lacks %s", sym); return; } \
    } while (0)
    DG_BIND(GetUniqueId, "ncclGetUniqueId");
    DG_BIND(CommInitRank, "ncclCommInitRank");
    DG_BIND(AllReduce, "ncclAllReduce");
    DG_BIND(CommDestroy, "ncclCommDestroy");
    DG_BIND(GetErrorString, "ncclGetErrorString");
#undef DG_BIND
    g_rccl.handle = h;
}

int rccl_ready(const char *fn)
{
    std::call_once(g_rccl_once, rccl_bind);
    if (!g_rccl.handle) { dg_set_error("%s: %s", fn, g_rccl.why); return DG_ERCCL; }
    return DG_OK;
}
int rccl_fail(const char *fn, const char *what, ncclResult_t r)
{
    dg_set_error("%s: %s failed: %s", fn, what, g_rccl.GetErrorString(r));
    return DG_ERCCL;
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == DG_COMM_ID_BYTES, "DG_COMM_ID_BYTES must be RCCL's unique-id size");

extern "C" int dg_comm_unique_id(void *id)
{
    DG_REQUIRE(id, "comm_unique_id: null id");
    int rc = rccl_ready("comm_unique_id");
    if (rc) return rc;
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail("comm_unique_id", "ncclGetUniqueId", r);
    memcpy(id, &u, sizeof(u));
    return DG_OK;
}

extern "C" int dg_comm_init(dg_comm_t **out, const void *id, int rank, int nranks)
{
    DG_REQUIRE(out && id && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: need out, id and 0 <= rank < nranks");
    *out = nullptr;
    int rc = rccl_ready("comm_init");
    if (rc) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { dg_set_error("comm_init: hipGetDevice failed"); return DG_EHIP; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&c, nranks, u, rank);
    if (r != ncclSuccess) return rccl_fail("comm_init", "ncclCommInitRank", r);
    dg_comm *h = new dg_comm();
    h->comm = c;
    h->rank = rank;
    h->nranks = nranks;
    h->device = dev;
    *out = h;
    return DG_OK;
}

extern "C" int dg_comm_rank(const dg_comm_t *c) { return c ? c->rank : -1; }
extern "C" int dg_comm_size(const dg_comm_t *c) { return c ? c->nranks : -1; }

extern "C" int dg_comm_allreduce_i64(dg_comm_t *c, int64_t *buf, int64_t count, dg_stream_t s)
{
    DG_REQUIRE(c && c->comm, "comm_allreduce_i64: null communicator");
    DG_REQUIRE(count >= 0 && (count == 0 || buf), "comm_allreduce_i64: bad buffer");
    if (count == 0) return DG_OK;
    ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, ncclInt64, ncclSum, (ncclComm_t)c->comm, (hipStream_t)s);
    if (r != ncclSuccess) return rccl_fail("comm_allreduce_i64", "ncclAllReduce", r);
    return DG_OK;
}

extern "C" int dg_comm_destroy(dg_comm_t *c)
{
    if (!c) return DG_OK;
    int rc = DG_OK;
    if (c->graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)c->graph_exec);
    if (c->comm) {
        ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)c->comm);
        if (r != ncclSuccess) rc = rccl_fail("comm_destroy", "ncclCommDestroy", r);
    }
    delete c;
    return rc;
}
