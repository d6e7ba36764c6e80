// The data-parallel exchange through the C ABI: one RCCL communicator owned by the library's caller, its all-reduce enqueued on the
// CALLER's stream -- the compute stream -- so that inside a captured training step it is a plain node between the gradient reduction
// and the launch that applies the update (no side stream, no fork / join around the collective: torch's ProcessGroupNCCL runs its
// collectives on a stream of its own, 17 us of fork / join per step in the captured form).
// RCCL is loaded at run time from the path the caller names (the copy PyTorch ships, so that both share one RCCL): the library itself
// has no link-time dependency on it and loads on boxes without it.  Replaces nothing in the reference -- its multi-GPU mode is keras'
// multi_gpu_model (DLWP/model/models.py:369-374), host-side averaging over replicas.
#include <dlfcn.h>
#include <mutex>
#include <string.h>
#include "common.h"

namespace dlwpcs {
namespace {

// the slice of nccl.h used here (RCCL 2.x ABI; /opt/rocm/include/rccl/rccl.h)
struct UniqueId { char internal[128]; };
typedef void *Comm;
typedef int Result;
enum { kFloat32 = 7, kSum = 0 };

struct Api {
    void *handle = nullptr;
    Result (*GetUniqueId)(UniqueId *) = nullptr;
    Result (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(Result) = nullptr;
    Result (*GetVersion)(int *) = nullptr;
};
Api g_api;
std::mutex g_mu;

const char *errstr(Result r) { return g_api.GetErrorString ? g_api.GetErrorString(r) : "?"; }

}  // namespace
}  // namespace dlwpcs

using namespace dlwpcs;

extern "C" int dlwpcs_comm_load(const char *librccl_path) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_api.handle) return DLWPCS_OK;
    if (!librccl_path) return fail(DLWPCS_E_INVALID, "comm_load: null path");
    void *h = dlopen(librccl_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(DLWPCS_E_UNSUPPORTED, "comm_load: %s", dlerror());
    Api a;
    a.handle = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.GetVersion = (decltype(a.GetVersion))dlsym(h, "ncclGetVersion");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString)
        return fail(DLWPCS_E_UNSUPPORTED, "comm_load: %s does not export the NCCL entry points", librccl_path);
    g_api = a;
    return DLWPCS_OK;
}

extern "C" int dlwpcs_comm_unique_id(void *id128) {
    if (!g_api.handle) return fail(DLWPCS_E_UNSUPPORTED, "comm_unique_id: dlwpcs_comm_load first");
    if (!id128) return fail(DLWPCS_E_INVALID, "comm_unique_id: null output");
    UniqueId id;
    const Result r = g_api.GetUniqueId(&id);
    if (r != 0) return fail(DLWPCS_E_LAUNCH, "comm_unique_id: %s", errstr(r));
    memcpy(id128, id.internal, sizeof(id.internal));
    return DLWPCS_OK;
}

extern "C" int dlwpcs_comm_init(void **comm, const void *id128, int rank, int world) {
    if (!g_api.handle) return fail(DLWPCS_E_UNSUPPORTED, "comm_init: dlwpcs_comm_load first");
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return fail(DLWPCS_E_INVALID, "comm_init: bad arguments (rank %d of %d)", rank, world);
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    Comm c = nullptr;
    const Result r = g_api.CommInitRank(&c, world, id, rank);      // (the communicator lives on the calling thread's current device)
    if (r != 0) return fail(DLWPCS_E_LAUNCH, "comm_init: ncclCommInitRank: %s", errstr(r));
    *comm = c;
    return DLWPCS_OK;
}

extern "C" int dlwpcs_comm_destroy(void *comm) {
    if (!comm) return DLWPCS_OK;
    if (!g_api.handle) return fail(DLWPCS_E_UNSUPPORTED, "comm_destroy: dlwpcs_comm_load first");
    const Result r = g_api.CommDestroy((Comm)comm);
    if (r != 0) return fail(DLWPCS_E_LAUNCH, "comm_destroy: %s", errstr(r));
    return DLWPCS_OK;
}

extern "C" int dlwpcs_allreduce_f32(void *comm, float *buf, size_t n, dlwpcs_stream_t stream) {
    if (!g_api.handle) return fail(DLWPCS_E_UNSUPPORTED, "allreduce_f32: dlwpcs_comm_load first");
    if (!comm || (!buf && n)) return fail(DLWPCS_E_INVALID, "allreduce_f32: null communicator / buffer");
    if (n == 0) return DLWPCS_OK;
    const Result r = g_api.AllReduce(buf, buf, n, kFloat32, kSum, (Comm)comm, (hipStream_t)stream);
    if (r != 0) return fail(DLWPCS_E_LAUNCH, "allreduce_f32: ncclAllReduce: %s", errstr(r));
    return DLWPCS_OK;
}
