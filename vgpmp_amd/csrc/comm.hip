// The one collective of the path: an in-place sum of the packed gradient + ELBO pieces over the ranks that share a
// sharded Monte-Carlo sample axis (the reference is single-process and has no counterpart).  RCCL is
// resolved at run time (dlopen): the library has no link-time dependency on it, problem-sharded and single-GPU use
// never touch it, and inside a PyTorch process the RCCL already loaded by torch (soname librccl.so.1) is the one used.
#include "vgpmp_device.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <new>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    bool ok = false;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.ok ? &r : nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) return nullptr;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
    return r.ok ? &r : nullptr;
}

}  // namespace

struct vgpmp_comm {
    ncclComm_t comm;
    int world, rank;
};

extern "C" {

int vgpmp_comm_unique_id(void* id_bytes) {
    if (!id_bytes) return VGPMP_E_ARG;
    static_assert(sizeof(ncclUniqueId) == VGPMP_COMM_ID_BYTES, "id size of include/vgpmp.h");
    Rccl* r = rccl();
    if (!r) return VGPMP_E_COMM;
    return r->GetUniqueId((ncclUniqueId*)id_bytes) == ncclSuccess ? 0 : VGPMP_E_COMM;
}

int vgpmp_comm_init(const void* id_bytes, int32_t world, int32_t rank, vgpmp_comm** comm) {
    if (!id_bytes || !comm) return VGPMP_E_ARG;
    if (world < 1 || rank < 0 || rank >= world) return VGPMP_E_SHAPE;
    Rccl* r = rccl();
    if (!r) return VGPMP_E_COMM;
    ncclUniqueId id;
    ::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    if (r->CommInitRank(&c, world, id, rank) != ncclSuccess) return VGPMP_E_COMM;      // current HIP device
    vgpmp_comm* out = new (std::nothrow) vgpmp_comm{c, world, rank};
    if (!out) { r->CommDestroy(c); return VGPMP_E_COMM; }
    *comm = out;
    return 0;
}

int vgpmp_allreduce_grads(vgpmp_comm* comm, double* dev_buf, size_t count, vgpmp_stream stream) {
    if (!comm || (!dev_buf && count)) return VGPMP_E_ARG;
    if (count == 0) return 0;
    Rccl* r = rccl();
    if (!r) return VGPMP_E_COMM;
    return r->AllReduce(dev_buf, dev_buf, count, ncclFloat64, ncclSum, comm->comm, (hipStream_t)stream) == ncclSuccess
               ? 0 : VGPMP_E_COMM;
}

int vgpmp_comm_destroy(vgpmp_comm* comm) {
    if (!comm) return 0;
    Rccl* r = rccl();
    int rc = (r && r->CommDestroy(comm->comm) == ncclSuccess) ? 0 : VGPMP_E_COMM;
    delete comm;
    return rc;
}

}  // extern "C"
