// Device context and the RCCL point-to-point / broadcast wrappers of the C ABI (ss_create / ss_destroy,
// ss_rccl_{send,recv,bcast}).
//
// Context.  The library keeps no state besides (a) the thread-local error string, (b) the runtime tuning knobs and
// (c) the tile-configuration table of the MFMA GEMM / conv kernels — (b) and (c) belong to ONE device: a process drives
// one GPU (one process per GPU, torch.distributed / RCCL between them).  ss_create(device) validates the device
// (gfx950), makes it current and returns the handle that owns that state; ss_destroy drops the table.  The op entry
// points take no handle (they act on the current HIP device, like the HIP runtime itself).
//
// RCCL.  The exchange step of the multi-GPU slot ring (seedstory/parallel.py: regressed image feature, KV-cache rows)
// as flat C calls for hosts that do not go through torch.distributed.  librccl is resolved at run time with dlopen (the
// copy already loaded by the host process — e.g. torch's — is reused), so the library itself has no link-time
// dependency on it and loads on boxes without RCCL; the calls fail with SS_ESTATE there.
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

#include "ss_common.h"

struct ss_context {
    int device;
    int cu_count;
    size_t hbm_bytes;
};

struct ss_rccl {
    ncclComm_t comm;
    int rank, nranks;
};

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (api.lib) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
            api.Send = (decltype(api.Send))dlsym(api.lib, "ncclSend");
            api.Recv = (decltype(api.Recv))dlsym(api.lib, "ncclRecv");
            api.Broadcast = (decltype(api.Broadcast))dlsym(api.lib, "ncclBroadcast");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Send || !api.Recv || !api.Broadcast)
                api.lib = nullptr;
        }
    }
    return api.lib ? &api : nullptr;
}

int rccl_check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return SS_OK;
    RcclApi* a = rccl_api();
    ss::set_error("RCCL error %d (%s) at %s", (int)r, a && a->GetErrorString ? a->GetErrorString(r) : "?", what);
    return SS_EHIP;
}

bool rccl_dtype(int dtype, ncclDataType_t* out) {
    switch (dtype) {
        case SS_F32: *out = ncclFloat32; return true;
        case SS_BF16: *out = ncclBfloat16; return true;
        case SS_F16: *out = ncclFloat16; return true;
        default: return false;
    }
}

}  // namespace

extern "C" {

int ss_create(int device, ss_context** out) {
    SS_REQUIRE(out, "ss_create: out == NULL");
    *out = nullptr;
    int n = 0;
    SS_HIP(hipGetDeviceCount(&n));
    SS_REQUIRE(device >= 0 && device < n, "ss_create: device %d out of range (%d visible)", device, n);
    hipDeviceProp_t p;
    SS_HIP(hipGetDeviceProperties(&p, device));
    SS_REQUIRE(strstr(p.gcnArchName, "gfx950") != nullptr, "ss_create: device %d is %s, this library is built for gfx950 only",
               device, p.gcnArchName);
    SS_HIP(hipSetDevice(device));
    ss_context* c = new ss_context{device, p.multiProcessorCount, (size_t)p.totalGlobalMem};
    *out = c;
    return SS_OK;
}

int ss_context_info(const ss_context* ctx, int64_t out[3]) {
    SS_REQUIRE(ctx && out, "ss_context_info: bad arguments");
    out[0] = ctx->device; out[1] = ctx->cu_count; out[2] = (int64_t)ctx->hbm_bytes;
    return SS_OK;
}

void ss_destroy(ss_context* ctx) {
    // The GEMM tile table is process-global (ops take no context handle) and every entry in it was measured on gfx950,
    // the only architecture ss_create accepts — it outlives a context: other users of the library in this process keep
    // their tuned tiles.  ss_tune_clear() is the explicit way to drop it.
    delete ctx;
}

int ss_rccl_unique_id(void* id_out_128_bytes) {
    SS_REQUIRE(id_out_128_bytes, "ss_rccl_unique_id: NULL buffer");
    RcclApi* a = rccl_api();
    if (!a) { ss::set_error("librccl could not be loaded"); return SS_ESTATE; }
    ncclUniqueId id;
    int rc = rccl_check(a->GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    memcpy(id_out_128_bytes, &id, sizeof(id));
    return SS_OK;
}

int ss_rccl_init(const void* id_128_bytes, int nranks, int rank, ss_rccl** out) {
    SS_REQUIRE(id_128_bytes && out && nranks > 0 && rank >= 0 && rank < nranks, "ss_rccl_init: bad arguments");
    *out = nullptr;
    RcclApi* a = rccl_api();
    if (!a) { ss::set_error("librccl could not be loaded"); return SS_ESTATE; }
    ncclUniqueId id;
    memcpy(&id, id_128_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    int rc = rccl_check(a->CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank");
    if (rc) return rc;
    *out = new ss_rccl{comm, rank, nranks};
    return SS_OK;
}

void ss_rccl_destroy(ss_rccl* c) {
    if (!c) return;
    RcclApi* a = rccl_api();
    if (a && c->comm) a->CommDestroy(c->comm);
    delete c;
}

int ss_rccl_send(ss_rccl* c, const void* buf, int64_t count, int dtype, int peer, void* stream) {
    SS_REQUIRE(c && buf && count >= 0 && peer >= 0 && peer < c->nranks && peer != c->rank, "ss_rccl_send: bad arguments");
    ncclDataType_t dt;
    SS_REQUIRE(rccl_dtype(dtype, &dt), "ss_rccl_send: unsupported dtype %d", dtype);
    return rccl_check(rccl_api()->Send(buf, (size_t)count, dt, peer, c->comm, (hipStream_t)stream), "ncclSend");
}

int ss_rccl_recv(ss_rccl* c, void* buf, int64_t count, int dtype, int peer, void* stream) {
    SS_REQUIRE(c && buf && count >= 0 && peer >= 0 && peer < c->nranks && peer != c->rank, "ss_rccl_recv: bad arguments");
    ncclDataType_t dt;
    SS_REQUIRE(rccl_dtype(dtype, &dt), "ss_rccl_recv: unsupported dtype %d", dtype);
    return rccl_check(rccl_api()->Recv(buf, (size_t)count, dt, peer, c->comm, (hipStream_t)stream), "ncclRecv");
}

int ss_rccl_bcast(ss_rccl* c, void* buf, int64_t count, int dtype, int root, void* stream) {
    SS_REQUIRE(c && buf && count >= 0 && root >= 0 && root < c->nranks, "ss_rccl_bcast: bad arguments");
    ncclDataType_t dt;
    SS_REQUIRE(rccl_dtype(dtype, &dt), "ss_rccl_bcast: unsupported dtype %d", dtype);
    return rccl_check(rccl_api()->Broadcast(buf, buf, (size_t)count, dt, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
}

}  // extern "C"
