// group.cpp -- mipt_group_*: the in-process multi-GPU form of the path (include/mipt.h).  One host thread drives n devices, as
// SDK/optixMultiGPU does (optixMultiGPU.cpp:562-594: for every state { cudaSetDevice; optixLaunch on its stream }, then a sync per
// device): every device renders its StaticWorkDistribution shard of the frame (sutil/WorkDistribution.h:34-91) into a compact
// float4 buffer, ONE grouped ncclAllGather over RCCL / xGMI hands every device every shard, and a local scatter kernel writes
// the canonical accum / frame buffers (the reference merges through pinned zero-copy host memory, optixMultiGPU.cu:282-292).
// RCCL is bound at run time (dlopen), so libmipt.so has no link dependency on it and single-GPU users never load it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mipt.h"

namespace {

// the few RCCL entry points used, with rccl.h's signatures (ncclResult_t = int, ncclComm_t = opaque pointer, ncclFloat = 7)
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load(std::string& why)
    {
        if (lib) return true;
        for (const char* name : { "librccl.so.1", "librccl.so" }) {
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already holds (PyTorch's), if any
            if (lib) break;
        }
        if (!lib) for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { why = std::string("librccl.so not found: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !AllGather || !GetErrorString) {
            why = "librccl.so lacks an expected symbol"; lib = nullptr; return false;
        }
        return true;
    }
};

struct Member {
    int          device = 0;
    MiptContext* ctx = nullptr;
    hipStream_t  stream = nullptr;
    void*        comm = nullptr;
    void*        compact = nullptr;     // this device's shard: num_samples float4
    void*        gathered = nullptr;    // every shard, rank-major: n * num_samples float4
    size_t       cap = 0;               // float4 entries compact holds
};

}  // namespace

struct MiptGroup {
    std::vector<Member> m;
    Rccl        rccl;
    bool        use_rccl = false;
    uint32_t    width = 0, height = 0;  // of the last launch
    std::string last_error;
    int fail(int code, const std::string& msg) { last_error = msg; return code; }
};

extern "C" {

int mipt_group_create(const int* hip_devices, uint32_t n, MiptLogCallback cb, void* cbdata, int log_level, MiptGroup** out)
{
    if (!out) return MIPT_ERROR_INVALID_VALUE;
    *out = nullptr;
    if (!hip_devices || n == 0 || n > 64) return MIPT_ERROR_INVALID_VALUE;
    for (uint32_t i = 0; i < n; ++i) for (uint32_t j = 0; j < i; ++j) if (hip_devices[i] == hip_devices[j]) return MIPT_ERROR_INVALID_VALUE;
    MiptGroup* g = new MiptGroup();
    g->m.resize(n);
    int rc = MIPT_SUCCESS;
    for (uint32_t i = 0; i < n && rc == MIPT_SUCCESS; ++i) {
        Member& mb = g->m[i];
        mb.device = hip_devices[i];
        rc = mipt_context_create(mb.device, cb, cbdata, log_level, &mb.ctx);
        if (rc == MIPT_SUCCESS && (hipSetDevice(mb.device) != hipSuccess || hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking) != hipSuccess)) rc = MIPT_ERROR_HIP;
    }
    if (rc == MIPT_SUCCESS) {
        // the exchange: RCCL communicators for all members of this process (ncclCommInitAll).  A single device needs no
        // exchange partner, but takes the same calls when the library is there, so that the path is exercised on one GPU.
        std::string why;
        if (g->rccl.load(why)) {
            std::vector<void*> comms(n, nullptr);
            const int r = g->rccl.CommInitAll(comms.data(), (int)n, hip_devices);
            if (r == 0) { for (uint32_t i = 0; i < n; ++i) g->m[i].comm = comms[i]; g->use_rccl = true; }
            else if (n > 1) { g->last_error = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r); rc = MIPT_ERROR_HIP; }
        } else if (n > 1) {
            g->last_error = why; rc = MIPT_ERROR_NOT_CONFIGURED;
        }
    }
    if (rc != MIPT_SUCCESS) {
        if (cb && log_level >= 2 && !g->last_error.empty()) cb(2, "GROUP", g->last_error.c_str(), cbdata);
        mipt_group_destroy(g);
        return rc;
    }
    *out = g;
    return MIPT_SUCCESS;
}

int mipt_group_destroy(MiptGroup* g)
{
    if (!g) return MIPT_ERROR_INVALID_VALUE;
    for (Member& mb : g->m) {
        (void)hipSetDevice(mb.device);
        if (mb.stream) (void)hipStreamSynchronize(mb.stream);
        if (mb.comm && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(mb.comm);
        if (mb.compact) (void)hipFree(mb.compact);
        if (mb.gathered) (void)hipFree(mb.gathered);
        if (mb.stream) (void)hipStreamDestroy(mb.stream);
        if (mb.ctx) (void)mipt_context_destroy(mb.ctx);
    }
    delete g;
    return MIPT_SUCCESS;
}

int mipt_group_size(const MiptGroup* g, uint32_t* n)
{
    if (!g || !n) return MIPT_ERROR_INVALID_VALUE;
    *n = (uint32_t)g->m.size();
    return MIPT_SUCCESS;
}

int mipt_group_member(MiptGroup* g, uint32_t index, MiptContext** ctx, int* hip_device, void** stream)
{
    if (!g || index >= g->m.size()) return MIPT_ERROR_INVALID_VALUE;
    if (ctx) *ctx = g->m[index].ctx;
    if (hip_device) *hip_device = g->m[index].device;
    if (stream) *stream = (void*)g->m[index].stream;
    return MIPT_SUCCESS;
}

const char* mipt_group_last_error_string(const MiptGroup* g) { return g ? g->last_error.c_str() : "null group"; }

int mipt_group_uses_rccl(const MiptGroup* g) { return g && g->use_rccl ? 1 : 0; }

int mipt_group_launch(MiptGroup* g, const MiptParams* params, unsigned int width, unsigned int height)
{
    if (!g || !params) return MIPT_ERROR_INVALID_VALUE;
    const uint32_t n = (uint32_t)g->m.size();
    uint32_t ns = 0;
    int rc = mipt_shard_num_samples(width, height, n, &ns);
    if (rc) return g->fail(rc, "mipt_shard_num_samples failed");
    for (uint32_t i = 0; i < n; ++i) {
        Member& mb = g->m[i];
        if (hipSetDevice(mb.device) != hipSuccess) return g->fail(MIPT_ERROR_HIP, "hipSetDevice failed");
        if (mb.cap < ns) {
            if (hipStreamSynchronize(mb.stream) != hipSuccess) return g->fail(MIPT_ERROR_HIP, "stream sync failed");
            if (mb.compact) (void)hipFree(mb.compact);
            if (mb.gathered) (void)hipFree(mb.gathered);
            mb.compact = mb.gathered = nullptr; mb.cap = 0;
            if (hipMalloc(&mb.compact, (size_t)ns * 16u) != hipSuccess || hipMalloc(&mb.gathered, (size_t)ns * 16u * n) != hipSuccess)
                return g->fail(MIPT_ERROR_OUT_OF_MEMORY, "shard buffers");
            // the padding slots of a shard (tiles beyond the image edge) are never written by the launch: gather zeros, not whatever
            // hipMalloc returned
            if (hipMemsetAsync(mb.compact, 0, (size_t)ns * 16u, mb.stream) != hipSuccess || hipMemsetAsync(mb.gathered, 0, (size_t)ns * 16u * n, mb.stream) != hipSuccess)
                return g->fail(MIPT_ERROR_HIP, "clearing the shard buffers failed");
            mb.cap = ns;
        }
        rc = mipt_launch_shard(mb.ctx, (void*)mb.stream, &params[i], width, height, i, n, mb.compact);
        if (rc) return g->fail(rc, std::string("device ") + std::to_string(mb.device) + ": " + mipt_last_error_string(mb.ctx));
    }
    g->width = width; g->height = height;
    return MIPT_SUCCESS;
}

int mipt_group_allgather(MiptGroup* g, void* const* accum_buffers, void* const* frame_buffers)
{
    if (!g || !accum_buffers) return MIPT_ERROR_INVALID_VALUE;
    if (g->width == 0) return g->fail(MIPT_ERROR_NOT_CONFIGURED, "mipt_group_allgather before mipt_group_launch");
    const uint32_t n = (uint32_t)g->m.size();
    uint32_t ns = 0;
    (void)mipt_shard_num_samples(g->width, g->height, n, &ns);
    if (g->use_rccl) {
        int r = g->rccl.GroupStart();
        for (uint32_t i = 0; i < n && r == 0; ++i) {
            Member& mb = g->m[i];
            if (hipSetDevice(mb.device) != hipSuccess) return g->fail(MIPT_ERROR_HIP, "hipSetDevice failed");
            r = g->rccl.AllGather(mb.compact, mb.gathered, (size_t)ns * 4u, 7 /* ncclFloat */, mb.comm, mb.stream);
        }
        const int r2 = g->rccl.GroupEnd();
        if (r != 0 || r2 != 0) return g->fail(MIPT_ERROR_HIP, std::string("ncclAllGather: ") + g->rccl.GetErrorString(r ? r : r2));
    } else {
        // one member without RCCL: the gathered buffer is its own shard
        Member& mb = g->m[0];
        if (hipSetDevice(mb.device) != hipSuccess ||
            hipMemcpyAsync(mb.gathered, mb.compact, (size_t)ns * 16u, hipMemcpyDeviceToDevice, mb.stream) != hipSuccess)
            return g->fail(MIPT_ERROR_HIP, "shard copy failed");
    }
    for (uint32_t i = 0; i < n; ++i) {
        Member& mb = g->m[i];
        if (!accum_buffers[i]) continue;                       // this device does not need the assembled frame
        const int rc = mipt_unpermute(mb.ctx, (void*)mb.stream, mb.gathered, n, g->width, g->height, accum_buffers[i],
                                      frame_buffers ? frame_buffers[i] : nullptr);
        if (rc) return g->fail(rc, mipt_last_error_string(mb.ctx));
    }
    return MIPT_SUCCESS;
}

int mipt_group_sync(MiptGroup* g)
{
    if (!g) return MIPT_ERROR_INVALID_VALUE;
    for (Member& mb : g->m) {
        if (hipSetDevice(mb.device) != hipSuccess || hipStreamSynchronize(mb.stream) != hipSuccess) return g->fail(MIPT_ERROR_HIP, "stream sync failed");
    }
    return MIPT_SUCCESS;
}

}  // extern "C"
