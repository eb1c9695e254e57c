// mipt_api.cpp -- the C ABI of libmipt.so (include/mipt.h): context, acceleration structures,
// hit-group records, and the host side of the wavefront launch.
//
// Mirrors the per-sample "state" functions of the reference (SDK/optixPathTracer/optixPathTracer.cpp):
//   createContext :555-573   -> mipt_context_create
//   buildMeshAccel :576-684  -> mipt_accel_build          (BVH owned by the library)
//   createSBT :829-898       -> mipt_sbt_set_hitgroups / mipt_sbt_set_miss
//   launchSubframe :488-511  -> mipt_launch               (params passed by value to the kernels,
//                                                          no cudaMemcpyAsync of Params needed)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mipt.h"
#include "bvh_build.h"
#include "detmath.h"
#include "kernels.h"

using namespace mipt;

// per-wave probe builds (make probe-waves PROBE_FLAGS=-DMIPT_PROBE_WAVES=1: this file and kernels.hip are compiled with the flag):
// only they read a record buffer's device address from the environment -- the shipped library has no such hook
#ifndef MIPT_PROBE_WAVES
#define MIPT_PROBE_WAVES 0
#endif
#ifndef MIPT_TOP_CACHE              // experiment (kernels.hip): a top-of-tree copy of the packed mirror behind the blob's mirrors
#define MIPT_TOP_CACHE 0
#endif

// the kept layouts: each figure below is what the reference's own headers state (oracle/ref_kat.cpp prints sizeof / offsetof from
// optixPathTracer.h, cuda/whitted.h, BufferView.h, Light.h, optix_types.h into tests/golden/ref_kat.json "layouts";
// tests/test_host_abi.py compares include/mipt.h, compiled as C, and the ctypes mirrors against that file)
static_assert(sizeof(MiptParams) == 152, "Params layout must match optixPathTracer.h:91-107");
static_assert(offsetof(MiptParams, accum_buffer) == 8 && offsetof(MiptParams, frame_buffer) == 16, "Params layout");
static_assert(offsetof(MiptParams, width) == 24 && offsetof(MiptParams, samples_per_launch) == 32, "Params layout");
static_assert(offsetof(MiptParams, eye) == 36 && offsetof(MiptParams, U) == 48 && offsetof(MiptParams, W) == 72, "Params layout");
static_assert(offsetof(MiptParams, light) == 84 && offsetof(MiptParams, handle) == 144, "Params layout");
static_assert(sizeof(MiptParallelogramLight) == 60, "ParallelogramLight layout");
static_assert(sizeof(MiptHitGroupData) == 32, "HitGroupData layout");
static_assert(sizeof(MiptBvhNode) == 64 && sizeof(MiptBvhTri) == 48, "BVH layout");
static_assert(sizeof(MiptInstance) == 80 && offsetof(MiptInstance, traversableHandle) == 64, "OptixInstance layout (include/optix_types.h)");
static_assert(sizeof(MiptBvhNode32) == 32, "packed node layout");
// whitted::LaunchParams offsets probed from the reference build
static_assert(sizeof(MiptWhittedLaunchParams) == 128 && sizeof(MiptBufferView) == 16, "whitted::LaunchParams layout");
static_assert(offsetof(MiptWhittedLaunchParams, accum_buffer) == 16 && offsetof(MiptWhittedLaunchParams, max_depth) == 32, "whitted::LaunchParams layout");
static_assert(offsetof(MiptWhittedLaunchParams, eye) == 40 && offsetof(MiptWhittedLaunchParams, lights) == 88, "whitted::LaunchParams layout");
static_assert(offsetof(MiptWhittedLaunchParams, miss_color) == 104 && offsetof(MiptWhittedLaunchParams, handle) == 120, "whitted::LaunchParams layout");

namespace {

struct Accel {
    HostBvh   host;                // scratch of the host builder (emptied after the upload)
    uint32_t  num_nodes = 0, num_tris = 0, depth = 0, leaf_max = 0;
    double    seconds = 0.0;
    int       builder = 0;         // 0 host binned SAH, 1 device LBVH, 2 device SAH, 3 host SAH with spatial splits (num_tris > n_prims: references)
    float4*   d_nodes = nullptr;   // one allocation: nodes | triangles | 64 B pad
    float4*   d_tris  = nullptr;   // = d_nodes + num_nodes * 4
    uint32_t  tri_byte_off = 0;    // 0 when the blob does not fit 32-bit offsets (then no streaming kernel)
    uint32_t  blob_bytes = 0;
    uint32_t  node32_off = 0;      // byte offset of the packed 32-byte BVH2 mirror behind the triangles (node formats 1 and 2); 0 = none
    uint32_t  node4_off = 0;       // ... and of the BVH4 mirror behind that (format 2)
    uint32_t  top_off = 0;         // experiment MIPT_TOP_CACHE
    uint32_t  node_fmt = 0;        // the format in effect: 0 fp32 only, 1 + packed BVH2 mirror, 2 + BVH4 mirror
    uint32_t  stack_need = 0;      // format 2: the most entries a lane's stack can hold while walking the BVH4 (<= 32 by construction)
    uint32_t  num_nodes4 = 0;
    uint32_t* d_mat   = nullptr;
    uint8_t*  d_height = nullptr;  // height of every node (topology only: made at build time, reused by every refit and by the BVH4 collapse)
    float*    d_cells = nullptr;        // MIPT_OPT_BUILD = 4, triangles cut: per triangle record the cell its box was clipped to (6 floats): a refit clips to it again
    uint32_t* d_prim_slot = nullptr;    // per ORIGINAL primitive: the position of its triangle record (host builder; null: not kept) -- the Whitted any-hit kernels
                                        // re-test a candidate's triangle through it (u, v, geometry flags of a key they hold)
    uint32_t  n_prims = 0;
    uint64_t  bytes   = 0;
    // two-level structure (mipt_accel_build_instances): d_nodes then holds the BVH over the instance boxes
    bool                       is_ias = false;
    DeviceInstance*            d_inst = nullptr;
    uint32_t*                  d_order = nullptr;
    std::vector<MiptBvhNode>   h_tlas;           // host copies for mipt_accel_export_instances
    std::vector<uint32_t>      h_order, h_base;
    std::vector<float>         h_inv;            // 12 per instance
    std::vector<unsigned long long> gas;         // child handle per instance
    uint32_t                   users = 0;        // instance structures that reference this mesh BVH
    // entry-pool form of a two-level structure (DeviceBvh::inst_fast): d_nodes is then ONE allocation, d_inst lies inside it
    bool                       inst_fast = false;
    bool                       stale = false;    // a mesh this structure references was refitted since: mipt_accel_update_instances first
    bool                       inst_slots = false;   // every instance record carries its mesh's d_prim_slot
    uint32_t                   inst_off = 0, inst_node4_off = 0, inst_tri_off = 0, inst_blob_bytes = 0, inst_root4_off = 0;
};

enum TimedKind { T_RAYGEN = 0, T_CLOSEST, T_SHADE, T_ANY, T_RESOLVE, T_KINDS };

}  // namespace

struct MiptContext {
    int             device = 0;
    int             num_cus = 256;
    MiptLogCallback cb = nullptr;
    void*           cbdata = nullptr;
    int             log_level = 0;
    std::string     last_error;

    std::unordered_map<unsigned long long, Accel*> accels;
    unsigned long long next_handle = 0x4d495054ull << 16;   // "MIPT"

    MiptHitGroupData* d_hitgroups = nullptr;
    uint32_t          n_hitgroups = 0;
    const void*       common_vertices = nullptr;   // set when all records share one vertex pointer
    MiptMissData      miss{};

    // options
    uint32_t max_depth = 0;
    int      count_visits = 0;
    int      blocks_per_cu = 0;
    int      batch_samples = 0;
    int      persistent = 3;
    int      build_mode = 2;              // MIPT_OPT_BUILD: 2 (default, round 4) on the device, top-down binned SAH + SAH treelets: the host builder's trees at a
                                          // fourteenth of its time; 1 on the device, Morton codes + radix tree + SAH treelets (fastest build: lbvh_build.hip);
                                          // 0 host binned SAH (bvh_build.cpp; the fallback of the other two for tiny inputs and trees deeper than the stack)
    int      split_below = 128;           // MIPT_OPT_BUILD = 4: ranges of at most this many references are split by objects only (env MIPT_SPLIT_BELOW, 64..2048)
    float    sbvh_extra = 0.5f;           // MIPT_OPT_BUILD = 3: references the spatial splits may add, as a fraction of the triangle count (env MIPT_SBVH_EXTRA)
    uint32_t build_host_below = 2048;     // MIPT_OPT_BUILD = 2: inputs of fewer triangles go to the host builder (env MIPT_BUILD_HOST_BELOW; 0: everything on the device)
    int      lbvh_treelets = 1;           // env MIPT_LBVH_TREELETS=0: the plain radix tree (A/B of the SAH treelets, lbvh_build.hip)
    int      lbvh_leaf = 4;               // env MIPT_LBVH_LEAF: subtrees of at most this many triangles become leaves of the device LBVH (1 .. 4)
    int      node_format = 2;             // MIPT_OPT_NODE_FORMAT: 2 fp16 boxes + BVH4 mirror, 1 fp16 boxes + packed 32-byte BVH2 mirror (where the
                                          // scene suits fp16 boxes), 0 fp32 boxes only
    uint32_t tail_paths = 1048576;         // MIPT_OPT_TAIL_PATHS (sweep: profiles/r01_tail_paths.txt)
    bool     bounces_on_device = false;   // launch statistics: iterations counted by k_advance / k_tail
    int      tile_group = 256;           // consecutive tiles dealt to one XCD at a time (env MIPT_TILE_GROUP)
    bool     tile_group_set = false;     // ... chosen by the environment: also for the cost-ordered launches, which default to 96
    int      tile_shift = 4;              // path tracer, full frames: 64 paths cover 16 x 4 pixels (8 x 8: closest-hit kernel of the
                                          // headline frame 0.440 instead of 0.413 ms; the Whitted and imgui_test raygens keep 8 x 8,
                                          // which is 1-3 % faster on their scenes)
    // whitted.cu launches: geometry arrays of the scene (caller-owned) + material table (library copy)
    WhGeometry        wh_geo{};
    MiptMaterialData* d_wh_materials = nullptr;
    uint32_t          n_wh_materials = 0;
    WhTexture*        d_wh_textures = nullptr;          // descriptor table on the device
    std::vector<void*> wh_texels;                       // device copies of the texel arrays
    uint32_t          n_wh_textures = 0;
    WhInstGeo*        d_wh_inst = nullptr;              // mipt_whitted_set_geometry_instanced: instance records, per-mesh tables,
    WhTables*         d_wh_meshes = nullptr;            // instance of every flattened primitive (library copies)
    uint32_t*         d_wh_prim_inst = nullptr;
    std::vector<uint32_t> wh_inst_base;                 // ... and their first flattened primitive + the total: compared with the structure a launch is given
    uint64_t          wh_inst_total = 0;
    bool              wh_any_hit = false, wh_blend = false;   // a MASK / BLEND material exists
    float4*           wh_e1 = nullptr;                  // second float4 per occlusion entry (capacity im_cap)
    float4*           wh_group = nullptr;               // per hit (capacity wh_group_cap)
    size_t            wh_group_cap = 0;
    float4*           wh_levels = nullptr;              // [MAX_TRACE_DEPTH][pixels] when BLEND materials exist
    size_t            wh_levels_cap = 0;
    // SDK/imgui_test and whitted launches: K occlusion rays per primary hit
    ShadowQueue im_sq{};
    uint32_t*   im_flags = nullptr;
    size_t      im_cap = 0;
    int      refill_min = 16, steal_min = 24, burst_bulk = 12, burst_dry = 8;   // tools/tune.py sweeps, profiles/r01_tuning*.txt
    // cost-ordered tiles of the primary closest-hit launch (TraceArgs::tile_order / tile_cost / tile_hist; TileOrderJob): the tiled
    // kernel records a cost class per tile, the first workgroups of the shade launch turn them into the next launch's order
    uint8_t*    d_tile_class = nullptr;  // written by the last primary launch that ran the tiled kernel
    uint32_t*   d_tile_hist = nullptr;   // 2 x [chunks][kOrderClasses]: the set a launch fills was cleared by the shade launch before it
    uint32_t    tile_hist_set = 0;       // the set the coming tiled launch fills
    uint32_t*   d_tile_order = nullptr;  // slot -> tile for the coming launch
    uint32_t    tile_n = 0;              // tiles the arrays cover
    uint32_t    tile_map = 0;            // world << 16 | rank of the launch the order was made by (another path -> pixel map: start over)
    bool        tile_order_valid = false;    // d_tile_order was made by a launch of the same tile count
    uint32_t*   d_tile_width = nullptr;  // the class width the tiled kernel uses, adapted by the order job (starts at tile_class_width)
    bool        tile_class_fixed = false;    // MIPT_TILE_CLASS given: that width, not adapted
    uint32_t    tile_class_width = 6;    // env MIPT_TILE_CLASS: cost units per class (headline frame, 16 classes: 6 / 8 / 12 -> 0.349 / 0.354 / 0.381 ms)
    uint32_t  tail_epoch = 0;            // TailArgs::epoch of the last path-kernel launch that made its own rays
    bool      debug_epoch_fault = false; // env MIPT_DEBUG_EPOCH_FAULT: the next such launches publish no epoch (tests the fault report)
    int       tail_refill = 1;           // env MIPT_TAIL_REFILL (TraceOpts::tail_refill)
    uint32_t  profile_launch = 0;        // launches since MIPT_OPT_PROFILE was set (mode 3 brackets every 8th)
    int       direct_resolve = 1;        // env MIPT_DIRECT_RESOLVE=0: always a k_resolve launch (A/B)
    int       frame1 = 0;                // env MIPT_FRAME1=1: one-bounce passes as ONE launch of tile waves (k_frame1).  Measured SLOWER -- headline frame
                                         // 0.654 against 0.549 ms for the three wavefront launches: the tiles whose rays are long are the tiles that cast
                                         // occlusion rays, and ~29 more pool rounds behind them lengthen exactly the waves the launch ends on -- so off
    bool      advance_pending = false;   // the last launch ended without its final k_advance step (mipt_get_launch_stats applies it)
    bool      bins_live = false;         // the last launch was the path tracer's: DeviceCounters::bins belong to its statistics
    int       fuse_raygen = 1;           // env MIPT_FUSE_RAYGEN=0: always a k_raygen launch of its own (A/B)
    int       tiled_bvh4 = 0;            // env MIPT_TILED_BVH4: experiments (kernels.hip launch_trace)
    int       inst_form = 0;             // env MIPT_INST_FORM: traversal form over two-level structures (TraceOpts::inst_form)
    int       tile_order_mode = 1;       // MIPT_OPT_TILE_ORDER: 0 image order, 1 (default) by the previous launch's costs
    int      profile = 0;

    // scratch arena of the build / refit calls (grow-only, bump-allocated per call): mipt_accel_update allocates nothing in its steady
    // state -- hipMalloc / hipFree of tens of MB cost more than the refit's kernels (round 3: 5.5 ms of a 1 M-triangle refit)
    char*  scratch = nullptr;
    size_t scratch_cap = 0, scratch_off = 0;
    void   scratch_reset() { scratch_off = 0; }
    void*  scratch_take(size_t bytes)
    {
        const size_t at = (scratch_off + 255) & ~(size_t)255;
        if (!scratch || at + bytes > scratch_cap) return nullptr;
        scratch_off = at + bytes;
        return scratch + at;
    }
    hipError_t scratch_ensure(size_t bytes)            // (only between calls' uses: the old block is released)
    {
        if (bytes <= scratch_cap) return hipSuccess;
        if (scratch) { (void)hipDeviceSynchronize(); (void)hipFree(scratch); scratch = nullptr; scratch_cap = 0; }
        const size_t want = bytes + bytes / 8 + 4096;
        const hipError_t e = hipMalloc((void**)&scratch, want);
        if (e == hipSuccess) scratch_cap = want;
        return e;
    }

    // launch buffers
    size_t          cap_paths = 0, cap_slots = 0;
    PathQueue       q[2]{};
    ShadowQueue     sq{};
    float4*         hits = nullptr;
    float4*         result = nullptr;
    uint32_t*       rg_seed = nullptr;
    DeviceCounters* d_ctr = nullptr;
    uint32_t*       h_pinned = nullptr;      // kRing live counts (run_wavefront)
    hipEvent_t      ring_ev[4] = {};
    hipStream_t     last_stream = nullptr;
    uint32_t        last_bounces = 0;

    // per-kernel timing (MIPT_OPT_PROFILE)
    struct Timed { hipEvent_t a, b; int kind; };
    std::vector<Timed>      timed;
    std::vector<hipEvent_t> event_pool;
    size_t                  events_used = 0;

    void log(unsigned level, const char* tag, const char* fmt, ...)
    {
        if (!cb || (int)level > log_level) return;
        char buf[1024];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        cb(level, tag, buf, cbdata);
    }
    int fail(int code, const char* fmt, ...)
    {
        char buf[1024];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        last_error = buf;
        if (cb && log_level >= 2) cb(2, "ERROR", buf, cbdata);
        return code;
    }
    hipEvent_t get_event()
    {
        if (events_used == event_pool.size()) {
            hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr;
            event_pool.push_back(e);
        }
        return event_pool[events_used++];
    }
};

#define HIPCHK(ctx, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return (ctx)->fail(e_ == hipErrorOutOfMemory ? MIPT_ERROR_OUT_OF_MEMORY : MIPT_ERROR_HIP, \
                               "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

namespace {

constexpr uint32_t kRing = 4, kLag = 2;     // run_wavefront's lagged read of the live-path count

struct TimedScope {
    MiptContext* c; hipStream_t st; int kind; bool on = false; hipEvent_t a = nullptr, b = nullptr;
    TimedScope(MiptContext* c_, hipStream_t st_, int kind_) : c(c_), st(st_), kind(kind_)
    {
        on = c->profile == 1 || ((c->profile == 2 || (c->profile == 3 && c->profile_launch % 8u == 0u)) && (kind_ == T_CLOSEST || kind_ == T_ANY));
        if (on) { a = c->get_event(); b = c->get_event(); if (a) (void)hipEventRecord(a, st); }
    }
    ~TimedScope()
    {
        if (on && a && b) { (void)hipEventRecord(b, st); c->timed.push_back({ a, b, kind }); }
    }
};

bool is_device_pointer(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof attr);
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// copies `bytes` from a host-or-device pointer into host storage
int fetch_to_host(MiptContext* ctx, const void* src, size_t bytes, std::vector<uint8_t>& dst)
{
    dst.resize(bytes);
    if (bytes == 0) return MIPT_SUCCESS;
    if (is_device_pointer(src)) HIPCHK(ctx, hipMemcpy(dst.data(), src, bytes, hipMemcpyDeviceToHost));
    else                        std::memcpy(dst.data(), src, bytes);
    return MIPT_SUCCESS;
}

void free_accel(Accel* a)
{
    if (!a) return;
    if (a->d_nodes) (void)hipFree(a->d_nodes);
    if (a->d_mat) (void)hipFree(a->d_mat);
    if (a->d_height) (void)hipFree(a->d_height);
    if (a->d_prim_slot) (void)hipFree(a->d_prim_slot);
    if (a->d_cells) (void)hipFree(a->d_cells);
    if (a->d_inst) (void)hipFree(a->d_inst);
    if (a->d_order) (void)hipFree(a->d_order);
    delete a;
}

void free_launch_buffers(MiptContext* c)
{
    for (int i = 0; i < 2; ++i) {
        if (c->q[i].ray_o) (void)hipFree(c->q[i].ray_o);
        if (c->q[i].state_a) (void)hipFree(c->q[i].state_a);
        if (c->q[i].state_b) (void)hipFree(c->q[i].state_b);
        c->q[i] = PathQueue{};
    }
    if (c->sq.ray_o) (void)hipFree(c->sq.ray_o);
    if (c->sq.contrib) (void)hipFree(c->sq.contrib);
    c->sq = ShadowQueue{};
    if (c->hits) (void)hipFree(c->hits);
    if (c->result) (void)hipFree(c->result);
    if (c->rg_seed) (void)hipFree(c->rg_seed);
    c->hits = nullptr; c->result = nullptr; c->rg_seed = nullptr;
    c->cap_paths = 0; c->cap_slots = 0;
}

// SoA queues: ray_o and ray_d are two planes of one allocation
int ensure_capacity(MiptContext* c, size_t n_paths, size_t n_slots)
{
    if (n_paths <= c->cap_paths && n_slots <= c->cap_slots) return MIPT_SUCCESS;
    const size_t np = std::max(n_paths, c->cap_paths), ns = std::max(n_slots, c->cap_slots);
    free_launch_buffers(c);
    for (int i = 0; i < 2; ++i) {
        HIPCHK(c, hipMalloc((void**)&c->q[i].ray_o, np * 2 * sizeof(float4)));
        c->q[i].ray_d = c->q[i].ray_o + np;
        HIPCHK(c, hipMalloc((void**)&c->q[i].state_a, np * sizeof(float4)));
        HIPCHK(c, hipMalloc((void**)&c->q[i].state_b, np * sizeof(uint2)));
    }
    HIPCHK(c, hipMalloc((void**)&c->sq.ray_o, np * 2 * sizeof(float4)));
    c->sq.ray_d = c->sq.ray_o + np;
    HIPCHK(c, hipMalloc((void**)&c->sq.contrib, np * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->hits, np * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->result, ns * sizeof(float4)));
    HIPCHK(c, hipMalloc((void**)&c->rg_seed, ns * sizeof(uint32_t)));
    c->cap_paths = np; c->cap_slots = ns;
    c->log(4, "LAUNCH", "queues sized for %zu paths / %zu slots (%.1f MB)", np, ns,
           (double)(np * (2 * (32 + 16 + 8) + 32 + 16 + 16) + ns * 20) / 1e6);
    return MIPT_SUCCESS;
}

TraceOpts trace_opts(const MiptContext* c)
{
    TraceOpts o{};
    o.persistent = c->persistent;
    o.count_visits = c->count_visits;
    o.blocks = c->num_cus * (c->blocks_per_cu > 0 ? c->blocks_per_cu : 16);
    o.blocks_user = c->blocks_per_cu > 0 ? 1 : 0; o.num_cus = c->num_cus;
    o.refill_min = c->refill_min; o.steal_min = c->steal_min; o.burst_bulk = c->burst_bulk; o.burst_dry = c->burst_dry;
    o.tile_group = c->tile_group;
    o.tail_refill = c->tail_refill;
    o.tiled_bvh4 = c->tiled_bvh4;
    o.inst_form = c->inst_form;
    return o;
}

int default_build_threads()
{
    int threads = (int)std::thread::hardware_concurrency();
    if (const char* e = std::getenv("MIPT_BUILD_THREADS")) { const int v = std::atoi(e); if (v >= 1) threads = v; }     // (tests: the tree does not depend on it)
    return threads < 1 ? 1 : (threads > 32 ? 32 : threads);
}

// Byte layout of the one BVH allocation: nodes (64 B each) | triangles (48 B each) | 64 B pad | node mirror (up to 64 B per node).
// Sets the 32-bit offsets the streaming kernels use (0 when the allocation does not fit them).  Node formats 1 and 2 round the
// boxes to fp16 in place and write the packed 32-byte mirror (pack_nodes32); format 2 then replaces the mirror by the BVH4
// collapse of the rounded tree, also on the device: a build or a refit ends without a host copy of the nodes (the one word the host
// reads is the root box, for the fp16 suitability test).  A scene that does not suit fp16 boxes keeps fp32 nodes.
hipError_t finish_blob(MiptContext* c, Accel* a)
{
    const size_t nb = (size_t)a->num_nodes * sizeof(MiptBvhNode), tb = (size_t)std::max<uint32_t>(a->num_tris, 1u) * sizeof(MiptBvhTri);
    const size_t n32 = (nb + tb + 64 + 127) & ~(size_t)127;                     // a mirror node never straddles a 128-byte line
    const size_t n4 = (n32 + (size_t)a->num_nodes * 32u + 127) & ~(size_t)127;   // the BVH4 mirror behind the packed BVH2 one
    const size_t total = n4 + (size_t)a->num_nodes * 64u;
    a->tri_byte_off = total < (1ull << 31) ? (uint32_t)nb : 0u;
    a->blob_bytes = a->tri_byte_off ? (uint32_t)total : 0u;
    a->node32_off = 0u; a->node4_off = 0u; a->node_fmt = 0u; a->stack_need = 0u; a->num_nodes4 = 0u;
    if (!a->d_height && a->num_nodes) {            // a build (a refit keeps the topology and with it the heights)
        hipError_t e = hipMalloc((void**)&a->d_height, a->num_nodes);
        if (e != hipSuccess) return e;
        const size_t hb = topo_heights_scratch_bytes(a->num_nodes);
        c->scratch_reset();
        if ((e = c->scratch_ensure(hb)) != hipSuccess) return e;
        if ((e = topo_heights(nullptr, a->d_nodes, a->num_nodes, a->d_height, c->scratch_take(hb), hb)) != hipSuccess) return e;
    }
    if (c->node_format != 0 && a->tri_byte_off != 0u && a->num_tris >= 2u) {
        bool ok = false;
        hipError_t e = pack_nodes32(nullptr, a->d_nodes, a->num_nodes, (char*)a->d_nodes + n32, &ok);
        if (e != hipSuccess) return e;
        if (!ok) return hipSuccess;
        a->node32_off = (uint32_t)n32; a->node_fmt = 1u;
#if MIPT_TOP_CACHE
        {
            const size_t top = (n4 + (size_t)a->num_nodes * 64u + 127) & ~(size_t)127;      // behind the BVH4 mirror's room (the allocations carry 4 KB for it)
            launch_build_top(nullptr, (const char*)a->d_nodes + n32, a->num_nodes, (char*)a->d_nodes + top);
            a->top_off = (uint32_t)top;
            if (top + 4096 > (1ull << 31)) a->top_off = 0u;
            else a->blob_bytes = (uint32_t)(top + 4096);
        }
#endif
        // the BVH4 mirror, made on the device from the packed mirror (collapse_bvh4, lbvh_build.hip) for either builder's tree
        if (c->node_format == 2 && a->num_nodes < (1u << 24)) {
            uint32_t cnt4 = 0u, need = 0u;
            const size_t sb = collapse_bvh4_scratch_bytes(a->num_nodes);
            c->scratch_reset();                                            // (the call's earlier temporaries are dead: pack_nodes32 has synchronised)
            if ((e = c->scratch_ensure(sb)) != hipSuccess) return e;
            e = collapse_bvh4(nullptr, (const char*)a->d_nodes + n32, a->d_height, a->num_nodes, a->depth, (char*)a->d_nodes + n4, &cnt4, &need, &ok, c->scratch_take(sb), sb);
            if (e != hipSuccess) return e;
            if (ok) { a->node_fmt = 2u; a->node4_off = (uint32_t)n4; a->stack_need = need; a->num_nodes4 = cnt4; }
        }
    }
    return hipSuccess;
}

Accel* find_accel(const MiptContext* c, unsigned long long h)
{
    auto it = c->accels.find(h);
    return it == c->accels.end() ? nullptr : it->second;
}

// single != null (the context): an instance structure of ONE instance -- one mesh under one node, most glTF files -- is handed to the
// kernels as that mesh's own BVH plus the instance's world -> object matrix (DeviceBvh::xf: every ray is transformed once, when its
// traversal starts).  The two-level walk would test the instance's world box, transform the ray the same way and walk the same mesh;
// the world box is a padded Arvo box made so that it never rejects a ray the object-space walk would let hit, so the hit records are the
// walk's (tests/test_instances.py: against the oracle's two-level walk, bit for bit) -- at the speed of a flattened scene instead of
// 1.2-1.6 x its time.  Not for counting launches (the walk's visit counters include the top level) nor where a kernel form is pinned.
DeviceBvh device_bvh(const Accel* a, const MiptContext* single = nullptr)
{
    if (single != nullptr && a->is_ias && a->inst_fast && a->gas.size() == 1 && !single->count_visits && single->persistent == 3 && single->inst_form == 0) {
        const Accel* g = find_accel(single, a->gas[0]);
        if (g != nullptr && !g->is_ias) {
            DeviceBvh m = device_bvh(g);
            if (m.pool_ok != 0u && m.tri_byte_off != 0u && m.node_fmt == 2u) {
                m.xf = reinterpret_cast<const DeviceInstance*>(reinterpret_cast<const char*>(a->d_nodes) + a->inst_off)[0].inv;
                return m;
            }
        }
    }
    DeviceBvh b{};
    b.nodes = a->d_nodes; b.tris = a->d_tris; b.mat_index = a->d_mat; b.tri_byte_off = a->tri_byte_off; b.blob_bytes = a->blob_bytes; b.depth = a->depth;
    b.pool_ok = (a->num_nodes < (1u << 25) && a->num_tris >= 2u && a->num_tris < (1u << 23) && a->leaf_max <= 4u) ? 1u : 0u;
    b.pool2_ok = (b.pool_ok && a->num_nodes < (1u << 24) && a->num_tris < (1u << 22)) ? 1u : 0u;
    b.node32_off = a->node32_off; b.node_fmt = a->node32_off ? a->node_fmt : 0u;
    b.node4_off = b.node_fmt == 2u ? a->node4_off : 0u; b.stack4 = a->stack_need; b.top_off = b.node_fmt >= 1u ? a->top_off : 0u;
    b.prim_slot = a->d_prim_slot;
    if (a->is_ias) {
        b.inst = a->inst_fast ? reinterpret_cast<const DeviceInstance*>(reinterpret_cast<const char*>(a->d_nodes) + a->inst_off) : a->d_inst;
        b.inst_order = a->d_order; b.n_inst = (uint32_t)a->gas.size(); b.pool_ok = b.pool2_ok = 0u; b.tri_byte_off = 0u; b.node32_off = 0u;
        b.node_fmt = 0u; b.node4_off = 0u; b.blob_bytes = 0u;
        b.inst_fast = a->inst_fast ? 1u : 0u; b.inst_off = a->inst_off; b.inst_node4_off = a->inst_node4_off; b.inst_tri_off = a->inst_tri_off;
        b.inst_blob_bytes = a->inst_blob_bytes; b.stack4 = a->stack_need; b.inst_root4_off = a->inst_root4_off;
        b.prim_slot = nullptr; b.inst_slots = a->inst_slots ? 1u : 0u;
    }
    return b;
}

int check_device_faults(MiptContext* ctx);

int check_launch(MiptContext* ctx, const MiptParams* p, unsigned width, unsigned height, unsigned depth, Accel** acc)
{
    if (!p) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "params is null");
    if (depth != 1) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "launch depth must be 1 (got %u)", depth);
    if (width == 0 || height == 0 || width != p->width || height != p->height)
        return ctx->fail(MIPT_ERROR_INVALID_VALUE, "launch size %ux%u does not match params %ux%u", width, height, p->width, p->height);
    if ((uint64_t)width * height > (1ull << 30)) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "launch too large");
    if (p->samples_per_launch == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "samples_per_launch must be >= 1");
    *acc = find_accel(ctx, p->handle);
    if (!*acc) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "params.handle 0x%llx is not a traversable of this context", p->handle);
    if ((*acc)->stale) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instance structure 0x%llx references a mesh that was refitted since it was built: call mipt_accel_update_instances first", p->handle);
    if (!ctx->d_hitgroups || ctx->n_hitgroups == 0) return ctx->fail(MIPT_ERROR_NOT_CONFIGURED, "no hit-group records set (mipt_sbt_set_hitgroups)");
    // a fault an EARLIER launch raised (callers that synchronise the stream themselves never pass mipt_sync): reported once, here
    return check_device_faults(ctx);
}

FrameConsts make_consts(const MiptContext* c, const MiptParams* p)
{
    FrameConsts fc{};
    fc.width = p->width; fc.height = p->height; fc.subframe_index = p->subframe_index;
    fc.samples_per_launch = p->samples_per_launch;
    std::memcpy(fc.eye, p->eye, 12); std::memcpy(fc.U, p->U, 12); std::memcpy(fc.V, p->V, 12); std::memcpy(fc.W, p->W, 12);
    fc.light = p->light;
    std::memcpy(fc.bg, c->miss.bg_color, 16);
    fc.max_depth = c->max_depth;
    return fc;
}

// the wavefront loop shared by full-frame and shard launches
// pixels of the frame that a rank owns under the 8x4 interleaved tile map
uint32_t shard_valid_pixels(uint32_t w, uint32_t h, uint32_t world, uint32_t rank)
{
    const uint32_t ns = shard_num_samples(w, h, world);
    uint32_t valid = 0;
    for (uint32_t tile = 0; tile < ns / 32u; ++tile) {
        int px, py;
        shard_pixel(w, world, rank, tile * 32u, px, py);
        if ((uint32_t)px >= w || (uint32_t)py >= h) continue;
        const uint32_t cols = w - (uint32_t)px < 8u ? w - (uint32_t)px : 8u;
        const uint32_t rows = h - (uint32_t)py < 4u ? h - (uint32_t)py : 4u;
        valid += cols * rows;
    }
    return valid;
}

uint32_t next_epoch(MiptContext* ctx)
{
    ctx->tail_epoch = (ctx->tail_epoch + 1u) & 0x7fffffffu;
    if (ctx->tail_epoch == 0u) ctx->tail_epoch = 1u;
    return ctx->tail_epoch | (ctx->debug_epoch_fault ? 0x80000000u : 0u);
}

// A kernel that gave up on a bounded wait has said so in the pinned fault word: after a synchronisation that is an error, once.
int check_device_faults(MiptContext* ctx)
{
    uint32_t* fw = ctx->h_pinned + kFaultWord;
    const uint32_t f = __atomic_exchange_n(fw, 0u, __ATOMIC_RELAXED);
    if (f & kFaultEpochTimeout)
        return ctx->fail(MIPT_ERROR_INTERNAL, "a path-kernel wave timed out waiting for the launch's first workgroup (epoch handshake): "
                                              "the ray counters of that launch may miss additions; pixels are not affected");
    if (f) return ctx->fail(MIPT_ERROR_INTERNAL, "device fault word 0x%x", f);
    return MIPT_SUCCESS;
}

int run_wavefront(MiptContext* ctx, hipStream_t st, const FrameConsts& fc_in, const Accel* acc, size_t n_slots,
                  float4* accum_out, uint32_t* frame_out)
{
    FrameConsts fc = fc_in;
    const uint32_t n_valid = fc.world == 0u ? fc.width * fc.height : shard_valid_pixels(fc.width, fc.height, fc.world, fc.rank);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // batched samples: every sample is its own path with its own accumulator, one wavefront pass in all
    const bool batch = ctx->batch_samples && fc.samples_per_launch > 1u &&
                       (uint64_t)fc.n_paths * fc.samples_per_launch < (1ull << 31);
    fc.paths_per_sample = fc.n_paths;
    fc.n_slots = (uint32_t)n_slots;
    fc.batch_spp = batch ? fc.samples_per_launch : 0u;
    const uint32_t passes = batch ? 1u : fc.samples_per_launch;
    const uint32_t pass_valid = batch ? n_valid * fc.samples_per_launch : n_valid;
    // one sample per pixel over a frame or a rank's shard of it: whichever launch ends a path writes its pixel, and k_resolve (13 us + a launch gap)
    // is not launched (ResolveJob; env MIPT_DIRECT_RESOLVE=0 for the A/B)
    ResolveJob rj{};
    rj.accum = accum_out; rj.frame = frame_out; rj.subframe_index = fc.subframe_index;
    rj.on = (ctx->direct_resolve && passes == 1u && !batch && !ctx->count_visits) ? 1u : 0u;       // (a rank's shard too: slot = its compact index)
    if (batch) fc.n_paths *= fc.samples_per_launch;
    const uint32_t n_paths = fc.n_paths;
    int rc = ensure_capacity(ctx, n_paths, batch ? n_slots * fc.samples_per_launch : n_slots);
    if (rc) return rc;
    ctx->last_stream = st;
    const DeviceBvh bvh = device_bvh(acc, ctx);
    const TraceOpts opts = trace_opts(ctx);
    bool advance_in_resolve = false;
    for (uint32_t s = 0; s < passes; ++s) {
        // MIPT_OPT_TAIL_PATHS is meant for the pool form of the tail kernel; the plain loop (BVHs that do not fit the pool's
        // packed entries, other kernel variants) pays off at a quarter of it, and not at all for depth-capped passes from
        // the start.  Depth-capped paths have few bounces left, so a switch inside the loop pays later: at half the count
        // (1 M-triangle scene, 1080p, depth 8: limit 2^18 / 2^19 / 2^20 -> 3.79 / 3.61 / 4.28 ms).
        // (the path kernels walk one-level BVHs, or -- k_tail_pool_inst -- a two-level structure in its entry-pool form)
        const bool inst_tail = acc->is_ias && tail_launch_makes_rays(bvh, opts);
        const bool use_tail = ctx->tail_paths != 0u && !ctx->count_visits && (!acc->is_ias || inst_tail);
        const bool pool_tail = inst_tail || (opts.persistent == 3 && bvh.pool_ok != 0u && bvh.tri_byte_off != 0u && bvh.depth <= 32u);
        const uint32_t tail_paths = !use_tail ? 0u : (pool_tail ? ctx->tail_paths : ctx->tail_paths / 4u);
        const uint32_t tail_limit = fc.max_depth != 0u ? tail_paths / 2u : tail_paths;
        // (a capped pass of a million paths is as fast in the wavefront kernels: half the headline frame 0.54 vs 0.56 ms)
        const uint32_t first_limit = fc.max_depth == 0u ? tail_paths : (pool_tail ? tail_paths / 4u * 3u : 0u);
        // raygen's first workgroup resets the queue counters (and, for the first pass, the launch statistics)
        // the primary rays of a frame (or of a rank's shard of it) walk their tiles in the order of the previous launch's costs (MIPT_OPT_TILE_ORDER) when this
        // pass runs the tiled kernel (launch_trace's choice: the pool form over a one-level BVH that fits it, more than 2560 rays
        // per CU, not a counting launch): the raygen launch clears the class counters this pass fills
        // (ONE predicate decides the kernel form: closest_launch_is_tiled() -- launch_trace consults the same function)
        const bool tiled_primary = ctx->tile_order_mode != 0 && !batch && closest_launch_is_tiled(bvh, n_paths, opts) && n_paths > first_limit;
        TileOrderJob job{};
        if (tiled_primary) {
            const uint32_t tiles = n_paths / 64u, chunks = (tiles + kOrderChunk - 1u) / kOrderChunk;
            if (ctx->tile_n != tiles) {
                HIPCHK(ctx, hipDeviceSynchronize());
                for (void* p : { (void*)ctx->d_tile_class, (void*)ctx->d_tile_hist, (void*)ctx->d_tile_order }) if (p) (void)hipFree(p);
                ctx->d_tile_class = nullptr; ctx->d_tile_hist = ctx->d_tile_order = nullptr; ctx->tile_n = 0; ctx->tile_order_valid = false;
                HIPCHK(ctx, hipMalloc((void**)&ctx->d_tile_class, tiles));
                HIPCHK(ctx, hipMalloc((void**)&ctx->d_tile_hist, (size_t)chunks * kOrderClasses * 8u));
                HIPCHK(ctx, hipMemset(ctx->d_tile_hist, 0, (size_t)chunks * kOrderClasses * 8u));
                ctx->tile_hist_set = 0;
                HIPCHK(ctx, hipMalloc((void**)&ctx->d_tile_order, (size_t)tiles * 4u));
                if (!ctx->d_tile_width) {
                    HIPCHK(ctx, hipMalloc((void**)&ctx->d_tile_width, 4u));
                    HIPCHK(ctx, hipMemcpy(ctx->d_tile_width, &ctx->tile_class_width, 4u, hipMemcpyHostToDevice));
                }
                ctx->tile_n = tiles;
            }
            if (ctx->tile_map != ((fc.world << 16) | fc.rank)) { ctx->tile_map = (fc.world << 16) | fc.rank; ctx->tile_order_valid = false; }
            job.cls = ctx->d_tile_class; job.order = ctx->d_tile_order; job.n_tiles = tiles; job.n_chunks = chunks;
            job.hist = ctx->d_tile_hist + (size_t)ctx->tile_hist_set * chunks * kOrderClasses;
            job.zero = ctx->d_tile_hist + (size_t)(ctx->tile_hist_set ^ 1u) * chunks * kOrderClasses;
            ctx->tile_hist_set ^= 1u;
            job.width = ctx->tile_class_fixed ? nullptr : ctx->d_tile_width;
        } else {
            ctx->tile_order_valid = false;
        }
        // the pass's primary rays: made by the closest-hit launch itself where that takes the tiled form (RaygenJob), else by k_raygen
        RaygenJob rg{};
        rg.fc = fc; rg.sample = s; rg.q = ctx->q[0]; rg.rg_seed = ctx->rg_seed; rg.result = ctx->result; rg.ctr = ctx->d_ctr;
        rg.n_valid = pass_valid; rg.zero_stats = s == 0u ? 1u : 0u;
        const bool tail_first = n_paths <= first_limit;                   // the pass runs in the path kernel from its first bounce
        const bool fused_raygen = ctx->fuse_raygen && (tail_first ? tail_launch_makes_rays(bvh, opts) : closest_launch_is_tiled(bvh, n_paths, opts));
        if (!fused_raygen) { TimedScope t(ctx, st, T_RAYGEN); launch_raygen(st, fc, s, ctx->q[0], ctx->rg_seed, ctx->result, ctx->d_ctr, pass_valid, s == 0u); }
        int cur = 0;
        if (n_paths <= first_limit) {
            // a pass this small runs in the tail kernel from the first bounce on (Cornell 512x512: 0.73 -> 0.37 ms)
            TimedScope t(ctx, st, T_CLOSEST);
            launch_tail(st, fc, bvh, ctx->d_hitgroups, ctx->n_hitgroups, ctx->q[0], ctx->result, ctx->d_ctr, n_paths, ctx->common_vertices, opts, rj,
                        fused_raygen ? &rg : nullptr, next_epoch(ctx));
            continue;
        }
        if (ctx->frame1 && fused_raygen && !batch && !ctx->count_visits && fc.max_depth == 1u) {
            // a one-bounce pass as ONE launch of tile waves (k_frame1): trace, shade, occlusion rays, pixels -- no shade / any-hit launch
            TraceOpts o2 = opts;
            if (tiled_primary) {
                o2.tile_cost = ctx->d_tile_class; o2.tile_hist = const_cast<uint32_t*>(job.hist); o2.tile_order = ctx->tile_order_valid ? ctx->d_tile_order : nullptr;
                o2.n_order = job.n_tiles; o2.class_width = ctx->tile_class_width; o2.class_width_dev = job.width;
                if (!ctx->tile_group_set && o2.tile_order != nullptr) o2.tile_group = 96;
                if (ctx->tile_order_mode == 2 && ctx->tile_order_valid) { o2.tile_cost = nullptr; job = TileOrderJob{}; }
            }
            bool done1 = false;
            { TimedScope t(ctx, st, T_CLOSEST);
              done1 = launch_frame1(st, fc, bvh, ctx->d_hitgroups, ctx->n_hitgroups, ctx->result, ctx->d_ctr, n_paths, ctx->common_vertices, o2, rj, rg,
                                    next_epoch(ctx), tiled_primary ? job : TileOrderJob{}, ctx->frame1 == 2 ? &ctx->sq : nullptr); }
            if (done1) {
                if (tiled_primary && job.n_chunks != 0u) ctx->tile_order_valid = true;
                if (ctx->frame1 == 2) {
                    // MIPT_FRAME1 = 2: the tile waves traced and shaded; the occlusion rays they queued are the wavefront form's any-hit launch
                    { TimedScope t(ctx, st, T_ANY);
                      launch_trace_any(st, bvh, ctx->sq.ray_o, ctx->sq.ray_d, 1, &ctx->d_ctr->n_shadow, n_paths, nullptr, ctx->sq.contrib,
                                       ctx->result, ctx->d_ctr, opts, rj); }
                    if (s + 1u == passes) advance_in_resolve = true;
                    else launch_advance(st, ctx->d_ctr, ctx->h_pinned);
                }
                continue;
            }
        }
        uint32_t bound = n_paths;      // upper bound of the live-path count (it only falls): the latest count the host has seen
        for (uint32_t bounce = 0;; ++bounce) {
            { TimedScope t(ctx, st, T_CLOSEST);
              TraceOpts o2 = opts;
              if (bounce == 0u && tiled_primary) {
                  // this pass walks the order the previous pass's shade launch made and records its own cost classes for the next
                  // (MIPT_TILE_ORDER=2, a diagnostic: the first order is kept)
                  o2.tile_cost = ctx->d_tile_class; o2.tile_hist = const_cast<uint32_t*>(job.hist); o2.tile_order = ctx->tile_order_valid ? ctx->d_tile_order : nullptr;
                  o2.n_order = job.n_tiles; o2.class_width = ctx->tile_class_width; o2.class_width_dev = job.width;
                  // in cost order a group of consecutive slots is a group of equally heavy tiles: smaller groups spread them over the
                  // XCDs better (32 / 48 / 64 / 96 / 128 / 256 / 512: closest-hit kernel 0.335 / 0.333 / 0.339 / 0.334 / 0.342 / 0.345 / 0.363 ms;
                  // at 32 the any-hit kernel that follows loses 25 us -- it inherits the L2 contents)
                  if (!ctx->tile_group_set && o2.tile_order != nullptr) o2.tile_group = 96;
                  if (ctx->tile_order_mode == 2 && ctx->tile_order_valid) { o2.tile_cost = nullptr; job = TileOrderJob{}; }
              }
#if MIPT_PROBE_WAVES
              if (bounce == 0u) {     // probe builds only, as in mipt_trace_closest (MIPT_PROBE_WAVES_ANY: the any-hit kernel's records)
                  if (const char* e = std::getenv("MIPT_PROBE_WAVES")) {
                      const unsigned long long addr = std::strtoull(e, nullptr, 0);
                      HIPCHK(ctx, hipMemcpyAsync(&ctx->d_ctr->dbg[0], &addr, 8, hipMemcpyHostToDevice, st));
                  }
                  if (const char* e = std::getenv("MIPT_PROBE_WAVES_ANY")) {
                      const unsigned long long addr = std::strtoull(e, nullptr, 0);
                      HIPCHK(ctx, hipMemcpyAsync(&ctx->d_ctr->dbg[1], &addr, 8, hipMemcpyHostToDevice, st));
                  }
              }
#endif
              launch_trace_closest(st, bvh, ctx->q[cur].ray_o, ctx->q[cur].ray_d, 1, &ctx->d_ctr->n_cur, bound, ctx->hits, ctx->d_ctr, o2, 0u,
                                   bounce == 0u && fused_raygen ? &rg : nullptr); }
            { TimedScope t(ctx, st, T_SHADE);
              launch_shade(st, fc, bvh, ctx->d_hitgroups, ctx->n_hitgroups, ctx->q[cur], ctx->hits, ctx->q[cur ^ 1], ctx->sq,
                           ctx->result, ctx->d_ctr, bound, ctx->common_vertices, bounce == 0u ? job : TileOrderJob{},
                           bounce == 0u && fused_raygen ? (1u | ((s == 0u || batch) ? 2u : 0u)) : 0u, rj); }
            if (bounce == 0u && job.n_chunks != 0u) ctx->tile_order_valid = true;
            { TimedScope t(ctx, st, T_ANY);
              launch_trace_any(st, bvh, ctx->sq.ray_o, ctx->sq.ray_d, 1, &ctx->d_ctr->n_shadow, bound, nullptr, ctx->sq.contrib,
                               ctx->result, ctx->d_ctr, opts, rj); }
            if (fc.max_depth != 0u && bounce + 1u >= fc.max_depth && s + 1u == passes) { advance_in_resolve = true; break; }   // k_resolve does it
            launch_advance(st, ctx->d_ctr, ctx->h_pinned + bounce % kRing);
            cur ^= 1;
            if (fc.max_depth != 0u) {
                if (bounce + 1u >= fc.max_depth) break;
                if (!use_tail) continue;      // bounded paths, no tail kernel: no host round trip, empty queues make the kernels exit at once
            }
            // Unbounded (reference behaviour): stop when Russian roulette has ended every path.  The live count after
            // bounce b is written by k_advance into a pinned ring and read kLag bounces later, so the stream never
            // drains while the host waits; the bounces launched past the end find empty queues and exit at once.
            HIPCHK(ctx, hipEventRecord(ctx->ring_ev[bounce % kRing], st));
            if (bounce >= kLag) {
                const uint32_t old = (bounce - kLag) % kRing;
                HIPCHK(ctx, hipEventSynchronize(ctx->ring_ev[old]));
                const uint32_t live = ctx->h_pinned[old];       // an upper bound of every later count
                if (live == 0u) break;
                bound = live < bound ? live : bound;            // sizes the grids and picks the kernel forms of the next bounces
                if (live <= tail_limit) {
                    // few paths left: one launch runs them to their ends instead of five per bounce
                    TimedScope t(ctx, st, T_CLOSEST);
                    launch_tail(st, fc, bvh, ctx->d_hitgroups, ctx->n_hitgroups, ctx->q[cur], ctx->result, ctx->d_ctr, live,
                                ctx->common_vertices, opts, rj);
                    break;
                }
            }
            if (bounce > 100000u) return ctx->fail(MIPT_ERROR_INTERNAL, "path length runaway");
        }
    }
    // (with ResolveJob the pixels are written already; the last bounce's queue bookkeeping, which rides on k_resolve, only feeds the
    //  statistics then and is applied to their copy when they are read: advance_pending)
    ctx->advance_pending = rj.on != 0u && advance_in_resolve;
    ctx->bins_live = true;
    if (!rj.on) { TimedScope t(ctx, st, T_RESOLVE); launch_resolve(st, fc, ctx->result, accum_out, frame_out, (uint32_t)n_slots, advance_in_resolve ? ctx->d_ctr : nullptr); }
    ctx->bounces_on_device = true;
    ++ctx->profile_launch;
    HIPCHK(ctx, hipGetLastError());
    return MIPT_SUCCESS;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

int mipt_abi_version(void) { return MIPT_ABI_VERSION; }

const char* mipt_error_name(int r)
{
    switch (r) {
    case MIPT_SUCCESS: return "MIPT_SUCCESS";
    case MIPT_ERROR_INVALID_VALUE: return "MIPT_ERROR_INVALID_VALUE";
    case MIPT_ERROR_OUT_OF_MEMORY: return "MIPT_ERROR_OUT_OF_MEMORY";
    case MIPT_ERROR_HIP: return "MIPT_ERROR_HIP";
    case MIPT_ERROR_NO_DEVICE: return "MIPT_ERROR_NO_DEVICE";
    case MIPT_ERROR_INVALID_HANDLE: return "MIPT_ERROR_INVALID_HANDLE";
    case MIPT_ERROR_NOT_CONFIGURED: return "MIPT_ERROR_NOT_CONFIGURED";
    case MIPT_ERROR_BVH_TOO_DEEP: return "MIPT_ERROR_BVH_TOO_DEEP";
    case MIPT_ERROR_INTERNAL: return "MIPT_ERROR_INTERNAL";
    default: return "MIPT_ERROR_UNKNOWN";
    }
}

int mipt_context_create(int hip_device, MiptLogCallback cb, void* cbdata, int log_level, MiptContext** out)
{
    if (!out) return MIPT_ERROR_INVALID_VALUE;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return MIPT_ERROR_NO_DEVICE; }
    if (hip_device < 0 || hip_device >= n) return MIPT_ERROR_INVALID_VALUE;
    if (hipSetDevice(hip_device) != hipSuccess) return MIPT_ERROR_HIP;
    if (hipFree(nullptr) != hipSuccess) return MIPT_ERROR_HIP;     // same idiom as cudaFree(0), optixPathTracer.cpp:558
    MiptContext* c = new MiptContext();
    c->device = hip_device; c->cb = cb; c->cbdata = cbdata; c->log_level = log_level;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, hip_device) == hipSuccess) {
        c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        c->log(4, "CONTEXT", "device %d: %s, %d CUs, arch %s", hip_device, prop.name, c->num_cus, prop.gcnArchName);
    }
    if (hipMalloc((void**)&c->d_ctr, sizeof(DeviceCounters)) != hipSuccess ||
        hipMemset(c->d_ctr, 0, sizeof(DeviceCounters)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_pinned, 64, hipHostMallocCoherent) != hipSuccess) {
        delete c;
        return MIPT_ERROR_OUT_OF_MEMORY;
    }
    std::memset(c->h_pinned, 0, 64);
    {   // kernels report a bounded wait that ran out through this pinned word (DeviceCounters::fault_host)
        uint32_t* fw = c->h_pinned + kFaultWord;
        if (hipMemcpy(&c->d_ctr->fault_host, &fw, sizeof fw, hipMemcpyHostToDevice) != hipSuccess) { delete c; return MIPT_ERROR_HIP; }
    }
    for (hipEvent_t& e : c->ring_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) { delete c; return MIPT_ERROR_HIP; }
    if (const char* e = std::getenv("MIPT_DEBUG_EPOCH_FAULT")) c->debug_epoch_fault = std::atoi(e) != 0;      // fault injection for the test of the report path
    // experiment hooks (tools/, CI matrix): initial values of MIPT_OPT_PERSISTENT / MIPT_OPT_STEAL_MIN
    if (const char* e = std::getenv("MIPT_PERSISTENT")) { const int v = std::atoi(e); if (v == 0 || v == 2 || v == 3) c->persistent = v; }
    if (const char* e = std::getenv("MIPT_NODE_FORMAT")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) c->node_format = v; }
    if (const char* e = std::getenv("MIPT_TILE_CLASS")) { const int v = std::atoi(e); if (v >= 1 && v <= 4096) { c->tile_class_width = (uint32_t)v; c->tile_class_fixed = true; } }
    if (const char* e = std::getenv("MIPT_TILE_CLASS_INIT")) { const int v = std::atoi(e); if (v >= 1 && v <= 4096) c->tile_class_width = (uint32_t)v; }   // adapted from there
    if (const char* e = std::getenv("MIPT_TILED_BVH4")) c->tiled_bvh4 = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_INST_FORM")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) c->inst_form = v; }
    if (const char* e = std::getenv("MIPT_BUILD")) { const int v = std::atoi(e); if (v >= 0 && v <= 4) c->build_mode = v; }
    if (const char* e = std::getenv("MIPT_SPLIT_BELOW")) { const int v = std::atoi(e); if (v >= 64 && v <= 2048) c->split_below = v; }
    if (const char* e = std::getenv("MIPT_SBVH_EXTRA")) { const double v = std::atof(e); if (v >= 0.0 && v <= 8.0) c->sbvh_extra = (float)v; }       // MIPT_OPT_BUILD at creation
    if (const char* e = std::getenv("MIPT_BUILD_HOST_BELOW")) { const long v = std::atol(e); if (v >= 0 && v <= (1l << 30)) c->build_host_below = (uint32_t)v; }
    if (const char* e = std::getenv("MIPT_LBVH_TREELETS")) c->lbvh_treelets = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_LBVH_LEAF")) { const int v = std::atoi(e); if (v >= 1 && v <= 4) c->lbvh_leaf = v; }
    if (const char* e = std::getenv("MIPT_TAIL_REFILL")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) c->tail_refill = v; }
    if (const char* e = std::getenv("MIPT_DIRECT_RESOLVE")) c->direct_resolve = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_FRAME1")) { const int v = std::atoi(e); c->frame1 = v == 1 || v == 2 ? v : 0; }
    if (const char* e = std::getenv("MIPT_FUSE_RAYGEN")) c->fuse_raygen = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_TILE_ORDER")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) c->tile_order_mode = v; }
    if (const char* e = std::getenv("MIPT_TILE_SHIFT")) { const int v = std::atoi(e); if (v >= 0 && v <= 6) c->tile_shift = v; }
    if (const char* e = std::getenv("MIPT_TILE_GROUP")) { const int v = std::atoi(e); if (v >= 0 && v <= (1 << 24)) { c->tile_group = v; c->tile_group_set = true; } }
    if (const char* e = std::getenv("MIPT_TAIL_PATHS")) { const int v = std::atoi(e); if (v >= 0 && v <= (1 << 30)) c->tail_paths = (uint32_t)v; }
    if (const char* e = std::getenv("MIPT_STEAL_MIN")) { const int v = std::atoi(e); if (v >= 1 && v <= 65) c->steal_min = v; }
    *out = c;
    return MIPT_SUCCESS;
}

int mipt_context_destroy(MiptContext* c)
{
    if (!c) return MIPT_ERROR_INVALID_VALUE;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : c->accels) free_accel(kv.second);
    free_launch_buffers(c);
    if (c->d_hitgroups) (void)hipFree(c->d_hitgroups);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->scratch) (void)hipFree(c->scratch);
    for (void* p : { (void*)c->d_tile_class, (void*)c->d_tile_hist, (void*)c->d_tile_order, (void*)c->d_tile_width }) if (p) (void)hipFree(p);
    if (c->im_sq.ray_o) (void)hipFree(c->im_sq.ray_o);
    if (c->im_sq.contrib) (void)hipFree(c->im_sq.contrib);
    if (c->im_flags) (void)hipFree(c->im_flags);
    if (c->d_wh_materials) (void)hipFree(c->d_wh_materials);
    if (c->d_wh_textures) (void)hipFree(c->d_wh_textures);
    for (void* p : { (void*)c->d_wh_inst, (void*)c->d_wh_meshes, (void*)c->d_wh_prim_inst }) if (p) (void)hipFree(p);
    if (c->wh_e1) (void)hipFree(c->wh_e1);
    if (c->wh_group) (void)hipFree(c->wh_group);
    if (c->wh_levels) (void)hipFree(c->wh_levels);
    for (void* t : c->wh_texels) if (t) (void)hipFree(t);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ring_ev) if (e) (void)hipEventDestroy(e);
    delete c;
    return MIPT_SUCCESS;
}

const char* mipt_last_error_string(const MiptContext* c) { return c ? c->last_error.c_str() : "null context"; }

int mipt_set_option(MiptContext* c, int option, uint64_t value)
{
    if (!c) return MIPT_ERROR_INVALID_VALUE;
    switch (option) {
    case MIPT_OPT_MAX_DEPTH: c->max_depth = (uint32_t)value; return MIPT_SUCCESS;
    case MIPT_OPT_COUNT_VISITS: c->count_visits = value ? 1 : 0; return MIPT_SUCCESS;
    case MIPT_OPT_TRACE_GRID:
        if (value > 256) return c->fail(MIPT_ERROR_INVALID_VALUE, "at most 256 single-wave workgroups per CU");
        c->blocks_per_cu = (int)value; return MIPT_SUCCESS;
    case MIPT_OPT_PERSISTENT:
        if (value != 0 && value != 2 && value != 3) return c->fail(MIPT_ERROR_INVALID_VALUE, "traversal kernel variant must be 0, 2 or 3");
        c->persistent = (int)value; return MIPT_SUCCESS;
    case MIPT_OPT_REFILL_MIN:
        if (value < 1 || value > 64) return c->fail(MIPT_ERROR_INVALID_VALUE, "refill threshold must be 1..64");
        c->refill_min = (int)value; return MIPT_SUCCESS;
    case MIPT_OPT_BATCH_SAMPLES: c->batch_samples = value ? 1 : 0; return MIPT_SUCCESS;
    case MIPT_OPT_TAIL_PATHS:
        if (value > (1u << 30)) return c->fail(MIPT_ERROR_INVALID_VALUE, "tail threshold must be 0 (off) .. 2^30 paths");
        c->tail_paths = (uint32_t)value; return MIPT_SUCCESS;
    case MIPT_OPT_BUILD:
        if (value > 4) return c->fail(MIPT_ERROR_INVALID_VALUE, "builder must be 0 (host SAH), 1 (device LBVH), 2 (device SAH), 3 (host SAH with spatial splits) or 4 (device SAH with spatial splits)");
        c->build_mode = (int)value; return MIPT_SUCCESS;
    case MIPT_OPT_TILE_ORDER:
        if (value > 1) return c->fail(MIPT_ERROR_INVALID_VALUE, "tile order must be 0 (image order) or 1 (by the previous launch's costs)");
        c->tile_order_mode = (int)value; c->tile_order_valid = false; return MIPT_SUCCESS;
    case MIPT_OPT_NODE_FORMAT:
        if (value > 2) return c->fail(MIPT_ERROR_INVALID_VALUE, "node format must be 0 (fp32 boxes), 1 (fp16 boxes, packed 32-byte BVH2 nodes) or 2 (fp16 boxes, BVH4 nodes)");
        c->node_format = (int)value; return MIPT_SUCCESS;
    case MIPT_OPT_BURST:
        if ((value & 0xffu) < 1 || (value & 0xffu) > 64 || ((value >> 8) & 0xffu) < 1 || ((value >> 8) & 0xffu) > 64)
            return c->fail(MIPT_ERROR_INVALID_VALUE, "burst lengths must be 1..64 (bulk | dry << 8)");
        c->burst_bulk = (int)(value & 0xffu); c->burst_dry = (int)((value >> 8) & 0xffu); return MIPT_SUCCESS;
    case MIPT_OPT_STEAL_MIN:
        if (value < 1 || value > 65) return c->fail(MIPT_ERROR_INVALID_VALUE, "steal threshold must be 1..65 (65 = never)");
        c->steal_min = (int)value; return MIPT_SUCCESS;
    case MIPT_OPT_PROFILE:
        if (c->last_stream || c->timed.size()) (void)hipStreamSynchronize(c->last_stream);
        c->profile = value == 2 || value == 3 ? (int)value : (value ? 1 : 0); c->timed.clear(); c->events_used = 0; c->profile_launch = 0; return MIPT_SUCCESS;
    default: return c->fail(MIPT_ERROR_INVALID_VALUE, "unknown option %d", option);
    }
}

int mipt_get_option(const MiptContext* c, int option, uint64_t* value)
{
    if (!c || !value) return MIPT_ERROR_INVALID_VALUE;
    switch (option) {
    case MIPT_OPT_MAX_DEPTH: *value = c->max_depth; return MIPT_SUCCESS;
    case MIPT_OPT_COUNT_VISITS: *value = (uint64_t)c->count_visits; return MIPT_SUCCESS;
    case MIPT_OPT_TRACE_GRID: *value = (uint64_t)c->blocks_per_cu; return MIPT_SUCCESS;
    case MIPT_OPT_PERSISTENT: *value = (uint64_t)c->persistent; return MIPT_SUCCESS;
    case MIPT_OPT_PROFILE: *value = (uint64_t)c->profile; return MIPT_SUCCESS;
    case MIPT_OPT_REFILL_MIN: *value = (uint64_t)c->refill_min; return MIPT_SUCCESS;
    case MIPT_OPT_STEAL_MIN: *value = (uint64_t)c->steal_min; return MIPT_SUCCESS;
    case MIPT_OPT_BATCH_SAMPLES: *value = (uint64_t)c->batch_samples; return MIPT_SUCCESS;
    case MIPT_OPT_BUILD: *value = (uint64_t)c->build_mode; return MIPT_SUCCESS;
    case MIPT_OPT_NODE_FORMAT: *value = (uint64_t)c->node_format; return MIPT_SUCCESS;
    case MIPT_OPT_TILE_ORDER: *value = (uint64_t)c->tile_order_mode; return MIPT_SUCCESS;
    case MIPT_OPT_TAIL_PATHS: *value = (uint64_t)c->tail_paths; return MIPT_SUCCESS;
    case MIPT_OPT_BURST: *value = (uint64_t)c->burst_bulk | ((uint64_t)c->burst_dry << 8); return MIPT_SUCCESS;
    default: return MIPT_ERROR_INVALID_VALUE;
    }
}

// ------------------------------------------------------------------------------------------------
// MIPT_OPT_BUILD = 1: the whole build on the device -- inputs that live in host memory are uploaded as they are, a
// gather kernel makes world-space triangles / SBT indices / geometry flags, lbvh_build.hip builds the tree.
// Returns MIPT_SUCCESS with *handle_out = 0 when the caller should use the host builder instead (tiny scenes, a
// tree deeper than the traversal stack).
// world-space triangles (9 floats each), SBT index and geometry flags per triangle of the build inputs, assembled on the
// device (k_gather_input); host arrays are uploaded as they are.  *n_out = 0: nothing to do on the device (tiny or huge input).
struct DeviceGather {
    std::vector<void*> temps;          // allocations of this call (what did not fit the context's arena)
    float* world = nullptr; uint32_t* mats = nullptr; uint32_t* gflags = nullptr;
    bool mats_owned = false;           // mats is an allocation of its own (a build keeps it as the accel's SBT index table)
    uint32_t n = 0, err = 0;
    hipError_t e = hipSuccess;
    ~DeviceGather() { for (void* p : temps) if (p) (void)hipFree(p); if (mats && mats_owned) (void)hipFree(mats); }
    uint32_t* release_mats() { uint32_t* m = mats; mats = nullptr; return m; }
};

// keep_mats: the SBT index table becomes an allocation of its own (builds keep it); extra_scratch: bytes the caller will take from the
// arena after this (sized in one go, so that the arena is never re-allocated while parts of it are in use)
static int gather_on_device(MiptContext* ctx, const MiptBuildInput* inputs, uint32_t num_inputs, DeviceGather& G, bool keep_mats, size_t extra_scratch)
{
    std::vector<uint32_t> counts(num_inputs);
    uint64_t total = 0;
    for (uint32_t k = 0; k < num_inputs; ++k) {
        const MiptBuildInput& in = inputs[k];
        const uint32_t stride = in.vertex_stride_bytes ? in.vertex_stride_bytes : 12u;
        if (stride < 12u || (stride & 3u)) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: vertex stride %u", k, stride);
        if (in.num_vertices && !in.vertices) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: null vertices", k);
        uint32_t n_tris = in.num_triangles;
        if (!in.indices || in.index_format == 0) {
            if (n_tris == 0) n_tris = in.num_vertices / 3u;
            if ((uint64_t)n_tris * 3u > in.num_vertices) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: %u triangles need %u vertices, got %u", k, n_tris, n_tris * 3u, in.num_vertices);
        } else if (in.index_format != 2u && in.index_format != 4u) {
            return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: index format %u (expected 0, 2 or 4)", k, in.index_format);
        }
        if (in.flags && in.num_sbt_records == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: flags without num_sbt_records", k);
        counts[k] = n_tris; total += n_tris;
    }
    if (total <= 4u || total * 36u >= (1ull << 31)) return MIPT_SUCCESS;
    const uint32_t n = (uint32_t)total;
    ctx->scratch_reset();
    {
        const hipError_t se = ctx->scratch_ensure((size_t)n * (36u + 4u + 4u) + 4096u + extra_scratch);
        if (se != hipSuccess) return ctx->fail(MIPT_ERROR_OUT_OF_MEMORY, "build scratch: %s", hipGetErrorString(se));
    }
    auto dev_alloc = [&](size_t bytes, void** out) {                    // the arena first, an allocation of the call's own when it is full
        if (void* p = ctx->scratch_take(bytes ? bytes : 4)) { *out = p; return hipSuccess; }
        hipError_t e = hipMalloc(out, bytes ? bytes : 4); if (e == hipSuccess) G.temps.push_back(*out); return e;
    };
    auto on_device = [&](const void* p, size_t bytes, const void** out) {          // device copy of a host array, or the pointer itself
        if (!p || bytes == 0) { *out = p; return hipSuccess; }
        if (is_device_pointer(p)) { *out = p; return hipSuccess; }
        void* d = nullptr;
        hipError_t e = dev_alloc(bytes, &d);
        if (e == hipSuccess) e = hipMemcpy(d, p, bytes, hipMemcpyHostToDevice);
        *out = d;
        return e;
    };
    uint32_t* d_err = nullptr;
    hipError_t e = dev_alloc((size_t)n * 36u, (void**)&G.world);
    if (e == hipSuccess) {
        if (keep_mats) { e = hipMalloc((void**)&G.mats, (size_t)n * 4u); G.mats_owned = e == hipSuccess; }
        else e = dev_alloc((size_t)n * 4u, (void**)&G.mats);
    }
    if (e == hipSuccess) e = dev_alloc((size_t)n * 4u, (void**)&G.gflags);
    if (e == hipSuccess) e = dev_alloc(4, (void**)&d_err);
    if (e == hipSuccess) e = hipMemset(d_err, 0, 4);
    uint32_t base = 0;
    for (uint32_t k = 0; k < num_inputs && e == hipSuccess; ++k) {
        const MiptBuildInput& in = inputs[k];
        GatherInput g{};
        const uint32_t stride = in.vertex_stride_bytes ? in.vertex_stride_bytes : 12u;
        const bool indexed = in.indices && in.index_format != 0;
        g.num_vertices = in.num_vertices; g.vertex_stride = stride; g.num_triangles = counts[k];
        g.index_format = indexed ? in.index_format : 0u;
        g.index_stride = indexed ? (in.index_stride_bytes ? in.index_stride_bytes : 3u * in.index_format) : 0u;
        e = on_device(in.vertices, in.num_vertices ? (size_t)(in.num_vertices - 1) * stride + 12u : 0u, &g.vertices);
        if (e == hipSuccess && indexed) e = on_device(in.indices, counts[k] ? (size_t)(counts[k] - 1) * g.index_stride + 3u * in.index_format : 0u, &g.indices);
        const void* p = nullptr;
        if (e == hipSuccess) { e = on_device(in.sbt_index_offsets, (size_t)counts[k] * 4u, &p); g.sbt_index_offsets = (const uint32_t*)p; }
        if (e == hipSuccess && in.flags) {                                          // the flags array is host memory by contract
            void* d = nullptr;
            e = dev_alloc((size_t)in.num_sbt_records * 4u, &d);
            if (e == hipSuccess) e = hipMemcpy(d, in.flags, (size_t)in.num_sbt_records * 4u, hipMemcpyHostToDevice);
            g.flags = (const uint32_t*)d; g.num_sbt_records = in.num_sbt_records;
        }
        if (in.transform3x4) { g.has_transform = 1u; std::memcpy(g.m, in.transform3x4, 48); }
        if (e == hipSuccess) launch_gather_input(nullptr, g, G.world, G.mats, G.gflags, base, d_err);
        base += counts[k];
    }
    if (e == hipSuccess) e = hipMemcpy(&G.err, d_err, 4, hipMemcpyDeviceToHost);
    G.e = e; G.n = n;
    return MIPT_SUCCESS;
}

static int accel_build_device(MiptContext* ctx, const MiptBuildInput* inputs, uint32_t num_inputs, unsigned long long* handle_out)
{
    *handle_out = 0;
    const auto t0 = std::chrono::steady_clock::now();
    DeviceGather G;
    const int grc = gather_on_device(ctx, inputs, num_inputs, G, /*keep_mats=*/true, 0);
    if (grc != MIPT_SUCCESS || G.n == 0u) return grc;
    const uint32_t n = G.n;
    hipError_t e = G.e;
    const uint32_t err = G.err;
    float* d_world = G.world; uint32_t* d_gf = G.gflags;
    float4* blob = nullptr; uint32_t nn = 0, dep = 0; bool ok = false;
    uint32_t* d_slot = nullptr;                                  // primitive -> triangle record (Accel::d_prim_slot)
    if (e == hipSuccess && err == 0u) e = hipMalloc((void**)&d_slot, (size_t)n * 4u);
    uint32_t records = n;                                        // > n when triangles were cut (MIPT_OPT_BUILD = 4)
    float* d_cells = nullptr;                                    // ... and then the records' cells
    if (e == hipSuccess && err == 0u) e = lbvh_build(nullptr, d_world, d_gf, n, ctx->lbvh_leaf, kStackEntries, &blob, &nn, &dep, &ok, ctx->lbvh_treelets != 0, d_slot, ctx->build_mode >= 2,
                                                      ctx->build_mode == 4 ? ctx->sbvh_extra : 0.0f, ctx->split_below, &records, &d_cells);
    if ((e != hipSuccess || err != 0u || !ok) && d_slot) { (void)hipFree(d_slot); d_slot = nullptr; }
    if ((e != hipSuccess || err != 0u || !ok) && d_cells) { (void)hipFree(d_cells); d_cells = nullptr; }
    if (e != hipSuccess) return ctx->fail(e == hipErrorOutOfMemory ? MIPT_ERROR_OUT_OF_MEMORY : MIPT_ERROR_HIP, "device BVH build failed: %s", hipGetErrorString(e));
    if (err != 0u) return ctx->fail(MIPT_ERROR_INVALID_VALUE, err == 1u ? "a vertex index is out of range" : "an SBT index is >= num_sbt_records");
    if (!ok) {
        ctx->log(3, "ACCEL", "device-built BVH is %u deep (stack holds %d): falling back to the host builder", dep, kStackEntries);
        return MIPT_SUCCESS;
    }
    uint32_t* d_mats = G.release_mats();
    Accel* a = new Accel();
    const size_t nb = (size_t)nn * sizeof(MiptBvhNode), tb = (size_t)records * sizeof(MiptBvhTri);
    a->n_prims = n; a->d_nodes = blob; a->d_tris = blob + nb / sizeof(float4); a->d_mat = d_mats;
    a->num_nodes = nn; a->num_tris = records; a->depth = dep; a->leaf_max = 4; a->builder = records > n ? 4 : (ctx->build_mode >= 2 ? 2 : 1);
    a->d_prim_slot = d_slot; a->d_cells = d_cells;
    if ((e = finish_blob(ctx, a)) != hipSuccess) { free_accel(a); return ctx->fail(MIPT_ERROR_HIP, "device BVH build: %s", hipGetErrorString(e)); }
    a->bytes = nb + tb + 64 + 256 + 8192 + (size_t)nn * 96u + (size_t)n * 4u + (d_slot ? (size_t)n * 4u : 0u) + (d_cells ? (size_t)records * 24u : 0u);   // the whole allocation (lbvh_build sizes it like the host path) + the SBT index table + the primitive -> record table
    a->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const unsigned long long h = ++ctx->next_handle;
    ctx->accels[h] = a;
    *handle_out = h;
    ctx->log(4, "ACCEL", "built BVH2 (device %s): %u tris (%u records), %u nodes, depth %u, %.1f MB, %.4f s", a->builder == 4 ? "SAH with spatial splits" : (a->builder == 2 ? "SAH" : "LBVH"), n, records, nn, dep,
             (double)a->bytes / 1e6, a->seconds);
    return MIPT_SUCCESS;
}

int mipt_accel_build(MiptContext* ctx, const MiptBuildInput* inputs, uint32_t num_inputs, unsigned long long* handle_out)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!inputs || num_inputs == 0 || !handle_out) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "accel_build: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // (the default builder hands inputs of a few thousand triangles to the host builder: a device build is ~0.6 ms of launches, read-backs and
    //  allocations whatever its size -- 300 triangles: 0.7 against 0.2 ms on the host, break-even near 2 000; the trees are alike)
    uint64_t n_hint = 0;
    for (uint32_t k = 0; k < num_inputs; ++k) n_hint += (inputs[k].indices && inputs[k].index_format != 0) ? inputs[k].num_triangles : inputs[k].num_vertices / 3u;
    if (ctx->build_mode == 1 || ((ctx->build_mode == 2 || ctx->build_mode == 4) && n_hint >= (uint64_t)ctx->build_host_below)) {
        const int rc = accel_build_device(ctx, inputs, num_inputs, handle_out);
        if (rc != MIPT_SUCCESS || *handle_out != 0ull) return rc;
    }
    std::vector<float>    world;     // 9 floats per triangle
    std::vector<uint32_t> mats, gflags;
    for (uint32_t k = 0; k < num_inputs; ++k) {
        const MiptBuildInput& in = inputs[k];
        const uint32_t stride = in.vertex_stride_bytes ? in.vertex_stride_bytes : 12u;
        if (stride < 12u || (stride & 3u)) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: vertex stride %u", k, stride);
        if (in.num_vertices && !in.vertices) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: null vertices", k);
        uint32_t n_tris = in.num_triangles;
        if (!in.indices || in.index_format == 0) {
            if (n_tris == 0) n_tris = in.num_vertices / 3u;
            if ((uint64_t)n_tris * 3u > in.num_vertices) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: %u triangles need %u vertices, got %u", k, n_tris, n_tris * 3u, in.num_vertices);
        } else if (in.index_format != 2u && in.index_format != 4u) {
            return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: index format %u (expected 0, 2 or 4)", k, in.index_format);
        }
        std::vector<uint8_t> vbuf, ibuf, mbuf;
        int rc = fetch_to_host(ctx, in.vertices, in.num_vertices ? (size_t)(in.num_vertices - 1) * stride + 12u : 0u, vbuf);
        if (rc) return rc;
        const bool indexed = in.indices && in.index_format != 0;
        const uint32_t istride = indexed ? (in.index_stride_bytes ? in.index_stride_bytes : 3u * in.index_format) : 0u;
        if (indexed) { rc = fetch_to_host(ctx, in.indices, n_tris ? (size_t)(n_tris - 1) * istride + 3u * in.index_format : 0u, ibuf); if (rc) return rc; }
        if (in.sbt_index_offsets) { rc = fetch_to_host(ctx, in.sbt_index_offsets, (size_t)n_tris * 4u, mbuf); if (rc) return rc; }
        const size_t base = world.size() / 9;
        world.resize((base + n_tris) * 9);
        mats.resize(base + n_tris);
        gflags.resize(base + n_tris, 0u);
        if (in.flags && in.num_sbt_records == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: flags without num_sbt_records", k);
        for (uint32_t t = 0; t < n_tris; ++t) {
            uint32_t vi[3];
            if (indexed) {
                const uint8_t* ip = ibuf.data() + (size_t)t * istride;
                for (int c = 0; c < 3; ++c) {
                    if (in.index_format == 2u) { uint16_t v; std::memcpy(&v, ip + 2 * c, 2); vi[c] = v; }
                    else                       { std::memcpy(&vi[c], ip + 4 * c, 4); }
                    if (vi[c] >= in.num_vertices) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: index %u out of range (%u vertices)", k, vi[c], in.num_vertices);
                }
            } else {
                vi[0] = 3 * t; vi[1] = 3 * t + 1; vi[2] = 3 * t + 2;
            }
            float* w = world.data() + (base + t) * 9;
            for (int c = 0; c < 3; ++c) {
                float v[3];
                std::memcpy(v, vbuf.data() + (size_t)vi[c] * stride, 12);
                if (in.transform3x4) {
                    const float* m = in.transform3x4;
                    for (int r = 0; r < 3; ++r) w[3 * c + r] = ((m[4 * r] * v[0] + m[4 * r + 1] * v[1]) + m[4 * r + 2] * v[2]) + m[4 * r + 3];
                } else {
                    w[3 * c] = v[0]; w[3 * c + 1] = v[1]; w[3 * c + 2] = v[2];
                }
            }
            uint32_t m = 0;
            if (in.sbt_index_offsets) std::memcpy(&m, mbuf.data() + (size_t)t * 4, 4);
            mats[base + t] = m;
            if (in.flags) {
                if (m >= in.num_sbt_records) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "input %u: SBT index %u of triangle %u >= num_sbt_records %u", k, m, t, in.num_sbt_records);
                gflags[base + t] = in.flags[m];
            }
        }
    }
    const uint32_t n_tris = (uint32_t)(world.size() / 9);
    Accel* a = new Accel();
    a->n_prims = n_tris;
    const size_t mb = std::max<size_t>(n_tris, 1) * sizeof(uint32_t);
    size_t nb = 0, tb = 0;
    std::vector<uint32_t> prim_slot;
    hipError_t e = hipSuccess;
    {
        // MIPT_OPT_BUILD = 3: spatial splits (bvh_build.cpp, build_sbvh): triangles may be referenced by several leaves; an input it cannot take
        // (a tree deeper than the stack) goes to the object-split builder like every other
        bool spatial = ctx->build_mode == 3 && n_tris > 4u && build_sbvh(world.data(), n_tris, 4, kStackEntries, default_build_threads(), ctx->sbvh_extra, a->host);
        if (spatial && (a->host.tris.size() >= (1u << 23) || a->host.nodes.size() >= (1u << 25))) spatial = false;      // (beyond the entry pool's packed words)
        if (!spatial && !build_bvh(world.data(), n_tris, 4, kStackEntries, default_build_threads(), a->host)) {
            delete a;
            return ctx->fail(MIPT_ERROR_BVH_TOO_DEEP, "BVH over %u triangles does not fit a %d-entry traversal stack", n_tris, kStackEntries);
        }
        a->builder = spatial ? 3 : 0;
        for (MiptBvhTri& t : a->host.tris) t.pad1 = gflags[t.prim];
        if (a->host.tris.size() >= n_tris && n_tris) {            // primitive -> one of its triangle records (the first: spatial splits leave several)
            prim_slot.assign(n_tris, 0xffffffffu);
            for (uint32_t i = 0; i < (uint32_t)a->host.tris.size(); ++i) { uint32_t& sl = prim_slot[a->host.tris[i].prim]; if (sl == 0xffffffffu) sl = i; }
        }
        nb = a->host.nodes.size() * sizeof(MiptBvhNode);
        tb = std::max<size_t>(a->host.tris.size(), 1) * sizeof(MiptBvhTri);
        e = hipMalloc((void**)&a->d_nodes, nb + tb + 64 + 256 + 8192 + a->host.nodes.size() * 96u);      // + room for the node mirrors (finish_blob)
        if (e == hipSuccess) {
            a->d_tris = a->d_nodes + nb / sizeof(float4);
            e = hipMemset((char*)a->d_nodes + nb + tb, 0, 64);
        }
        if (e == hipSuccess) e = hipMemcpy(a->d_nodes, a->host.nodes.data(), nb, hipMemcpyHostToDevice);
        if (e == hipSuccess && !a->host.tris.empty()) e = hipMemcpy(a->d_tris, a->host.tris.data(), a->host.tris.size() * sizeof(MiptBvhTri), hipMemcpyHostToDevice);
        a->num_nodes = (uint32_t)a->host.nodes.size(); a->num_tris = (uint32_t)a->host.tris.size();
        a->depth = a->host.depth; a->leaf_max = a->host.leaf_max; a->seconds = a->host.seconds;
        a->host = HostBvh();                                   // the device copy is the one that is used from here on
    }
    if (e == hipSuccess) e = finish_blob(ctx, a);
    if (e == hipSuccess) e = hipMalloc((void**)&a->d_mat, mb);
    if (e == hipSuccess && n_tris && prim_slot.size() == n_tris) {
        e = hipMalloc((void**)&a->d_prim_slot, (size_t)n_tris * 4u);
        if (e == hipSuccess) e = hipMemcpy(a->d_prim_slot, prim_slot.data(), (size_t)n_tris * 4u, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && n_tris) e = hipMemcpy(a->d_mat, mats.data(), (size_t)n_tris * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        free_accel(a);
        return ctx->fail(e == hipErrorOutOfMemory ? MIPT_ERROR_OUT_OF_MEMORY : MIPT_ERROR_HIP, "accel upload failed: %s", hipGetErrorString(e));
    }
    a->bytes = nb + tb + 64 + 256 + 8192 + (size_t)a->num_nodes * 96u + mb + (a->d_prim_slot ? (size_t)n_tris * 4u : 0u);   // the whole allocation: nodes | triangles | pad | room for both node mirrors, + the SBT index table + the primitive -> record table
    const unsigned long long h = ++ctx->next_handle;
    ctx->accels[h] = a;
    *handle_out = h;
    ctx->log(4, "ACCEL", "built BVH2 (%s): %u tris, %u nodes, depth %u, %.1f MB, %.3f s", a->builder == 3 ? "host SAH with spatial splits" : "host SAH", n_tris,
             a->num_nodes, a->depth, (double)a->bytes / 1e6, a->seconds);
    return MIPT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// Two-level structures: what optixAccelBuild makes of OPTIX_BUILD_INPUT_TYPE_INSTANCES (Scene::buildInstanceAccel,
// sutil/Scene.cpp:1134-1212).  The top level is a BVH2 over the instances' world boxes, built by the same binned-SAH builder
// (every box enters as the degenerate triangle lo-hi-lo, whose bounds are the box).
namespace {

// world -> object matrix of a row-major 3x4 object -> world transform, in double, rounded once
bool invert3x4(const float* m, float* out)
{
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double r[9] = { A / det, -(b * i - c * h) / det, (b * f - c * e) / det,
                          B / det, (a * i - c * g) / det, -(a * f - c * d) / det,
                          C / det, -(a * h - b * g) / det, (a * e - b * d) / det };
    const double tx = m[3], ty = m[7], tz = m[11];
    for (int k = 0; k < 3; ++k) {
        out[4 * k] = (float)r[3 * k]; out[4 * k + 1] = (float)r[3 * k + 1]; out[4 * k + 2] = (float)r[3 * k + 2];
        out[4 * k + 3] = (float)-(r[3 * k] * tx + r[3 * k + 1] * ty + r[3 * k + 2] * tz);
    }
    for (int k = 0; k < 12; ++k) if (!std::isfinite(out[k])) return false;
    return true;
}

// Fills `a` (fresh or being rebuilt) from the instance records.  Transactional: everything is validated and built into local
// tables and FRESH device allocations first; `a` is touched -- the new tables swapped in, the old device copies released -- only
// after the last step has succeeded.  On failure `a` is exactly what it was (a failed mipt_accel_update_instances leaves the
// structure, its mesh references and its export tables intact) and ctx holds the message.
struct IasScratch {
    float4* d_nodes = nullptr; DeviceInstance* d_inst = nullptr; uint32_t* d_order = nullptr; uint32_t* d_mat = nullptr;
    ~IasScratch() { for (void* p : { (void*)d_nodes, (void*)d_inst, (void*)d_order, (void*)d_mat }) if (p) (void)hipFree(p); }
};

int build_ias(MiptContext* ctx, Accel* a, const MiptInstance* inst, uint32_t n)
{
    std::vector<float> boxes((size_t)n * 9);                       // lo, hi, lo per instance
    std::vector<DeviceInstance> recs(n);
    std::vector<float> h_inv((size_t)n * 12, 0.0f);
    std::vector<uint32_t> h_base(n, 0u);
    std::vector<unsigned long long> gas(n, 0ull);
    uint64_t total_tris = 0; uint32_t max_depth = 0;
    bool all_slots = n != 0u;
    for (uint32_t k = 0; k < n; ++k) {
        Accel* g = find_accel(ctx, inst[k].traversableHandle);
        if (!g || g->is_ias) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "instance %u: traversableHandle 0x%llx is not a triangle acceleration structure of this context",
                                              k, inst[k].traversableHandle);
        if (!invert3x4(inst[k].transform, &h_inv[(size_t)k * 12])) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instance %u: singular transform", k);
        // the mesh's object-space box = the union of its root's two child boxes, read back from the device
        MiptBvhNode root;
        HIPCHK(ctx, hipMemcpy(&root, g->d_nodes, sizeof root, hipMemcpyDeviceToHost));
        const float lo[3] = { std::fmin(root.c0x[0], root.c1x[0]), std::fmin(root.c0y[0], root.c1y[0]), std::fmin(root.c0z[0], root.c1z[0]) };
        const float hi[3] = { std::fmax(root.c0x[1], root.c1x[1]), std::fmax(root.c0y[1], root.c1y[1]), std::fmax(root.c0z[1], root.c1z[1]) };
        // world box (Arvo), grown by a few ulps of its largest coordinate: the box is tested against the WORLD ray, the
        // triangles against the OBJECT-space ray, and the two roundings must never disagree about a grazing hit
        const float* m = inst[k].transform;
        float* bx = &boxes[(size_t)k * 9];
        float mag = 0.0f;
        for (int r = 0; r < 3; ++r) {
            double l = m[4 * r + 3], h = m[4 * r + 3];
            for (int c = 0; c < 3; ++c) {
                const double p = (double)m[4 * r + c] * lo[c], q = (double)m[4 * r + c] * hi[c];
                l += std::fmin(p, q); h += std::fmax(p, q);
            }
            bx[r] = (float)l; bx[3 + r] = (float)h;
            mag = std::fmax(mag, std::fmax(std::fabs(bx[r]), std::fabs(bx[3 + r])));
        }
        const float pad = mag * 1.9073486e-6f + 1e-30f;              // 2^-19 of the largest coordinate
        for (int r = 0; r < 3; ++r) { bx[r] -= pad; bx[3 + r] += pad; bx[6 + r] = bx[r]; }
        DeviceInstance& d = recs[k];
        std::memcpy(d.inv, &h_inv[(size_t)k * 12], 48);
        d.nodes = g->d_nodes; d.tris = g->d_tris; d.prim_base = (uint32_t)total_tris;
        d.prim_slot = g->d_prim_slot; all_slots = all_slots && g->d_prim_slot != nullptr;
        h_base[k] = (uint32_t)total_tris; gas[k] = inst[k].traversableHandle;
        total_tris += g->n_prims;
        max_depth = std::max(max_depth, g->depth);
        if (total_tris >= (1ull << 31)) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instances reference more than 2^31 triangles");
    }
    HostBvh top;
    if (!build_bvh(boxes.data(), n, 1, kStackEntries, 1, top)) return ctx->fail(MIPT_ERROR_BVH_TOO_DEEP, "instance BVH over %u instances is too deep", n);
    std::vector<uint32_t> h_order(top.tris.size());
    for (size_t k = 0; k < top.tris.size(); ++k) h_order[k] = top.tris[k].prim;
    // The entry-pool form (k_trace_static_inst, the path kernel) needs every mesh's BVH4 mirror and at least two instances; the whole
    // structure then lives in ONE allocation behind one buffer descriptor: top-level fp32 nodes | instance records | BVH4 nodes
    // (the top level's mirror in the first T slots, then each distinct mesh's, references rebased) | the meshes' triangles.
    struct MeshSlot { const Accel* g; uint32_t node_base, tri_base; };
    std::vector<MeshSlot> meshes;
    std::unordered_map<unsigned long long, size_t> mesh_of;
    const uint32_t T = (uint32_t)top.nodes.size();
    bool fast = ctx->node_format == 2 && n >= 1u && T >= 1u;      // (one instance -- most glTF files -- too: its top level is a root with one child)
    uint64_t n4_total = T, tri_total = 0;
    for (uint32_t k = 0; k < n && fast; ++k) {
        if (mesh_of.count(gas[k])) continue;
        const Accel* g = find_accel(ctx, gas[k]);
        if (g->node_fmt != 2u || g->node4_off == 0u || g->num_nodes4 == 0u || g->leaf_max > 4u) { fast = false; break; }
        mesh_of[gas[k]] = meshes.size();
        meshes.push_back({ g, (uint32_t)n4_total, (uint32_t)tri_total });
        n4_total += g->num_nodes4; tri_total += g->num_tris;
    }
    const size_t o_inst = ((size_t)T * sizeof(MiptBvhNode) + 255) & ~(size_t)255;
    const size_t o_r4 = (o_inst + (size_t)n * sizeof(DeviceInstance) + 127) & ~(size_t)127;        // 16 B per top-level BVH4 node (at most T of them)
    const size_t o_n4 = (o_r4 + (size_t)T * 16u + 127) & ~(size_t)127;
    const size_t o_tri = (o_n4 + (size_t)n4_total * 64u + 127) & ~(size_t)127;
    const size_t arena = o_tri + (size_t)tri_total * sizeof(MiptBvhTri) + 64;
    if (n4_total >= (1ull << 25) || tri_total >= (1ull << 23) || arena >= (1ull << 31)) fast = false;      // the pool's packed entries, 32-bit offsets
    // fresh device copies; the structure's current ones stay in place (and in use by nothing: the callers synchronise) until all exist
    IasScratch S;
    uint32_t stack4 = 0u;
    if (fast) {
        for (uint32_t k = 0; k < n; ++k) recs[k].root4 = meshes[mesh_of[gas[k]]].node_base;
        HIPCHK(ctx, hipMalloc((void**)&S.d_nodes, arena));
        char* base = reinterpret_cast<char*>(S.d_nodes);
        HIPCHK(ctx, hipMemcpy(base, top.nodes.data(), (size_t)T * sizeof(MiptBvhNode), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(base + o_inst, recs.data(), (size_t)n * sizeof(DeviceInstance), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemset(base + o_tri + (size_t)tri_total * sizeof(MiptBvhTri), 0, 64));
        for (const MeshSlot& m : meshes) {
            launch_rebase_nodes4(nullptr, reinterpret_cast<const char*>(m.g->d_nodes) + m.g->node4_off, base + o_n4 + (size_t)m.node_base * 64u, m.g->num_nodes4, m.node_base, m.tri_base);
            HIPCHK(ctx, hipMemcpyAsync(base + o_tri + (size_t)m.tri_base * sizeof(MiptBvhTri), m.g->d_tris, (size_t)m.g->num_tris * sizeof(MiptBvhTri), hipMemcpyDeviceToDevice, nullptr));
        }
    } else {
        HIPCHK(ctx, hipMalloc((void**)&S.d_nodes, top.nodes.size() * sizeof(MiptBvhNode)));
        HIPCHK(ctx, hipMemcpy(S.d_nodes, top.nodes.data(), top.nodes.size() * sizeof(MiptBvhNode), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMalloc((void**)&S.d_inst, (size_t)n * sizeof(DeviceInstance)));
        HIPCHK(ctx, hipMemcpy(S.d_inst, recs.data(), (size_t)n * sizeof(DeviceInstance), hipMemcpyHostToDevice));
    }
    HIPCHK(ctx, hipMalloc((void**)&S.d_order, std::max<size_t>(h_order.size(), 1) * 4u));
    if (!h_order.empty()) HIPCHK(ctx, hipMemcpy(S.d_order, h_order.data(), h_order.size() * 4u, hipMemcpyHostToDevice));
    if (fast) {
        // the top level's mirrors, on the device: heights, fp16 boxes (the fp32 nodes are rounded in place, as a mesh's are), BVH4 collapse
        // into the first T slots of the BVH4 array; its leaves then name instances instead of positions of inst_order
        char* base = reinterpret_cast<char*>(S.d_nodes);
        const size_t hb = topo_heights_scratch_bytes(T), cb = collapse_bvh4_scratch_bytes(T);
        ctx->scratch_reset();
        HIPCHK(ctx, ctx->scratch_ensure(hb + cb + (size_t)T * (32u + 1u) + 4096u));
        uint8_t* d_h = (uint8_t*)ctx->scratch_take(T);
        void* d_n32 = ctx->scratch_take((size_t)T * 32u);
        HIPCHK(ctx, topo_heights(nullptr, S.d_nodes, T, d_h, ctx->scratch_take(hb), hb));
        bool ok = false;
        HIPCHK(ctx, pack_nodes32(nullptr, S.d_nodes, T, d_n32, &ok));
        uint32_t cnt4 = 0u;
        // (the top level is collapsed under a stack budget that leaves room for the deepest mesh: k_trace_tiled_inst keeps ONE 32-row
        //  stack per lane for both levels, + 1 for the node a lane stands at)
        uint32_t mesh_need = 0u;
        for (const MeshSlot& m : meshes) mesh_need = std::max(mesh_need, m.g->stack_need);
        const uint32_t tl_budget = mesh_need + 1u < 31u ? std::max<uint32_t>(31u - mesh_need, top.depth) : 32u;
        if (ok) HIPCHK(ctx, collapse_bvh4(nullptr, d_n32, d_h, T, top.depth, base + o_n4, &cnt4, &stack4, &ok, ctx->scratch_take(cb), cb, tl_budget));
        if (ok) {
            launch_tlas4_leaves(nullptr, base + o_n4, cnt4, S.d_order, reinterpret_cast<const DeviceInstance*>(base + o_inst), reinterpret_cast<uint32_t*>(base + o_r4));
            HIPCHK(ctx, hipStreamSynchronize(nullptr)); HIPCHK(ctx, hipGetLastError());
        }
        if (!ok) {
            // (a top level that does not suit fp16 boxes: the one-ray-per-lane walk over the same allocation)
            fast = false;
            HIPCHK(ctx, hipMalloc((void**)&S.d_inst, (size_t)n * sizeof(DeviceInstance)));
            HIPCHK(ctx, hipMemcpy(S.d_inst, recs.data(), (size_t)n * sizeof(DeviceInstance), hipMemcpyHostToDevice));
        }
    }
    // hit-group record index per FLATTENED primitive: the meshes' own tables, one after the other
    HIPCHK(ctx, hipMalloc((void**)&S.d_mat, std::max<uint64_t>(total_tris, 1) * 4u));
    for (uint32_t k = 0; k < n; ++k) {
        const Accel* g = find_accel(ctx, inst[k].traversableHandle);
        if (g->n_prims) HIPCHK(ctx, hipMemcpy(S.d_mat + h_base[k], g->d_mat, (size_t)g->n_prims * 4u, hipMemcpyDeviceToDevice));
    }
    // ---- commit: nothing below can fail
    std::swap(a->d_nodes, S.d_nodes); std::swap(a->d_inst, S.d_inst); std::swap(a->d_order, S.d_order); std::swap(a->d_mat, S.d_mat);   // ~IasScratch frees the old ones
    a->h_inv.swap(h_inv); a->h_base.swap(h_base); a->gas.swap(gas); a->h_order.swap(h_order); a->h_tlas = top.nodes;
    a->d_tris = nullptr; a->tri_byte_off = 0; a->blob_bytes = 0; a->node32_off = 0;
    a->num_nodes = (uint32_t)top.nodes.size(); a->num_tris = (uint32_t)total_tris; a->n_prims = (uint32_t)total_tris;
    a->depth = top.depth + max_depth; a->leaf_max = 1; a->is_ias = true; a->stale = false;
    uint32_t mesh_stack = 0u;
    for (const MeshSlot& m : meshes) mesh_stack = std::max(mesh_stack, m.g->stack_need);
    a->inst_slots = all_slots;
    a->inst_fast = fast; a->stack_need = fast ? stack4 + mesh_stack : 0u;      // the most entries a lane's stack over BOTH levels can hold
    a->inst_off = fast ? (uint32_t)o_inst : 0u; a->inst_node4_off = fast ? (uint32_t)o_n4 : 0u; a->inst_tri_off = fast ? (uint32_t)o_tri : 0u;
    a->inst_blob_bytes = fast ? (uint32_t)arena : 0u; a->inst_root4_off = fast ? (uint32_t)o_r4 : 0u;
    a->bytes = (fast ? arena : top.nodes.size() * sizeof(MiptBvhNode) + (size_t)n * sizeof(DeviceInstance)) + (size_t)n * 4u + total_tris * 4u;
    return MIPT_SUCCESS;
}

}  // namespace

int mipt_accel_build_instances(MiptContext* ctx, const MiptInstance* instances, uint32_t num_instances, unsigned long long* handle_out)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!instances || num_instances == 0 || !handle_out) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "accel_build_instances: null argument");
    if (num_instances >= (1u << 27)) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "too many instances");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    Accel* a = new Accel();
    const int rc = build_ias(ctx, a, instances, num_instances);
    if (rc != MIPT_SUCCESS) { free_accel(a); return rc; }
    for (unsigned long long h : a->gas) find_accel(ctx, h)->users++;
    a->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const unsigned long long h = ++ctx->next_handle;
    ctx->accels[h] = a;
    *handle_out = h;
    ctx->log(4, "ACCEL", "built instance structure: %u instances, %u top-level nodes, %u triangles referenced, depth %u, %.4f s", num_instances,
             a->num_nodes, a->num_tris, a->depth, a->seconds);
    return MIPT_SUCCESS;
}

int mipt_accel_update_instances(MiptContext* ctx, unsigned long long handle, const MiptInstance* instances, uint32_t num_instances)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    Accel* a = find_accel(ctx, handle);
    if (!a || !a->is_ias) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "0x%llx is not an instance acceleration structure", handle);
    if (!instances || num_instances != a->gas.size()) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "accel_update_instances: %u instances given, built from %zu", num_instances, a->gas.size());
    for (uint32_t k = 0; k < num_instances; ++k)
        if (instances[k].traversableHandle != a->gas[k]) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "accel_update_instances: instance %u refers to another mesh than at build time", k);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    return build_ias(ctx, a, instances, num_instances);           // a few thousand boxes: rebuilding the top level IS the update
}

int mipt_accel_export_instances(const MiptContext* ctx, unsigned long long handle, MiptBvhNode* nodes_out, uint32_t* order_out, float* inv12_out,
                                uint32_t* prim_base_out, unsigned long long* gas_handles_out)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    const Accel* a = find_accel(ctx, handle);
    if (!a || !a->is_ias) return MIPT_ERROR_INVALID_HANDLE;
    // the top level straight from device memory, the tables from the host copies they were uploaded from
    if (hipSetDevice(ctx->device) != hipSuccess) return MIPT_ERROR_HIP;
    if (nodes_out && hipMemcpy(nodes_out, a->d_nodes, (size_t)a->num_nodes * sizeof(MiptBvhNode), hipMemcpyDeviceToHost) != hipSuccess) return MIPT_ERROR_HIP;
    if (order_out) std::memcpy(order_out, a->h_order.data(), a->h_order.size() * 4u);
    if (inv12_out) std::memcpy(inv12_out, a->h_inv.data(), a->h_inv.size() * 4u);
    if (prim_base_out) std::memcpy(prim_base_out, a->h_base.data(), a->h_base.size() * 4u);
    if (gas_handles_out) std::memcpy(gas_handles_out, a->gas.data(), a->gas.size() * 8u);
    return MIPT_SUCCESS;
}

int mipt_accel_update(MiptContext* ctx, unsigned long long handle, const MiptBuildInput* inputs, uint32_t num_inputs)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!inputs || num_inputs == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "accel_update: null argument");
    Accel* a = find_accel(ctx, handle);
    if (!a) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "unknown traversable 0x%llx", handle);
    if (a->is_ias) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "0x%llx is an instance structure: use mipt_accel_update_instances", handle);
    // (a tree with spatial splits -- builders 3 and 4: more records than triangles -- is refitted like any other.  Builder 4 keeps the CELL every
    //  record's box was clipped to, and a record's new box is its moved triangle's box cut by that cell: the tree keeps what the cuts bought as
    //  long as triangles stay near their places.  Builder 3 keeps no cells: every record gets the box of its WHOLE triangle -- a sound tree,
    //  exact records, but the overlap the splits removed is back until the next build)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());                         // launches that still read the old boxes
    const auto t0 = std::chrono::steady_clock::now();
    DeviceGather G;
    const int grc = gather_on_device(ctx, inputs, num_inputs, G, /*keep_mats=*/false, 0);
    if (grc != MIPT_SUCCESS) return grc;
    if (G.n == 0u || G.n != a->n_prims || a->num_nodes == 0u)
        return ctx->fail(MIPT_ERROR_INVALID_VALUE, "accel_update: the inputs hold %u triangles, the traversable was built from %u", G.n, a->n_prims);
    if (G.e != hipSuccess) return ctx->fail(MIPT_ERROR_HIP, "accel_update: %s", hipGetErrorString(G.e));
    if (G.err != 0u) return ctx->fail(MIPT_ERROR_INVALID_VALUE, G.err == 1u ? "a vertex index is out of range" : "an SBT index is >= num_sbt_records");
    if (!a->d_height) return ctx->fail(MIPT_ERROR_INTERNAL, "accel_update: the traversable has no height table");
    hipError_t e = lbvh_refit(nullptr, G.world, a->num_tris, a->d_nodes, a->num_nodes, a->d_tris, a->d_height, a->depth, a->d_cells);       // (num_tris: the triangle RECORDS)
    if (e == hipSuccess) e = finish_blob(ctx, a);            // the refitted boxes are fp32: round and pack them again
    if (e != hipSuccess) return ctx->fail(MIPT_ERROR_HIP, "accel_update: %s", hipGetErrorString(e));
    // an instance structure that references this mesh holds copies of its mirrors and a world box made from its old root box: it has to
    // be brought up to date (mipt_accel_update_instances, as OptiX requires an IAS update after a GAS update) before it is traced again
    if (a->users != 0u)
        for (auto& kv : ctx->accels)
            if (kv.second->is_ias && std::find(kv.second->gas.begin(), kv.second->gas.end(), handle) != kv.second->gas.end()) kv.second->stale = true;
    ctx->log(4, "ACCEL", "refitted BVH2: %u tris, %u nodes, %.4f s", a->num_tris, a->num_nodes,
             std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return MIPT_SUCCESS;
}

int mipt_accel_destroy(MiptContext* ctx, unsigned long long handle)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    auto it = ctx->accels.find(handle);
    if (it == ctx->accels.end()) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "unknown traversable 0x%llx", handle);
    if (it->second->users != 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "traversable 0x%llx is still referenced by %u instance structure(s)", handle, it->second->users);
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (unsigned long long g : it->second->gas) if (Accel* m = find_accel(ctx, g)) if (m->users) m->users--;
    free_accel(it->second);
    ctx->accels.erase(it);
    return MIPT_SUCCESS;
}

int mipt_accel_info(const MiptContext* ctx, unsigned long long handle, MiptAccelInfo* info)
{
    if (!ctx || !info) return MIPT_ERROR_INVALID_VALUE;
    const Accel* a = find_accel(ctx, handle);
    if (!a) return MIPT_ERROR_INVALID_HANDLE;
    info->num_nodes = a->num_nodes;
    info->num_triangles = a->num_tris;
    info->depth = a->depth;
    info->leaf_max = a->leaf_max;
    info->device_bytes = a->bytes;
    info->build_seconds = a->seconds;
    info->node_format = a->node32_off ? a->node_fmt : 0u; info->num_instances = a->is_ias ? (uint32_t)a->gas.size() : 0u;
    if (a->is_ias && a->inst_fast) info->node_format = 2u;       // the entry-pool arena: BVH4 mirrors of the top level and of every mesh
    info->stack_entries = info->node_format == 2u ? a->stack_need : a->depth;
    info->mirror_nodes = info->node_format == 2u ? a->num_nodes4 : (info->node_format == 1u ? a->num_nodes : 0u);
    return MIPT_SUCCESS;
}

int mipt_accel_export(const MiptContext* ctx, unsigned long long handle, MiptBvhNode* nodes_out, MiptBvhTri* tris_out)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    const Accel* a = find_accel(ctx, handle);
    if (!a) return MIPT_ERROR_INVALID_HANDLE;
    if (a->is_ias) return MIPT_ERROR_INVALID_HANDLE;             // mipt_accel_export_instances
    // exported from DEVICE memory, so the checker sees what the kernels read
    (void)hipSetDevice(ctx->device);
    if (nodes_out && hipMemcpy(nodes_out, a->d_nodes, (size_t)a->num_nodes * sizeof(MiptBvhNode), hipMemcpyDeviceToHost) != hipSuccess) return MIPT_ERROR_HIP;
    if (tris_out && a->num_tris &&
        hipMemcpy(tris_out, a->d_tris, (size_t)a->num_tris * sizeof(MiptBvhTri), hipMemcpyDeviceToHost) != hipSuccess) return MIPT_ERROR_HIP;
    return MIPT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
struct MiptHostBvh { HostBvh bvh; };

int mipt_host_bvh_create(const float* vertices9, uint32_t num_triangles, MiptHostBvh** out)
{
    if (!out || (num_triangles && !vertices9)) return MIPT_ERROR_INVALID_VALUE;
    MiptHostBvh* h = new MiptHostBvh();
    if (!build_bvh(vertices9, num_triangles, 4, kStackEntries, default_build_threads(), h->bvh)) { delete h; return MIPT_ERROR_BVH_TOO_DEEP; }
    *out = h;
    return MIPT_SUCCESS;
}

int mipt_host_bvh_create_spatial(const float* vertices9, uint32_t num_triangles, float extra, MiptHostBvh** out)
{
    if (!out || (num_triangles && !vertices9) || !(extra >= 0.0f) || extra > 4.0f) return MIPT_ERROR_INVALID_VALUE;
    MiptHostBvh* h = new MiptHostBvh();
    // (as mipt_accel_build with MIPT_OPT_BUILD = 3: inputs the spatial builder does not take get the object-split tree)
    const bool ok = (num_triangles > 4u && build_sbvh(vertices9, num_triangles, 4, kStackEntries, default_build_threads(), extra, h->bvh)) ||
                    build_bvh(vertices9, num_triangles, 4, kStackEntries, default_build_threads(), h->bvh);
    if (!ok) { delete h; return MIPT_ERROR_BVH_TOO_DEEP; }
    *out = h;
    return MIPT_SUCCESS;
}

int mipt_host_bvh_info(const MiptHostBvh* h, MiptAccelInfo* info)
{
    if (!h || !info) return MIPT_ERROR_INVALID_VALUE;
    info->num_nodes = (uint32_t)h->bvh.nodes.size();
    info->num_triangles = (uint32_t)h->bvh.tris.size();
    info->depth = h->bvh.depth;
    info->leaf_max = h->bvh.leaf_max;
    info->device_bytes = 0;
    info->build_seconds = h->bvh.seconds;
    info->node_format = 0u; info->num_instances = 0u; info->stack_entries = h->bvh.depth; info->mirror_nodes = 0u;
    return MIPT_SUCCESS;
}

int mipt_host_bvh_copy(const MiptHostBvh* h, MiptBvhNode* nodes_out, MiptBvhTri* tris_out)
{
    if (!h) return MIPT_ERROR_INVALID_VALUE;
    if (nodes_out) std::memcpy(nodes_out, h->bvh.nodes.data(), h->bvh.nodes.size() * sizeof(MiptBvhNode));
    if (tris_out && !h->bvh.tris.empty()) std::memcpy(tris_out, h->bvh.tris.data(), h->bvh.tris.size() * sizeof(MiptBvhTri));
    return MIPT_SUCCESS;
}

int mipt_host_bvh_destroy(MiptHostBvh* h)
{
    delete h;
    return MIPT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
int mipt_sbt_set_hitgroups(MiptContext* ctx, const MiptHitGroupData* records, uint32_t count)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!records || count == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "sbt_set_hitgroups: no records");
    for (uint32_t i = 0; i < count; ++i)
        if (!records[i].vertices) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "hit-group record %u has a null vertex pointer", i);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    if (ctx->d_hitgroups) { (void)hipFree(ctx->d_hitgroups); ctx->d_hitgroups = nullptr; ctx->n_hitgroups = 0; }
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_hitgroups, (size_t)count * sizeof(MiptHitGroupData)));
    HIPCHK(ctx, hipMemcpy(ctx->d_hitgroups, records, (size_t)count * sizeof(MiptHitGroupData), hipMemcpyHostToDevice));
    ctx->n_hitgroups = count;
    ctx->common_vertices = records[0].vertices;
    for (uint32_t i = 1; i < count; ++i) if (records[i].vertices != records[0].vertices) ctx->common_vertices = nullptr;
    return MIPT_SUCCESS;
}

int mipt_sbt_set_miss(MiptContext* ctx, const MiptMissData* miss)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!miss) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "sbt_set_miss: null");
    ctx->miss = *miss;
    return MIPT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
int mipt_launch(MiptContext* ctx, void* stream, const MiptParams* p, unsigned int width, unsigned int height, unsigned int depth)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    Accel* acc = nullptr;
    int rc = check_launch(ctx, p, width, height, depth, &acc);
    if (rc) return rc;
    if (!p->accum_buffer) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "params.accum_buffer is null");
    FrameConsts fc = make_consts(ctx, p);
    fc.world = 0; fc.rank = 0;
    fc.tile_shift = (uint32_t)ctx->tile_shift;
    const uint32_t tw = 1u << fc.tile_shift, th = 64u >> fc.tile_shift;
    const uint32_t tiles = ((width + tw - 1u) / tw) * ((height + th - 1u) / th);
    fc.n_paths = tiles * 64u;
    return run_wavefront(ctx, (hipStream_t)stream, fc, acc, (size_t)width * height, (float4*)p->accum_buffer, (uint32_t*)p->frame_buffer);
}

int mipt_launch_shard(MiptContext* ctx, void* stream, const MiptParams* p, unsigned int width, unsigned int height,
                      unsigned int rank, unsigned int world, void* compact_accum)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    Accel* acc = nullptr;
    int rc = check_launch(ctx, p, width, height, 1, &acc);
    if (rc) return rc;
    if (world == 0 || rank >= world) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "rank %u / world %u", rank, world);
    if (!compact_accum) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "compact_accum is null");
    FrameConsts fc = make_consts(ctx, p);
    fc.world = world; fc.rank = rank;
    fc.n_paths = shard_num_samples(width, height, world);
    return run_wavefront(ctx, (hipStream_t)stream, fc, acc, fc.n_paths, (float4*)compact_accum, nullptr);
}

int mipt_unpermute(MiptContext* ctx, void* stream, const void* gathered, unsigned int world, unsigned int width,
                   unsigned int height, void* accum_buffer, void* frame_buffer)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!gathered || !accum_buffer || world == 0 || width == 0 || height == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "unpermute: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->last_stream = (hipStream_t)stream;
    launch_unpermute((hipStream_t)stream, (const float4*)gathered, world, width, height, (float4*)accum_buffer, (uint32_t*)frame_buffer);
    HIPCHK(ctx, hipGetLastError());
    return MIPT_SUCCESS;
}

int mipt_shard_num_samples(unsigned int width, unsigned int height, unsigned int world, uint32_t* n_out)
{
    if (!n_out || world == 0) return MIPT_ERROR_INVALID_VALUE;
    *n_out = shard_num_samples(width, height, world);
    return MIPT_SUCCESS;
}

int mipt_shard_pixel(unsigned int width, unsigned int height, unsigned int world, unsigned int rank, uint32_t sample_idx,
                     int32_t* px, int32_t* py)
{
    (void)height;
    if (!px || !py || world == 0 || rank >= world) return MIPT_ERROR_INVALID_VALUE;
    int x, y;
    shard_pixel(width, world, rank, sample_idx, x, y);
    *px = x; *py = y;
    return MIPT_SUCCESS;
}

int mipt_sync(MiptContext* ctx)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->last_stream));
    return check_device_faults(ctx);
}

int mipt_get_launch_stats(MiptContext* ctx, MiptLaunchStats* stats)
{
    if (!ctx || !stats) return MIPT_ERROR_INVALID_VALUE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->last_stream));
    if (const int frc = check_device_faults(ctx)) return frc;
    DeviceCounters h;
    HIPCHK(ctx, hipMemcpy(&h, ctx->d_ctr, sizeof h, hipMemcpyDeviceToHost));
    for (uint32_t b = 0; ctx->bins_live && b < kStatBins; ++b) {      // the path kernels' end-of-wave statistics (DeviceCounters::bins)
        h.closest_rays += h.bins[b * kStatBinWords]; h.shadow_rays += h.bins[b * kStatBinWords + 1u];
        h.tail_max = std::max(h.tail_max, h.bins[b * kStatBinWords + 2u]);
    }
    if (ctx->advance_pending) {            // advance() of the launch's last bounce (kernels.hip), on the copy
        if (h.n_cur) h.iterations += 1;
        h.closest_rays += h.n_next; h.shadow_rays += h.n_shadow;
    }
    stats->closest_rays = h.closest_rays; stats->shadow_rays = h.shadow_rays;
    stats->node_visits = h.node_visits[0] + h.node_visits[1]; stats->tri_tests = h.tri_tests[0] + h.tri_tests[1];
    stats->shadow_node_visits = h.node_visits[1]; stats->shadow_tri_tests = h.tri_tests[1];
    if (std::getenv("MIPT_PROBE_PRINT"))       // probe builds (csrc/Makefile probe-%): the in-kernel stamps of the last launch
        std::fprintf(stderr, "probe dbg: %llu %llu %llu %llu %llu %llu\n", h.dbg[0], h.dbg[1], h.dbg[2], h.dbg[3], h.dbg[4], h.dbg[5]);
    stats->bounces = ctx->bounces_on_device ? (uint32_t)(h.iterations + h.tail_max) : ctx->last_bounces; stats->reserved = 0;
    return MIPT_SUCCESS;
}

int mipt_get_launch_timings(MiptContext* ctx, MiptLaunchTimings* out)
{
    if (!ctx || !out) return MIPT_ERROR_INVALID_VALUE;
    std::memset(out, 0, sizeof *out);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->last_stream));
    for (const auto& t : ctx->timed) {
        float ms = 0.0f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, t.a, t.b));
        switch (t.kind) {
        case T_RAYGEN: out->ms_raygen += ms; break;
        case T_CLOSEST: out->ms_trace_closest += ms; out->n_trace_closest++; break;
        case T_SHADE: out->ms_shade += ms; break;
        case T_ANY: out->ms_trace_any += ms; out->n_trace_any++; break;
        case T_RESOLVE: out->ms_resolve += ms; break;
        default: break;
        }
    }
    ctx->timed.clear(); ctx->events_used = 0;      // sums cover every launch since the previous call
    return MIPT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// SDK/imgui_test
// ------------------------------------------------------------------------------------------------
static_assert(sizeof(MiptImguiCamera) == 92 && sizeof(MiptImguiLight) == 44 && sizeof(MiptImguiParams) == 128,
              "imgui_test device-object layouts (camera.h:155-173, light.h:10-51, optixTriangle.h:42-108)");

static int ensure_group_capacity(MiptContext* ctx, size_t need);

int mipt_imgui_launch(MiptContext* ctx, void* stream, const MiptImguiParams* p, unsigned int width, unsigned int height, unsigned int depth)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!p) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "params is null");
    if (depth != 1 || width == 0 || height == 0 || width != p->image_width || height != p->image_height)
        return ctx->fail(MIPT_ERROR_INVALID_VALUE, "launch size %ux%ux%u does not match params %ux%u", width, height, depth, p->image_width, p->image_height);
    if (p->samples_per_frame == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "samples_per_frame must be >= 1");
    if (p->dt == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "dt is 0: call frame_step() before the launch (tracer_window.cpp:92)");
    if (!p->camera || !p->film || !p->normals || !p->mat_indices || !p->materials || p->nmaterials <= 0)
        return ctx->fail(MIPT_ERROR_INVALID_VALUE, "camera / film / normals / mat_indices / materials must be set");
    if (p->nlights < 0 || p->nlights > 31 || (p->nlights > 0 && !p->lights)) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "0..31 lights supported");
    Accel* acc = find_accel(ctx, p->handle);
    if (!acc) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "params.handle 0x%llx is not a traversable of this context", p->handle);
    if (acc->stale) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instance structure 0x%llx references a mesh that was refitted since it was built: call mipt_accel_update_instances first", p->handle);
    if (const int frc = check_device_faults(ctx)) return frc;         // raised by an earlier launch, reported once
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint32_t n_paths = ((width + 7u) / 8u) * ((height + 7u) / 8u) * 64u;
    int rc = ensure_capacity(ctx, n_paths, (size_t)width * height);
    if (rc) return rc;
    rc = ensure_group_capacity(ctx, (size_t)n_paths * (size_t)(p->nlights + 1));
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ctx->last_stream = st;
    ctx->advance_pending = false; ctx->bins_live = false;
    HIPCHK(ctx, hipMemsetAsync(&ctx->d_ctr->closest_rays, 0, (14 + kStatBins * kStatBinWords) * sizeof(unsigned long long), st));
    imgui_run(st, *p, n_paths, device_bvh(acc, ctx), ctx->q[0], ctx->hits, ctx->im_sq, ctx->im_flags, ctx->result, ctx->rg_seed,
              ctx->d_ctr, trace_opts(ctx));
    ctx->last_bounces = p->samples_per_frame; ctx->bounces_on_device = false;
    HIPCHK(ctx, hipGetLastError());
    return MIPT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// SDK/cuda/whitted.cu (optixMeshViewer's pipeline)
// ------------------------------------------------------------------------------------------------
static_assert(sizeof(MiptLight) == 36, "struct Light layout (cuda/Light.h:33-71)");

static int ensure_group_capacity(MiptContext* ctx, size_t need)
{
    if (need <= ctx->im_cap) return MIPT_SUCCESS;
    HIPCHK(ctx, hipDeviceSynchronize());
    if (ctx->im_sq.ray_o) (void)hipFree(ctx->im_sq.ray_o);
    if (ctx->im_sq.contrib) (void)hipFree(ctx->im_sq.contrib);
    if (ctx->im_flags) (void)hipFree(ctx->im_flags);
    if (ctx->wh_e1) (void)hipFree(ctx->wh_e1);
    ctx->im_sq = ShadowQueue{}; ctx->im_flags = nullptr; ctx->wh_e1 = nullptr; ctx->im_cap = 0;
    HIPCHK(ctx, hipMalloc((void**)&ctx->im_sq.ray_o, need * 2 * sizeof(float4)));
    ctx->im_sq.ray_d = ctx->im_sq.ray_o + need;
    HIPCHK(ctx, hipMalloc((void**)&ctx->im_sq.contrib, need * sizeof(float4)));
    HIPCHK(ctx, hipMalloc((void**)&ctx->im_flags, need * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc((void**)&ctx->wh_e1, need * sizeof(float4)));
    ctx->im_cap = need;
    return MIPT_SUCCESS;
}

// materials + textures of the Whitted programs (library copies); shared by the two set_geometry forms
static int set_whitted_tables(MiptContext* ctx, const MiptMaterialData* materials, uint32_t num_materials, const MiptTextureData* textures, uint32_t num_textures)
{
    if (!materials || num_materials == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "whitted_set_geometry: >= 1 material required");
    if (num_textures && !textures) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "whitted_set_geometry: num_textures without textures");
    for (uint32_t i = 0; i < num_textures; ++i)
        if (textures[i].texels && textures[i].width && textures[i].bits != 8 && textures[i].bits != 16)
            return ctx->fail(MIPT_ERROR_INVALID_VALUE, "texture %u: %u bits per component (8 or 16 supported, Scene.cpp:587-600)", i, textures[i].bits);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    if (ctx->d_wh_materials) { (void)hipFree(ctx->d_wh_materials); ctx->d_wh_materials = nullptr; }
    if (ctx->d_wh_textures) { (void)hipFree(ctx->d_wh_textures); ctx->d_wh_textures = nullptr; }
    for (void* t : ctx->wh_texels) if (t) (void)hipFree(t);
    ctx->wh_texels.clear(); ctx->n_wh_textures = 0;
    for (void** p : { (void**)&ctx->d_wh_inst, (void**)&ctx->d_wh_meshes, (void**)&ctx->d_wh_prim_inst }) if (*p) { (void)hipFree(*p); *p = nullptr; }
    ctx->wh_geo = WhGeometry{};
    ctx->wh_inst_base.clear(); ctx->wh_inst_total = 0;
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_wh_materials, (size_t)num_materials * sizeof(MiptMaterialData)));
    HIPCHK(ctx, hipMemcpy(ctx->d_wh_materials, materials, (size_t)num_materials * sizeof(MiptMaterialData), hipMemcpyHostToDevice));
    ctx->n_wh_materials = num_materials;
    ctx->wh_any_hit = ctx->wh_blend = false;
    for (uint32_t i = 0; i < num_materials; ++i) {
        if (materials[i].alpha_mode != 0) ctx->wh_any_hit = true;
        if (materials[i].alpha_mode == 2) ctx->wh_blend = true;
    }
    if (num_textures) {
        std::vector<WhTexture> table(num_textures);
        for (uint32_t i = 0; i < num_textures; ++i) {
            const MiptTextureData& t = textures[i];
            WhTexture& d = table[i];
            d = WhTexture{};
            d.address_s = t.address_s; d.address_t = t.address_t; d.filter = t.filter;
            void* dev = nullptr;
            if (t.texels && t.width && t.height) {
                const size_t bytes = (size_t)t.width * t.height * 4u * (t.bits / 8u);
                HIPCHK(ctx, hipMalloc(&dev, bytes));
                ctx->wh_texels.push_back(dev);
                HIPCHK(ctx, hipMemcpy(dev, t.texels, bytes, hipMemcpyHostToDevice));
                d.width = t.width; d.height = t.height; d.bits = t.bits; d.texels = dev;
            }
        }
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_wh_textures, (size_t)num_textures * sizeof(WhTexture)));
        HIPCHK(ctx, hipMemcpy(ctx->d_wh_textures, table.data(), (size_t)num_textures * sizeof(WhTexture), hipMemcpyHostToDevice));
        ctx->n_wh_textures = num_textures;
    }
    return MIPT_SUCCESS;
}

static WhTables typed_tables(const MiptWhittedGeometry& geo)
{
    WhTables g{};
    g.vertices4 = (const float4*)geo.vertices4; g.normals4 = (const float4*)geo.normals4; g.material_index = geo.material_index;
    for (int j = 0; j < 2; ++j) { g.texcoords[j] = (const float2*)geo.texcoords[j]; g.texcoord_present[j] = geo.texcoord_present[j]; }
    g.colors4 = (const float4*)geo.colors4; g.color_present = geo.color_present;
    return g;
}

int mipt_whitted_set_geometry(MiptContext* ctx, const MiptWhittedGeometry* geo, const MiptMaterialData* materials, uint32_t num_materials,
                              const MiptTextureData* textures, uint32_t num_textures)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!geo || !geo->vertices4) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "whitted_set_geometry: vertices required");
    const int rc = set_whitted_tables(ctx, materials, num_materials, textures, num_textures);
    if (rc) return rc;
    WhGeometry g{};
    static_cast<WhTables&>(g) = typed_tables(*geo);
    ctx->wh_geo = g;
    return MIPT_SUCCESS;
}

// Scene::createSBT's shading data (Scene.cpp:1368-1480): per-mesh object-space buffers + one record per instance
int mipt_whitted_set_geometry_instanced(MiptContext* ctx, const MiptWhittedMesh* meshes, uint32_t num_meshes, const MiptWhittedInstance* instances,
                                        uint32_t num_instances, const MiptMaterialData* materials, uint32_t num_materials,
                                        const MiptTextureData* textures, uint32_t num_textures)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!meshes || num_meshes == 0 || !instances || num_instances == 0) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "whitted_set_geometry_instanced: meshes and instances required");
    std::vector<WhTables> tabs(num_meshes);
    for (uint32_t m = 0; m < num_meshes; ++m) {
        if (meshes[m].num_triangles && !meshes[m].geometry.vertices4) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "mesh %u: vertices required", m);
        tabs[m] = typed_tables(meshes[m].geometry);
    }
    std::vector<WhInstGeo> recs(num_instances);
    std::vector<uint32_t> prim_inst;
    uint64_t total = 0;
    for (uint32_t i = 0; i < num_instances; ++i) {
        if (instances[i].mesh >= num_meshes) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instance %u: mesh %u of %u", i, instances[i].mesh, num_meshes);
        WhInstGeo& r = recs[i];
        std::memcpy(r.m, instances[i].transform, 48);
        const float* m = r.m;           // (rows of 4: m[4 r + c]; the upper 3x3's cofactors exactly as mipt_scene_flatten_normals forms them, scene.cpp)
        r.cof[0] = m[5] * m[10] - m[6] * m[9]; r.cof[1] = m[6] * m[8] - m[4] * m[10]; r.cof[2] = m[4] * m[9] - m[5] * m[8];
        r.cof[3] = m[2] * m[9] - m[1] * m[10]; r.cof[4] = m[0] * m[10] - m[2] * m[8]; r.cof[5] = m[1] * m[8] - m[0] * m[9];
        r.cof[6] = m[1] * m[6] - m[2] * m[5];  r.cof[7] = m[2] * m[4] - m[0] * m[6];  r.cof[8] = m[0] * m[5] - m[1] * m[4];
        const float det = (m[0] * r.cof[0] + m[1] * r.cof[1]) + m[2] * r.cof[2];
        r.inv_det = det != 0.0f ? 1.0f / det : 0.0f;
        r.mesh = instances[i].mesh; r.prim_base = (uint32_t)total;
        total += meshes[instances[i].mesh].num_triangles;
        if (total >= (1ull << 31)) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instances reference more than 2^31 triangles");
        prim_inst.resize((size_t)total, i);
    }
    const int rc = set_whitted_tables(ctx, materials, num_materials, textures, num_textures);
    if (rc) return rc;
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_wh_meshes, (size_t)num_meshes * sizeof(WhTables)));
    HIPCHK(ctx, hipMemcpy(ctx->d_wh_meshes, tabs.data(), (size_t)num_meshes * sizeof(WhTables), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_wh_inst, (size_t)num_instances * sizeof(WhInstGeo)));
    HIPCHK(ctx, hipMemcpy(ctx->d_wh_inst, recs.data(), (size_t)num_instances * sizeof(WhInstGeo), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_wh_prim_inst, std::max<size_t>(prim_inst.size(), 1) * 4u));
    if (!prim_inst.empty()) HIPCHK(ctx, hipMemcpy(ctx->d_wh_prim_inst, prim_inst.data(), prim_inst.size() * 4u, hipMemcpyHostToDevice));
    WhGeometry g{};
    g.prim_inst = ctx->d_wh_prim_inst; g.inst = ctx->d_wh_inst; g.meshes = ctx->d_wh_meshes;
    ctx->wh_geo = g;
    ctx->wh_inst_base.resize(num_instances);
    for (uint32_t i = 0; i < num_instances; ++i) ctx->wh_inst_base[i] = recs[i].prim_base;
    ctx->wh_inst_total = total;
    return MIPT_SUCCESS;
}

int mipt_whitted_launch(MiptContext* ctx, void* stream, const MiptWhittedLaunchParams* p, unsigned int width, unsigned int height, unsigned int depth)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    if (!p) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "params is null");
    if (depth != 1 || width == 0 || height == 0 || width != p->width || height != p->height)
        return ctx->fail(MIPT_ERROR_INVALID_VALUE, "launch size %ux%ux%u does not match params %ux%u", width, height, depth, p->width, p->height);
    if (!p->accum_buffer) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "params.accum_buffer is null");
    if (p->lights.count > 31 || (p->lights.count && !p->lights.data)) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "0..31 lights supported");
    if (p->lights.count && p->lights.byte_stride != 0 && p->lights.byte_stride != sizeof(MiptLight))
        return ctx->fail(MIPT_ERROR_INVALID_VALUE, "lights must be tightly packed (BufferView stride %u)", (unsigned)p->lights.byte_stride);
    if ((!ctx->wh_geo.vertices4 && !ctx->wh_geo.prim_inst) || !ctx->d_wh_materials) return ctx->fail(MIPT_ERROR_NOT_CONFIGURED, "mipt_whitted_set_geometry was not called");
    Accel* acc = find_accel(ctx, p->handle);
    if (!acc) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "params.handle 0x%llx is not a traversable of this context", p->handle);
    if (acc->stale) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instance structure 0x%llx references a mesh that was refitted since it was built: call mipt_accel_update_instances first", p->handle);
    if (ctx->wh_geo.prim_inst) {
        // per-instance shading tables are indexed by the hits' flattened primitive numbers: the tables and the structure must count the
        // same triangles, and over an instance structure every instance must start at the same number (same instances, same order)
        if (ctx->wh_inst_total != acc->n_prims)
            return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instanced shading tables hold %llu triangles, traversable 0x%llx has %u", (unsigned long long)ctx->wh_inst_total, p->handle, acc->n_prims);
        if (acc->is_ias && acc->h_base != ctx->wh_inst_base)
            return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instanced shading tables do not follow the instance order of traversable 0x%llx (instances / triangles per instance differ)", p->handle);
    }
    if (const int frc = check_device_faults(ctx)) return frc;         // raised by an earlier launch, reported once
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint32_t n_paths = ((width + 7u) / 8u) * ((height + 7u) / 8u) * 64u;
    int rc = ensure_capacity(ctx, n_paths, (size_t)width * height);
    if (rc) return rc;
    rc = ensure_group_capacity(ctx, (size_t)n_paths * (size_t)(p->lights.count + 1));
    if (rc) return rc;
    if (ctx->wh_group_cap < n_paths) {
        HIPCHK(ctx, hipDeviceSynchronize());
        if (ctx->wh_group) (void)hipFree(ctx->wh_group);
        ctx->wh_group = nullptr; ctx->wh_group_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->wh_group, (size_t)n_paths * sizeof(float4)));
        ctx->wh_group_cap = n_paths;
    }
    const size_t level_words = (size_t)width * height * 8u;
    if (ctx->wh_blend && ctx->wh_levels_cap < level_words) {
        HIPCHK(ctx, hipDeviceSynchronize());
        if (ctx->wh_levels) (void)hipFree(ctx->wh_levels);
        ctx->wh_levels = nullptr; ctx->wh_levels_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->wh_levels, level_words * sizeof(float4)));
        ctx->wh_levels_cap = level_words;
    }
    hipStream_t st = (hipStream_t)stream;
    ctx->last_stream = st;
    ctx->advance_pending = false; ctx->bins_live = false;
    HIPCHK(ctx, hipMemsetAsync(&ctx->d_ctr->closest_rays, 0, (14 + kStatBins * kStatBinWords) * sizeof(unsigned long long), st));
    WhBuffers B{};
    B.q[0] = ctx->q[0]; B.q[1] = ctx->q[1]; B.hits = ctx->hits; B.sh = ctx->im_sq; B.e1 = ctx->wh_e1; B.flags = ctx->im_flags;
    B.group = ctx->wh_group; B.n_levels = ctx->rg_seed;
    B.max_levels = ctx->wh_blend ? 8u : 1u;                     // whitted::MAX_TRACE_DEPTH
    B.levels = ctx->wh_blend ? ctx->wh_levels : ctx->result;
    B.any_hit_programs = ctx->wh_any_hit;
    B.live_ring = ctx->h_pinned; B.ring_events = reinterpret_cast<void* const*>(ctx->ring_ev);
    whitted_run(st, *p, n_paths, device_bvh(acc, ctx->wh_any_hit ? nullptr : ctx), ctx->wh_geo, ctx->d_wh_materials, ctx->n_wh_materials, ctx->d_wh_textures,
                ctx->n_wh_textures, B, ctx->d_ctr, trace_opts(ctx));
    ctx->last_bounces = B.max_levels; ctx->bounces_on_device = false;
    HIPCHK(ctx, hipGetLastError());
    return MIPT_SUCCESS;
}

void mipt_imgui_camera_set_fd(MiptImguiCamera* c, float new_fd)
{
    const f3 d = mk3(c->lookat[0], c->lookat[1], c->lookat[2]) - mk3(c->eye[0], c->eye[1], c->eye[2]);
    c->fd = new_fd / length(d);                                         // camera.h:36
}

void mipt_imgui_camera_compute_uvw(MiptImguiCamera* c)                 // camera.h:110-123
{
    f3 w = (mk3(c->lookat[0], c->lookat[1], c->lookat[2]) - mk3(c->eye[0], c->eye[1], c->eye[2])) * c->fd;
    const float wlen = length(w);
    f3 u = normalize(cross(w, mk3(c->up[0], c->up[1], c->up[2])));
    f3 v = normalize(cross(u, w));
    const float vlen = wlen * tanf(0.5f * c->fov * 3.14159265358979323846f / 180.0f);
    v = v * vlen;
    u = u * vlen;
    c->u[0] = u.x; c->u[1] = u.y; c->u[2] = u.z;
    c->v[0] = v.x; c->v[1] = v.y; c->v[2] = v.z;
    c->w[0] = w.x; c->w[1] = w.y; c->w[2] = w.z;
}

// ------------------------------------------------------------------------------------------------
int mipt_trace_closest(MiptContext* ctx, void* stream, unsigned long long handle, const void* rays, uint32_t n, void* hits)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    const Accel* a = find_accel(ctx, handle);
    if (!a) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "unknown traversable 0x%llx", handle);
    if (a->stale) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instance structure 0x%llx references a mesh that was refitted since it was built: call mipt_accel_update_instances first", handle);
    if (n == 0) return MIPT_SUCCESS;
    if (!rays || !hits) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "trace_closest: null buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    ctx->last_stream = st;
    ctx->advance_pending = false; ctx->bins_live = false;
    HIPCHK(ctx, hipMemsetAsync(&ctx->d_ctr->closest_rays, 0, (14 + kStatBins * kStatBinWords) * sizeof(unsigned long long), st));
    launch_begin_sample(st, ctx->d_ctr, n, n);
#if MIPT_PROBE_WAVES
    if (const char* e = std::getenv("MIPT_PROBE_WAVES")) {          // probe builds only: address of a caller-allocated device buffer for per-wave records
        const unsigned long long addr = std::strtoull(e, nullptr, 0);
        HIPCHK(ctx, hipMemcpyAsync(&ctx->d_ctr->dbg[0], &addr, 8, hipMemcpyHostToDevice, st));
    }
#endif
    const float4* r = (const float4*)rays;
    { TimedScope t(ctx, st, T_CLOSEST);
      launch_trace_closest(st, device_bvh(a, ctx), r, r + 1, 2, nullptr, n, (float4*)hits, ctx->d_ctr, trace_opts(ctx)); }
    HIPCHK(ctx, hipGetLastError());
    return MIPT_SUCCESS;
}

int mipt_trace_any(MiptContext* ctx, void* stream, unsigned long long handle, const void* rays, uint32_t n, void* occluded)
{
    if (!ctx) return MIPT_ERROR_INVALID_VALUE;
    const Accel* a = find_accel(ctx, handle);
    if (!a) return ctx->fail(MIPT_ERROR_INVALID_HANDLE, "unknown traversable 0x%llx", handle);
    if (a->stale) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "instance structure 0x%llx references a mesh that was refitted since it was built: call mipt_accel_update_instances first", handle);
    if (n == 0) return MIPT_SUCCESS;
    if (!rays || !occluded) return ctx->fail(MIPT_ERROR_INVALID_VALUE, "trace_any: null buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    ctx->last_stream = st;
    ctx->advance_pending = false; ctx->bins_live = false;
    HIPCHK(ctx, hipMemsetAsync(&ctx->d_ctr->closest_rays, 0, (14 + kStatBins * kStatBinWords) * sizeof(unsigned long long), st));
    launch_begin_sample(st, ctx->d_ctr, n, 0);
#if MIPT_PROBE_WAVES
    if (const char* e = std::getenv("MIPT_PROBE_WAVES_ANY")) {      // probe builds only: per-wave records of k_trace_static (tools/probe_shadow_order.py)
        const unsigned long long addr = std::strtoull(e, nullptr, 0);
        HIPCHK(ctx, hipMemcpyAsync(&ctx->d_ctr->dbg[1], &addr, 8, hipMemcpyHostToDevice, st));
    }
#endif
    const float4* r = (const float4*)rays;
    { TimedScope t(ctx, st, T_ANY);
      launch_trace_any(st, device_bvh(a, ctx), r, r + 1, 2, nullptr, n, (uint32_t*)occluded, nullptr, nullptr, ctx->d_ctr, trace_opts(ctx)); }
    HIPCHK(ctx, hipGetLastError());
    return MIPT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------
void mipt_camera_uvw(const float eye[3], const float lookat[3], const float up[3], float fovY_deg, float aspect, float fod,
                     float U[3], float V[3], float W[3])
{
    // sutil::Camera::UVWFrame, SDK/sutil/Camera.cpp:34-46
    f3 w = (mk3(lookat[0], lookat[1], lookat[2]) - mk3(eye[0], eye[1], eye[2])) * fod;
    const float wlen = length(w);
    f3 u = normalize(cross(w, mk3(up[0], up[1], up[2])));
    f3 v = normalize(cross(u, w));
    const float vlen = wlen * tanf(0.5f * fovY_deg * 3.14159265358979323846f / 180.0f);
    v = v * vlen;
    const float ulen = vlen * aspect;
    u = u * ulen;
    U[0] = u.x; U[1] = u.y; U[2] = u.z;
    V[0] = v.x; V[1] = v.y; V[2] = v.z;
    W[0] = w.x; W[1] = w.y; W[2] = w.z;
}

int mipt_scene_random_triangles(uint32_t n, uint32_t seed_tag, float edge_scale, float* out)
{
    if (!out && n) return MIPT_ERROR_INVALID_VALUE;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t seed = tea4(i, seed_tag);
        float c[3], e1[3], e2[3];
        for (int k = 0; k < 3; ++k) c[k] = rnd(seed);
        for (int k = 0; k < 3; ++k) e1[k] = (rnd(seed) - 0.5f) * 2.0f * edge_scale;
        for (int k = 0; k < 3; ++k) e2[k] = (rnd(seed) - 0.5f) * 2.0f * edge_scale;
        float* v = out + (size_t)i * 12;
        for (int k = 0; k < 3; ++k) { v[k] = c[k]; v[4 + k] = c[k] + e1[k]; v[8 + k] = c[k] + e2[k]; }
        v[3] = 0.0f; v[7] = 0.0f; v[11] = 0.0f;
    }
    return MIPT_SUCCESS;
}

}  // extern "C"
