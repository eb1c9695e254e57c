// kernels.h -- host-visible launch interface of the HIP kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mipt.h"

namespace mipt {

constexpr int      kStackEntries  = 32;     // traversal stack entries per lane == builder depth cap
constexpr int      kStreamStack = 32;       // k_trace_stream: whole stack in LDS (8 KB + 2 KB merge slots per wave = 16 waves per CU)
constexpr uint32_t kInvalidSlot   = 0xffffffffu;
constexpr int      kNumSegments   = 8;      // one ray-range segment per XCD
constexpr int      kCursorStride  = 32;     // uint32 words between cursors (own 128-B line each)
constexpr uint32_t kStreamChunk   = 64;     // rays a wave takes from a segment cursor at a time (one 8x8 tile)
constexpr uint32_t kShadeBlock    = 512;    // threads per shade workgroup == paths per queue-append atomic

// device-resident queue counters + statistics (one per context)
struct DeviceCounters {
    uint32_t n_cur;            // live paths entering this bounce
    uint32_t n_next;           // continuation rays emitted by shade
    uint32_t n_shadow;         // shadow rays emitted by shade
    uint32_t bounce;
    uint32_t cursor_closest[kNumSegments * kCursorStride];
    uint32_t cursor_shadow[kNumSegments * kCursorStride];
    unsigned long long closest_rays, shadow_rays;
    unsigned long long node_visits[2], tri_tests[2];   // [0] closest-hit kernel, [1] any-hit kernel
    unsigned long long iterations;                      // wavefront bounces that had live paths (+ the tails of finished passes)
    unsigned long long tail_max;                        // deepest path of the current pass's k_tail launch (bounces it ran)
    unsigned long long dbg[6];
    // what the waves of a path kernel add when they end, spread over kStatBins cache lines (bin = workgroup % kStatBins; words 0 / 1 / 2
    // of a bin: continuation rays, shadow rays, most bounces): atomics on ONE address take ~8 ns each whether they return or not, and
    // 8 000 waves ending within the same 0.2 ms queued 25 us of them behind a rank's 1/8 shard.  Read as sums / a maximum (tail_stats_*).
    unsigned long long bins[64 * 16];
    uint32_t epoch;            // path kernels that do their own begin_sample (TailArgs): the launch whose first workgroup is through it
    uint32_t n_open;           // Whitted any-hit programs, seeded radiance launches: rays whose closest candidate the program ignored (k_wh_seed)
    uint32_t* fault_host;      // pinned host word (MiptContext::h_pinned + kFaultWord): a kernel that gave up on a bounded wait ORs its
                               // kFault* bit into it; mipt_sync / mipt_get_launch_stats / the next launch turn it into an error
};
constexpr uint32_t kStatBins = 64, kStatBinWords = 16;      // DeviceCounters::bins
constexpr uint32_t kFaultWord = 8;              // index into MiptContext::h_pinned (words 0 .. 3: run_wavefront's ring of live counts)
constexpr uint32_t kFaultEpochTimeout = 1u;     // tail_await_epoch ran out: the statistics race it guards against may have happened

struct DeviceBvh {
    const float4* nodes;       // 4 x float4 per MiptBvhNode
    const float4* tris;        // 3 x float4 per MiptBvhTri
    const uint32_t* mat_index; // per ORIGINAL primitive: hit-group record index
    uint32_t tri_byte_off;     // tris == (char*)nodes + tri_byte_off (one allocation); 0 if that offset needs > 32 bits
    uint32_t blob_bytes;       // size of that allocation (buffer-descriptor range)
    uint32_t depth;            // deepest leaf (stack entries a ray can need)
    uint32_t pool2_ok;         // ... and k_tail_pool2's (7-bit slots): < 2^24 nodes, < 2^22 triangles
    uint32_t pool_ok;          // fits the entry pool's packed words: < 2^25 nodes, 2 .. 2^23 - 1 triangles, leaves of 1 .. 4
    // two-level (instance) structure: `nodes` is then the BVH over the instance boxes, its leaves list positions of inst_order
    const struct DeviceInstance* inst;      // null: an ordinary triangle BVH
    const uint32_t* inst_order;
    uint32_t n_inst;
    uint32_t node32_off;       // byte offset, in the same allocation, of the entry-pool kernels' node mirror; 0 = this BVH has none
    uint32_t node4_off;        // byte offset of the BVH4 mirror (64-byte nodes of four fp16 boxes) behind the packed one; 0 = none
    uint32_t stack4;           // the most entries a lane's stack can need over the BVH4 mirror
    // two-level structures on the entry pool (instances.hip, k_trace_static_inst / the path kernel's two-level form): `nodes` is then
    // the base of ONE allocation -- top-level fp32 nodes | DeviceInstance records | BVH4 nodes (the top level's mirror first, then
    // every referenced mesh's mirror with its references rebased into this array) | the meshes' triangles -- walked through one
    // buffer descriptor.  inst_fast = 0: only the one-ray-per-lane walk (k_trace_inst) can traverse the structure.
    uint32_t inst_fast, inst_off, inst_node4_off, inst_tri_off, inst_blob_bytes;
    uint32_t inst_root4_off;   // per top-level BVH4 node 4 words: the BVH4 root of the mesh behind each LEAF child (an instance)
    const uint32_t* prim_slot; // one-level: per ORIGINAL primitive the position of its triangle record (null: not kept by the builder that made this BVH)
    uint32_t inst_slots;       // two-level: every DeviceInstance::prim_slot is set
    uint32_t top_off;          // experiment MIPT_TOP_CACHE: byte offset of the top-of-tree copy of the packed mirror (64 nodes + their real indices); 0 = none
    const float* xf;           // not null: this is the mesh of an instance structure of ONE instance, walked as an ordinary BVH -- every ray is taken into
                               // the instance's object space when its traversal starts (trav_init; the 12 floats of DeviceInstance::inv, device memory)
    uint32_t node_fmt;         // 0: fp32 64-byte BVH2 nodes only; 1: + packed 32-byte BVH2 nodes at node32_off (MiptBvhNode32: the boxes of
                               // `nodes` as fp16 -- the two arrays hold the SAME values), walked by every entry-pool kernel; 2: + the BVH4
                               // mirror at node4_off, walked by the kernels that live on the pool alone (static chunks, path kernels)
};

// one instance of a two-level structure (built by mipt_accel_build_instances from an OptixInstance-compatible record)
struct DeviceInstance {
    float         inv[12];     // world -> object, row-major 3x4
    const float4* nodes;       // the mesh's BVH (64-byte nodes) ...
    const float4* tris;        // ... and triangles (48 bytes each), object space
    uint32_t      prim_base;   // flattened index of the mesh's primitive 0 within this instance
    uint32_t      root4;       // entry-pool form: index of the mesh's BVH4 root in the structure's BVH4 array (DeviceBvh::inst_node4_off)
    const uint32_t* prim_slot; // the mesh's table: mesh-local primitive -> position of its triangle record (null: not kept)
};

static_assert(sizeof(DeviceInstance) == 80, "the pool kernels fetch an instance record in 16-byte pieces");

// SoA path queues, capacity `cap` entries each
struct PathQueue {
    float4* ray_o;             // origin.xyz, tmin
    float4* ray_d;             // direction.xyz, tmax
    float4* state_a;           // attenuation.xyz, seed (bits)
    uint2*  state_b;           // result slot, depth
};

struct ShadowQueue {
    float4* ray_o;
    float4* ray_d;
    float4* contrib;           // radiance*attenuation .xyz, result slot (bits)
};

struct FrameConsts {           // the by-value part of Params (optixPathTracer.h:91-107)
    uint32_t width, height, subframe_index, samples_per_launch;
    float eye[3], U[3], V[3], W[3];
    MiptParallelogramLight light;
    float bg[4];
    uint32_t max_depth;
    // sharding: world == 0 -> full frame (slot = pixel index, path order by pixel tiles of 64: tile_shift)
    uint32_t world, rank, n_paths;   // n_paths: padded path count of one wavefront pass
    // batched samples (MIPT_OPT_BATCH_SAMPLES): all samples_per_launch samples are paths of ONE pass
    uint32_t batch_spp;              // 0 = one pass per sample (reference accumulation order), else samples_per_launch
    uint32_t paths_per_sample;       // padded paths of one sample
    uint32_t n_slots;                // accumulator slots of one sample
    uint32_t tile_shift;             // full frames: a tile of 64 paths covers (1 << tile_shift) x (64 >> tile_shift) pixels
};

// on != 0: the launch that ENDS a path writes its pixel (k_resolve's arithmetic for one slot: resolve_pixel) -- launches of one sample
// per pixel over a full frame, where every pixel has exactly one path; k_resolve is then not launched.  A path ends in the shade
// launch (no continuation ray; if it casts a last occlusion ray, that ray's queue entry carries bit 31 of the slot and the any-hit
// launch writes the pixel) or in a path kernel.
struct ResolveJob {
    float4*   accum;
    uint32_t* frame;           // may be null
    uint32_t  subframe_index;
    uint32_t  on;
};

struct RaygenJob {             // the arguments of k_raygen
    FrameConsts     fc;
    uint32_t        sample;
    PathQueue       q;
    uint32_t*       rg_seed;
    float4*         result;
    DeviceCounters* ctr;
    uint32_t        n_valid, zero_stats;
};

struct TraceOpts {
    int  persistent;           // 3: tiles + entry pool (default), 2: streaming waves, 0: one thread per ray
    int  refill_min, steal_min, burst_bulk, burst_dry; // k_trace_stream knobs
    int  tile_group;           // k_trace_pool, tiled: consecutive tiles dealt to one XCD at a time (0: a contiguous eighth each)
    int  count_visits;
    int  blocks;               // persistent grid size (workgroups of 64)
    int  blocks_user, num_cus; // the grid was set with MIPT_OPT_TRACE_GRID; CUs of the device
    const uint32_t* tile_order; uint8_t* tile_cost; uint32_t* tile_hist; uint32_t n_order, class_width;   // tiled closest-hit launches: TraceArgs
    const uint32_t* class_width_dev;
    int  tail_refill;          // k_tail_pool2 as one set of resident waves that refill ended paths from the queue: 0 never, 1 big trees, 2 always
    int  tiled_bvh4;           // experiments: the tiled kernel walks the BVH4 mirror too (when a lane's stack over it fits 32 rows)
    int  inst_form;            // two-level structures (instances.hip): 0 by launch size, 1 the pool-only form, 2 one ray per lane (env MIPT_INST_FORM)
};

// The next primary launch's tile order, made inside the launches the pass has anyway: the tiled kernel leaves a class byte per
// tile and per-chunk class histograms (TraceArgs), the first `n_chunks` workgroups of the shade launch turn them into the order
// (stable counting sort: heaviest class first, image order inside a class) and clear the histograms of the launch after.
constexpr uint32_t kOrderChunk = 512, kOrderClasses = 16;
struct TileOrderJob {
    const uint8_t*  cls;      // [n_tiles]
    const uint32_t* hist;     // [n_chunks][kOrderClasses]
    uint32_t*       zero;     // [n_chunks][kOrderClasses] the histograms of the NEXT tiled launch, cleared here (the two sets alternate)
    uint32_t*       width;    // null, or the device word the tiled kernel takes its class width from: adapted here to the frame's costs
    uint32_t*       order;    // [n_tiles] out: slot -> tile
    uint32_t        n_tiles, n_chunks;
};

// ---- launches (all asynchronous on `st`) ----
void launch_begin_sample(hipStream_t st, DeviceCounters* ctr, uint32_t n_paths, uint32_t n_valid);
void launch_tail(hipStream_t st, const FrameConsts& fc, const DeviceBvh& bvh, const MiptHitGroupData* hitgroups, uint32_t n_hitgroups,
                 PathQueue cur, float4* result, DeviceCounters* ctr, uint32_t n_max, const void* common_vertices, const struct TraceOpts& o,
                 const struct ResolveJob& resolve, const struct RaygenJob* rg = nullptr, uint32_t epoch = 0u);
// a one-bounce pass (depth cap 1) whose closest-hit launch would take the tiled form, as ONE launch: tile waves that make their paths,
// trace them, shade them, trace their occlusion rays and write their pixels (k_frame1) + the next launch's tile order (k_tile_order).
// false: nothing was launched (the pass does not qualify), the caller runs the wavefront form
bool launch_frame1(hipStream_t st, const FrameConsts& fc, const DeviceBvh& bvh, const MiptHitGroupData* hitgroups, uint32_t n_hitgroups,
                   float4* result, DeviceCounters* ctr, uint32_t n_paths, const void* common_vertices, const struct TraceOpts& o,
                   const struct ResolveJob& resolve, const struct RaygenJob& rg, uint32_t epoch, const struct TileOrderJob& order,
                   const ShadowQueue* defer_to = nullptr);      // not null: the occlusion rays go to this queue (ctr->n_shadow), for the any-hit launch of the wavefront form
// rg is allowed (the launch makes the pass's primary paths itself: no k_raygen in front of it) when the pool forms of the path kernel run
bool tail_launch_makes_rays(const DeviceBvh& bvh, const struct TraceOpts& o);
void launch_advance(hipStream_t st, DeviceCounters* ctr, uint32_t* host_live = nullptr);   // host_live: pinned word that receives the new live count
// ctr != null: the kernel also does launch_begin_sample's work for the pass (and clears the launch statistics when zero_stats)
void launch_raygen(hipStream_t st, const FrameConsts& fc, uint32_t sample, PathQueue q, uint32_t* rg_seed, float4* result,
                   DeviceCounters* ctr = nullptr, uint32_t n_valid = 0, bool zero_stats = false,
                   uint32_t* zero_words = nullptr, uint32_t n_zero = 0);     // words the launch clears
// rg != null (only where closest_launch_is_tiled()): the launch makes its rays itself -- k_raygen's work for the pass, folded into
// the kernel that traces them (no launch of its own, the rays are not read back)
void launch_trace_closest(hipStream_t st, const DeviceBvh& bvh, const float4* ray_o, const float4* ray_d, uint32_t stride,
                          const uint32_t* n_ptr, uint32_t n_max, float4* hits, DeviceCounters* ctr, const TraceOpts& o,
                          uint32_t ray_flags = 0u /* MiptRayFlags */, const struct RaygenJob* rg = nullptr);
// the closest-hit launch of n_max rays takes the tiled form (k_trace_pool), not k_trace_static / k_trace_stream / k_trace_simple
bool closest_launch_is_tiled(const DeviceBvh& bvh, uint32_t n_max, const TraceOpts& o);
// mode 1: occluded flags (uint32 per ray); mode 2: accumulate contrib into result when unoccluded
void launch_trace_any(hipStream_t st, const DeviceBvh& bvh, const float4* ray_o, const float4* ray_d, uint32_t stride,
                      const uint32_t* n_ptr, uint32_t n_max, uint32_t* occluded, const float4* contrib, float4* result,
                      DeviceCounters* ctr, const TraceOpts& o, const ResolveJob& resolve = ResolveJob{});
void launch_shade(hipStream_t st, const FrameConsts& fc, const DeviceBvh& bvh, const MiptHitGroupData* hitgroups,
                  uint32_t n_hitgroups, PathQueue cur, const float4* hits, PathQueue next, ShadowQueue shadow,
                  float4* result, DeviceCounters* ctr, uint32_t n_max, const void* common_vertices, const TileOrderJob& order = TileOrderJob{},
                  uint32_t primary = 0u,     // ShadeArgs::primary: the pass's rays were made by the closest-hit launch (RaygenJob)
                  const ResolveJob& resolve = ResolveJob{});
void launch_resolve(hipStream_t st, const FrameConsts& fc, const float4* result, float4* accum, uint32_t* frame, uint32_t n_slots,
                    DeviceCounters* advance_ctr = nullptr);   // advance_ctr: also do the last bounce's launch_advance
// SDK/imgui_test pipeline (imgui_kernels.hip)
void imgui_run(hipStream_t st, const MiptImguiParams& p, uint32_t n_paths, const DeviceBvh& bvh, PathQueue q, float4* hits,
               ShadowQueue sh, uint32_t* occluded, float4* result, uint32_t* rg_seed, DeviceCounters* ctr, const TraceOpts& opts);
// two-level structures: a mesh's BVH4 mirror copied into the structure's array with inner references moved by node_base and leaf
// descriptors by tri_base; the top level's leaf descriptors turned from positions of inst_order into instance indices
void launch_rebase_nodes4(hipStream_t st, const void* src, void* dst, uint32_t n4, uint32_t node_base, uint32_t tri_base);
void launch_tlas4_leaves(hipStream_t st, void* nodes4, uint32_t n4, const uint32_t* order, const struct DeviceInstance* inst, uint32_t* root4_out);
void launch_build_top(hipStream_t st, const void* d_nodes32, uint32_t nn, void* d_top);      // experiment MIPT_TOP_CACHE (kernels.hip)
// device-side build-input assembly + BVH build (lbvh_build.hip); every pointer of GatherInput is a DEVICE pointer
struct GatherInput {
    const void* vertices; uint32_t num_vertices, vertex_stride;
    const void* indices; uint32_t index_format, index_stride, num_triangles;
    const uint32_t* sbt_index_offsets; const uint32_t* flags; uint32_t num_sbt_records;
    uint32_t has_transform; float m[12];
};
void launch_gather_input(hipStream_t st, const GatherInput& in, float* world, uint32_t* mats, uint32_t* gflags, uint32_t base, uint32_t* error);
// *ok = false when the tree is deeper than depth_cap
// refit of an existing BVH2 (same topology) to new world-space vertices (9 floats per original primitive)
// Node format 1 (MIPT_OPT_NODE_FORMAT): rounds every child box of the 64-byte nodes OUTWARD to fp16-representable values, in
// place, and writes the packed 32-byte mirror (MiptBvhNode32) the entry-pool kernels fetch.  *ok = false (nothing changed)
// when the scene does not suit fp16 boxes: a coordinate beyond +-32768, or an extent below a quarter of the largest one.
hipError_t pack_nodes32(hipStream_t st, float4* d_nodes, uint32_t nn, void* d_nodes32, bool* ok);
// Node format 2: the BVH4 mirror (64-byte nodes of four fp16 boxes) of the packed BVH2 mirror, made on the device (lbvh_build.hip)
// (scratch: device memory of at least *_scratch_bytes() for the call's temporaries -- the context's arena, so that a refit allocates
//  nothing in its steady state; null: the function allocates and frees its own)
// height of every node (1 = both children are leaves; the root's = the tree's depth): topology only, computed once per build
size_t topo_heights_scratch_bytes(uint32_t nn);
hipError_t topo_heights(hipStream_t st, const float4* d_nodes, uint32_t nn, uint8_t* d_height, void* scratch = nullptr, size_t scratch_bytes = 0);
size_t collapse_bvh4_scratch_bytes(uint32_t nn);
hipError_t collapse_bvh4(hipStream_t st, const void* d_nodes32, const uint8_t* d_height, uint32_t nn, uint32_t depth, void* d_nodes4, uint32_t* n4,
                         uint32_t* stack_need, bool* ok, void* scratch = nullptr, size_t scratch_bytes = 0,
                         uint32_t budget = 32u);       // budget: the most stack entries a lane may need while walking the mirror
hipError_t lbvh_refit(hipStream_t st, const float* d_world, uint32_t n, float4* d_nodes, uint32_t nn, float4* d_tris, const uint8_t* d_height, uint32_t depth,
                      const float* d_cells = nullptr);        // n: triangle RECORDS; d_cells: 6 floats per record of a tree built with MIPT_OPT_BUILD = 4, or null
hipError_t lbvh_build(hipStream_t st, const float* d_world, const uint32_t* d_gflags, uint32_t n, int leaf_max, int depth_cap,
                      float4** d_blob, uint32_t* num_nodes, uint32_t* depth, bool* ok, bool sah_treelets = true,
                      uint32_t* d_prim_slot = nullptr,        // d_prim_slot: n words, filled with primitive -> position of its triangle record
                      bool sah_top = false,                   // the levels above the treelets by top-down binned SAH instead of Morton codes (MIPT_OPT_BUILD = 2)
                      float spatial_extra = 0.0f, int ret_below = 256, uint32_t* num_records = nullptr, float** d_cells = nullptr);      // MIPT_OPT_BUILD = 4: triangles cut at split planes (lbvh_build.hip: struct SpSplit)

// SDK/cuda/whitted.cu pipeline (whitted_kernels.hip)
struct WhTables {                  // MiptWhittedGeometry with typed pointers: the shading tables of a flattened scene, or of one mesh
    const float4* vertices4; const float4* normals4; const uint32_t* material_index;
    const float2* texcoords[2]; const uint8_t* texcoord_present[2];
    const float4* colors4; const uint8_t* color_present;
};
struct WhInstGeo {                 // one instance of mipt_whitted_set_geometry_instanced
    float    m[12];                // object -> world, row-major 3x4
    float    cof[9];               // cofactors of its upper 3x3 (= det * inverse transpose) ...
    float    inv_det;              // ... and 1 / det (0 when singular): mipt_scene_flatten_normals' arithmetic
    uint32_t mesh, prim_base;      // its mesh; flattened index of its primitive 0
};
struct WhGeometry : WhTables {     // prim_inst == null: the flattened tables above, indexed by the flattened primitive
    const uint32_t*  prim_inst;    // per flattened primitive: its instance
    const WhInstGeo* inst;
    const WhTables*  meshes;       // per-mesh OBJECT-space tables, indexed by the mesh-local primitive
};
struct WhTexture {                 // MiptTextureData with the texels in device memory
    uint32_t width, height, bits, address_s, address_t, filter;
    const void* texels;
};
struct WhBuffers {                 // launch storage owned by the context
    PathQueue q[2];                // rays of the current / next level
    float4* hits;
    ShadowQueue sh;                // K entries per hit: occlusion ray + E0
    float4* e1;
    uint32_t* flags;               // occlusion result per entry: uint32 flag (fast kernels) or float attenuation (any-hit programs)
    float4* group;
    float4* levels;                // [max_levels][pixels]
    uint32_t* n_levels;            // per pixel
    uint32_t max_levels;           // 1 without BLEND materials, MAX_TRACE_DEPTH with
    bool any_hit_programs;         // a MASK / BLEND material exists: traversal runs the any-hit programs
    uint32_t* live_ring;           // pinned host words (4): the number of rays a level leaves for the next one, written by k_wh_advance ...
    void* const* ring_events;      // ... and 4 hipEvent_t recorded behind it: the host stops launching levels two behind the first empty one (null: all levels)
};
void whitted_run(hipStream_t st, const MiptWhittedLaunchParams& p, uint32_t n_paths, const DeviceBvh& bvh, const WhGeometry& geo,
                 const MiptMaterialData* d_materials, uint32_t n_materials, const WhTexture* d_textures, uint32_t n_textures,
                 const WhBuffers& B, DeviceCounters* ctr, const TraceOpts& opts);
void launch_unpermute(hipStream_t st, const float4* gathered, uint32_t world, uint32_t width, uint32_t height,
                      float4* accum, uint32_t* frame);

}  // namespace mipt
