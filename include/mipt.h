/*
 * mipt.h -- C ABI of libmipt.so, the MI355X-native wavefront path tracer that
 * replaces the optixTrace / raygen / closest-hit / miss pipeline of
 * SDK/optixPathTracer (and, later, SDK/imgui_test) of awegsche/OptiX_Raytracer.
 *
 * Plain C: pointers, sizes, PODs.  No torch / HIP types in any signature
 * (a stream is passed as `void*` holding a hipStream_t; NULL = default stream).
 * Every entry point returns a MiptResult (0 = success), like OptixResult
 * (reference: include/optix_types.h, OPTIX_CHECK in SDK/sutil/Exception.h:82-112).
 * Contexts are not thread safe (same as the reference samples: one host thread,
 * one stream per state, SDK/optixPathTracer/optixPathTracer.cpp:442).
 *
 * File:line citations are relative to /root/reference.
 */
#ifndef MIPT_H
#define MIPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPT_ABI_VERSION 4   /* 2: MiptAccelInfo.node_format, MIPT_OPT_NODE_FORMAT, mipt_group_*; 3: mipt_whitted_set_geometry_instanced, mipt_scene_mesh_normals / _texcoords / _colors;
                                4: MIPT_OPT_BUILD = 2 (device SAH builder, the default), 3 and 4 (host / device SAH with spatial splits), mipt_host_bvh_create_spatial */

typedef enum MiptResult {
    MIPT_SUCCESS                 = 0,
    MIPT_ERROR_INVALID_VALUE     = 1,
    MIPT_ERROR_OUT_OF_MEMORY     = 2,
    MIPT_ERROR_HIP               = 3,   /* a HIP runtime call failed; see mipt_last_error_string */
    MIPT_ERROR_NO_DEVICE         = 4,
    MIPT_ERROR_INVALID_HANDLE    = 5,   /* traversable handle unknown to this context */
    MIPT_ERROR_NOT_CONFIGURED    = 6,   /* launch before hit-group records / accel were set */
    MIPT_ERROR_BVH_TOO_DEEP      = 7,
    MIPT_ERROR_INTERNAL          = 8
} MiptResult;

typedef struct MiptContext MiptContext;

/* same signature as the OptiX log callback used by the samples
 * (SDK/optixPathTracer/optixPathTracer.cpp:530-533) */
typedef void (*MiptLogCallback)(unsigned int level, const char* tag, const char* message, void* cbdata);

/* ---------------------------------------------------------------------------
 * Structs kept bit-for-bit from the reference
 * ------------------------------------------------------------------------- */

/* ParallelogramLight, SDK/optixPathTracer/optixPathTracer.h:82-88 (60 bytes) */
typedef struct MiptParallelogramLight {
    float corner[3];
    float v1[3], v2[3];
    float normal[3];
    float emission[3];
} MiptParallelogramLight;

/* Params, SDK/optixPathTracer/optixPathTracer.h:91-107 -- 152 bytes, align 8:
 * subframe_index 0, accum_buffer 8, frame_buffer 16, width 24, height 28,
 * samples_per_launch 32, eye 36, U 48, V 60, W 72, light 84, handle 144.
 * accum_buffer: width*height float4 (device), index y*width+x, row 0 = bottom.
 * frame_buffer: width*height uchar4 (device), sRGB8, alpha 255. */
typedef struct MiptParams {
    unsigned int           subframe_index;
    void*                  accum_buffer;
    void*                  frame_buffer;
    unsigned int           width;
    unsigned int           height;
    unsigned int           samples_per_launch;
    float                  eye[3];
    float                  U[3];
    float                  V[3];
    float                  W[3];
    MiptParallelogramLight light;
    unsigned long long     handle;      /* OptixTraversableHandle stays an opaque u64 */
} MiptParams;

/* MissData, optixPathTracer.h:115-118 */
typedef struct MiptMissData {
    float bg_color[4];
} MiptMissData;

/* HitGroupData, optixPathTracer.h:121-126 (32 bytes).  `vertices` is a DEVICE
 * pointer to float4-strided vertices, 3 per primitive, owned by the caller and
 * kept alive until the context is destroyed (optixPathTracer.cpp:877). */
typedef struct MiptHitGroupData {
    float       emission_color[3];
    float       diffuse_color[3];
    const void* vertices;
} MiptHitGroupData;

/* ---------------------------------------------------------------------------
 * BVH memory layout (device and export): linear arrays, 64-byte nodes.
 * ------------------------------------------------------------------------- */

/* inner node: both children's boxes + references.
 * ref >= 0 : inner node index;  ref < 0 : leaf, ~ref = (first_tri << 4) | count */
typedef struct MiptBvhNode {
    float    c0x[2], c0y[2];   /* child0 lo.x hi.x lo.y hi.y */
    float    c1x[2], c1y[2];   /* child1 lo.x hi.x lo.y hi.y */
    float    c0z[2], c1z[2];   /* child0 lo.z hi.z | child1 lo.z hi.z */
    int32_t  c0, c1;
    uint32_t pad0, pad1;
} MiptBvhNode;

/* Node format 1 (MIPT_OPT_NODE_FORMAT, the default where the scene suits it): every child box of the nodes above is
 * rounded OUTWARD to fp16-representable values when the BVH is built or refitted, and the entry-pool traversal kernels fetch
 * this packed mirror instead -- 32 bytes, two 16-byte loads per visit instead of four.  Both arrays hold the SAME box
 * values (a half converts to float exactly), so every kernel, mipt_accel_export and an external checker see one BVH.
 * Wider boxes only weaken pruning: hit records do not change.  c0/c1 as in MiptBvhNode. */
typedef struct MiptBvhNode32 {
    uint16_t c0x[2], c0y[2], c0z[2];   /* child0 lo.x hi.x lo.y hi.y lo.z hi.z, IEEE binary16 bit patterns */
    int32_t  c0;
    uint16_t c1x[2], c1y[2], c1z[2];
    int32_t  c1;
} MiptBvhNode32;

/* triangle in leaf order: v0, e1 = v1 - v0, e2 = v2 - v0 (fp32), original index */
typedef struct MiptBvhTri {
    float v0[3]; uint32_t prim;
    float e1[3]; uint32_t pad1;   /* MiptGeometryFlags of the triangle */
    float e2[3]; uint32_t pad2;
} MiptBvhTri;

typedef struct MiptAccelInfo {
    uint32_t num_nodes;
    uint32_t num_triangles;
    uint32_t depth;            /* deepest leaf; traversal stack needs <= depth entries */
    uint32_t leaf_max;
    uint64_t device_bytes;
    double   build_seconds;
    uint32_t node_format;      /* the format in effect (MIPT_OPT_NODE_FORMAT): 2 BVH4 mirror, 1 packed BVH2 mirror, 0 fp32 boxes only */
    uint32_t num_instances;    /* > 0: an instance (two-level) structure; num_nodes = its top level, num_triangles = all referenced */
    uint32_t stack_entries;    /* entries a lane's traversal stack can need over the structure the tiled kernel walks (format 2: the
                                  BVH4 mirror's worst case, <= 32; else = depth) */
    uint32_t mirror_nodes;     /* nodes of the mirror (format 2: BVH4 nodes; format 1: = num_nodes; format 0: 0) */
} MiptAccelInfo;

/* one triangle build input == OptixBuildInputTriangleArray as the samples use it
 * (optixPathTracer.cpp:610-621; sutil/Scene.cpp:900-960; imgui_test/triangle_gas.cpp:196-214) */
typedef struct MiptBuildInput {
    const void*     vertices;            /* host or device pointer (detected) */
    uint32_t        num_vertices;
    uint32_t        vertex_stride_bytes; /* 0 => 12 (packed float3) */
    const void*     indices;             /* NULL => non-indexed, 3 consecutive vertices per triangle */
    uint32_t        index_format;        /* 0 none, 2 = uint16 triplets, 4 = uint32 triplets */
    uint32_t        index_stride_bytes;  /* bytes between triplets; 0 => 3*index_format */
    uint32_t        num_triangles;       /* indexed: number of triplets; else num_vertices/3 */
    const uint32_t* sbt_index_offsets;   /* per-triangle hit-group record index (host or device), NULL => 0 */
    const float*    transform3x4;        /* optional row-major 3x4 object->world (host), NULL => identity */
    const uint32_t* flags;               /* OptixBuildInputTriangleArray::flags: one MiptGeometryFlags word per SBT record
                                            (host), selected by sbt_index_offsets; NULL => MIPT_GEOMETRY_FLAG_NONE */
    uint32_t        num_sbt_records;     /* length of flags */
} MiptBuildInput;

/* OptixGeometryFlags / OptixRayFlags values the samples use (optix_types.h); stored per triangle in MiptBvhTri::pad1 */
typedef enum MiptGeometryFlags {
    MIPT_GEOMETRY_FLAG_NONE                          = 0,
    MIPT_GEOMETRY_FLAG_DISABLE_ANYHIT                = 1u << 0,
    MIPT_GEOMETRY_FLAG_REQUIRE_SINGLE_ANYHIT_CALL    = 1u << 1,   /* always true here: a triangle sits in one leaf */
    MIPT_GEOMETRY_FLAG_DISABLE_TRIANGLE_FACE_CULLING = 1u << 2
} MiptGeometryFlags;
typedef enum MiptRayFlags {
    MIPT_RAY_FLAG_NONE                        = 0,
    MIPT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES  = 1u << 4    /* back = the ray sees the (v1-v0) x (v2-v0) side from behind */
} MiptRayFlags;

typedef struct MiptLaunchStats {
    uint64_t closest_rays;     /* primary + continuation rays traced by the last launch */
    uint64_t shadow_rays;      /* occlusion rays traced by the last launch */
    uint64_t node_visits;      /* 64-B nodes fetched by both traversal kernels; only counted when */
    uint64_t tri_tests;        /* 48-B triangles fetched;  MIPT_OPT_COUNT_VISITS is on           */
    uint64_t shadow_node_visits; /* the any-hit kernel's share of node_visits */
    uint64_t shadow_tri_tests;   /* the any-hit kernel's share of tri_tests   */
    uint32_t bounces;          /* wavefront iterations executed */
    uint32_t reserved;
} MiptLaunchStats;

typedef enum MiptOption {
    MIPT_OPT_MAX_DEPTH    = 1, /* 0 = unbounded (reference behaviour, Russian roulette only);
                                  k = at most k closest-hit queries per path (build extension) */
    MIPT_OPT_COUNT_VISITS = 2, /* 1 = traversal kernels count node visits / triangle tests */
    MIPT_OPT_TRACE_GRID   = 3, /* traversal workgroups (one wave each) per CU; 0 = default: one per 64-ray tile of the launch,
                                  16 .. 128 per CU */
    MIPT_OPT_PERSISTENT   = 4, /* traversal kernels: 3 (default) closest hit in 64-ray tiles per wave, one ray per lane, then a
                                  per-wave entry pool for the tail of each tile; any-hit launches and closest-hit launches
                                  with few rays in static chunks of 8 .. 64 rays per wave, on the pool from the root;
                                  2 persistent streaming waves with lane refill and subtree stealing (also used when a BVH
                                  does not fit the pool's packed entries: >= 2^25 nodes or >= 2^23 triangles, and by counting
                                  launches); 0 one thread per ray (tight while-while loop; also the
                                  fallback for trees deeper than 32). */
    MIPT_OPT_PROFILE      = 5, /* 1 = bracket every kernel of a launch with HIP events (mipt_get_launch_timings),
                                  2 = only the two traversal kernels (10 -> 4 event records per frame),
                                  3 = those of every 8th launch only (a record costs ~4 us of stream time: mode 2 is 2.6 % of
                                  the headline frame, mode 3 0.3 %) */
    MIPT_OPT_REFILL_MIN   = 7, /* variant 2: idle lanes that trigger a swap-in/prefetch round (default 16) */
    MIPT_OPT_STEAL_MIN    = 8, /* finished lanes of a tile that make the wave go over to its entry pool (variant 3) / idle
                                  lanes that trigger a round in which they take stack entries of busy rays once the ray
                                  range is exhausted (variant 2); default 24, 65 = never (3: when all are finished) */
    MIPT_OPT_BATCH_SAMPLES = 10, /* 0 (default): one wavefront pass per sample, contributions added in the reference's
                                  order (bit-identical to the oracle); 1: all samples_per_launch samples of a launch are
                                  paths of ONE pass with per-sample accumulators (same values to rounding, far fewer passes) */
    MIPT_OPT_BUILD        = 12, /* builder of mipt_accel_build, all in the same node / triangle layout; a render does not depend on it:
                                  2 (default since ABI 4): on the device, as optixAccelBuild -- top-down binned SAH (32 bins per axis, all open
                                  ranges of a level at once) down to ranges of 64 triangles, each of which one wave then builds by full-sweep
                                  SAH; leaves by the SAH termination rule.  7 ms per million triangles, frames within 2 % of those over the
                                  host builder's tree on every scene measured (profiles/r04_builders.txt), the same tree in every run; inputs of fewer than
                                  2048 triangles are built on the host (a device build costs ~0.6 ms of launches whatever its size);
                                  1: on the device, fastest build (~ PREFER_FAST_BUILD): Morton codes, radix sort, Karras' radix tree, the same
                                  64-triangle treelets: 3 ms per million triangles, frames equal on evenly spread triangles (the headline
                                  scene), 10-20 % slower on scenes whose top levels need SAH (large triangles among small ones, height fields);
                                  3: binned SAH with SPATIAL SPLITS on the host cores (SBVH: a triangle that straddles a split plane is referenced
                                  on both sides, each reference with the conservatively clipped bounds of its part; at most 50 % more triangle
                                  records, env MIPT_SBVH_EXTRA): the best tree for a static scene (~ PREFER_FAST_TRACE) -- frames over scenes of
                                  triangles as large as the gaps between them 6-7 % faster (the headline scene 0.557 -> 0.520 ms), 0-3 % elsewhere
                                  -- at ~3 s per million triangles; MiptAccelInfo::num_triangles then counts RECORDS; mipt_accel_update
                                  gives every record the box of its WHOLE moved triangle (a sound tree, but the overlap the splits removed is back:
                                  build it again instead);
                                  4: spatial splits ON THE DEVICE (~ PREFER_FAST_TRACE at the cost of a build): builder 2 whose level loop cuts the
                                  triangles that straddle a split plane in two wherever the split's sides overlap (the plane is the object
                                  split's own; clipped bounds in double, rounded outward; the same reference budget, MIPT_SBVH_EXTRA), down to
                                  ranges of 128 references (env MIPT_SPLIT_BELOW), which are then split by objects only: builder 3's frames
                                  (the headline scene 0.519 against 0.516 ms) from a build of ~12 ms per million triangles instead of ~3 s; the
                                  same tree in every run; num_triangles counts RECORDS; mipt_accel_update clips every record's moved triangle to the cell the
                                  builder cut it to (kept, 24 bytes per record): the same vertices give the same tree back, small motions keep most of
                                  what the cuts bought (1 M triangles moved by 5 % of their size: 0.549 -> 0.594 ms, a refitted tree of builder 2:
                                  0.581) -- scenes that are refitted every frame are builder 2's; an input it gives up
                                  on (lists that overflow) is built as by 2, and the records stay below 2^23 whatever the budget (the entry pool's index width);
                                  0: binned SAH on the host cores (155 ms per million triangles).  1 and 2 fall back to 0 for inputs of
                                  <= 4 triangles and when their tree is deeper than the 32-entry traversal stack */
    MIPT_OPT_TAIL_PATHS   = 13, /* once at most this many paths of a pass are alive (the count is read back two bounces late;
                                  half the value when MAX_DEPTH caps the paths), one kernel runs each of them to its
                                  end, a wave per 64 paths, instead of five kernels per bounce; a pass that starts with at
                                  most this many paths (three quarters of it when capped) starts there.  Same pixels, same ray counts.  Default 1048576, 0 = off */
    MIPT_OPT_NODE_FORMAT  = 14, /* set before mipt_accel_build.  What the entry-pool traversal kernels walk:
                                  2 (default): child boxes rounded outward to fp16 values + a BVH4 mirror of the tree (64-byte nodes
                                     of four fp16 boxes, collapsed on the device so that a lane's stack never needs more than 32 entries),
                                     for either builder's tree;
                                  1: the rounded boxes + the packed 32-byte BVH2 mirror (MiptBvhNode32);
                                  0: fp32 boxes, 64-byte nodes only.
                                  Scenes that do not suit fp16 boxes (a coordinate beyond +-32768, or the largest extent below a
                                  quarter of the largest coordinate) silently keep format 0.  The 64-byte BVH2 nodes stay the
                                  canonical tree in every format (mipt_accel_export, counting launches, instance walk). */
    MIPT_OPT_TILE_ORDER   = 15, /* the primary closest-hit launch of a full frame walks its 64-ray tiles 1 (default): heaviest first,
                                  by their cost (bulk steps + pool rounds) in the PREVIOUS launch -- a tile's cost varies 1 : 5, and in
                                  image order the kernel ends in a long drain while late, heavy tiles run alone.  The order is made
                                  on the device inside the launches a frame has anyway (16 cost classes, stable counting sort:
                                  image order inside a class keeps the L2 locality) and only schedules: it cannot change a result.
                                  0: image order.  The first launch after a change of size (or of a shard's rank) and launches with
                                  batched samples run in image order. */
    MIPT_OPT_BURST        = 9  /* streaming waves: traversal steps per bookkeeping round, value = bulk | dry << 8
                                  (bulk: while fresh rays remain; dry: once the ray range is exhausted) */
} MiptOption;

/* per-kernel device time (ms), summed over every kernel launched since the previous
 * mipt_get_launch_timings call (or since MIPT_OPT_PROFILE was switched on); only filled
 * while MIPT_OPT_PROFILE is on.  Reading the timings synchronises the stream and resets the sums. */
typedef struct MiptLaunchTimings {
    float    ms_raygen, ms_trace_closest, ms_shade, ms_trace_any, ms_resolve;
    uint32_t n_trace_closest, n_trace_any;   /* kernel launches behind the sums */
} MiptLaunchTimings;

/* ---------------------------------------------------------------------------
 * Context   (replaces createContext, optixPathTracer.cpp:555-573)
 * ------------------------------------------------------------------------- */
int         mipt_abi_version(void);
int         mipt_context_create(int hip_device, MiptLogCallback cb, void* cbdata, int log_level, MiptContext** out);
int         mipt_context_destroy(MiptContext* ctx);
const char* mipt_error_name(int result);
const char* mipt_last_error_string(const MiptContext* ctx);
int         mipt_set_option(MiptContext* ctx, int option, uint64_t value);
int         mipt_get_option(const MiptContext* ctx, int option, uint64_t* value);

/* ---------------------------------------------------------------------------
 * Acceleration structure   (replaces buildMeshAccel / optixAccelBuild+Compact,
 * optixPathTracer.cpp:576-684; library owns BVH memory)
 * ------------------------------------------------------------------------- */
int mipt_accel_build(MiptContext* ctx, const MiptBuildInput* inputs, uint32_t num_inputs,
                     unsigned long long* handle_out);
/* Refit (= optixAccelBuild with OPTIX_BUILD_OPERATION_UPDATE, include/optix_types.h; used by SDK/optixDynamicGeometry): the
 * same inputs as the build with moved vertices -- triangle count, order, SBT offsets and flags unchanged.  The tree keeps
 * its topology, triangle records and boxes are recomputed on the device; traversal results are those of a fresh build
 * (hits do not depend on the tree), traversal cost grows as the geometry drifts from the layout it was built for. */
int mipt_accel_update(MiptContext* ctx, unsigned long long handle, const MiptBuildInput* inputs, uint32_t num_inputs);
int mipt_accel_destroy(MiptContext* ctx, unsigned long long handle);
int mipt_accel_info(const MiptContext* ctx, unsigned long long handle, MiptAccelInfo* info);
/* copies the linear BVH to host arrays sized from mipt_accel_info (inspection, external checkers) */
int mipt_accel_export(const MiptContext* ctx, unsigned long long handle,
                      MiptBvhNode* nodes_out, MiptBvhTri* tris_out);

/* ---------------------------------------------------------------------------
 * Instance (two-level) acceleration structures   (replaces optixAccelBuild over OPTIX_BUILD_INPUT_TYPE_INSTANCES as
 * Scene::buildInstanceAccel uses it, SDK/sutil/Scene.cpp:1134-1212).  MiptInstance IS OptixInstance (include/optix_types.h,
 * 80 bytes: row-major 3x4 object->world transform, instanceId, sbtOffset, visibilityMask, flags, traversableHandle, pad[2]);
 * traversableHandle must be a triangle structure of mipt_accel_build in the same context, built from OBJECT-space vertices.
 * The library builds a BVH over the instances' world boxes and keeps the world->object matrices; rays are transformed at the
 * instance boundary.  A hit in an instance structure reports the FLATTENED primitive index
 *     sum of the triangle counts of instances 0 .. k-1  +  the mesh-local primitive index
 * i.e. the order of mipt_scene_flatten*, so hit-group records, the Whitted geometry arrays and the launch functions are used
 * exactly as with one flattened BVH.  instanceId / sbtOffset / visibilityMask / flags are stored but not interpreted (the samples
 * use mask 1 and no flags).  mipt_accel_update_instances takes the same instances with new transforms (an instance moved).
 * Alpha-tested / blended materials (the Whitted any-hit programs) work over an instance structure too (k_wh_trace_pool_inst).
 * ------------------------------------------------------------------------- */
typedef struct MiptInstance {
    float              transform[12];
    unsigned int       instanceId;
    unsigned int       sbtOffset;
    unsigned int       visibilityMask;
    unsigned int       flags;
    unsigned long long traversableHandle;
    unsigned int       pad[2];
} MiptInstance;
int mipt_accel_build_instances(MiptContext* ctx, const MiptInstance* instances /* host */, uint32_t num_instances,
                               unsigned long long* handle_out);
int mipt_accel_update_instances(MiptContext* ctx, unsigned long long handle, const MiptInstance* instances, uint32_t num_instances);
/* for external checkers: top-level nodes (num_nodes of mipt_accel_info), then per instance (num_instances entries each, nullable):
 * order_out = instance index per leaf slot, inv12_out = world->object matrices (12 floats each), prim_base_out, gas_handles_out */
int mipt_accel_export_instances(const MiptContext* ctx, unsigned long long handle, MiptBvhNode* nodes_out, uint32_t* order_out,
                                float* inv12_out, uint32_t* prim_base_out, unsigned long long* gas_handles_out);

/* Host-only build (no device needed): the same builder, for inspection and for checking the BVH
 * without a GPU.  vertices9: num_triangles * 9 floats (v0 v1 v2, world space). */
typedef struct MiptHostBvh MiptHostBvh;
int mipt_host_bvh_create(const float* vertices9, uint32_t num_triangles, MiptHostBvh** out);
/* the tree of MIPT_OPT_BUILD = 3 (spatial splits; `extra`: added references per triangle at most, 0..4; mipt_host_bvh_info's
 * num_triangles counts the triangle RECORDS, duplicates included) */
int mipt_host_bvh_create_spatial(const float* vertices9, uint32_t num_triangles, float extra, MiptHostBvh** out);
int mipt_host_bvh_info(const MiptHostBvh* bvh, MiptAccelInfo* info);
int mipt_host_bvh_copy(const MiptHostBvh* bvh, MiptBvhNode* nodes_out, MiptBvhTri* tris_out);
int mipt_host_bvh_destroy(MiptHostBvh* bvh);

/* ---------------------------------------------------------------------------
 * Shader binding table   (replaces createSBT, optixPathTracer.cpp:829-898;
 * the programs themselves are fixed-function kernels of this library)
 * ------------------------------------------------------------------------- */
int mipt_sbt_set_hitgroups(MiptContext* ctx, const MiptHitGroupData* records, uint32_t count);
int mipt_sbt_set_miss(MiptContext* ctx, const MiptMissData* miss);

/* ---------------------------------------------------------------------------
 * Launch   (replaces cudaMemcpyAsync(params) + optixLaunch in launchSubframe,
 * optixPathTracer.cpp:488-511).  `params` is a HOST pointer to the 152-byte
 * struct; width/height/depth as optixLaunch (depth must be 1).
 * ------------------------------------------------------------------------- */
int mipt_launch(MiptContext* ctx, void* stream, const MiptParams* params,
                unsigned int width, unsigned int height, unsigned int depth);
int mipt_sync(MiptContext* ctx);
int mipt_get_launch_stats(MiptContext* ctx, MiptLaunchStats* stats);
int mipt_get_launch_timings(MiptContext* ctx, MiptLaunchTimings* timings);

/* ---------------------------------------------------------------------------
 * Ray-buffer queries (precedent: SDK/optixRaycasting, Ray/Hit buffers,
 * optixRaycastingKernels.h:35-47).  rays: n * {ox,oy,oz,tmin, dx,dy,dz,tmax}
 * (device).  hits: n * {t, prim(as bits), u, v} (device); miss => t = -1, prim = ~0.
 * occluded: n * uint32 (device), 1 = any hit in (tmin,tmax).
 * ------------------------------------------------------------------------- */
int mipt_trace_closest(MiptContext* ctx, void* stream, unsigned long long handle,
                       const void* rays, uint32_t n, void* hits);
int mipt_trace_any(MiptContext* ctx, void* stream, unsigned long long handle,
                   const void* rays, uint32_t n, void* occluded);

/* ---------------------------------------------------------------------------
 * Multi-GPU image-tile sharding (precedent: SDK/optixMultiGPU +
 * StaticWorkDistribution, sutil/WorkDistribution.h:34-91: 8x4 tiles, strip of
 * `world` tiles, tile->rank rotated per strip row).  Each rank renders its
 * pixels into a compact float4 buffer of mipt_shard_num_samples() entries;
 * the caller all-gathers the shards (RCCL) and calls mipt_unpermute.
 * ------------------------------------------------------------------------- */
int mipt_shard_num_samples(unsigned int width, unsigned int height, unsigned int world, uint32_t* n_out);
int mipt_shard_pixel(unsigned int width, unsigned int height, unsigned int world, unsigned int rank,
                     uint32_t sample_idx, int32_t* px, int32_t* py);
/* params->accum_buffer / frame_buffer are ignored; compact_accum (device, in/out) holds
 * this rank's running average, indexed by sample index. */
int mipt_launch_shard(MiptContext* ctx, void* stream, const MiptParams* params,
                      unsigned int width, unsigned int height,
                      unsigned int rank, unsigned int world, void* compact_accum);
/* gathered: world * num_samples float4 (device), rank-major; writes canonical accum + frame */
int mipt_unpermute(MiptContext* ctx, void* stream, const void* gathered, unsigned int world,
                   unsigned int width, unsigned int height, void* accum_buffer, void* frame_buffer);

/* ---------------------------------------------------------------------------
 * In-process multi-GPU group (shape of SDK/optixMultiGPU: ONE host thread, every device its own state and stream,
 * optixMultiGPU.cpp:562-594 launchSubframe, :1093-1102 per-device scene).  The group owns one MiptContext and one stream per
 * device, the compact shard buffers and the RCCL communicators (ncclCommInitAll; librccl is bound with dlopen when the
 * group is created).  The caller replicates the scene: for every member it builds the acceleration structure and sets the
 * hit-group records through that member's context (mipt_group_member), exactly as the sample does per PathTracerState.
 *   mipt_group_launch    params[i] is member i's Params (its own traversable handle; accum / frame pointers are ignored);
 *                        every member renders its StaticWorkDistribution shard of the frame on its stream, asynchronously.
 *   mipt_group_allgather ONE grouped ncclAllGather of the compact float4 shards, then on every member i with
 *                        accum_buffers[i] != NULL the scatter into its canonical accum_buffer / frame_buffer (device
 *                        pointers on that member's device; frame_buffers or any of its entries may be NULL).
 *   mipt_group_sync      waits for every member's stream (the sample's per-device CUDA_SYNC_CHECK).
 * A one-device group works without RCCL.  Not thread safe.
 * ------------------------------------------------------------------------- */
typedef struct MiptGroup MiptGroup;
int         mipt_group_create(const int* hip_devices, uint32_t n, MiptLogCallback cb, void* cbdata, int log_level, MiptGroup** out);
int         mipt_group_destroy(MiptGroup* group);
int         mipt_group_size(const MiptGroup* group, uint32_t* n_out);
int         mipt_group_member(MiptGroup* group, uint32_t index, MiptContext** ctx_out, int* hip_device_out, void** stream_out);
int         mipt_group_launch(MiptGroup* group, const MiptParams* params, unsigned int width, unsigned int height);
int         mipt_group_allgather(MiptGroup* group, void* const* accum_buffers, void* const* frame_buffers);
int         mipt_group_sync(MiptGroup* group);
int         mipt_group_uses_rccl(const MiptGroup* group);      /* 1 when the exchange goes through RCCL */
const char* mipt_group_last_error_string(const MiptGroup* group);

/* ---------------------------------------------------------------------------
 * Layouts of the glTF launch path kept for its drop-in: whitted::LaunchParams
 * (SDK/cuda/whitted.h:59-77, 128 bytes) and BufferView<T> (SDK/cuda/BufferView.h:33-49, 16 bytes).
 * mipt_whitted_launch (below) runs the opaque, untextured subset of cuda/whitted.cu on them.
 * ------------------------------------------------------------------------- */
typedef struct MiptBufferView {
    unsigned long long data;            /* CUdeviceptr */
    unsigned int       count;
    unsigned short     byte_stride;
    unsigned short     elmt_byte_size;
} MiptBufferView;

typedef struct MiptWhittedLaunchParams {
    unsigned int       width, height, subframe_index;
    void*              accum_buffer;    /* float4* */
    void*              frame_buffer;    /* uchar4* */
    int                max_depth;
    float              scene_epsilon;
    float              eye[3], U[3], V[3], W[3];
    MiptBufferView     lights;          /* BufferView<Light> */
    float              miss_color[3];
    unsigned long long handle;
} MiptWhittedLaunchParams;

/* MaterialData::Texture (SDK/cuda/MaterialData.h:60-72) with the CUDA texture object replaced by the index of the
 * scene's texture; offset / rotation / scale come from KHR_texture_transform (Scene.cpp:214-265) */
typedef struct MiptTextureInfo {
    int32_t texture;              /* index for mipt_scene_texture, -1 = none */
    int32_t texcoord;             /* TEXCOORD set, 0 or 1 */
    float   offset[2];
    float   rotation[2];          /* { sin, cos } */
    float   scale[2];
} MiptTextureInfo;

/* MaterialData (SDK/cuda/MaterialData.h:33-130), the PBR branch */
typedef struct MiptMaterialData {
    float   base_color[4];
    float   metallic, roughness;
    float   emissive_factor[3];
    float   alpha_cutoff;
    int32_t alpha_mode;           /* 0 OPAQUE, 1 MASK, 2 BLEND */
    int32_t double_sided;
    MiptTextureInfo base_color_tex, metallic_roughness_tex, normal_tex, emissive_tex;
} MiptMaterialData;

/* one glTF texture = sampler state + decoded image (Scene::addImage / addSampler, Scene.cpp:576-652).
 * texels: RGBA, 8 or 16 bits per component, row 0 = top row of the image; owned by the scene (host memory).
 * An image the loader cannot decode (not a non-interlaced PNG or a Huffman-coded 8-bit JPEG) yields width = height = 0 and
 * texels = NULL: materials that use it render untextured. */
typedef struct MiptTextureData {
    uint32_t    width, height, bits;
    uint32_t    address_s, address_t;     /* 0 wrap, 1 clamp, 2 mirror (cudaTextureAddressMode order) */
    uint32_t    filter;                   /* 0 point, 1 linear */
    int32_t     image;                    /* glTF image index */
    const void* texels;
} MiptTextureData;

/* struct Light, SDK/cuda/Light.h:33-71 (36 bytes): type 0 POINT {color, intensity, position, falloff},
 * type 1 AMBIENT {color} */
typedef struct MiptLight {
    int   type;
    float color[3];
    float intensity;
    float position[3];
    int   falloff;
} MiptLight;

/* The Whitted / glTF-PBR programs of cuda/whitted.cu (__raygen__pinhole :44-97, __miss__constant_radiance :139-142,
 * __closesthit__radiance :149-289) as wavefront stages: pinhole raygen (pixel centre at subframe 0), LocalGeometry
 * (cuda/LocalGeometry.h:59-176: interpolated position / normal / COLOR_0 / two texcoord sets with dpdu, dpdv),
 * base-colour (linearised), metallic-roughness, emissive and normal textures with KHR_texture_transform
 * (cuda/LocalShading.h:37-53), GGX direct lighting from the point / ambient lights in params->lights with one
 * occlusion ray per point light, progressive lerp + sRGB8.  Geometry comes as world-space triangles
 * (mipt_scene_flatten*): the SBT records of Scene::createSBT (Scene.cpp:1368-1470) become device arrays, a
 * material table and a texture table.  Radiance rays cull back faces of single-sided materials
 * (OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES, whitted_cuda.h:110; per-triangle geometry flags from MiptBuildInput::flags).
 * When a material is MASK or BLEND the traversal runs the any-hit programs (cuda/whitted.cu:100-136: alpha test for
 * radiance rays, accumulated attenuation for occlusion rays -- a ray's candidates in the order of (t, primitive index), which OptiX
 * leaves open) and BLEND hits continue with a ray from the hit distance,
 * composed back to front over at most MAX_TRACE_DEPTH = 8 levels (:268-286).
 * Texture fetches use fp32 bilinear weights (the hardware's 8-bit weights are not modelled). */
typedef struct MiptWhittedGeometry {
    const void*     vertices4;            /* device: 3 float4 per triangle, world space */
    const void*     normals4;             /* device, nullable: 3 float4 {normal, has_normal} per triangle */
    const uint32_t* material_index;       /* device, nullable: per triangle; >= num_materials selects the last material */
    const float*    texcoords[2];         /* device, nullable: 3 float2 per triangle (TEXCOORD_0 / _1) */
    const uint8_t*  texcoord_present[2];  /* device, nullable: per triangle; NULL = present wherever texcoords is given */
    const float*    colors4;              /* device, nullable: 3 float4 per triangle (COLOR_0) */
    const uint8_t*  color_present;        /* device, nullable */
} MiptWhittedGeometry;
/* geometry arrays stay caller-owned (device); materials (host array) and textures (host array, host texels) are
 * copied into the context.  textures may be NULL / 0. */
int mipt_whitted_set_geometry(MiptContext* ctx, const MiptWhittedGeometry* geometry,
                              const MiptMaterialData* materials, uint32_t num_materials,
                              const MiptTextureData* textures, uint32_t num_textures);
/* The same with the shading tables the reference keeps (Scene::createSBT, Scene.cpp:1368-1480: one set of buffers per MESH, one SBT
 * record per instance x primitive over them): per-mesh OBJECT-space tables + instance records instead of flattened world-space arrays,
 * so that instancing saves shading memory as well as BVH memory.  A hit's flattened primitive index (instances in order, mesh-local
 * index inside: what the instance structure of mipt_accel_build_instances over the same records reports) is resolved through a table
 * of one word per flattened primitive; a corner is taken to world space in the kernel with the arithmetic mipt_scene_flatten uses on
 * the host (and a normal with mipt_scene_flatten_normals'), so frames are bit-identical to those over the flattened arrays.
 * meshes[m].geometry: DEVICE pointers as in MiptWhittedGeometry (caller-owned, alive until the next set_geometry* call), vertices and
 * normals in object space (normals4: {n, 1} or all 0); instances[i].transform: row-major 3x4 object -> world (the OptixInstance's).
 * ORDER: instances[] must list the same instances in the same order, with the same triangle count per mesh, as the records the
 * traversable of the launch was built from (mipt_accel_build_instances), or -- for a flattened BVH -- as the flatten that made its
 * vertex buffer.  mipt_whitted_launch checks what the library can see and fails with MIPT_ERROR_INVALID_VALUE otherwise: the total
 * triangle count against the traversable's, and over an instance structure every instance's first flattened primitive. */
typedef struct MiptWhittedMesh {
    MiptWhittedGeometry geometry;
    uint32_t            num_triangles;
    uint32_t            reserved;
} MiptWhittedMesh;
typedef struct MiptWhittedInstance {
    uint32_t mesh;
    float    transform[12];
} MiptWhittedInstance;
int mipt_whitted_set_geometry_instanced(MiptContext* ctx, const MiptWhittedMesh* meshes, uint32_t num_meshes,
                                        const MiptWhittedInstance* instances, uint32_t num_instances,
                                        const MiptMaterialData* materials, uint32_t num_materials,
                                        const MiptTextureData* textures, uint32_t num_textures);
/* replaces optixLaunch( scene.pipeline(), ... ) of optixMeshViewer.cpp:283-308; params is a HOST pointer */
int mipt_whitted_launch(MiptContext* ctx, void* stream, const MiptWhittedLaunchParams* params,
                        unsigned int width, unsigned int height, unsigned int depth);

/* ---------------------------------------------------------------------------
 * SDK/imgui_test (the fork author's interactive tracer): the same traversal kernels under its own
 * raygen / closest-hit / miss programs (imgui_test/optixTriangle.cu:103-268).  The structs keep the
 * layouts of the sample's device objects, so its host code can upload them unchanged.
 * ------------------------------------------------------------------------- */

/* class Camera, imgui_test/camera.h:155-173 (92 bytes) */
typedef struct MiptImguiCamera {
    float   eye[3], lookat[3], up[3];
    uint8_t ortho, pad_[3];
    float   fov, fd, aperture, speed;
    float   u[3], v[3], w[3];
} MiptImguiCamera;

/* class LightVariant, imgui_test/light.h:10-51 (44 bytes): 40-byte union + tag.
 *   tag 0 PointLight       (point_light.h):       a = position,  b = {lumi.xyz, dark.x}, c = dark.yz, -
 *   tag 1 DirectionalLight (directional_light.h): a = direction, b = {lumi.xyz, jitter}, c = dark.xyz
 *   tag 2 VolumetricLight  (volumetric_light.h):  a = position,  b = {radius, lumi.xyz}, c = dark.xyz */
typedef struct MiptImguiLight {
    float   a[3];
    float   b[4];
    float   c[3];
    int32_t tag;
} MiptImguiLight;

/* struct Params, imgui_test/optixTriangle.h:42-108 (128 bytes); every pointer is a DEVICE pointer */
typedef struct MiptImguiParams {
    unsigned int        image_width, image_height, samples_per_frame;
    const MiptImguiCamera* camera;
    unsigned int        dt;
    uint8_t             dirty;           /* bool: overwrite the film instead of adding to it */
    void*               image;           /* uchar4 per pixel */
    float*              film;            /* packed float3 per pixel */
    float               tfactor;
    unsigned long long  handle;
    const float*        normals;         /* packed float3, 3 per primitive */
    const float*        vertices;        /* packed float3, 3 per primitive */
    const int*          mat_indices;
    int                 nmat_indices;
    const MiptImguiLight* lights;
    int                 nlights;
    const float*        materials;       /* DiffuseMaterial: one float3 colour each (diffuse.h:3-16) */
    int                 nmaterials;
} MiptImguiParams;

/* replaces optixLaunch in TracerWindow::run (imgui_test/tracer_window.cpp:101); params is a HOST pointer */
int  mipt_imgui_launch(MiptContext* ctx, void* stream, const MiptImguiParams* params,
                       unsigned int width, unsigned int height, unsigned int depth);
/* Camera::set_fd / Camera::compute_uvw (camera.h:36, 110-123), host side */
void mipt_imgui_camera_set_fd(MiptImguiCamera* camera, float new_fd);
void mipt_imgui_camera_compute_uvw(MiptImguiCamera* camera);

/* ---------------------------------------------------------------------------
 * glTF scenes   (replaces sutil::loadScene + the host data of sutil::Scene,
 * SDK/sutil/Scene.cpp:267-550, Scene.h:53-159; host only, no device needed).
 * Instances are flattened to world-space triangles instead of GAS/IAS builds.
 * ------------------------------------------------------------------------- */
typedef struct MiptScene MiptScene;

typedef struct MiptSceneInfo {
    uint32_t num_buffers, num_images, num_textures, num_materials;
    uint32_t num_meshes, num_instances, num_cameras, num_triangles;   /* triangles after flattening */
    float    aabb_min[3], aabb_max[3];                                 /* union of the instances' world boxes */
} MiptSceneInfo;

int mipt_scene_load(const char* filename, MiptScene** out, char* errbuf, size_t errlen);   /* .gltf (+ .bin / data: URIs) or .glb */
int mipt_scene_destroy(MiptScene* scene);
int mipt_scene_info(const MiptScene* scene, MiptSceneInfo* info);
/* Scene::camera() (Scene.cpp:777-791): first glTF camera (lookat = scene aabb centre, :687-688) or the default camera */
int mipt_scene_camera(const MiptScene* scene, float eye[3], float lookat[3], float up[3], float* fovY_deg, float* aspect);
int mipt_scene_material(const MiptScene* scene, uint32_t index, MiptMaterialData* out);
/* vertices4_out: num_triangles*3 float4 {x,y,z,0}; material_index_out: num_triangles (0xffffffff = none) */
int mipt_scene_flatten(const MiptScene* scene, float* vertices4_out, uint32_t* material_index_out);
/* Per-mesh OBJECT-space triangles and the instance records, for two-level builds (Scene::buildMeshAccels /
 * buildInstanceAccel, Scene.cpp:817-1212): mesh m's triangles in the order mipt_scene_flatten emits them inside one instance
 * (vertices4_out: n*3 float4 {x,y,z,0}; material_index_out nullable); instance i's mesh and row-major 3x4 object->world matrix. */
int mipt_scene_mesh_num_triangles(const MiptScene* scene, uint32_t mesh, uint32_t* n_out);
int mipt_scene_mesh_triangles(const MiptScene* scene, uint32_t mesh, float* vertices4_out, uint32_t* material_index_out);
int mipt_scene_instance(const MiptScene* scene, uint32_t index, uint32_t* mesh_out, float transform12_out[12]);
/* ... and the per-mesh shading attributes in the same order, as the flatten_* functions below hand them out per instance
 * (mipt_whitted_set_geometry_instanced): normals4_out: n*3 float4 {OBJECT-space normal as stored, 1}, all 0 where a primitive has no
 * NORMAL; uv2_out / rgba4_out / present_out as mipt_scene_flatten_texcoords / _colors */
int mipt_scene_mesh_normals(const MiptScene* scene, uint32_t mesh, float* normals4_out);
int mipt_scene_mesh_texcoords(const MiptScene* scene, uint32_t mesh, uint32_t set, float* uv2_out, uint8_t* present_out);
int mipt_scene_mesh_colors(const MiptScene* scene, uint32_t mesh, float* rgba4_out, uint8_t* present_out);
/* normals4_out: num_triangles*3 float4 {world-space normal (M^-T n, un-normalised), 1}; all 0 where a primitive has no NORMAL */
int mipt_scene_flatten_normals(const MiptScene* scene, float* normals4_out);
/* texture i of the scene (glTF "textures"); the texel pointer stays valid until mipt_scene_destroy */
int mipt_scene_texture(const MiptScene* scene, uint32_t index, MiptTextureData* out);
/* TEXCOORD_<set> per corner: uv2_out = num_triangles*3 float2; present_out (nullable) = num_triangles bytes, 0 where
 * the primitive has no such attribute (uv then 0).  Float VEC2 accessors only, as the reference reads them. */
int mipt_scene_flatten_texcoords(const MiptScene* scene, uint32_t set, float* uv2_out, uint8_t* present_out);
/* COLOR_0 per corner: rgba4_out = num_triangles*3 float4 (1,1,1,1 where absent); float VEC4 accessors only */
int mipt_scene_flatten_colors(const MiptScene* scene, float* rgba4_out, uint8_t* present_out);

/* ---------------------------------------------------------------------------
 * Image output   (replaces sutil::saveImage, SDK/sutil/sutil.cpp:542-709; host only)
 * ------------------------------------------------------------------------- */
typedef enum MiptImageFormat {          /* sutil::BufferImageFormat, SDK/sutil/sutil.h:53-58 */
    MIPT_IMAGE_UNSIGNED_BYTE4 = 0,
    MIPT_IMAGE_FLOAT4         = 1,
    MIPT_IMAGE_FLOAT3         = 2
} MiptImageFormat;
/* Chooses the container by the last three characters of `filename`, as the reference does:
 *   ppm: any format; written top row first (buffer row 0 is the bottom row); float channels -> toSRGB unless
 *        disable_srgb_conversion, then int(256 x) clamped to 0..255
 *   png: UNSIGNED_BYTE4 only, RGBA8, flipped vertically;   exr: FLOAT3 / FLOAT4 only, half floats, buffer row order.
 * Unsupported combinations fail with the reference's exception text in errbuf.  data is a HOST pointer. */
int mipt_save_image(const char* filename, const void* data, uint32_t width, uint32_t height, int pixel_format,
                    int disable_srgb_conversion, char* errbuf, size_t errlen);

/* ---------------------------------------------------------------------------
 * Host helpers mirroring sutil (kept API: SDK/sutil/Camera.h:38-76)
 * ------------------------------------------------------------------------- */
/* Camera::UVWFrame, SDK/sutil/Camera.cpp:34-46 (fod = m_fod, the fork's focal scale) */
void mipt_camera_uvw(const float eye[3], const float lookat[3], const float up[3],
                     float fovY_deg, float aspect, float fod, float U[3], float V[3], float W[3]);
/* deterministic synthetic scene of SURVEY.md 8(d) "C4": n triangles in the unit cube,
 * written as 3 float4 {x,y,z,0} per triangle (the reference's Vertex layout) to host memory */
int  mipt_scene_random_triangles(uint32_t n, uint32_t seed_tag, float edge_scale, float* vertices4_out);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H */
