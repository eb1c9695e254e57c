// instances.hip -- traversal of a two-level (instance) acceleration structure: what optixTraverse does over the IAS that
// sutil::Scene::buildInstanceAccel builds (SDK/sutil/Scene.cpp:1134-1212: one OptixInstance per glTF node with a mesh, row-major
// 3x4 object->world transform, the mesh's GAS as child).  Build-defined arithmetic, identical sequence in oracle/pt_oracle.c
// (inst_traverse / gas_traverse):
//   * the world ray walks a BVH2 over the instances' world boxes (the same 64-byte nodes, near child first);
//   * at a leaf, for each listed instance in turn, the ray is taken into object space -- o' = M o + t, d' = M d with the exported
//     world->object matrix, every row fmaf(m2, z, fmaf(m1, y, m0 * x)) (+ m3) -- WITHOUT re-normalising d', so a distance t is
//     the same point in both spaces and tmin / tmax / the running best carry over;
//   * the mesh's own BVH2 is walked to its end (same slab test, same Moeller-Trumbore test as the one-level kernels);
//   * a hit reports the FLATTENED primitive index prim_base + local index (the order of mipt_scene_flatten), ties at equal t go
//     to the lower flattened index.
// k_trace_inst: one ray per thread, one LDS stack of 64 entries per lane shared by both levels ([entry][lane]: bank = lane) -- the
// straightforward form (the per-ray visiting order of the oracle, exact visit counters; counting launches and structures whose
// meshes have no BVH4 mirror).  k_trace_static_inst / tail_trace_inst: the entry pool of kernels.hip over both levels (round 3).
#include <cstdlib>

#include "kernels.h"
#include "detmath.h"
#include "wave_util.h"
#include "trav_core.h"
#include "inst_core.h"
#include "pool_core.h"
#include "inst_pool.h"

namespace mipt {

namespace {

struct Best { float t, u, v; uint32_t prim; };

// one mesh BVH with the object-space ray; candidates enter the WORLD record with their flattened index.  Returns true when an
// any-hit query may stop.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool walk_mesh(const DeviceInstance& in, const RayO& r, float tmin, float tmax, Best& best, const Stack2& stk, int& sp,
                                          bool cull_back, uint32_t& n_visits, uint32_t& n_tests)
{
    const int base = sp;
    int node = 0;
    while (node != kSentinel) {
        while (node >= 0) {
            if (COUNT) ++n_visits;
            node = step_inner(in.nodes, node, r, tmin, best.t, stk, sp, base);
        }
        if (node == kSentinel) break;
        const uint32_t desc = ~(uint32_t)node, first = desc >> 4, cnt = desc & 15u;
        for (uint32_t i = 0; i < cnt; ++i) {
            const float4* tp = in.tris + (size_t)(first + i) * 3u;
            const float4 a = tp[0], b = tp[1], c = tp[2];
            if (COUNT) ++n_tests;
            const f3 e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
            const f3 p = fcross(r.d, e2);
            const float det = fdot(e1, p);
            if (det == 0.0f) continue;
            if (cull_back && det < 0.0f && (f2u(b.w) & 4u) == 0u) continue;
            const float inv = 1.0f / det;
            const f3 tv = r.o - mk3(a.x, a.y, a.z);
            const float u = fdot(tv, p) * inv;
            const f3 q = fcross(tv, e1);
            const float v = fdot(r.d, q) * inv;
            const float t = fdot(e2, q) * inv;
            if (!(u >= 0.0f && v >= 0.0f && u + v <= 1.0f)) continue;
            if (!(t > tmin && t < tmax)) continue;
            const uint32_t gp = in.prim_base + f2u(a.w);
            if (ANY) { best.t = t; best.prim = gp; sp = base; return true; }
            if (best.prim == 0xffffffffu || t < best.t || (t == best.t && gp < best.prim)) { best.t = t; best.prim = gp; best.u = u; best.v = v; }
        }
        node = stk.pop(sp, base);
    }
    return false;
}

template <int MODE, bool COUNT>
__global__ __launch_bounds__(64) void k_trace_inst(TraceArgs a)
{
    __shared__ uint32_t stk_lds[kInstStack * 64];
    const int lane = (int)threadIdx.x;
    const Stack2 stk{ stk_lds, lane };
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    const bool cull_back = MODE == 0 && (a.ray_flags & 16u) != 0u;
    uint32_t n_visits = 0, n_tests = 0;
    for (uint32_t ray = blockIdx.x * 64u + threadIdx.x; ray < n; ray += gridDim.x * 64u) {
        const float4 ro = a.ray_o[(size_t)ray * a.stride], rd = a.ray_d[(size_t)ray * a.stride];
        RayO W;
        W.o = mk3(ro.x, ro.y, ro.z); W.d = mk3(rd.x, rd.y, rd.z);
        prepare(W);
        const float tmin = ro.w, tmax = rd.w;
        Best best{ tmax, 0.0f, 0.0f, 0xffffffffu };
        int sp = 0;
        int node = tmax > tmin ? 0 : kSentinel;         // empty interval: a miss without touching the structure
        bool stop = false;
        while (node != kSentinel && !stop) {
            while (node >= 0) {
                if (COUNT) ++n_visits;
                node = step_inner(a.bvh.nodes, node, W, tmin, best.t, stk, sp, 0);
            }
            if (node == kSentinel) break;
            const uint32_t desc = ~(uint32_t)node, first = desc >> 4, cnt = desc & 15u;
            for (uint32_t i = 0; i < cnt && !stop; ++i) {
                const DeviceInstance& in = a.bvh.inst[a.bvh.inst_order[first + i]];
                RayO O;
                const f3 m = xform_vec(in.inv, W.o);
                O.o = mk3(m.x + in.inv[3], m.y + in.inv[7], m.z + in.inv[11]);
                O.d = xform_vec(in.inv, W.d);
                prepare(O);
                stop = walk_mesh<MODE != 0, COUNT>(in, O, tmin, tmax, best, stk, sp, cull_back, n_visits, n_tests);
            }
            if (!stop) node = stk.pop(sp, 0);
        }
        Trav T;
        T.tbest = best.t; T.prim = best.prim; T.u = best.u; T.v = best.v;
        finish_ray<MODE>(a, T, ray, MODE == 0 ? best.prim != 0xffffffffu : stop);
    }
    if (COUNT) {
        wave_add64(&a.ctr->node_visits[MODE != 0], n_visits);
        wave_add64(&a.ctr->tri_tests[MODE != 0], n_tests);
    }
}


// static chunks of `ppw` rays per wave, all on the pool from the root (kernels.hip's k_trace_static over two levels)
#ifndef MIPT_INST_XR_CACHE
#define MIPT_INST_XR_CACHE 0                                // 1 (probe builds): the LDS cache of object-space rays above.  Measured SLOWER: MeshViewer over 151 /
                                                            // 2 001 instances 0.942 / 1.559 ms against 0.882 / 1.437 ms without -- its 2 KB of LDS and ~10 VGPRs cost
                                                            // a wave per SIMD, its reads sit in front of the round's loads, and a ray's entries alternate instances
#endif
#ifndef MIPT_INST_ROWS
#define MIPT_INST_ROWS 24                                   // 6 KB of LDS per wave: 576 entries of 8 bytes + the merge state
#endif
constexpr int kInstRows = MIPT_INST_ROWS;
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kInstRows <= 24 ? 6 : 5, kInstRows <= 24 ? 6 : 5))) void k_trace_static_inst(TraceArgs a, uint32_t ppw)
{
    constexpr int kCap = (kInstRows * 64 - 384) / 2;
    constexpr bool CACHE = MIPT_INST_XR_CACHE != 0;
    __shared__ __attribute__((aligned(16))) uint2 stk2[kInstRows * 32 + (CACHE ? kXrWords / 2 : 0)];
    uint32_t* const words = reinterpret_cast<uint32_t*>(stk2);
    uint32_t* const xr = words + kInstRows * 64;
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(words + 2 * kCap);
    float* const sh_u = reinterpret_cast<float*>(words + 2 * kCap + 128);
    float* const sh_v = reinterpret_cast<float*>(words + 2 * kCap + 192);
    uint32_t* const sh_done = words + 2 * kCap + 320;
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.inst_blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    const uint32_t chunks = (n + ppw - 1u) / ppw;
    const bool cull_back = MODE == 0 && (a.ray_flags & 16u) != 0u;
#if MIPT_INST_STATS
    uint32_t ist[6] = { 0u, 0u, 0u, 0u, 0u, 0u };
#endif
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint32_t ray = chunk * ppw + (uint32_t)lane;
        const bool mine = (uint32_t)lane < ppw && ray < n;
        Trav T;
        T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f;
        T.o = mk3(0.f, 0.f, 0.f); T.d = T.o; T.idir = T.o; T.nood = T.o; T.tmin = 0.0f; T.tmax = 0.0f;
        if (mine) trav_init(T, a.ray_o[(size_t)ray * a.stride], a.ray_d[(size_t)ray * a.stride]);
        tail_trace_inst<MODE, kCap, CACHE>(blob, a.bvh, stk2, sh_best, sh_u, sh_v, sh_done, lane, T, mine && T.node != kSentinel, cull_back INST_STATS_ARG, xr);
        if (mine) {
            if (MODE == 0) {
                const unsigned long long key = sh_best[lane];
                T.tbest = ord_float((uint32_t)(key >> 32)); T.prim = (uint32_t)key; T.u = sh_u[lane]; T.v = sh_v[lane];
            }
            finish_ray<MODE>(a, T, ray, MODE == 0 ? T.prim != 0xffffffffu : sh_done[lane] != 0u);
        }
    }
#if MIPT_INST_STATS
    if (lane == 0) for (int k = 0; k < 6; ++k) atomicAdd(&a.ctr->dbg[k], (unsigned long long)ist[k]);      // rounds, live lanes, top-level nodes, instance nodes, triangles, pool fill
#endif
}

// ---- tiles of 64 rays, one ray per lane, then the pool: kernels.hip's k_trace_pool over two levels ------------------------------------
// Bulk phase: every lane walks its own ray depth first, near child first, with its stack in LDS ([entry][lane]) shared by both levels,
// and keeps the ray of the space it is in -- world, or the object space of the instance it has entered -- in its registers: the
// transform is applied ONCE per instance entered (the pool above applies it in every round, and fetches the matrix for it: with the
// memory pipeline as the limiter that is what made the pool-only form 2-4 x slower than a flattened tree on scenes of many
// overlapping instances), and a hit found in a near instance prunes the ones behind it before they are opened.  One unified step per
// iteration, as kernels.hip's stream_step: each lane fetches 64 bytes -- its BVH4 node, its next triangle, or (standing at a top-level
// leaf) the instance's record -- then, after one wait, does its slab test / triangle test / transform.  A lane leaves an instance
// when its stack is back at the height it entered with; the world ray is kept in six registers and its reciprocals recomputed.
// Tail phase: when half the tile is finished the remaining stack entries go into the pool above (64-bit entries tagged with the
// instance they belong to) and all 64 lanes work on them.
constexpr int kTiledInstRows = 32;
template <int MODE>
__device__ __forceinline__ void inst_step(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, uint32_t* stk, const int lane, const bool has_ray, Trav& T,
                                          f3& wo, f3& wd, uint32_t& cur, int& base_sp, uint32_t& pbase, const bool cull_back)
{
    const bool act = has_ray && T.node != kSentinel;
    const bool inner = act && T.node >= 0;
    const bool leafref = act && T.node < 0;
    const bool enter = leafref && cur == 0u;               // a top-level leaf: an instance
    const bool at_tri = leafref && cur != 0u;
    const uint32_t desc = ~(uint32_t)T.node;
    const uint32_t lfirst = desc >> 4, lcnt = desc & 15u;
    uint32_t off = bvh.inst_node4_off, off3 = off + 48u;
    if (inner)  { off = bvh.inst_node4_off + ((uint32_t)T.node << 6); off3 = off + 48u; }
    if (at_tri) { off = bvh.inst_tri_off + lfirst * 48u; off3 = off + 32u; }
    if (enter)  { off = bvh.inst_off + lfirst * (uint32_t)sizeof(DeviceInstance); off3 = off + (uint32_t)offsetof(DeviceInstance, prim_base); }
    const u32x4 q0 = ldq(blob, off), q1 = ldq(blob, off + 16u), q2 = ldq(blob, off + 32u);
    u32x4 q3 = u32x4{ 0u, 0u, 0u, (uint32_t)kNoChild };
    if (inner || enter) q3 = ldq(blob, off3);
    asm volatile("" ::: "memory");
    bool need_pop = false;
    if (inner) {
        const Slab4 b = slab_children4(q0, q1, q2, q3, T.idir.x, T.idir.y, T.idir.z, T.nood.x, T.nood.y, T.nood.z, T.tmin, T.tbest);
        if (b.n == 0) need_pop = true;
        else {
            if (b.n > 3) { stk[T.sp * 64 + lane] = (uint32_t)b.ref[3]; ++T.sp; }          // farthest first, the nearest is walked next
            if (b.n > 2) { stk[T.sp * 64 + lane] = (uint32_t)b.ref[2]; ++T.sp; }
            if (b.n > 1) { stk[T.sp * 64 + lane] = (uint32_t)b.ref[1]; ++T.sp; }
            T.node = b.ref[0];
        }
    }
    if (at_tri) {
        bool stop = false;
        if (lcnt != 0u) {
            const float4 ta = make_float4(u2f(q0.x), u2f(q0.y), u2f(q0.z), u2f(q0.w)), tb = make_float4(u2f(q1.x), u2f(q1.y), u2f(q1.z), u2f(q1.w)),
                         tc = make_float4(u2f(q2.x), u2f(q2.y), u2f(q2.z), u2f(q2.w));
            const f3 e1 = mk3(tb.x, tb.y, tb.z), e2 = mk3(tc.x, tc.y, tc.z);
            const f3 p = fcross(T.d, e2);
            const float det = fdot(e1, p);
            const float inv = 1.0f / det;
            const f3 tv = T.o - mk3(ta.x, ta.y, ta.z);
            const float u = fdot(tv, p) * inv;
            const f3 q = fcross(tv, e1);
            const float v = fdot(T.d, q) * inv;
            const float t = fdot(e2, q) * inv;
            const uint32_t prim = pbase + f2u(ta.w);          // the flattened index
            const bool ok = det != 0.0f && u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > T.tmin && t < T.tmax &&
                            (!cull_back || det > 0.0f || (f2u(tb.w) & 4u) != 0u);
            if (MODE != 0) {
                stop = ok;
                if (ok) { T.tbest = t; T.prim = prim; }
            } else if (ok && (t < T.tbest || (t == T.tbest && prim < T.prim))) {
                T.tbest = t; T.prim = prim; T.u = u; T.v = v;
            }
        }
        if (stop) T.node = kSentinel;
        else if (lcnt > 1u) T.node = ~(int)(((lfirst + 1u) << 4) | (lcnt - 1u));
        else need_pop = true;
    }
    if (__ballot(enter) != 0ull) {
        if (enter) {
            const RayO O = to_object(q0, q1, q2, wo, wd);
            T.o = O.o; T.d = O.d; T.idir = O.idir; T.nood = O.nood;
            pbase = q3.x; cur = lfirst + 1u; base_sp = T.sp;
            T.node = (int)q3.y;                               // the mesh's BVH4 root
        }
    }
    if (need_pop) {
        if (cur != 0u && T.sp == base_sp) {                   // the instance is exhausted: back to the world ray
            RayO W; W.o = wo; W.d = wd; prepare(W);
            T.o = wo; T.d = wd; T.idir = W.idir; T.nood = W.nood;
            cur = 0u;
        }
        if (T.sp == 0) T.node = kSentinel;
        else { --T.sp; T.node = (int)stk[T.sp * 64 + lane]; }
    }
}

#ifndef MIPT_INST_TILED_WAVES
#define MIPT_INST_TILED_WAVES 5
#endif
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MIPT_INST_TILED_WAVES, MIPT_INST_TILED_WAVES))) void k_trace_tiled_inst(TraceArgs a)
{
    constexpr int kWords = kTiledInstRows * 64;
    constexpr int kCap = (kWords - 384) / 2;
    __shared__ __attribute__((aligned(16))) uint32_t lds[kWords];                  // (uint2 entries, 64-bit LDS atomics on sh_best)
    uint2* const stk2 = reinterpret_cast<uint2*>(lds);
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(lds + 2 * kCap);
    float* const sh_u = reinterpret_cast<float*>(lds + 2 * kCap + 128);
    float* const sh_v = reinterpret_cast<float*>(lds + 2 * kCap + 192);
    uint32_t* const sh_done = lds + 2 * kCap + 320;
    const int lane = (int)threadIdx.x;
    const bool cull_back = MODE == 0 && (a.ray_flags & 16u) != 0u;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.inst_blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    // a lane's pool entries take two words where its stack entries took one: the rows are rewritten in place, so at most 32 lanes may
    // still be busy at the switch (row e is read before (e + 1) * 32 entries = (e + 1) * 64 words are written)
    // ... and the pool must hold what they bring: a busy lane has at most stack4 entries on its stack + the one it stands at, and
    // pool_round_inst keeps psize <= kCap - 120 only if it starts there (the one-level kernel's steal_floor, kernels.hip)
    const int busy_max = (kCap - 120) / (int)(a.bvh.stack4 + 1u);
    int steal_min = a.steal_min < 32 ? 32 : (a.steal_min < 64 ? a.steal_min : 64);
    if (steal_min < 64 - busy_max) steal_min = 64 - busy_max;
    const uint32_t tiles = (n + 63u) >> 6, tiles_seg = (tiles + kNumSegments - 1) / kNumSegments;
    const uint32_t xcd = blockIdx.x & (kNumSegments - 1), waves_seg = gridDim.x / kNumSegments;
#if MIPT_INST_STATS
    uint32_t ist[6] = { 0u, 0u, 0u, 0u, 0u, 0u };
#endif
    for (uint32_t k = blockIdx.x / kNumSegments;; k += waves_seg) {
        uint32_t tile = xcd * tiles_seg + k;                 // groups of tile_group consecutive tiles dealt to the XCDs in turn (k_trace_pool)
        if (a.tile_group != 0u) {
            const uint32_t gq = k / a.tile_group, r = k - gq * a.tile_group;
            tile = (gq * kNumSegments + xcd) * a.tile_group + r;
        } else if (k >= tiles_seg) break;
        if (tile >= tiles) break;
        const uint32_t ray = tile * 64u + (uint32_t)lane;
        Trav T;
        T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f;
        T.o = mk3(0.f, 0.f, 0.f); T.d = T.o; T.idir = T.o; T.nood = T.o; T.tmin = 0.0f; T.tmax = 0.0f;
        bool has_ray = false;
        if (ray < n) {
            trav_init(T, a.ray_o[(size_t)ray * a.stride], a.ray_d[(size_t)ray * a.stride]);
            has_ray = T.node != kSentinel;
            if (!has_ray) finish_ray<MODE>(a, T, ray, false);          // empty interval: a miss, nothing is touched
        }
        f3 wo = T.o, wd = T.d;
        uint32_t cur = 0u, pbase = 0u;
        int base_sp = 0;
        // ============================================================ bulk: one ray per lane
        while (__popcll(__ballot(!has_ray)) < steal_min) {
            for (int bi = 0; bi < a.burst_bulk; ++bi) {
                if (bi != 0 && __ballot(has_ray && T.node != kSentinel) == 0ull) break;
                inst_step<MODE>(blob, a.bvh, lds, lane, has_ray, T, wo, wd, cur, base_sp, pbase, cull_back);
#if MIPT_INST_STATS
                ist[0] += 1u; ist[1] += (uint32_t)__popcll(__ballot(has_ray));
#endif
            }
            const bool done = has_ray && T.node == kSentinel;
            if (done) { finish_ray<MODE>(a, T, ray, T.prim != 0xffffffffu); has_ray = false; }
        }
        if (__ballot(has_ray) == 0ull) continue;
        // ============================================================ tail: the wave's entry pool
        // the per-lane stacks become tagged pool entries, row by row and in place: an entry above the height the lane entered its
        // instance with belongs to that instance, the ones below to the top level
        int psize = 0;
        for (int e = 0; e < kTiledInstRows; ++e) {
            const bool in = has_ray && T.sp > e;
            const unsigned long long m = __ballot(in);
            if (m == 0ull) break;
            uint32_t v = 0u;
            if (in) v = lds[e * 64 + lane];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (in) stk2[psize + (int)lane_rank(m)] = make_uint2(pool_pack((int)v, (uint32_t)lane), (cur != 0u && e >= base_sp) ? cur : 0u);
            psize += __popcll(m);
        }
        {
            const unsigned long long m = __ballot(has_ray);           // the node (or triangle, or instance) each busy lane stands at
            if (has_ray) stk2[psize + (int)lane_rank(m)] = make_uint2(pool_pack(T.node, (uint32_t)lane), cur);
            psize += __popcll(m);
        }
        // the pool reads the WORLD ray from its owner's registers
        if (cur != 0u) { RayO W; W.o = wo; W.d = wd; prepare(W); T.o = wo; T.d = wd; T.idir = W.idir; T.nood = W.nood; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        sh_best[lane] = ((unsigned long long)ord_bits(T.tbest + 0.0f) << 32) | (unsigned long long)T.prim;
        sh_u[lane] = T.u; sh_v[lane] = T.v; sh_done[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        while (psize > 0) pool_round_inst<MODE, kCap>(blob, a.bvh, stk2, sh_best, sh_u, sh_v, sh_done, lane, T, psize, cull_back INST_STATS_ARG);
        if (has_ray) {
            const unsigned long long key = sh_best[lane];
            T.tbest = ord_float((uint32_t)(key >> 32)); T.prim = (uint32_t)key; T.u = sh_u[lane]; T.v = sh_v[lane];
            finish_ray<MODE>(a, T, ray, MODE == 0 ? T.prim != 0xffffffffu : sh_done[lane] != 0u);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // the next tile's stacks overwrite the merge slots
    }
#if MIPT_INST_STATS
    if (lane == 0) for (int k = 0; k < 6; ++k) atomicAdd(&a.ctr->dbg[k], (unsigned long long)ist[k]);
#endif
}

template <int MODE>
void launch_mode(hipStream_t st, const TraceArgs& a, bool count, const TraceOpts& o)
{
    // o.inst_form (env MIPT_INST_FORM at context creation): 0 = by launch size, 1 = the pool-only form for every launch, 2 = one ray per lane
    if (a.bvh.inst_fast && !count && o.persistent == 3 && o.inst_form != 2) {
        // rays per wave as launch_trace has them for one-level BVHs (kernels.hip): 64 for big any-hit launches, 32 / 16 / 8 for smaller ones
        // (any hit, 2 M shadow rays of the MeshViewer frame over 151 / 2 001 instances: 16 / 32 / 48 / 64 rays per wave give frames of
        //  1.62 / 1.29 / 1.20 / 1.18 and 2.43 / 1.94 / 1.79 / 1.77 ms)
        const uint32_t cus = (uint32_t)(o.num_cus > 0 ? o.num_cus : 256);
        const uint32_t n_est = MODE != 0 && a.n_ptr != nullptr ? a.n_max / 4u : a.n_max;
        const uint32_t q = n_est / (cus * 16u);
        const uint32_t ppw = MODE != 0 ? (n_est > cus * 2048u ? 64u : (q >= 24u ? 32u : (q >= 12u ? 16u : 8u))) : (q >= 96u ? 32u : (q >= 12u ? 16u : 8u));
        unsigned long long want = ((unsigned long long)n_est + ppw - 1u) / ppw;
        want = want < cus * 16ull ? cus * 16ull : (want > cus * 256ull ? cus * 256ull : want);
        const uint32_t g = (uint32_t)((want + kNumSegments - 1) / kNumSegments * kNumSegments);
        // big CLOSEST-hit launches: tiles of 64 rays, one ray per lane first (k_trace_tiled_inst), when a lane's stack over both levels
        // fits its rows.  Any-hit launches keep the pool-only form (as over one-level BVHs: occlusion rays start at scattered points
        // and the per-lane phase buys them nothing -- 151 instances: 376 against 454 us; 2 001: 762 against 692 us in round 3; round 4, whole MeshViewer
        // frames with the tiled form for the occlusion rays too: 150 / 500 / 1 000 / 2 000 instances 0.927 / 1.153 / 1.335 / 1.465 ms against 0.834 / 1.051 / 1.299 / 1.377).
        if (o.inst_form == 0 && MODE == 0 && a.bvh.stack4 + 1u <= (uint32_t)kTiledInstRows && n_est > cus * 512u) {
            TraceArgs b = a;
            b.tile_group = o.tile_group > 0 ? (uint32_t)o.tile_group : 0u;
            unsigned long long tw = ((unsigned long long)a.n_max + 63ull) / 64ull;
            tw = tw < cus * 16ull ? cus * 16ull : (tw > cus * 128ull ? cus * 128ull : tw);
            const uint32_t tg = (uint32_t)((tw + kNumSegments - 1) / kNumSegments * kNumSegments);
            hipLaunchKernelGGL((k_trace_tiled_inst<MODE>), dim3(tg), dim3(64), 0, st, b);
            return;
        }
        hipLaunchKernelGGL((k_trace_static_inst<MODE>), dim3(g), dim3(64), 0, st, a, ppw);
        return;
    }
    uint32_t blocks = (a.n_max + 63u) / 64u;
    if (blocks > 16384u) blocks = 16384u;
    if (count) hipLaunchKernelGGL((k_trace_inst<MODE, true>), dim3(blocks), dim3(64), 0, st, a);
    else       hipLaunchKernelGGL((k_trace_inst<MODE, false>), dim3(blocks), dim3(64), 0, st, a);
}

}  // namespace

void launch_trace_inst(hipStream_t st, const TraceArgs& a, int mode, bool count, const TraceOpts& o)
{
    if (mode == 0) launch_mode<0>(st, a, count, o);
    else if (mode == 1) launch_mode<1>(st, a, count, o);
    else launch_mode<2>(st, a, count, o);
}

}  // namespace mipt
