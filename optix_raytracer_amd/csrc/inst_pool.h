// inst_pool.h -- the entry pool over a two-level (instance) structure: pool_round_inst / tail_trace_inst, shared by instances.hip (the
// traversal launches) and kernels.hip (the path kernel over instance structures, k_tail_pool_inst).  Moved out of instances.hip unchanged.
#pragma once
#include "kernels.h"
#include "detmath.h"
#include "wave_util.h"
#include "trav_core.h"
#include "inst_core.h"
#include "pool_core.h"

namespace mipt {

// ---- the entry pool over a two-level structure ----------------------------------------------------------------------------------------
// The same pool as kernels.hip's pool_round (a wave's rays traversed together: each round pops up to 64 entries, whichever rays they
// belong to, one per lane), with 64-bit entries: .x as there (leaf : 1 | payload : 25 | ray slot : 6), .y = 0 for an entry of the top
// level, instance + 1 for an entry inside an instance.  All nodes are BVH4 nodes of ONE array (DeviceBvh::inst_node4_off): the top
// level's mirror, then every mesh's with its references rebased (mipt_accel_build_instances), so an entry inside a mesh looks like an
// entry of a one-level BVH.  A lane's round is one of
//   top-level node:      fetch 64 B + the table of its leaf children's mesh roots, slab-test four boxes against the WORLD ray, push the hit
//                        children: a node with .y = 0, a leaf (an instance) as its mesh's ROOT with .y = instance + 1
//   top-level leaf:      (an instance that came from a lane's own stack: k_trace_tiled_inst) fetch its mesh's root, push it with .y = instance + 1
//   node in an instance: fetch 64 B + the instance's world->object matrix (issued together: the entry names both), take the ray
//                        into object space (inst_core.h's arithmetic: the oracle's), slab-test, push with the same .y
//   triangle:            fetch 48 B + the matrix + prim_base, test in object space, merge t (the same point in both spaces) and the
//                        FLATTENED primitive index into the ray's slot: min t, tie -> lowest index
// Hit records do not depend on the visiting order, so they equal k_trace_inst's and the oracle's bit for bit; visit counters differ
// (counting launches keep k_trace_inst).
#ifndef MIPT_INST_STATS            // probe builds (make probe-inststats PROBE_FLAGS=-DMIPT_INST_STATS=1): per-launch sums in DeviceCounters::dbg
#define MIPT_INST_STATS 0
#endif
#if MIPT_INST_STATS
#define INST_STATS_PARAM , uint32_t* ist
#define INST_STATS_ARG , ist
#else
#define INST_STATS_PARAM
#define INST_STATS_ARG
#endif

// slab_children4 that also says WHICH child each hit is (its index rides in the two low mantissa bits of the sort key: the order
// of near-equal entry distances is free, results do not depend on it)
struct Slab4i { int ref[4]; uint32_t idx[4]; int n; };
__device__ __forceinline__ Slab4i slab_children4i(const u32x4 q0, const u32x4 q1, const u32x4 q2, const u32x4 q3, const float ix, const float iy, const float iz,
                                                  const float nx, const float ny, const float nz, const float tmn, const float tcur)
{
    float t[4]; int r[4];
    slab_piece(q0, ix, iy, iz, nx, ny, nz, tmn, tcur, t[0], r[0]);
    slab_piece(q1, ix, iy, iz, nx, ny, nz, tmn, tcur, t[1], r[1]);
    slab_piece(q2, ix, iy, iz, nx, ny, nz, tmn, tcur, t[2], r[2]);
    slab_piece(q3, ix, iy, iz, nx, ny, nz, tmn, tcur, t[3], r[3]);
    const float inf = __builtin_inff();
    Slab4i o;
    o.n = (t[0] < inf ? 1 : 0) + (t[1] < inf ? 1 : 0) + (t[2] < inf ? 1 : 0) + (t[3] < inf ? 1 : 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = t[c] < inf ? u2f((f2u(t[c]) & ~3u) | (uint32_t)c) : inf;
    cmp_swap(t[0], r[0], t[1], r[1]); cmp_swap(t[2], r[2], t[3], r[3]);
    cmp_swap(t[0], r[0], t[2], r[2]); cmp_swap(t[1], r[1], t[3], r[3]);
    cmp_swap(t[1], r[1], t[2], r[2]);
#pragma unroll
    for (int c = 0; c < 4; ++c) { o.ref[c] = r[c]; o.idx[c] = f2u(t[c]) & 3u; }
    return o;
}

// the object-space ray of instance record pieces m0 .. m2 (rows of the world->object matrix): instances.hip's header, the oracle's sequence
__device__ __forceinline__ RayO to_object(const u32x4 m0, const u32x4 m1, const u32x4 m2, const f3 wo, const f3 wd)
{
    const float inv[12] = { u2f(m0.x), u2f(m0.y), u2f(m0.z), u2f(m0.w), u2f(m1.x), u2f(m1.y), u2f(m1.z), u2f(m1.w), u2f(m2.x), u2f(m2.y), u2f(m2.z), u2f(m2.w) };
    RayO O;
    const f3 m = xform_vec(inv, wo);
    O.o = mk3(m.x + inv[3], m.y + inv[7], m.z + inv[11]);
    O.d = xform_vec(inv, wd);
    prepare(O);
    return O;
}

// CACHE: the object-space ray of each ray slot's CURRENT instance is kept in LDS (xr: per slot a tag = instance + 1, then o', d'): a
// round inside that instance reads it instead of fetching the instance's matrix (3 of the round's 7 loads) and transforming again.
// One entry per slot, replaced on a miss; a slot whose entries alternate between instances misses on each change, no more.
constexpr int kXrWords = 64 * 8;                           // [slot][8]: tag, o'.xyz, d'.xyz, writer
// TWO (the path kernel's two-rays-per-pass form, k_tail_pool2_inst): a slot holds TWO rays -- its closest-hit ray in the owner lane's registers
// (T) and the occlusion ray of the bounce before in LDS (sray, as k_tail_pool2 keeps it); bit 31 of an entry's .y says which one the entry
// belongs to (instance numbers stay below 2^31).  An occlusion entry behaves as MODE 2 (first hit sets sh_done[slot]), a closest-hit
// entry as MODE 0 (sh_best[slot]).
template <int MODE, int CAP, bool CACHE = false, int AH = 0, bool TWO = false>
__device__ __forceinline__ void pool_round_inst(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, uint2* stk, unsigned long long* sh_best,
                                                float* sh_u, float* sh_v, uint32_t* sh_done, const int lane, const Trav& T, int& psize, const bool cull_back INST_STATS_PARAM,
                                                uint32_t* xr = nullptr, const float* sray = nullptr)
{
    int k = psize < 64 ? psize : 64;
    const int room = CAP - 120 - psize;
    if (3 * k > room) k = room > 3 ? room / 3 : 1;
    const bool have = lane < k;
    uint2 ent = make_uint2(0u, 0u);
    if (have) ent = stk[psize - 1 - lane];
    psize -= k;
    uint32_t slot = ent.x & 63u;
    const uint32_t sbit = TWO ? ent.y & 0x80000000u : 0u;      // the entry belongs to the slot's occlusion ray
    const bool is_s = TWO && have && sbit != 0u;
    if (TWO) ent.y &= 0x7fffffffu;
    bool dead = false;
    if (MODE != 0) {                                       // an entry of a ray found occluded in the meantime: take one more from the top
        dead = have && sh_done[slot] != 0u;
        const unsigned long long dm = __ballot(dead);
        if (dm != 0ull && psize > 0) {
            const int nd = __popcll(dm), k2 = nd < psize ? nd : psize;
            const int r = (int)lane_rank(dm);
            if (dead && r < k2) { ent = stk[psize - 1 - r]; slot = ent.x & 63u; dead = sh_done[slot] != 0u; }
            psize -= k2;
        }
    }
    // the WORLD ray of that slot, from its owner's registers
    float ox = shfl_f(T.o.x, slot), oy = shfl_f(T.o.y, slot), oz = shfl_f(T.o.z, slot);
    float dx = shfl_f(T.d.x, slot), dy = shfl_f(T.d.y, slot), dz = shfl_f(T.d.z, slot);
    float ix = shfl_f(T.idir.x, slot), iy = shfl_f(T.idir.y, slot), iz = shfl_f(T.idir.z, slot);
    float nx = shfl_f(T.nood.x, slot), ny = shfl_f(T.nood.y, slot), nz = shfl_f(T.nood.z, slot);
    float tmn = shfl_f(T.tmin, slot), tmx = shfl_f(T.tmax, slot);
    if (TWO) {
        if (__ballot(is_s) != 0ull) {
            if (is_s) {
                ox = sray[0 * 64 + slot]; oy = sray[1 * 64 + slot]; oz = sray[2 * 64 + slot];
                dx = sray[3 * 64 + slot]; dy = sray[4 * 64 + slot]; dz = sray[5 * 64 + slot];
                ix = sray[6 * 64 + slot]; iy = sray[7 * 64 + slot]; iz = sray[8 * 64 + slot];
                nx = sray[9 * 64 + slot]; ny = sray[10 * 64 + slot]; nz = sray[11 * 64 + slot];
                tmn = sray[12 * 64 + slot]; tmx = sray[13 * 64 + slot];
            }
        }
    }
    const uint32_t skp = AH != 0 ? shfl_u(T.skip, slot) : 0xffffffffu;       // continuation key of the any-hit-program kernels (Trav::skip)
    const uint32_t ahk = AH != 0 ? shfl_u(T.ahk, slot) : 1u;                 // ... and how many candidates the ray wants from this traversal
    float tcur = tmx;
    bool live = have;
    if (TWO) {
        if (have) {
            if (is_s) { if (sh_done[slot] != 0u) live = false; }
            else      tcur = ord_float((uint32_t)(sh_best[slot] >> 32));
        }
    }
    else if (MODE == 0) { if (have) tcur = ord_float((uint32_t)(sh_best[(AH != 0 && ahk > 1u ? (kAhLevels - 1) * 64 : 0) + slot] >> 32)); }      // AH: pruned at the last wanted candidate
    else if (dead) live = false;
    const bool leaf = (ent.x >> 31) != 0u;
    const uint32_t payload = (ent.x >> 6) & 0x1ffffffu;
    const uint32_t lfirst = payload >> 2, lcnt = (payload & 3u) + 1u;
    const bool in_inst = ent.y != 0u;
    const bool do_inner = live && !leaf;
    const bool do_tri = live && leaf && in_inst;
    const bool do_enter = live && leaf && !in_inst;        // a top-level leaf from a lane's stack: lfirst is the instance
#if MIPT_INST_STATS
    ist[0] += 1u; ist[1] += (uint32_t)__popcll(__ballot(live)); ist[2] += (uint32_t)__popcll(__ballot(do_inner && !in_inst));
    ist[3] += (uint32_t)__popcll(__ballot(do_inner && in_inst)); ist[4] += (uint32_t)__popcll(__ballot(do_tri)); ist[5] += (uint32_t)psize;
#endif
    const uint32_t inst_rec = bvh.inst_off + (in_inst ? ent.y - 1u : lfirst) * (uint32_t)sizeof(DeviceInstance);
    // ------------------------------------------------------------ this round's loads, all issued before the one wait
    uint32_t off = bvh.inst_node4_off;                     // idle lanes re-read node 0 (a broadcast)
    if (do_inner) off = bvh.inst_node4_off + (payload << 6);
    if (do_tri)   off = bvh.inst_tri_off + lfirst * 48u;
    u32x4 q0 = ldq(blob, off), q1 = ldq(blob, off + 16u), q2 = ldq(blob, off + 32u);
    u32x4 q3 = u32x4{ 0u, 0u, 0u, (uint32_t)kNoChild };
    if (do_inner) q3 = ldq(blob, off + 48u);
    u32x4 m0 = u32x4{ 0u, 0u, 0u, 0u }, m1 = m0, m2 = m0;
    uint32_t rec_word = 0u;                                // prim_base (triangles) or root4 (entering an instance)
    u32x4 r4 = u32x4{ 0u, 0u, 0u, 0u };                    // top-level node: the BVH4 root of the mesh behind each leaf child
    bool xr_hit = false;
    float cx0 = 0.f, cx1 = 0.f, cx2 = 0.f, cx3 = 0.f, cx4 = 0.f, cx5 = 0.f;
    if (CACHE && live && in_inst) {
        const uint4 c0 = *reinterpret_cast<const uint4*>(xr + slot * 8u);
        const uint4 c1 = *reinterpret_cast<const uint4*>(xr + slot * 8u + 4u);
        xr_hit = c0.x == ent.y;
        cx0 = u2f(c0.y); cx1 = u2f(c0.z); cx2 = u2f(c0.w); cx3 = u2f(c1.x); cx4 = u2f(c1.y); cx5 = u2f(c1.z);
    }
    const bool need_m = live && in_inst && !xr_hit;
    if (!CACHE || __ballot(need_m) != 0ull) { if (need_m) { m0 = ldq(blob, inst_rec); m1 = ldq(blob, inst_rec + 16u); m2 = ldq(blob, inst_rec + 32u); } }
    if (do_tri)   rec_word = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(blob, (int)(inst_rec + (uint32_t)offsetof(DeviceInstance, prim_base)), 0, 0);
    if (do_enter) rec_word = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(blob, (int)(inst_rec + (uint32_t)offsetof(DeviceInstance, root4)), 0, 0);
    if (do_inner && !in_inst) r4 = ldq(blob, bvh.inst_root4_off + (payload << 4));
    asm volatile("" ::: "memory");
    if (live && in_inst) {
        RayO O;
        if (CACHE && xr_hit) { O.o = mk3(cx0, cx1, cx2); O.d = mk3(cx3, cx4, cx5); prepare(O); }
        else O = to_object(m0, m1, m2, mk3(ox, oy, oz), mk3(dx, dy, dz));
        ox = O.o.x; oy = O.o.y; oz = O.o.z; dx = O.d.x; dy = O.d.y; dz = O.d.z;
        ix = O.idir.x; iy = O.idir.y; iz = O.idir.z; nx = O.nood.x; ny = O.nood.y; nz = O.nood.z;
    }
    if (CACHE && __ballot(need_m) != 0ull) {
        // a miss replaces the slot's entry: ONE writer per slot (several lanes may hold entries of one ray in different instances)
        if (need_m) xr[slot * 8u + 7u] = (uint32_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (need_m && xr[slot * 8u + 7u] == (uint32_t)lane) {
            *reinterpret_cast<uint4*>(xr + slot * 8u) = make_uint4(ent.y, f2u(ox), f2u(oy), f2u(oz));
            xr[slot * 8u + 4u] = f2u(dx); xr[slot * 8u + 5u] = f2u(dy); xr[slot * 8u + 6u] = f2u(dz);
        }
    }
    uint32_t push0 = 0u, push1 = 0u, push2 = 0u, push3 = 0u;
    uint32_t y0 = ent.y | sbit, y1 = y0, y2 = y0, y3 = y0;
    int npush = 0;
    if (do_inner) {
        float tmn_b = tmn, tcur_b = tcur;                     // the box tests' interval (AH: a little wider, ah_box_interval; nood of the space the entry is in)
        if (AH != 0) ah_box_interval(nx, ny, nz, tmn, tcur, tmn_b, tcur_b);
        const Slab4i b = slab_children4i(q0, q1, q2, q3, ix, iy, iz, nx, ny, nz, tmn_b, tcur_b);
        npush = b.n;                                          // push1 = the nearest (ends on top), push0 the next, then push2, push3
        int ref[4] = { b.ref[0], b.ref[1], b.ref[2], b.ref[3] };
        uint32_t yy[4] = { ent.y, ent.y, ent.y, ent.y };
        if (!in_inst) {
            // a leaf child of the top level is an instance: what is pushed is its mesh's root (from the table beside the node) under
            // the instance's number -- no round of its own for "entering"
#pragma unroll
            for (int c = 0; c < 4; ++c) if (ref[c] < 0) {
                const uint32_t kk = (~(uint32_t)ref[c]) >> 4;
                const uint32_t id = b.idx[c];
                ref[c] = (int)(id == 0u ? r4.x : (id == 1u ? r4.y : (id == 2u ? r4.z : r4.w)));
                yy[c] = kk + 1u;
            }
        }
        if (b.n > 0) { push1 = pool_pack(ref[0], slot); y1 = yy[0] | sbit; }
        if (b.n > 1) { push0 = pool_pack(ref[1], slot); y0 = yy[1] | sbit; }
        if (b.n > 2) { push2 = pool_pack(ref[2], slot); y2 = yy[2] | sbit; }
        if (b.n > 3) { push3 = pool_pack(ref[3], slot); y3 = yy[3] | sbit; }
    }
    if (do_enter) { push1 = pool_pack((int)rec_word, slot); y1 = (lfirst + 1u) | sbit; npush = 1; }
    if (do_tri) {
        const float4 n0 = make_float4(u2f(q0.x), u2f(q0.y), u2f(q0.z), u2f(q0.w)), n1 = make_float4(u2f(q1.x), u2f(q1.y), u2f(q1.z), u2f(q1.w)),
                     n2 = make_float4(u2f(q2.x), u2f(q2.y), u2f(q2.z), u2f(q2.w));
        const f3 rd = mk3(dx, dy, dz);
        const f3 e1 = mk3(n1.x, n1.y, n1.z), e2 = mk3(n2.x, n2.y, n2.z);
        const f3 p = fcross(rd, e2);
        const float det = fdot(e1, p);
        const float inv = 1.0f / det;
        const f3 tv = mk3(ox, oy, oz) - mk3(n0.x, n0.y, n0.z);
        const float u = fdot(tv, p) * inv;
        const f3 q = fcross(tv, e1);
        const float v = fdot(rd, q) * inv;
        const float t = fdot(e2, q) * inv;
        const uint32_t prim = rec_word + f2u(n0.w);           // the flattened index
        const bool ok = det != 0.0f && u >= 0.0f && v >= 0.0f && u + v <= 1.0f && (t > tmn || (AH != 0 && t == tmn && prim > skp)) && t < tmx &&
                        (!cull_back || det > 0.0f || (f2u(n1.w) & 4u) != 0u);
        bool stop = false;
        if (TWO ? is_s : MODE != 0) {
            if (ok) { sh_done[slot] = 1u; stop = true; }
        } else if (AH == 2 && ok && (f2u(n1.w) & 1u) != 0u) {
            // AH 2 (occlusion rays with any-hit programs): a candidate WITHOUT program ends the ray wherever it lies in the interval -- the
            // slot's key drops below every real one (t = -inf: the ray's other entries fail their box tests at once), the owner sees it
#pragma unroll
            for (int l = 0; l < kAhLevels; ++l) atomicMin(&sh_best[l * 64 + slot], kAhOccludedKey);
        } else if (ok) {
            unsigned long long key = ((unsigned long long)ord_bits(t + 0.0f) << 32) | (unsigned long long)prim;
            if (AH != 0) {
                // AH: the slot's kAhLevels least keys, level l the (l + 1)-th: every level takes the minimum of what reaches it and hands
                // the larger key down -- whatever the interleaving with other lanes' candidates, level l ends as the (l + 1)-th least
#pragma unroll
                for (int l = 0; l < kAhLevels; ++l) { const unsigned long long prev = atomicMin(&sh_best[l * 64 + slot], key); key = prev == key ? ~0ull : (prev > key ? prev : key); }     // (prev == key: the SAME candidate through another reference -- a tree with spatial splits -- is there already)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            } else {
                atomicMin(&sh_best[slot], key);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (*(volatile unsigned long long*)&sh_best[slot] == key) { sh_u[slot] = u; sh_v[slot] = v; }
            }
        }
        if (!stop && lcnt > 1u) {
            push1 = 0x80000000u | ((lfirst + 1u) << 8) | ((lcnt - 2u) << 6) | slot;
            npush = 1;
        }
    }
    {
        const unsigned long long m4 = __ballot(npush >= 4);
        if (m4 != 0ull) { if (npush >= 4) stk[psize + (int)lane_rank(m4)] = make_uint2(push3, y3); psize += __popcll(m4); }
        const unsigned long long m3 = __ballot(npush >= 3);
        if (m3 != 0ull) { if (npush >= 3) stk[psize + (int)lane_rank(m3)] = make_uint2(push2, y2); psize += __popcll(m3); }
        const unsigned long long m2_ = __ballot(npush >= 2);
        if (m2_ != 0ull) { if (npush >= 2) stk[psize + (int)lane_rank(m2_)] = make_uint2(push0, y0); psize += __popcll(m2_); }
        const unsigned long long m1_ = __ballot(npush >= 1);
        if (m1_ != 0ull) { if (npush >= 1) stk[psize + (int)lane_rank(m1_)] = make_uint2(push1, y1); psize += __popcll(m1_); }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// a wave's rays (one per lane where has_ray) through the pool from the top level's root; results in sh_best / sh_u / sh_v / sh_done
template <int MODE, int CAP, bool CACHE = false, int AH = 0>
__device__ __forceinline__ void tail_trace_inst(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, uint2* stk, unsigned long long* sh_best,
                                                float* sh_u, float* sh_v, uint32_t* sh_done, const int lane, const Trav& T, const bool has_ray, const bool cull_back INST_STATS_PARAM,
                                                uint32_t* xr = nullptr)
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    sh_best[lane] = ((unsigned long long)ord_bits(T.tbest + 0.0f) << 32) | 0xffffffffull;
    if (AH != 0) { for (int l = 1; l < kAhLevels; ++l) sh_best[l * 64 + lane] = sh_best[lane]; }       // (the levels lie over sh_u / sh_v / sh_done)
    else { sh_u[lane] = 0.0f; sh_v[lane] = 0.0f; sh_done[lane] = 0u; }
    if (CACHE) xr[lane * 8] = 0u;                            // no instance cached for this slot's (new) ray
    const unsigned long long m = __ballot(has_ray);
    if (has_ray) stk[(int)lane_rank(m)] = make_uint2(pool_pack(0, (uint32_t)lane), 0u);
    int psize = __popcll(m);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    while (psize > 0) pool_round_inst<MODE, CAP, CACHE, AH>(blob, bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, T, psize, cull_back INST_STATS_ARG, xr);
}


}  // namespace mipt
