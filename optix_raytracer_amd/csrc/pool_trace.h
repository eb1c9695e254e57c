// pool_trace.h -- the entry-pool traversal of a one-level BVH: a step's loads (load_step), one round of a wave's pool (pool_round)
// and a chunk of rays traversed from the root (tail_trace).  Moved out of kernels.hip unchanged (round 3) so that the Whitted
// any-hit-program kernels (whitted_kernels.hip) traverse with the same code; AH = 1 / 2 adds their continuation key (below).
#pragma once
#include "kernels.h"
#include "detmath.h"
#include "wave_util.h"
#include "trav_core.h"
#include "pool_core.h"

namespace mipt {

// this step's 64 (fp32 nodes) or 32 / 48 (packed nodes / triangles) bytes of a lane: inner lanes fetch their node, leaf
// lanes their next triangle, idle lanes re-read node 0 (a broadcast).  The memory pipeline (TA / TD) spends 16 cycles per
// dwordx4 wave instruction whatever the exec mask (profiles/r02_l1_*), so the packed form's third piece is fetched by leaf
// lanes only and skipped by the wave when no lane stands at a leaf.
// ---- sensitivity probes (tools/probe_bound.sh; never defined in the product build) -------------------------------------------------
// MIPT_PROBE_VMEM = n: n more 16-byte loads per step (of the lane's own first piece: L1 hits) -- what a step costs in the
// vector-memory pipeline.  MIPT_PROBE_VALU = n: n more independent VALU instructions per step -- what it costs in issue slots.
// The time a probe adds per instruction, against the kernel's time per instruction, says which resource the kernel is bound by.
#ifndef MIPT_PROBE_VMEM
#define MIPT_PROBE_VMEM 0
#endif
#ifndef MIPT_PROBE_VALU
#define MIPT_PROBE_VALU 0
#endif
// MIPT_PROBE_STAMP: s_memtime stamps around a step's loads (issue -> data back) in k_trace_pool; per wave: cycles waited, cycles
// in the rest of the step, steps; summed into DeviceCounters::dbg (tools/probe_bound.sh prints them)
#ifndef MIPT_PROBE_STAMP
#define MIPT_PROBE_STAMP 0
#endif
// MIPT_PROBE_WAVES: per wave of k_trace_pool {lifetime, cycles in the bulk phase, bulk steps, pool rounds} and its start / end on
// the device-wide 100 MHz counter, into the buffer whose device address the host put into DeviceCounters::dbg[0]
// (mipt_trace_closest, env MIPT_PROBE_WAVES; tools/probe_waves.py): five SMEM stamps per wave
#ifndef MIPT_PROBE_WAVES
#define MIPT_PROBE_WAVES 0
#endif
#if MIPT_PROBE_STAMP
#define PROBE_PARAM , unsigned long long& pw_wait
#define PROBE_ARG , pw_wait
#else
#define PROBE_PARAM
#define PROBE_ARG
#endif
__device__ __forceinline__ float probe_valu(float seed)
{
    float a0 = seed, a1 = seed + 1.0f, a2 = seed + 2.0f, a3 = seed + 3.0f;
#pragma unroll
    for (int k = 0; k < MIPT_PROBE_VALU / 4; ++k) {
        a0 = __builtin_fmaf(a0, 1.0000001f, 1e-30f); a1 = __builtin_fmaf(a1, 1.0000002f, 1e-30f);
        a2 = __builtin_fmaf(a2, 1.0000003f, 1e-30f); a3 = __builtin_fmaf(a3, 1.0000004f, 1e-30f);
    }
    return (a0 + a1) + (a2 + a3);
}

// ---- experiment MIPT_TOP_CACHE = 1 (probe builds): the first levels of the packed mirror in the wave's LDS -----------------------------
// 63 nodes in breadth-first order (k_build_top), their references to each other flagged (kTopFlag | position), copied into LDS behind
// the stacks by every wave; a lane at such a node reads it with two ds_read_b128 instead of two buffer loads, and the wave issues no
// vector-memory instruction in a step whose lanes all stand in the top.
#ifndef MIPT_TOP_CACHE
#define MIPT_TOP_CACHE 0
#endif
constexpr uint32_t kTopFlag = 0x40000000u, kTopNodes = 64u, kTopWords = kTopNodes * 8u + kTopNodes;     // nodes (8 words each), then their real indices

template <int Q, bool TOP = false>
__device__ __forceinline__ void load_step(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, const bool inner_in, const bool tri,
                                          const uint32_t node, const uint32_t lfirst, u32x4& n0, u32x4& n1, u32x4& n2, u32x4& n3 PROBE_PARAM,
                                          const uint32_t* top_lds = nullptr)
{
#if MIPT_PROBE_STAMP
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
    const bool in_top = TOP && inner_in && (node & kTopFlag) != 0u;
    const bool inner = inner_in && !in_top;
    if (TOP && Q == 1) {
        n0 = u32x4{ 0u, 0u, 0u, 0u }; n1 = n0; n2 = n0; n3 = n0;
        if (__ballot(inner || tri) != 0ull) {
            uint32_t off = bvh.node32_off;
            if (inner) off = bvh.node32_off + (node << 5);
            if (tri)   off = bvh.tri_byte_off + lfirst * 48u;
            const u32x4 g0 = ldq(blob, off), g1 = ldq(blob, off + 16u);
            if (!in_top) { n0 = g0; n1 = g1; }
            if (tri) n2 = ldq(blob, off + 32u);
        }
        if (__ballot(in_top) != 0ull) {
            if (in_top) {
                const uint4* p = reinterpret_cast<const uint4*>(top_lds + (node & (kTopNodes - 1u)) * 8u);
                const uint4 a0 = p[0], a1 = p[1];
                n0 = u32x4{ a0.x, a0.y, a0.z, a0.w }; n1 = u32x4{ a1.x, a1.y, a1.z, a1.w };
            }
        }
        asm volatile("" ::: "memory");
        return;
    }
    uint32_t off = Q == 2 ? bvh.node4_off : (Q == 1 ? bvh.node32_off : 0u);
    if (inner) off = Q == 2 ? bvh.node4_off + (node << 6) : (Q == 1 ? bvh.node32_off + (node << 5) : node << 6);
    if (tri)   off = bvh.tri_byte_off + lfirst * 48u;
    n0 = ldq(blob, off); n1 = ldq(blob, off + 16u);
    if (Q == 1) {
        n2 = u32x4{ 0u, 0u, 0u, 0u }; n3 = n2;
        if (tri) n2 = ldq(blob, off + 32u);
    } else if (Q == 2) {
        n2 = ldq(blob, off + 32u);                          // third child, or the triangle's third piece
        n3 = u32x4{ 0u, 0u, 0u, (uint32_t)kNoChild };
        if (inner) n3 = ldq(blob, off + 48u);
    } else {
        n2 = ldq(blob, off + 32u); n3 = ldq(blob, off + 48u);
    }
#if MIPT_PROBE_VMEM > 0
    {
        u32x4 acc = u32x4{ 0u, 0u, 0u, 0u };
#pragma unroll
        for (int k = 0; k < MIPT_PROBE_VMEM; ++k) {
            uint32_t o2 = off + 16u * (uint32_t)(k & 1);
            asm volatile("" : "+v"(o2));                   // opaque: not merged with the step's own loads
            const u32x4 t = ldq(blob, o2); acc.x ^= t.x; acc.y ^= t.y; acc.z ^= t.z; acc.w ^= t.w;
        }
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && n0.x == 0x12345678u) n1.x ^= 1u;     // keeps the loads alive, never changes a result
    }
#endif
#if MIPT_PROBE_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pw_wait += __builtin_amdgcn_s_memtime() - pt0;
#endif
    asm volatile("" ::: "memory");
}


// One round of a wave's entry pool (k_trace_pool's tail phase, k_tail_pool): pop up to 64 packed (entry, ray slot) words,
// each lane processes one -- an inner node: slab-test both children against the ray's current best t and push the
// survivors, nearer on top; a leaf reference: test its first triangle and push the rest of the leaf.  The ray of slot s is
// read from the registers of lane s; its best hit is merged in LDS (min t, tie -> lowest primitive).
template <int MODE, int CAP, int Q, int AH = 0>
__device__ __forceinline__ void pool_round(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, uint32_t* stk,
                                           unsigned long long* sh_best, float* sh_u, float* sh_v, uint32_t* sh_done,
                                           const int lane, const Trav& T, int& psize, const bool cull_back PROBE_PARAM)
{
    constexpr int kPoolCap = CAP;
    // ------------------------------------------------------------ pop up to 64 entries (each lane adds at most one net entry)
    int k = psize < 64 ? psize : 64;
    const int room = kPoolCap - (Q == 2 ? 120 : 104) - psize;  // 64 for ray starts, 40 (BVH4: 56) for a single-lane descent
    if (Q == 2) { if (3 * k > room) k = room > 3 ? room / 3 : 1; }      // a popped BVH4 node may push four entries
    else        { if (k > room) k = room > 1 ? room : 1; }
    const bool have = lane < k;
    uint32_t ent = 0u;
    if (have) ent = stk[psize - 1 - lane];
    psize -= k;
    uint32_t slot = ent & 63u;
#ifndef MIPT_ANY_SECOND_POP
#define MIPT_ANY_SECOND_POP 1       // 0: the A/B (any-hit kernel of the headline frame 0.1905 -> 0.1838 ms with it)
#endif
    bool dead = false;
    if (MODE != 0 && MIPT_ANY_SECOND_POP) {
        // an entry of a ray that has been found occluded in the meantime costs its lane the round (18 % of the popped entries on the
        // headline frame's shadow rays): such a lane takes one more entry from the top
        dead = have && sh_done[slot] != 0u;
        const unsigned long long dm = __ballot(dead);
        if (dm != 0ull && psize > 0) {
            const int nd = __popcll(dm), k2 = nd < psize ? nd : psize;
            const int r = (int)lane_rank(dm);
            if (dead && r < k2) { ent = stk[psize - 1 - r]; slot = ent & 63u; dead = sh_done[slot] != 0u; }
            psize -= k2;
        }
    }
    // the ray of that slot, from its owner's registers
    const float ox = shfl_f(T.o.x, slot), oy = shfl_f(T.o.y, slot), oz = shfl_f(T.o.z, slot);
    const float dx = shfl_f(T.d.x, slot), dy = shfl_f(T.d.y, slot), dz = shfl_f(T.d.z, slot);
    const float ix = shfl_f(T.idir.x, slot), iy = shfl_f(T.idir.y, slot), iz = shfl_f(T.idir.z, slot);
    const float nx = shfl_f(T.nood.x, slot), ny = shfl_f(T.nood.y, slot), nz = shfl_f(T.nood.z, slot);
    const float tmn = shfl_f(T.tmin, slot), tmx = shfl_f(T.tmax, slot);
    const uint32_t skp = AH != 0 ? shfl_u(T.skip, slot) : 0xffffffffu;       // continuation key of the any-hit-program kernels (Trav::skip)
    const uint32_t ahk = AH != 0 ? shfl_u(T.ahk, slot) : 1u;                 // ... and how many candidates the ray wants from this traversal
    float tcur = tmx;
    bool live = have;
    if (MODE == 0) { if (have) tcur = ord_float((uint32_t)(sh_best[(AH != 0 && ahk > 1u ? (kAhLevels - 1) * 64 : 0) + slot] >> 32)); }      // AH: pruned at the last wanted candidate
    else if (MIPT_ANY_SECOND_POP) { if (dead) live = false; }
    else           { if (have && sh_done[slot] != 0u) live = false; }
    const bool leaf = (ent >> 31) != 0u;
    const uint32_t payload = (ent >> 6) & 0x1ffffffu;
    const uint32_t lfirst = payload >> 2, lcnt = (payload & 3u) + 1u;
    const bool do_inner = live && !leaf;
    const bool do_tri   = live && leaf;
    u32x4 q0, q1, q2, q3;
    load_step<Q>(blob, bvh, do_inner, do_tri, payload, lfirst, q0, q1, q2, q3 PROBE_ARG);
#if MIPT_PROBE_VALU > 0
    if (probe_valu(tmn) == 12345.678f) tcur = 0.0f;
#endif
    const float4 n0 = make_float4(u2f(q0.x), u2f(q0.y), u2f(q0.z), u2f(q0.w)), n1 = make_float4(u2f(q1.x), u2f(q1.y), u2f(q1.z), u2f(q1.w)),
                 n2 = make_float4(u2f(q2.x), u2f(q2.y), u2f(q2.z), u2f(q2.w));
    uint32_t push0 = 0u, push1 = 0u, push2 = 0u, push3 = 0u;
    int npush = 0;
    float tmn_b = tmn, tcur_b = tcur;                         // the box tests' interval (AH: a little wider, ah_box_interval)
    if (AH != 0) ah_box_interval(nx, ny, nz, tmn, tcur, tmn_b, tcur_b);
    if (Q == 2 && do_inner) {
        const Slab4 b = slab_children4(q0, q1, q2, q3, ix, iy, iz, nx, ny, nz, tmn_b, tcur_b);
        npush = b.n;                                          // push1 = the nearest (ends on top), push0 the next, then push2, push3
        if (b.n > 0) push1 = pool_pack(b.ref[0], slot);
        if (b.n > 1) push0 = pool_pack(b.ref[1], slot);
        if (b.n > 2) push2 = pool_pack(b.ref[2], slot);
        if (b.n > 3) push3 = pool_pack(b.ref[3], slot);
    }
    if (Q != 2 && do_inner) {
        const SlabOut b = slab_children<Q>(q0, q1, q2, q3, ix, iy, iz, nx, ny, nz, tmn_b, tcur_b);
        const bool h0 = b.tn0 <= b.tf0, h1 = b.tn1 <= b.tf1;
        const int c0 = b.c0, c1 = b.c1;
        if (h0 && h1) {
            const bool swp = b.tn1 < b.tn0;
            push0 = pool_pack(swp ? c0 : c1, slot);           // farther child below ...
            push1 = pool_pack(swp ? c1 : c0, slot);           // ... the nearer one on top
            npush = 2;
        } else if (h0 || h1) {
            push1 = pool_pack(h0 ? c0 : c1, slot);
            npush = 1;
        }
    }
    if (do_tri) {
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
        const uint32_t prim = f2u(n0.w);
        const bool ok = det != 0.0f && u >= 0.0f && v >= 0.0f && u + v <= 1.0f && (t > tmn || (AH != 0 && t == tmn && prim > skp)) && t < tmx &&
                        (!cull_back || det > 0.0f || (f2u(n1.w) & 4u) != 0u);
        bool stop = false;
        if (MODE != 0) {
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
                for (int l = 0; l < kAhLevels; ++l) { const unsigned long long prev = atomicMin(&sh_best[l * 64 + slot], key); key = prev > key ? prev : key; }
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
        if (Q == 2) {                                         // the farthest children of all lanes first
            const unsigned long long m4 = __ballot(npush >= 4);
            if (m4 != 0ull) {
                if (npush >= 4) stk[psize + (int)lane_rank(m4)] = push3;
                psize += __popcll(m4);
            }
            const unsigned long long m3 = __ballot(npush >= 3);
            if (m3 != 0ull) {
                if (npush >= 3) stk[psize + (int)lane_rank(m3)] = push2;
                psize += __popcll(m3);
            }
        }
        const unsigned long long m2 = __ballot(npush >= 2);
        if (m2 != 0ull) {
            if (npush >= 2) stk[psize + (int)lane_rank(m2)] = push0;
            psize += __popcll(m2);
        }
        const unsigned long long m1 = __ballot(npush >= 1);
        if (m1 != 0ull) {
            if (npush >= 1) stk[psize + (int)lane_rank(m1)] = push1;
            psize += __popcll(m1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}


template <int MODE, int Q, int CAP = 24 * 64 - 384, int AH = 0>
__device__ __forceinline__ uint2 tail_trace(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, uint32_t* stk,
                                           unsigned long long* sh_best, float* sh_u, float* sh_v, uint32_t* sh_done,
                                           const int lane, const Trav& T, const bool has_ray, const bool cull_back = false)
{
    constexpr int kPoolCap = CAP;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    sh_best[lane] = ((unsigned long long)ord_bits(T.tbest + 0.0f) << 32) | 0xffffffffull;
    if (AH != 0) { for (int l = 1; l < kAhLevels; ++l) sh_best[l * 64 + lane] = sh_best[lane]; }       // (the levels lie over sh_u / sh_v / sh_done)
    else { sh_u[lane] = 0.0f; sh_v[lane] = 0.0f; sh_done[lane] = 0u; }
    const unsigned long long m = __ballot(has_ray);
    if (has_ray) stk[(int)lane_rank(m)] = pool_pack(0, (uint32_t)lane);
    int psize = __popcll(m);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#if MIPT_PROBE_STAMP
    unsigned long long pw_wait = 0ull;
#endif
    uint2 work = make_uint2(0u, 0u);          // rounds, entries offered to them (probe builds read it)
    while (psize > 0) {
        work.x += 1u; work.y += (uint32_t)(psize < 64 ? psize : 64);
#if MIPT_PROBE_WAVES
        if (MODE != 0) {                      // upper 16 bits of .x: popped entries whose ray has been found occluded already
            const int k = psize < 64 ? psize : 64;
            const bool dead = lane < k && sh_done[stk[psize - 1 - lane] & 63u] != 0u;
            work.x += (uint32_t)__popcll(__ballot(dead)) << 16;
        }
#endif
        pool_round<MODE, kPoolCap, Q, AH>(blob, bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, T, psize, cull_back PROBE_ARG);
    }
    return work;
}

}  // namespace mipt
