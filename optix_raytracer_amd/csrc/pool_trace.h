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
// MIPT_PROBE_UNI: per launch of k_trace_pool, into DeviceCounters::dbg: bulk steps, steps whose inner lanes stand at ONE node, of those
// the ones without a lane at a leaf, the sum of distinct nodes per step (counted up to 8), steps with <= 2 / <= 4 distinct nodes
#ifndef MIPT_PROBE_UNI
#define MIPT_PROBE_UNI 0
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

// ---- MIPT_SCALAR_TOP: wave-uniform node fetches through the scalar cache (round 4) -------------------------------------------------
// The rays of a tile start at the root together and keep to common nodes for the first levels (a 16 x 4-pixel tile of primary rays is
// narrower than the boxes of the top ~ten levels).  A step whose inner lanes ALL stand at one node fetches it with one s_load_dwordx8
// (scalar data cache, its own path to the L2) instead of two lane-wide buffer loads of 64 equal addresses -- which cost the CU's
// vector-memory pipeline ~21 cycles each (profiles/r03_micro_vmem.txt), the resource this kernel is bound by.  Lanes at a leaf in such a
// step fetch their triangle with masked loads as before.  The visiting order is untouched: results and counters are the same bit for bit.
#ifndef MIPT_SCALAR_TOP
#define MIPT_SCALAR_TOP 0          // measured SLOWER (round 4: closest-hit kernel 0.347 against 0.337 ms): kept as a probe build
#endif
template <int Q, bool TOP = false, bool SC = false>
__device__ __forceinline__ void load_step(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, const bool inner_in, const bool tri,
                                          const uint32_t node, const uint32_t lfirst, u32x4& n0, u32x4& n1, u32x4& n2, u32x4& n3 PROBE_PARAM,
                                          const uint32_t* top_lds = nullptr)
{
#if MIPT_PROBE_STAMP
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    if (SC && Q == 1 && !TOP) {
        const unsigned long long im = __ballot(inner_in);
        if (im != 0ull) {
            const uint32_t un = (uint32_t)__builtin_amdgcn_readlane((int)node, (int)__builtin_ctzll(im));
            if (__ballot(inner_in && node != un) == 0ull) {
                typedef const __attribute__((address_space(4))) u32x4* ConstQ;        // constant address space: selected as s_load
                const ConstQ p = (ConstQ)((unsigned long long)bvh.nodes + (unsigned long long)bvh.node32_off + ((unsigned long long)un << 5));
                const u32x4 s0 = p[0], s1 = p[1];
                n0 = s0; n1 = s1; n2 = u32x4{ 0u, 0u, 0u, 0u }; n3 = n2;
                if (__ballot(tri) != 0ull) {
                    if (tri) {
                        const uint32_t off = bvh.tri_byte_off + lfirst * 48u;
                        n0 = ldq(blob, off); n1 = ldq(blob, off + 16u); n2 = ldq(blob, off + 32u);
                    }
                }
                asm volatile("" ::: "memory");
                return;
            }
        }
    }
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


// ---- quad rounds (round 4): a THIN pool -- at most 16 entries -- is served by quads of lanes, one entry per quad ---------------------
// The kernels that live on the pool are bound by the CHAIN of their rounds wherever waves are thin: the last paths of a Russian-roulette
// pass (one or two live rays per wave for tens of bounces), a rank's 1/8 shard (rounds a third full), the end of every chunk.  A thin
// round pays for 64 lanes and uses a few: each busy lane fetches a whole BVH4 node (4 loads), tests four boxes, sorts them and pushes up
// to four entries in four ballot stages -- ~360 instructions that ONE wave issues at ~4 cycles each, whatever the other lanes do -- and a
// leaf of k triangles takes k rounds (a lane tests one and pushes the rest).  With <= 16 entries the wave gives every entry a QUAD: lane
// c of the quad fetches child c's 16-byte piece (one load instruction for the round) and tests that one box; the quad ranks its hit
// children by entry distance with three DPP quad broadcasts; lane c of a leaf entry tests triangle first + c, so a leaf is ONE round.
// Same slab and triangle arithmetic, same merge (ds_min_u64 on the order-preserving key); hit records do not depend on the visiting
// order, so frames and records are unchanged bit for bit (the existing parity tests run every pool kernel both ways: MIPT_QUAD_MAX).
// SB: slot bits of an entry (6: pool_round; 7: pool_round2, whose bit 6 marks the lane's occlusion ray, kept in LDS: sray).
// MODE 0 closest / 1, 2 any / 3 per entry by that bit.
#ifndef MIPT_QUAD_MAX
#define MIPT_QUAD_MAX 16          // entries (0: never).  A pool of more runs the 64-entry round.
#endif
template <int J>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, true));
}

template <int MODE, int SB>
__device__ __forceinline__ void pool_round_quad(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, uint32_t* stk,
                                                unsigned long long* sh_best, float* sh_u, float* sh_v, uint32_t* sh_done, const float* sray,
                                                const int lane, const Trav& T, int& psize, const bool cull_back)
{
    constexpr uint32_t kSlotMask = (1u << SB) - 1u, kPayMask = (1u << (31 - SB)) - 1u;
    const int k = psize < 16 ? psize : 16;
    const int e = lane >> 2;
    const uint32_t c = (uint32_t)lane & 3u;
    const bool have = e < k;
    uint32_t ent = 0u;
    if (have) ent = stk[psize - 1 - e];
    psize -= k;
    const uint32_t slotw = ent & kSlotMask, ls = slotw & 63u;
    const bool is_s = SB == 7 && have && (slotw & 64u) != 0u;          // pool2: the slot's occlusion ray
    float ox = shfl_f(T.o.x, ls), oy = shfl_f(T.o.y, ls), oz = shfl_f(T.o.z, ls);
    float dx = shfl_f(T.d.x, ls), dy = shfl_f(T.d.y, ls), dz = shfl_f(T.d.z, ls);
    float ix = shfl_f(T.idir.x, ls), iy = shfl_f(T.idir.y, ls), iz = shfl_f(T.idir.z, ls);
    float nx = shfl_f(T.nood.x, ls), ny = shfl_f(T.nood.y, ls), nz = shfl_f(T.nood.z, ls);
    float tmn = shfl_f(T.tmin, ls), tmx = shfl_f(T.tmax, ls);
    if (SB == 7) {
        if (__ballot(is_s) != 0ull) {
            if (is_s) {
                ox = sray[0 * 64 + ls]; oy = sray[1 * 64 + ls]; oz = sray[2 * 64 + ls];
                dx = sray[3 * 64 + ls]; dy = sray[4 * 64 + ls]; dz = sray[5 * 64 + ls];
                ix = sray[6 * 64 + ls]; iy = sray[7 * 64 + ls]; iz = sray[8 * 64 + ls];
                nx = sray[9 * 64 + ls]; ny = sray[10 * 64 + ls]; nz = sray[11 * 64 + ls];
                tmn = sray[12 * 64 + ls]; tmx = sray[13 * 64 + ls];
            }
        }
    }
    const bool any_sem = MODE == 3 ? is_s : MODE != 0;                 // this entry's ray ends at its first hit
    float tcur = tmx;
    bool live = have;
    if (have) {
        if (any_sem) { if (sh_done[ls] != 0u) live = false; }
        else tcur = ord_float((uint32_t)(sh_best[ls] >> 32));
    }
    const bool leaf = (ent >> 31) != 0u;
    const uint32_t payload = (ent >> SB) & kPayMask;
    const uint32_t lfirst = payload >> 2, lcnt = (payload & 3u) + 1u;
    const bool do_inner = live && !leaf;
    const bool do_tri = live && leaf && c < lcnt;
    // ------------------------------------------------------------ this round's loads: a child's piece, or a triangle
    uint32_t off = bvh.node4_off;                                       // idle lanes re-read the root's first piece (a broadcast)
    if (do_inner) off = bvh.node4_off + (payload << 6) + (c << 4);
    if (do_tri)   off = bvh.tri_byte_off + (lfirst + c) * 48u;
    const u32x4 q0 = ldq(blob, off);
    u32x4 q1 = u32x4{ 0u, 0u, 0u, 0u }, q2 = q1;
    if (__ballot(do_tri) != 0ull) { if (do_tri) { q1 = ldq(blob, off + 16u); q2 = ldq(blob, off + 32u); } }
    asm volatile("" ::: "memory");
    // ------------------------------------------------------------ one box per lane, ranked inside the quad
    float key = __builtin_inff();
    int ref = kNoChild;
    if (do_inner) slab_piece(q0, ix, iy, iz, nx, ny, nz, tmn, tcur, key, ref);
    const bool hit = key < __builtin_inff();
    uint32_t rank = 0u;
    if (MODE == 0 || MODE == 3) {
        // entries ahead of this one in the quad: a smaller entry distance, or an equal one in a lower lane (all lanes take part in the DPP reads)
        const float k0 = quad_bcast<0>(key), k1 = quad_bcast<1>(key), k2 = quad_bcast<2>(key), k3 = quad_bcast<3>(key);
        rank = (k0 < key || (k0 == key && 0u < c) ? 1u : 0u) + (k1 < key || (k1 == key && 1u < c) ? 1u : 0u) +
               (k2 < key || (k2 == key && 2u < c) ? 1u : 0u) + (k3 < key || (k3 == key && 3u < c) ? 1u : 0u);
    }
    // ------------------------------------------------------------ one triangle per lane of a leaf's quad
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
        const uint32_t prim = f2u(n0.w);
        const bool ok = det != 0.0f && u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > tmn && t < tmx &&
                        (!cull_back || det > 0.0f || (f2u(n1.w) & 4u) != 0u);
        if (ok) {
            if (any_sem) sh_done[ls] = 1u;
            else {
                const unsigned long long kk = ((unsigned long long)ord_bits(t + 0.0f) << 32) | (unsigned long long)prim;
                atomicMin(&sh_best[ls], kk);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (*(volatile unsigned long long*)&sh_best[ls] == kk) { sh_u[ls] = u; sh_v[ls] = v; }
            }
        }
    }
    // ------------------------------------------------------------ pushes: every quad's farthest first, the nearest on top
    uint32_t word = 0u;
    if (hit) {
        if (ref >= 0) word = ((uint32_t)ref << SB) | slotw;
        else { const uint32_t desc = ~(uint32_t)ref; word = 0x80000000u | ((desc >> 4) << (SB + 2)) | (((desc - 1u) & 3u) << SB) | slotw; }
    }
    if (MODE == 0 || MODE == 3) {
        const unsigned long long m3 = __ballot(hit && rank == 3u), m2 = __ballot(hit && rank == 2u), m1 = __ballot(hit && rank == 1u), m0 = __ballot(hit && rank == 0u);
        const int c3 = __popcll(m3), c2 = __popcll(m2), c1 = __popcll(m1);
        if (hit) {
            const unsigned long long mm = rank == 0u ? m0 : (rank == 1u ? m1 : (rank == 2u ? m2 : m3));
            const int base = rank == 0u ? c3 + c2 + c1 : (rank == 1u ? c3 + c2 : (rank == 2u ? c3 : 0));
            stk[psize + base + (int)lane_rank(mm)] = word;
        }
        psize += c3 + c2 + c1 + __popcll(m0);
    } else {
        const unsigned long long m = __ballot(hit);                      // (an occlusion ray does not care which of its boxes comes first)
        if (hit) stk[psize + (int)lane_rank(m)] = word;
        psize += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
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
    if (Q == 2 && AH == 0 && MIPT_QUAD_MAX > 0) {
        if (psize <= MIPT_QUAD_MAX) { pool_round_quad<MODE, 6>(blob, bvh, stk, sh_best, sh_u, sh_v, sh_done, nullptr, lane, T, psize, cull_back); return; }
    }
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
