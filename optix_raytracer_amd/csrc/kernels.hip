// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// Wavefront replacement of the OptiX megakernel that optixLaunch runs in
// SDK/optixPathTracer (file:line below relative to /root/reference):
//
//   k_raygen          <- __raygen__rg prologue            optixPathTracer.cu:249-281
//   k_trace_pool<0>   <- optixTraverse closest hit        optixPathTracer.cu:184-197   (driver-internal BVH walk)
//   k_shade           <- __closesthit__radiance / __miss__radiance + the raygen bounce loop body
//                                                           optixPathTracer.cu:291-303, 323-413
//   k_trace_static<2> <- traceOcclusion (any hit)         optixPathTracer.cu:218-240 + radiance add :292
//   k_resolve         <- raygen epilogue                  optixPathTracer.cu:308-319
//   k_tail_pool       <- the raygen bounce loop as one kernel, for the last paths of a pass  optixPathTracer.cu:283-305
//   k_unpermute       <- multi-GPU merge                  optixMultiGPU.cu:282-292 (scatter by StaticWorkDistribution)
//
// Traversal is cache-latency / instruction-issue bound pointer chasing (64-B nodes, 48-B triangles); MFMA is not used.
// Default closest-hit kernel: 64-ray tiles, one ray per lane with the per-lane traversal stack in LDS ([entry][lane] ->
// conflict free), then a per-wave pool of stack entries that all 64 lanes work on; any-hit launches and small launches run
// on that pool from the root.  Other variants: k_trace_stream (persistent waves, lane refill from per-XCD cursors, subtree
// stealing; also the counting launches), k_trace_simple (one thread per ray).  Wave-ballot compaction of the output queues.
// Floating point: -ffp-contract=off, explicit fmaf only where the specification says so.
#include "kernels.h"
#include "detmath.h"
#include "wave_util.h"
#include "trav_core.h"
#include "pool_core.h"
#include "inst_pool.h"
#include "pool_trace.h"

namespace mipt {

// ------------------------------------------------------------------------------------------------
// traversal core (build-defined conventions, identical sequence in oracle/pt_oracle.c):
//   hit iff tmin < t < tmax; double sided; det == 0 rejected; u >= 0, v >= 0, u + v <= 1;
//   closest-hit tie -> lowest original primitive index; near child first, tie -> child 0 first.
// ------------------------------------------------------------------------------------------------
// leaf step: Moeller-Trumbore on (v0, e1, e2); returns true when ANY mode may terminate
template <bool ANY>
__device__ __forceinline__ bool visit_leaf(Trav& T, const float4* __restrict__ tris, uint32_t& n_tests, bool cull_back)
{
    const uint32_t desc = ~(uint32_t)T.node;
    const uint32_t first = desc >> 4, cnt = desc & 15u;
    for (uint32_t i = 0; i < cnt; ++i) {
        const float4* tp = tris + (size_t)(first + i) * 3u;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        ++n_tests;
        const f3 e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
        const f3 p  = fcross(T.d, e2);
        const float det = fdot(e1, p);
        if (det == 0.0f) continue;
        if (cull_back && det < 0.0f && (f2u(b.w) & 4u) == 0u) continue;       // back face of a single-sided triangle
        const float inv = 1.0f / det;
        const f3 tv = T.o - mk3(a.x, a.y, a.z);
        const float u = fdot(tv, p) * inv;
        const f3 q = fcross(tv, e1);
        const float v = fdot(T.d, q) * inv;
        const float t = fdot(e2, q) * inv;
        if (!(u >= 0.0f && v >= 0.0f && u + v <= 1.0f)) continue;
        if (!(t > T.tmin && t < T.tmax)) continue;
        const uint32_t prim = f2u(a.w);
        if (ANY) { T.tbest = t; T.prim = prim; return true; }
        if (t < T.tbest || (t == T.tbest && prim < T.prim)) { T.tbest = t; T.prim = prim; T.u = u; T.v = v; }
    }
    return false;
}

// ---- streaming waves: the default traversal kernel -------------------------------------------
// Every lane is a small state machine (at an inner node | at a leaf with k triangles left | idle) and
// the wave runs ONE loop.  Per iteration every lane loads 64 bytes -- its node, or its next triangle
// (nodes and triangles share one allocation, so inner lanes and leaf lanes execute the same four
// buffer loads: uniform descriptor + 32-bit lane offset) -- then, after ONE wait, inner lanes do a slab
// step and leaf lanes one branch-free triangle test.  Lanes never sit out a whole inner or leaf phase
// as in the classic while-while loop (measured there: 19-27 % VALU lane utilisation, profiles/r01_*;
// tools/sim_divergence.py models the schedules).
//   * Refill: a finished lane is swapped onto a ray that was prefetched into registers while it was
//     still traversing, in batches, so refilling exposes no memory latency.
//   * Drain (STEAL): once the ray range is exhausted the frame time is the dependent-load chain of the
//     few longest rays (hundreds of steps).  Idle lanes then take the BOTTOM entry of a busy lane's
//     traversal stack -- an independent subtree of the same ray -- and traverse it with their own
//     running best; a per-ray slot in LDS merges the partial results (min t, tie -> lowest primitive)
//     and the last participant writes the hit record.  Sequential traversal visits a subset of what the
//     participants visit, so the result is the same; visit counters are only exact without stealing,
//     hence COUNT builds do not steal.
// The per-ray visiting order of a non-shared ray is unchanged (a lane never takes an inner step while
// it owes a triangle test), so hit records and visit counters equal the oracle's.
// One step of the lane state machine (see k_trace_stream): every lane loads 64 bytes -- its node, or its next
// triangle -- then inner lanes do a slab step and leaf lanes one branch-free triangle test.  `sb` is the bottom of
// the lane's stack (entries below it were given away); shared by k_trace_stream and the bulk phase of k_trace_pool.
template <int MODE, bool COUNT, int Q = 0, bool TOP = false, bool SC = false>
__device__ __forceinline__ void stream_step(const TraceArgs& a, const __amdgpu_buffer_rsrc_t blob, uint32_t* stk, int lane, bool has_ray, Trav& T,
                                            int sb, bool cull_back, uint32_t& n_visits, uint32_t& n_tests PROBE_PARAM, const uint32_t* top_lds = nullptr)
{
    // ------------------------------------------------------------ this iteration's load
    const bool inner   = has_ray && T.node >= 0;
    const bool at_leaf = has_ray && T.node < 0 && T.node != kSentinel;
    const uint32_t ldesc = ~(uint32_t)T.node;
    const uint32_t lfirst = ldesc >> 4, lcnt = ldesc & 15u;
    u32x4 q0, q1, q2, q3;
    load_step<Q, TOP, SC>(blob, a.bvh, inner, at_leaf && lcnt != 0u, (uint32_t)T.node, lfirst, q0, q1, q2, q3 PROBE_ARG, top_lds);
#if MIPT_PROBE_VALU > 0
    if (probe_valu(T.tmin) == 12345.678f) T.tbest = 0.0f;                 // never true (tmin is 0.01 or a hit distance)
#endif
    // ------------------------------------------------------------ one inner-node step
    if (Q == 2 && inner) {
        const Slab4 b = slab_children4(q0, q1, q2, q3, T.idir.x, T.idir.y, T.idir.z, T.nood.x, T.nood.y, T.nood.z, T.tmin, T.tbest);
        if (b.n == 0) {
            if (T.sp == sb) T.node = kSentinel;
            else { --T.sp; T.node = (int)stk[T.sp * 64 + lane]; }
        } else {
            if (b.n > 3) { stk[T.sp * 64 + lane] = (uint32_t)b.ref[3]; ++T.sp; }          // farthest first, the nearest is walked next
            if (b.n > 2) { stk[T.sp * 64 + lane] = (uint32_t)b.ref[2]; ++T.sp; }
            if (b.n > 1) { stk[T.sp * 64 + lane] = (uint32_t)b.ref[1]; ++T.sp; }
            T.node = b.ref[0];
        }
    }
    if (Q != 2 && inner) {
        if (COUNT) ++n_visits;
        const SlabOut b = slab_children<Q>(q0, q1, q2, q3, T.idir.x, T.idir.y, T.idir.z, T.nood.x, T.nood.y, T.nood.z, T.tmin, T.tbest);
        const bool h0 = b.tn0 <= b.tf0, h1 = b.tn1 <= b.tf1;
        const int c0 = b.c0, c1 = b.c1;
        if (h0 && h1) {
            const bool swp = b.tn1 < b.tn0;
            stk[T.sp * 64 + lane] = (uint32_t)(swp ? c0 : c1);
            ++T.sp;
            T.node = swp ? c1 : c0;
        } else if (h0 || h1) {
            T.node = h0 ? c0 : c1;
        } else if (T.sp == sb) {
            T.node = kSentinel;
        } else {
            --T.sp;
            T.node = (int)stk[T.sp * 64 + lane];
        }
    }
    // ------------------------------------------------------------ one triangle test (branch free)
    if (at_leaf) {
        bool stop = false;
        if (lcnt != 0u) {
            if (COUNT) ++n_tests;
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
            const uint32_t prim = f2u(ta.w);
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
        else if (T.sp == sb) T.node = kSentinel;
        else { --T.sp; T.node = (int)stk[T.sp * 64 + lane]; }
    }
}

template <int MODE, bool COUNT>
__global__ __launch_bounds__(64) void k_trace_stream(TraceArgs a)
{
    constexpr bool STEAL = !COUNT;
    __shared__ uint32_t stk[kStreamStack * 64];          // [entry][lane]: bank = lane
    __shared__ float    sh_t[64], sh_u[64], sh_v[64];    // per-ray merge slots, indexed by the owner's lane
    __shared__ uint32_t sh_prim[64], sh_ray[64];
    __shared__ int      sh_pending[64];
    __shared__ uint32_t sh_src[64], sh_node[64];         // steal round: donor lane / donated reference per pair
    const int lane = (int)threadIdx.x;
    const bool cull_back = MODE == 0 && (a.ray_flags & 16u) != 0u;   // MIPT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES: det < 0 is a back face
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    const uint32_t seg_len = (((n + kNumSegments - 1) / kNumSegments) + 63u) & ~63u;
    uint32_t seg = blockIdx.x & (kNumSegments - 1);
    uint32_t tried = 0;
    uint32_t pool_next = 0, pool_end = 0;
    const int refill_min = a.refill_min, steal_min = a.steal_min;
    // nodes, then triangles at tri_byte_off, in one allocation
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.blob_bytes, 0x00020000);

    bool     has_ray = false, nxt_valid = false, shared = false;
    uint32_t ray = 0, nxt_idx = 0, own = (uint32_t)lane;
    int      sb = 0;                                      // stack bottom (entries below were given away)
    float4   nxt_o = make_float4(0.f, 0.f, 0.f, 0.f), nxt_d = nxt_o;
    uint32_t n_visits = 0, n_tests = 0;
    Trav T;
    T.node = kSentinel; T.sp = 0;
    sh_pending[lane] = 0;

    for (;;) {
        // ------------------------------------------------------------ swap in prefetched rays, prefetch more
        const unsigned long long idle = __ballot(!has_ray);
        const int n_idle = __popcll(idle);
        const bool dry = tried >= (uint32_t)kNumSegments && pool_next == pool_end;
        if (n_idle == 64 || (n_idle >= refill_min && (!dry || __ballot(nxt_valid && !has_ray) != 0ull))) {
            if (!has_ray && nxt_valid) {
                ray = nxt_idx;
                trav_init(T, nxt_o, nxt_d, a.bvh.xf);
                sb = 0; shared = false; own = (uint32_t)lane;
                has_ray = T.node != kSentinel;
                if (!has_ray) finish_ray<MODE>(a, T, ray, false);      // empty interval: a miss, no BVH access
                nxt_valid = false;
            }
            const unsigned long long want = __ballot(!nxt_valid);
            const uint32_t n_want = (uint32_t)__popcll(want);
            const uint32_t rank = lane_rank(want);
            uint32_t given = 0;
            while (given < n_want) {
                if (pool_next == pool_end) {
                    if (tried >= (uint32_t)kNumSegments) break;
                    uint32_t c = 0;
                    if (lane == 0) c = atomicAdd(&a.cursors[seg * kCursorStride], kStreamChunk);
                    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                    const uint32_t lo = seg * seg_len;
                    uint32_t hi = lo + seg_len; hi = hi < n ? hi : n;
                    if (lo >= hi || c >= hi - lo) { seg = (seg + 1u) & (kNumSegments - 1); ++tried; continue; }
                    pool_next = lo + c;
                    pool_end  = pool_next + kStreamChunk < hi ? pool_next + kStreamChunk : hi;
                }
                const uint32_t avail = pool_end - pool_next;
                const uint32_t need  = n_want - given;
                const uint32_t take  = avail < need ? avail : need;
                if (!nxt_valid && rank >= given && rank < given + take) {
                    nxt_idx = pool_next + (rank - given);
                    nxt_o = a.ray_o[(size_t)nxt_idx * a.stride];      // lands while this lane keeps traversing
                    nxt_d = a.ray_d[(size_t)nxt_idx * a.stride];
                    nxt_valid = true;
                }
                pool_next += take;
                given += take;
            }
            if (__ballot(has_ray || nxt_valid) == 0ull) break;
        }
        // ------------------------------------------------------------ drain: idle lanes take subtrees of busy rays
        if (STEAL && dry && n_idle >= steal_min) {
            const bool thief  = !has_ray && !nxt_valid;
            // a lane's merge slot may still belong to an earlier ray of this lane whose helpers are not done:
            // such a lane does not donate from a new, not yet shared ray
            const bool victim = has_ray && T.sp > sb && (shared || sh_pending[lane] == 0);
            const unsigned long long tm = __ballot(thief), vm = __ballot(victim);
            const uint32_t nt = (uint32_t)__popcll(tm), nv = (uint32_t)__popcll(vm);
            const uint32_t np = nt < nv ? nt : nv;
            if (np != 0u) {
                const uint32_t rt = lane_rank(tm), rv = lane_rank(vm);
                const bool donate = victim && rv < np, receive = thief && rt < np;
                if (donate) {
                    if (!shared) {          // first donation of this ray: open its merge slot
                        sh_t[lane] = 0.0f; sh_u[lane] = 0.0f; sh_v[lane] = 0.0f;
                        sh_prim[lane] = 0xffffffffu; sh_ray[lane] = ray; sh_pending[lane] = 1;
                        shared = true;      // own == lane already
                    }
                    sh_src[rv]  = (uint32_t)lane;
                    sh_node[rv] = stk[sb * 64 + lane];
                    ++sb;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (donate) atomicAdd(&sh_pending[own], 1);
                uint32_t src = (uint32_t)lane, got = 0u;
                if (receive) { src = sh_src[rt]; got = sh_node[rt]; }
                // the thief copies the donor's ray; all lanes take part in the shuffles
                const float ox = shfl_f(T.o.x, src), oy = shfl_f(T.o.y, src), oz = shfl_f(T.o.z, src);
                const float dx = shfl_f(T.d.x, src), dy = shfl_f(T.d.y, src), dz = shfl_f(T.d.z, src);
                const float ix = shfl_f(T.idir.x, src), iy = shfl_f(T.idir.y, src), iz = shfl_f(T.idir.z, src);
                const float nx = shfl_f(T.nood.x, src), ny = shfl_f(T.nood.y, src), nz = shfl_f(T.nood.z, src);
                const float tmn = shfl_f(T.tmin, src), tmx = shfl_f(T.tmax, src), tbs = shfl_f(T.tbest, src);
                const uint32_t ow = shfl_u(own, src);
                if (receive) {
                    T.o = mk3(ox, oy, oz); T.d = mk3(dx, dy, dz); T.idir = mk3(ix, iy, iz); T.nood = mk3(nx, ny, nz);
                    T.tmin = tmn; T.tmax = tmx; T.tbest = tbs;
                    T.u = 0.0f; T.v = 0.0f; T.prim = 0xffffffffu;
                    T.node = (int)got; T.sp = 0; sb = 0;
                    own = ow; shared = true; has_ray = true;
                }
            }
        }
        // any-hit: a participant stops as soon as another one has found an occluder
        if (STEAL && MODE != 0 && has_ray && shared && sh_prim[own] != 0xffffffffu) T.node = kSentinel;
        // ------------------------------------------------------------ a burst of traversal steps
        // The bookkeeping above (ballots, refill, steal round) is amortised over `burst` steps; a lane that
        // finishes inside a burst idles until its end.
        const int burst = dry ? a.burst_dry : a.burst_bulk;
        for (int bi = 0; bi < burst; ++bi) {
            if (bi != 0 && __ballot(has_ray && T.node != kSentinel) == 0ull) break;
#if MIPT_PROBE_STAMP
            unsigned long long pw_wait = 0ull;
#endif
            stream_step<MODE, COUNT>(a, blob, stk, lane, has_ray, T, sb, cull_back, n_visits, n_tests PROBE_ARG);
        }
        // ------------------------------------------------------------ retire
        const bool done = has_ray && T.node == kSentinel;
        if (done && !shared) finish_ray<MODE>(a, T, ray, T.prim != 0xffffffffu);
        if (STEAL) {
            // participants of a shared ray merge into its slot one after the other (same wave: program order)
            unsigned long long fm = __ballot(done && shared);
            while (fm != 0ull) {
                const int l = __builtin_ctzll(fm);
                if (lane == l) {
                    if (T.prim != 0xffffffffu) {
                        const uint32_t sp_ = sh_prim[own];
                        const float st_ = sh_t[own];
                        if (MODE != 0 || sp_ == 0xffffffffu || T.tbest < st_ || (T.tbest == st_ && T.prim < sp_)) {
                            sh_t[own] = T.tbest; sh_prim[own] = T.prim; sh_u[own] = T.u; sh_v[own] = T.v;
                        }
                    }
                    const int left = sh_pending[own] - 1;
                    sh_pending[own] = left;
                    if (left == 0) {
                        Trav R = T;
                        R.tbest = sh_t[own]; R.prim = sh_prim[own]; R.u = sh_u[own]; R.v = sh_v[own];
                        finish_ray<MODE>(a, R, sh_ray[own], R.prim != 0xffffffffu);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                fm &= fm - 1ull;
            }
        }
        if (done) has_ray = false;
    }
    if (COUNT) {
        wave_add64(&a.ctr->node_visits[MODE != 0], n_visits);
        wave_add64(&a.ctr->tri_tests[MODE != 0], n_tests);
    }
}

// ---- tiles + entry pool: the default traversal kernel -------------------------------------------------------
// Bulk phase: the launch's rays are cut into tiles of 64 consecutive rays (one 16x4 pixel tile for primary rays);
// a wave takes tile after tile (statically: wave w of XCD x takes tiles w, w + W, ... of the x-th eighth -- no
// cursor, no atomic; with the default grid of one wave per tile up to 128 per CU the hardware's workgroup
// dispatcher is the load balancer) and walks its 64 rays with the lane state machine of stream_step.
// Tail phase: when `steal_min` lanes of the tile are finished the wave stops being "one ray per lane".  Every
// pending traversal-stack entry of every lane is moved, in place, into ONE stack of (entry, ray slot) pairs --
// the wave's entry pool -- and each round pops up to 64 entries, whichever rays they belong to; each lane
// processes one (an inner node: slab-test both children and push the survivors, nearer last; a leaf reference:
// test its first triangle and push the rest of the leaf).  A ray with many pending subtrees is traversed by
// many lanes at once instead of being a dependent-load chain of hundreds of steps in one lane.
// A launch with few rays (shadow rays of a small frame, a shard of a multi-GPU frame) runs on the pool from the
// start: waves fetch rays 16 at a time from per-XCD cursors whenever their pool runs low, so 64 lanes serve
// 16-32 rays and the rays spread over the waves by work.
// (Measured and dropped: persistent waves that refill finished lanes from ray-range cursors with register
//  prefetch -- the design of k_trace_stream -- in the bulk phase: 0.60 ms against 0.55 ms for the closest-hit
//  kernel of the headline frame, and slower still on 1/2 and 1/4 frames.)
// The ray of slot s stays in the registers of lane s (read by the others with ds_bpermute); its running best
// hit lives in LDS as one 64-bit word (order-preserving t bits << 32 | primitive) merged with ds_min_u64:
// min t, tie -> lowest primitive, the conventions of the sequential traversal, which visits a subset of what
// the pool visits -- same result, but visit counters differ, so COUNT launches use k_trace_stream.
// Packed entry: bit 31 leaf | 25-bit payload | 6-bit slot; inner payload = node index, leaf payload =
// first << 2 | (count - 1): leaves of 1 .. 4 triangles, < 2^23 triangles.  The launcher only selects this kernel for BVHs
// that fit (DeviceBvh::pool_ok; the builders make leaves of at most 4, and only the forced root over 0 or 1 triangles has
// an empty one).

// The wave's private range of ray indices (all members wave-uniform).  Each per-XCD segment of the ray array is
// split into a STATIC part -- one first chunk per wave of that XCD, taken without any atomic -- and a dynamic
// part handed out by the segment's cursor.  Same-address atomics serialise in L2 at tens of ns each, so 4096
// waves that all start (or all run dry) at the same moment must not queue on 8 cache lines: the first chunk
// is implicit, and a wave whose segment has run out LOOKS at all 8 cursors with one 8-lane load instead of
// probing them with 8 consecutive atomics.  (A load in front of EVERY grab, or a shared "exhausted" flag
// word, were both measured slower: closest-hit kernel 0.63 -> 0.85 ms on the headline frame.)
struct RayRange {
    uint32_t seg, next, end, tried;
    uint32_t stat;            // rays per segment covered by the static first chunks
    bool     dry;             // every segment is exhausted
};

__device__ __forceinline__ void range_init(RayRange& r, uint32_t n, uint32_t seg_len, uint32_t first_chunk)
{
    r.seg = blockIdx.x & (kNumSegments - 1);
    r.stat = (gridDim.x / kNumSegments) * first_chunk;            // the launcher makes gridDim.x a multiple of 8
    const uint32_t lo = r.seg * seg_len;
    uint32_t hi = lo + seg_len; hi = hi < n ? hi : n;
    const uint32_t mine = lo + (blockIdx.x / kNumSegments) * first_chunk;
    r.next = mine < hi ? mine : 0u;
    r.end  = mine < hi ? (mine + first_chunk < hi ? mine + first_chunk : hi) : 0u;
    r.dry  = false; r.tried = 0u;
}

// makes [next, end) non-empty by grabbing up to `chunk` rays; false when every segment is exhausted
__device__ __forceinline__ bool range_grab(const TraceArgs& a, RayRange& r, uint32_t n, uint32_t seg_len, uint32_t chunk, int lane)
{
    while (r.next == r.end) {
        if (r.dry) return false;
        const uint32_t lo = r.seg * seg_len;
        uint32_t hi = lo + seg_len; hi = hi < n ? hi : n;
        const uint32_t len = hi > lo ? hi - lo : 0u;
        uint32_t c = len;
        if (len > r.stat) {
            if (lane == 0) c = atomicAdd(&a.cursors[r.seg * kCursorStride], chunk);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        }
        if (len > r.stat && c < len - r.stat) {
            r.next = lo + r.stat + c;
            r.end  = r.next + chunk < hi ? r.next + chunk : hi;
            break;
        }
        r.seg = (r.seg + 1u) & (kNumSegments - 1);
        if (++r.tried >= (uint32_t)kNumSegments) { r.dry = true; return false; }
    }
    return true;
}

// ROWS: entries of a lane's stack = rows of the LDS block (24 when the tree is at most 24 deep -- 7.5 KB of LDS per
// wave and <= 96 VGPRs let 20 waves share a CU instead of 16 -- else 32)
// a tile's sort key for the next launch's order: 0 = heaviest, 255 = lightest; classes of `width` cost units (bulk steps + pool rounds)
__device__ __forceinline__ uint32_t tile_class(uint32_t cost, uint32_t width)
{
    const uint32_t q = cost / width;
    return (kOrderClasses - 1u) - (q < kOrderClasses - 1u ? q : kOrderClasses - 1u);
}
__device__ __forceinline__ void record_tile_cost(const TraceArgs& a, uint32_t tile, uint32_t cost)
{
    const uint32_t c = tile_class(cost, a.class_width_dev != nullptr ? *a.class_width_dev : a.class_width);
    a.tile_cost[tile] = (uint8_t)c;
    atomicAdd(&a.tile_hist[(tile / kOrderChunk) * kOrderClasses + c], 1u);
}

#ifndef MIPT_TILED_ROWS        // experiments: LDS rows / waves per SIMD of the tiled kernel's small build (trees up to that many levels)
#define MIPT_TILED_ROWS 24
#endif
#ifndef MIPT_TILED_WAVES
#define MIPT_TILED_WAVES 6
#endif
__device__ __forceinline__ void begin_sample(DeviceCounters* c, uint32_t i, uint32_t n_paths, uint32_t n_valid, uint32_t zero_stats);
template <int WR>
__device__ __forceinline__ bool raygen_path(const FrameConsts& fc, uint32_t sample, const PathQueue& q, uint32_t* rg_seed, float4* result,
                                            uint32_t i, float4& ro, float4& rd, uint32_t& slot_out, uint32_t& seed_out);
// a RaygenJob that is the kernel's `Member` argument, read through the kernel-argument segment where it is used: as a plain
// argument its ~50 words would stay in SGPRs across the traversal loops (k_trace_pool spilled 17 SGPRs and 18 VGPRs that way)
#if defined(__HIP_DEVICE_COMPILE__)
#define MIPT_RG_JOB_PTR(Type, member, fallback) \
    ((const __attribute__((address_space(4))) RaygenJob*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Type, member)))
#else
#define MIPT_RG_JOB_PTR(Type, member, fallback) (&(fallback))          /* (the host pass only parses the kernels) */
#endif

// RG: the launch makes the rays it traces (the pass's primary rays: raygen_path, k_raygen's body) and does the pass's queue
// bookkeeping -- a raygen launch of its own is 25 us of stores plus a launch gap in front of this kernel, which has the store
// bandwidth to spare, and the rays need not be read back.
template <int MODE, int ROWS, int Q, bool RG = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ROWS <= 24 ? MIPT_TILED_WAVES : 5, ROWS <= 24 ? MIPT_TILED_WAVES : 5)))
void k_trace_pool(TraceArgs a, RaygenJob rg)
{
    // ONE LDS block per wave: in the bulk phase ROWS rows of 64 lanes (the lanes' stacks), in the tail phase the entry
    // pool in its first kPoolCap words and the per-slot merge state in the last 384 (written only after the stacks
    // have been moved into the pool).  ROWS = 24: 6 KB, 26 waves per CU by LDS; 24 by registers (<= 80 VGPRs).
    constexpr int kPoolCap = ROWS * 64 - 384;
    constexpr bool TOP = MIPT_TOP_CACHE != 0 && Q == 1 && MODE == 0;
    __shared__ __attribute__((aligned(16))) uint32_t stk[ROWS * 64 + (TOP ? kTopWords : 0u)];
    const uint32_t* const top_lds = stk + ROWS * 64;
    const bool use_top = TOP && a.bvh.top_off != 0u;
    if (use_top) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.bvh.nodes) + a.bvh.top_off);
        for (uint32_t i = threadIdx.x; i < kTopWords; i += 64u) stk[ROWS * 64 + i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(stk + kPoolCap);   // per ray slot
    float* const sh_u = reinterpret_cast<float*>(stk + kPoolCap + 128);
    float* const sh_v = reinterpret_cast<float*>(stk + kPoolCap + 192);
    uint32_t* const sh_done = stk + kPoolCap + 320;                             // any-hit: an occluder was found
    const int lane = (int)threadIdx.x;
    const bool cull_back = MODE == 0 && (a.ray_flags & 16u) != 0u;   // MIPT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES: det < 0 is a back face
    uint32_t n = a.n_max;
    if (!RG && a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    if (RG && blockIdx.x == 0u) begin_sample(rg.ctr, threadIdx.x, rg.fc.n_paths, rg.n_valid, rg.zero_stats);     // nothing in this launch reads them
    static_assert(alignof(RaygenJob) == 8 && alignof(TraceArgs) == 8, "RaygenJob follows TraceArgs in the kernel-argument segment at the next multiple of 8");
    // the stacks of the still busy lanes must fit the pool: at most (kPoolCap - 168) / ROWS busy lanes at the switch
    const int steal_floor = 64 - (kPoolCap - 168) / ROWS;
    const int steal_min = a.steal_min < 64 ? (a.steal_min > steal_floor ? a.steal_min : steal_floor) : 64;   // 64: switch when the wave is empty
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.blob_bytes, 0x00020000);
    // the launch is cut into tiles of 64 consecutive rays, dealt out to the waves without any atomic
    const uint32_t tiles = (n + 63u) >> 6, tiles_seg = (tiles + kNumSegments - 1) / kNumSegments;
    const uint32_t xcd = blockIdx.x & (kNumSegments - 1), waves_seg = gridDim.x / kNumSegments;   // gridDim.x is a multiple of 8
    Trav T;
    T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f;
#if MIPT_PROBE_STAMP
    unsigned long long pw_wait = 0ull, pw_bulk = 0ull, pw_pool = 0ull, pw_steps = 0ull, pw_rounds = 0ull;
    const unsigned long long pw_t0 = __builtin_amdgcn_s_memtime();
#endif
#if MIPT_PROBE_WAVES
    const unsigned long long wv_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long wv_r0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz, one counter for the whole device
    unsigned long long wv_t1 = wv_t0;
    uint32_t wv_steps = 0u, wv_rounds = 0u;
#endif

#if MIPT_PROBE_UNI
    uint32_t pu[6] = { 0u, 0u, 0u, 0u, 0u, 0u };
#endif
    for (uint32_t k = blockIdx.x / kNumSegments;; k += waves_seg) {
        // ============================================================ bulk: one ray per lane, one tile at a time
        // Groups of tile_group consecutive tiles (256: two rows of 16x4-pixel tiles at 1080p) are dealt to the XCDs in turn: a
        // group keeps an XCD's L2 on one region of the image, and dealing them round balances the XCDs -- with one contiguous
        // eighth of the image per XCD the ones that get its top and bottom (rays that miss everything) idle while the middle
        // ones work: closest-hit kernel of the headline frame 0.514 -> 0.458 ms, imgui_test at 1080p 5.49 -> 4.76 ms
        // (group size 1 / 8 / 64 / 256 / 512 / 1024 / 2048: frame 0.804 / 0.795 / 0.797 / 0.792 / 0.789 / 0.802 / 0.825 ms).
        uint32_t tile = xcd * tiles_seg + k;
        if (a.tile_group != 0u) {
            const uint32_t gq = k / a.tile_group, r = k - gq * a.tile_group;
            tile = (gq * kNumSegments + xcd) * a.tile_group + r;
        } else if (k >= tiles_seg) break;
        if (tile >= tiles) break;
        // the slot's tile: by last launch's costs, heaviest first -- a tile's life varies 1 : 5 (median to worst), and with the
        // launch order of the image the kernel ends in a drain of a third of its duration during which the long tiles that
        // started late run alone (tools/probe_waves.py); results do not depend on which wave walks which tile
        if (a.tile_order != nullptr && a.n_order == tiles) {
            const uint32_t t2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tile_order[tile]);      // the same word in every lane: keep it scalar
            tile = t2 < tiles ? t2 : tile;
        }
        uint32_t cost = 0u;
        const uint32_t ray = tile * 64u + (uint32_t)lane;
        bool has_ray = false;
        if (ray < n) {
            if (RG) {
                // the job is read from the kernel-argument segment HERE, once per tile: as plain kernel arguments its ~50 words stay
                // in SGPRs across the traversal loop, which then spills (106 SGPRs, 17 of them into VGPR lanes, 18 VGPRs to scratch)
#if defined(__HIP_DEVICE_COMPILE__)
                typedef const __attribute__((address_space(4))) RaygenJob* JobPtr;
                JobPtr rp = (JobPtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                     ((sizeof(TraceArgs) + 7u) & ~size_t(7)));
                asm volatile("" : "+s"(rp));
#else
                const RaygenJob* rp = &rg;            // (the host pass only parses this)
#endif
                const FrameConsts fc = rp->fc;
                const PathQueue q = rp->q;
                float4 ro, rd; uint32_t slot_, seed_;
                raygen_path<1>(fc, rp->sample, q, rp->rg_seed, rp->result, ray, ro, rd, slot_, seed_);
                trav_init(T, ro, rd, a.bvh.xf);
            } else {
                trav_init(T, a.ray_o[(size_t)ray * a.stride], a.ray_d[(size_t)ray * a.stride], a.bvh.xf);
            }
            has_ray = T.node != kSentinel;
            if (!has_ray) finish_ray<MODE>(a, T, ray, false);      // empty interval: a miss, no BVH access
            if (use_top && has_ray) T.node = (int)kTopFlag;        // the root's copy
        }
#if MIPT_PROBE_STAMP
        const unsigned long long pw_b0 = __builtin_amdgcn_s_memtime();
#endif
        while (__popcll(__ballot(!has_ray)) < steal_min) {
            for (int bi = 0; bi < a.burst_bulk; ++bi) {
                if (bi != 0 && __ballot(has_ray && T.node != kSentinel) == 0ull) break;
                uint32_t nv_ = 0u, nt_ = 0u;
                if (use_top) stream_step<MODE, false, Q, TOP>(a, blob, stk, lane, has_ray, T, 0, cull_back, nv_, nt_ PROBE_ARG, top_lds);
                else         stream_step<MODE, false, Q, false, MIPT_SCALAR_TOP != 0>(a, blob, stk, lane, has_ray, T, 0, cull_back, nv_, nt_ PROBE_ARG);
#if MIPT_PROBE_UNI
                {   // how coherent are a tile's lanes?  distinct nodes among the inner lanes of this step (counted up to 8)
                    unsigned long long im = __ballot(has_ray && T.node >= 0);
                    const bool leafs = __ballot(has_ray && T.node < 0 && T.node != kSentinel) != 0ull;
                    uint32_t distinct = 0u;
                    while (im != 0ull && distinct < 8u) {
                        const int un = __builtin_amdgcn_readlane(T.node, (int)__builtin_ctzll(im));
                        im &= ~__ballot(has_ray && T.node == un);
                        ++distinct;
                    }
                    pu[0] += 1u; pu[1] += distinct == 1u ? 1u : 0u; pu[2] += (distinct == 1u && !leafs) ? 1u : 0u;
                    pu[3] += distinct; pu[4] += (distinct >= 1u && distinct <= 2u) ? 1u : 0u; pu[5] += (distinct >= 1u && distinct <= 4u) ? 1u : 0u;
                }
#endif
#if MIPT_PROBE_STAMP
                ++pw_steps;
#endif
#if MIPT_PROBE_WAVES
                ++wv_steps;
#endif
                ++cost;
            }
            const bool done = has_ray && T.node == kSentinel;
            if (done) { finish_ray<MODE>(a, T, ray, T.prim != 0xffffffffu); has_ray = false; }
        }
#if MIPT_PROBE_STAMP
        const unsigned long long pw_b1 = __builtin_amdgcn_s_memtime();
        pw_bulk += pw_b1 - pw_b0;
#endif
#if MIPT_PROBE_WAVES
        wv_t1 = __builtin_amdgcn_s_memtime();
#endif
        if (__ballot(has_ray) == 0ull) {                             // the whole tile is finished
            if (lane == 0 && a.tile_cost != nullptr && a.n_order == tiles) record_tile_cost(a, tile, cost);
            continue;
        }

        // ============================================================ tail: one entry pool per wave
        // move the per-lane stacks into the pool, row by row and in place: row e is read before at most
        // (entries so far) + 64 <= (e + 1) * 64 words are written
        int psize = 0;
        for (int e = 0; e < ROWS; ++e) {
            const bool in = has_ray && T.sp > e;
            const unsigned long long m = __ballot(in);
            if (m == 0ull) break;
            uint32_t v = 0u;
            if (in) v = stk[e * 64 + lane];
            if (TOP && in && (int)v >= 0 && (v & kTopFlag) != 0u) v = top_lds[kTopNodes * 8u + (v & (kTopNodes - 1u))];     // a cached node: its real index
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (in) stk[psize + (int)lane_rank(m)] = pool_pack((int)v, (uint32_t)lane);
            psize += __popcll(m);
        }
        {
            const unsigned long long m = __ballot(has_ray);           // the node each busy lane stands at
            int cur = T.node;
            if (TOP && has_ray && cur >= 0 && ((uint32_t)cur & kTopFlag) != 0u) cur = (int)top_lds[kTopNodes * 8u + ((uint32_t)cur & (kTopNodes - 1u))];
            if (has_ray) stk[psize + (int)lane_rank(m)] = pool_pack(cur, (uint32_t)lane);
            psize += __popcll(m);
        }
        // open the merge slots (they overlay the top rows of the stacks: only now) with what each lane has found so far
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        sh_best[lane] = ((unsigned long long)ord_bits(T.tbest + 0.0f) << 32) | (unsigned long long)T.prim;
        sh_u[lane] = T.u; sh_v[lane] = T.v; sh_done[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        while (psize > 0) {
            pool_round<MODE, kPoolCap, Q>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, T, psize, cull_back PROBE_ARG);
#if MIPT_PROBE_STAMP
            ++pw_rounds;
#endif
#if MIPT_PROBE_WAVES
            ++wv_rounds;
#endif
            ++cost;
        }
        if (lane == 0 && a.tile_cost != nullptr && a.n_order == tiles) record_tile_cost(a, tile, cost);
#if MIPT_PROBE_STAMP
        pw_pool += __builtin_amdgcn_s_memtime() - pw_b1;
#endif
        if (has_ray) {
            const unsigned long long key = sh_best[lane];
            T.tbest = ord_float((uint32_t)(key >> 32)); T.prim = (uint32_t)key; T.u = sh_u[lane]; T.v = sh_v[lane];
            finish_ray<MODE>(a, T, ray, MODE == 0 ? T.prim != 0xffffffffu : sh_done[lane] != 0u);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // the next tile's stacks overwrite the merge slots
    }
#if MIPT_PROBE_UNI
    if (lane == 0) for (int q = 0; q < 6; ++q) atomicAdd(&a.ctr->dbg[q], (unsigned long long)pu[q]);
#endif
#if MIPT_PROBE_WAVES
    if (lane == 0 && a.ctr->dbg[0] != 0ull) {
        uint4* rec = reinterpret_cast<uint4*>(a.ctr->dbg[0]);
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        const unsigned long long r2 = __builtin_amdgcn_s_memrealtime();
        rec[blockIdx.x] = make_uint4((uint32_t)(t2 - wv_t0), (uint32_t)(wv_t1 - wv_t0), wv_steps, wv_rounds);
        rec[gridDim.x + blockIdx.x] = make_uint4((uint32_t)wv_r0, (uint32_t)(wv_r0 >> 32), (uint32_t)r2, (uint32_t)(r2 >> 32));
    }
#endif
#if MIPT_PROBE_STAMP
    if (lane == 0) {
        atomicAdd(&a.ctr->dbg[0], __builtin_amdgcn_s_memtime() - pw_t0);     // wave lifetime
        atomicAdd(&a.ctr->dbg[1], pw_wait);                                  // loads issued -> data back, all steps and rounds
        atomicAdd(&a.ctr->dbg[2], pw_bulk);  atomicAdd(&a.ctr->dbg[3], pw_pool);
        atomicAdd(&a.ctr->dbg[4], pw_steps); atomicAdd(&a.ctr->dbg[5], pw_rounds);
    }
#endif
}

// ---- one thread per ray (no refill); same per-ray visiting order ----
template <int MODE, bool COUNT>
__global__ __launch_bounds__(64) void k_trace_simple(TraceArgs a)
{
    __shared__ uint32_t stk_lds[kStackEntries * 64];
    const int lane = (int)threadIdx.x;
    const Stack stk{ stk_lds, lane };
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    uint32_t n_visits = 0, n_tests = 0;
    for (uint32_t item = blockIdx.x * 64u + threadIdx.x; item < n; item += gridDim.x * 64u) {
        const uint32_t ray = item;
        Trav T;
        trav_init(T, a.ray_o[(size_t)ray * a.stride], a.ray_d[(size_t)ray * a.stride], a.bvh.xf);
        bool hit = false;
        while (T.node != kSentinel) {
            while (T.node >= 0) {
                if (COUNT) ++n_visits;
                visit_inner(T, a.bvh.nodes, stk);
            }
            if (T.node != kSentinel) {
                if (visit_leaf<(MODE != 0)>(T, a.bvh.tris, n_tests, MODE == 0 && (a.ray_flags & 16u) != 0u)) { hit = true; break; }
                T.node = stk.pop(T.sp);
            }
        }
        if (MODE == 0) hit = T.prim != 0xffffffffu;
        finish_ray<MODE>(a, T, ray, hit);
    }
    if (COUNT) {
        wave_add64(&a.ctr->node_visits[MODE != 0], n_visits);
        wave_add64(&a.ctr->tri_tests[MODE != 0], n_tests);
    }
}

// ------------------------------------------------------------------------------------------------
// queue bookkeeping
// ------------------------------------------------------------------------------------------------
// what a pass starts with; zero_stats: the first pass of a launch also clears the launch statistics (14 words from closest_rays on)
__device__ __forceinline__ void begin_sample(DeviceCounters* c, uint32_t i, uint32_t n_paths, uint32_t n_valid, uint32_t zero_stats)
{
    // (called by the 64+ threads of a launch's first workgroup.)  The bins of the path kernels' end-of-wave statistics: the previous pass's
    // deepest path is folded into `iterations` and cleared; the first pass of a launch clears the ray sums too.
    unsigned long long tmax = i < kStatBins ? c->bins[i * kStatBinWords + 2u] : 0ull;
    if (i < kStatBins) {
        c->bins[i * kStatBinWords + 2u] = 0ull;
        if (zero_stats) { c->bins[i * kStatBinWords] = 0ull; c->bins[i * kStatBinWords + 1u] = 0ull; }
    }
    if (i < 64u) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_down((int)(uint32_t)(tmax >> 32), off, 64) << 32) | (uint32_t)__shfl_down((int)(uint32_t)tmax, off, 64);
            tmax = o > tmax ? o : tmax;
        }
    }
    if (i == 0) {
        if (zero_stats) {
            unsigned long long* w = &c->closest_rays;
            for (int k = 0; k < ((MIPT_PROBE_STAMP || MIPT_PROBE_UNI) ? 14 : 8); ++k) w[k] = 0ull;     // (dbg[]: only the stamp probe's sums are per launch)
            tmax = 0ull;
        }
        c->closest_rays += n_valid; c->n_cur = n_paths; c->n_next = 0; c->n_shadow = 0; c->bounce = 0;
        c->iterations += (c->tail_max > tmax ? c->tail_max : tmax); c->tail_max = 0;       // the previous pass's tail (iterations is zeroed per launch)
    }
    if (i < kNumSegments) { c->cursor_closest[i * kCursorStride] = 0; c->cursor_shadow[i * kCursorStride] = 0; }
}

// a path-kernel wave's end-of-life statistics: continuation rays, shadow rays, bounces of its deepest path -- into the wave's bin
__device__ __forceinline__ void tail_stats(DeviceCounters* c, uint32_t n_closest, uint32_t n_shadow, uint32_t n_iter)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_closest += (uint32_t)__shfl_down((int)n_closest, off, 64);
        n_shadow += (uint32_t)__shfl_down((int)n_shadow, off, 64);
        const uint32_t v = (uint32_t)__shfl_down((int)n_iter, off, 64);
        n_iter = v > n_iter ? v : n_iter;
    }
    if ((threadIdx.x & 63u) == 0u) {
        unsigned long long* b = &c->bins[(blockIdx.x % kStatBins) * kStatBinWords];
        if (n_closest) atomicAdd(&b[0], (unsigned long long)n_closest);
        if (n_shadow) atomicAdd(&b[1], (unsigned long long)n_shadow);
        if (n_iter) atomicMax(&b[2], (unsigned long long)n_iter);
    }
}

__global__ void k_begin_sample(DeviceCounters* c, uint32_t n_paths, uint32_t n_valid)
{
    begin_sample(c, threadIdx.x, n_paths, n_valid, 0u);
}

__device__ __forceinline__ void advance(DeviceCounters* c, uint32_t i, uint32_t* host_live)
{
    if (i == 0) {
        const uint32_t live = c->n_next;
        if (c->n_cur) c->iterations += 1;
        c->closest_rays += live; c->shadow_rays += c->n_shadow; c->n_cur = live; c->n_next = 0; c->n_shadow = 0; c->bounce += 1;
        if (host_live) __hip_atomic_store(host_live, live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (i < kNumSegments) { c->cursor_closest[i * kCursorStride] = 0; c->cursor_shadow[i * kCursorStride] = 0; }
}

__global__ void k_advance(DeviceCounters* c, uint32_t* host_live)
{
    advance(c, threadIdx.x, host_live);
}

// ------------------------------------------------------------------------------------------------
// raygen: optixPathTracer.cu:249-281.  Path i -> pixel by 16x4-pixel tiles (one wave = one tile; fc.tile_shift) for full
// frames, or by the StaticWorkDistribution map for a rank's shard.  The raygen-local seed advances
// by the two jitter draws only (the path's own stream is discarded, :260-278), so it is kept per slot.
// ------------------------------------------------------------------------------------------------
// With fc.batch_spp != 0 every sample of the launch is a path of its own (path i -> sample i / paths_per_sample)
// with its own accumulator slot `sample * n_slots + slot`; the raygen seed of sample s is reached by 2*s LCG
// steps from the pixel's tea<4> seed, exactly the state the sequential loop would have.
// The first workgroup also does the pass's queue bookkeeping (begin_sample: a launch of its own costs ~8 us with its gap).
// one path of the pass: writes its queue entry, the raygen seed and (first sample) the cleared accumulator; returns the ray.
// WR = 1 (the tiled kernel's own raygen; read by the first shade launch of the pass only, ShadeArgs::primary): the entry is
// direction + (slot, seed) -- 24 bytes instead of 56 -- because origin, interval, attenuation and depth of a primary path are
// constants, and the accumulator is not cleared: the shade launch starts from zero instead of reading it.
// WR = 2 (the path kernels' own raygen): no entry at all, the path stays in the lane's registers (slot_out, seed_out).
template <int WR>
__device__ __forceinline__ bool raygen_path(const FrameConsts& fc, uint32_t sample, const PathQueue& q, uint32_t* rg_seed, float4* result,
                                            uint32_t i, float4& ro, float4& rd, uint32_t& slot_out, uint32_t& seed_out)
{
    constexpr bool COMPACT = WR != 0;
    slot_out = kInvalidSlot; seed_out = 0u;
    uint32_t i0 = i;
    if (fc.batch_spp != 0u) { sample = i / fc.paths_per_sample; i0 = i - sample * fc.paths_per_sample; }
    int px, py; uint32_t slot; bool valid;
    if (fc.world == 0u) {
        const uint32_t ts = fc.tile_shift, tw = 1u << ts, th = 64u >> ts;
        const uint32_t tiles_x = (fc.width + tw - 1u) >> ts;
        const uint32_t tile = i0 >> 6, in = i0 & 63u;
        const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        px = (int)(tx * tw + (in & (tw - 1u)));
        py = (int)(ty * th + (in >> ts));
        valid = (uint32_t)px < fc.width && (uint32_t)py < fc.height;
        slot = (uint32_t)py * fc.width + (uint32_t)px;
    } else {
        shard_pixel(fc.width, fc.world, fc.rank, i0, px, py);
        valid = (uint32_t)px < fc.width && (uint32_t)py < fc.height;
        slot = i0;
    }
    if (fc.batch_spp != 0u) slot += sample * fc.n_slots;
    if (!valid) {
        // padding lane: a ray with an empty interval, killed by shade
        ro = make_float4(0.f, 0.f, 0.f, 1.0f);
        rd = make_float4(0.f, 0.f, 1.f, 0.0f);
        if (!COMPACT) { q.ray_o[i] = ro; q.ray_d[i] = rd; q.state_a[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (WR != 2) q.state_b[i] = make_uint2(kInvalidSlot, 0u);
        if (fc.world != 0u && (sample == 0u || fc.batch_spp != 0u)) result[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        return false;
    }
    uint32_t seed;
    if (fc.batch_spp != 0u) {
        seed = tea4((uint32_t)py * fc.width + (uint32_t)px, fc.subframe_index);
        for (uint32_t k = 0; k < 2u * sample; ++k) seed = 1664525u * seed + 1013904223u;
    } else {
        seed = sample == 0u ? tea4((uint32_t)py * fc.width + (uint32_t)px, fc.subframe_index) : rg_seed[slot];
    }
    const float jx = rnd(seed);
    const float jy = rnd(seed);
    if (fc.batch_spp == 0u) rg_seed[slot] = seed;
    const float dx = 2.0f * (((float)px + jx) / (float)fc.width) - 1.0f;
    const float dy = 2.0f * (((float)py + jy) / (float)fc.height) - 1.0f;
    const f3 U = mk3(fc.U[0], fc.U[1], fc.U[2]), V = mk3(fc.V[0], fc.V[1], fc.V[2]), W = mk3(fc.W[0], fc.W[1], fc.W[2]);
    const f3 dir = normalize(dx * U + dy * V + W);
    ro = make_float4(fc.eye[0], fc.eye[1], fc.eye[2], 0.01f);
    rd = make_float4(dir.x, dir.y, dir.z, 1e16f);
    slot_out = slot; seed_out = seed;
    if (WR == 2) return true;
    q.ray_d[i]   = rd;
    if (COMPACT) {
        q.state_b[i] = make_uint2(slot, seed);
    } else {
        q.ray_o[i]   = ro;
        q.state_a[i] = make_float4(1.0f, 1.0f, 1.0f, u2f(seed));
        q.state_b[i] = make_uint2(slot, 0u);
        if (sample == 0u || fc.batch_spp != 0u) result[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return true;
}

__global__ __launch_bounds__(256) void k_raygen(FrameConsts fc, uint32_t sample, PathQueue q, uint32_t* rg_seed, float4* result,
                                                DeviceCounters* ctr, uint32_t n_valid, uint32_t zero_stats, uint32_t* zero_words, uint32_t n_zero)
{
    if (ctr && blockIdx.x == 0u) begin_sample(ctr, threadIdx.x, fc.n_paths, n_valid, zero_stats);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_zero) zero_words[i] = 0u;
    if (i >= fc.n_paths) return;
    float4 ro, rd; uint32_t slot, seed;
    raygen_path<0>(fc, sample, q, rg_seed, result, i, ro, rd, slot, seed);
}

// ------------------------------------------------------------------------------------------------
// shade: __closesthit__radiance (optixPathTracer.cu:338-413), __miss__radiance (:323-335) and the
// caller's per-bounce bookkeeping (:291-303).  Emits the NEE shadow ray and the continuation ray
// into compacted queues.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cosine_sample_hemisphere(float u1, float u2, f3& p)
{
    const float r   = __builtin_sqrtf(u1);
    const float phi = (2.0f * 3.14159265358979323846f) * u2;
    float sn, cs;
    det_sincos(phi, sn, cs);
    p.x = r * cs;
    p.y = r * sn;
    p.z = __builtin_sqrtf(__builtin_fmaxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
}

__device__ __forceinline__ f3 onb_inverse_transform(f3 n, f3 p)
{
    f3 b;
    if (__builtin_fabsf(n.x) > __builtin_fabsf(n.z)) b = mk3(-n.y, n.x, 0.0f);
    else                                             b = mk3(0.0f, -n.z, n.y);
    b = normalize(b);
    const f3 t = cross(b, n);
    return p.x * t + p.y * b + p.z * n;
}

struct ShadeArgs {
    FrameConsts             fc;
    DeviceBvh               bvh;
    const MiptHitGroupData* hitgroups;
    uint32_t                n_hitgroups;
    PathQueue               cur;
    const float4*           hits;
    PathQueue               next;
    ShadowQueue             shadow;
    float4*                 result;
    DeviceCounters*         ctr;
    uint32_t                n_max;
    const void*             common_vertices;   // non-null when every hit-group record points at the same vertex buffer
    TileOrderJob            order;             // n_chunks != 0: the first n_chunks workgroups make the next launch's tile order
    ResolveJob              resolve;           // on: a path that ends in this launch gets its pixel written (ResolveJob)
    uint32_t                primary;           // k_shade: bit 0: the entries are raygen_path<true>'s (state_b = slot, seed; depth 0, attenuation 1,
                                               // origin = the eye); bit 1: the accumulators start at zero (first sample of a launch)
};

struct ShadeOut {
    bool   push_next, push_shadow;
    float4 nro, nrd, nsa; uint2 nsb;      // continuation ray + path state
    float4 sro, srd, sc;                  // shadow ray + its contribution (.w = result slot)
};

// One path at one bounce: the arithmetic of the miss / closest-hit programs and of the caller's bookkeeping, shared by
// k_shade (wavefront bounces) and k_tail (the last few paths) so that both produce the same bits.  `res` is the path's
// accumulator (read and written back by the caller); load_ray fetches the ray only when the path hit something.
// LAST: the caller knows that the depth cap ends every path at this bounce (max_depth == 1: the headline frame) -- the continuation
// direction (a sincos, two square roots, an orthonormal basis) and the roulette draw are then dead work; the two draws that pick the
// direction still advance the seed, so the light sample is the same.
template <bool LAST = false, class LoadRay>
__device__ __forceinline__ void shade_path(const ShadeArgs& a, float4 hit, float4 sa, uint32_t slot, uint32_t depth, float4& res,
                                           ShadeOut& o, LoadRay load_ray)
{
    o.push_next = false; o.push_shadow = false;
    f3 att = mk3(sa.x, sa.y, sa.z);
    uint32_t seed = f2u(sa.w);
    if (hit.x < 0.0f) {
        // miss: radiance = bg, emitted = 0, done (optixPathTracer.cu:323-335)
        const f3 rad = mk3(a.fc.bg[0], a.fc.bg[1], a.fc.bg[2]) * att;
        res.x += 0.0f; res.y += 0.0f; res.z += 0.0f;
        res.x += rad.x; res.y += rad.y; res.z += rad.z;
        return;
    }
    float4 ro4, rd4;
    load_ray(ro4, rd4);
    const f3 ray_o = mk3(ro4.x, ro4.y, ro4.z), ray_d = mk3(rd4.x, rd4.y, rd4.z);
    const uint32_t prim = f2u(hit.y);
    uint32_t m = a.bvh.mat_index[prim];
    m = m < a.n_hitgroups ? m : a.n_hitgroups - 1u;
    // all records usually share one vertex buffer (optixPathTracer.cpp:877): then the fetch
    // does not have to wait for the record
    const float4* vb0 = reinterpret_cast<const float4*>(a.common_vertices);
    const MiptHitGroupData rec = a.hitgroups[m];
    const float4* vb = (vb0 ? vb0 : reinterpret_cast<const float4*>(rec.vertices)) + (size_t)prim * 3u;
    const float4 a0 = vb[0], a1 = vb[1], a2 = vb[2];
    const f3 v0 = mk3(a0.x, a0.y, a0.z), v1 = mk3(a1.x, a1.y, a1.z), v2 = mk3(a2.x, a2.y, a2.z);
    const f3 N0 = normalize(cross(v1 - v0, v2 - v0));
    const f3 N  = faceforward(N0, -ray_d, N0);
    const f3 P  = ray_o + hit.x * ray_d;

    // emitted only for camera rays (:352-355); added before the radiance term (:291)
    if (depth == 0u) { res.x += rec.emission_color[0]; res.y += rec.emission_color[1]; res.z += rec.emission_color[2]; }
    else             { res.x += 0.0f; res.y += 0.0f; res.z += 0.0f; }

    const float z1 = rnd(seed);
    const float z2 = rnd(seed);
    f3 new_dir = mk3(0.0f, 0.0f, 0.0f);
    if (!LAST) {
        f3 w_in;
        cosine_sample_hemisphere(z1, z2, w_in);
        new_dir = onb_inverse_transform(N, w_in);
    }
    att = att * mk3(rec.diffuse_color[0], rec.diffuse_color[1], rec.diffuse_color[2]);

    const float l1 = rnd(seed);
    const float l2 = rnd(seed);
    const MiptParallelogramLight& lt = a.fc.light;
    const f3 lv1 = mk3(lt.v1[0], lt.v1[1], lt.v1[2]), lv2 = mk3(lt.v2[0], lt.v2[1], lt.v2[2]);
    const f3 light_pos = mk3(lt.corner[0], lt.corner[1], lt.corner[2]) + lv1 * l1 + lv2 * l2;
    const f3 toL = light_pos - P;
    const float Ldist = length(toL);
    const f3 L = normalize(toL);
    const float nDl  = dot(N, L);
    const float LnDl = -dot(mk3(lt.normal[0], lt.normal[1], lt.normal[2]), L);
    if (nDl > 0.0f && LnDl > 0.0f) {
        const float A = length(cross(lv1, lv2));
        const float weight = nDl * LnDl * A / (3.14159265358979323846f * Ldist * Ldist);
        const f3 radiance = mk3(lt.emission[0], lt.emission[1], lt.emission[2]) * weight;
        const f3 c = radiance * att;
        o.push_shadow = true;
        o.sro = make_float4(P.x, P.y, P.z, 0.01f);
        o.srd = make_float4(L.x, L.y, L.z, Ldist - 0.01f);
        o.sc  = make_float4(c.x, c.y, c.z, u2f(slot));
    }
    if (LAST) return;
    // Russian roulette + continuation (:294-303)
    const float lum = dot(att, mk3(0.30f, 0.59f, 0.11f));
    const bool done = rnd(seed) > lum;
    const bool capped = a.fc.max_depth != 0u && depth + 1u >= a.fc.max_depth;
    if (!done && !capped) {
        const float inv = 1.0f / lum;
        att = att * inv;
        o.push_next = true;
        o.nro = make_float4(P.x, P.y, P.z, 0.01f);
        o.nrd = make_float4(new_dir.x, new_dir.y, new_dir.z, 1e16f);
        o.nsa = make_float4(att.x, att.y, att.z, u2f(seed));
        o.nsb = make_uint2(slot, depth + 1u);
    }
}

// One global atomic per 512-path tile and queue (not per wave): the queue-length words are single
// addresses, and ~32 k returning atomics on one word serialise at ~88 per microsecond (measured: the
// per-wave version spent 0.37 of its 0.44 ms there).
// The order of the next primary launch's tiles, one chunk of kOrderChunk tiles per workgroup: slot of a tile = tiles of heavier
// classes (all chunks) + tiles of its class in earlier chunks + tiles of its class before it in this chunk.
__device__ void tile_order_job(const TileOrderJob& j, uint32_t chunk)
{
    constexpr uint32_t kSegs = kOrderChunk / 64u;
    __shared__ uint32_t s_tot[kOrderClasses], s_pre[kOrderClasses], s_first[kOrderClasses], s_seg[kSegs][kOrderClasses], s_ok;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, waves = blockDim.x >> 6;
    if (tid < kOrderClasses) { s_tot[tid] = 0u; s_pre[tid] = 0u; }
    if (j.zero != nullptr && tid < kOrderClasses) j.zero[chunk * kOrderClasses + tid] = 0u;
    __syncthreads();
    {   // class totals over all chunks / over the chunks before this one
        const uint32_t k = tid % kOrderClasses, stripe = tid / kOrderClasses, stripes = blockDim.x / kOrderClasses;
        uint32_t tot = 0u, pre = 0u;
        for (uint32_t c = stripe; c < j.n_chunks; c += stripes) {
            const uint32_t h = j.hist[c * kOrderClasses + k];
            tot += h; pre += c < chunk ? h : 0u;
        }
        if (tot) atomicAdd(&s_tot[k], tot);
        if (pre) atomicAdd(&s_pre[k], pre);
    }
    // tiles of every class in every 64-tile segment of the chunk
    for (uint32_t g = wave; g < kSegs; g += waves) {
        const uint32_t tile = chunk * kOrderChunk + g * 64u + lane;
        const uint32_t cl = tile < j.n_tiles ? j.cls[tile] : 0xffu;
        uint32_t mine = 0u;
        for (uint32_t k = 0; k < kOrderClasses; ++k) { const uint32_t c = (uint32_t)__popcll(__ballot(cl == k)); if (lane == k) mine = c; }
        if (lane < kOrderClasses) s_seg[g][lane] = mine;
    }
    __syncthreads();
    if (tid < kOrderClasses) {
        uint32_t first = s_pre[tid];
        for (uint32_t k = 0; k < tid; ++k) first += s_tot[k];
        s_first[tid] = first;
    }
    // The ranks below are a permutation only if the launch before this one recorded EVERY tile (the histograms then sum to the tile
    // count).  If it did not -- another kernel form ran than the host assumed -- the class bytes are stale: emit image order rather
    // than an order with duplicate and missing tiles, which would leave pixels untraced.
    if (tid == 0u) {
        uint32_t total = 0u;
        for (uint32_t k = 0; k < kOrderClasses; ++k) total += s_tot[k];
        s_ok = total == j.n_tiles ? 1u : 0u;
    }
    // The class width follows the frame: the heaviest class (it is open-ended, its tiles stay in image order) should hold about a
    // third of the tiles that cost anything -- on the headline frame that is 6 units (0.36 of them; 5 / 6 / 7 units: 0.355 /
    // 0.344 / 0.349 ms), and a fixed unit would mean nothing on a scene whose tiles cost a tenth or ten times as much.  One step
    // per launch, with a dead band, so a steady frame keeps its width.
    if (j.width != nullptr && chunk == 0u && tid == 0u) {
        const uint32_t busy = j.n_tiles - s_tot[kOrderClasses - 1u];            // tiles above the lightest class
        uint32_t w = *j.width;
        if (busy >= 64u) {
            if (s_tot[0] * 100u > busy * 42u && w < 4096u) ++w;
            else if (s_tot[0] * 100u < busy * 30u && w > 1u) --w;
        }
        *j.width = w;
    }
    __syncthreads();
    if (s_ok == 0u) {
        for (uint32_t t = chunk * kOrderChunk + tid; t < (chunk + 1u) * kOrderChunk && t < j.n_tiles; t += blockDim.x) j.order[t] = t;
        return;
    }
    for (uint32_t g = wave; g < kSegs; g += waves) {
        const uint32_t tile = chunk * kOrderChunk + g * 64u + lane;
        const uint32_t cl = tile < j.n_tiles ? j.cls[tile] : 0xffu;
        uint32_t rank = 0u;
        for (uint32_t k = 0; k < kOrderClasses; ++k) {
            const unsigned long long m = __ballot(cl == k);
            if (cl == k) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
        if (cl < kOrderClasses) {
            uint32_t slot = s_first[cl] + rank;
            for (uint32_t g2 = 0; g2 < g; ++g2) slot += s_seg[g2][cl];
            if (slot < j.n_tiles) j.order[slot] = tile;
        }
    }
}

template <bool LAST>           // LAST: max_depth == 1 -- every path ends at this bounce (shade_path<LAST>)
__global__ __launch_bounds__(kShadeBlock) void k_shade(ShadeArgs a)
{
    __shared__ uint32_t s_cnt[2], s_base[2];
    if (blockIdx.x < a.order.n_chunks) { tile_order_job(a.order, blockIdx.x); return; }
    const uint32_t first_block = a.order.n_chunks, stride = gridDim.x - a.order.n_chunks;
    uint32_t n = a.ctr->n_cur;
    n = n < a.n_max ? n : a.n_max;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t tile = blockIdx.x - first_block; tile * kShadeBlock < n; tile += stride) {
        const uint32_t i = tile * kShadeBlock + threadIdx.x;
        ShadeOut o;
        o.push_next = false; o.push_shadow = false;
        if (i < n) {
            const uint2 sb = a.cur.state_b[i];
            const bool compact = (a.primary & 1u) != 0u;
            const uint32_t slot = sb.x, depth = compact ? 0u : sb.y;
            if (slot != kInvalidSlot) {
                const float4 hit = a.hits[i];
                const float4 sa = compact ? make_float4(1.0f, 1.0f, 1.0f, u2f(sb.y)) : a.cur.state_a[i];
                float4 res = (a.primary & 2u) != 0u ? make_float4(0.f, 0.f, 0.f, 0.f) : a.result[slot];
                shade_path<LAST>(a, hit, sa, slot, depth, res, o, [&](float4& ro4, float4& rd4) {
                    ro4 = compact ? make_float4(a.fc.eye[0], a.fc.eye[1], a.fc.eye[2], 0.01f) : a.cur.ray_o[i];
                    rd4 = a.cur.ray_d[i];
                });
                if (a.resolve.on && !o.push_next) {
                    // the path ends here: with a last occlusion ray pending the any-hit launch finishes the pixel, else it is finished
                    if (o.push_shadow) { o.sc.w = u2f(f2u(o.sc.w) | 0x80000000u); a.result[slot] = res; }
                    else resolve_pixel(a.resolve, slot, res);
                } else {
                    a.result[slot] = res;
                }
            }
        }
        const bool push_shadow = o.push_shadow, push_next = o.push_next;
        // ---- compacting appends, aggregated over the block
        const unsigned long long ms = __ballot(push_shadow), mn = __ballot(push_next);
        uint32_t wo_s = 0, wo_n = 0;
        if ((threadIdx.x & 63u) == 0u) {
            if (ms) wo_s = atomicAdd(&s_cnt[0], (uint32_t)__popcll(ms));
            if (mn) wo_n = atomicAdd(&s_cnt[1], (uint32_t)__popcll(mn));
        }
        wo_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wo_s);
        wo_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)wo_n);
        __syncthreads();
        if (threadIdx.x < 2) {
            const uint32_t c = s_cnt[threadIdx.x];
            s_base[threadIdx.x] = c ? atomicAdd(threadIdx.x == 0 ? &a.ctr->n_shadow : &a.ctr->n_next, c) : 0u;
            s_cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        if (push_shadow) {
            const uint32_t js = s_base[0] + wo_s + lane_rank(ms);
            a.shadow.ray_o[js] = o.sro; a.shadow.ray_d[js] = o.srd; a.shadow.contrib[js] = o.sc;
        }
        if (push_next) {
            const uint32_t jn = s_base[1] + wo_n + lane_rank(mn);
            a.next.ray_o[jn] = o.nro; a.next.ray_d[jn] = o.nrd; a.next.state_a[jn] = o.nsa; a.next.state_b[jn] = o.nsb;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// tail: when Russian roulette has left only a few thousand paths, a wavefront bounce is five nearly empty kernels
// (~40 us).  k_tail runs every remaining path to its end in ONE launch, one path per lane -- the raygen loop of
// optixPathTracer.cu:283-305 as the reference has it -- with the per-ray traversal of k_trace_simple and shade_path,
// so the accumulators receive the same terms in the same order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_tail(ShadeArgs a)
{
    __shared__ uint32_t stk_lds[kStackEntries * 64];
    const int lane = (int)threadIdx.x;
    const Stack stk{ stk_lds, lane };
    uint32_t n = a.ctr->n_cur;
    n = n < a.n_max ? n : a.n_max;
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    uint32_t n_closest = 0, n_shadow = 0, n_iter = 0, n_tests = 0;
    uint32_t slot = kInvalidSlot, depth = 0;
    if (i < n) { const uint2 sb = a.cur.state_b[i]; slot = sb.x; depth = sb.y; }
    if (slot != kInvalidSlot) {
        float4 ro4 = a.cur.ray_o[i], rd4 = a.cur.ray_d[i], sa = a.cur.state_a[i];
        float4 res = a.result[slot];
        for (;;) {
            Trav T;
            trav_init(T, ro4, rd4, a.bvh.xf);
            while (T.node != kSentinel) {
                while (T.node >= 0) visit_inner(T, a.bvh.nodes, stk);
                if (T.node != kSentinel) { (void)visit_leaf<false>(T, a.bvh.tris, n_tests, false); T.node = stk.pop(T.sp); }
            }
            const float4 hit = T.prim != 0xffffffffu ? make_float4(T.tbest, u2f(T.prim), T.u, T.v)
                                                     : make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
            ++n_iter;
            ShadeOut o;
            shade_path(a, hit, sa, slot, depth, res, o, [&](float4& o4, float4& d4) { o4 = ro4; d4 = rd4; });
            if (o.push_shadow) {
                ++n_shadow;
                trav_init(T, o.sro, o.srd, a.bvh.xf);
                bool occluded = false;
                while (T.node != kSentinel) {
                    while (T.node >= 0) visit_inner(T, a.bvh.nodes, stk);
                    if (T.node != kSentinel) {
                        if (visit_leaf<true>(T, a.bvh.tris, n_tests, false)) { occluded = true; break; }
                        T.node = stk.pop(T.sp);
                    }
                }
                if (!occluded) { res.x += o.sc.x; res.y += o.sc.y; res.z += o.sc.z; }      // finish_ray<2>
            }
            if (!o.push_next) break;
            ++n_closest;
            ro4 = o.nro; rd4 = o.nrd; sa = o.nsa; depth = o.nsb.y;
        }
        if (a.resolve.on) resolve_pixel(a.resolve, slot, res); else a.result[slot] = res;
    }
    tail_stats(a.ctr, n_closest, n_shadow, n_iter);
}

// ---- the same loop with the wave's 64 rays traversed together on the entry pool (pool_round): a bounce then lasts as long
// as the deepest chain of dependent node fetches, not as long as the longest ray's whole visiting sequence (1 M-triangle
// scene: one bounce of 65 k paths 0.51 -> 0.1 ms).  Hit records are those of the sequential traversal (k_trace_pool's
// argument).  Measured and dropped: one ray per lane first (stream_step) and the pool only for the tail of the tile, as in
// k_trace_pool's tiled form -- 164 VGPRs, and capped at 128 or 96 it is slower than this (Cornell 512x512 0.41 / 0.53 vs
// 0.38 ms; the headline frame run this way 1.05 / 1.08 vs 0.95 ms).
// The path kernels' arguments.  rg.ctr != null: the pass starts in this launch, which then makes the primary paths itself
// (raygen_path<2>: path i of the pass instead of queue entry i, nothing written but the raygen seed) and does the pass's queue
// bookkeeping in its first workgroup, instead of a k_raygen launch in front of it.  That bookkeeping reads, clears and adds to the
// launch statistics, to which every wave adds at its end -- on a small scene a wave can be through its paths before the first
// workgroup is through eight dependent global accesses (seen: one continuation ray missing from closest_rays on Cornell 96x64).
// So the first workgroup publishes `epoch` (a number the host counts up per such launch) in DeviceCounters when it is done, and a
// wave looks for it before its own additions: the first workgroup is dispatched first and waits for nobody, so the wait ends
// (and is bounded anyway); it is almost never entered.
struct TailArgs {
    ShadeArgs a;
    uint32_t  ppw, refill;
    RaygenJob rg;
    uint32_t  epoch;
};

// (bit 31 of `epoch` is the host's fault injection, env MIPT_DEBUG_EPOCH_FAULT at context creation: the first workgroup then
//  publishes nothing and every wave runs into the time-out below -- tests/test_gpu_parity.py checks that this is reported)
__device__ __forceinline__ void tail_publish_epoch(DeviceCounters* c, uint32_t epoch)
{
    if (threadIdx.x == 0u && (epoch >> 31) == 0u) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&c->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// The wait is bounded (a wave must never hang the device): ~2^17 polls of 512 cycles, tens of milliseconds.  Running out of it is
// NOT silent: the wave raises kFaultEpochTimeout in the context's pinned fault word, and the host's next synchronising call fails.
__device__ __forceinline__ void tail_await_epoch(DeviceCounters* c, uint32_t epoch)
{
    if (threadIdx.x == 0u) {
        bool seen = false;
        for (int spin = 0; spin < (1 << 17); ++spin) {
            if (__hip_atomic_load(&c->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) { seen = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!seen) {
            uint32_t* f = __hip_atomic_load(&c->fault_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f) __hip_atomic_fetch_or(f, kFaultEpochTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// one path of a path kernel: from the queue, or made here
__device__ __forceinline__ void tail_load_path(const TailArgs& k, const bool use_rg, const uint32_t j, uint32_t& slot, uint32_t& depth,
                                               float4& ro4, float4& rd4, float4& sa, float4& res)
{
    const ShadeArgs& a = k.a;
    if (use_rg) {
        auto rp = MIPT_RG_JOB_PTR(TailArgs, rg, k.rg);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(rp));
#endif
        const FrameConsts fc = rp->fc;
        const uint32_t sample = rp->sample;
        uint32_t seed = 0u;
        const bool valid = raygen_path<2>(fc, sample, PathQueue{}, rp->rg_seed, rp->result, j, ro4, rd4, slot, seed);
        depth = 0u;
        sa = make_float4(1.0f, 1.0f, 1.0f, u2f(seed));
        res = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && !(sample == 0u || fc.batch_spp != 0u)) res = rp->result[slot];       // later samples add to the accumulator
    } else {
        const uint2 sb = a.cur.state_b[j];
        slot = sb.x; depth = sb.y;
        if (slot != kInvalidSlot) { ro4 = a.cur.ray_o[j]; rd4 = a.cur.ray_d[j]; sa = a.cur.state_a[j]; res = a.result[slot]; }
    }
}

template <int Q>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tail_pool(TailArgs k)      // (<= 128 VGPRs: four waves per SIMD, as it has run since round 1)
{
    const ShadeArgs& a = k.a;
    const uint32_t ppw = k.ppw;
    const bool use_rg = k.rg.ctr != nullptr;
    if (use_rg && blockIdx.x == 0u) { begin_sample(k.rg.ctr, threadIdx.x, k.rg.fc.n_paths, k.rg.n_valid, k.rg.zero_stats); tail_publish_epoch(k.rg.ctr, k.epoch); }
    constexpr int kPoolCap = 24 * 64 - 384;
    __shared__ __attribute__((aligned(16))) uint32_t stk[24 * 64];                 // (sh_best: 64-bit LDS atomics)
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(stk + kPoolCap);
    float* const sh_u = reinterpret_cast<float*>(stk + kPoolCap + 128);
    float* const sh_v = reinterpret_cast<float*>(stk + kPoolCap + 192);
    uint32_t* const sh_done = stk + kPoolCap + 320;
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (!use_rg) { const uint32_t m = a.ctr->n_cur; n = m < n ? m : n; }
    const uint32_t i = blockIdx.x * ppw + threadIdx.x;
    uint32_t n_closest = 0, n_shadow = 0, n_iter = 0;
    uint32_t slot = kInvalidSlot, depth = 0;
    float4 ro4 = make_float4(0.f, 0.f, 0.f, 0.f), rd4 = ro4, sa = ro4, res = ro4;
    if (i < n && threadIdx.x < ppw) tail_load_path(k, use_rg, i, slot, depth, ro4, rd4, sa, res);
    bool active = slot != kInvalidSlot;
    while (__ballot(active) != 0ull) {
        Trav T;
        trav_init(T, ro4, rd4, a.bvh.xf);
        tail_trace<0, Q>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, T, active && T.node != kSentinel);
        const unsigned long long key = sh_best[lane];
        const uint32_t prim = (uint32_t)key;
        const float4 hit = prim != 0xffffffffu ? make_float4(ord_float((uint32_t)(key >> 32)), u2f(prim), sh_u[lane], sh_v[lane])
                                               : make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
        ShadeOut o;
        o.push_next = false; o.push_shadow = false;
        if (active) {
            ++n_iter;
            shade_path(a, hit, sa, slot, depth, res, o, [&](float4& o4, float4& d4) { o4 = ro4; d4 = rd4; });
        }
        const bool sh = active && o.push_shadow;
        if (__ballot(sh) != 0ull) {
            Trav S;
            trav_init(S, sh ? o.sro : ro4, sh ? o.srd : rd4, a.bvh.xf);
            tail_trace<2, Q>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, S, sh && S.node != kSentinel);
            if (sh) {
                ++n_shadow;
                if (sh_done[lane] == 0u) { res.x += o.sc.x; res.y += o.sc.y; res.z += o.sc.z; }      // finish_ray<2>
            }
        }
        active = active && o.push_next;
        if (active) { ++n_closest; ro4 = o.nro; rd4 = o.nrd; sa = o.nsa; depth = o.nsb.y; }
    }
    if (slot != kInvalidSlot) { if (a.resolve.on) resolve_pixel(a.resolve, slot, res); else a.result[slot] = res; }
    if (use_rg) tail_await_epoch(a.ctr, k.epoch);
    tail_stats(a.ctr, n_closest, n_shadow, n_iter);
}

// ---- k_tail_pool1: k_tail_pool for passes whose depth cap is ONE bounce (the headline frame's shards at N >= 4) -----------------------
// Same traversal, same shading arithmetic, no loop: nothing of the path outlives its first bounce, so the ray, its attenuation and the
// depth are dead once the shadow ray exists, and the continuation direction is never made (shade_path<LAST>).  The point is
// registers: k_tail_pool needs 118 VGPRs (16 waves per CU) and a rank's 1/8 shard -- 8 100 waves of 32 paths -- runs as two
// generations of chains of ~75 dependent pool rounds; fewer registers put more of them on the chip at once.
#ifndef MIPT_TAIL1_WAVES
#define MIPT_TAIL1_WAVES 6
#endif
template <int Q>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MIPT_TAIL1_WAVES, MIPT_TAIL1_WAVES))) void k_tail_pool1(TailArgs k)
{
    const ShadeArgs& a = k.a;
    const uint32_t ppw = k.ppw;
    const bool use_rg = k.rg.ctr != nullptr;
    if (use_rg && blockIdx.x == 0u) { begin_sample(k.rg.ctr, threadIdx.x, k.rg.fc.n_paths, k.rg.n_valid, k.rg.zero_stats); tail_publish_epoch(k.rg.ctr, k.epoch); }
    constexpr int kPoolCap = 24 * 64 - 384;
    __shared__ __attribute__((aligned(16))) uint32_t stk[24 * 64];                 // (sh_best: 64-bit LDS atomics)
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(stk + kPoolCap);
    float* const sh_u = reinterpret_cast<float*>(stk + kPoolCap + 128);
    float* const sh_v = reinterpret_cast<float*>(stk + kPoolCap + 192);
    uint32_t* const sh_done = stk + kPoolCap + 320;
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (!use_rg) { const uint32_t m = a.ctr->n_cur; n = m < n ? m : n; }
    const uint32_t i = blockIdx.x * ppw + threadIdx.x;
    uint32_t slot = kInvalidSlot, depth = 0;
    float4 ro4 = make_float4(0.f, 0.f, 0.f, 0.f), rd4 = ro4, sa = ro4, res = ro4;
    if (i < n && threadIdx.x < ppw) tail_load_path(k, use_rg, i, slot, depth, ro4, rd4, sa, res);
    const bool active = slot != kInvalidSlot;
    ShadeOut o;
    o.push_next = false; o.push_shadow = false;
    {
        Trav T;
        trav_init(T, ro4, rd4, a.bvh.xf);
        tail_trace<0, Q>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, T, active && T.node != kSentinel);
        const unsigned long long key = sh_best[lane];
        const uint32_t prim = (uint32_t)key;
        const float4 hit = prim != 0xffffffffu ? make_float4(ord_float((uint32_t)(key >> 32)), u2f(prim), sh_u[lane], sh_v[lane])
                                               : make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
        if (active) shade_path<true>(a, hit, sa, slot, depth, res, o, [&](float4& o4, float4& d4) { o4 = ro4; d4 = rd4; });
    }
    const bool sh = active && o.push_shadow;
    if (__ballot(sh) != 0ull) {
        Trav S;
        trav_init(S, sh ? o.sro : make_float4(0.f, 0.f, 0.f, 1.0f), sh ? o.srd : make_float4(0.f, 0.f, 1.f, 0.0f), a.bvh.xf);
        tail_trace<2, Q>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, S, sh && S.node != kSentinel);
        if (sh && sh_done[lane] == 0u) { res.x += o.sc.x; res.y += o.sc.y; res.z += o.sc.z; }      // finish_ray<2>
    }
    if (active) { if (a.resolve.on) resolve_pixel(a.resolve, slot, res); else a.result[slot] = res; }
#ifndef MIPT_PROBE_NO_TAIL_STATS
    if (use_rg) tail_await_epoch(a.ctr, k.epoch);
    tail_stats(a.ctr, 0u, sh ? 1u : 0u, active ? 1u : 0u);
#endif
}

// ---- k_frame1: a one-bounce pass (depth cap 1: the headline frame) as ONE launch of tile waves (round 4) -------------------------------
// k_trace_pool's tile wave -- raygen for its 64 paths, one ray per lane over the packed BVH2, then the wave's entry pool -- keeps going
// where the wavefront form ends the kernel: with every lane's hit in registers it shades its 64 paths (shade_path<LAST>, the body of
// k_shade), traces the tile's occlusion rays (a fifth of the paths on the headline frame: 13 per tile on average, thin rounds -- quad
// rounds) on the pool over the BVH4 mirror from the root, adds the unoccluded terms in the reference's order and writes the pixels.
// What it replaces: the shade launch (0.04 ms: as long as the dependent fetches of the paths that hit), the any-hit launch (0.18 ms, its
// last 40 % a drain with 16 of 25 wave slots occupied), two launch gaps, the hit records, the compact path entries and the shadow queue
// (written by one kernel, read back by the next).  The occlusion work of a tile now hides behind the other 23 tile waves of its CU.
// Same arithmetic in the same order per path, so frames are the wavefront form's bit for bit; ray counters from the waves' bins.
// The next launch's tile order is made by k_tile_order (the job that rode in the shade launch's first workgroups).
struct Frame1Args {
    TraceArgs t;               // bvh, steal_min, burst_bulk, tile_group, the tile order tables (ray arrays unused: the kernel makes its rays)
    TailArgs  k;               // ShadeArgs (hit groups, frame constants, result, resolve), RaygenJob, epoch
};

// DEFER (MIPT_FRAME1 = 2): the tile wave traces and SHADES its paths, and leaves the occlusion rays to the any-hit launch of the wavefront form
// (shadow queue, one append per tile) -- what is saved is the shade launch, the hit records and the compact path entries.  Bit-identical and
// SLOWER too: trace + shade 0.436 ms in the tile waves against 0.318 + 0.043 as two launches (frame 0.612 against 0.518).  Not registers (path
// state made again after the traversal, shading arguments read late: 52 bytes of scratch, none in the loop), not the queue counter (see
// below): the shading of a tile is a chain of four or five dependent fetches (hit group, vertices, accumulator, pixel) by ONE wave that holds
// a tile slot meanwhile, in a memory system the traversal keeps saturated -- k_shade runs the same chains 2 M wide at full occupancy.
template <int ROWS, int Q, int QS, bool DEFER>      // Q: node format the closest-hit phases walk (0 fp32 / 1 packed); QS: the occlusion pool's (2: BVH4 mirror)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ROWS <= 24 ? MIPT_TILED_WAVES : 5, ROWS <= 24 ? MIPT_TILED_WAVES : 5)))
void k_frame1(Frame1Args f)
{
    const TraceArgs& a = f.t;
    constexpr int kPoolCap = ROWS * 64 - 384;
    __shared__ __attribute__((aligned(16))) uint32_t stk[ROWS * 64];
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(stk + kPoolCap);
    float* const sh_u = reinterpret_cast<float*>(stk + kPoolCap + 128);
    float* const sh_v = reinterpret_cast<float*>(stk + kPoolCap + 192);
    uint32_t* const sh_done = stk + kPoolCap + 320;
    const int lane = (int)threadIdx.x;
    if (blockIdx.x == 0u) { begin_sample(f.k.rg.ctr, threadIdx.x, f.k.rg.fc.n_paths, f.k.rg.n_valid, f.k.rg.zero_stats); tail_publish_epoch(f.k.rg.ctr, f.k.epoch); }
    const uint32_t n = a.n_max;
    const int steal_floor = 64 - (kPoolCap - 168) / ROWS;
    const int steal_min = a.steal_min < 64 ? (a.steal_min > steal_floor ? a.steal_min : steal_floor) : 64;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.blob_bytes, 0x00020000);
    const uint32_t tiles = (n + 63u) >> 6, tiles_seg = (tiles + kNumSegments - 1) / kNumSegments;
    const uint32_t xcd = blockIdx.x & (kNumSegments - 1), waves_seg = gridDim.x / kNumSegments;
    Trav T;
    T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f;
    uint32_t n_shadow = 0u, n_iter = 0u;
    bool awaited = false;
    for (uint32_t k = blockIdx.x / kNumSegments;; k += waves_seg) {
        uint32_t tile = xcd * tiles_seg + k;
        if (a.tile_group != 0u) {
            const uint32_t gq = k / a.tile_group, r = k - gq * a.tile_group;
            tile = (gq * kNumSegments + xcd) * a.tile_group + r;
        } else if (k >= tiles_seg) break;
        if (tile >= tiles) break;
        if (a.tile_order != nullptr && a.n_order == tiles) {
            const uint32_t t2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tile_order[tile]);
            tile = t2 < tiles ? t2 : tile;
        }
        uint32_t cost = 0u;
        const uint32_t ray = tile * 64u + (uint32_t)lane;
        // ============================================================ raygen (optixPathTracer.cu:249-281): the path stays in the lane
        uint32_t slot = kInvalidSlot, seed = 0u;
        float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
        bool has_ray = false;
        T.prim = 0xffffffffu; T.node = kSentinel; T.sp = 0;
        if (ray < n) {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef const __attribute__((address_space(4))) RaygenJob* JobPtr;
            JobPtr rp = (JobPtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Frame1Args, k) + offsetof(TailArgs, rg));
            asm volatile("" : "+s"(rp));
#else
            const RaygenJob* rp = &f.k.rg;            // (the host pass only parses this)
#endif
            const FrameConsts fc = rp->fc;
            const uint32_t sample = rp->sample;
            float4 ro, rd;
            uint32_t slot0 = kInvalidSlot, seed0 = 0u;
            const bool valid0 = raygen_path<2>(fc, sample, PathQueue{}, rp->rg_seed, rp->result, ray, ro, rd, slot0, seed0);
            if (valid0) { trav_init(T, ro, rd); has_ray = T.node != kSentinel; }      // (launch_frame1 refuses a BVH with xf: shading reads the ray from T)
        }
        // ============================================================ closest hit: one ray per lane, then the wave's entry pool (k_trace_pool)
        while (__popcll(__ballot(!has_ray)) < steal_min) {
            for (int bi = 0; bi < a.burst_bulk; ++bi) {
                if (bi != 0 && __ballot(has_ray && T.node != kSentinel) == 0ull) break;
                uint32_t nv_ = 0u, nt_ = 0u;
                stream_step<0, false, Q>(a, blob, stk, lane, has_ray, T, 0, false, nv_, nt_ PROBE_ARG);
                ++cost;
            }
            if (has_ray && T.node == kSentinel) has_ray = false;           // its record stays in T
        }
        if (__ballot(has_ray) != 0ull) {
            int psize = 0;
            for (int e = 0; e < ROWS; ++e) {
                const bool in = has_ray && T.sp > e;
                const unsigned long long m = __ballot(in);
                if (m == 0ull) break;
                uint32_t v = 0u;
                if (in) v = stk[e * 64 + lane];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (in) stk[psize + (int)lane_rank(m)] = pool_pack((int)v, (uint32_t)lane);
                psize += __popcll(m);
            }
            {
                const unsigned long long m = __ballot(has_ray);
                if (has_ray) stk[psize + (int)lane_rank(m)] = pool_pack(T.node, (uint32_t)lane);
                psize += __popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            sh_best[lane] = ((unsigned long long)ord_bits(T.tbest + 0.0f) << 32) | (unsigned long long)T.prim;
            sh_u[lane] = T.u; sh_v[lane] = T.v; sh_done[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            while (psize > 0) {
                pool_round<0, kPoolCap, Q>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, T, psize, false PROBE_ARG);
                ++cost;
            }
            if (has_ray) {
                const unsigned long long key = sh_best[lane];
                T.tbest = ord_float((uint32_t)(key >> 32)); T.prim = (uint32_t)key; T.u = sh_u[lane]; T.v = sh_v[lane];
            }
        }
        // ============================================================ shade (optixPathTracer.cu:323-413, shade_path) and the occlusion rays
        // (the path's slot, seed and accumulator are made AGAIN here rather than held in registers across the traversal: the tile kernel
        //  has none to spare -- 80 VGPRs at six waves per SIMD -- and held values went to scratch inside the loop)
        if (ray < n) {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef const __attribute__((address_space(4))) RaygenJob* JobPtr2;
            JobPtr2 rp = (JobPtr2)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Frame1Args, k) + offsetof(TailArgs, rg));
            asm volatile("" : "+s"(rp));
#else
            const RaygenJob* rp = &f.k.rg;
#endif
            const FrameConsts fc = rp->fc;
            const uint32_t sample = rp->sample;
            float4 ro, rd;
            slot = kInvalidSlot;
            const bool valid2 = raygen_path<2>(fc, sample, PathQueue{}, rp->rg_seed, rp->result, ray, ro, rd, slot, seed);
            res = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid2 && !(sample == 0u || fc.batch_spp != 0u)) res = rp->result[slot];
        }
        const bool valid = slot != kInvalidSlot;
        // (the shading arguments are read from the kernel-argument segment HERE: held from the kernel's entry they cost the traversal loop its
        //  scalar registers)
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const __attribute__((address_space(4))) ShadeArgs* ShadePtr;
        ShadePtr sap = (ShadePtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Frame1Args, k) + offsetof(TailArgs, a));
        asm volatile("" : "+s"(sap));
        const ShadeArgs sh_a = *sap;
#else
        const ShadeArgs& sh_a = f.k.a;
#endif
        const float4 hit = T.prim != 0xffffffffu ? make_float4(T.tbest, u2f(T.prim), T.u, T.v) : make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
        ShadeOut o;
        o.push_next = false; o.push_shadow = false;
        if (valid) {
            n_iter = 1u;
            shade_path<true>(sh_a, hit, make_float4(1.0f, 1.0f, 1.0f, u2f(seed)), slot, 0u, res, o,
                             [&](float4& o4, float4& d4) { o4 = make_float4(T.o.x, T.o.y, T.o.z, 0.01f); d4 = make_float4(T.d.x, T.d.y, T.d.z, 1e16f); });
        }
        const bool sh = valid && o.push_shadow;
        if (DEFER) {
            // as k_shade: the path ends here; with an occlusion ray pending the any-hit launch finishes the pixel, else it is finished now
            const unsigned long long ms = __ballot(sh);
            uint32_t base = 0u;
            if (ms != 0ull) {
                // (the first workgroup's begin_sample clears the queue counter: no append before its epoch is seen)
                if (!awaited) { tail_await_epoch(f.k.rg.ctr, f.k.epoch); awaited = true; }
                // (6.5 k appends on one counter in 0.3 ms: not what this form waits for -- a probe build with one counter per XCD took the same 0.44 ms)
                if (lane == 0) base = atomicAdd(&sh_a.ctr->n_shadow, (uint32_t)__popcll(ms));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            }
            if (valid) {
                if (sh_a.resolve.on) {
                    if (sh) { o.sc.w = u2f(f2u(o.sc.w) | 0x80000000u); sh_a.result[slot] = res; }
                    else resolve_pixel(sh_a.resolve, slot, res);
                } else sh_a.result[slot] = res;
                if (sh) {
                    const uint32_t js = base + lane_rank(ms);
                    sh_a.shadow.ray_o[js] = o.sro; sh_a.shadow.ray_d[js] = o.srd; sh_a.shadow.contrib[js] = o.sc;
                }
            }
            if (lane == 0 && a.tile_cost != nullptr && a.n_order == tiles) record_tile_cost(a, tile, cost);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            continue;
        }
        if (__ballot(sh) != 0ull) {
            Trav S;
            trav_init(S, sh ? o.sro : make_float4(0.f, 0.f, 0.f, 1.0f), sh ? o.srd : make_float4(0.f, 0.f, 1.f, 0.0f), a.bvh.xf);
            const uint2 work = tail_trace<2, QS, kPoolCap>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, S, sh && S.node != kSentinel);
            cost += work.x & 0xffffu;
            if (sh) {
                ++n_shadow;
                if (sh_done[lane] == 0u) { res.x += o.sc.x; res.y += o.sc.y; res.z += o.sc.z; }      // finish_ray<2>
            }
        }
        if (valid) { if (sh_a.resolve.on) resolve_pixel(sh_a.resolve, slot, res); else sh_a.result[slot] = res; }
        if (lane == 0 && a.tile_cost != nullptr && a.n_order == tiles) record_tile_cost(a, tile, cost);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // the next tile's stacks overwrite the merge slots
    }
    if (!DEFER || !awaited) tail_await_epoch(f.k.rg.ctr, f.k.epoch);
    if (!DEFER) tail_stats(f.k.rg.ctr, 0u, n_shadow, n_iter);        // (DEFER: the queue bookkeeping of the wavefront form counts the occlusion rays)
}

// the next tiled launch's order from this one's cost classes (TileOrderJob): the work of k_shade's first workgroups, for passes without a shade launch
__global__ __launch_bounds__(kShadeBlock) void k_tile_order(TileOrderJob j)
{
    tile_order_job(j, blockIdx.x);
}

// ---- k_tail_pool_inst: k_tail_pool over a two-level (instance) structure (round 3) ------------------------------------------------------
// The same loop -- closest hit, shade_path, the occlusion ray, the next bounce -- with the wave's rays traversed on the two-level entry pool
// (inst_pool.h: 64-bit entries tagged with the instance they belong to).  A path-traced launch over an instance structure used to run
// wavefront bounces to the end: five nearly empty kernels per bounce for the last few hundred paths of a Russian-roulette pass.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tail_pool_inst(TailArgs k)
{
    const ShadeArgs& a = k.a;
    const uint32_t ppw = k.ppw;
    const bool use_rg = k.rg.ctr != nullptr;
    if (use_rg && blockIdx.x == 0u) { begin_sample(k.rg.ctr, threadIdx.x, k.rg.fc.n_paths, k.rg.n_valid, k.rg.zero_stats); tail_publish_epoch(k.rg.ctr, k.epoch); }
    constexpr int kRows = 24, kCap = (kRows * 64 - 384) / 2;
    __shared__ __attribute__((aligned(16))) uint2 stk2[kRows * 32];
    uint32_t* const words = reinterpret_cast<uint32_t*>(stk2);
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(words + 2 * kCap);
    float* const sh_u = reinterpret_cast<float*>(words + 2 * kCap + 128);
    float* const sh_v = reinterpret_cast<float*>(words + 2 * kCap + 192);
    uint32_t* const sh_done = words + 2 * kCap + 320;
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.inst_blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (!use_rg) { const uint32_t m = a.ctr->n_cur; n = m < n ? m : n; }
    const uint32_t i = blockIdx.x * ppw + threadIdx.x;
    uint32_t n_closest = 0, n_shadow = 0, n_iter = 0;
    uint32_t slot = kInvalidSlot, depth = 0;
    float4 ro4 = make_float4(0.f, 0.f, 0.f, 0.f), rd4 = ro4, sa = ro4, res = ro4;
    if (i < n && threadIdx.x < ppw) tail_load_path(k, use_rg, i, slot, depth, ro4, rd4, sa, res);
    bool active = slot != kInvalidSlot;
#if MIPT_INST_STATS
    uint32_t ist[6] = { 0u, 0u, 0u, 0u, 0u, 0u };
#endif
    while (__ballot(active) != 0ull) {
        Trav T;
        trav_init(T, ro4, rd4, a.bvh.xf);
        tail_trace_inst<0, kCap>(blob, a.bvh, stk2, sh_best, sh_u, sh_v, sh_done, lane, T, active && T.node != kSentinel, false INST_STATS_ARG);
        const unsigned long long key = sh_best[lane];
        const uint32_t prim = (uint32_t)key;
        const float4 hit = prim != 0xffffffffu ? make_float4(ord_float((uint32_t)(key >> 32)), u2f(prim), sh_u[lane], sh_v[lane])
                                               : make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
        ShadeOut o;
        o.push_next = false; o.push_shadow = false;
        if (active) {
            ++n_iter;
            shade_path(a, hit, sa, slot, depth, res, o, [&](float4& o4, float4& d4) { o4 = ro4; d4 = rd4; });
        }
        const bool sh = active && o.push_shadow;
        if (__ballot(sh) != 0ull) {
            Trav S;
            trav_init(S, sh ? o.sro : ro4, sh ? o.srd : rd4, a.bvh.xf);
            tail_trace_inst<2, kCap>(blob, a.bvh, stk2, sh_best, sh_u, sh_v, sh_done, lane, S, sh && S.node != kSentinel, false INST_STATS_ARG);
            if (sh) {
                ++n_shadow;
                if (sh_done[lane] == 0u) { res.x += o.sc.x; res.y += o.sc.y; res.z += o.sc.z; }      // finish_ray<2>
            }
        }
        active = active && o.push_next;
        if (active) { ++n_closest; ro4 = o.nro; rd4 = o.nrd; sa = o.nsa; depth = o.nsb.y; }
    }
    if (slot != kInvalidSlot) { if (a.resolve.on) resolve_pixel(a.resolve, slot, res); else a.result[slot] = res; }
    if (use_rg) tail_await_epoch(a.ctr, k.epoch);
    tail_stats(a.ctr, n_closest, n_shadow, n_iter);
}

// ---- k_tail_pool2: the occlusion ray of bounce b and the closest-hit ray of bounce b + 1 of every path in ONE pool pass ------
// The tail of a pass is bound by the chain of dependent fetches per bounce (closest hit, then shading, then the occlusion ray);
// the two rays are independent once bounce b is shaded, so tracing them together halves the chain.  Entries carry a 7-bit slot
// (bit 6: the lane's occlusion ray, kept in LDS; else its closest-hit ray, in the lane's registers): bit 31 leaf | 24-bit
// payload | 7-bit slot, i.e. < 2^24 nodes and < 2^22 triangles (DeviceBvh::pool2_ok).  The accumulator still receives its terms
// in the reference's order: the occlusion result of bounce b is added after the pass, before bounce b + 1 is shaded.
__device__ __forceinline__ uint32_t pool_pack2(int ref, uint32_t slot7)
{
    if (ref >= 0) return ((uint32_t)ref << 7) | slot7;
    const uint32_t desc = ~(uint32_t)ref;
    return 0x80000000u | ((desc >> 4) << 9) | (((desc - 1u) & 3u) << 7) | slot7;
}

template <int CAP, int Q>
__device__ __forceinline__ void pool_round2(const __amdgpu_buffer_rsrc_t blob, const DeviceBvh& bvh, uint32_t* stk,
                                            unsigned long long* sh_best, float* sh_u, float* sh_v, uint32_t* sh_done, const float* sray,
                                            const int lane, const Trav& T, int& psize)
{
    if (Q == 2 && MIPT_QUAD_MAX > 0) {
        if (psize <= MIPT_QUAD_MAX) { pool_round_quad<3, 7>(blob, bvh, stk, sh_best, sh_u, sh_v, sh_done, sray, lane, T, psize, false); return; }
    }
    int k = psize < 64 ? psize : 64;
    const int room = CAP - (Q == 2 ? 120 : 104) - psize;
    if (Q == 2) { if (3 * k > room) k = room > 3 ? room / 3 : 1; }
    else        { if (k > room) k = room > 1 ? room : 1; }
    const bool have = lane < k;
    uint32_t ent = 0u;
    if (have) ent = stk[psize - 1 - lane];
    psize -= k;
    const uint32_t slot7 = ent & 127u, ls = slot7 & 63u;
    const bool is_s = have && (slot7 & 64u) != 0u;
    float ox = shfl_f(T.o.x, ls), oy = shfl_f(T.o.y, ls), oz = shfl_f(T.o.z, ls);
    float dx = shfl_f(T.d.x, ls), dy = shfl_f(T.d.y, ls), dz = shfl_f(T.d.z, ls);
    float ix = shfl_f(T.idir.x, ls), iy = shfl_f(T.idir.y, ls), iz = shfl_f(T.idir.z, ls);
    float nx = shfl_f(T.nood.x, ls), ny = shfl_f(T.nood.y, ls), nz = shfl_f(T.nood.z, ls);
    float tmn = shfl_f(T.tmin, ls), tmx = shfl_f(T.tmax, ls);
    if (__ballot(is_s) != 0ull) {
        if (is_s) {
            ox = sray[0 * 64 + ls]; oy = sray[1 * 64 + ls]; oz = sray[2 * 64 + ls];
            dx = sray[3 * 64 + ls]; dy = sray[4 * 64 + ls]; dz = sray[5 * 64 + ls];
            ix = sray[6 * 64 + ls]; iy = sray[7 * 64 + ls]; iz = sray[8 * 64 + ls];
            nx = sray[9 * 64 + ls]; ny = sray[10 * 64 + ls]; nz = sray[11 * 64 + ls];
            tmn = sray[12 * 64 + ls]; tmx = sray[13 * 64 + ls];
        }
    }
    float tcur = tmx;
    bool live = have;
    if (have) {
        if (is_s) { if (sh_done[ls] != 0u) live = false; }
        else      tcur = ord_float((uint32_t)(sh_best[ls] >> 32));
    }
    const bool leaf = (ent >> 31) != 0u;
    const uint32_t payload = (ent >> 7) & 0xffffffu;
    const uint32_t lfirst = payload >> 2, lcnt = (payload & 3u) + 1u;
    const bool do_inner = live && !leaf;
    const bool do_tri   = live && leaf;
    u32x4 q0, q1, q2, q3;
#if MIPT_PROBE_STAMP
    unsigned long long pw_wait = 0ull;
#endif
    load_step<Q>(blob, bvh, do_inner, do_tri, payload, lfirst, q0, q1, q2, q3 PROBE_ARG);
    const float4 n0 = make_float4(u2f(q0.x), u2f(q0.y), u2f(q0.z), u2f(q0.w)), n1 = make_float4(u2f(q1.x), u2f(q1.y), u2f(q1.z), u2f(q1.w)),
                 n2 = make_float4(u2f(q2.x), u2f(q2.y), u2f(q2.z), u2f(q2.w));
    uint32_t push0 = 0u, push1 = 0u, push2 = 0u, push3 = 0u;
    int npush = 0;
    if (Q == 2 && do_inner) {
        const Slab4 b = slab_children4(q0, q1, q2, q3, ix, iy, iz, nx, ny, nz, tmn, tcur);
        npush = b.n;
        if (b.n > 0) push1 = pool_pack2(b.ref[0], slot7);
        if (b.n > 1) push0 = pool_pack2(b.ref[1], slot7);
        if (b.n > 2) push2 = pool_pack2(b.ref[2], slot7);
        if (b.n > 3) push3 = pool_pack2(b.ref[3], slot7);
    }
    if (Q != 2 && do_inner) {
        const SlabOut b = slab_children<Q>(q0, q1, q2, q3, ix, iy, iz, nx, ny, nz, tmn, tcur);
        const bool h0 = b.tn0 <= b.tf0, h1 = b.tn1 <= b.tf1;
        const int c0 = b.c0, c1 = b.c1;
        if (h0 && h1) {
            const bool swp = b.tn1 < b.tn0;
            push0 = pool_pack2(swp ? c0 : c1, slot7);
            push1 = pool_pack2(swp ? c1 : c0, slot7);
            npush = 2;
        } else if (h0 || h1) {
            push1 = pool_pack2(h0 ? c0 : c1, slot7);
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
        const bool ok = det != 0.0f && u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > tmn && t < tmx;
        bool stop = false;
        if (ok) {
            if (is_s) { sh_done[ls] = 1u; stop = true; }
            else {
                const unsigned long long key = ((unsigned long long)ord_bits(t + 0.0f) << 32) | (unsigned long long)prim;
                atomicMin(&sh_best[ls], key);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (*(volatile unsigned long long*)&sh_best[ls] == key) { sh_u[ls] = u; sh_v[ls] = v; }
            }
        }
        if (!stop && lcnt > 1u) {
            push1 = 0x80000000u | ((lfirst + 1u) << 9) | ((lcnt - 2u) << 7) | slot7;
            npush = 1;
        }
    }
    {
        if (Q == 2) {
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

// ---- k_tail_pool2_inst: k_tail_pool2's two rays per pass over a two-level structure (round 4) --------------------------------------------
// The occlusion ray of bounce b and the closest-hit ray of bounce b + 1 of every path in ONE pass of the two-level pool (inst_pool.h,
// pool_round_inst<.., TWO>: bit 31 of an entry's instance word names the slot's occlusion ray, which waits in LDS): the chain of
// dependent pool rounds per bounce is the longer of the two traversals instead of their sum.  The accumulator receives its terms in
// the reference's order, as in k_tail_pool2 (the light term of bounce b is added after the pass, before bounce b + 1 is shaded).  No refill.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tail_pool2_inst(TailArgs k)
{
    const ShadeArgs& a = k.a;
    const uint32_t ppw = k.ppw;
    const bool use_rg = k.rg.ctr != nullptr;
    if (use_rg && blockIdx.x == 0u) { begin_sample(k.rg.ctr, threadIdx.x, k.rg.fc.n_paths, k.rg.n_valid, k.rg.zero_stats); tail_publish_epoch(k.rg.ctr, k.epoch); }
    constexpr int kRows = 24, kCap = (kRows * 64 - 384) / 2;
    __shared__ __attribute__((aligned(16))) uint2 stk2[kRows * 32 + 7 * 64];          // + the occlusion rays: 14 words per slot
    uint32_t* const words = reinterpret_cast<uint32_t*>(stk2);
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(words + 2 * kCap);
    float* const sh_u = reinterpret_cast<float*>(words + 2 * kCap + 128);
    float* const sh_v = reinterpret_cast<float*>(words + 2 * kCap + 192);
    uint32_t* const sh_done = words + 2 * kCap + 320;
    float* const sray = reinterpret_cast<float*>(words + kRows * 64);
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.inst_blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (!use_rg) { const uint32_t m = a.ctr->n_cur; n = m < n ? m : n; }
    const uint32_t i = blockIdx.x * ppw + threadIdx.x;
    uint32_t n_closest = 0, n_shadow = 0, n_iter = 0;
    uint32_t slot = kInvalidSlot, depth = 0;
    float4 ro4 = make_float4(0.f, 0.f, 0.f, 0.f), rd4 = ro4, sa = ro4, res = ro4;
    if (i < n && threadIdx.x < ppw) tail_load_path(k, use_rg, i, slot, depth, ro4, rd4, sa, res);
    bool active = slot != kInvalidSlot;
    bool has_s = false;                          // an occlusion ray of the previous bounce is waiting (in LDS)
    float cx = 0.f, cy = 0.f, cz = 0.f;          // what it adds when nothing is in its way
    while (__ballot(active || has_s) != 0ull) {
        Trav T;
        T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f;
        T.o = mk3(0.f, 0.f, 0.f); T.d = T.o; T.idir = T.o; T.nood = T.o; T.tmin = 0.0f; T.tmax = 0.0f;
        if (active) trav_init(T, ro4, rd4, a.bvh.xf);
        const bool has_c = active && T.node != kSentinel;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        sh_best[lane] = ((unsigned long long)ord_bits(T.tbest + 0.0f) << 32) | 0xffffffffull;
        sh_u[lane] = 0.0f; sh_v[lane] = 0.0f; sh_done[lane] = 0u;
        const unsigned long long mc = __ballot(has_c), ms = __ballot(has_s);
        if (has_c) stk2[(int)lane_rank(mc)] = make_uint2(pool_pack(0, (uint32_t)lane), 0u);
        int psize = __popcll(mc);
        if (has_s) stk2[psize + (int)lane_rank(ms)] = make_uint2(pool_pack(0, (uint32_t)lane), 0x80000000u);
        psize += __popcll(ms);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        while (psize > 0) pool_round_inst<0, kCap, false, 0, true>(blob, a.bvh, stk2, sh_best, sh_u, sh_v, sh_done, lane, T, psize, false, nullptr, sray);
        // the previous bounce's light term first: the accumulator gets its terms in the reference's order
        if (has_s) {
            if (sh_done[lane] == 0u) { res.x += cx; res.y += cy; res.z += cz; }      // finish_ray<2>
            has_s = false;
        }
        if (active) {
            const unsigned long long key = sh_best[lane];
            const uint32_t prim = (uint32_t)key;
            const float4 hit = prim != 0xffffffffu ? make_float4(ord_float((uint32_t)(key >> 32)), u2f(prim), sh_u[lane], sh_v[lane])
                                                   : make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
            ShadeOut o;
            o.push_next = false; o.push_shadow = false;
            ++n_iter;
            shade_path(a, hit, sa, slot, depth, res, o, [&](float4& o4, float4& d4) { o4 = ro4; d4 = rd4; });
            if (o.push_shadow) {
                ++n_shadow;
                Trav S;
                trav_init(S, o.sro, o.srd, a.bvh.xf);
                if (S.node != kSentinel) {
                    sray[0 * 64 + lane] = S.o.x;    sray[1 * 64 + lane] = S.o.y;     sray[2 * 64 + lane] = S.o.z;
                    sray[3 * 64 + lane] = S.d.x;    sray[4 * 64 + lane] = S.d.y;     sray[5 * 64 + lane] = S.d.z;
                    sray[6 * 64 + lane] = S.idir.x; sray[7 * 64 + lane] = S.idir.y;  sray[8 * 64 + lane] = S.idir.z;
                    sray[9 * 64 + lane] = S.nood.x; sray[10 * 64 + lane] = S.nood.y; sray[11 * 64 + lane] = S.nood.z;
                    sray[12 * 64 + lane] = S.tmin;  sray[13 * 64 + lane] = S.tmax;
                    has_s = true; cx = o.sc.x; cy = o.sc.y; cz = o.sc.z;
                } else {
                    res.x += o.sc.x; res.y += o.sc.y; res.z += o.sc.z;              // an empty interval is not occluded
                }
            }
            active = o.push_next;
            if (active) { ++n_closest; ro4 = o.nro; rd4 = o.nrd; sa = o.nsa; depth = o.nsb.y; }
        }
    }
    if (slot != kInvalidSlot) { if (a.resolve.on) resolve_pixel(a.resolve, slot, res); else a.result[slot] = res; }
    if (use_rg) tail_await_epoch(a.ctr, k.epoch);
    tail_stats(a.ctr, n_closest, n_shadow, n_iter);
}

template <int Q>
// `refill` != 0: the launch has fewer waves than ppw-path chunks (one set of resident waves), and a wave whose paths have ended
// takes new ones from the rest of the queue -- the survivors of a Russian-roulette pass thin out geometrically, and a wave that
// walks its last two paths for forty bounces does so at the price of sixty-four.  Path ranges are dealt per XCD segment
// (DeviceCounters::cursor_closest, zero at the start of a bounce): segment s = blockIdx % 8 holds the paths [s, s + 1) * seg_len,
// its waves start on the first chunks and the cursor counts what has been taken beyond them.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tail_pool2(TailArgs k)
{
    const ShadeArgs& a = k.a;
    const uint32_t ppw = k.ppw, refill = k.refill;
    const bool use_rg = k.rg.ctr != nullptr;
    if (use_rg && blockIdx.x == 0u) { begin_sample(k.rg.ctr, threadIdx.x, k.rg.fc.n_paths, k.rg.n_valid, k.rg.zero_stats); tail_publish_epoch(k.rg.ctr, k.epoch); }
    constexpr int kCap = 1152;
    static_assert(kCap % 2 == 0, "sh_best holds 64-bit keys");
    __shared__ __attribute__((aligned(16))) uint32_t lds[kCap + 128 + 3 * 64 + 14 * 64];
    uint32_t* const stk = lds;
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(lds + kCap);
    float* const sh_u = reinterpret_cast<float*>(lds + kCap + 128);
    float* const sh_v = reinterpret_cast<float*>(lds + kCap + 192);
    uint32_t* const sh_done = lds + kCap + 256;
    float* const sray = reinterpret_cast<float*>(lds + kCap + 320);
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (!use_rg) { const uint32_t m = a.ctr->n_cur; n = m < n ? m : n; }
    // without refill: wave w owns the paths [w, w + 1) * ppw.  With: the segment's range and this wave's first chunk in it
    const uint32_t seg = blockIdx.x & (kNumSegments - 1), waves_seg = gridDim.x / kNumSegments;
    const uint32_t seg_len = refill ? (n + kNumSegments - 1) / kNumSegments : 0u;
    const uint32_t seg_lo = seg * seg_len, seg_hi = seg_lo + seg_len < n ? seg_lo + seg_len : n;
    uint32_t i = refill ? seg_lo + (blockIdx.x / kNumSegments) * ppw + threadIdx.x : blockIdx.x * ppw + threadIdx.x;
    const uint32_t i_end = refill ? seg_hi : n;
    uint32_t n_closest = 0, n_shadow = 0, n_iter = 0, n_iter_max = 0;
    uint32_t slot = kInvalidSlot, depth = 0;
    float4 ro4 = make_float4(0.f, 0.f, 0.f, 0.f), rd4 = ro4, sa = ro4, res = ro4;
    if (i < i_end && threadIdx.x < ppw) tail_load_path(k, use_rg, i, slot, depth, ro4, rd4, sa, res);
    bool active = slot != kInvalidSlot;
    bool has_s = false;                          // an occlusion ray of the previous bounce is waiting (in LDS)
    float cx = 0.f, cy = 0.f, cz = 0.f;          // what it adds when nothing is in its way
    uint32_t fslot = kInvalidSlot;               // refill: the waiting occlusion ray is the last term of the ENDED path of this slot
                                                 // (its accumulator is parked in result[fslot]; the lane may carry a new path already)
    bool dry = refill == 0u;                     // nothing left to take
    bool synced = false;                         // (own raygen) the first workgroup's bookkeeping has been seen
    uint32_t try_seg = seg;
    while (__ballot(active || has_s) != 0ull || !dry) {
        // ---- refill: lanes below ppw without a path take the next paths of the queue
        if (!dry) {
            const bool want = !active && (uint32_t)lane < ppw;
            const unsigned long long mw = __ballot(want);
            const uint32_t kfree = (uint32_t)__popcll(mw);
            if (kfree >= refill) {
                if (use_rg && !synced) { tail_await_epoch(a.ctr, k.epoch); synced = true; }      // the cursors are cleared by the first workgroup                   // enough of the wave is free: one cursor atomic for all of them
                uint32_t base = 0u, lo = 0u, hi = 0u;
                for (;;) {
                    lo = try_seg * seg_len + waves_seg * ppw; hi = (try_seg + 1u) * seg_len < n ? (try_seg + 1u) * seg_len : n;
                    if (lane == 0) base = atomicAdd(&a.ctr->cursor_closest[try_seg * kCursorStride], kfree);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (lo + base < hi) break;                                   // at least one path for us
                    try_seg = (try_seg + 1u) & (kNumSegments - 1);               // this segment is used up: the next one, round once
                    if (try_seg == seg) { dry = true; break; }
                }
                if (!dry) {
                    const uint32_t j = lo + base + lane_rank(mw);
                    if (want && j < hi) {
                        tail_load_path(k, use_rg, j, slot, depth, ro4, rd4, sa, res);
                        if (slot != kInvalidSlot) { n_iter = 0u; active = true; }
                    }
                }
            }
        }
        Trav T;
        T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f;
        T.o = mk3(0.f, 0.f, 0.f); T.d = T.o; T.idir = T.o; T.nood = T.o; T.tmin = 0.0f; T.tmax = 0.0f;
        if (active) trav_init(T, ro4, rd4, a.bvh.xf);
        const bool has_c = active && T.node != kSentinel;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        sh_best[lane] = ((unsigned long long)ord_bits(T.tbest + 0.0f) << 32) | 0xffffffffull;
        sh_u[lane] = 0.0f; sh_v[lane] = 0.0f; sh_done[lane] = 0u;
        const unsigned long long mc = __ballot(has_c), ms = __ballot(has_s);
        if (has_c) stk[(int)lane_rank(mc)] = pool_pack2(0, (uint32_t)lane);
        int psize = __popcll(mc);
        if (has_s) stk[psize + (int)lane_rank(ms)] = pool_pack2(0, 64u + (uint32_t)lane);
        psize += __popcll(ms);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        while (psize > 0) pool_round2<kCap, Q>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, sray, lane, T, psize);
        // the previous bounce's light term first: the accumulator gets its terms in the reference's order
        if (has_s) {
            if (fslot != kInvalidSlot) {          // ... of a path that has ended: finish its pixel
                float4 r = a.result[fslot];
                if (sh_done[lane] == 0u) { r.x += cx; r.y += cy; r.z += cz; }
                if (a.resolve.on) resolve_pixel(a.resolve, fslot, r); else a.result[fslot] = r;
                fslot = kInvalidSlot;
            } else if (sh_done[lane] == 0u) { res.x += cx; res.y += cy; res.z += cz; }      // finish_ray<2>
            has_s = false;
        }
        if (active) {
            const unsigned long long key = sh_best[lane];
            const uint32_t prim = (uint32_t)key;
            const float4 hit = prim != 0xffffffffu ? make_float4(ord_float((uint32_t)(key >> 32)), u2f(prim), sh_u[lane], sh_v[lane])
                                                   : make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
            ShadeOut o;
            ++n_iter;
            shade_path(a, hit, sa, slot, depth, res, o, [&](float4& o4, float4& d4) { o4 = ro4; d4 = rd4; });
            if (o.push_shadow) {
                ++n_shadow;
                Trav S;
                trav_init(S, o.sro, o.srd, a.bvh.xf);
                if (S.node != kSentinel) {
                    sray[0 * 64 + lane] = S.o.x;    sray[1 * 64 + lane] = S.o.y;     sray[2 * 64 + lane] = S.o.z;
                    sray[3 * 64 + lane] = S.d.x;    sray[4 * 64 + lane] = S.d.y;     sray[5 * 64 + lane] = S.d.z;
                    sray[6 * 64 + lane] = S.idir.x; sray[7 * 64 + lane] = S.idir.y;  sray[8 * 64 + lane] = S.idir.z;
                    sray[9 * 64 + lane] = S.nood.x; sray[10 * 64 + lane] = S.nood.y; sray[11 * 64 + lane] = S.nood.z;
                    sray[12 * 64 + lane] = S.tmin;  sray[13 * 64 + lane] = S.tmax;
                    has_s = true; cx = o.sc.x; cy = o.sc.y; cz = o.sc.z;
                } else {
                    res.x += o.sc.x; res.y += o.sc.y; res.z += o.sc.z;              // an empty interval is not occluded
                }
            }
            active = o.push_next;
            if (active) { ++n_closest; ro4 = o.nro; rd4 = o.nrd; sa = o.nsa; depth = o.nsb.y; }
            else if (refill) {
                // the path has ended and the lane may get another: its pixel is finished now, or -- with its last occlusion ray
                // still to come -- its accumulator waits in result[] for that ray's pass
                n_iter_max = n_iter > n_iter_max ? n_iter : n_iter_max;
                if (has_s) { a.result[slot] = res; fslot = slot; }
                else if (a.resolve.on) resolve_pixel(a.resolve, slot, res);
                else a.result[slot] = res;
                slot = kInvalidSlot;
            }
        }
    }
    if (slot != kInvalidSlot) { if (a.resolve.on) resolve_pixel(a.resolve, slot, res); else a.result[slot] = res; }
    n_iter = n_iter > n_iter_max ? n_iter : n_iter_max;
    if (use_rg && !synced) tail_await_epoch(a.ctr, k.epoch);
    tail_stats(a.ctr, n_closest, n_shadow, n_iter);
}

// Static chunks of `ppw` rays per wave, traversed together on the entry pool from the root (tail_trace): every any-hit launch
// of the default variant and its closest-hit launches with few rays (launch_trace has the sizes and the measurements).
// Occlusion rays start at scattered hit points and stop at the first hit, so the one-ray-per-lane bulk phase of k_trace_pool
// buys nothing for them: 421 k shadow rays of the headline frame 0.265 -> 0.228 ms with 32 rays per wave, 8.3 M occlusion rays
// of the MeshViewer frame at 4K 6.64 -> 6.07 ms with 64 (big closest-hit launches: 0.512 -> 0.529 ms, they keep the tiled form).
// A launch with few rays lasts as long as ONE wave needs for its rays, so there a wave gets 8 .. 32 and all 64 lanes serve them.
// LDS rows and waves per SIMD of k_trace_static: 20 rows (5 KB: a pool of 896 entries -- pool_round pops fewer when it runs full) and
// <= 72 VGPRs give 28 waves per CU instead of 25 with the 24 rows of the other pool kernels: any-hit kernel of the headline frame
// 0.1937 -> 0.1877 ms, MeshViewer at 4K 5.59 -> 5.51 ms; 22 rows change nothing, 8 waves per SIMD (64 VGPRs, spilling) 0.216 / 0.226 ms
#ifndef MIPT_STATIC_ROWS
#define MIPT_STATIC_ROWS 20
#endif
#ifndef MIPT_STATIC_WAVES
#define MIPT_STATIC_WAVES 7
#endif
#define MIPT_STATIC_ATTR __attribute__((amdgpu_waves_per_eu(MIPT_STATIC_WAVES, MIPT_STATIC_WAVES)))
template <int MODE, int Q>
__global__ __launch_bounds__(64) MIPT_STATIC_ATTR void k_trace_static(TraceArgs a, uint32_t ppw)
{
    constexpr int kPoolCap = MIPT_STATIC_ROWS * 64 - 384;
    static_assert(kPoolCap % 2 == 0, "sh_best holds 64-bit keys");
    __shared__ __attribute__((aligned(16))) uint32_t stk[MIPT_STATIC_ROWS * 64];
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(stk + kPoolCap);
    float* const sh_u = reinterpret_cast<float*>(stk + kPoolCap + 128);
    float* const sh_v = reinterpret_cast<float*>(stk + kPoolCap + 192);
    uint32_t* const sh_done = stk + kPoolCap + 320;
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    const uint32_t chunks = (n + ppw - 1u) / ppw;
#if MIPT_PROBE_WAVES
    const unsigned long long wv_t0 = __builtin_amdgcn_s_memtime(), wv_r0 = __builtin_amdgcn_s_memrealtime();
    uint2 wv_work = make_uint2(0u, 0u); uint32_t wv_rays = 0u;
#endif
    // consecutive chunks go to different XCDs: occlusion rays of neighbouring image regions cost very different amounts, and
    // a contiguous band of the queue per XCD (as the tiled closest-hit kernel has it) leaves XCDs idle (MeshViewer 4K: 6.73 vs 6.07 ms)
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint32_t ray = chunk * ppw + (uint32_t)lane;
        const bool mine = (uint32_t)lane < ppw && ray < n;
        Trav T;
        T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f;
        T.o = mk3(0.f, 0.f, 0.f); T.d = T.o; T.idir = T.o; T.nood = T.o; T.tmin = 0.0f; T.tmax = 0.0f;
        if (mine) trav_init(T, a.ray_o[(size_t)ray * a.stride], a.ray_d[(size_t)ray * a.stride], a.bvh.xf);
        const uint2 work = tail_trace<MODE, Q, kPoolCap>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, T, mine && T.node != kSentinel,
                                               MODE == 0 && (a.ray_flags & 16u) != 0u);
        (void)work;
#if MIPT_PROBE_WAVES
        wv_work.x += work.x; wv_work.y += work.y; wv_rays += (uint32_t)__popcll(__ballot(mine));
#endif
        if (mine) {
            if (MODE == 0) {
                const unsigned long long key = sh_best[lane];
                T.tbest = ord_float((uint32_t)(key >> 32)); T.prim = (uint32_t)key; T.u = sh_u[lane]; T.v = sh_v[lane];
            }
            finish_ray<MODE>(a, T, ray, MODE == 0 ? T.prim != 0xffffffffu : sh_done[lane] != 0u);
        }
    }
#if MIPT_PROBE_WAVES
    if (lane == 0 && a.ctr->dbg[1] != 0ull && blockIdx.x < 40000u) {      // env MIPT_PROBE_WAVES_ANY
        uint4* rec = reinterpret_cast<uint4*>(a.ctr->dbg[1]);
        const unsigned long long t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
        rec[blockIdx.x] = make_uint4((uint32_t)(t2 - wv_t0), wv_rays, wv_work.y, wv_work.x);
        rec[40000u + blockIdx.x] = make_uint4((uint32_t)wv_r0, (uint32_t)(wv_r0 >> 32), (uint32_t)r2, (uint32_t)(r2 >> 32));
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// resolve: optixPathTracer.cu:308-319
// ------------------------------------------------------------------------------------------------
// advance_ctr != null: the first workgroup also does the queue bookkeeping of the launch's last bounce (k_advance's work)
__global__ __launch_bounds__(256) void k_resolve(FrameConsts fc, const float4* result, float4* accum, uint32_t* frame, uint32_t n_slots,
                                                 DeviceCounters* advance_ctr)
{
    if (advance_ctr && blockIdx.x == 0u) advance(advance_ctr, threadIdx.x, nullptr);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    float4 r = result[i];
    // batched samples: per-sample sums added in sample order (association differs from the sequential loop,
    // values agree to rounding: within the 1e-5 tolerance, not bit for bit)
    for (uint32_t s = 1; s < fc.batch_spp; ++s) {
        const float4 t = result[(size_t)s * n_slots + i];
        r.x += t.x; r.y += t.y; r.z += t.z;
    }
    const float inv_spp = 1.0f / (float)fc.samples_per_launch;
    f3 c = mk3(r.x, r.y, r.z) * inv_spp;
    if (fc.subframe_index > 0u) {
        const float a = 1.0f / (float)(fc.subframe_index + 1u);
        const float4 p = accum[i];
        c = lerp3(mk3(p.x, p.y, p.z), c, a);
    }
    accum[i] = make_float4(c.x, c.y, c.z, 1.0f);
    if (frame) frame[i] = make_color(c);
}

__global__ __launch_bounds__(256) void k_unpermute(const float4* gathered, uint32_t world, uint32_t width, uint32_t height,
                                                   uint32_t ns, float4* accum, uint32_t* frame)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= world * ns) return;
    const uint32_t rank = i / ns, s = i - rank * ns;
    int px, py;
    shard_pixel(width, world, rank, s, px, py);
    if ((uint32_t)px >= width || (uint32_t)py >= height) return;
    const float4 v = gathered[i];
    const uint32_t p = (uint32_t)py * width + (uint32_t)px;
    accum[p] = v;
    if (frame) frame[p] = make_color(mk3(v.x, v.y, v.z));
}

// experiment MIPT_TOP_CACHE: the first 63 nodes of the packed mirror in breadth-first order, references among them flagged; word
// 8 * kTopNodes + i = the real index of copy i.  One thread (63 nodes: tens of microseconds, once per build / refit).
__global__ void k_build_top(const uint4* __restrict__ n32, uint32_t nn, uint32_t* top)
{
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    uint32_t real[kTopNodes];
    uint32_t head = 0u, tail = 1u;
    real[0] = 0u;
    for (; head < tail; ++head) {
        uint4 c[2] = { n32[2u * real[head]], n32[2u * real[head] + 1u] };
        for (int k = 0; k < 2; ++k) {
            const int ref = (int)c[k].w;
            if (ref >= 0 && (uint32_t)ref < nn && tail < kTopNodes - 1u) { real[tail] = (uint32_t)ref; c[k].w = kTopFlag | tail; ++tail; }
        }
        for (int k = 0; k < 2; ++k) { top[head * 8u + 4u * k] = c[k].x; top[head * 8u + 4u * k + 1u] = c[k].y; top[head * 8u + 4u * k + 2u] = c[k].z; top[head * 8u + 4u * k + 3u] = c[k].w; }
        top[kTopNodes * 8u + head] = real[head];
    }
}

void launch_build_top(hipStream_t st, const void* d_nodes32, uint32_t nn, void* d_top)
{
    hipLaunchKernelGGL(k_build_top, dim3(1), dim3(64), 0, st, reinterpret_cast<const uint4*>(d_nodes32), nn, reinterpret_cast<uint32_t*>(d_top));
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
void launch_begin_sample(hipStream_t st, DeviceCounters* ctr, uint32_t n_paths, uint32_t n_valid)
{
    hipLaunchKernelGGL(k_begin_sample, dim3(1), dim3(64), 0, st, ctr, n_paths, n_valid);
}

void launch_advance(hipStream_t st, DeviceCounters* ctr, uint32_t* host_live)
{
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, st, ctr, host_live);
}

void launch_raygen(hipStream_t st, const FrameConsts& fc, uint32_t sample, PathQueue q, uint32_t* rg_seed, float4* result,
                   DeviceCounters* ctr, uint32_t n_valid, bool zero_stats, uint32_t* zero_words, uint32_t n_zero)
{
    const uint32_t blocks = (fc.n_paths + 255u) / 256u;
    if (n_zero > fc.n_paths) n_zero = 0u;                             // never: 16 words per 32768 paths
    hipLaunchKernelGGL(k_raygen, dim3(blocks), dim3(256), 0, st, fc, sample, q, rg_seed, result, ctr, n_valid, zero_stats ? 1u : 0u,
                       zero_words, n_zero);
}

template <int MODE>
static void launch_simple(hipStream_t st, const TraceArgs& a, const TraceOpts& o)
{
    uint32_t blocks = (a.n_max + 63u) / 64u;
    if (blocks > 16384u) blocks = 16384u;
    if (o.count_visits) hipLaunchKernelGGL((k_trace_simple<MODE, true>), dim3(blocks), dim3(64), 0, st, a);
    else                hipLaunchKernelGGL((k_trace_simple<MODE, false>), dim3(blocks), dim3(64), 0, st, a);
}

bool closest_launch_is_tiled(const DeviceBvh& bvh, uint32_t n_max, const TraceOpts& o)
{
    if (n_max == 0u || bvh.inst) return false;
    if (!(o.persistent == 3 && !o.count_visits && bvh.pool_ok && bvh.tri_byte_off != 0u && bvh.depth <= (uint32_t)kStreamStack)) return false;
    return n_max > (uint32_t)(o.num_cus > 0 ? o.num_cus : 256) * 2560u;       // below: static chunks on the pool (launch_trace)
}

template <int MODE>
static void launch_trace(hipStream_t st, TraceArgs a, const TraceOpts& o, const RaygenJob* rgp = nullptr)
{
    if (a.n_max == 0) return;
    const RaygenJob rg = rgp ? *rgp : RaygenJob{};
    if (a.bvh.inst) { launch_trace_inst(st, a, MODE, o.count_visits != 0, o); return; }       // two-level structure: instances.hip
    if (o.persistent >= 2 && a.bvh.tri_byte_off != 0u && a.bvh.depth <= (uint32_t)kStreamStack) {
        int blocks = ((o.blocks > 0 ? o.blocks : 4096) + kNumSegments - 1) / kNumSegments * kNumSegments;
        const bool pool = o.persistent == 3 && !o.count_visits && a.bvh.pool_ok;
        a.tile_group = o.tile_group > 0 ? (uint32_t)o.tile_group : 0u;
        if (pool) {
            const uint32_t cus = (uint32_t)(o.num_cus > 0 ? o.num_cus : 256);
            // Static chunks of a few rays per wave, on the entry pool from the root (k_trace_static): every any-hit launch,
            // and closest-hit launches of at most 2560 rays per CU (655 k) -- there a wave per 64-ray tile would leave the
            // step at the time one wave needs for its tile.  The host only knows the capacity of a launch; a shadow-ray
            // queue sized for one ray per path typically holds a fifth to a quarter of it.
            //   any hit, rays per wave: 64 from an estimated 2048 rays per CU on, 32 below (421 k rays: 24 / 32 / 40 / 48 / 64
            //   take 0.241 / 0.228 / 0.238 / 0.247 / 0.258 ms), 16 / 8 below 384 / 192 per CU (53 k rays: 32 / 16 / 8 take
            //   0.086 / 0.064 / 0.073 ms; 26 k rays: 0.081 / 0.057 / 0.047 ms)
            //   closest hit: 32 from 1536 rays per CU on (518 k rays: 16 / 32 / 64 take 0.236 / 0.226 / 0.256 ms, the tiled form
            //   0.248; 1.04 M rays: 0.438 / 0.344 / 0.327 against 0.300 tiled), 16 below (259 k: 0.164 / 0.169 / 0.247, the
            //   cursor-fed pool this replaces 0.215; 130 k: 0.115 against 0.149), 8 below 192 per CU (32 k: 0.067 against 0.086)
            const uint32_t n_est = MODE != 0 && a.n_ptr != nullptr ? a.n_max / 4u : a.n_max;     // no device-side count: n_max is exact
            const uint32_t q = n_est / (cus * 16u);
            uint32_t ppw = 0u;
            if (MODE != 0)                    ppw = n_est > cus * 2048u ? 64u : (q >= 24u ? 32u : (q >= 12u ? 16u : 8u));
            else if (!closest_launch_is_tiled(a.bvh, a.n_max, o)) ppw = q >= 96u ? 32u : (q >= 12u ? 16u : 8u);
            if (MODE != 0) { static const int e_ppw = [] { const char* e = std::getenv("MIPT_ANY_PPW"); return e ? std::atoi(e) : 0; }(); if (e_ppw >= 1 && e_ppw <= 64) ppw = (uint32_t)e_ppw; }
            if (ppw != 0u) {
                unsigned long long want = ((unsigned long long)n_est + ppw - 1u) / ppw;
                want = want < cus * 16ull ? cus * 16ull : (want > cus * 256ull ? cus * 256ull : want);
                if (o.blocks_user) want = (unsigned long long)blocks;
                const uint32_t g = (uint32_t)((want + kNumSegments - 1) / kNumSegments * kNumSegments);
                if (a.bvh.node_fmt == 2)      hipLaunchKernelGGL((k_trace_static<MODE, 2>), dim3(g), dim3(64), 0, st, a, ppw);
                else if (a.bvh.node_fmt == 1) hipLaunchKernelGGL((k_trace_static<MODE, 1>), dim3(g), dim3(64), 0, st, a, ppw);
                else                          hipLaunchKernelGGL((k_trace_static<MODE, 0>), dim3(g), dim3(64), 0, st, a, ppw);
            } else if constexpr (MODE == 0) {
                // tiled form: one wave per tile of 64 rays, at most 128 waves per CU (beyond that a wave takes several tiles);
                // the ray count may be smaller on the device (n_ptr), surplus waves leave at once
                if (!o.blocks_user) {
                    unsigned long long want = ((unsigned long long)a.n_max + 63ull) / 64ull;
                    want = want < cus * 16ull ? cus * 16ull : (want > cus * 128ull ? cus * 128ull : want);
                    blocks = (int)((want + kNumSegments - 1) / kNumSegments * kNumSegments);
                }
                // (the tiled kernel keeps the BVH2 when a BVH4 mirror exists too: a lane's stack over the BVH4 can need 32 rows -- 8 KB
                //  per wave, 20 waves per CU instead of 24 -- which gives back what the shorter walk gains: 0.404 against 0.406 ms on
                //  the headline frame; MIPT_TILED_BVH4=1 selects it for experiments when the stack fits)
#define MIPT_LAUNCH_TILED(ROWS, Q) do { \
                    if (rgp) hipLaunchKernelGGL((k_trace_pool<0, ROWS, Q, true>), dim3(blocks), dim3(64), 0, st, a, rg); \
                    else     hipLaunchKernelGGL((k_trace_pool<0, ROWS, Q, false>), dim3(blocks), dim3(64), 0, st, a, rg); } while (0)
                if (a.bvh.node_fmt == 2 && o.tiled_bvh4 && a.bvh.stack4 <= 32u) {
                    if (a.bvh.stack4 <= 24u) MIPT_LAUNCH_TILED(24, 2); else MIPT_LAUNCH_TILED(32, 2);
                } else if (a.bvh.node_fmt >= 1) {
                    if (a.bvh.depth <= (uint32_t)MIPT_TILED_ROWS) MIPT_LAUNCH_TILED(MIPT_TILED_ROWS, 1); else MIPT_LAUNCH_TILED(32, 1);
                } else {
                    if (a.bvh.depth <= 24u) MIPT_LAUNCH_TILED(24, 0); else MIPT_LAUNCH_TILED(32, 0);
                }
#undef MIPT_LAUNCH_TILED
            }
        }
        else if (o.count_visits) hipLaunchKernelGGL((k_trace_stream<MODE, true>), dim3(blocks), dim3(64), 0, st, a);
        else                hipLaunchKernelGGL((k_trace_stream<MODE, false>), dim3(blocks), dim3(64), 0, st, a);
    } else {
        launch_simple<MODE>(st, a, o);
    }
}

void launch_trace_closest(hipStream_t st, const DeviceBvh& bvh, const float4* ray_o, const float4* ray_d, uint32_t stride,
                          const uint32_t* n_ptr, uint32_t n_max, float4* hits, DeviceCounters* ctr, const TraceOpts& o, uint32_t ray_flags,
                          const RaygenJob* rg)
{
    TraceArgs a{};
    a.ray_flags = ray_flags;
    a.bvh = bvh; a.ray_o = ray_o; a.ray_d = ray_d; a.stride = stride; a.n_ptr = n_ptr; a.n_max = n_max;
    a.hits = hits; a.cursors = ctr->cursor_closest; a.ctr = ctr;
    a.refill_min = o.refill_min; a.steal_min = o.steal_min; a.burst_bulk = o.burst_bulk; a.burst_dry = o.burst_dry;
    a.tile_order = o.tile_order; a.tile_cost = o.tile_cost; a.tile_hist = o.tile_hist; a.n_order = o.n_order; a.class_width_dev = o.class_width_dev; a.class_width = o.class_width ? o.class_width : 6u;
    launch_trace<0>(st, a, o, rg && closest_launch_is_tiled(bvh, n_max, o) ? rg : nullptr);
}

void launch_trace_any(hipStream_t st, const DeviceBvh& bvh, const float4* ray_o, const float4* ray_d, uint32_t stride,
                      const uint32_t* n_ptr, uint32_t n_max, uint32_t* occluded, const float4* contrib, float4* result,
                      DeviceCounters* ctr, const TraceOpts& o, const ResolveJob& resolve)
{
    TraceArgs a{};
    a.resolve = resolve;
    a.bvh = bvh; a.ray_o = ray_o; a.ray_d = ray_d; a.stride = stride; a.n_ptr = n_ptr; a.n_max = n_max;
    a.occluded = occluded; a.contrib = contrib; a.result = result; a.cursors = ctr->cursor_shadow; a.ctr = ctr;
    a.refill_min = o.refill_min; a.steal_min = o.steal_min; a.burst_bulk = o.burst_bulk; a.burst_dry = o.burst_dry;
    if (occluded) launch_trace<1>(st, a, o);
    else          launch_trace<2>(st, a, o);
}

void launch_shade(hipStream_t st, const FrameConsts& fc, const DeviceBvh& bvh, const MiptHitGroupData* hitgroups,
                  uint32_t n_hitgroups, PathQueue cur, const float4* hits, PathQueue next, ShadowQueue shadow,
                  float4* result, DeviceCounters* ctr, uint32_t n_max, const void* common_vertices, const TileOrderJob& order, uint32_t primary,
                  const ResolveJob& resolve)
{
    ShadeArgs a{};
    a.order = order; a.primary = primary; a.resolve = resolve;
    a.fc = fc; a.bvh = bvh; a.hitgroups = hitgroups; a.n_hitgroups = n_hitgroups; a.cur = cur; a.hits = hits;
    a.next = next; a.shadow = shadow; a.result = result; a.ctr = ctr; a.n_max = n_max; a.common_vertices = common_vertices;
    // one workgroup per 512 paths, in path order, up to 65536 (beyond that they stride): with 2048 striding workgroups the shadow
    // queue interleaved tiles 2048 apart, and the any-hit launch that follows was 8 us slower on the headline frame (0.556 -> 0.548 ms)
    uint32_t blocks = (n_max + kShadeBlock - 1u) / kShadeBlock;
    if (blocks > 65536u) blocks = 65536u;
    if (blocks == 0u) blocks = 1u;
    blocks += order.n_chunks;
    if (fc.max_depth == 1u) hipLaunchKernelGGL(k_shade<true>, dim3(blocks), dim3(kShadeBlock), 0, st, a);
    else                    hipLaunchKernelGGL(k_shade<false>, dim3(blocks), dim3(kShadeBlock), 0, st, a);
}

void launch_tail(hipStream_t st, const FrameConsts& fc, const DeviceBvh& bvh, const MiptHitGroupData* hitgroups, uint32_t n_hitgroups,
                 PathQueue cur, float4* result, DeviceCounters* ctr, uint32_t n_max, const void* common_vertices, const TraceOpts& o,
                 const ResolveJob& resolve, const RaygenJob* rg, uint32_t epoch)
{
    if (n_max == 0u) return;
    TailArgs k{};
    if (rg) { k.rg = *rg; k.epoch = epoch; }
    ShadeArgs& a = k.a;
    a.resolve = resolve;
    a.fc = fc; a.bvh = bvh; a.hitgroups = hitgroups; a.n_hitgroups = n_hitgroups; a.cur = cur;
    a.result = result; a.ctr = ctr; a.n_max = n_max; a.common_vertices = common_vertices;
    if (bvh.inst) {                        // a two-level structure in its entry-pool form (run_wavefront comes here for no other)
        const uint32_t slots = (uint32_t)(o.num_cus > 0 ? o.num_cus : 256) * 16u;
        uint32_t ppw = ((n_max + slots - 1u) / slots + 15u) & ~15u;
        ppw = ppw < 16u ? 16u : (ppw > 64u ? 64u : ppw);
        k.ppw = ppw;
        // more than one bounce: two rays per pass (k_tail_pool2_inst; env MIPT_TAIL_INST2=0: the one-ray form, A/B)
        static const int e_two = [] { const char* e = std::getenv("MIPT_TAIL_INST2"); return e ? std::atoi(e) : 1; }();
        if (fc.max_depth != 1u && e_two) hipLaunchKernelGGL(k_tail_pool2_inst, dim3((n_max + ppw - 1u) / ppw), dim3(64), 0, st, k);
        else                             hipLaunchKernelGGL(k_tail_pool_inst, dim3((n_max + ppw - 1u) / ppw), dim3(64), 0, st, k);
        return;
    }
    // the pool variant needs a BVH that fits k_trace_pool's packed entries; MIPT_OPT_PERSISTENT != 3 selects the plain loop
    if (o.persistent == 3 && bvh.pool_ok && bvh.tri_byte_off != 0u && bvh.depth <= 32u)
    {
        // paths per wave: 64, except for a pass of few bounces (depth cap <= 8) that is too small to give every wave slot two
        // waves -- there the step is the time ONE wave needs for its paths, and fewer paths per wave (the idle lanes still
        // work in the pool rounds) are faster: a 1/8 shard of the headline frame 0.36 -> 0.28 ms, 65 k paths of depth 4
        // 2.37 -> 1.62 ms (profiles/r01_tail_paths.txt).
        uint32_t ppw = 64u;
        if (fc.max_depth >= 1u && fc.max_depth <= 8u) {
            const uint32_t slots = (uint32_t)(o.num_cus > 0 ? o.num_cus : 256) * 32u;
            ppw = ((n_max + slots - 1u) / slots + 15u) & ~15u;
            ppw = ppw < 16u ? 16u : (ppw > 64u ? 64u : ppw);
        } else {
            // longer paths: as thin as ONE set of resident waves allows (16 per CU) -- a second set would only start when the long
            // tails of the first have ended (65 k paths of unbounded depth, 1 M-triangle scene: 1.67 -> 1.50 ms)
            const uint32_t slots = (uint32_t)(o.num_cus > 0 ? o.num_cus : 256) * 16u;
            ppw = ((n_max + slots - 1u) / slots + 15u) & ~15u;
            ppw = ppw < 16u ? 16u : (ppw > 64u ? 64u : ppw);
        }
        if (fc.max_depth != 1u && bvh.pool2_ok != 0u)
        {                                                                                                    // more than one bounce: two rays per pass
            // more chunks than ONE set of resident waves (16 per CU): that set, each wave refilling its ended paths from the rest
            // of the queue (k_tail_pool2) -- where a pass is long enough to pay for the refill's own latency (a cursor atomic and two
            // dependent loads inside the wave's loop): unbounded depth on the 1 M-triangle scene at 1080p 4.00 -> 3.59 ms (refill at
            // 1/4 / 1/2 / 3/4 / all lanes free: 3.59 / 3.60 / 3.67 / 3.84), but Cornell 1080p spp 16 8.28 -> 8.78 ms, a 100 k-triangle
            // scene 6.03 -> 6.20 ms, the 240 k-triangle hall unchanged -- hence only over big trees (>= 2^18 nodes)
            const int e_refill = o.tail_refill;          // env MIPT_TAIL_REFILL at context creation: 0 never, 1 (default) as described, 2 whenever it can
            const uint32_t resident = (uint32_t)(o.num_cus > 0 ? o.num_cus : 256) * 16u / kNumSegments * kNumSegments;
            uint32_t g = (n_max + ppw - 1u) / ppw, refill = 0u;
            const bool long_paths = (fc.max_depth == 0u || fc.max_depth > 8u) && bvh.tri_byte_off / 64u >= (1u << 18);
            if (g > resident && (e_refill == 2 || (e_refill == 1 && long_paths))) { g = resident; refill = ppw / 4u; }     // refill when a quarter is free
            k.ppw = ppw; k.refill = refill;
            if (bvh.node_fmt == 2)      hipLaunchKernelGGL(k_tail_pool2<2>, dim3(g), dim3(64), 0, st, k);
            else if (bvh.node_fmt == 1) hipLaunchKernelGGL(k_tail_pool2<1>, dim3(g), dim3(64), 0, st, k);
            else                        hipLaunchKernelGGL(k_tail_pool2<0>, dim3(g), dim3(64), 0, st, k);
            return;
        }
        k.ppw = ppw;
        static const int e_t1 = [] { const char* e = std::getenv("MIPT_TAIL1"); return e ? std::atoi(e) : 1; }();      // 0: k_tail_pool for one-bounce passes too (A/B)
        if (fc.max_depth == 1u && e_t1) {
            static const int e_ppw1 = [] { const char* e = std::getenv("MIPT_TAIL1_PPW"); return e ? std::atoi(e) : 0; }();
            if (e_ppw1 >= 8 && e_ppw1 <= 64) k.ppw = ppw = (uint32_t)e_ppw1;
            if (bvh.node_fmt == 2)      hipLaunchKernelGGL(k_tail_pool1<2>, dim3((n_max + ppw - 1u) / ppw), dim3(64), 0, st, k);
            else if (bvh.node_fmt == 1) hipLaunchKernelGGL(k_tail_pool1<1>, dim3((n_max + ppw - 1u) / ppw), dim3(64), 0, st, k);
            else                        hipLaunchKernelGGL(k_tail_pool1<0>, dim3((n_max + ppw - 1u) / ppw), dim3(64), 0, st, k);
            return;
        }
        if (bvh.node_fmt == 2)      hipLaunchKernelGGL(k_tail_pool<2>, dim3((n_max + ppw - 1u) / ppw), dim3(64), 0, st, k);
        else if (bvh.node_fmt == 1) hipLaunchKernelGGL(k_tail_pool<1>, dim3((n_max + ppw - 1u) / ppw), dim3(64), 0, st, k);
        else                        hipLaunchKernelGGL(k_tail_pool<0>, dim3((n_max + ppw - 1u) / ppw), dim3(64), 0, st, k);
    }
    else
        hipLaunchKernelGGL(k_tail, dim3((n_max + 63u) / 64u), dim3(64), 0, st, a);          // (never with rg: tail_launch_makes_rays)
}

// a one-bounce pass whose closest-hit launch would take the tiled form: ONE launch (k_frame1) + the order job
bool launch_frame1(hipStream_t st, const FrameConsts& fc, const DeviceBvh& bvh, const MiptHitGroupData* hitgroups, uint32_t n_hitgroups,
                   float4* result, DeviceCounters* ctr, uint32_t n_paths, const void* common_vertices, const TraceOpts& o,
                   const ResolveJob& resolve, const RaygenJob& rg, uint32_t epoch, const TileOrderJob& order, const ShadowQueue* defer_to)
{
    if (fc.max_depth != 1u || !closest_launch_is_tiled(bvh, n_paths, o) || bvh.node_fmt < 1u || bvh.xf != nullptr) return false;
    if (o.tiled_bvh4) return false;                       // (the BVH4 experiment of the tiled kernel keeps the wavefront form)
    Frame1Args f{};
    TraceArgs& a = f.t;
    a.bvh = bvh; a.n_max = n_paths; a.ctr = ctr;
    a.steal_min = o.steal_min; a.burst_bulk = o.burst_bulk; a.burst_dry = o.burst_dry;
    a.tile_group = o.tile_group > 0 ? (uint32_t)o.tile_group : 0u;
    a.tile_order = o.tile_order; a.tile_cost = o.tile_cost; a.tile_hist = o.tile_hist; a.n_order = o.n_order; a.class_width_dev = o.class_width_dev;
    a.class_width = o.class_width ? o.class_width : 6u;
    f.k.rg = rg; f.k.epoch = epoch;
    ShadeArgs& s = f.k.a;
    s.resolve = resolve; s.fc = fc; s.bvh = bvh; s.hitgroups = hitgroups; s.n_hitgroups = n_hitgroups;
    s.result = result; s.ctr = ctr; s.n_max = n_paths; s.common_vertices = common_vertices;
    if (defer_to != nullptr) s.shadow = *defer_to;
    const uint32_t cus = (uint32_t)(o.num_cus > 0 ? o.num_cus : 256);
    unsigned long long want = ((unsigned long long)n_paths + 63ull) / 64ull;
    want = want < cus * 16ull ? cus * 16ull : (want > cus * 128ull ? cus * 128ull : want);
    const int blocks = o.blocks_user ? ((o.blocks + kNumSegments - 1) / kNumSegments * kNumSegments) : (int)((want + kNumSegments - 1) / kNumSegments * kNumSegments);
    const bool deep = bvh.depth > (uint32_t)MIPT_TILED_ROWS;
    if (defer_to != nullptr) {
        if (!deep) hipLaunchKernelGGL((k_frame1<MIPT_TILED_ROWS, 1, 2, true>), dim3(blocks), dim3(64), 0, st, f);
        else       hipLaunchKernelGGL((k_frame1<32, 1, 2, true>), dim3(blocks), dim3(64), 0, st, f);
    } else if (bvh.node_fmt == 2) {
        if (!deep) hipLaunchKernelGGL((k_frame1<MIPT_TILED_ROWS, 1, 2, false>), dim3(blocks), dim3(64), 0, st, f);
        else       hipLaunchKernelGGL((k_frame1<32, 1, 2, false>), dim3(blocks), dim3(64), 0, st, f);
    } else {
        if (!deep) hipLaunchKernelGGL((k_frame1<MIPT_TILED_ROWS, 1, 1, false>), dim3(blocks), dim3(64), 0, st, f);
        else       hipLaunchKernelGGL((k_frame1<32, 1, 1, false>), dim3(blocks), dim3(64), 0, st, f);
    }
    if (order.n_chunks != 0u) hipLaunchKernelGGL(k_tile_order, dim3(order.n_chunks), dim3(kShadeBlock), 0, st, order);
    return true;
}

bool tail_launch_makes_rays(const DeviceBvh& bvh, const TraceOpts& o)
{
    if (bvh.inst) return o.persistent == 3 && bvh.inst_fast != 0u && o.inst_form != 2;          // k_tail_pool_inst
    return o.persistent == 3 && bvh.pool_ok && bvh.tri_byte_off != 0u && bvh.depth <= 32u;      // the pool forms of launch_tail
}

void launch_resolve(hipStream_t st, const FrameConsts& fc, const float4* result, float4* accum, uint32_t* frame, uint32_t n_slots,
                    DeviceCounters* advance_ctr)
{
    const uint32_t blocks = (n_slots + 255u) / 256u;
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, st, fc, result, accum, frame, n_slots, advance_ctr);
}

void launch_unpermute(hipStream_t st, const float4* gathered, uint32_t world, uint32_t width, uint32_t height,
                      float4* accum, uint32_t* frame)
{
    const uint32_t ns = shard_num_samples(width, height, world);
    const uint32_t total = world * ns;
    hipLaunchKernelGGL(k_unpermute, dim3((total + 255u) / 256u), dim3(256), 0, st, gathered, world, width, height, ns, accum, frame);
}

}  // namespace mipt
