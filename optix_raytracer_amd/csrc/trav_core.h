// trav_core.h -- the pieces of the BVH2 traversal shared by every traversal kernel (kernels.hip, whitted_kernels.hip):
// ray set-up, the per-lane LDS stack, one inner-node step.  Build-defined conventions, identical sequence in
// oracle/pt_oracle.c:
//   hit iff tmin < t < tmax; double sided unless a launch culls back faces; det == 0 rejected; u >= 0, v >= 0,
//   u + v <= 1; closest-hit tie -> lowest original primitive index; near child first, tie -> child 0 first.
#pragma once
#include "kernels.h"
#include "detmath.h"

namespace mipt {

__device__ __forceinline__ float fdot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3    fcross(f3 a, f3 b)
{
    return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
               __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}

constexpr int kSentinel = (int)0x80000000;

struct Trav {
    f3       o, d, idir, nood;   // nood = -(o * idir)
    float    tmin, tmax, tbest;
    float    u, v;
    uint32_t prim;
    int      node, sp;
    uint32_t ahk;                // AH kernels only: how many of the ray's nearest candidates this traversal is to find (1 or kAhLevels)
    uint32_t skip;               // AH kernels only (pool_trace.h): with tmin = the t of a candidate the any-hit program ignored, that
                                 // candidate's primitive -- candidates AT tmin count when their primitive index is greater
};

// rows of a 3x4 matrix times a direction: fmaf(m2, z, fmaf(m1, y, m0 * x)) per row (the instance walk's arithmetic, inst_core.h)
__device__ __forceinline__ f3 xform_vec(const float* m, f3 d)
{
    return mk3(__builtin_fmaf(m[2], d.z, __builtin_fmaf(m[1], d.y, m[0] * d.x)), __builtin_fmaf(m[6], d.z, __builtin_fmaf(m[5], d.y, m[4] * d.x)),
               __builtin_fmaf(m[10], d.z, __builtin_fmaf(m[9], d.y, m[8] * d.x)));
}

// xf != null (DeviceBvh::xf: an instance structure of ONE instance walked as its mesh's own BVH): the ray is taken into the instance's
// object space first -- o' = M o + t, d' = M d, not re-normalised, exactly as the two-level walk does when it enters the instance
// (instances.hip) -- so that t, u, v and the primitive come out as that walk has them; tmin / tmax carry over.
__device__ __forceinline__ void trav_init(Trav& T, float4 ro, float4 rd, const float* __restrict__ xf = nullptr)
{
    T.o = mk3(ro.x, ro.y, ro.z);
    T.d = mk3(rd.x, rd.y, rd.z);
    if (xf != nullptr) {
        const f3 m = xform_vec(xf, T.o);
        T.o = mk3(m.x + xf[3], m.y + xf[7], m.z + xf[11]);
        T.d = xform_vec(xf, T.d);
    }
    T.tmin = ro.w; T.tmax = rd.w; T.tbest = rd.w;
    const float eps = 8.27180613e-25f;   // 2^-80: keeps 1/d finite for axis-parallel rays
    const float dx = __builtin_fabsf(T.d.x) > eps ? T.d.x : __builtin_copysignf(eps, T.d.x);
    const float dy = __builtin_fabsf(T.d.y) > eps ? T.d.y : __builtin_copysignf(eps, T.d.y);
    const float dz = __builtin_fabsf(T.d.z) > eps ? T.d.z : __builtin_copysignf(eps, T.d.z);
    T.idir = mk3(1.0f / dx, 1.0f / dy, 1.0f / dz);
    T.nood = mk3(-(T.o.x * T.idir.x), -(T.o.y * T.idir.y), -(T.o.z * T.idir.z));
    T.u = 0.0f; T.v = 0.0f; T.prim = 0xffffffffu;
    T.node = T.tmax > T.tmin ? 0 : kSentinel;   // empty interval: a miss without touching the BVH
    T.sp = 0;
}

// Per-lane traversal stack in LDS, laid out [entry][lane]: bank = lane, conflict free.  The builder caps the
// tree depth at kStackEntries, so it cannot overflow.
struct Stack {
    uint32_t* lds;       // kStackEntries * 64 words
    int       lane;
    __device__ __forceinline__ void push(int& sp, uint32_t v) const { lds[sp * 64 + lane] = v; ++sp; }
    __device__ __forceinline__ int pop(int& sp) const
    {
        if (sp == 0) return kSentinel;
        --sp;
        return (int)lds[sp * 64 + lane];
    }
};

// one inner-node step: fetch 64 B, slab-test both children, descend into the nearer, push the other
__device__ __forceinline__ void visit_inner(Trav& T, const float4* __restrict__ nodes, const Stack& stk)
{
    const float4* np = nodes + (size_t)(uint32_t)T.node * 4u;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    const float ax0 = __builtin_fmaf(n0.x, T.idir.x, T.nood.x), ax1 = __builtin_fmaf(n0.y, T.idir.x, T.nood.x);
    const float ay0 = __builtin_fmaf(n0.z, T.idir.y, T.nood.y), ay1 = __builtin_fmaf(n0.w, T.idir.y, T.nood.y);
    const float az0 = __builtin_fmaf(n2.x, T.idir.z, T.nood.z), az1 = __builtin_fmaf(n2.y, T.idir.z, T.nood.z);
    const float bx0 = __builtin_fmaf(n1.x, T.idir.x, T.nood.x), bx1 = __builtin_fmaf(n1.y, T.idir.x, T.nood.x);
    const float by0 = __builtin_fmaf(n1.z, T.idir.y, T.nood.y), by1 = __builtin_fmaf(n1.w, T.idir.y, T.nood.y);
    const float bz0 = __builtin_fmaf(n2.z, T.idir.z, T.nood.z), bz1 = __builtin_fmaf(n2.w, T.idir.z, T.nood.z);
    const float tn0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax0, ax1), __builtin_fminf(ay0, ay1)),
                                      __builtin_fmaxf(__builtin_fminf(az0, az1), T.tmin));
    const float tf0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax0, ax1), __builtin_fmaxf(ay0, ay1)),
                                      __builtin_fminf(__builtin_fmaxf(az0, az1), T.tbest));
    const float tn1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(bx0, bx1), __builtin_fminf(by0, by1)),
                                      __builtin_fmaxf(__builtin_fminf(bz0, bz1), T.tmin));
    const float tf1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(bx0, bx1), __builtin_fmaxf(by0, by1)),
                                      __builtin_fminf(__builtin_fmaxf(bz0, bz1), T.tbest));
    const bool h0 = tn0 <= tf0;
    const bool h1 = tn1 <= tf1;
    int c0 = (int)f2u(n3.x), c1 = (int)f2u(n3.y);
    if (!h0 && !h1) {
        T.node = stk.pop(T.sp);
    } else if (h0 && h1) {
        const bool swp = tn1 < tn0;
        const int nearc = swp ? c1 : c0;
        const int farc  = swp ? c0 : c1;
        stk.push(T.sp, (uint32_t)farc);
        T.node = nearc;
    } else {
        T.node = h0 ? c0 : c1;
    }
}

// ---- arguments and result write-back shared by every traversal kernel (kernels.hip, instances.hip) ----
struct TraceArgs {
    DeviceBvh       bvh;
    const float4*   ray_o;
    const float4*   ray_d;
    uint32_t        stride;       // float4 elements between consecutive rays (1 = SoA planes, 2 = interleaved)
    const uint32_t* n_ptr;        // device-side ray count (may be null)
    uint32_t        n_max;
    float4*         hits;         // MODE 0
    uint32_t*       occluded;     // MODE 1
    const float4*   contrib;      // MODE 2
    float4*         result;       // MODE 2
    uint32_t*       cursors;      // kNumSegments cursors, kCursorStride apart
    DeviceCounters* ctr;
    int             refill_min;   // k_trace_stream: idle lanes that trigger a swap-in / prefetch round
    int             steal_min;    // k_trace_stream: idle lanes that trigger a steal round once the ray range is exhausted
    int             burst_bulk;   // k_trace_stream: traversal steps per bookkeeping round while fresh rays remain
    int             burst_dry;    // ... and once the ray range is exhausted
    uint32_t        ray_flags;    // MiptRayFlags of every ray of the launch (closest hit: back-face culling)
    uint32_t        tile_group;   // k_trace_pool, tiled: consecutive tiles dealt to one XCD at a time (0: a contiguous eighth each)
    // k_trace_pool, tiled: the launch walks its tiles in the order `tile_order` gives (slot -> tile: a permutation made from the
    // previous launch's costs by a stable counting sort, heaviest class first, image order inside a class: tile_order_job();
    // null or another tile count: image order) and records every tile's cost class in `tile_cost` and the class histogram of
    // every chunk of kOrderChunk tiles in `tile_hist` (null: not recorded)
    const uint32_t* tile_order;
    uint8_t*        tile_cost;    // one byte per tile: its cost CLASS (0 = heaviest .. kOrderClasses-1; tile_class())
    uint32_t*       tile_hist;    // [chunk][class] counts, zero when the launch starts
    uint32_t        class_width;  // cost units per class ...
    const uint32_t* class_width_dev;   // ... or, when not null, read from here (the shade launch's order job adapts it: TileOrderJob::width)
    ResolveJob      resolve;      // MODE 2: an entry whose slot carries bit 31 is its path's last term -- write the pixel
    uint32_t        n_order;      // tiles the tables are laid out for (they are used only when the launch has exactly that many)
};

// k_resolve (optixPathTracer.cu:308-319) for one pixel of a launch with one sample per pixel: `r` is the finished accumulator
__device__ __forceinline__ void resolve_pixel(const ResolveJob& rj, uint32_t slot, float4 r)
{
    const float inv_spp = 1.0f / 1.0f;
    f3 c = mk3(r.x, r.y, r.z) * inv_spp;
    if (rj.subframe_index > 0u) {
        const float a = 1.0f / (float)(rj.subframe_index + 1u);
        const float4 p = rj.accum[slot];
        c = lerp3(mk3(p.x, p.y, p.z), c, a);
    }
    rj.accum[slot] = make_float4(c.x, c.y, c.z, 1.0f);
    if (rj.frame) rj.frame[slot] = make_color(c);
}

template <int MODE>
__device__ __forceinline__ void finish_ray(const TraceArgs& a, const Trav& T, uint32_t ray, bool hit)
{
    if (MODE == 0) {
        a.hits[ray] = hit ? make_float4(T.tbest, u2f(T.prim), T.u, T.v) : make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
    } else if (MODE == 1) {
        a.occluded[ray] = hit ? 1u : 0u;
    } else {
        if (a.resolve.on) {
            const float4 c = a.contrib[ray];
            const uint32_t sw = f2u(c.w), slot = sw & 0x7fffffffu;
            if ((sw >> 31) != 0u) {                      // the path's last term: its pixel is finished here
                float4 r = a.result[slot];
                if (!hit) { r.x += c.x; r.y += c.y; r.z += c.z; }
                resolve_pixel(a.resolve, slot, r);
            } else if (!hit) {
                float4 r = a.result[slot];
                r.x += c.x; r.y += c.y; r.z += c.z;
                a.result[slot] = r;
            }
        } else if (!hit) {
            // result += radiance * attenuation (optixPathTracer.cu:292); one path per slot is in flight
            const float4 c = a.contrib[ray];
            const uint32_t slot = f2u(c.w);
            float4 r = a.result[slot];
            r.x += c.x; r.y += c.y; r.z += c.z;
            a.result[slot] = r;
        }
    }
}


void launch_trace_inst(hipStream_t st, const TraceArgs& a, int mode, bool count, const struct TraceOpts& o);       // instances.hip

}  // namespace mipt
