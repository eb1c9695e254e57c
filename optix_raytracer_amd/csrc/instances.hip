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
// One ray per thread, one LDS stack of 64 entries per lane shared by both levels ([entry][lane]: bank = lane).  This is the
// straightforward form (the per-ray visiting order of the oracle, exact visit counters); the entry-pool kernels of kernels.hip
// stay one-level.
#include "kernels.h"
#include "detmath.h"
#include "wave_util.h"
#include "trav_core.h"

namespace mipt {

namespace {

constexpr int kInstStack = 64;          // TLAS depth + deepest mesh BVH, both capped at 32 by the builders

struct Stack2 {
    uint32_t* lds; int lane;
    __device__ __forceinline__ void push(int& sp, int v) const { lds[sp * 64 + lane] = (uint32_t)v; ++sp; }
    __device__ __forceinline__ int pop(int& sp, int base) const
    {
        if (sp == base) return kSentinel;
        --sp;
        return (int)lds[sp * 64 + lane];
    }
};

struct RayO { f3 o, d, idir, nood; };

__device__ __forceinline__ void prepare(RayO& r)
{
    const float eps = 8.27180613e-25f;   // 2^-80, as trav_init
    const float dx = __builtin_fabsf(r.d.x) > eps ? r.d.x : __builtin_copysignf(eps, r.d.x);
    const float dy = __builtin_fabsf(r.d.y) > eps ? r.d.y : __builtin_copysignf(eps, r.d.y);
    const float dz = __builtin_fabsf(r.d.z) > eps ? r.d.z : __builtin_copysignf(eps, r.d.z);
    r.idir = mk3(1.0f / dx, 1.0f / dy, 1.0f / dz);
    r.nood = mk3(-(r.o.x * r.idir.x), -(r.o.y * r.idir.y), -(r.o.z * r.idir.z));
}

__device__ __forceinline__ f3 xform_vec(const float* m, f3 d)
{
    return mk3(__builtin_fmaf(m[2], d.z, __builtin_fmaf(m[1], d.y, m[0] * d.x)), __builtin_fmaf(m[6], d.z, __builtin_fmaf(m[5], d.y, m[4] * d.x)),
               __builtin_fmaf(m[10], d.z, __builtin_fmaf(m[9], d.y, m[8] * d.x)));
}

// slab test of both children of one 64-byte node; returns the next node (near child, far one pushed) or a popped entry
__device__ __forceinline__ int step_inner(const float4* __restrict__ nodes, int node, const RayO& r, float tmin, float tbest, const Stack2& stk,
                                          int& sp, int base)
{
    const float4* np = nodes + (size_t)(uint32_t)node * 4u;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    const float ax0 = __builtin_fmaf(n0.x, r.idir.x, r.nood.x), ax1 = __builtin_fmaf(n0.y, r.idir.x, r.nood.x);
    const float ay0 = __builtin_fmaf(n0.z, r.idir.y, r.nood.y), ay1 = __builtin_fmaf(n0.w, r.idir.y, r.nood.y);
    const float az0 = __builtin_fmaf(n2.x, r.idir.z, r.nood.z), az1 = __builtin_fmaf(n2.y, r.idir.z, r.nood.z);
    const float bx0 = __builtin_fmaf(n1.x, r.idir.x, r.nood.x), bx1 = __builtin_fmaf(n1.y, r.idir.x, r.nood.x);
    const float by0 = __builtin_fmaf(n1.z, r.idir.y, r.nood.y), by1 = __builtin_fmaf(n1.w, r.idir.y, r.nood.y);
    const float bz0 = __builtin_fmaf(n2.z, r.idir.z, r.nood.z), bz1 = __builtin_fmaf(n2.w, r.idir.z, r.nood.z);
    const float tn0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax0, ax1), __builtin_fminf(ay0, ay1)), __builtin_fmaxf(__builtin_fminf(az0, az1), tmin));
    const float tf0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax0, ax1), __builtin_fmaxf(ay0, ay1)), __builtin_fminf(__builtin_fmaxf(az0, az1), tbest));
    const float tn1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(bx0, bx1), __builtin_fminf(by0, by1)), __builtin_fmaxf(__builtin_fminf(bz0, bz1), tmin));
    const float tf1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(bx0, bx1), __builtin_fmaxf(by0, by1)), __builtin_fminf(__builtin_fmaxf(bz0, bz1), tbest));
    const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
    const int c0 = (int)f2u(n3.x), c1 = (int)f2u(n3.y);
    if (!h0 && !h1) return stk.pop(sp, base);
    if (h0 && h1) {
        const bool swp = tn1 < tn0;
        stk.push(sp, swp ? c0 : c1);
        return swp ? c1 : c0;
    }
    return h0 ? c0 : c1;
}

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

template <int MODE>
void launch_mode(hipStream_t st, const TraceArgs& a, bool count)
{
    uint32_t blocks = (a.n_max + 63u) / 64u;
    if (blocks > 16384u) blocks = 16384u;
    if (count) hipLaunchKernelGGL((k_trace_inst<MODE, true>), dim3(blocks), dim3(64), 0, st, a);
    else       hipLaunchKernelGGL((k_trace_inst<MODE, false>), dim3(blocks), dim3(64), 0, st, a);
}

}  // namespace

void launch_trace_inst(hipStream_t st, const TraceArgs& a, int mode, bool count)
{
    if (mode == 0) launch_mode<0>(st, a, count);
    else if (mode == 1) launch_mode<1>(st, a, count);
    else launch_mode<2>(st, a, count);
}

}  // namespace mipt
