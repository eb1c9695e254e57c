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
#include "inst_core.h"

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
