// inst_core.h -- the pieces of the two-level (instance) walk shared by instances.hip (plain closest / any hit) and
// whitted_kernels.hip (the same walk with the Whitted any-hit programs at every candidate): the per-lane LDS stack over both
// levels, the object-space ray, the slab step over a 64-byte node.  Arithmetic: instances.hip's header comment.
#pragma once
#include "kernels.h"
#include "detmath.h"
#include "trav_core.h"

namespace mipt {

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

// (xform_vec: trav_core.h)

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

}  // namespace mipt
