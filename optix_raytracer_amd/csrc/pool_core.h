// pool_core.h -- the pieces of the entry-pool traversal shared by kernels.hip (one-level BVHs) and instances.hip (two-level
// structures): 16-byte loads through a buffer descriptor, the slab tests over the node formats, the packed pool entries, the
// order-preserving hit keys.  Moved out of kernels.hip unchanged (round 3).
#pragma once
#include "kernels.h"
#include "detmath.h"
#include "trav_core.h"

namespace mipt {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte load through a buffer descriptor: uniform base in SGPRs + 32-bit lane offset, hardware
// range check (an out-of-range offset reads zeros instead of faulting), never split by the compiler
__device__ __forceinline__ float4 ld16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(u2f(v.x), u2f(v.y), u2f(v.z), u2f(v.w));
}

__device__ __forceinline__ u32x4 ldq(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
}

// ---- the slab test of both children of an inner node, from either node format ------------------------------------------------
// Q = false: the 64-byte fp32 node (MiptBvhNode) in pieces n0 .. n3.  Q = true: the packed 32-byte node (MiptBvhNode32) in
// pieces n0, n1 -- child c: {lo.x | hi.x << 16, lo.y | hi.y << 16, lo.z | hi.z << 16} as fp16, then its reference.  The fp16
// values ARE the values the 64-byte node holds (pack_nodes32 rounds the fp32 boxes outward to them), a half converts to float
// exactly, and the plane enters the same fmaf -- so both formats give the same tn / tf bit for bit (v_fma_mix_f32 reads the
// half operand directly: no conversion instruction).
__device__ __forceinline__ float half_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float half_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
struct SlabOut { float tn0, tf0, tn1, tf1; int c0, c1; };

template <int Q>
__device__ __forceinline__ SlabOut slab_children(const u32x4 n0, const u32x4 n1, const u32x4 n2, const u32x4 n3, const float ix, const float iy,
                                                 const float iz, const float nx, const float ny, const float nz, const float tmn, const float tcur)
{
    float ax0, ax1, ay0, ay1, az0, az1, bx0, bx1, by0, by1, bz0, bz1;
    SlabOut o;
    if (Q == 1) {
        // (halves taken by shift / truncate of each dword: a bit_cast of the dword to a half2 vector miscompiles on ROCm 7.2 --
        // every v_fma_mix_f32 then reads element 0 of the loaded vector, whichever element the source names)
        ax0 = __builtin_fmaf(half_lo(n0.x), ix, nx); ax1 = __builtin_fmaf(half_hi(n0.x), ix, nx);
        ay0 = __builtin_fmaf(half_lo(n0.y), iy, ny); ay1 = __builtin_fmaf(half_hi(n0.y), iy, ny);
        az0 = __builtin_fmaf(half_lo(n0.z), iz, nz); az1 = __builtin_fmaf(half_hi(n0.z), iz, nz);
        bx0 = __builtin_fmaf(half_lo(n1.x), ix, nx); bx1 = __builtin_fmaf(half_hi(n1.x), ix, nx);
        by0 = __builtin_fmaf(half_lo(n1.y), iy, ny); by1 = __builtin_fmaf(half_hi(n1.y), iy, ny);
        bz0 = __builtin_fmaf(half_lo(n1.z), iz, nz); bz1 = __builtin_fmaf(half_hi(n1.z), iz, nz);
        o.c0 = (int)n0.w; o.c1 = (int)n1.w;
    } else {
        ax0 = __builtin_fmaf(u2f(n0.x), ix, nx); ax1 = __builtin_fmaf(u2f(n0.y), ix, nx);
        ay0 = __builtin_fmaf(u2f(n0.z), iy, ny); ay1 = __builtin_fmaf(u2f(n0.w), iy, ny);
        az0 = __builtin_fmaf(u2f(n2.x), iz, nz); az1 = __builtin_fmaf(u2f(n2.y), iz, nz);
        bx0 = __builtin_fmaf(u2f(n1.x), ix, nx); bx1 = __builtin_fmaf(u2f(n1.y), ix, nx);
        by0 = __builtin_fmaf(u2f(n1.z), iy, ny); by1 = __builtin_fmaf(u2f(n1.w), iy, ny);
        bz0 = __builtin_fmaf(u2f(n2.z), iz, nz); bz1 = __builtin_fmaf(u2f(n2.w), iz, nz);
        o.c0 = (int)n3.x; o.c1 = (int)n3.y;
    }
    o.tn0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax0, ax1), __builtin_fminf(ay0, ay1)), __builtin_fmaxf(__builtin_fminf(az0, az1), tmn));
    o.tf0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax0, ax1), __builtin_fmaxf(ay0, ay1)), __builtin_fminf(__builtin_fmaxf(az0, az1), tcur));
    o.tn1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(bx0, bx1), __builtin_fminf(by0, by1)), __builtin_fmaxf(__builtin_fminf(bz0, bz1), tmn));
    o.tf1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(bx0, bx1), __builtin_fmaxf(by0, by1)), __builtin_fminf(__builtin_fmaxf(bz0, bz1), tcur));
    return o;
}

// ---- node format 2: four children per node (BVH4 mirror of the BVH2, mipt_api.cpp Bvh4Builder) ------------------------------------
// 64 bytes = four 16-byte pieces, piece c = child c: {lo.x | hi.x << 16, lo.y | hi.y << 16, lo.z | hi.z << 16} as fp16, then its
// reference (>= 0: BVH4 node, < 0: the BVH2's leaf descriptor, kNoChild: empty slot).  One step tests four boxes and orders the
// hit children by entry distance: 0.63 x the steps of the BVH2 walk per primary ray, 0.58 x per shadow ray (1 M-triangle
// scene).  Hit records cannot depend on the tree (closest: min t, tie -> lowest primitive; any: a boolean), so frames equal
// those of the other formats bit for bit.
constexpr int kNoChild = 0x7fffffff;
struct Slab4 { float t[4]; int ref[4]; int n; };       // the hit children, nearest first; n of them

__device__ __forceinline__ void slab_piece(const u32x4 q, const float ix, const float iy, const float iz, const float nx, const float ny, const float nz,
                                           const float tmn, const float tcur, float& key, int& ref)
{
    const float x0 = __builtin_fmaf(half_lo(q.x), ix, nx), x1 = __builtin_fmaf(half_hi(q.x), ix, nx);
    const float y0 = __builtin_fmaf(half_lo(q.y), iy, ny), y1 = __builtin_fmaf(half_hi(q.y), iy, ny);
    const float z0 = __builtin_fmaf(half_lo(q.z), iz, nz), z1 = __builtin_fmaf(half_hi(q.z), iz, nz);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fmaxf(__builtin_fminf(z0, z1), tmn));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fminf(__builtin_fmaxf(z0, z1), tcur));
    ref = (int)q.w;
    key = (tn <= tf && ref != kNoChild) ? tn : __builtin_inff();
}
__device__ __forceinline__ void cmp_swap(float& ka, int& ra, float& kb, int& rb)
{
    const bool sw = kb < ka;
    const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}
__device__ __forceinline__ Slab4 slab_children4(const u32x4 q0, const u32x4 q1, const u32x4 q2, const u32x4 q3, const float ix, const float iy,
                                                const float iz, const float nx, const float ny, const float nz, const float tmn, const float tcur)
{
    Slab4 o;
    slab_piece(q0, ix, iy, iz, nx, ny, nz, tmn, tcur, o.t[0], o.ref[0]);
    slab_piece(q1, ix, iy, iz, nx, ny, nz, tmn, tcur, o.t[1], o.ref[1]);
    slab_piece(q2, ix, iy, iz, nx, ny, nz, tmn, tcur, o.t[2], o.ref[2]);
    slab_piece(q3, ix, iy, iz, nx, ny, nz, tmn, tcur, o.t[3], o.ref[3]);
    cmp_swap(o.t[0], o.ref[0], o.t[1], o.ref[1]); cmp_swap(o.t[2], o.ref[2], o.t[3], o.ref[3]);     // 5-comparator network
    cmp_swap(o.t[0], o.ref[0], o.t[2], o.ref[2]); cmp_swap(o.t[1], o.ref[1], o.t[3], o.ref[3]);
    cmp_swap(o.t[1], o.ref[1], o.t[2], o.ref[2]);
    const float inf = __builtin_inff();
    o.n = (o.t[0] < inf ? 1 : 0) + (o.t[1] < inf ? 1 : 0) + (o.t[2] < inf ? 1 : 0) + (o.t[3] < inf ? 1 : 0);
    return o;
}

__device__ __forceinline__ float shfl_f(float v, uint32_t src) { return __shfl(v, (int)src, 64); }
__device__ __forceinline__ uint32_t shfl_u(uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)src, 64); }

__device__ __forceinline__ uint32_t pool_pack(int ref, uint32_t slot)
{
    if (ref >= 0) return ((uint32_t)ref << 6) | slot;
    const uint32_t desc = ~(uint32_t)ref;
    return 0x80000000u | ((desc >> 4) << 8) | (((desc - 1u) & 3u) << 6) | slot;
}
// (2 / 3 / 5 / 8 levels, 1080p over 782 k alpha-textured triangles: BLEND everywhere 24.8 / 20.8 / 17.9 / 17.3 ms, hard-alpha MASK 2.81 / 2.74 / 2.86 /
//  3.20 ms, boxes MASK 2.16 / 2.15 / 2.24 / 2.52 ms: three fit the merge words and cost the scenes whose rays end early nothing)
#ifndef MIPT_AH_LEVELS
#define MIPT_AH_LEVELS 3
#endif
constexpr int kAhLevels = MIPT_AH_LEVELS;     // AH kernels: a traversal can keep a ray's 3 nearest candidates (keys only: level l of slot s at sh_best[l * 64 + s], over the
                                 // merge words sh_u / sh_v / sh_done, which the AH kernels do not use -- they re-test a candidate's triangle for u, v)
// AH kernels: the interval the BOX tests of a traversal use, a little wider than the ray's.  A traversal of the any-hit kernels starts at
// tmin = the t of the candidate the program just ignored and is pruned at a candidate's t, and a box that holds a candidate within
// rounding of such a t (a coincident surface, a decal a few ulps in front) can fail its slab test by an ulp although the triangle test --
// which decides, and is exact about the key -- would pass.  The oracle's traversal never stands at such a t (it collects a ray's
// candidates from the ray's own tmin in one walk), so the two would part ways.  Wider box intervals only weaken pruning.  The margin:
// 2^-19 of the magnitudes that enter a slab distance (box * idir - origin * idir).  (Boxes rounded outward to fp16, node formats 1 and 2,
// hide this; fp32 boxes -- format 0, scenes that do not suit fp16 boxes -- do not: tests/test_whitted.py runs both.)
__device__ __forceinline__ void ah_box_interval(const float nx, const float ny, const float nz, const float tmn, const float tcur, float& tmn_b, float& tcur_b)
{
    const float mg = (__builtin_fabsf(nx) + __builtin_fabsf(ny) + __builtin_fabsf(nz)) * 1.9073486e-6f;
    tmn_b = tmn - (mg + __builtin_fabsf(tmn) * 1.9073486e-6f);
    tcur_b = tcur == -__builtin_inff() ? tcur : tcur + (mg + __builtin_fabsf(tcur) * 1.9073486e-6f);
}
constexpr unsigned long long kAhOccludedKey = 0x007fffffull << 32;      // ord_bits(-inf) in the t half: below every key of a real candidate (t > tmin)
__device__ __forceinline__ uint32_t ord_bits(float f) { const uint32_t b = f2u(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float    ord_float(uint32_t k) { return u2f(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }


}  // namespace mipt
