// whitted_kernels.hip -- the glTF / PBR programs of SDK/cuda/whitted.cu as wavefront stages:
//   k_wh_raygen  <- __raygen__pinhole            (whitted.cu:44-97: pixel centre at subframe 0, jitter later)
//   k_wh_shade   <- __closesthit__radiance        (:149-289) up to its occlusion queries, __miss__constant_radiance (:139-142),
//                   getLocalGeometry              (cuda/LocalGeometry.h:59-127) on world-space triangles
//   k_wh_gather  <- the `result +=` lines once the occlusion flags are known, in light order
//   k_wh_resolve <- progressive lerp + make_color (:82-96)
//   k_wh_trace   <- optixTrace with the any-hit programs (:100-136) when the scene has MASK / BLEND materials
// Point and ambient lights; the BLEND continuation ray (:268-286) makes the pipeline a loop over up to
// MAX_TRACE_DEPTH levels whose results are composed back to front, like the recursion.
// schlick's powf(x, 5) is x^2 * x^2 * x (detmath: no libm transcendentals).
#include <cstdlib>
#include <cstring>

#include "kernels.h"
#include "detmath.h"
#include "wave_util.h"
#include "trav_core.h"
#include "inst_core.h"
#include "pool_core.h"
#include "pool_trace.h"
#include "inst_pool.h"

namespace mipt {

struct WhConsts {
    uint32_t width, height, subframe_index, n_paths;
    float eye[3], U[3], V[3], W[3], miss_color[3];
    const MiptLight* lights; uint32_t nlights;
    WhGeometry geo;
    const MiptMaterialData* materials; uint32_t nmaterials;
    const WhTexture* textures; uint32_t ntextures;
};

__device__ __forceinline__ f3 ld3w(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ float pow5(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x4 * x; }
__device__ __forceinline__ f3 w_schlick(f3 spec, float VdH) { return spec + (mk3(1.0f, 1.0f, 1.0f) - spec) * pow5(1.0f - VdH); }
__device__ __forceinline__ float w_vis(float NdL, float NdV, float alpha)
{
    const float a2 = alpha * alpha;
    const float g0 = NdL * __builtin_sqrtf(NdV * NdV * (1.0f - a2) + a2);
    const float g1 = NdV * __builtin_sqrtf(NdL * NdL * (1.0f - a2) + a2);
    return 2.0f * NdL * NdV / (g0 + g1);
}
__device__ __forceinline__ float w_ggx(float NdH, float alpha)
{
    const float a2 = alpha * alpha;
    const float x = NdH * NdH * (a2 - 1.0f) + 1.0f;
    return a2 / (3.14159265358979323846f * x * x);
}

// ---- tex2D<float4> for the texture objects of Scene::addSampler (Scene.cpp:624-652): normalised coordinates,
// normalised-float read, point or bilinear filter (identical sequence in oracle/pt_oracle.c: orc_sample_texture)
__device__ __forceinline__ int wrap_index(int i, int n, uint32_t mode)
{
    if (mode == 1u) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    if (mode == 2u) { int p = i % (2 * n); if (p < 0) p += 2 * n; return p < n ? p : 2 * n - 1 - p; }
    int p = i % n; return p < 0 ? p + n : p;
}
__device__ __forceinline__ float4 fetch_texel(const WhTexture& t, int x, int y)
{
    const size_t at = ((size_t)y * t.width + (size_t)x) * 4;
    if (t.bits == 16u) {
        const ushort4 p = *reinterpret_cast<const ushort4*>((const uint16_t*)t.texels + at);
        return make_float4((float)p.x / 65535.0f, (float)p.y / 65535.0f, (float)p.z / 65535.0f, (float)p.w / 65535.0f);
    }
    const uchar4 p = *reinterpret_cast<const uchar4*>((const uint8_t*)t.texels + at);
    return make_float4((float)p.x / 255.0f, (float)p.y / 255.0f, (float)p.z / 255.0f, (float)p.w / 255.0f);
}
__device__ __forceinline__ float lerp1(float a, float b, float t) { return a + t * (b - a); }
__device__ float4 sample_texture(const WhTexture& t, float u, float v)
{
    const int W = (int)t.width, H = (int)t.height;
    if (t.filter == 0u)
        return fetch_texel(t, wrap_index((int)__builtin_floorf(u * (float)W), W, t.address_s), wrap_index((int)__builtin_floorf(v * (float)H), H, t.address_t));
    const float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    const float x0f = __builtin_floorf(fx), y0f = __builtin_floorf(fy);
    const float a = fx - x0f, b = fy - y0f;
    const int x0 = wrap_index((int)x0f, W, t.address_s), x1 = wrap_index((int)x0f + 1, W, t.address_s);
    const int y0 = wrap_index((int)y0f, H, t.address_t), y1 = wrap_index((int)y0f + 1, H, t.address_t);
    const float4 t00 = fetch_texel(t, x0, y0), t10 = fetch_texel(t, x1, y0), t01 = fetch_texel(t, x0, y1), t11 = fetch_texel(t, x1, y1);
    return make_float4(lerp1(lerp1(t00.x, t10.x, a), lerp1(t01.x, t11.x, a), b), lerp1(lerp1(t00.y, t10.y, a), lerp1(t01.y, t11.y, a), b),
                       lerp1(lerp1(t00.z, t10.z, a), lerp1(t01.z, t11.z, a), b), lerp1(lerp1(t00.w, t10.w, a), lerp1(t01.w, t11.w, a), b));
}

struct WhTexcoord { float u, v; f3 dpdu, dpdv; };
struct WhLocalGeom { f3 P, N; WhTexcoord tc[2]; float4 color; };

constexpr uint32_t kWhMaxTraceDepth = 8u;                // whitted::MAX_TRACE_DEPTH, cuda/whitted.h:42

// sampleTexture<float4> (cuda/LocalShading.h:37-53): false when the material has no such (decodable) texture
__device__ __forceinline__ bool sample_material_texture(const WhConsts& c, const MiptTextureInfo& ti, const WhLocalGeom& g, float4& out)
{
    if (ti.texture < 0 || (uint32_t)ti.texture >= c.ntextures) return false;
    const WhTexture t = c.textures[ti.texture];
    if (t.width == 0u || !t.texels) return false;
    const WhTexcoord& tc = g.tc[ti.texcoord == 1 ? 1 : 0];
    const float ux = tc.u * ti.scale[0], uy = tc.v * ti.scale[1];
    const float tu = (ux * ti.rotation[1] + uy * ti.rotation[0]) + ti.offset[0];
    const float tv = (ux * -ti.rotation[0] + uy * ti.rotation[1]) + ti.offset[1];
    out = sample_texture(t, tu, tv);
    return true;
}

// The shading tables of a hit: the flattened ones and the primitive index itself, or (mipt_whitted_set_geometry_instanced) the tables of
// the instance's MESH, the mesh-local index and the instance's record -- what the reference's per-instance SBT records point at.
struct WhSite { WhTables t; uint32_t l; const WhInstGeo* inst; };
__device__ __forceinline__ WhSite geo_site(const WhConsts& c, uint32_t prim)
{
    WhSite s;
    s.t = c.geo; s.l = prim; s.inst = nullptr;
    if (c.geo.prim_inst) {
        s.inst = &c.geo.inst[c.geo.prim_inst[prim]];
        s.t = c.geo.meshes[s.inst->mesh];
        s.l = prim - s.inst->prim_base;
    }
    return s;
}
// a corner into world space: mipt_scene_flatten's m4_apply, term by term (scene.cpp:164: sum = 0; sum += m[j] * v[j])
__device__ __forceinline__ f3 inst_point(const WhInstGeo& I, float4 v)
{
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { float sum = 0.0f; sum += I.m[4 * i] * v.x; sum += I.m[4 * i + 1] * v.y; sum += I.m[4 * i + 2] * v.z; sum += I.m[4 * i + 3] * 1.0f; r[i] = sum; }
    return mk3(r[0], r[1], r[2]);
}
// a normal: mipt_scene_flatten_normals' (M^-T) n, un-normalised
__device__ __forceinline__ f3 inst_normal(const WhInstGeo& I, float4 n)
{
    return mk3(((I.cof[0] * n.x + I.cof[1] * n.y) + I.cof[2] * n.z) * I.inv_det, ((I.cof[3] * n.x + I.cof[4] * n.y) + I.cof[5] * n.z) * I.inv_det,
               ((I.cof[6] * n.x + I.cof[7] * n.y) + I.cof[8] * n.z) * I.inv_det);
}

// getLocalGeometry (cuda/LocalGeometry.h:59-176) on world-space corners
__device__ __forceinline__ void local_geometry(const WhConsts& c, uint32_t prim_in, float bu, float bv, WhLocalGeom& g)
{
    const WhSite site = geo_site(c, prim_in);
    const WhTables& G = site.t;
    const uint32_t prim = site.l;
    const float4 a0 = G.vertices4[(size_t)prim * 3u], a1 = G.vertices4[(size_t)prim * 3u + 1u], a2 = G.vertices4[(size_t)prim * 3u + 2u];
    f3 P0 = mk3(a0.x, a0.y, a0.z), P1 = mk3(a1.x, a1.y, a1.z), P2 = mk3(a2.x, a2.y, a2.z);
    if (site.inst) { P0 = inst_point(*site.inst, a0); P1 = inst_point(*site.inst, a1); P2 = inst_point(*site.inst, a2); }
    const float b0 = 1.0f - bu - bv;
    g.P = b0 * P0 + bu * P1 + bv * P2;
    g.color = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (G.colors4 && (!G.color_present || G.color_present[prim])) {
        const float4* cc = G.colors4 + (size_t)prim * 3u;
        const float4 c0 = cc[0], c1 = cc[1], c2 = cc[2];
        g.color = make_float4((b0 * c0.x + bu * c1.x) + bv * c2.x, (b0 * c0.y + bu * c1.y) + bv * c2.y,
                              (b0 * c0.z + bu * c1.z) + bv * c2.z, (b0 * c0.w + bu * c1.w) + bv * c2.w);
    }
    g.N = normalize(cross(P1 - P0, P2 - P0));
    if (G.normals4) {
        const float4 n0 = G.normals4[(size_t)prim * 3u], n1 = G.normals4[(size_t)prim * 3u + 1u], n2 = G.normals4[(size_t)prim * 3u + 2u];
        if (n0.w != 0.0f) {
            f3 N0 = mk3(n0.x, n0.y, n0.z), N1 = mk3(n1.x, n1.y, n1.z), N2 = mk3(n2.x, n2.y, n2.z);
            if (site.inst) { N0 = inst_normal(*site.inst, n0); N1 = inst_normal(*site.inst, n1); N2 = inst_normal(*site.inst, n2); }
            g.N = normalize(b0 * N0 + bu * N1 + bv * N2);
        }
    }
    const f3 dp1 = P0 - P2, dp2 = P1 - P2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        WhTexcoord& tc = g.tc[j];
        if (G.texcoords[j] && (!G.texcoord_present[j] || G.texcoord_present[j][prim])) {
            const float2* uv = G.texcoords[j] + (size_t)prim * 3u;
            const float2 uv0 = uv[0], uv1 = uv[1], uv2 = uv[2];
            tc.u = (b0 * uv0.x + bu * uv1.x) + bv * uv2.x;
            tc.v = (b0 * uv0.y + bu * uv1.y) + bv * uv2.y;
            const float du1 = uv0.x - uv2.x, du2 = uv1.x - uv2.x, dv1 = uv0.y - uv2.y, dv2 = uv1.y - uv2.y;
            const float det = du1 * dv2 - dv1 * du2;
            const float invdet = 1.0f / det;
            tc.dpdu = (dv2 * dp1 - dv1 * dp2) * invdet;
            tc.dpdv = ((-du2) * dp1 + du1 * dp2) * invdet;
        } else {
            tc.u = bu; tc.v = bv;
            tc.dpdu = -dp1;
            tc.dpdv = -dp1 + dp2;
        }
    }
}

__device__ __forceinline__ uint32_t prim_material(const WhConsts& c, uint32_t prim)
{
    uint32_t mi = 0u;
    if (c.geo.prim_inst) {
        const WhInstGeo& I = c.geo.inst[c.geo.prim_inst[prim]];
        const uint32_t* tab = c.geo.meshes[I.mesh].material_index;
        mi = tab ? tab[prim - I.prim_base] : 0u;
    } else mi = c.geo.material_index ? c.geo.material_index[prim] : 0u;
    return mi >= c.nmaterials ? c.nmaterials - 1u : mi;
}

// ---- optixTrace with the any-hit programs of cuda/whitted.cu:100-136 ------------------------------------
// Used instead of the fast kernels when a scene has MASK / BLEND materials.  MODE 0: traceRadiance (closest hit, back-face
// culling, __anyhit__radiance) -> hit records; MODE 1: traceOcclusion (__anyhit__occlusion) -> attenuation per ray.
//
// The programs run DEFERRED: a ray's candidates are taken in the order of (t, primitive index) -- a closest-hit traversal finds the
// first, the any-hit program is run on it, and when the program ignores it (optixIgnoreIntersection: below the cutoff, or a surface
// that only attenuates) the ray is traced again for the closest candidate BEYOND the key (t, primitive): tmin = t, and candidates AT
// tmin count when their primitive index is greater (Trav::skip; coincident surfaces are visited one by one).  That is
//   * MODE 0: the closest candidate the program accepts -- what a traversal that runs the program at every candidate reports;
//   * MODE 1: the product of the attenuations (1 - alpha) in the order of (t, primitive index), 0 as soon as a candidate is opaque or
//     the product reaches 0.  OptiX leaves the order of any-hit invocations open; this order is the build's definition
//     (oracle/pt_oracle.c: whitted occlusion), independent of the acceleration structure.
// So the traversal itself is the entry pool's (pool_trace.h / inst_pool.h, AH form) -- k_wh_trace_pool / k_wh_trace_pool_inst: a
// wave's chunk of rays, all lanes serving whichever rays are still open -- and the texture fetch happens once per candidate that
// IS a ray's next one instead of inside the traversal loop at every candidate in the ray's interval (round 2's one-ray-per-thread
// walk).  A ray's first traversal finds ONE candidate (pruned like a closest-hit ray: rays that end at their first solid texel pay no
// more); once a candidate has been ignored the ray asks for its THREE nearest per traversal (kAhLevels: per slot three 64-bit keys
// in LDS, each level keeping the minimum of what reaches it and handing the larger key down; pruned at the third) -- a third of the
// restarts for rays through many surfaces.  The pool reports keys only; u, v and the geometry flags of a key come from testing its
// triangle again (candidate_record: the builders' primitive -> record tables).
// Measured, 1080p subframe over 2 000 instances / 782 k triangles, ALL of them alpha-textured (tools/bench_alpha.py,
// profiles/r03_bench_alpha.txt): MASK materials with hard alpha (a ray ends at the first solid texel) 2.22 ms against 1.10 ms for the
// same scene with opaque materials; BLEND films everywhere -- an occlusion ray crosses 7 attenuating surfaces on average, up to 78, and
// none ends it, so every ray is traversed to its end -- 18.0 ms (one candidate per traversal, 32 rays per wave: 33.8) against round 2's 45.5 ms.
// Tried for the occlusion rays and dropped: ONE traversal that records every candidate of the interval (16 records per ray,
// narrower windows for rays with more) and then sorts -- without a running closest hit nothing prunes the traversal, and the rays that
// end early pay for the whole interval: 20 ms on the MASK scene, 40-76 ms on the BLEND scene.
// k_wh_trace / k_wh_trace_inst: the same loop one ray per thread, one candidate per traversal: structures the pool cannot walk, and
// trees of up to 8 levels (wh_trace_on_pool).
__device__ __forceinline__ bool candidate_alpha(const WhConsts& c, uint32_t prim, float u, float v, uint32_t& mi, float& alpha)
{
    mi = prim_material(c, prim);
    const MiptTextureInfo ti = c.materials[mi].base_color_tex;
    if (ti.texture < 0 || (uint32_t)ti.texture >= c.ntextures) return false;
    WhLocalGeom g;
    local_geometry(c, prim, u, v, g);
    float4 tx;
    if (!sample_material_texture(c, ti, g, tx)) return false;
    alpha = tx.w;
    return true;
}

// the any-hit program at a ray's next candidate; true: the ray is finished (MODE 0: the candidate is its hit; MODE 1: occluded)
template <int MODE>
__device__ __forceinline__ bool anyhit_step(const WhConsts& c, uint32_t gflags, uint32_t prim, float u, float v, float& occlusion)
{
    if ((gflags & 1u) == 0u) {                                                           // any-hit program enabled
        uint32_t mi; float alpha;
        if (candidate_alpha(c, prim, u, v, mi, alpha)) {
            const MiptMaterialData& m = c.materials[mi];
            if (MODE == 0) {
                if (alpha < m.alpha_cutoff) return false;                                // optixIgnoreIntersection
            } else if (m.alpha_mode != 0) {
                if (m.alpha_mode == 1 && alpha < m.alpha_cutoff) return false;
                const float att = occlusion * (1.0f - alpha);
                if (att > 0.0f) { occlusion = att; return false; }
            }
        }
    }
    if (MODE != 0) occlusion = 0.0f;                                                     // optixTerminateRay
    return true;
}

// rays per wave, from the ray count the DEVICE knows (a level's queue length; the host only knows the capacity): a chunk's traversals
// run one after the other, so a launch of few rays finishes sooner with few rays per wave -- all 64 lanes serve them.  Big launches take
// 64 (1080p over 782 k triangles, everything MASK / boxes MASK / BLEND everywhere: one level 2.74 / 2.17 / 21.0 ms with 32, 2.27 / 1.75 /
// 19.0 with 64; two levels, whose pool entries are twice as wide, with a pool of 40 LDS rows instead of 24 -- 64 rays overfill the
// smaller one: 3.69 / 2.90 / 29.0 ms with 32 rays and 24 rows, 2.95 / 2.33 / 26.7 with 64 and 40; one level gains nothing from more rows).
__device__ __forceinline__ uint32_t ah_rays_per_wave(uint32_t fixed, uint32_t n, uint32_t cus)
{
    if (fixed != 0u) return fixed;
    const uint32_t q = n / (cus * 64u);
    return q >= 64u ? 64u : (q >= 32u ? 32u : (q >= 16u ? 16u : (q >= 8u ? 8u : 4u)));
}

struct WhTraceArgs {
    DeviceBvh bvh;
    const float4* ray_o; const float4* ray_d;
    const uint32_t* n_ptr; uint32_t n_max;
    float4* hits; float* attenuation;
    uint32_t ppw, cus;             // ppw 0: chosen on the device (ah_rays_per_wave)
    uint32_t first_k;              // candidates a ray's FIRST traversal asks for (1; kAhLevels for the occlusion rays of scenes with BLEND materials)
    const uint32_t* open_list;     // MODE 0, seeded launch: the rays to trace (k_wh_seed; n_ptr counts them), each from beyond the candidate in hits[ray]
};

// A candidate the traversal reports by its key (t, flattened primitive) alone: u, v and the geometry flags by testing its triangle again --
// the record through the builders' tables (DeviceBvh::prim_slot; two-level: the instance is the last one whose prim_base is <= prim,
// then DeviceInstance::prim_slot), the ray taken to object space as the traversal takes it, the traversal's own sequence of operations:
// t comes out as the key has it, bit for bit.
__device__ __forceinline__ void candidate_record(const DeviceBvh& bvh, f3 o, f3 d, const uint32_t prim, float& t, float& u, float& v, uint32_t& gflags)
{
    const float4* tp;
    if (bvh.inst) {
        uint32_t lo = 0u, hi = bvh.n_inst;                  // invariant: inst[lo].prim_base <= prim < inst[hi].prim_base (hi = n: the end)
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (bvh.inst[mid].prim_base <= prim) lo = mid; else hi = mid; }
        const DeviceInstance& in = bvh.inst[lo];
        tp = in.tris + (size_t)in.prim_slot[prim - in.prim_base] * 3u;
        const u32x4* m = reinterpret_cast<const u32x4*>(&in);
        const RayO O = to_object(m[0], m[1], m[2], o, d);
        o = O.o; d = O.d;
    } else {
        tp = bvh.tris + (size_t)bvh.prim_slot[prim] * 3u;
    }
    const float4 n0 = tp[0], n1 = tp[1], n2 = tp[2];
    const f3 e1 = mk3(n1.x, n1.y, n1.z), e2 = mk3(n2.x, n2.y, n2.z);
    const f3 p = fcross(d, e2);
    const float det = fdot(e1, p);
    const float inv = 1.0f / det;
    const f3 tv = o - mk3(n0.x, n0.y, n0.z);
    u = fdot(tv, p) * inv;
    const f3 q = fcross(tv, e1);
    v = fdot(d, q) * inv;
    t = fdot(e2, q) * inv;
    gflags = f2u(n1.w);
}

// one ray's bookkeeping after a traversal.  `lv`: the slot's candidate levels (sh_best: level l at lv[l * 64]), the T.ahk nearest remaining
// candidates in order.  The program runs over them; the first one it accepts ends the ray; when it ignores them all the ray goes on
// beyond the last -- and asks for kAhLevels candidates per traversal from then on (a ray
// whose first candidate is ignored is likely to meet more such surfaces; a ray's FIRST traversal keeps to one candidate, so that rays
// that end at their first solid texel are pruned as closest-hit rays are).
template <int MODE>
__device__ __forceinline__ void anyhit_advance(const WhConsts& c, const DeviceBvh& bvh, Trav& T, bool& open, float& occlusion, float4& rec, const unsigned long long* lv)
{
    const uint32_t want = T.ahk;
    for (uint32_t l = 0; l < want; ++l) {
        const unsigned long long key = lv[l * 64u];
        if (MODE != 0 && key == kAhOccludedKey) { occlusion = 0.0f; open = false; return; }  // a candidate without program somewhere in the interval
        const uint32_t prim = (uint32_t)key;
        if (prim == 0xffffffffu) { open = false; return; }                                   // nothing left in the interval
        float t, u, v; uint32_t gflags;
        candidate_record(bvh, T.o, T.d, prim, t, u, v, gflags);
        t = ord_float((uint32_t)(key >> 32));
        if (anyhit_step<MODE>(c, gflags, prim, u, v, occlusion)) { open = false; rec = make_float4(t, u2f(prim), u, v); return; }
        T.tmin = t; T.skip = prim;
    }
    T.ahk = (uint32_t)kAhLevels;
}

// A seeded launch (MODE 0): the fast closest-hit kernels -- tiles of 64 rays for a level's millions of rays -- have found every ray's
// closest candidate with all geometry taken as opaque; that candidate is exactly the first one of the loop below.  k_wh_seed runs the
// program on it, one ray per thread: accepted (or no program, or a miss) -> the hit record stands; IGNORED -> the ray's index joins
// the list `open` (one atomic per workgroup), and only the listed rays, densely packed into chunks, are traversed here, starting
// beyond their candidate.  Default over two-level structures only.  1080p, 2 000 instances, boxes MASK / everything MASK
// (profiles/r03_bench_alpha.txt): two-level 3.29 -> 2.86 / 4.41 -> 3.92 ms; one level 2.10 -> 2.12 / 2.87 -> 2.90 ms -- there the
// tiled launch (0.35 ms) + this filter (0.05) + the listed rays (0.37: few rays, but each ignored candidate is one more traversal from
// the root, and the longest such chain sets the launch's duration) only just beat the unseeded launch (0.81).
// (Seeding inside the traversal kernel instead, ray i in lane i: a chunk with ONE open ray costs a traversal's latency all the same,
// and most chunks have one -- no gain at all.)
__global__ __launch_bounds__(256) void k_wh_seed(WhConsts c, DeviceBvh bvh, const float4* ray_o, const float4* ray_d, const uint32_t* n_ptr, uint32_t n_max,
                                                 const float4* hits, uint32_t* open_list, DeviceCounters* ctr)
{
    __shared__ uint32_t s2[2];
    if (threadIdx.x == 0u) s2[0] = 0u;
    __syncthreads();
    uint32_t n = n_max;
    if (n_ptr) { const uint32_t m = *n_ptr; n = m < n ? m : n; }
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {       // (uniform trip count: block_append synchronises)
        const uint32_t ray = base + threadIdx.x;
        bool push = false;
        if (ray < n) {
            const float4 h = hits[ray];
            const uint32_t prim = f2u(h.y);
            float occlusion = 1.0f;
            if (prim != 0xffffffffu) {
                float t, u, v; uint32_t gflags;
                const float4 ro = ray_o[ray], rd = ray_d[ray];
                candidate_record(bvh, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), prim, t, u, v, gflags);
                push = !anyhit_step<0>(c, gflags, prim, h.z, h.w, occlusion);
            }
        }
        const uint32_t at = block_append(push, &ctr->n_open, s2);
        if (push) open_list[at] = ray;
    }
}

#ifndef MIPT_WH_AH_ROWS
#define MIPT_WH_AH_ROWS 24
#endif
#ifndef MIPT_WH_INST_ROWS
#define MIPT_WH_INST_ROWS 40       // (two-level pool entries are 8 bytes: 40 rows hold what 64 rays per wave need, ah_rays_per_wave)
#endif
// (experiments: 7 waves per SIMD -- 72 VGPRs -- change nothing, 2.27 / 1.80 / 18.6 ms on the three alpha scenes; 20 LDS rows cost 30-50 %
//  with 64 rays per wave: the pool runs full)
#ifdef MIPT_WH_AH_WAVES
#define MIPT_WH_AH_ATTR __attribute__((amdgpu_waves_per_eu(MIPT_WH_AH_WAVES, MIPT_WH_AH_WAVES)))
#else
#define MIPT_WH_AH_ATTR
#endif
template <int MODE, int Q>
__global__ __launch_bounds__(64) MIPT_WH_AH_ATTR void k_wh_trace_pool(WhConsts c, WhTraceArgs a)
{
    constexpr int kLevelWords = kAhLevels * 128 > 384 ? kAhLevels * 128 : 384;      // the candidate levels behind the pool (3 fit the merge words)
    constexpr int kPoolCap = MIPT_WH_AH_ROWS * 64 - 384;
    static_assert(kPoolCap % 2 == 0, "sh_best holds 64-bit keys: it must start on an even word");
    __shared__ __attribute__((aligned(16))) uint32_t stk[MIPT_WH_AH_ROWS * 64 - 384 + kLevelWords];
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(stk + kPoolCap);
    float* const sh_u = reinterpret_cast<float*>(stk + kPoolCap + 128);
    float* const sh_v = reinterpret_cast<float*>(stk + kPoolCap + 192);
    uint32_t* const sh_done = stk + kPoolCap + 320;
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    const uint32_t ppw = ah_rays_per_wave(a.ppw, n, a.cus);
    const uint32_t chunks = (n + ppw - 1u) / ppw;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        uint32_t ray = chunk * ppw + (uint32_t)lane;
        const bool mine = (uint32_t)lane < ppw && ray < n;
        if (MODE == 0 && a.open_list != nullptr && mine) ray = a.open_list[ray];
        Trav T;
        T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f; T.skip = 0xffffffffu; T.ahk = a.first_k;
        T.o = mk3(0.f, 0.f, 0.f); T.d = T.o; T.idir = T.o; T.nood = T.o; T.tmin = 0.0f; T.tmax = 0.0f;
        if (mine) trav_init(T, a.ray_o[ray], a.ray_d[ray]);
        if (MODE == 0 && a.open_list != nullptr && mine) { const float4 h = a.hits[ray]; T.tmin = h.x; T.skip = f2u(h.y); T.ahk = (uint32_t)kAhLevels; }      // beyond the ignored candidate
        bool open = mine && T.node != kSentinel;
        float occlusion = 1.0f;
        float4 rec = make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
        while (__ballot(open) != 0ull) {
            tail_trace<0, Q, kPoolCap, MODE == 0 ? 1 : 2>(blob, a.bvh, stk, sh_best, sh_u, sh_v, sh_done, lane, T, open, MODE == 0);
            if (open) anyhit_advance<MODE>(c, a.bvh, T, open, occlusion, rec, sh_best + lane);
        }
        if (mine) {
            if (MODE == 0) a.hits[ray] = rec;
            else           a.attenuation[ray] = occlusion;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_wh_trace_pool_inst(WhConsts c, WhTraceArgs a)
{
    constexpr int kRows = MIPT_WH_INST_ROWS;
    constexpr int kLevelWords = kAhLevels * 128 > 384 ? kAhLevels * 128 : 384;
    constexpr int kCap = (kRows * 64 - 384) / 2;
    __shared__ __attribute__((aligned(16))) uint2 stk2[(kRows * 64 - 384 + kLevelWords) / 2];
    uint32_t* const words = reinterpret_cast<uint32_t*>(stk2);
    unsigned long long* const sh_best = reinterpret_cast<unsigned long long*>(words + 2 * kCap);
    float* const sh_u = reinterpret_cast<float*>(words + 2 * kCap + 128);
    float* const sh_v = reinterpret_cast<float*>(words + 2 * kCap + 192);
    uint32_t* const sh_done = words + 2 * kCap + 320;
    const int lane = (int)threadIdx.x;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bvh.nodes, 0, (int)a.bvh.inst_blob_bytes, 0x00020000);
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    const uint32_t ppw = ah_rays_per_wave(a.ppw, n, a.cus);
    const uint32_t chunks = (n + ppw - 1u) / ppw;
#if MIPT_INST_STATS
    uint32_t ist[6] = { 0u, 0u, 0u, 0u, 0u, 0u };
#endif
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        uint32_t ray = chunk * ppw + (uint32_t)lane;
        const bool mine = (uint32_t)lane < ppw && ray < n;
        if (MODE == 0 && a.open_list != nullptr && mine) ray = a.open_list[ray];
        Trav T;
        T.node = kSentinel; T.sp = 0; T.tbest = 0.0f; T.prim = 0xffffffffu; T.u = 0.0f; T.v = 0.0f; T.skip = 0xffffffffu; T.ahk = a.first_k;
        T.o = mk3(0.f, 0.f, 0.f); T.d = T.o; T.idir = T.o; T.nood = T.o; T.tmin = 0.0f; T.tmax = 0.0f;
        if (mine) trav_init(T, a.ray_o[ray], a.ray_d[ray]);
        if (MODE == 0 && a.open_list != nullptr && mine) { const float4 h = a.hits[ray]; T.tmin = h.x; T.skip = f2u(h.y); T.ahk = (uint32_t)kAhLevels; }      // beyond the ignored candidate
        bool open = mine && T.node != kSentinel;
        float occlusion = 1.0f;
        float4 rec = make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
        while (__ballot(open) != 0ull) {
            tail_trace_inst<0, kCap, false, MODE == 0 ? 1 : 2>(blob, a.bvh, stk2, sh_best, sh_u, sh_v, sh_done, lane, T, open, MODE == 0 INST_STATS_ARG);
            if (open) anyhit_advance<MODE>(c, a.bvh, T, open, occlusion, rec, sh_best + lane);
        }
        if (mine) {
            if (MODE == 0) a.hits[ray] = rec;
            else           a.attenuation[ray] = occlusion;
        }
    }
}

// ---- the same loop, one ray per thread (structures without the entry pool's mirrors) -----------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void k_wh_trace(WhConsts c, WhTraceArgs a)
{
    __shared__ uint32_t stk_lds[kStackEntries * 64];
    const int lane = (int)threadIdx.x;
    const Stack stk{ stk_lds, lane };
    const DeviceBvh& bvh = a.bvh;
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    for (uint32_t ray = blockIdx.x * 64u + threadIdx.x; ray < n; ray += gridDim.x * 64u) {
        const float4 ro = a.ray_o[ray], rd = a.ray_d[ray];
        float occlusion = 1.0f;
        float4 rec = make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
        float tmin = ro.w; uint32_t skip = 0xffffffffu;
        for (bool open = true; open;) {
            Trav T;
            trav_init(T, make_float4(ro.x, ro.y, ro.z, tmin), rd);
            uint32_t gflags_best = 0u;
            float best = T.tbest;                                  // exact: the triangle tests' bound; T.tmin / T.tbest: the box tests' (ah_box_interval)
            const float tmax = T.tmax;
            float box_lo, box_hi;
            ah_box_interval(T.nood.x, T.nood.y, T.nood.z, tmin, best, box_lo, box_hi);
            T.tmin = box_lo; T.tbest = box_hi;
            while (T.node != kSentinel) {
                while (T.node >= 0) visit_inner(T, bvh.nodes, stk);
                if (T.node == kSentinel) break;
                const uint32_t desc = ~(uint32_t)T.node;
                const uint32_t first = desc >> 4, cnt = desc & 15u;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const float4* tp = bvh.tris + (size_t)(first + i) * 3u;
                    const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                    const f3 e1 = mk3(tb.x, tb.y, tb.z), e2 = mk3(tc.x, tc.y, tc.z);
                    const f3 p = fcross(T.d, e2);
                    const float det = fdot(e1, p);
                    if (det == 0.0f) continue;
                    const float inv = 1.0f / det;
                    const f3 tv = T.o - mk3(ta.x, ta.y, ta.z);
                    const float u = fdot(tv, p) * inv;
                    const f3 q = fcross(tv, e1);
                    const float v = fdot(T.d, q) * inv;
                    const float t = fdot(e2, q) * inv;
                    if (!(u >= 0.0f && v >= 0.0f && u + v <= 1.0f)) continue;
                    const uint32_t prim = f2u(ta.w), gflags = f2u(tb.w);
                    if (!((t > tmin || (t == tmin && prim > skip)) && t < tmax)) continue;
                    if (MODE == 0 && det < 0.0f && (gflags & 4u) == 0u) continue;            // back face of a single-sided triangle
                    if (MODE != 0 && (gflags & 1u) != 0u) { occlusion = 0.0f; T.prim = 0xffffffffu; T.sp = 0; T.node = kSentinel; break; }   // no program: occluded, wherever it lies
                    if (t < best || (t == best && prim < T.prim)) {
                        best = t; T.prim = prim; T.u = u; T.v = v; gflags_best = gflags;
                        ah_box_interval(T.nood.x, T.nood.y, T.nood.z, tmin, best, box_lo, box_hi);
                        T.tbest = box_hi;
                    }
                }
                T.node = stk.pop(T.sp);
            }
            if (T.prim == 0xffffffffu) break;
            if (anyhit_step<MODE>(c, gflags_best, T.prim, T.u, T.v, occlusion)) { rec = make_float4(best, u2f(T.prim), T.u, T.v); open = false; }
            else { tmin = best; skip = T.prim; }
        }
        if (MODE == 0) a.hits[ray] = rec;
        else           a.attenuation[ray] = occlusion;
    }
}

// ... over a two-level (instance) structure (instances.hip's walk: the world ray over the instances' boxes, the object-space ray --
// not re-normalised, so distances carry over -- through each mesh).  A candidate's primitive is its FLATTENED index (prim_base +
// local), which is what the shading tables and candidate_alpha are laid out by.
template <int MODE>
__global__ __launch_bounds__(64) void k_wh_trace_inst(WhConsts c, WhTraceArgs a)
{
    __shared__ uint32_t stk_lds[kInstStack * 64];
    const int lane = (int)threadIdx.x;
    const Stack2 stk{ stk_lds, lane };
    const DeviceBvh& bvh = a.bvh;
    uint32_t n = a.n_max;
    if (a.n_ptr) { const uint32_t m = *a.n_ptr; n = m < n ? m : n; }
    for (uint32_t ray = blockIdx.x * 64u + threadIdx.x; ray < n; ray += gridDim.x * 64u) {
        const float4 ro = a.ray_o[ray], rd = a.ray_d[ray];
        RayO W;
        W.o = mk3(ro.x, ro.y, ro.z); W.d = mk3(rd.x, rd.y, rd.z);
        prepare(W);
        const float tmax = rd.w;
        float tmin = ro.w; uint32_t skip = 0xffffffffu;
        float occlusion = 1.0f;
        float4 rec = make_float4(-1.0f, u2f(0xffffffffu), 0.0f, 0.0f);
        for (bool open = tmax > tmin; open;) {
            float tbest = tmax, bu = 0.0f, bv = 0.0f; uint32_t bprim = 0xffffffffu, bflags = 0u;
            int sp = 0;
            int node = 0;
            bool dead = false;                                  // MODE 1: a candidate without program was met
            while (node != kSentinel && !dead) {
                float w_lo, w_hi;                                   // the box tests' interval, a little wider than the ray's (ah_box_interval)
                ah_box_interval(W.nood.x, W.nood.y, W.nood.z, tmin, tbest, w_lo, w_hi);
                while (node >= 0) node = step_inner(bvh.nodes, node, W, w_lo, w_hi, stk, sp, 0);
                if (node == kSentinel) break;
                const uint32_t idesc = ~(uint32_t)node, ifirst = idesc >> 4, icnt = idesc & 15u;
                for (uint32_t k = 0; k < icnt && !dead; ++k) {
                    const DeviceInstance& in = bvh.inst[bvh.inst_order[ifirst + k]];
                    RayO O;
                    const f3 mo = xform_vec(in.inv, W.o);
                    O.o = mk3(mo.x + in.inv[3], mo.y + in.inv[7], mo.z + in.inv[11]);
                    O.d = xform_vec(in.inv, W.d);
                    prepare(O);
                    const int base = sp;
                    int mn = 0;
                    while (mn != kSentinel && !dead) {
                        float o_lo, o_hi;
                        ah_box_interval(O.nood.x, O.nood.y, O.nood.z, tmin, tbest, o_lo, o_hi);
                        while (mn >= 0) mn = step_inner(in.nodes, mn, O, o_lo, o_hi, stk, sp, base);
                        if (mn == kSentinel) break;
                        const uint32_t desc = ~(uint32_t)mn, first = desc >> 4, cnt = desc & 15u;
                        for (uint32_t i = 0; i < cnt; ++i) {
                            const float4* tp = in.tris + (size_t)(first + i) * 3u;
                            const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                            const f3 e1 = mk3(tb.x, tb.y, tb.z), e2 = mk3(tc.x, tc.y, tc.z);
                            const f3 p = fcross(O.d, e2);
                            const float det = fdot(e1, p);
                            if (det == 0.0f) continue;
                            const float inv = 1.0f / det;
                            const f3 tv = O.o - mk3(ta.x, ta.y, ta.z);
                            const float u = fdot(tv, p) * inv;
                            const f3 q = fcross(tv, e1);
                            const float v = fdot(O.d, q) * inv;
                            const float t = fdot(e2, q) * inv;
                            if (!(u >= 0.0f && v >= 0.0f && u + v <= 1.0f)) continue;
                            const uint32_t prim = in.prim_base + f2u(ta.w), gflags = f2u(tb.w);
                            if (!((t > tmin || (t == tmin && prim > skip)) && t < tmax)) continue;
                            if (MODE == 0 && det < 0.0f && (gflags & 4u) == 0u) continue;        // back face of a single-sided triangle
                            if (MODE != 0 && (gflags & 1u) != 0u) { occlusion = 0.0f; dead = true; break; }
                            if (bprim == 0xffffffffu || t < tbest || (t == tbest && prim < bprim)) { tbest = t; bprim = prim; bu = u; bv = v; bflags = gflags; }
                        }
                        mn = stk.pop(sp, base);
                    }
                    sp = base;
                }
                node = stk.pop(sp, 0);
            }
            if (dead || bprim == 0xffffffffu) break;
            if (anyhit_step<MODE>(c, bflags, bprim, bu, bv, occlusion)) { rec = make_float4(tbest, u2f(bprim), bu, bv); open = false; }
            else { tmin = tbest; skip = bprim; }
        }
        if (MODE == 0) a.hits[ray] = rec;
        else           a.attenuation[ray] = occlusion;
    }
}

// ---- raygen ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wh_raygen(WhConsts c, PathQueue q, uint32_t* n_levels)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_paths) return;
    const uint32_t tiles_x = (c.width + 7u) >> 3;
    const uint32_t tile = i >> 6, in = i & 63u;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const uint32_t px = tx * 8u + (in & 7u), py = ty * 8u + (in >> 3);
    if (px >= c.width || py >= c.height) {
        q.ray_o[i] = make_float4(0.f, 0.f, 0.f, 1.0f);
        q.ray_d[i] = make_float4(0.f, 0.f, 1.f, 0.0f);
        q.state_b[i] = make_uint2(kInvalidSlot, 0u);
        return;
    }
    uint32_t seed = tea4(py * c.width + px, c.subframe_index);
    float jx = 0.5f, jy = 0.5f;
    if (c.subframe_index != 0u) { jx = rnd(seed); jy = rnd(seed); }
    const float dx = 2.0f * (((float)px + jx) / (float)c.width) - 1.0f;
    const float dy = 2.0f * (((float)py + jy) / (float)c.height) - 1.0f;
    const f3 dir = normalize(dx * ld3w(c.U) + dy * ld3w(c.V) + ld3w(c.W));
    q.ray_o[i] = make_float4(c.eye[0], c.eye[1], c.eye[2], 0.0f);
    q.ray_d[i] = make_float4(dir.x, dir.y, dir.z, 1e16f);
    q.state_b[i] = make_uint2(py * c.width + px, 0u);                            // payload depth 0
    n_levels[py * c.width + px] = 0u;
}

// ---- closest-hit shading of one level ---------------------------------------------------------------------
// Per hit one group of K = nlights + 1 entries (entry 0 = emission, entry 1 + i = light i): E0 = {colour, intensity
// or a tag}, E1 = {diffuse + specular term, N.L} and the occlusion ray of a point light; the gather kernel multiplies
// them with the attenuation in the reference's order.  tags in E0.w: -1 skip, -2 add E1.xyz as it is.
struct WhLevel {
    PathQueue cur, next;
    const float4* hits;
    ShadowQueue sh;              // occlusion rays + E0 in .contrib
    float4* e1;
    float4* group;               // per hit: {base alpha, is BLEND, slot bits, -}
    float4* levels;              // [level][n_pixels]: {radiance of the level, 1 - alpha of a BLEND hit}
    uint32_t* n_levels;          // per pixel: levels written so far
    uint32_t level;              // = payload depth of the rays of this level
    uint32_t n_pixels;
    uint32_t bound;              // capacity of the queues
};

__global__ __launch_bounds__(256) void k_wh_shade(WhConsts c, WhLevel L, DeviceCounters* ctr)
{
    __shared__ uint32_t s_app[2][2];
    if (threadIdx.x < 4u) s_app[threadIdx.x >> 1][threadIdx.x & 1u] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_cur = ctr->n_cur < L.bound ? ctr->n_cur : L.bound;
    bool is_hit = false;
    uint32_t slot = kInvalidSlot;
    float4 hit = make_float4(-1.f, 0.f, 0.f, 0.f);
    if (i < n_cur) {
        slot = L.cur.state_b[i].x;
        if (slot != kInvalidSlot) {
            hit = L.hits[i];
            if (hit.x < 0.0f) {                                                  // __miss__constant_radiance
                L.levels[(size_t)L.level * L.n_pixels + slot] = make_float4(c.miss_color[0], c.miss_color[1], c.miss_color[2], 0.f);
                L.n_levels[slot] = L.level + 1u;
            } else is_hit = true;
        }
    }
    const uint32_t j = block_append(is_hit, &ctr->n_shadow, s_app[0]);           // group index (scaled to rays by k_wh_scale); one atomic per workgroup
    // ---- surface
    bool spawn = false;
    float4 ro4 = make_float4(0.f, 0.f, 0.f, 0.f), rd4 = ro4;
    if (is_hit) {
        const uint32_t K = c.nlights + 1u;
        const uint32_t prim = f2u(hit.y);
        ro4 = L.cur.ray_o[i]; rd4 = L.cur.ray_d[i];
        const f3 ray_d = mk3(rd4.x, rd4.y, rd4.z);
        WhLocalGeom g;
        local_geometry(c, prim, hit.z, hit.w, g);
        const MiptMaterialData m = c.materials[prim_material(c, prim)];
        float4 bc = make_float4(m.base_color[0] * g.color.x, m.base_color[1] * g.color.y, m.base_color[2] * g.color.z, m.base_color[3] * g.color.w);
        float4 tx;
        if (sample_material_texture(c, m.base_color_tex, g, tx)) {
            bc.x *= det_pow22(tx.x); bc.y *= det_pow22(tx.y); bc.z *= det_pow22(tx.z);      // linearize; not the alpha channel
            bc.w *= tx.w;
        }
        const f3 base = mk3(bc.x, bc.y, bc.z);
        float metallic = m.metallic, roughness = m.roughness;
        float4 mr = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        sample_material_texture(c, m.metallic_roughness_tex, g, mr);                        // (occlusion, roughness, metallic)
        roughness *= mr.y;
        metallic *= mr.z;
        const float F0 = 0.04f;
        const f3 diff_color = (base * (1.0f - F0)) * (1.0f - metallic);
        const f3 f0v = mk3(F0, F0, F0);
        const f3 spec_color = f0v + metallic * (base - f0v);
        const float alpha = roughness * roughness;
        float4 emt = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        sample_material_texture(c, m.emissive_tex, g, emt);
        const f3 em = ld3w(m.emissive_factor) * mk3(emt.x, emt.y, emt.z);
        f3 N = g.N;
        float4 nt;
        if (sample_material_texture(c, m.normal_tex, g, nt)) {
            const float nx = 2.0f * nt.x - 1.0f, ny = 2.0f * nt.y - 1.0f, nz = 2.0f * nt.z - 1.0f;
            const float tx_ = nx * m.normal_tex.rotation[1] + ny * -m.normal_tex.rotation[0];
            const float ty_ = nx * m.normal_tex.rotation[0] + ny * m.normal_tex.rotation[1];
            const WhTexcoord& tc = g.tc[m.normal_tex.texcoord == 1 ? 1 : 0];
            N = normalize((tx_ * normalize(tc.dpdu) + ty_ * normalize(tc.dpdv)) + nz * g.N);
        }
        const f3 P = g.P;
        if (dot(N, ray_d) > 0.0f) N = -N;
        const uint32_t depth = L.level + 1u;
        const size_t base_i = (size_t)j * K;
        const float4 no_ray_o = make_float4(0.f, 0.f, 0.f, 1.0f), no_ray_d = make_float4(0.f, 0.f, 1.f, 0.0f);   // empty interval
        L.sh.ray_o[base_i] = no_ray_o; L.sh.ray_d[base_i] = no_ray_d;
        L.sh.contrib[base_i] = make_float4(0.f, 0.f, 0.f, -2.0f);
        L.e1[base_i] = make_float4(em.x, em.y, em.z, 0.f);
        for (uint32_t li = 0; li < c.nlights; ++li) {
            const MiptLight l = c.lights[li];
            float4 e0 = make_float4(0.f, 0.f, 0.f, -1.0f), e1v = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 ro = no_ray_o, rd = no_ray_d;
            if (l.type == 0) {
                if (depth < kWhMaxTraceDepth) {
                    const f3 toL = ld3w(l.position) - P;
                    const float Ld = length(toL);
                    const float invd = 1.0f / Ld;
                    const f3 Lv = toL * invd;
                    const f3 V = -normalize(ray_d);
                    const f3 H = normalize(Lv + V);
                    const float NdL = dot(N, Lv), NdV = dot(N, V), NdH = dot(N, H), VdH = dot(V, H);
                    if (NdL > 0.0f && NdV > 0.0f) {
                        const f3 F = w_schlick(spec_color, VdH);
                        const float G = w_vis(NdL, NdV, alpha);
                        const float D = w_ggx(NdH, alpha);
                        const f3 diff = ((mk3(1.0f, 1.0f, 1.0f) - F) * diff_color) * (1.0f / 3.14159265358979323846f);
                        const f3 spec = (F * G) * D;
                        const f3 ds = diff + spec;
                        e0 = make_float4(l.color[0], l.color[1], l.color[2], l.intensity);
                        e1v = make_float4(ds.x, ds.y, ds.z, NdL);
                        if (!(l.intensity >= 0.0f)) e0.w = -1.0f;                        // negative / NaN intensities would collide with the tags
                        ro = make_float4(P.x, P.y, P.z, 0.001f);
                        rd = make_float4(Lv.x, Lv.y, Lv.z, Ld - 0.001f);
                    }
                }
            } else if (l.type == 1) {
                const f3 amb = ld3w(l.color) * base;
                e0.w = -2.0f;
                e1v = make_float4(amb.x, amb.y, amb.z, 0.f);
            }
            L.sh.ray_o[base_i + 1u + li] = ro;
            L.sh.ray_d[base_i + 1u + li] = rd;
            L.sh.contrib[base_i + 1u + li] = e0;
            L.e1[base_i + 1u + li] = e1v;
        }
        const bool blend = m.alpha_mode == 2;
        L.group[j] = make_float4(bc.w, blend ? 1.0f : 0.0f, u2f(slot), 0.f);
        L.n_levels[slot] = L.level + 1u;
        spawn = blend && depth < kWhMaxTraceDepth;                                          // continuation ray, whitted.cu:272-283
    }
    const uint32_t jn = block_append(spawn, &ctr->n_next, s_app[1]);
    if (spawn) {
        L.next.ray_o[jn] = make_float4(ro4.x, ro4.y, ro4.z, hit.x);                         // tmin = optixGetRayTmax() = the hit distance
        L.next.ray_d[jn] = make_float4(rd4.x, rd4.y, rd4.z, 1e16f);
        L.next.state_b[jn] = make_uint2(slot, L.level + 1u);
    }
}

__global__ void k_wh_scale(DeviceCounters* ctr, uint32_t K)
{
    if (threadIdx.x == 0) { ctr->bounce = ctr->n_shadow; ctr->n_shadow = ctr->n_shadow * K; }   // bounce := number of groups
}

// adds the entries of a group in the reference's order: result = 0 + emission, then the lights
__global__ __launch_bounds__(1024) void k_wh_gather(uint32_t K, WhLevel L, const uint32_t* occluded, const float* attenuation, DeviceCounters* ctr)
{
    __shared__ uint32_t s_traced;
    if (threadIdx.x == 0u) s_traced = 0u;
    __syncthreads();
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = j < ctr->bounce;
    const size_t base = (size_t)j * K;
    f3 r = mk3(0.0f, 0.0f, 0.0f);
    uint32_t traced = 0;
    for (uint32_t k = 0; mine && k < K; ++k) {
        const float4 e0 = L.sh.contrib[base + k], e1 = L.e1[base + k];
        if (e0.w == -1.0f) continue;
        if (e0.w == -2.0f) { r = r + mk3(e1.x, e1.y, e1.z); continue; }
        ++traced;
        const float att = attenuation ? attenuation[base + k] : (occluded[base + k] ? 0.0f : 1.0f);
        if (att > 0.0f) {
            const f3 lc = ((mk3(e0.x, e0.y, e0.z) * att) * e0.w) * e1.w;
            r = r + lc * mk3(e1.x, e1.y, e1.z);
        }
    }
    if (mine) {
        const float4 gh = L.group[j];
        const uint32_t slot = f2u(gh.z);
        float w = 0.0f;
        if (gh.y != 0.0f) { r = r * gh.x; w = 1.0f - gh.x; }                                // ALPHA_MODE_BLEND: result *= base_color.w
        L.levels[(size_t)L.level * L.n_pixels + slot] = make_float4(r.x, r.y, r.z, w);
    }
    block_add64(&ctr->shadow_rays, traced, &s_traced);                                      // occlusion rays really cast: one atomic per 1024 groups
}

__global__ void k_wh_advance(DeviceCounters* c, uint32_t* host_live)
{
    const uint32_t i = threadIdx.x;
    if (i == 0) {
        if (host_live) *host_live = c->n_next;                  // (pinned host memory: what the next level will find)
        c->closest_rays += c->n_next; c->n_cur = c->n_next; c->n_next = 0; c->n_shadow = 0; c->bounce = 0; c->n_open = 0;
    }
    if (i < kNumSegments) { c->cursor_closest[i * kCursorStride] = 0; c->cursor_shadow[i * kCursorStride] = 0; }
}

// composes the levels back to front (result += alpha_payload.result * (1 - base_color.w), whitted.cu:285), then the
// progressive lerp + sRGB8 of the raygen program (:82-96)
__global__ __launch_bounds__(256) void k_wh_resolve(WhConsts c, const float4* levels, const uint32_t* n_levels, float4* accum, uint32_t* frame)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_pixels = c.width * c.height;
    if (i >= n_pixels) return;
    const uint32_t n = n_levels[i];
    f3 col = mk3(0.0f, 0.0f, 0.0f);
    if (n != 0u) {
        const float4 last = levels[(size_t)(n - 1u) * n_pixels + i];
        col = mk3(last.x, last.y, last.z);
        for (uint32_t l = n - 1u; l-- > 0u;) {
            const float4 a = levels[(size_t)l * n_pixels + i];
            col = mk3(a.x, a.y, a.z) + col * mk3(a.w, a.w, a.w);
        }
    }
    if (c.subframe_index > 0u) {
        const float a = 1.0f / (float)(c.subframe_index + 1u);
        const float4 p = accum[i];
        col = lerp3(mk3(p.x, p.y, p.z), col, a);
    }
    accum[i] = make_float4(col.x, col.y, col.z, 1.0f);
    if (frame) frame[i] = make_color(col);
}

// The launch of the any-hit-program traversal: the entry pool's form when the structure has the mirrors for it (a one-level BVH
// whose packed entries fit, a two-level structure in its arena form), else one ray per thread.  MIPT_WH_ANYHIT=walk forces the
// latter (tests compare the two).  Rays per wave: chosen on the device from the queue's length (ah_rays_per_wave).
// MIPT_WH_ANYHIT = walk: one ray per thread (what structures without the pool's mirrors get); unseeded / seeded: never / wherever the
// record tables exist a fast closest-hit launch in front of the radiance rays (default: over two-level structures, where it gains)
static int wh_anyhit_env()
{
    static const int v = [] { const char* e = std::getenv("MIPT_WH_ANYHIT"); return !e ? 0 : (std::strcmp(e, "walk") == 0 ? 1 : (std::strcmp(e, "unseeded") == 0 ? 2 : (std::strcmp(e, "seeded") == 0 ? 3 : 0))); }();
    return v;
}
static bool wh_trace_on_pool(const DeviceBvh& bvh, const TraceOpts& o)
{
    // (a tree of a few levels -- the alpha fixture's 14 triangles -- is walked faster one ray per thread: a chunk on the pool costs a
    //  handful of rounds whatever the tree, 0.82 against 0.55 ms per 1080p subframe; from 2 k triangles on the pool wins, 0.48 against 0.65)
    // (and the builders' primitive -> record tables: the pool reports a candidate by its key, candidate_record() finds the rest)
    return wh_anyhit_env() != 1 && o.persistent == 3 && bvh.depth > 8u &&
           (bvh.inst ? bvh.inst_fast != 0u && bvh.inst_slots != 0u : (bvh.pool_ok && bvh.tri_byte_off != 0u && bvh.prim_slot != nullptr));
}

template <int MODE>
static void launch_wh_trace(hipStream_t st, const WhConsts& c, const DeviceBvh& bvh, const float4* ray_o, const float4* ray_d, const uint32_t* n_ptr,
                            uint32_t n_max, float4* hits, float* attenuation, const TraceOpts& o, const uint32_t* open_list = nullptr, bool blend = false)
{
    if (n_max == 0u) return;
    static const int e_ppw = [] { const char* e = std::getenv("MIPT_WH_AH_PPW"); return e ? std::atoi(e) : 0; }();
    WhTraceArgs a{};
    a.bvh = bvh; a.ray_o = ray_o; a.ray_d = ray_d; a.n_ptr = n_ptr; a.n_max = n_max; a.hits = hits; a.attenuation = attenuation;
    a.open_list = open_list;
    static const int e_k = [] { const char* e = std::getenv("MIPT_WH_AH_FIRST"); return e ? std::atoi(e) : 0; }();      // (experiments: 1 / kAhLevels everywhere)
    a.first_k = e_k >= 1 && e_k <= kAhLevels ? (uint32_t)e_k : (MODE == 1 && blend ? (uint32_t)kAhLevels : 1u);
    const bool pool = wh_trace_on_pool(bvh, o);
    if (!pool) {
        const int tblocks = o.blocks > 0 ? o.blocks : 4096;
        if (bvh.inst) hipLaunchKernelGGL((k_wh_trace_inst<MODE>), dim3(tblocks), dim3(64), 0, st, c, a);
        else          hipLaunchKernelGGL((k_wh_trace<MODE>), dim3(tblocks), dim3(64), 0, st, c, a);
        return;
    }
    const uint32_t cus = (uint32_t)(o.num_cus > 0 ? o.num_cus : 256);
    a.ppw = e_ppw >= 1 && e_ppw <= 64 ? (uint32_t)e_ppw : 0u; a.cus = cus;
    unsigned long long want = ((unsigned long long)n_max + 3u) / 4u;
    static const int e_grid = [] { const char* e = std::getenv("MIPT_WH_AH_GRID"); return e ? std::atoi(e) : 0; }();      // waves per CU at most (experiments)
    const unsigned long long cap = cus * (unsigned long long)(e_grid >= 1 && e_grid <= 1024 ? e_grid : 256);
    want = want < cus * 16ull ? cus * 16ull : (want > cap ? cap : want);
    const uint32_t g = (uint32_t)want;
    if (bvh.inst)                hipLaunchKernelGGL((k_wh_trace_pool_inst<MODE>), dim3(g), dim3(64), 0, st, c, a);
    else if (bvh.node_fmt == 2u) hipLaunchKernelGGL((k_wh_trace_pool<MODE, 2>), dim3(g), dim3(64), 0, st, c, a);
    else if (bvh.node_fmt == 1u) hipLaunchKernelGGL((k_wh_trace_pool<MODE, 1>), dim3(g), dim3(64), 0, st, c, a);
    else                         hipLaunchKernelGGL((k_wh_trace_pool<MODE, 0>), dim3(g), dim3(64), 0, st, c, a);
}

void whitted_run(hipStream_t st, const MiptWhittedLaunchParams& p, uint32_t n_paths, const DeviceBvh& bvh, const WhGeometry& geo,
                 const MiptMaterialData* d_materials, uint32_t n_materials, const WhTexture* d_textures, uint32_t n_textures,
                 const WhBuffers& B, DeviceCounters* ctr, const TraceOpts& opts)
{
    WhConsts c{};
    c.width = p.width; c.height = p.height; c.subframe_index = p.subframe_index; c.n_paths = n_paths;
    for (int k = 0; k < 3; ++k) { c.eye[k] = p.eye[k]; c.U[k] = p.U[k]; c.V[k] = p.V[k]; c.W[k] = p.W[k]; c.miss_color[k] = p.miss_color[k]; }
    c.lights = reinterpret_cast<const MiptLight*>(p.lights.data); c.nlights = p.lights.count;
    c.geo = geo; c.materials = d_materials; c.nmaterials = n_materials; c.textures = d_textures; c.ntextures = n_textures;
    const uint32_t K = c.nlights + 1u;
    const uint32_t blocks = (n_paths + 255u) / 256u;
    const uint32_t n_pixels = p.width * p.height;
    launch_begin_sample(st, ctr, n_paths, n_pixels);
    if (B.any_hit_programs) (void)hipMemsetAsync(&ctr->n_open, 0, sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_wh_raygen, dim3(blocks), dim3(256), 0, st, c, B.q[0], B.n_levels);
    for (uint32_t level = 0; level < B.max_levels; ++level) {
        const PathQueue cur = B.q[level & 1u], next = B.q[(level & 1u) ^ 1u];
        // (any-hit programs: the fast launch first when the structure's builders kept the geometry flags per primitive -- anyhit_seed)
        const bool seeded = B.any_hit_programs && wh_anyhit_env() != 2 && wh_trace_on_pool(bvh, opts) &&
                            (bvh.inst != nullptr || wh_anyhit_env() == 3);
        if (B.any_hit_programs && !seeded) launch_wh_trace<0>(st, c, bvh, cur.ray_o, cur.ray_d, &ctr->n_cur, n_paths, B.hits, nullptr, opts);
        else
            launch_trace_closest(st, bvh, cur.ray_o, cur.ray_d, 1, &ctr->n_cur, n_paths, B.hits, ctr, opts,
                                 MIPT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES);          // traceRadiance, whitted_cuda.h:110
        if (seeded) {
            uint32_t* const open_list = reinterpret_cast<uint32_t*>(B.group);          // (the group records are written by the shade launch that follows)
            hipLaunchKernelGGL(k_wh_seed, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, st, c, bvh, cur.ray_o, cur.ray_d, &ctr->n_cur, n_paths, (const float4*)B.hits, open_list, ctr);
            launch_wh_trace<0>(st, c, bvh, cur.ray_o, cur.ray_d, &ctr->n_open, n_paths, B.hits, nullptr, opts, open_list);
        }
        WhLevel L{};
        L.cur = cur; L.next = next; L.hits = B.hits; L.sh = B.sh; L.e1 = B.e1; L.group = B.group; L.levels = B.levels; L.n_levels = B.n_levels;
        L.level = level; L.n_pixels = n_pixels; L.bound = n_paths;
        hipLaunchKernelGGL(k_wh_shade, dim3(blocks), dim3(256), 0, st, c, L, ctr);
        hipLaunchKernelGGL(k_wh_scale, dim3(1), dim3(64), 0, st, ctr, K);
        if (B.any_hit_programs) launch_wh_trace<1>(st, c, bvh, B.sh.ray_o, B.sh.ray_d, &ctr->n_shadow, n_paths * K, nullptr, (float*)B.flags, opts, nullptr, B.max_levels > 1u);
        else
            launch_trace_any(st, bvh, B.sh.ray_o, B.sh.ray_d, 1, &ctr->n_shadow, n_paths * K, B.flags, nullptr, nullptr, ctr, opts);
        hipLaunchKernelGGL(k_wh_gather, dim3((blocks + 3u) / 4u), dim3(1024), 0, st, K, L, (const uint32_t*)B.flags,
                           B.any_hit_programs ? (const float*)B.flags : (const float*)nullptr, ctr);
        // Levels beyond the deepest chain of BLEND surfaces find empty queues and cost seven launches each: the count a level leaves behind
        // goes to a pinned ring, and the host reads it two levels later (the stream never drains) and stops behind the first empty level
        // (run_wavefront's scheme; the 14-triangle alpha fixture, 8 levels of which 3 carry rays: 0.56 -> 0.52 ms per 1080p subframe).
        constexpr uint32_t kRing = 4u, kLag = 2u;
        const bool lagged = B.live_ring != nullptr && B.ring_events != nullptr && B.max_levels > kLag + 1u;
        hipLaunchKernelGGL(k_wh_advance, dim3(1), dim3(64), 0, st, ctr, lagged ? B.live_ring + level % kRing : (uint32_t*)nullptr);
        if (lagged) {
            (void)hipEventRecord((hipEvent_t)B.ring_events[level % kRing], st);
            if (level >= kLag) {
                const uint32_t old = (level - kLag) % kRing;
                if (hipEventSynchronize((hipEvent_t)B.ring_events[old]) == hipSuccess && B.live_ring[old] == 0u) break;
            }
        }
    }
    const uint32_t pb = (n_pixels + 255u) / 256u;
    hipLaunchKernelGGL(k_wh_resolve, dim3(pb), dim3(256), 0, st, c, (const float4*)B.levels, (const uint32_t*)B.n_levels, (float4*)p.accum_buffer,
                       (uint32_t*)p.frame_buffer);
}

}  // namespace mipt
