// whitted_kernels.hip -- the glTF / PBR programs of SDK/cuda/whitted.cu as wavefront stages:
//   k_wh_raygen  <- __raygen__pinhole            (whitted.cu:44-97: pixel centre at subframe 0, jitter later)
//   k_wh_shade   <- __closesthit__radiance        (:149-289) up to its occlusion queries, __miss__constant_radiance (:139-142),
//                   getLocalGeometry              (cuda/LocalGeometry.h:59-127) on world-space triangles
//   k_wh_gather  <- the `result +=` lines once the occlusion flags are known, in light order
//   k_wh_resolve <- progressive lerp + make_color (:82-96)
// Subset: opaque, untextured materials (base colour / metallic / roughness / emissive factors), point and
// ambient lights.  schlick's powf(x, 5) is x^2 * x^2 * x (detmath: no libm transcendentals).
#include "kernels.h"
#include "detmath.h"
#include "wave_util.h"

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

__global__ __launch_bounds__(256) void k_wh_raygen(WhConsts c, PathQueue q)
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
    q.state_b[i] = make_uint2(py * c.width + px, 0u);
}

// per primary hit one group of K = nlights + 1 entries: entry 0 = emission (no ray), entry 1 + i = light i
__global__ __launch_bounds__(256) void k_wh_shade(WhConsts c, PathQueue cur, const float4* hits, ShadowQueue sh, float4* result,
                                                  DeviceCounters* ctr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool is_hit = false;
    uint32_t slot = kInvalidSlot;
    float4 hit = make_float4(-1.f, 0.f, 0.f, 0.f);
    if (i < c.n_paths) {
        slot = cur.state_b[i].x;
        if (slot != kInvalidSlot) {
            hit = hits[i];
            if (hit.x < 0.0f) result[slot] = make_float4(c.miss_color[0], c.miss_color[1], c.miss_color[2], 0.f);
            else is_hit = true;
        }
    }
    const uint32_t j = wave_append(is_hit, &ctr->n_next);
    if (!is_hit) return;
    const uint32_t K = c.nlights + 1u;
    const uint32_t prim = f2u(hit.y);
    const float4 rd4 = cur.ray_d[i];
    const f3 ray_d = mk3(rd4.x, rd4.y, rd4.z);
    // ---- getLocalGeometry (cuda/LocalGeometry.h:59-176) on world-space corners
    WhLocalGeom g;
    const float4 a0 = c.geo.vertices4[(size_t)prim * 3u], a1 = c.geo.vertices4[(size_t)prim * 3u + 1u], a2 = c.geo.vertices4[(size_t)prim * 3u + 2u];
    const f3 P0 = mk3(a0.x, a0.y, a0.z), P1 = mk3(a1.x, a1.y, a1.z), P2 = mk3(a2.x, a2.y, a2.z);
    const float bu = hit.z, bv = hit.w, b0 = 1.0f - hit.z - hit.w;
    g.P = b0 * P0 + bu * P1 + bv * P2;
    g.color = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (c.geo.colors4 && (!c.geo.color_present || c.geo.color_present[prim])) {
        const float4* cc = c.geo.colors4 + (size_t)prim * 3u;
        const float4 c0 = cc[0], c1 = cc[1], c2 = cc[2];
        g.color = make_float4((b0 * c0.x + bu * c1.x) + bv * c2.x, (b0 * c0.y + bu * c1.y) + bv * c2.y,
                              (b0 * c0.z + bu * c1.z) + bv * c2.z, (b0 * c0.w + bu * c1.w) + bv * c2.w);
    }
    g.N = normalize(cross(P1 - P0, P2 - P0));
    if (c.geo.normals4) {
        const float4 n0 = c.geo.normals4[(size_t)prim * 3u], n1 = c.geo.normals4[(size_t)prim * 3u + 1u], n2 = c.geo.normals4[(size_t)prim * 3u + 2u];
        if (n0.w != 0.0f) g.N = normalize(b0 * mk3(n0.x, n0.y, n0.z) + bu * mk3(n1.x, n1.y, n1.z) + bv * mk3(n2.x, n2.y, n2.z));
    }
    const f3 dp1 = P0 - P2, dp2 = P1 - P2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        WhTexcoord& tc = g.tc[j];
        if (c.geo.texcoords[j] && (!c.geo.texcoord_present[j] || c.geo.texcoord_present[j][prim])) {
            const float2* uv = c.geo.texcoords[j] + (size_t)prim * 3u;
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
    // ---- material (cuda/whitted.cu:155-215)
    uint32_t mi = c.geo.material_index ? c.geo.material_index[prim] : 0u;
    if (mi >= c.nmaterials) mi = c.nmaterials - 1u;
    const MiptMaterialData m = c.materials[mi];
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
    const size_t base_i = (size_t)j * K;
    sh.ray_o[base_i] = make_float4(0.f, 0.f, 0.f, 1.0f);                 // empty interval: never occluded
    sh.ray_d[base_i] = make_float4(0.f, 0.f, 1.f, 0.0f);
    sh.contrib[base_i] = make_float4(em.x, em.y, em.z, u2f(slot));
    for (uint32_t li = 0; li < c.nlights; ++li) {
        const MiptLight l = c.lights[li];
        f3 contrib = mk3(0.0f, 0.0f, 0.0f);
        float4 ro = make_float4(0.f, 0.f, 0.f, 1.0f), rd = make_float4(0.f, 0.f, 1.f, 0.0f);
        if (l.type == 0) {
            const f3 toL = ld3w(l.position) - P;
            const float Ld = length(toL);
            const float invd = 1.0f / Ld;
            const f3 L = toL * invd;
            const f3 V = -normalize(ray_d);
            const f3 H = normalize(L + V);
            const float NdL = dot(N, L), NdV = dot(N, V), NdH = dot(N, H), VdH = dot(V, H);
            if (NdL > 0.0f && NdV > 0.0f) {
                const f3 F = w_schlick(spec_color, VdH);
                const float G = w_vis(NdL, NdV, alpha);
                const float D = w_ggx(NdH, alpha);
                const f3 diff = ((mk3(1.0f, 1.0f, 1.0f) - F) * diff_color) * (1.0f / 3.14159265358979323846f);
                const f3 spec = (F * G) * D;
                const f3 lc = ((ld3w(l.color) * 1.0f) * l.intensity) * NdL;
                contrib = lc * (diff + spec);
                ro = make_float4(P.x, P.y, P.z, 0.001f);
                rd = make_float4(L.x, L.y, L.z, Ld - 0.001f);
            }
        } else if (l.type == 1) {
            contrib = ld3w(l.color) * base;
        }
        sh.ray_o[base_i + 1u + li] = ro;
        sh.ray_d[base_i + 1u + li] = rd;
        sh.contrib[base_i + 1u + li] = make_float4(contrib.x, contrib.y, contrib.z, u2f(slot));
    }
}

__global__ void k_wh_scale(DeviceCounters* ctr, uint32_t K)
{
    if (threadIdx.x == 0) ctr->n_shadow = ctr->n_next * K;
}

__global__ __launch_bounds__(256) void k_wh_gather(uint32_t K, const float4* ray_d, const float4* contrib, const uint32_t* occluded,
                                                   float4* result, DeviceCounters* ctr)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ctr->n_next) return;
    const size_t base = (size_t)j * K;
    f3 r = mk3(0.0f, 0.0f, 0.0f);
    uint32_t slot = 0;
    uint32_t traced = 0;
    for (uint32_t k = 0; k < K; ++k) {
        const float4 cc = contrib[base + k];
        slot = f2u(cc.w);
        if (!occluded[base + k]) r = r + mk3(cc.x, cc.y, cc.z);
        traced += ray_d[base + k].w > 0.0f ? 1u : 0u;
    }
    result[slot] = make_float4(r.x, r.y, r.z, 0.f);
    if (traced) atomicAdd(&ctr->shadow_rays, (unsigned long long)traced);     // occlusion rays really cast
}

__global__ __launch_bounds__(256) void k_wh_resolve(WhConsts c, const float4* result, float4* accum, uint32_t* frame)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.width * c.height) return;
    const float4 r = result[i];
    f3 col = mk3(r.x, r.y, r.z);
    if (c.subframe_index > 0u) {
        const float a = 1.0f / (float)(c.subframe_index + 1u);
        const float4 p = accum[i];
        col = lerp3(mk3(p.x, p.y, p.z), col, a);
    }
    accum[i] = make_float4(col.x, col.y, col.z, 1.0f);
    if (frame) frame[i] = make_color(col);
}

void whitted_run(hipStream_t st, const MiptWhittedLaunchParams& p, uint32_t n_paths, const DeviceBvh& bvh, const WhGeometry& geo,
                 const MiptMaterialData* d_materials, uint32_t n_materials, const WhTexture* d_textures, uint32_t n_textures,
                 PathQueue q, float4* hits, ShadowQueue sh, uint32_t* occluded, float4* result, DeviceCounters* ctr, const TraceOpts& opts)
{
    WhConsts c{};
    c.width = p.width; c.height = p.height; c.subframe_index = p.subframe_index; c.n_paths = n_paths;
    for (int k = 0; k < 3; ++k) { c.eye[k] = p.eye[k]; c.U[k] = p.U[k]; c.V[k] = p.V[k]; c.W[k] = p.W[k]; c.miss_color[k] = p.miss_color[k]; }
    c.lights = reinterpret_cast<const MiptLight*>(p.lights.data); c.nlights = p.lights.count;
    c.geo = geo; c.materials = d_materials; c.nmaterials = n_materials; c.textures = d_textures; c.ntextures = n_textures;
    const uint32_t K = c.nlights + 1u;
    const uint32_t blocks = (n_paths + 255u) / 256u;
    launch_begin_sample(st, ctr, n_paths, p.width * p.height);
    hipLaunchKernelGGL(k_wh_raygen, dim3(blocks), dim3(256), 0, st, c, q);
    launch_trace_closest(st, bvh, q.ray_o, q.ray_d, 1, &ctr->n_cur, n_paths, hits, ctr, opts);
    hipLaunchKernelGGL(k_wh_shade, dim3(blocks), dim3(256), 0, st, c, q, hits, sh, result, ctr);
    hipLaunchKernelGGL(k_wh_scale, dim3(1), dim3(64), 0, st, ctr, K);
    launch_trace_any(st, bvh, sh.ray_o, sh.ray_d, 1, &ctr->n_shadow, n_paths * K, occluded, nullptr, nullptr, ctr, opts);
    hipLaunchKernelGGL(k_wh_gather, dim3(blocks), dim3(256), 0, st, K, sh.ray_d, sh.contrib, occluded, result, ctr);
    const uint32_t pb = (p.width * p.height + 255u) / 256u;
    hipLaunchKernelGGL(k_wh_resolve, dim3(pb), dim3(256), 0, st, c, result, (float4*)p.accum_buffer, (uint32_t*)p.frame_buffer);
}

}  // namespace mipt
