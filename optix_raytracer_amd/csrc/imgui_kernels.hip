// imgui_kernels.hip -- the programs of SDK/imgui_test (imgui_test/optixTriangle.cu:103-268) as wavefront
// stages around the shared traversal kernels:
//   k_imgui_raygen  <- __raygen__rg prologue + Camera::compute_ray (camera.h:127-144)
//   k_imgui_shade   <- __closesthit__ch (:174-268) up to its occlusion queries, __miss__ms (:153-171)
//   k_imgui_gather  <- the `result +=` lines of __closesthit__ch once the occlusion flags are known
//   k_imgui_resolve <- film / image update of __raygen__rg (:143-149)
// A primary hit issues nlights shadow rays (tmin .01, tmax 1.0 in units of the UN-normalised light vector)
// and one cosine-sampled probe ray of which only hit / no hit is used; all of them go through the any-hit
// kernel, and their contributions are added per path in the reference's order by k_imgui_gather.
#include "kernels.h"
#include "detmath.h"
#include "wave_util.h"

namespace mipt {

struct ImguiConsts {
    uint32_t width, height, samples_per_frame, dt, dirty, n_paths;
    const MiptImguiCamera* camera;
    const MiptImguiLight*  lights;    int nlights;
    const float*           materials; int nmaterials;
    const int*             mat_indices; int nmat_indices;
    const float*           normals;
};

__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

__device__ __forceinline__ bool path_pixel(uint32_t i, uint32_t width, uint32_t height, int& px, int& py)
{
    const uint32_t tiles_x = (width + 7u) >> 3;
    const uint32_t tile = i >> 6, in = i & 63u;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    px = (int)(tx * 8u + (in & 7u));
    py = (int)(ty * 8u + (in >> 3));
    return (uint32_t)px < width && (uint32_t)py < height;
}

__global__ __launch_bounds__(256) void k_imgui_raygen(ImguiConsts c, uint32_t sample, PathQueue q, uint32_t* rg_seed, float4* result)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_paths) return;
    int px, py;
    if (!path_pixel(i, c.width, c.height, px, py)) {
        q.ray_o[i] = make_float4(0.f, 0.f, 0.f, 1.0f);
        q.ray_d[i] = make_float4(0.f, 0.f, 1.f, 0.0f);
        q.state_b[i] = make_uint2(kInvalidSlot, 0u);
        return;
    }
    const uint32_t slot = (uint32_t)py * c.width + (uint32_t)px;
    uint32_t seed = sample == 0u ? tea4((uint32_t)px + c.width * (uint32_t)py, c.dt) : rg_seed[slot];
    const MiptImguiCamera cam = *c.camera;
    float dx = 2.0f * ((float)px / (float)c.width) - 1.0f;
    float dy = 2.0f * ((float)py / (float)c.height) - 1.0f;
    const f3 u = ld3(cam.u), v = ld3(cam.v), w = ld3(cam.w), eye = ld3(cam.eye);
    f3 origin, direction;
    if (cam.ortho) {
        direction = normalize(dx * u + dy * v + w);
        origin    = eye + dx * u + dy * v;
    } else {
        const float ax = (rnd(seed) - 0.5f) * cam.aperture;
        const float ay = (rnd(seed) - 0.5f) * cam.aperture;
        dx = dx - ax; dy = dy - ay;
        direction = normalize(dx * u + dy * v + w);
        origin    = eye + ax * u + ay * v;
    }
    rg_seed[slot] = seed;
    q.ray_o[i] = make_float4(origin.x, origin.y, origin.z, 0.0f);
    q.ray_d[i] = make_float4(direction.x, direction.y, direction.z, 1e16f);
    q.state_b[i] = make_uint2(slot, 0u);
    if (sample == 0u) result[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ f3 light_wi(const MiptImguiLight& l, f3 p, uint32_t& seed)
{
    if (l.tag == 0) return ld3(l.a) - p;
    if (l.tag == 1) {
        const float j = l.b[3];
        const float jx = j * rnd(seed), jy = j * rnd(seed), jz = j * rnd(seed);
        return ld3(l.a) + mk3(jx, jy, jz);
    }
    const float r = l.b[0];
    const float jx = r * rnd(seed), jy = r * rnd(seed), jz = r * rnd(seed);
    return ld3(l.a) + mk3(jx, jy, jz) - p;
}
__device__ __forceinline__ f3 light_lumi(const MiptImguiLight& l)
{
    return l.tag == 2 ? mk3(l.b[1], l.b[2], l.b[3]) : mk3(l.b[0], l.b[1], l.b[2]);
}

__device__ __forceinline__ void cosine_sample(float u1, float u2, f3& p)
{
    const float r = __builtin_sqrtf(u1);
    const float phi = (2.0f * 3.14159265358979323846f) * u2;
    float sn, cs;
    det_sincos(phi, sn, cs);
    p.x = r * cs; p.y = r * sn;
    p.z = __builtin_sqrtf(__builtin_fmaxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
}
__device__ __forceinline__ f3 onb_inverse(f3 n, f3 p)
{
    f3 b = __builtin_fabsf(n.x) > __builtin_fabsf(n.z) ? mk3(-n.y, n.x, 0.0f) : mk3(0.0f, -n.z, n.y);
    b = normalize(b);
    const f3 t = cross(b, n);
    return p.x * t + p.y * b + p.z * n;
}

// One entry group of K = nlights + 1 occlusion rays per primary hit, group j at [j*K, (j+1)*K).
__global__ __launch_bounds__(256) void k_imgui_shade(ImguiConsts c, PathQueue cur, const float4* hits, ShadowQueue sh,
                                                     float4* result, DeviceCounters* ctr)
{
    __shared__ uint32_t s_app[2];
    if (threadIdx.x < 2u) s_app[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;     // grid covers n_paths rounded up to 256
    bool is_hit = false;
    uint32_t slot = kInvalidSlot;
    float4 hit = make_float4(-1.f, 0.f, 0.f, 0.f);
    if (i < c.n_paths) {
        slot = cur.state_b[i].x;
        if (slot != kInvalidSlot) {
            hit = hits[i];
            if (hit.x < 0.0f) {
                const float4 rd = cur.ray_d[i];                   // __miss__ms: direction * 0.5 + 0.5
                float4 r = result[slot];
                r.x += rd.x * 0.5f + 0.5f; r.y += rd.y * 0.5f + 0.5f; r.z += rd.z * 0.5f + 0.5f;
                result[slot] = r;
            } else is_hit = true;
        }
    }
    // group index: one atomic per workgroup (a returning atomic per wave on the one counter word serialises: wave_util.h)
    const uint32_t j = block_append(is_hit, &ctr->n_next, s_app);
    if (!is_hit) return;
    const uint32_t K = (uint32_t)c.nlights + 1u;
    const uint32_t prim = f2u(hit.y);
    const float4 ro4 = cur.ray_o[i], rd4 = cur.ray_d[i];
    const f3 ray_o = mk3(ro4.x, ro4.y, ro4.z), ray_d = mk3(rd4.x, rd4.y, rd4.z);
    const float* nb = c.normals + (size_t)prim * 9u;
    const f3 n0 = ld3(nb), n1 = ld3(nb + 3), n2 = ld3(nb + 6);
    const f3 normal = hit.z * n1 + hit.w * n2 + (1.0f - hit.z - hit.w) * n0;
    const f3 P = ray_o + hit.x * ray_d + normal * 0.0001f;
    const uint32_t px = slot % c.width, py = slot / c.width;
    uint32_t seed = tea4(px + c.width * py, c.dt);               // re-seeded identically for every sample (:201)
    int mi = (prim < (uint32_t)c.nmat_indices) ? c.mat_indices[prim] : 0;
    if (mi < 0 || mi >= c.nmaterials) mi = 0;
    const f3 color = ld3(c.materials + 3 * mi);
    const size_t base = (size_t)j * K;
    for (int li = 0; li < c.nlights; ++li) {
        const MiptImguiLight l = c.lights[li];
        const f3 wi = light_wi(l, P, seed);
        const float ndotwi = dot(normal, wi);
        f3 contrib = (color * light_lumi(l)) * ndotwi;
        if (ndotwi < 0.0f) contrib = mk3(0.0f, 0.0f, 0.0f);       // adds shadow_color = 0 whatever the query says
        sh.ray_o[base + li] = make_float4(P.x, P.y, P.z, 0.01f);
        sh.ray_d[base + li] = make_float4(wi.x, wi.y, wi.z, 1.0f);
        sh.contrib[base + li] = make_float4(contrib.x, contrib.y, contrib.z, u2f(slot));
    }
    const float u1 = rnd(seed), u2 = rnd(seed);
    f3 out;
    cosine_sample(u1, u2, out);
    out = onb_inverse(normal, out);
    const f3 amb = mk3(0.01f, 0.01f, 0.01f) * color;
    sh.ray_o[base + K - 1u] = make_float4(P.x, P.y, P.z, 0.01f);
    sh.ray_d[base + K - 1u] = make_float4(out.x, out.y, out.z, 1e16f);
    sh.contrib[base + K - 1u] = make_float4(amb.x, amb.y, amb.z, u2f(slot));
}

__global__ void k_imgui_scale(DeviceCounters* ctr, uint32_t K)
{
    if (threadIdx.x == 0) ctr->n_shadow = ctr->n_next * K;
}

__global__ void k_imgui_end_sample(DeviceCounters* ctr)
{
    if (threadIdx.x == 0) ctr->shadow_rays += ctr->n_shadow;
}

__global__ __launch_bounds__(256) void k_imgui_gather(uint32_t K, const float4* contrib, const uint32_t* occluded, float4* result,
                                                      DeviceCounters* ctr)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ctr->n_next) return;
    const size_t base = (size_t)j * K;
    f3 payload = mk3(0.0f, 0.0f, 0.0f);
    uint32_t slot = 0;
    for (uint32_t k = 0; k + 1u < K; ++k) {
        const float4 cc = contrib[base + k];
        slot = f2u(cc.w);
        if (!occluded[base + k]) payload = payload + mk3(cc.x, cc.y, cc.z);
        else                     payload = payload + mk3(0.0f, 0.0f, 0.0f);
    }
    const float4 ca = contrib[base + K - 1u];
    slot = f2u(ca.w);
    payload = payload + (occluded[base + K - 1u] ? mk3(0.01f, 0.01f, 0.01f) : mk3(ca.x, ca.y, ca.z));
    float4 r = result[slot];
    r.x += payload.x; r.y += payload.y; r.z += payload.z;
    result[slot] = r;
}

__global__ __launch_bounds__(256) void k_imgui_resolve(ImguiConsts c, const float4* result, float* film, uint32_t* image)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.width * c.height) return;
    const float4 r = result[i];
    f3 fl = mk3(r.x, r.y, r.z);
    if (!c.dirty) fl = ld3(film + (size_t)i * 3u) + fl;
    film[(size_t)i * 3u] = fl.x; film[(size_t)i * 3u + 1u] = fl.y; film[(size_t)i * 3u + 2u] = fl.z;
    if (image) {
        const float inv = 1.0f / (float)c.dt;
        image[i] = make_color(fl * inv);
    }
}

// ---- host side ----
void imgui_run(hipStream_t st, const MiptImguiParams& p, uint32_t n_paths, const DeviceBvh& bvh, PathQueue q, float4* hits,
               ShadowQueue sh, uint32_t* occluded, float4* result, uint32_t* rg_seed, DeviceCounters* ctr, const TraceOpts& opts)
{
    ImguiConsts c{};
    c.width = p.image_width; c.height = p.image_height; c.samples_per_frame = p.samples_per_frame; c.dt = p.dt;
    c.dirty = p.dirty ? 1u : 0u; c.n_paths = n_paths;
    c.camera = p.camera; c.lights = p.lights; c.nlights = p.nlights; c.materials = p.materials; c.nmaterials = p.nmaterials;
    c.mat_indices = p.mat_indices; c.nmat_indices = p.nmat_indices; c.normals = p.normals;
    const uint32_t K = (uint32_t)p.nlights + 1u;
    const uint32_t blocks = (n_paths + 255u) / 256u;
    for (uint32_t s = 0; s < p.samples_per_frame; ++s) {
        launch_begin_sample(st, ctr, n_paths, p.image_width * p.image_height);
        hipLaunchKernelGGL(k_imgui_raygen, dim3(blocks), dim3(256), 0, st, c, s, q, rg_seed, result);
        launch_trace_closest(st, bvh, q.ray_o, q.ray_d, 1, &ctr->n_cur, n_paths, hits, ctr, opts);
        hipLaunchKernelGGL(k_imgui_shade, dim3(blocks), dim3(256), 0, st, c, q, hits, sh, result, ctr);
        hipLaunchKernelGGL(k_imgui_scale, dim3(1), dim3(64), 0, st, ctr, K);
        launch_trace_any(st, bvh, sh.ray_o, sh.ray_d, 1, &ctr->n_shadow, n_paths * K, occluded, nullptr, nullptr, ctr, opts);
        hipLaunchKernelGGL(k_imgui_gather, dim3(blocks), dim3(256), 0, st, K, sh.contrib, occluded, result, ctr);
        hipLaunchKernelGGL(k_imgui_end_sample, dim3(1), dim3(64), 0, st, ctr);
    }
    const uint32_t pb = (p.image_width * p.image_height + 255u) / 256u;
    hipLaunchKernelGGL(k_imgui_resolve, dim3(pb), dim3(256), 0, st, c, result, p.film, (uint32_t*)p.image);
}

}  // namespace mipt
