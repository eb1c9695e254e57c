// ref_kat.cpp -- known-answer-vector generator (test infrastructure).
//
// This file is OURS; it #includes the reference's own host-callable headers
// straight from /root/reference (never copied) and prints, as JSON, what the
// reference's code computes for a fixed list of inputs:
//   SDK/cuda/random.h          tea<4>, lcg, rnd
//   SDK/sutil/vec_math.h       float3 helpers
//   SDK/cuda/helpers.h         make_color / toSRGB / quantizeUnsigned8Bits
//   SDK/sutil/Camera.{h,cpp}   Camera::UVWFrame
//   SDK/sutil/WorkDistribution.h  StaticWorkDistribution
//   SDK/optixPathTracer/optixPathTracer.h, SDK/cuda/whitted.h, BufferView.h, Light.h, MaterialData.h,
//   include/optix_types.h      sizeof / offsetof of every struct whose layout the C ABI keeps ("layouts")
// The CUDA vector types come from the cuda_runtime headers that the triton
// wheel ships in this image.  Output -> tests/golden/ref_kat.json (committed),
// which pins oracle/pt_oracle.c.  Built only where /root/reference exists
// (oracle/Makefile target `ref`), binary lands in oracle/_ref/.
#include <cuda_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

#include <sutil/vec_math.h>
#include <cuda/random.h>
#include <cuda/helpers.h>
#include <sutil/Camera.h>
#include <sutil/WorkDistribution.h>
#include <cstddef>
#include <optix_types.h>
#include <cuda/whitted.h>
namespace pt {                         // optixPathTracer.h and whitted.h both define Params-like names at file scope
#include <optixPathTracer/optixPathTracer.h>
}

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

static void pf3(const char* name, float3 v, bool comma = true)
{
    std::printf("\"%s\": [%u, %u, %u]%s", name, bits(v.x), bits(v.y), bits(v.z), comma ? ", " : "");
}

int main()
{
    std::printf("{\n");

    // ---- tea<4> + rnd streams -------------------------------------------------
    std::printf("\"tea4\": [\n");
    const uint32_t tea_in[][2] = { {0,0}, {1,0}, {2567,0}, {0,1}, {262143,0}, {2073599,0}, {2073599,7},
                                   {123456789u, 0x5EEDu}, {0xffffffffu, 0xffffffffu}, {999999, 0x5EED} };
    const int n_tea = sizeof(tea_in) / sizeof(tea_in[0]);
    for (int i = 0; i < n_tea; ++i) {
        uint32_t s = tea<4>(tea_in[i][0], tea_in[i][1]);
        uint32_t seed = s;
        std::printf("  {\"v0\": %u, \"v1\": %u, \"out\": %u, \"rnd_bits\": [", tea_in[i][0], tea_in[i][1], s);
        for (int k = 0; k < 6; ++k) std::printf("%u%s", bits(rnd(seed)), k < 5 ? ", " : "");
        std::printf("], \"seed_after\": %u}%s\n", seed, i < n_tea - 1 ? "," : "");
    }
    std::printf("],\n");

    // ---- make_color -------------------------------------------------------------
    std::printf("\"make_color\": [\n");
    std::vector<float3> cols = { {0.18f,0.09f,0.045f}, {0.5f,0.25f,0.125f}, {2.f,1.f,0.5f}, {0.f,0.0031308f,0.0031307f},
                                 {-1.f, 1e-8f, 0.999999f}, {0.0031309f, 0.01f, 0.02f} };
    {
        uint32_t seed = 12345u;
        for (int i = 0; i < 250; ++i) {
            float a = rnd(seed), b = rnd(seed), c = rnd(seed);
            float sc = (i % 3 == 0) ? 0.02f : (i % 3 == 1 ? 0.3f : 1.2f);
            cols.push_back(make_float3(a * sc, b * sc, c * sc));
        }
    }
    for (size_t i = 0; i < cols.size(); ++i) {
        uchar4 q = make_color(cols[i]);
        std::printf("  {\"c\": [%u, %u, %u], \"rgba\": [%u, %u, %u, %u]}%s\n", bits(cols[i].x), bits(cols[i].y),
                    bits(cols[i].z), q.x, q.y, q.z, q.w, i + 1 < cols.size() ? "," : "");
    }
    std::printf("],\n");

    // ---- Camera::UVWFrame ---------------------------------------------------------
    std::printf("\"camera\": [\n");
    struct Cam { float3 eye, lookat, up; float fov, aspect, fod; };
    const Cam cams[] = {
        { {278.f,273.f,-900.f}, {278.f,273.f,330.f}, {0.f,1.f,0.f}, 35.f, 1.0f, 1.0f },
        { {278.f,273.f,-900.f}, {278.f,273.f,330.f}, {0.f,1.f,0.f}, 35.f, 1920.f/1080.f, 1.0f },
        { {0.5f,0.5f,-1.5f}, {0.5f,0.5f,0.5f}, {0.f,1.f,0.f}, 35.f, 1920.f/1080.f, 1.0f },
        { {0.f,1.f,-10.f}, {0.f,0.1f,0.f}, {0.f,7.3e-6f,1.f}, 45.f, 1.5f, 1.0f },
        { {4.0011301f,4.63263988f,-4.31078005f}, {0.134406984f,0.869496822f,-0.037014991f},
          {-0.00425204914f,0.00783693418f,0.00452797301f}, 37.8491669f, 1920.f/1080.f, 1.0f },
        { {1.f,2.f,3.f}, {-4.f,0.5f,9.f}, {0.1f,0.9f,0.2f}, 60.f, 0.75f, 2.5f },
    };
    const int n_cam = sizeof(cams) / sizeof(cams[0]);
    for (int i = 0; i < n_cam; ++i) {
        sutil::Camera cam(cams[i].eye, cams[i].lookat, cams[i].up, cams[i].fov, cams[i].aspect);
        cam.setFocalLength(cams[i].fod);
        float3 U, V, W;
        cam.UVWFrame(U, V, W);
        std::printf("  {");
        pf3("eye", cams[i].eye); pf3("lookat", cams[i].lookat); pf3("up", cams[i].up);
        std::printf("\"fovY\": %u, \"aspect\": %u, \"fod\": %u, ", bits(cams[i].fov), bits(cams[i].aspect), bits(cams[i].fod));
        pf3("U", U); pf3("V", V); pf3("W", W, false);
        std::printf("}%s\n", i < n_cam - 1 ? "," : "");
    }
    std::printf("],\n");

    // ---- float3 helpers -----------------------------------------------------------
    std::printf("\"vec\": [\n");
    {
        uint32_t seed = 777u;
        const int n = 64;
        for (int i = 0; i < n; ++i) {
            float3 a = make_float3(rnd(seed) * 4.f - 2.f, rnd(seed) * 4.f - 2.f, rnd(seed) * 4.f - 2.f);
            float3 b = make_float3(rnd(seed) * 600.f - 300.f, rnd(seed) * 600.f, rnd(seed) - 0.5f);
            float  t = rnd(seed);
            float3 nrm = normalize(a);
            float3 crs = cross(a, b);
            float3 ff  = faceforward(nrm, -b, nrm);
            float3 lrp = lerp(a, b, t);
            float3 dv  = a / t;
            std::printf("  {");
            pf3("a", a); pf3("b", b);
            std::printf("\"t\": %u, \"dot\": %u, \"length\": %u, ", bits(t), bits(dot(a, b)), bits(length(a)));
            pf3("normalize", nrm); pf3("cross", crs); pf3("faceforward", ff); pf3("lerp", lrp); pf3("div", dv, false);
            std::printf("}%s\n", i < n - 1 ? "," : "");
        }
    }
    std::printf("],\n");

    // ---- StaticWorkDistribution -----------------------------------------------------
    std::printf("\"workdist\": [\n");
    {
        const int cfg[][3] = { {1920,1080,1}, {1920,1080,2}, {1920,1080,4}, {1920,1080,8}, {512,512,8}, {100,50,3}, {64,8,2} };
        const int n_cfg = sizeof(cfg) / sizeof(cfg[0]);
        for (int i = 0; i < n_cfg; ++i) {
            StaticWorkDistribution wd;
            wd.setRasterSize(cfg[i][0], cfg[i][1]);
            wd.setNumGPUs(cfg[i][2]);
            int ns = wd.numSamples(0);
            std::printf("  {\"width\": %d, \"height\": %d, \"gpus\": %d, \"num_samples\": %d, \"probe\": [", cfg[i][0], cfg[i][1], cfg[i][2], ns);
            // probes: (gpu, sample) -> pixel
            uint32_t seed = 4242u + i;
            for (int k = 0; k < 48; ++k) {
                int gpu = (int)(lcg(seed) % (uint32_t)cfg[i][2]);
                int smp = k < 4 ? k * 37 % ns : (int)(lcg(seed) % (uint32_t)ns);
                int2 px = wd.getSamplePixel(gpu, smp);
                std::printf("[%d, %d, %d, %d]%s", gpu, smp, px.x, px.y, k < 47 ? ", " : "");
            }
            std::printf("]}%s\n", i < n_cfg - 1 ? "," : "");
        }
    }
    std::printf("],\n");

    // ---- first primary rays of the Cornell launch (raygen arithmetic, sample 0) -------
    std::printf("\"primary\": [\n");
    {
        sutil::Camera cam(make_float3(278.f,273.f,-900.f), make_float3(278.f,273.f,330.f), make_float3(0.f,1.f,0.f), 35.f, 1.0f);
        float3 U, V, W; cam.UVWFrame(U, V, W);
        const int w = 512, h = 512;
        const int pix[][2] = { {0,0}, {511,511}, {7,5}, {256,256}, {100,400}, {511,0} };
        const int n_pix = sizeof(pix) / sizeof(pix[0]);
        for (int i = 0; i < n_pix; ++i) {
            unsigned int seed = tea<4>(pix[i][1] * w + pix[i][0], 0);
            unsigned int seed0 = seed;
            // same expressions as optixPathTracer.cu:267-274, evaluated by the reference's operators
            // the reference writes make_float2( rnd(seed), rnd(seed) ): argument evaluation order is
            // unspecified in C++; nvcc/clang evaluate left to right, which is what we pin here.
            const float jit_first = rnd(seed);
            const float jit_second = rnd(seed);
            const float2 jit = make_float2(jit_first, jit_second);
            const float2 d = 2.0f * make_float2((static_cast<float>(pix[i][0]) + jit.x) / static_cast<float>(w),
                                                (static_cast<float>(pix[i][1]) + jit.y) / static_cast<float>(h)) - 1.0f;
            float3 dir = normalize(d.x * U + d.y * V + W);
            std::printf("  {\"x\": %d, \"y\": %d, \"w\": %d, \"h\": %d, \"seed0\": %u, \"jitter\": [%u, %u], ", pix[i][0], pix[i][1], w, h, seed0, bits(jit.x), bits(jit.y));
            pf3("dir", dir, false);
            std::printf("}%s\n", i < n_pix - 1 ? "," : "");
        }
    }
    std::printf("],\n");

    // ---- layouts the C ABI keeps: stated by the reference's own headers ---------------------------------------
    // {"struct": {"sizeof": n, "alignof": a, "fields": {"name": [offset, size], ...}}, ...}
#define LAY_BEGIN(name, T) std::printf("  \"%s\": {\"sizeof\": %zu, \"alignof\": %zu, \"fields\": {", name, sizeof(T), alignof(T))
#define LAY_F(T, f, last) std::printf("\"%s\": [%zu, %zu]%s", #f, offsetof(T, f), sizeof(((T*)0)->f), last ? "" : ", ")
#define LAY_END(last) std::printf("}}%s\n", last ? "" : ",")
    std::printf("\"layouts\": {\n");
    LAY_BEGIN("Params", pt::Params);
    LAY_F(pt::Params, subframe_index, 0); LAY_F(pt::Params, accum_buffer, 0); LAY_F(pt::Params, frame_buffer, 0);
    LAY_F(pt::Params, width, 0); LAY_F(pt::Params, height, 0); LAY_F(pt::Params, samples_per_launch, 0);
    LAY_F(pt::Params, eye, 0); LAY_F(pt::Params, U, 0); LAY_F(pt::Params, V, 0); LAY_F(pt::Params, W, 0);
    LAY_F(pt::Params, light, 0); LAY_F(pt::Params, handle, 1);
    LAY_END(0);
    LAY_BEGIN("ParallelogramLight", pt::ParallelogramLight);
    LAY_F(pt::ParallelogramLight, corner, 0); LAY_F(pt::ParallelogramLight, v1, 0); LAY_F(pt::ParallelogramLight, v2, 0);
    LAY_F(pt::ParallelogramLight, normal, 0); LAY_F(pt::ParallelogramLight, emission, 1);
    LAY_END(0);
    LAY_BEGIN("HitGroupData", pt::HitGroupData);
    LAY_F(pt::HitGroupData, emission_color, 0); LAY_F(pt::HitGroupData, diffuse_color, 0); LAY_F(pt::HitGroupData, vertices, 1);
    LAY_END(0);
    LAY_BEGIN("MissData", pt::MissData);
    LAY_F(pt::MissData, bg_color, 1);
    LAY_END(0);
    LAY_BEGIN("RadiancePRD", pt::RadiancePRD);
    LAY_F(pt::RadiancePRD, attenuation, 0); LAY_F(pt::RadiancePRD, seed, 0); LAY_F(pt::RadiancePRD, depth, 0);
    LAY_F(pt::RadiancePRD, emitted, 0); LAY_F(pt::RadiancePRD, radiance, 0); LAY_F(pt::RadiancePRD, origin, 0);
    LAY_F(pt::RadiancePRD, direction, 0); LAY_F(pt::RadiancePRD, done, 1);
    LAY_END(0);
    typedef whitted::LaunchParams WLP;
    LAY_BEGIN("whitted::LaunchParams", WLP);
    LAY_F(WLP, width, 0); LAY_F(WLP, height, 0); LAY_F(WLP, subframe_index, 0); LAY_F(WLP, accum_buffer, 0); LAY_F(WLP, frame_buffer, 0);
    LAY_F(WLP, max_depth, 0); LAY_F(WLP, scene_epsilon, 0); LAY_F(WLP, eye, 0); LAY_F(WLP, U, 0); LAY_F(WLP, V, 0); LAY_F(WLP, W, 0);
    LAY_F(WLP, lights, 0); LAY_F(WLP, miss_color, 0); LAY_F(WLP, handle, 1);
    LAY_END(0);
    typedef BufferView<Light> BVL;
    LAY_BEGIN("BufferView", BVL);
    LAY_F(BVL, data, 0); LAY_F(BVL, count, 0); LAY_F(BVL, byte_stride, 0); LAY_F(BVL, elmt_byte_size, 1);
    LAY_END(0);
    LAY_BEGIN("Light", Light);
    LAY_F(Light, type, 0); LAY_F(Light, point, 0); LAY_F(Light, point.color, 0); LAY_F(Light, point.intensity, 0);
    LAY_F(Light, point.position, 0); LAY_F(Light, point.falloff, 0); LAY_F(Light, ambient, 0); LAY_F(Light, ambient.color, 1);
    LAY_END(0);
    typedef MaterialData::Texture MTex;
    LAY_BEGIN("MaterialData::Texture", MTex);
    LAY_F(MTex, texcoord, 0); LAY_F(MTex, tex, 0); LAY_F(MTex, texcoord_offset, 0); LAY_F(MTex, texcoord_rotation, 0); LAY_F(MTex, texcoord_scale, 1);
    LAY_END(0);
    LAY_BEGIN("MaterialData", MaterialData);
    LAY_F(MaterialData, type, 0); LAY_F(MaterialData, normal_tex, 0); LAY_F(MaterialData, alpha_mode, 0); LAY_F(MaterialData, alpha_cutoff, 0);
    LAY_F(MaterialData, emissive_factor, 0); LAY_F(MaterialData, emissive_tex, 0); LAY_F(MaterialData, doubleSided, 0);
    LAY_F(MaterialData, pbr, 0); LAY_F(MaterialData, pbr.base_color, 0); LAY_F(MaterialData, pbr.metallic, 0); LAY_F(MaterialData, pbr.roughness, 0);
    LAY_F(MaterialData, pbr.base_color_tex, 0); LAY_F(MaterialData, pbr.metallic_roughness_tex, 1);
    LAY_END(0);
    LAY_BEGIN("whitted::HitGroupData", whitted::HitGroupData);
    LAY_F(whitted::HitGroupData, geometry_data, 0); LAY_F(whitted::HitGroupData, material_data, 1);
    LAY_END(0);
    LAY_BEGIN("OptixInstance", OptixInstance);
    LAY_F(OptixInstance, transform, 0); LAY_F(OptixInstance, instanceId, 0); LAY_F(OptixInstance, sbtOffset, 0);
    LAY_F(OptixInstance, visibilityMask, 0); LAY_F(OptixInstance, flags, 0); LAY_F(OptixInstance, traversableHandle, 0); LAY_F(OptixInstance, pad, 1);
    LAY_END(1);
    std::printf("}\n}\n");
    return 0;
}
