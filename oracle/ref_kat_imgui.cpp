// ref_kat_imgui.cpp -- known-answer-vector generator for the SDK/imgui_test restatement (test infrastructure).
//
// This file is OURS; it #includes the reference's own host-callable classes straight from /root/reference (never copied)
//   SDK/imgui_test/camera.h            Camera::set_fd :36, compute_uvw :110-123, compute_ray :127-144
//   SDK/imgui_test/light.h             LightVariant::wi / lumi :19-40 (point_light.h:20-25, directional_light.h:19-24,
//                                      volumetric_light.h:22-27)
//   SDK/imgui_test/diffuse.h           DiffuseMaterial::f :8-12
//   SDK/imgui_test/optixTriangle.h     struct Params :42-108 (sizeof / offsetof only)
// and prints, as JSON, the object layouts and what those functions compute for fixed inputs.  Output ->
// tests/golden/ref_kat_imgui.json (committed), which pins the imgui_test part of oracle/pt_oracle.c and the layouts of
// MiptImguiCamera / MiptImguiLight / MiptImguiParams (include/mipt.h).
//
// COMPILER: built with clang++ (oracle/Makefile target `ref`).  Three of these functions draw random numbers in sibling
// arguments of one call -- camera.h:139 make_float2(rnd, rnd), directional_light.h:21 and volumetric_light.h:24
// make_float3(rnd, rnd, rnd) -- and C++ leaves the evaluation order of function arguments unspecified.  The reference's
// device code is compiled by nvcc, whose device compiler is clang/NVVM based and evaluates arguments left to right;
// clang++ does the same on the host (g++ goes right to left), so this generator is built with clang++ to pin THAT order.
#define private public                    // the layout probes below read private members; the classes are not modified
#include <cuda_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <sutil/vec_math.h>
#include <cuda/random.h>
#include <optixTriangle.h>
#undef private

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static void pf3(const char* name, float3 v, bool comma = true)
{
    std::printf("\"%s\": [%u, %u, %u]%s", name, bits(v.x), bits(v.y), bits(v.z), comma ? ", " : "");
}
template <class T> static void words(const char* name, const T& obj, bool comma = true)
{
    uint32_t w[(sizeof(T) + 3) / 4] = {};
    std::memcpy(w, &obj, sizeof(T));
    std::printf("\"%s\": [", name);
    for (size_t i = 0; i < sizeof(w) / 4; ++i) std::printf("%u%s", w[i], i + 1 < sizeof(w) / 4 ? ", " : "");
    std::printf("]%s", comma ? ", " : "");
}
#define OFF(T, m) (unsigned)offsetof(T, m)

int main()
{
    std::printf("{\n");
    // ---- layouts -----------------------------------------------------------------------------------------------------
    std::printf("\"layout\": {\"Camera\": {\"size\": %u, \"m_eye\": %u, \"m_lookat\": %u, \"m_up\": %u, \"m_ortho\": %u, \"m_fov\": %u, "
                "\"m_fd\": %u, \"m_aperture\": %u, \"m_speed\": %u, \"u\": %u, \"v\": %u, \"w\": %u},\n",
                (unsigned)sizeof(Camera), OFF(Camera, m_eye), OFF(Camera, m_lookat), OFF(Camera, m_up), OFF(Camera, m_ortho), OFF(Camera, m_fov),
                OFF(Camera, m_fd), OFF(Camera, m_aperture), OFF(Camera, m_speed), OFF(Camera, u), OFF(Camera, v), OFF(Camera, w));
    std::printf("  \"LightVariant\": {\"size\": %u, \"_payload\": %u, \"_tag\": %u}, \"PointLight\": {\"size\": %u, \"m_position\": %u, \"m_lumi\": %u, \"m_dark\": %u},\n",
                (unsigned)sizeof(LightVariant), OFF(LightVariant, _payload), OFF(LightVariant, _tag), (unsigned)sizeof(PointLight),
                OFF(PointLight, m_position), OFF(PointLight, m_lumi), OFF(PointLight, m_dark));
    std::printf("  \"DirectionalLight\": {\"size\": %u, \"m_direction\": %u, \"m_lumi\": %u, \"m_jitter\": %u, \"m_dark\": %u},\n",
                (unsigned)sizeof(DirectionalLight), OFF(DirectionalLight, m_direction), OFF(DirectionalLight, m_lumi), OFF(DirectionalLight, m_jitter),
                OFF(DirectionalLight, m_dark));
    std::printf("  \"VolumetricLight\": {\"size\": %u, \"m_position\": %u, \"m_radius\": %u, \"m_lumi\": %u, \"m_dark\": %u}, \"DiffuseMaterial\": {\"size\": %u, \"m_color\": %u},\n",
                (unsigned)sizeof(VolumetricLight), OFF(VolumetricLight, m_position), OFF(VolumetricLight, m_radius), OFF(VolumetricLight, m_lumi),
                OFF(VolumetricLight, m_dark), (unsigned)sizeof(DiffuseMaterial), OFF(DiffuseMaterial, m_color));
    std::printf("  \"Params\": {\"size\": %u, \"image_width\": %u, \"image_height\": %u, \"samples_per_frame\": %u, \"camera\": %u, \"dt\": %u, \"dirty\": %u, "
                "\"image\": %u, \"film\": %u, \"tfactor\": %u, \"handle\": %u, \"normals\": %u, \"vertices\": %u, \"mat_indices\": %u, \"nmat_indices\": %u, "
                "\"lights\": %u, \"nlights\": %u, \"materials\": %u, \"nmaterials\": %u}},\n",
                (unsigned)sizeof(Params), OFF(Params, image_width), OFF(Params, image_height), OFF(Params, samples_per_frame), OFF(Params, camera),
                OFF(Params, dt), OFF(Params, dirty), OFF(Params, image), OFF(Params, film), OFF(Params, tfactor), OFF(Params, handle), OFF(Params, normals),
                OFF(Params, vertices), OFF(Params, mat_indices), OFF(Params, nmat_indices), OFF(Params, lights), OFF(Params, nlights),
                OFF(Params, materials), OFF(Params, nmaterials));

    // ---- Camera: set_fd, compute_uvw, compute_ray ------------------------------------------------------------------------
    struct CamIn { float3 eye, lookat, up; float fov, fd_in, aperture; bool ortho; };
    const CamIn cams[] = {
        { {0.f, 1.f, -2.f}, {0.f, 0.f, 0.f}, {0.f, 1.f, 0.000073f}, 45.f, 2.2360679f, 0.0f, false },     // the class defaults (camera.h:155-173)
        { {0.f, 1.f, -2.f}, {0.f, 0.f, 0.f}, {0.f, 1.f, 0.000073f}, 45.f, 1.5f, 0.05f, false },           // thin lens
        { {3.f, 2.5f, -4.f}, {0.2f, 0.4f, 0.1f}, {0.1f, 0.9f, 0.2f}, 60.f, 7.0f, 0.2f, false },
        { {3.f, 2.5f, -4.f}, {0.2f, 0.4f, 0.1f}, {0.1f, 0.9f, 0.2f}, 30.f, 4.0f, 0.2f, true },            // orthographic: no draws
        { {-10.f, 0.3f, 0.5f}, {1.f, 1.f, 1.f}, {0.f, 1.f, 0.f}, 75.f, 0.5f, 1.0f, false },
    };
    std::printf("\"camera\": [\n");
    const int n_cam = sizeof(cams) / sizeof(cams[0]);
    for (int i = 0; i < n_cam; ++i) {
        Camera cam;
        cam.set_eye(cams[i].eye); cam.set_lookat(cams[i].lookat); cam.set_up(cams[i].up);
        cam.set_fov(cams[i].fov); cam.set_aperture(cams[i].aperture); cam.set_ortho(cams[i].ortho);
        cam.set_fd(cams[i].fd_in);
        cam.compute_uvw();
        std::printf("  {");
        pf3("eye", cams[i].eye); pf3("lookat", cams[i].lookat); pf3("up", cams[i].up);
        std::printf("\"fov\": %u, \"fd_in\": %u, \"aperture\": %u, \"ortho\": %d, \"m_fd\": %u, ", bits(cams[i].fov), bits(cams[i].fd_in),
                    bits(cams[i].aperture), cams[i].ortho ? 1 : 0, bits(cam.m_fd));
        pf3("u", cam.u); pf3("v", cam.v); pf3("w", cam.w);
        words("object", cam);
        std::printf("\"rays\": [");
        const unsigned dims[][2] = { {400, 300}, {1024, 768}, {1920, 1080} };
        const unsigned pix[][2] = { {0, 0}, {399, 299}, {17, 250}, {200, 150}, {1, 0}, {333, 7} };
        int first = 1;
        for (const auto& dm : dims) for (const auto& px : pix) {
            const uint3 idx = make_uint3(px[0] * dm[0] / 400u, px[1] * dm[1] / 300u, 0u), dim = make_uint3(dm[0], dm[1], 1u);
            unsigned int seed = tea<4>(idx.x + dim.x * idx.y, 4u * (1u + px[0] % 3u));
            const unsigned int seed0 = seed;
            float3 o, d;
            cam.compute_ray(idx, dim, o, d, seed);
            float3 o2, d2;
            cam.compute_ray(idx, dim, o2, d2, seed);                      // the second sample of the pixel continues the stream
            std::printf("%s{\"idx\": [%u, %u], \"dim\": [%u, %u], \"seed0\": %u, ", first ? "" : ", ", idx.x, idx.y, dim.x, dim.y, seed0);
            pf3("origin", o); pf3("direction", d); pf3("origin2", o2); pf3("direction2", d2);
            std::printf("\"seed_after\": %u}", seed);
            first = 0;
        }
        std::printf("]}%s\n", i + 1 < n_cam ? "," : "");
    }
    std::printf("],\n");

    // ---- LightVariant: object bytes, wi, lumi ------------------------------------------------------------------------------
    std::vector<LightVariant> lights;
    lights.push_back(LightVariant(PointLight(make_float3(1.f, 1.f, 0.f), make_float3(1.f, 1.f, 1.f))));
    lights.push_back(LightVariant(PointLight(make_float3(-3.f, 4.5f, 2.25f), make_float3(10.f, 8.f, 6.f))));
    lights.push_back(LightVariant(DirectionalLight(make_float3(1.f, 1.f, 0.f), make_float3(0.5f, 0.25f, 0.125f), 0.0f)));
    lights.push_back(LightVariant(DirectionalLight(make_float3(-0.3f, 1.f, 0.4f), make_float3(2.f, 2.f, 1.5f), 0.25f)));
    lights.push_back(LightVariant(VolumetricLight(make_float3(0.f, 3.f, 0.f), 0.5f, make_float3(4.f, 3.f, 2.f))));
    lights.push_back(LightVariant(VolumetricLight(make_float3(2.f, 1.f, -1.f), 2.0f, make_float3(0.1f, 0.2f, 0.3f))));
    std::printf("\"lights\": [\n");
    for (size_t i = 0; i < lights.size(); ++i) {
        std::printf("  {\"tag\": %d, ", (int)lights[i]._tag);
        words("object", lights[i]);
        pf3("lumi", lights[i].lumi());
        std::printf("\"wi\": [");
        uint32_t s = 99u + (uint32_t)i;
        for (int k = 0; k < 8; ++k) {
            const float3 p = make_float3(rnd(s) * 4.f - 2.f, rnd(s) * 2.f, rnd(s) * 4.f - 2.f);       // (our own inputs; any order is fine, they are printed)
            unsigned int seed = tea<4>(1000u * (unsigned)i + (unsigned)k, 8u);
            const unsigned int seed0 = seed;
            const float3 a = lights[i].wi(p, seed);
            const float3 b = lights[i].wi(p, seed);                       // a second light of the same kind continues the stream
            std::printf("%s{", k ? ", " : "");
            pf3("p", p);
            std::printf("\"seed0\": %u, ", seed0);
            pf3("wi", a); pf3("wi2", b);
            std::printf("\"seed_after\": %u}", seed);
        }
        std::printf("]}%s\n", i + 1 < lights.size() ? "," : "");
    }
    std::printf("],\n");

    // ---- DiffuseMaterial::f ---------------------------------------------------------------------------------------------------
    std::printf("\"diffuse\": [");
    const float3 colors[] = { {0.5f, 0.5f, 0.5f}, {0.9f, 0.1f, 0.3f}, {0.f, 1.f, 0.25f} };
    for (int i = 0; i < 3; ++i) {
        DiffuseMaterial m(colors[i]);
        unsigned int seed = 5u;
        const float3 f = m.f(make_float3(1.f, 2.f, 3.f), make_float3(0.f, 1.f, 0.f), make_float3(0.f, 1.f, 0.f), seed);
        std::printf("%s{", i ? ", " : "");
        pf3("color", colors[i]); pf3("f", f);
        words("object", m);
        std::printf("\"seed_after\": %u}", seed);
    }
    std::printf("]\n}\n");
    return 0;
}
