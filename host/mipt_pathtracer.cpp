// mipt_pathtracer -- C++ host of the path-tracer sample above the libmipt C ABI.
//
// Mirrors SDK/optixPathTracer/optixPathTracer.cpp in file-output mode (main :927-1093): the same state
// functions in the same order, the same command-line flags (--file/-f, --dim=WxH, --launch-samples/-s),
// the same Params / ParallelogramLight / HitGroupData / MissData structs; sutil::HIPOutputBuffer<uchar4> (= CUDAOutputBuffer,
// CUDA_DEVICE or ZERO_COPY) and sutil::Camera come from include/mipt_sutil.hpp; the image is written by mipt_save_image (sutil::saveImage's rules, sutil.cpp:542-709: rows are
// flipped on write, row 0 of the buffer is the bottom of the image).  createModule / createProgramGroups /
// createPipeline (:686-826) have no counterpart.  Extra flags: --scene cornell|soup, --tris N,
// --max-depth K (0 = Russian roulette only, the reference's behaviour), --subframes N.
#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/mipt_sutil.hpp"     // sutil::Camera, HIPOutputBuffer<T>, Exception, MIPT_CHECK over include/mipt.h

namespace {

using sutil::Camera;
using sutil::Exception;
using sutil::HIPOutputBuffer;
typedef float3 float3_;
typedef float4 float4_;
typedef uchar4 uchar4_;

// ---- structs of optixPathTracer.h:82-126, unchanged ---------------------------------------------
struct ParallelogramLight { float3_ corner, v1, v2, normal, emission; };
struct Params {
    unsigned int subframe_index;
    float4_*     accum_buffer;
    uchar4_*     frame_buffer;
    unsigned int width, height, samples_per_launch;
    float3_      eye, U, V, W;
    ParallelogramLight light;
    unsigned long long handle;
};
static_assert(sizeof(Params) == sizeof(MiptParams), "Params must stay layout compatible");
struct Vertex { float x, y, z, pad; };                                     // optixPathTracer.cpp:96-99

struct PathTracerState {                                                    // optixPathTracer.cpp:110-131
    MiptContext*       context = nullptr;
    unsigned long long gas_handle = 0;
    Vertex*            d_vertices = nullptr;
    hipStream_t        stream = nullptr;
    Params             params{};
};

// ---- scene data ----------------------------------------------------------------------------------
std::vector<Vertex>   g_vertices;
std::vector<uint32_t> g_mat_indices;
std::vector<float3_>  g_emission_colors, g_diffuse_colors;
Camera                camera;
int                   samples_per_launch = 16;                              // optixPathTracer.cpp:75
std::string           scene_name = "cornell";
uint32_t              soup_tris = 1000000;

void quad(int mat, float3_ a, float3_ b, float3_ c, float3_ d)
{
    const float3_ v[6] = { a, b, c, a, c, d };
    for (const float3_& p : v) g_vertices.push_back({ p.x, p.y, p.z, 0.0f });
    g_mat_indices.push_back((uint32_t)mat); g_mat_indices.push_back((uint32_t)mat);
}

void buildSceneTables(PathTracerState& state)
{
    if (scene_name == "cornell") {                                          // optixPathTracer.cpp:144-314, as quads
        const float W = 556.0f, H = 548.8f, D = 559.2f;
        quad(0, {0,0,0}, {0,0,D}, {W,0,D}, {W,0,0});          quad(0, {0,H,0}, {W,H,0}, {W,H,D}, {0,H,D});
        quad(0, {0,0,D}, {0,H,D}, {W,H,D}, {W,0,D});          quad(1, {0,0,0}, {0,H,0}, {0,H,D}, {0,0,D});
        quad(2, {W,0,0}, {W,0,D}, {W,H,D}, {W,H,0});
        quad(0, {130,165,65}, {82,165,225}, {242,165,274}, {290,165,114});
        quad(0, {290,0,114}, {290,165,114}, {240,165,272}, {240,0,272});
        quad(0, {130,0,65}, {130,165,65}, {290,165,114}, {290,0,114});
        quad(0, {82,0,225}, {82,165,225}, {130,165,65}, {130,0,65});
        quad(0, {240,0,272}, {240,165,272}, {82,165,225}, {82,0,225});
        quad(0, {423,330,247}, {265,330,296}, {314,330,455}, {472,330,406});
        quad(0, {423,0,247}, {423,330,247}, {472,330,406}, {472,0,406});
        quad(0, {472,0,406}, {472,330,406}, {314,330,456}, {314,0,456});
        quad(0, {314,0,456}, {314,330,456}, {265,330,296}, {265,0,296});
        quad(0, {265,0,296}, {265,330,296}, {423,330,247}, {423,0,247});
        quad(3, {343,548.6f,227}, {213,548.6f,227}, {213,548.6f,332}, {343,548.6f,332});
        g_emission_colors = { {0,0,0}, {0,0,0}, {0,0,0}, {15,15,5} };
        g_diffuse_colors  = { {0.80f,0.80f,0.80f}, {0.05f,0.80f,0.05f}, {0.80f,0.05f,0.05f}, {0.50f,0.00f,0.00f} };
        state.params.light = { {343.0f,548.5f,227.0f}, {0,0,105.0f}, {-130.0f,0,0}, {0,-1,0}, {15,15,5} };   // :435-439
        camera.setEye({278.0f,273.0f,-900.0f}); camera.setLookat({278.0f,273.0f,330.0f});                    // :538-541
    } else {                                                                // SURVEY.md 8(d) "C4"
        g_vertices.resize((size_t)soup_tris * 3);
        if (mipt_scene_random_triangles(soup_tris, 0x5EED, 0.01f, &g_vertices[0].x) != MIPT_SUCCESS) throw Exception("scene generation failed");
        g_mat_indices.assign(soup_tris, 0u);
        g_emission_colors = { {0,0,0} };
        g_diffuse_colors  = { {0.8f,0.8f,0.8f} };
        state.params.light = { {0.25f,1.5f,0.25f}, {0.5f,0,0}, {0,0,0.5f}, {0,-1,0}, {15,15,5} };
        camera.setEye({0.5f,0.5f,-1.5f}); camera.setLookat({0.5f,0.5f,0.5f});
    }
    camera.setUp({0,1,0}); camera.setFovY(35.0f);
}

void context_log_cb(unsigned int level, const char* tag, const char* message, void*)     // optixPathTracer.cpp:530-533
{
    std::cerr << "[" << std::setw(2) << level << "][" << std::setw(12) << tag << "]: " << message << "\n";
}

void createContext(PathTracerState& state)                                  // :555-573
{
    const int rc = mipt_context_create(0, &context_log_cb, nullptr, 4, &state.context);
    if (rc != MIPT_SUCCESS) throw Exception(std::string("mipt_context_create failed: ") + mipt_error_name(rc));
}

void buildMeshAccel(PathTracerState& state)                                 // :576-684
{
    const size_t bytes = g_vertices.size() * sizeof(Vertex);
    HIP_CHECK(hipMalloc((void**)&state.d_vertices, bytes));
    HIP_CHECK(hipMemcpy(state.d_vertices, g_vertices.data(), bytes, hipMemcpyHostToDevice));
    MiptBuildInput in{};
    in.vertices = state.d_vertices; in.num_vertices = (uint32_t)g_vertices.size(); in.vertex_stride_bytes = sizeof(Vertex);
    in.sbt_index_offsets = g_mat_indices.data();
    MIPT_CHECK(state.context, mipt_accel_build(state.context, &in, 1, &state.gas_handle));
}

void createSBT(PathTracerState& state)                                      // :829-898
{
    MiptMissData ms{ { 0.0f, 0.0f, 0.0f, 0.0f } };
    MIPT_CHECK(state.context, mipt_sbt_set_miss(state.context, &ms));
    std::vector<MiptHitGroupData> recs(g_emission_colors.size());
    for (size_t i = 0; i < recs.size(); ++i) {
        std::memcpy(recs[i].emission_color, &g_emission_colors[i], 12);
        std::memcpy(recs[i].diffuse_color, &g_diffuse_colors[i], 12);
        recs[i].vertices = state.d_vertices;
    }
    MIPT_CHECK(state.context, mipt_sbt_set_hitgroups(state.context, recs.data(), (uint32_t)recs.size()));
}

void initLaunchParams(PathTracerState& state)                               // :424-445
{
    HIP_CHECK(hipMalloc((void**)&state.params.accum_buffer, (size_t)state.params.width * state.params.height * sizeof(float4_)));
    state.params.frame_buffer = nullptr;
    state.params.samples_per_launch = (unsigned)samples_per_launch;
    state.params.subframe_index = 0u;
    state.params.handle = state.gas_handle;
    HIP_CHECK(hipStreamCreate(&state.stream));
}

void handleCameraUpdate(Params& params)                                     // :448-457
{
    camera.setAspectRatio(static_cast<float>(params.width) / static_cast<float>(params.height));
    params.eye = camera.eye();
    camera.UVWFrame(params.U, params.V, params.W);
}

void launchSubframe(HIPOutputBuffer<uchar4_>& output_buffer, PathTracerState& state)     // :488-511
{
    state.params.frame_buffer = output_buffer.map();
    output_buffer.setStream(state.stream);
    MIPT_CHECK(state.context, mipt_launch(state.context, state.stream, reinterpret_cast<const MiptParams*>(&state.params),
                           state.params.width, state.params.height, 1));
    output_buffer.unmap();
    MIPT_CHECK(state.context, mipt_sync(state.context));
}

void saveImage(const char* fname, uchar4_* data, int width, int height)                   // :1056-1062
{
    sutil::ImageBuffer buffer;
    buffer.data = data; buffer.width = (unsigned)width; buffer.height = (unsigned)height; buffer.pixel_format = sutil::UNSIGNED_BYTE4;
    sutil::saveImage(fname, buffer, false);                                                 // .ppm or .png, as the reference
}

void cleanupState(PathTracerState& state)                                   // :901-918
{
    if (state.context) mipt_context_destroy(state.context);
    if (state.d_vertices) (void)hipFree(state.d_vertices);
    if (state.params.accum_buffer) (void)hipFree(state.params.accum_buffer);
    if (state.stream) (void)hipStreamDestroy(state.stream);
}

[[noreturn]] void printUsageAndExit(const char* argv0)
{
    std::cerr << "Usage  : " << argv0 << " [options]\n"
              << "Options: --file | -f <filename>      File for image output (required: there is no display path)\n"
              << "         --launch-samples | -s       Number of samples per pixel per launch (default 16)\n"
              << "         --dim=<width>x<height>      Set image dimensions; defaults to 768x768\n"
              << "         --scene cornell|soup        --tris <n>   --max-depth <k>   --subframes <n>\n"
              << "         --zero-copy                 frame buffer in pinned, mapped host memory (CUDAOutputBufferType::ZERO_COPY)\n";
    std::exit(0);
}

}  // namespace

int main(int argc, char* argv[])
{
    PathTracerState state;
    state.params.width = 768; state.params.height = 768;
    std::string outfile;
    uint64_t max_depth = 0; int subframes = 1;
    sutil::HIPOutputBufferType output_buffer_type = sutil::HIPOutputBufferType::HIP_DEVICE;
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg == "--help" || arg == "-h") printUsageAndExit(argv[0]);
        else if ((arg == "--file" || arg == "-f") && i < argc - 1) outfile = argv[++i];
        else if (arg.substr(0, 6) == "--dim=") {
            int w = 0, h = 0;
            if (std::sscanf(arg.c_str() + 6, "%dx%d", &w, &h) != 2 || w <= 0 || h <= 0) printUsageAndExit(argv[0]);
            state.params.width = (unsigned)w; state.params.height = (unsigned)h;
        }
        else if ((arg == "--launch-samples" || arg == "-s") && i < argc - 1) samples_per_launch = std::atoi(argv[++i]);
        else if (arg == "--zero-copy") output_buffer_type = sutil::HIPOutputBufferType::ZERO_COPY;
        else if (arg == "--scene" && i < argc - 1) scene_name = argv[++i];
        else if (arg == "--tris" && i < argc - 1) soup_tris = (uint32_t)std::atoll(argv[++i]);
        else if (arg == "--max-depth" && i < argc - 1) max_depth = (uint64_t)std::atoll(argv[++i]);
        else if (arg == "--subframes" && i < argc - 1) subframes = std::atoi(argv[++i]);
        else { std::cerr << "Unknown option '" << argv[i] << "'\n"; printUsageAndExit(argv[0]); }
    }
    if (outfile.empty()) printUsageAndExit(argv[0]);
    try {
        buildSceneTables(state);
        createContext(state);
        buildMeshAccel(state);
        createSBT(state);
        initLaunchParams(state);
        MIPT_CHECK(state.context, mipt_set_option(state.context, MIPT_OPT_MAX_DEPTH, max_depth));
        HIPOutputBuffer<uchar4_> output_buffer(output_buffer_type, (int)state.params.width, (int)state.params.height);
        handleCameraUpdate(state.params);
        for (int s = 0; s < subframes; ++s) {                              // the progressive loop of :1021-1045, headless
            state.params.subframe_index = (unsigned)s;
            launchSubframe(output_buffer, state);
        }
        MiptLaunchStats st{};
        MIPT_CHECK(state.context, mipt_get_launch_stats(state.context, &st));
        std::cerr << "last launch: " << st.closest_rays << " closest + " << st.shadow_rays << " shadow rays, "
                  << st.bounces << " wavefront iterations\n";
        saveImage(outfile.c_str(), output_buffer.getHostPointer(), output_buffer.width(), output_buffer.height());
        cleanupState(state);
    } catch (std::exception& e) {                                          // :1089-1093
        std::cerr << "Caught exception: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
