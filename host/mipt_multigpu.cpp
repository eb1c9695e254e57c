// mipt_multigpu -- C++ host of the multi-GPU path tracer above libmipt's in-process group API (mipt_group_*), written against
// include/mipt_sutil.hpp.
//
// Mirrors SDK/optixMultiGPU/optixMultiGPU.cpp in file-output mode: ONE host thread, one PathTracerState per device with its own
// context, stream, replicated scene and acceleration structure (:1093-1102), launchSubframe iterating the devices (:562-594), the
// image split by StaticWorkDistribution (8x4-pixel tiles dealt round the devices, sutil/WorkDistribution.h:34-91).  Where the
// sample merges through one zero-copy host buffer (optixMultiGPU.cu:282-292), this one all-gathers the devices' compact float4
// shards over RCCL / xGMI (ONE grouped ncclAllGather) and scatters them on device 0.  Flags as the sample: --file | -f,
// --dim=WxH, --launch-samples | -s; extra: --devices a,b,..  (default: every visible device), --subframes N, --max-depth K.
// The scene is the Cornell box of optixPathTracer.cpp:144-314, which optixMultiGPU shares.
#include <array>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "../include/mipt_sutil.hpp"

namespace {

struct ParallelogramLight { float3 corner, v1, v2, normal, emission; };     // optixMultiGPU.h
struct Params {
    unsigned int subframe_index;
    float4*      accum_buffer;
    uchar4*      frame_buffer;
    unsigned int width, height, samples_per_launch;
    float3       eye, U, V, W;
    ParallelogramLight light;
    unsigned long long handle;
};
static_assert(sizeof(Params) == sizeof(MiptParams), "Params must stay layout compatible");
struct Vertex { float x, y, z, pad; };

struct PathTracerState {                                                    // optixMultiGPU.cpp:105-135, one per device
    int                device_idx = -1;
    MiptContext*       context = nullptr;                                   // owned by the group
    unsigned long long gas_handle = 0;
    Vertex*            d_vertices = nullptr;
    hipStream_t        stream = nullptr;                                    // the group's stream of this device
    Params             params{};
};

std::vector<Vertex>   g_vertices;
std::vector<uint32_t> g_mat_indices;
std::vector<float3>   g_emission_colors, g_diffuse_colors;
sutil::Camera         camera;
int                   samples_per_launch = 16, width = 768, height = 768;

void quad(int mat, float3 a, float3 b, float3 c, float3 d)
{
    const float3 v[6] = { a, b, c, a, c, d };
    for (const float3& p : v) g_vertices.push_back({ p.x, p.y, p.z, 0.0f });
    g_mat_indices.push_back((uint32_t)mat); g_mat_indices.push_back((uint32_t)mat);
}
float3 f3(float x, float y, float z) { return make_float3(x, y, z); }

void buildSceneTables()
{
    const float W = 556.0f, H = 548.8f, D = 559.2f;
    quad(0, f3(0,0,0), f3(0,0,D), f3(W,0,D), f3(W,0,0));          quad(0, f3(0,H,0), f3(W,H,0), f3(W,H,D), f3(0,H,D));
    quad(0, f3(0,0,D), f3(0,H,D), f3(W,H,D), f3(W,0,D));          quad(1, f3(0,0,0), f3(0,H,0), f3(0,H,D), f3(0,0,D));
    quad(2, f3(W,0,0), f3(W,0,D), f3(W,H,D), f3(W,H,0));
    quad(0, f3(130,165,65), f3(82,165,225), f3(242,165,274), f3(290,165,114));
    quad(0, f3(290,0,114), f3(290,165,114), f3(240,165,272), f3(240,0,272));
    quad(0, f3(130,0,65), f3(130,165,65), f3(290,165,114), f3(290,0,114));
    quad(0, f3(82,0,225), f3(82,165,225), f3(130,165,65), f3(130,0,65));
    quad(0, f3(240,0,272), f3(240,165,272), f3(82,165,225), f3(82,0,225));
    quad(0, f3(423,330,247), f3(265,330,296), f3(314,330,455), f3(472,330,406));
    quad(0, f3(423,0,247), f3(423,330,247), f3(472,330,406), f3(472,0,406));
    quad(0, f3(472,0,406), f3(472,330,406), f3(314,330,456), f3(314,0,456));
    quad(0, f3(314,0,456), f3(314,330,456), f3(265,330,296), f3(265,0,296));
    quad(0, f3(265,0,296), f3(265,330,296), f3(423,330,247), f3(423,0,247));
    quad(3, f3(343,548.6f,227), f3(213,548.6f,227), f3(213,548.6f,332), f3(343,548.6f,332));
    g_emission_colors = { f3(0,0,0), f3(0,0,0), f3(0,0,0), f3(15,15,5) };
    g_diffuse_colors  = { f3(0.80f,0.80f,0.80f), f3(0.05f,0.80f,0.05f), f3(0.80f,0.05f,0.05f), f3(0.50f,0.00f,0.00f) };
    camera.setEye(f3(278.0f,273.0f,-900.0f)); camera.setLookat(f3(278.0f,273.0f,330.0f));
    camera.setUp(f3(0,1,0)); camera.setFovY(35.0f);
}

void context_log_cb(unsigned int level, const char* tag, const char* message, void*)
{
    std::cerr << "[" << std::setw(2) << level << "][" << std::setw(12) << tag << "]: " << message << "\n";
}

void buildMeshAccel(PathTracerState& state)
{
    HIP_CHECK(hipSetDevice(state.device_idx));
    const size_t bytes = g_vertices.size() * sizeof(Vertex);
    HIP_CHECK(hipMalloc((void**)&state.d_vertices, bytes));
    HIP_CHECK(hipMemcpy(state.d_vertices, g_vertices.data(), bytes, hipMemcpyHostToDevice));
    MiptBuildInput in{};
    in.vertices = state.d_vertices; in.num_vertices = (uint32_t)g_vertices.size(); in.vertex_stride_bytes = sizeof(Vertex);
    in.sbt_index_offsets = g_mat_indices.data();
    MIPT_CHECK(state.context, mipt_accel_build(state.context, &in, 1, &state.gas_handle));
}

void createSBT(PathTracerState& state)
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

void initLaunchParams(PathTracerState& state)
{
    state.params.width = (unsigned)width; state.params.height = (unsigned)height;
    state.params.accum_buffer = nullptr; state.params.frame_buffer = nullptr;      // the group owns the compact shard buffers
    state.params.samples_per_launch = (unsigned)samples_per_launch;
    state.params.subframe_index = 0u;
    state.params.light = { f3(343.0f,548.5f,227.0f), f3(0,0,105.0f), f3(-130.0f,0,0), f3(0,-1,0), f3(15,15,5) };
    state.params.handle = state.gas_handle;
    camera.setAspectRatio(static_cast<float>(width) / static_cast<float>(height));
    state.params.eye = camera.eye();
    camera.UVWFrame(state.params.U, state.params.V, state.params.W);
}

[[noreturn]] void printUsageAndExit(const char* argv0)
{
    std::cerr << "Usage  : " << argv0 << " [options]\n"
              << "Options: --file | -f <filename>      File for image output (required)\n"
              << "         --launch-samples | -s       Number of samples per pixel per launch (default 16)\n"
              << "         --dim=<width>x<height>      Set image dimensions; defaults to 768x768\n"
              << "         --devices a,b,..            HIP devices to use (default: all)   --subframes <n>   --max-depth <k>\n";
    std::exit(0);
}

}  // namespace

int main(int argc, char* argv[])
{
    std::string outfile;
    std::vector<int> devices;
    uint64_t max_depth = 0; int subframes = 1;
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg == "--help" || arg == "-h") printUsageAndExit(argv[0]);
        else if ((arg == "--file" || arg == "-f") && i < argc - 1) outfile = argv[++i];
        else if (arg.substr(0, 6) == "--dim=") {
            if (std::sscanf(arg.c_str() + 6, "%dx%d", &width, &height) != 2 || width <= 0 || height <= 0) printUsageAndExit(argv[0]);
        }
        else if ((arg == "--launch-samples" || arg == "-s") && i < argc - 1) samples_per_launch = std::atoi(argv[++i]);
        else if (arg == "--subframes" && i < argc - 1) subframes = std::atoi(argv[++i]);
        else if (arg == "--max-depth" && i < argc - 1) max_depth = (uint64_t)std::atoll(argv[++i]);
        else if (arg == "--devices" && i < argc - 1) {
            const std::string list = argv[++i];
            for (size_t p = 0; p < list.size();) { devices.push_back(std::atoi(list.c_str() + p)); p = list.find(',', p); if (p == std::string::npos) break; ++p; }
        }
        else { std::cerr << "Unknown option '" << argv[i] << "'\n"; printUsageAndExit(argv[0]); }
    }
    if (outfile.empty()) printUsageAndExit(argv[0]);
    MiptGroup* group = nullptr;
    try {
        if (devices.empty()) {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw sutil::Exception(MIPT_ERROR_NO_DEVICE, "no HIP device: MIPT_ERROR_NO_DEVICE");
            for (int d = 0; d < n; ++d) devices.push_back(d);
        }
        buildSceneTables();
        const int rc = mipt_group_create(devices.data(), (uint32_t)devices.size(), &context_log_cb, nullptr, 4, &group);
        if (rc != MIPT_SUCCESS) throw sutil::Exception(rc, std::string("mipt_group_create failed: ") + mipt_error_name(rc));
        std::vector<PathTracerState> states(devices.size());
        for (size_t i = 0; i < states.size(); ++i) {                        // createContext / buildMeshAccel / createSBT / initLaunchParams per device
            void* st = nullptr;
            MIPT_CHECK(nullptr, mipt_group_member(group, (uint32_t)i, &states[i].context, &states[i].device_idx, &st));
            states[i].stream = (hipStream_t)st;
            buildMeshAccel(states[i]);
            createSBT(states[i]);
            initLaunchParams(states[i]);
            MIPT_CHECK(states[i].context, mipt_set_option(states[i].context, MIPT_OPT_MAX_DEPTH, max_depth));
        }
        // the assembled frame lives on device 0 (the sample's single output buffer)
        HIP_CHECK(hipSetDevice(devices[0]));
        float4* d_accum = nullptr;
        HIP_CHECK(hipMalloc((void**)&d_accum, (size_t)width * height * sizeof(float4)));
        sutil::HIPOutputBuffer<uchar4> output_buffer(sutil::HIPOutputBufferType::HIP_DEVICE, width, height);
        output_buffer.setDevice(devices[0]);
        output_buffer.setStream(states[0].stream);
        std::vector<MiptParams> params(states.size());
        std::vector<void*> accum(states.size(), nullptr), frame(states.size(), nullptr);
        for (int s = 0; s < subframes; ++s) {                              // launchSubframe, :562-594
            for (size_t i = 0; i < states.size(); ++i) { states[i].params.subframe_index = (unsigned)s; std::memcpy(&params[i], &states[i].params, sizeof(MiptParams)); }
            accum[0] = d_accum; frame[0] = output_buffer.map();
            const int r1 = mipt_group_launch(group, params.data(), (unsigned)width, (unsigned)height);
            if (r1 != MIPT_SUCCESS) throw sutil::Exception(r1, std::string("mipt_group_launch: ") + mipt_group_last_error_string(group));
            const int r2 = mipt_group_allgather(group, accum.data(), frame.data());
            if (r2 != MIPT_SUCCESS) throw sutil::Exception(r2, std::string("mipt_group_allgather: ") + mipt_group_last_error_string(group));
            output_buffer.unmap();
            const int r3 = mipt_group_sync(group);
            if (r3 != MIPT_SUCCESS) throw sutil::Exception(r3, std::string("mipt_group_sync: ") + mipt_group_last_error_string(group));
        }
        std::cerr << devices.size() << " device(s), exchange through " << (mipt_group_uses_rccl(group) ? "RCCL" : "a device copy") << "\n";
        sutil::ImageBuffer buffer;
        buffer.data = output_buffer.getHostPointer();
        buffer.width = output_buffer.width(); buffer.height = output_buffer.height(); buffer.pixel_format = sutil::UNSIGNED_BYTE4;
        sutil::saveImage(outfile.c_str(), buffer, false);
        HIP_CHECK(hipFree(d_accum));
        for (PathTracerState& st : states) { HIP_CHECK(hipSetDevice(st.device_idx)); if (st.d_vertices) HIP_CHECK(hipFree(st.d_vertices)); }
        mipt_group_destroy(group);
    } catch (std::exception& e) {
        std::cerr << "Caught exception: " << e.what() << "\n";
        if (group) mipt_group_destroy(group);
        return 1;
    }
    return 0;
}
