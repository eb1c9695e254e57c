// mipt_meshviewer -- C++ host of the glTF viewer above the libmipt C ABI, written against include/mipt_sutil.hpp.
//
// Mirrors SDK/optixMeshViewer/optixMeshViewer.cpp in file-output mode (main :347-486): sutil::loadScene + scene.finalize(),
// initCameraState :326, initLaunchParams :190-232 (the two point lights placed from the scene box, miss colour 0.1),
// handleCameraUpdate :235-247, launchSubframe :283-308, sutil::saveImage.  Flags as the sample: --model <file.gltf>, --file | -f,
// --dim=WxH, --no-gl-interop (accepted, there is no GL here).  Extra flags: --subframes N, --zero-copy (CUDAOutputBufferType::
// ZERO_COPY), --flattened / --two-level (one BVH over world-space triangles / per-mesh structures under an instance structure,
// the reference's layout; default: sutil::Scene's AUTO -- flattened for scenes of several instances up to 4 M triangles, else two-level).
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "../include/mipt_sutil.hpp"

namespace {

bool    camera_changed = true;
sutil::Camera camera;
int32_t width = 1024, height = 768;                                        // optixMeshViewer.cpp:84-85
whitted::LaunchParams params = {};
hipStream_t stream = nullptr;

void context_log_cb(unsigned int level, const char* tag, const char* message, void*)
{
    std::cerr << "[" << std::setw(2) << level << "][" << std::setw(12) << tag << "]: " << message << "\n";
}

void initLaunchParams(const sutil::Scene& scene)                            // :190-232
{
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&params.accum_buffer), (size_t)width * height * sizeof(float4)));
    params.frame_buffer = nullptr;                                          // set when the output buffer is mapped
    params.width = (unsigned)width; params.height = (unsigned)height;      // the sample's handleResize sets these (:250-271)
    params.subframe_index = 0u;
    const float loffset = scene.aabb().maxExtent();
    std::vector<Light> lights(2);
    lights[0].type = Light::Type::POINT;
    lights[0].point.color = make_float3(1.0f, 1.0f, 0.8f);
    lights[0].point.intensity = 5.0f;
    lights[0].point.position = sutil::vadd(scene.aabb().center(), make_float3(loffset, loffset, loffset));
    lights[0].point.falloff = Light::Falloff::QUADRATIC;
    lights[1].type = Light::Type::POINT;
    lights[1].point.color = make_float3(0.8f, 0.8f, 1.0f);
    lights[1].point.intensity = 3.0f;
    lights[1].point.position = sutil::vadd(scene.aabb().center(), make_float3(-loffset, 0.5f * loffset, -0.5f * loffset));
    lights[1].point.falloff = Light::Falloff::QUADRATIC;
    params.lights.count = static_cast<uint32_t>(lights.size());
    void* d_lights = nullptr;
    HIP_CHECK(hipMalloc(&d_lights, lights.size() * sizeof(Light)));
    HIP_CHECK(hipMemcpy(d_lights, lights.data(), lights.size() * sizeof(Light), hipMemcpyHostToDevice));
    params.lights.data = reinterpret_cast<unsigned long long>(d_lights);
    params.lights.byte_stride = 0; params.lights.elmt_byte_size = sizeof(Light);
    params.miss_color = make_float3(0.1f, 0.1f, 0.1f);
    params.max_depth = (int)whitted::MAX_TRACE_DEPTH;
    params.scene_epsilon = 1e-4f;
    HIP_CHECK(hipStreamCreate(&stream));
    params.handle = scene.traversableHandle();
}

void handleCameraUpdate(whitted::LaunchParams& p)                           // :235-247
{
    if (!camera_changed) return;
    camera_changed = false;
    camera.setAspectRatio(static_cast<float>(width) / static_cast<float>(height));
    p.eye = camera.eye();
    camera.UVWFrame(p.U, p.V, p.W);
}

void updateState(whitted::LaunchParams& p)                                  // :274-281
{
    if (camera_changed) p.subframe_index = 0;
    handleCameraUpdate(p);
}

void launchSubframe(sutil::HIPOutputBuffer<uchar4>& output_buffer, const sutil::Scene& scene)       // :283-308
{
    uchar4* result_buffer_data = output_buffer.map();
    params.frame_buffer = result_buffer_data;
    output_buffer.setStream(stream);
    MIPT_CHECK(scene.context(), mipt_whitted_launch(scene.context(), stream, reinterpret_cast<const MiptWhittedLaunchParams*>(&params), (unsigned)width, (unsigned)height, 1));
    output_buffer.unmap();
    HIP_SYNC_CHECK();
}

void initCameraState(const sutil::Scene& scene)                             // :326-335
{
    camera = scene.camera();
    camera_changed = true;
}

void cleanup()                                                              // :338-343
{
    HIP_CHECK(hipFree(reinterpret_cast<void*>(params.accum_buffer)));
    HIP_CHECK(hipFree(reinterpret_cast<void*>(params.lights.data)));
    if (stream) HIP_CHECK(hipStreamDestroy(stream));
}

[[noreturn]] void printUsageAndExit(const char* argv0)
{
    std::cerr << "Usage  : " << argv0 << " [options]\n"
              << "Options: --file | -f <filename>      File for image output (required: there is no display path)\n"
              << "         --model <model.gltf>        Specify model to render (required)\n"
              << "         --dim=<width>x<height>      Set image dimensions; defaults to 1024x768\n"
              << "         --no-gl-interop             accepted for compatibility\n"
              << "         --subframes <n>   --zero-copy   --flattened | --two-level\n"
              << "         --help | -h                 Print this usage message\n";
    std::exit(0);
}

}  // namespace

int main(int argc, char* argv[])
{
    sutil::HIPOutputBufferType output_buffer_type = sutil::HIPOutputBufferType::HIP_DEVICE;
    sutil::Scene::AccelMode accel_mode = sutil::Scene::AccelMode::AUTO;
    std::string outfile, infile;
    int subframes = 1;
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg == "--help" || arg == "-h") printUsageAndExit(argv[0]);
        else if (arg == "--no-gl-interop") output_buffer_type = sutil::HIPOutputBufferType::HIP_DEVICE;
        else if (arg == "--zero-copy") output_buffer_type = sutil::HIPOutputBufferType::ZERO_COPY;
        else if (arg == "--flattened") accel_mode = sutil::Scene::AccelMode::FLATTENED;
        else if (arg == "--two-level") accel_mode = sutil::Scene::AccelMode::TWO_LEVEL;
        else if (arg == "--model" && i < argc - 1) infile = argv[++i];
        else if ((arg == "--file" || arg == "-f") && i < argc - 1) outfile = argv[++i];
        else if (arg == "--subframes" && i < argc - 1) subframes = std::atoi(argv[++i]);
        else if (arg.substr(0, 6) == "--dim=") {
            int w = 0, h = 0;
            if (std::sscanf(arg.c_str() + 6, "%dx%d", &w, &h) != 2 || w <= 0 || h <= 0) printUsageAndExit(argv[0]);
            width = w; height = h;
        } else { std::cerr << "Unknown option '" << argv[i] << "'\n"; printUsageAndExit(argv[0]); }
    }
    if (infile.empty()) { std::cerr << "--model argument required" << std::endl; printUsageAndExit(argv[0]); }
    if (outfile.empty()) printUsageAndExit(argv[0]);
    try {
        sutil::Scene scene;
        scene.setLogCallback(&context_log_cb, nullptr, 4);
        scene.setAccelMode(accel_mode);
        sutil::loadScene(infile.c_str(), scene);
        scene.finalize();
        initCameraState(scene);
        initLaunchParams(scene);
        {
            sutil::HIPOutputBuffer<uchar4> output_buffer(output_buffer_type, width, height);
            for (int s = 0; s < subframes; ++s) {
                updateState(params);
                launchSubframe(output_buffer, scene);
                ++params.subframe_index;
            }
            MiptLaunchStats st{};
            MIPT_CHECK(scene.context(), mipt_get_launch_stats(scene.context(), &st));
            std::cerr << (scene.twoLevel() ? "two-level" : "flattened") << " structure; last launch: " << st.closest_rays << " radiance + "
                      << st.shadow_rays << " occlusion rays\n";
            sutil::ImageBuffer buffer;
            buffer.data = output_buffer.getHostPointer();
            buffer.width = output_buffer.width(); buffer.height = output_buffer.height();
            buffer.pixel_format = sutil::UNSIGNED_BYTE4;
            sutil::saveImage(outfile.c_str(), buffer, false);
        }
        cleanup();
    } catch (std::exception& e) {
        std::cerr << "Caught exception: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
