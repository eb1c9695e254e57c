// mipt_sutil.hpp -- the C++ host API the reference's samples are written against, kept over libmipt's C ABI (SURVEY.md 8b
// "signatures kept bit-for-bit").  Header only, C++17, needs <hip/hip_runtime.h> and include/mipt.h; link libmipt.so.
//
//   sutil::Exception, MIPT_CHECK / HIP_CHECK / HIP_SYNC_CHECK     SDK/sutil/Exception.h:82-225 (OPTIX_CHECK / CUDA_CHECK / CUDA_SYNC_CHECK)
//   sutil::Camera                                                  SDK/sutil/Camera.h:38-76 (the fork's m_fod included, :64,75)
//   sutil::Aabb  (the members the samples call)                    SDK/sutil/Aabb.h
//   sutil::HIPOutputBuffer<PIXEL_FORMAT>, HIPOutputBufferType      SDK/sutil/CUDAOutputBuffer.h:45-94, 203-290 (CUDA_DEVICE and
//                                                                  ZERO_COPY; GL_INTEROP / CUDA_P2P have no counterpart: no GL here)
//   sutil::Scene, sutil::loadScene                                 SDK/sutil/Scene.h:53-159 (host data + finalize(): context, accels,
//                                                                  Whitted geometry tables; pipeline / SBT members do not exist)
//   whitted::LaunchParams, Light, BufferView<T>                    SDK/cuda/whitted.h:59-77, cuda/Light.h:33-71, cuda/BufferView.h:33-49
//   sutil::saveImage / ImageBuffer / BufferImageFormat             SDK/sutil/sutil.h:53-71, sutil.cpp:542-709
//
// Method names, argument meaning and error behaviour (exceptions carrying the failing call) follow the reference; the bodies
// are this project's.  File:line citations are relative to /root/reference.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "mipt.h"

namespace sutil {

// ---------------------------------------------------------------------------------------------------------- errors
class Exception : public std::runtime_error {
public:
    explicit Exception(const char* msg) : std::runtime_error(msg) {}
    explicit Exception(const std::string& msg) : std::runtime_error(msg) {}
    Exception(int mipt_result, const std::string& msg) : std::runtime_error(msg), m_result(mipt_result) {}
    int result() const { return m_result; }          // the MiptResult behind it (0 when the failure was not a library call)
private:
    int m_result = 0;
};

inline void miptCheck(int res, const MiptContext* ctx, const char* call, const char* file, unsigned int line)
{
    if (res == MIPT_SUCCESS) return;
    std::stringstream ss;
    ss << "mipt call '" << call << "' failed: " << mipt_error_name(res);
    if (ctx) ss << ": " << mipt_last_error_string(ctx);
    ss << " (" << file << ":" << line << ")\n";
    throw Exception(res, ss.str());
}
inline void hipCheck(hipError_t error, const char* call, const char* file, unsigned int line)
{
    if (error == hipSuccess) return;
    std::stringstream ss;
    ss << "HIP call (" << call << " ) failed with error: '" << hipGetErrorString(error) << "' (" << file << ":" << line << ")\n";
    throw Exception(ss.str());
}
inline void hipSyncCheck(const char* file, unsigned int line)
{
    (void)hipDeviceSynchronize();
    const hipError_t error = hipGetLastError();
    if (error == hipSuccess) return;
    std::stringstream ss;
    ss << "HIP error on synchronize with error '" << hipGetErrorString(error) << "' (" << file << ":" << line << ")\n";
    throw Exception(ss.str());
}

}  // namespace sutil

#define MIPT_CHECK(ctx, call) ::sutil::miptCheck((call), (ctx), #call, __FILE__, __LINE__)
#define HIP_CHECK(call) ::sutil::hipCheck((call), #call, __FILE__, __LINE__)
#define HIP_SYNC_CHECK() ::sutil::hipSyncCheck(__FILE__, __LINE__)

namespace sutil {

// ---------------------------------------------------------------------------------------------------------- small vector helpers
// (float3 / float4 / uchar4 and make_float3 come from the HIP headers; the operation order is vec_math.h's)
inline float3 vsub(const float3& a, const float3& b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 vadd(const float3& a, const float3& b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 vscale(const float3& a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
inline float  vdot(const float3& a, const float3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float  vlength(const float3& a) { return std::sqrt(vdot(a, a)); }
inline float3 vnormalize(const float3& a) { return vscale(a, 1.0f / std::sqrt(vdot(a, a))); }

// ---------------------------------------------------------------------------------------------------------- Camera (Camera.h:38-76)
class Camera {
public:
    Camera() : m_eye(make_float3(1.0f, 1.0f, 1.0f)), m_lookat(make_float3(0.0f, 0.0f, 0.0f)), m_up(make_float3(0.0f, 1.0f, 0.0f)), m_fovY(35.0f), m_aspectRatio(1.0f) {}
    Camera(const float3& eye, const float3& lookat, const float3& up, float fovY, float aspectRatio)
        : m_eye(eye), m_lookat(lookat), m_up(up), m_fovY(fovY), m_aspectRatio(aspectRatio) {}

    float3 direction() const { return vnormalize(vsub(m_lookat, m_eye)); }
    void   setDirection(const float3& dir) { m_lookat = vadd(m_eye, vscale(dir, vlength(vsub(m_lookat, m_eye)))); }

    const float3& eye() const { return m_eye; }
    void          setEye(const float3& val) { m_eye = val; }
    const float3& lookat() const { return m_lookat; }
    void          setLookat(const float3& val) { m_lookat = val; }
    const float3& up() const { return m_up; }
    void          setUp(const float3& val) { m_up = val; }
    const float&  fovY() const { return m_fovY; }
    void          setFovY(const float& val) { m_fovY = val; }
    const float&  aspectRatio() const { return m_aspectRatio; }
    void          setAspectRatio(const float& val) { m_aspectRatio = val; }
    void          setFocalLength(float val) { m_fod = val; }

    // UVW forms an orthogonal, but not orthonormal basis (Camera.cpp:34-46; computed by the library's host code)
    void UVWFrame(float3& U, float3& V, float3& W) const
    {
        mipt_camera_uvw(&m_eye.x, &m_lookat.x, &m_up.x, m_fovY, m_aspectRatio, m_fod, &U.x, &V.x, &W.x);
    }

private:
    float3 m_eye, m_lookat, m_up;
    float  m_fovY, m_aspectRatio;
    float  m_fod = 1.0f;
};

// ---------------------------------------------------------------------------------------------------------- Aabb (Aabb.h)
struct Aabb {
    float3 m_min = make_float3(1e37f, 1e37f, 1e37f), m_max = make_float3(-1e37f, -1e37f, -1e37f);
    bool   valid() const { return m_min.x <= m_max.x && m_min.y <= m_max.y && m_min.z <= m_max.z; }
    float3 center() const { return vscale(vadd(m_min, m_max), 0.5f); }
    float3 extent() const { return vsub(m_max, m_min); }
    float  maxExtent() const { const float3 e = extent(); return std::fmax(std::fmax(e.x, e.y), e.z); }
};

// ---------------------------------------------------------------------------------------------------------- output buffer
enum class HIPOutputBufferType {
    HIP_DEVICE = 0,   // CUDA_DEVICE: device memory, getHostPointer() copies
    GL_INTEROP = 1,   // not available (no GL in this build): the constructor throws
    ZERO_COPY  = 2,   // pinned, mapped host memory the kernels write through the fabric; getHostPointer() is the buffer itself
    HIP_P2P    = 3    // CUDA_P2P: treated as HIP_DEVICE
};

template <typename PIXEL_FORMAT>
class HIPOutputBuffer {
public:
    HIPOutputBuffer(HIPOutputBufferType type, int32_t width, int32_t height) : m_type(type)
    {
        if (type == HIPOutputBufferType::GL_INTEROP) throw Exception("HIPOutputBuffer: GL_INTEROP is not available (no GL display path); use ZERO_COPY or HIP_DEVICE");
        resize(width, height);
    }
    ~HIPOutputBuffer()
    {
        if (m_type == HIPOutputBufferType::ZERO_COPY) { if (m_host_zcopy_pixels) (void)hipHostFree(m_host_zcopy_pixels); }
        else if (m_device_pixels) (void)hipFree(m_device_pixels);
    }
    HIPOutputBuffer(const HIPOutputBuffer&) = delete;
    HIPOutputBuffer& operator=(const HIPOutputBuffer&) = delete;

    void setDevice(int32_t device_idx) { m_device_idx = device_idx; }
    void setStream(hipStream_t stream) { m_stream = stream; }

    void resize(int32_t width, int32_t height)
    {
        if (width < 1) width = 1;                                       // ensureMinimumSize (sutil.cpp)
        if (height < 1) height = 1;
        if (m_width == width && m_height == height) return;
        m_width = width; m_height = height;
        makeCurrent();
        const size_t bytes = (size_t)m_width * m_height * sizeof(PIXEL_FORMAT);
        if (m_type == HIPOutputBufferType::ZERO_COPY) {
            if (m_host_zcopy_pixels) HIP_CHECK(hipHostFree(reinterpret_cast<void*>(m_host_zcopy_pixels)));
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m_host_zcopy_pixels), bytes, hipHostMallocPortable | hipHostMallocMapped));
            HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&m_device_pixels), reinterpret_cast<void*>(m_host_zcopy_pixels), 0));
        } else {
            if (m_device_pixels) HIP_CHECK(hipFree(reinterpret_cast<void*>(m_device_pixels)));
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m_device_pixels), bytes));
        }
        if (!m_host_pixels.empty()) m_host_pixels.resize((size_t)m_width * m_height);
    }

    PIXEL_FORMAT* map() { return m_device_pixels; }                    // nothing to do for either mode
    void unmap() { makeCurrent(); HIP_CHECK(hipStreamSynchronize(m_stream)); }

    int32_t width() const { return m_width; }
    int32_t height() const { return m_height; }

    PIXEL_FORMAT* getHostPointer()
    {
        if (m_type == HIPOutputBufferType::ZERO_COPY) return m_host_zcopy_pixels;
        m_host_pixels.resize((size_t)m_width * m_height);
        makeCurrent();
        HIP_CHECK(hipMemcpy(static_cast<void*>(m_host_pixels.data()), map(), (size_t)m_width * m_height * sizeof(PIXEL_FORMAT), hipMemcpyDeviceToHost));
        unmap();
        return m_host_pixels.data();
    }

private:
    void makeCurrent() { HIP_CHECK(hipSetDevice(m_device_idx)); }

    HIPOutputBufferType       m_type;
    int32_t                   m_width = 0, m_height = 0;
    PIXEL_FORMAT*             m_device_pixels = nullptr;
    PIXEL_FORMAT*             m_host_zcopy_pixels = nullptr;
    std::vector<PIXEL_FORMAT> m_host_pixels;
    hipStream_t               m_stream = nullptr;
    int32_t                   m_device_idx = 0;
};

// ---------------------------------------------------------------------------------------------------------- images (sutil.h:53-71)
enum BufferImageFormat { UNSIGNED_BYTE4 = MIPT_IMAGE_UNSIGNED_BYTE4, FLOAT4 = MIPT_IMAGE_FLOAT4, FLOAT3 = MIPT_IMAGE_FLOAT3 };
struct ImageBuffer {
    void*             data = nullptr;
    unsigned int      width = 0, height = 0;
    BufferImageFormat pixel_format = UNSIGNED_BYTE4;
};
inline void saveImage(const char* filename, const ImageBuffer& buffer, bool disable_srgb_conversion)
{
    char err[512] = { 0 };
    const int rc = mipt_save_image(filename, buffer.data, buffer.width, buffer.height, (int)buffer.pixel_format, disable_srgb_conversion ? 1 : 0, err, sizeof err);
    if (rc != MIPT_SUCCESS) throw Exception(rc, err[0] ? err : "saveImage failed");
}

}  // namespace sutil

// ---------------------------------------------------------------------------------------------------------- whitted launch structs
template <typename T>
struct BufferView {                                                     // cuda/BufferView.h:33-49 (16 bytes)
    unsigned long long data = 0;
    unsigned int       count = 0;
    unsigned short     byte_stride = 0;
    unsigned short     elmt_byte_size = 0;
};
struct Light {                                                          // cuda/Light.h:33-71 (36 bytes), POINT / AMBIENT
    enum class Falloff : int { NONE = 0, LINEAR, QUADRATIC };
    enum class Type : int { POINT = 0, AMBIENT = 1 };
    struct Point { float3 color; float intensity; float3 position; Falloff falloff; };
    struct Ambient { float3 color; };
    Type type;
    union { Point point; Ambient ambient; };
    Light() : type(Type::POINT) { std::memset(&point, 0, sizeof point); point.color = make_float3(1.0f, 1.0f, 1.0f); point.intensity = 1.0f; point.falloff = Falloff::QUADRATIC; }
};
static_assert(sizeof(Light) == sizeof(MiptLight), "struct Light must stay layout compatible with cuda/Light.h");
namespace whitted {
const unsigned int MAX_TRACE_DEPTH = 8u;                                // cuda/whitted.h:42
struct LaunchParams {                                                   // cuda/whitted.h:59-77 (128 bytes)
    unsigned int       width, height, subframe_index;
    float4*            accum_buffer;
    uchar4*            frame_buffer;
    int                max_depth;
    float              scene_epsilon;
    float3             eye, U, V, W;
    BufferView<Light>  lights;
    float3             miss_color;
    unsigned long long handle;
};
static_assert(sizeof(LaunchParams) == sizeof(MiptWhittedLaunchParams), "whitted::LaunchParams must stay layout compatible");
}  // namespace whitted

namespace sutil {

// ---------------------------------------------------------------------------------------------------------- Scene (Scene.h:53-159)
// loadScene parses the glTF (mipt_scene_load); finalize() does what Scene::finalize does (Scene.cpp:803-815): creates the
// context, uploads the geometry, builds the acceleration structures -- one triangle structure per mesh plus the instance
// structure over them (buildMeshAccels + buildInstanceAccel), or ONE structure over the flattened world-space triangles when
// asked for or, by default (AccelMode::AUTO), when the scene has several instances and at most 4 M flattened triangles (FLATTENED: 1.2-1.3 x faster
// over MANY instances at twenty times the memory; a scene of ONE instance -- most glTF files -- is walked as its mesh's own BVH either way) -- and hands the Whitted programs their geometry (per-mesh
// object-space tables + instance records for the two-level form, as the reference's createSBT keeps them; flattened world-space
// arrays otherwise), material and texture tables.  The programs are fixed-function kernels: pipeline() / sbt() do not exist.
class Scene {
public:
    enum class AccelMode { AUTO, TWO_LEVEL, FLATTENED };
    static constexpr uint32_t kAutoFlattenTriangles = 4u << 20;        // ~0.6 GB of structure and shading tables

    Scene() = default;
    ~Scene() { cleanup(); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    void setDevice(int device_idx) { m_device = device_idx; }
    void setAccelMode(AccelMode m) { m_mode = m; }
    void setLogCallback(MiptLogCallback cb, void* cbdata, int level) { m_cb = cb; m_cbdata = cbdata; m_log_level = level; }

    Camera camera() const                                              // Scene::camera(), Scene.cpp:777-791
    {
        float eye[3], lookat[3], up[3], fovY, aspect;
        MIPT_CHECK(nullptr, mipt_scene_camera(m_scene, eye, lookat, up, &fovY, &aspect));
        return Camera(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]), make_float3(up[0], up[1], up[2]), fovY, aspect);
    }
    Aabb aabb() const
    {
        Aabb b;
        b.m_min = make_float3(m_info.aabb_min[0], m_info.aabb_min[1], m_info.aabb_min[2]);
        b.m_max = make_float3(m_info.aabb_max[0], m_info.aabb_max[1], m_info.aabb_max[2]);
        return b;
    }
    unsigned long long                   traversableHandle() const { return m_handle; }
    MiptContext*                         context() const { return m_context; }
    const std::vector<MiptMaterialData>& materials() const { return m_materials; }
    const MiptSceneInfo&                 info() const { return m_info; }
    bool                                 twoLevel() const { return m_two_level; }
    const MiptScene*                     hostScene() const { return m_scene; }

    void finalize()
    {
        if (!m_scene) throw Exception("Scene::finalize: no scene loaded");
        createContext();
        // AUTO.  The reference builds the instance structure for every scene (Scene.cpp:803-815).  Here a scene of ONE instance is walked as its
        // mesh's own BVH either way, and a scene of MANY instances renders 1.2-1.3 x faster flattened, at the price of one triangle record per
        // instance and triangle (twenty times the memory of the measured scenes) -- which a 288 GB card pays without noticing up to a few million
        // triangles: AUTO flattens scenes of several instances whose flattened form has at most kAutoFlattenTriangles, and builds the instance
        // structure for everything else.  TWO_LEVEL / FLATTENED pin the choice.
        m_two_level = m_mode == AccelMode::TWO_LEVEL ||
                      (m_mode == AccelMode::AUTO && (m_info.num_instances <= 1u || m_info.num_triangles > kAutoFlattenTriangles));
        if (m_two_level) { buildMeshAccels(); buildInstanceAccel(); createInstancedTables(); }
        else             { uploadGeometry(); buildFlattenedAccel(); createGeometryTables(); }
    }

    void cleanup()
    {
        if (m_context) {
            (void)hipSetDevice(m_device);
            (void)mipt_context_destroy(m_context);
            m_context = nullptr;
        }
        for (void* p : m_device_allocs) if (p) (void)hipFree(p);
        m_device_allocs.clear();
        if (m_scene) { (void)mipt_scene_destroy(m_scene); m_scene = nullptr; }
        m_handle = 0;
    }

    void createContext()
    {
        if (m_context) return;
        const int rc = mipt_context_create(m_device, m_cb, m_cbdata, m_log_level, &m_context);
        if (rc != MIPT_SUCCESS) throw Exception(rc, std::string("mipt_context_create failed: ") + mipt_error_name(rc));
    }

private:
    friend void loadScene(const std::string& filename, Scene& scene);

    template <typename T> T* upload(const std::vector<T>& host)
    {
        void* d = nullptr;
        HIP_CHECK(hipMalloc(&d, host.empty() ? 16 : host.size() * sizeof(T)));
        m_device_allocs.push_back(d);
        if (!host.empty()) HIP_CHECK(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
        return static_cast<T*>(d);
    }

    // OptixGeometryFlags per material as Scene::buildMeshAccels picks them (Scene.cpp:904-966)
    std::vector<uint32_t> geometryFlags() const
    {
        std::vector<uint32_t> f;
        for (const MiptMaterialData& m : m_materials) {
            const uint32_t base = m.alpha_mode == 0 ? MIPT_GEOMETRY_FLAG_DISABLE_ANYHIT : (m.alpha_mode == 1 ? MIPT_GEOMETRY_FLAG_NONE : MIPT_GEOMETRY_FLAG_REQUIRE_SINGLE_ANYHIT_CALL);
            f.push_back(base | (m.double_sided ? (uint32_t)MIPT_GEOMETRY_FLAG_DISABLE_TRIANGLE_FACE_CULLING : 0u));
        }
        return f;
    }

    void uploadGeometry()
    {
        HIP_CHECK(hipSetDevice(m_device));
        const size_t n = m_info.num_triangles;
        if (n == 0) throw Exception("scene has no triangles");
        std::vector<float>    v(n * 12), nrm(n * 12);
        std::vector<uint32_t> mat(n);
        MIPT_CHECK(nullptr, mipt_scene_flatten(m_scene, v.data(), mat.data()));
        MIPT_CHECK(nullptr, mipt_scene_flatten_normals(m_scene, nrm.data()));
        m_flat_material = mat;
        m_d_vertices = upload(v);
        bool has_normals = false;
        for (size_t i = 3; i < nrm.size(); i += 4) has_normals = has_normals || nrm[i] != 0.0f;
        m_d_normals = has_normals ? upload(nrm) : nullptr;
        std::vector<uint32_t> mi = mat;
        m_d_material_index = upload(mi);
        for (uint32_t set = 0; set < 2; ++set) {
            std::vector<float> uv(n * 6); std::vector<uint8_t> present(n);
            MIPT_CHECK(nullptr, mipt_scene_flatten_texcoords(m_scene, set, uv.data(), present.data()));
            bool any = false;
            for (uint8_t p : present) any = any || p != 0;
            m_d_texcoords[set] = any ? upload(uv) : nullptr;
            m_d_texcoord_present[set] = any ? upload(present) : nullptr;
        }
        std::vector<float> col(n * 12); std::vector<uint8_t> cpresent(n);
        MIPT_CHECK(nullptr, mipt_scene_flatten_colors(m_scene, col.data(), cpresent.data()));
        bool any = false;
        for (uint8_t p : cpresent) any = any || p != 0;
        m_d_colors = any ? upload(col) : nullptr;
        m_d_color_present = any ? upload(cpresent) : nullptr;
    }

    std::vector<uint32_t> clampMaterials(const std::vector<uint32_t>& mat) const     // "no material" (0xffffffff) selects the last table entry
    {
        std::vector<uint32_t> out(mat.size());
        const uint32_t last = (uint32_t)m_materials.size() - 1u;
        for (size_t i = 0; i < mat.size(); ++i) out[i] = mat[i] < last ? mat[i] : last;
        return out;
    }

    void buildFlattenedAccel()
    {
        const std::vector<uint32_t> flags = geometryFlags(), sbt = clampMaterials(m_flat_material);
        MiptBuildInput in{};
        in.vertices = m_d_vertices; in.num_vertices = m_info.num_triangles * 3u; in.vertex_stride_bytes = 16;
        in.sbt_index_offsets = sbt.data(); in.flags = flags.data(); in.num_sbt_records = (uint32_t)flags.size();
        MIPT_CHECK(m_context, mipt_accel_build(m_context, &in, 1, &m_handle));
    }

    void buildMeshAccels()                                              // Scene.cpp:817-1131: one triangle structure per mesh, object space
    {
        const std::vector<uint32_t> flags = geometryFlags();
        m_mesh_handles.assign(m_info.num_meshes, 0ull);
        m_mesh_tables.assign(m_info.num_meshes, MiptWhittedMesh{});
        for (uint32_t m = 0; m < m_info.num_meshes; ++m) {
            uint32_t n = 0;
            MIPT_CHECK(nullptr, mipt_scene_mesh_num_triangles(m_scene, m, &n));
            if (n == 0) continue;                                       // a mesh without triangle primitives is never instanced below
            std::vector<float> v((size_t)n * 12); std::vector<uint32_t> mat(n);
            MIPT_CHECK(nullptr, mipt_scene_mesh_triangles(m_scene, m, v.data(), mat.data()));
            float* dv = upload(v);
            m_mesh_tables[m].geometry.vertices4 = dv; m_mesh_tables[m].num_triangles = n;
            m_mesh_tables[m].geometry.material_index = upload(mat);
            const std::vector<uint32_t> sbt = clampMaterials(mat);
            MiptBuildInput in{};
            in.vertices = dv; in.num_vertices = n * 3u; in.vertex_stride_bytes = 16;
            in.sbt_index_offsets = sbt.data(); in.flags = flags.data(); in.num_sbt_records = (uint32_t)flags.size();
            MIPT_CHECK(m_context, mipt_accel_build(m_context, &in, 1, &m_mesh_handles[m]));
        }
    }

    void buildInstanceAccel()                                           // Scene.cpp:1134-1212
    {
        std::vector<MiptInstance> inst;
        for (uint32_t i = 0; i < m_info.num_instances; ++i) {
            uint32_t mesh = 0;
            MiptInstance oi;
            std::memset(&oi, 0, sizeof oi);
            MIPT_CHECK(nullptr, mipt_scene_instance(m_scene, i, &mesh, oi.transform));
            if (m_mesh_handles[mesh] == 0ull) throw Exception("Scene::buildInstanceAccel: an instance refers to a mesh without triangles");
            oi.flags = 0; oi.instanceId = i; oi.sbtOffset = 0; oi.visibilityMask = 1;
            oi.traversableHandle = m_mesh_handles[mesh];
            inst.push_back(oi);
        }
        MIPT_CHECK(m_context, mipt_accel_build_instances(m_context, inst.data(), (uint32_t)inst.size(), &m_handle));
    }

    // the data side of Scene::createSBT (Scene.cpp:1368-1480) as the reference keeps it: one set of OBJECT-space buffers per mesh, one
    // record per instance -- instancing saves shading memory as well as BVH memory (mipt_whitted_set_geometry_instanced)
    void createInstancedTables()
    {
        HIP_CHECK(hipSetDevice(m_device));
        if (m_info.num_triangles == 0) throw Exception("scene has no triangles");
        for (uint32_t m = 0; m < m_info.num_meshes; ++m) {
            const uint32_t n = m_mesh_tables[m].num_triangles;
            if (n == 0) continue;
            MiptWhittedGeometry& g = m_mesh_tables[m].geometry;
            std::vector<float> nrm((size_t)n * 12);
            MIPT_CHECK(nullptr, mipt_scene_mesh_normals(m_scene, m, nrm.data()));
            bool any = false;
            for (size_t i = 3; i < nrm.size(); i += 4) any = any || nrm[i] != 0.0f;
            g.normals4 = any ? upload(nrm) : nullptr;
            for (uint32_t set = 0; set < 2; ++set) {
                std::vector<float> uv((size_t)n * 6); std::vector<uint8_t> present(n);
                MIPT_CHECK(nullptr, mipt_scene_mesh_texcoords(m_scene, m, set, uv.data(), present.data()));
                any = false;
                for (uint8_t p : present) any = any || p != 0;
                g.texcoords[set] = any ? upload(uv) : nullptr;
                g.texcoord_present[set] = any ? upload(present) : nullptr;
            }
            std::vector<float> col((size_t)n * 12); std::vector<uint8_t> cpresent(n);
            MIPT_CHECK(nullptr, mipt_scene_mesh_colors(m_scene, m, col.data(), cpresent.data()));
            any = false;
            for (uint8_t p : cpresent) any = any || p != 0;
            g.colors4 = any ? upload(col) : nullptr;
            g.color_present = any ? upload(cpresent) : nullptr;
        }
        std::vector<MiptWhittedInstance> inst(m_info.num_instances);
        for (uint32_t i = 0; i < m_info.num_instances; ++i) MIPT_CHECK(nullptr, mipt_scene_instance(m_scene, i, &inst[i].mesh, inst[i].transform));
        std::vector<MiptTextureData> tex(m_info.num_textures);
        for (uint32_t t = 0; t < m_info.num_textures; ++t) MIPT_CHECK(nullptr, mipt_scene_texture(m_scene, t, &tex[t]));
        MIPT_CHECK(m_context, mipt_whitted_set_geometry_instanced(m_context, m_mesh_tables.data(), (uint32_t)m_mesh_tables.size(), inst.data(), (uint32_t)inst.size(),
                                                                  m_materials.data(), (uint32_t)m_materials.size(), tex.empty() ? nullptr : tex.data(), (uint32_t)tex.size()));
    }

    void createGeometryTables()                                         // ... and over flattened world-space arrays (AccelMode::FLATTENED)
    {
        std::vector<MiptTextureData> tex(m_info.num_textures);
        for (uint32_t t = 0; t < m_info.num_textures; ++t) MIPT_CHECK(nullptr, mipt_scene_texture(m_scene, t, &tex[t]));
        MiptWhittedGeometry geo{};
        geo.vertices4 = m_d_vertices; geo.normals4 = m_d_normals; geo.material_index = m_d_material_index;
        for (int j = 0; j < 2; ++j) { geo.texcoords[j] = m_d_texcoords[j]; geo.texcoord_present[j] = m_d_texcoord_present[j]; }
        geo.colors4 = m_d_colors; geo.color_present = m_d_color_present;
        MIPT_CHECK(m_context, mipt_whitted_set_geometry(m_context, &geo, m_materials.data(), (uint32_t)m_materials.size(), tex.empty() ? nullptr : tex.data(), (uint32_t)tex.size()));
    }

    MiptScene*                    m_scene = nullptr;
    MiptSceneInfo                 m_info{};
    std::vector<MiptMaterialData> m_materials;                          // the scene's materials + one default entry ("no material")
    std::vector<uint32_t>         m_flat_material;
    MiptContext*                  m_context = nullptr;
    MiptLogCallback               m_cb = nullptr; void* m_cbdata = nullptr; int m_log_level = 0;
    int                           m_device = 0;
    AccelMode                     m_mode = AccelMode::AUTO;
    bool                          m_two_level = false;
    unsigned long long            m_handle = 0;
    std::vector<unsigned long long> m_mesh_handles;
    std::vector<MiptWhittedMesh>  m_mesh_tables;                        // two-level scenes: per-mesh object-space shading tables
    std::vector<void*>            m_device_allocs;
    float *m_d_vertices = nullptr, *m_d_normals = nullptr, *m_d_texcoords[2] = { nullptr, nullptr }, *m_d_colors = nullptr;
    uint32_t*                     m_d_material_index = nullptr;
    uint8_t *m_d_texcoord_present[2] = { nullptr, nullptr }, *m_d_color_present = nullptr;
};

inline void loadScene(const std::string& filename, Scene& scene)        // Scene.cpp:267-550
{
    scene.cleanup();
    char err[512] = { 0 };
    const int rc = mipt_scene_load(filename.c_str(), &scene.m_scene, err, sizeof err);
    if (rc != MIPT_SUCCESS) throw Exception(rc, std::string("loadScene( ") + filename + " ): " + (err[0] ? err : mipt_error_name(rc)));
    MIPT_CHECK(nullptr, mipt_scene_info(scene.m_scene, &scene.m_info));
    scene.m_materials.resize(scene.m_info.num_materials);
    for (uint32_t i = 0; i < scene.m_info.num_materials; ++i) MIPT_CHECK(nullptr, mipt_scene_material(scene.m_scene, i, &scene.m_materials[i]));
    // primitives without a material use the defaults of MaterialData() (cuda/MaterialData.h:44-51): appended as the last entry
    MiptMaterialData def;
    std::memset(&def, 0, sizeof def);
    def.base_color[0] = def.base_color[1] = def.base_color[2] = def.base_color[3] = 1.0f;
    def.metallic = 1.0f; def.roughness = 1.0f;
    MiptTextureInfo none; std::memset(&none, 0, sizeof none);
    none.texture = -1; none.rotation[1] = 1.0f; none.scale[0] = none.scale[1] = 1.0f;
    def.base_color_tex = def.metallic_roughness_tex = def.normal_tex = def.emissive_tex = none;
    scene.m_materials.push_back(def);
}

}  // namespace sutil
