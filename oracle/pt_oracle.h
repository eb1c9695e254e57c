/*
 * pt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's path-tracer hot path
 * (SDK/optixPathTracer/optixPathTracer.cu:249-413 and the headers it uses),
 * plus a CPU BVH2 builder / traversal that stands in for the closed-source
 * OptiX traversal (optixTraverse, optixPathTracer.cu:184-197, 227-238).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libmipt.so) never links or calls it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - RNG, camera frame, sRGB quantisation, float3 helpers: PINNED by known
 *     answer vectors generated from the reference's own headers
 *     (oracle/ref_kat.cpp -> tests/golden/ref_kat.json).
 *   - raygen / closest-hit / miss control flow: restated from the reference
 *     source; the reference has no tests or golden images.
 *   - BVH traversal + ray/triangle test: PARITY UNPINNED -- the reference's
 *     arithmetic lives in libnvoptix / RT cores and is not in the tree.  The
 *     conventions used here are this build's own definition.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- launch parameters: same field order as `struct Params`
 *      (SDK/optixPathTracer/optixPathTracer.h:91-107), host pointers. ---- */
typedef struct OrcLight {
    float corner[3], v1[3], v2[3], normal[3], emission[3];
} OrcLight;

typedef struct OrcParams {
    uint32_t subframe_index;
    float*   accum_buffer;   /* width*height float4, row 0 = bottom */
    uint8_t* frame_buffer;   /* width*height uchar4 */
    uint32_t width, height, samples_per_launch;
    float    eye[3], U[3], V[3], W[3];
    OrcLight light;
    uint64_t handle;         /* unused by the oracle */
} OrcParams;

typedef struct OrcStats {
    uint64_t closest_rays;   /* primary + continuation rays   */
    uint64_t shadow_rays;    /* occlusion rays actually traced */
    uint64_t node_visits;    /* 64-byte inner nodes fetched (closest + shadow) */
    uint64_t tri_tests;      /* 48-byte triangles fetched   (closest + shadow) */
    uint64_t max_depth_seen; /* deepest bounce index reached   */
    uint64_t shadow_node_visits; /* the occlusion rays' share of node_visits */
    uint64_t shadow_tri_tests;   /* the occlusion rays' share of tri_tests   */
} OrcStats;

typedef struct OrcScene OrcScene;

enum { ORC_MODE_BRUTE = 0, ORC_MODE_BVH = 1, ORC_MODE_INST = 2 /* two-level: orc_scene_set_instances */ };

/* ---- scalar pieces (each pinned by a KAT) ---- */
uint32_t orc_tea4(uint32_t v0, uint32_t v1);
uint32_t orc_lcg(uint32_t* prev);
float    orc_rnd(uint32_t* prev);
void     orc_camera_uvw(const float eye[3], const float lookat[3], const float up[3],
                        float fovY_deg, float aspect, float fod,
                        float U[3], float V[3], float W[3]);
void     orc_make_color(const float c[3], uint8_t out[4]);
void     orc_vec_kat(const float a[3], const float b[3], float t, float out[17]);
void     orc_det_sincos(float x, float* s, float* c);
float    orc_det_pow_inv24(float x);
float    orc_det_pow22(float x);
void     orc_cosine_sample_hemisphere(float u1, float u2, float p[3]);
void     orc_onb_inverse_transform(const float n[3], float p[3]);
void     orc_work_pixel(int width, int height, int num_gpus, int gpu, int sample_idx, int* px, int* py);
int      orc_work_num_samples(int width, int height, int num_gpus);

/* ---- scene ---- */
OrcScene* orc_scene_create(const float* vertices4 /* n_tris*3 float4 */, uint32_t n_tris,
                           const uint32_t* mat_index /* n_tris or NULL */,
                           const float* emission3, const float* diffuse3, uint32_t n_mat,
                           const float bg[4]);
void      orc_scene_destroy(OrcScene*);
void      orc_scene_set_prim_flags(OrcScene*, const uint32_t* flags_per_prim);   /* OptixGeometryFlags; NULL clears */
/* own binned-SAH builder */
int       orc_scene_build_bvh(OrcScene*, int leaf_max, int max_depth);
/* adopt a BVH in the layout of include/mipt.h (MiptBvhNode / MiptBvhTri) */
int       orc_scene_set_bvh(OrcScene*, const void* nodes, uint32_t n_nodes,
                            const void* tris, uint32_t n_tris);
uint32_t  orc_scene_num_nodes(const OrcScene*);
int       orc_scene_bvh_depth(const OrcScene*);
/* adopt the product's two-level (instance) structure: the BVH over the instance boxes (n_tlas 64-byte nodes, leaves list
 * positions of `order`), order[n_inst] = instance index per leaf slot, per instance the world->object 3x4 matrix, the
 * flattened index of its primitive 0 and its mesh's BVH (MiptBvhNode / MiptBvhTri arrays).  The scene's vertex / material
 * arrays stay the flattened world-space ones.  Rendering with mode ORC_MODE_INST then walks this structure. */
int       orc_scene_set_instances(OrcScene*, const void* tlas_nodes, uint32_t n_tlas, const uint32_t* order, uint32_t n_inst,
                                  const float* inv12, const uint32_t* prim_base, const void* const* gas_nodes,
                                  const uint32_t* gas_n_nodes, const void* const* gas_tris, const uint32_t* gas_n_tris);
/* structural validation of the adopted/built BVH; 0 = ok */
int       orc_scene_validate_bvh(const OrcScene*);

/* ---- traversal: rays = n * 8 floats (ox,oy,oz,tmin, dx,dy,dz,tmax) ---- */
void orc_trace_closest(const OrcScene*, int mode, const float* rays, uint32_t n,
                       float* hit_t, uint32_t* hit_prim, float* hit_u, float* hit_v, OrcStats* stats);
void orc_trace_any(const OrcScene*, int mode, const float* rays, uint32_t n,
                   uint8_t* occluded, OrcStats* stats);

uint32_t orc_trace_profile(const OrcScene*, const float* ray8, int any, uint16_t* pairs, uint32_t max_pairs);

/* ---- full render of rows [y0,y1) (or of an explicit pixel list) ----
 * max_depth: 0 = unbounded (the reference's behaviour: Russian roulette only),
 *            k>0 = stop after k closest-hit queries per path (build extension). */
int orc_render_rows(const OrcScene*, int mode, const OrcParams* p, uint32_t max_depth,
                    uint32_t y0, uint32_t y1, OrcStats* stats);
int orc_render_pixels(const OrcScene*, int mode, const OrcParams* p, uint32_t max_depth,
                      const uint32_t* pixel_index, uint32_t n, float* compact_accum4,
                      const float* compact_prev4, OrcStats* stats);

/* ---- SDK/imgui_test programs (optixTriangle.cu:103-268), restated ----
 * Object layouts are the reference's device objects: Camera 92 B (camera.h:155-173), LightVariant 44 B
 * (light.h:10-51: 40-byte union + tag 0 Point / 1 Directional / 2 Volumetric), DiffuseMaterial 12 B. */
typedef struct OrcImguiCamera {
    float   eye[3], lookat[3], up[3];
    uint8_t ortho, pad_[3];
    float   fov, fd, aperture, speed;
    float   u[3], v[3], w[3];
} OrcImguiCamera;

typedef struct OrcImguiLight {
    float   a[3];        /* Point: position | Directional: direction | Volumetric: position */
    float   b[4];        /* Point: lumi[3], dark.x | Directional: lumi[3], jitter | Volumetric: radius, lumi[3] */
    float   c[3];        /* rest of the union */
    int32_t tag;
} OrcImguiLight;

typedef struct OrcImguiFrame {
    uint32_t image_width, image_height, samples_per_frame, dt;
    int32_t  dirty;
    const OrcImguiCamera* camera;
    const OrcImguiLight*  lights;    int32_t nlights;
    const float*          materials; int32_t nmaterials;   /* float3 colour each (diffuse.h:3-16) */
    const int32_t*        mat_indices;                     /* per primitive */
    const float*          normals;                         /* 3 packed float3 per primitive (triangle_gas.cpp:16-75) */
    float*                film;                            /* width*height float3, in/out */
    uint8_t*              image;                           /* width*height uchar4 */
} OrcImguiFrame;

void orc_imgui_camera_compute_uvw(OrcImguiCamera* cam);                       /* camera.h:110-123 */
void orc_imgui_camera_set_fd(OrcImguiCamera* cam, float new_fd);              /* camera.h:36 */
void orc_imgui_compute_ray(const OrcImguiCamera* cam, uint32_t ix, uint32_t iy, uint32_t dimx, uint32_t dimy, uint32_t* seed,
                           float origin[3], float direction[3]);                /* camera.h:127-144 */
void orc_imgui_light_wi(const OrcImguiLight* l, const float p[3], uint32_t* seed, float wi[3], float lumi[3]);   /* light.h:19-40 */
int  orc_imgui_render_rows(const OrcScene*, int mode, const OrcImguiFrame* f, uint32_t y0, uint32_t y1, OrcStats* stats);

/* ---- SDK/cuda/whitted.cu programs (__raygen__pinhole :44-97, __miss__constant_radiance :139-142,
 * __closesthit__radiance :149-289), opaque + untextured subset, on world-space triangles ---- */
typedef struct OrcLightW {            /* struct Light, cuda/Light.h:33-71 (36 bytes) */
    int32_t type;                     /* 0 POINT, 1 AMBIENT */
    float   color[3], intensity, position[3];
    int32_t falloff;
} OrcLightW;

typedef struct OrcTexInfoW {           /* MaterialData::Texture (cuda/MaterialData.h:60-72); texture < 0: none */
    int32_t texture, texcoord;
    float   offset[2], rotation[2] /* sin, cos */, scale[2];
} OrcTexInfoW;

typedef struct OrcMaterialW {         /* MaterialData, PBR branch (cuda/MaterialData.h) */
    float   base_color[4], metallic, roughness, emissive_factor[3];
    float   alpha_cutoff;
    int32_t alpha_mode, double_sided;
    OrcTexInfoW base_color_tex, metallic_roughness_tex, normal_tex, emissive_tex;
} OrcMaterialW;

typedef struct OrcTextureW {          /* cudaTextureObject state of Scene::addSampler (Scene.cpp:624-652) + texels */
    uint32_t width, height, bits;     /* RGBA, 8 or 16 bits per component, row 0 = top; width 0: undecodable image */
    uint32_t address_s, address_t;    /* 0 wrap, 1 clamp, 2 mirror */
    uint32_t filter;                  /* 0 point, 1 linear */
    int32_t  image;
    const void* texels;
} OrcTextureW;

typedef struct OrcWhittedFrame {      /* whitted::LaunchParams (cuda/whitted.h:59-77) + the scene arrays */
    uint32_t width, height, subframe_index;
    float*   accum_buffer;            /* float4 */
    uint8_t* frame_buffer;            /* uchar4 */
    float    eye[3], U[3], V[3], W[3];
    const OrcLightW* lights; uint32_t nlights;
    float    miss_color[3];
    const float*        normals4;     /* 3 float4 per triangle {world normal, has_normal} or NULL */
    const uint32_t*     material_index;
    const OrcMaterialW* materials; uint32_t nmaterials;
    const float*        texcoords[2];        /* 3 float2 per triangle or NULL */
    const uint8_t*      texcoord_present[2]; /* per triangle or NULL (= present wherever texcoords is given) */
    const float*        colors4;             /* 3 float4 per triangle or NULL */
    const uint8_t*      color_present;
    const OrcTextureW*  textures; uint32_t ntextures;
} OrcWhittedFrame;

/* tex2D<float4> of a texture object created by Scene::addSampler: normalised coordinates, normalised-float read,
 * point or bilinear filter (fp32 weights; the hardware's 8-bit weights are not modelled) */
void orc_sample_texture(const OrcTextureW* t, float u, float v, float out[4]);

int orc_whitted_render_rows(const OrcScene*, int mode, const OrcWhittedFrame* f, uint32_t y0, uint32_t y1, OrcStats* stats);

/* primary rays exactly as raygen emits them for sample 0 (for traversal tests) */
void orc_primary_rays(const OrcParams* p, uint32_t y0, uint32_t y1, float* rays8);

#ifdef __cplusplus
}
#endif
#endif
