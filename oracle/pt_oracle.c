/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See pt_oracle.h for scope and parity status.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -fno-fast-math -shared -fPIC
 * (fmaf must be a single fused op; no implicit contraction anywhere.)
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).  Nothing here is copied: the reference is C++/CUDA on
 * float3 operator overloads, this is scalar C.
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small float helpers; operation ORDER follows SDK/sutil/vec_math.h          */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_ld(const float* p) { return v3_make(p[0], p[1], p[2]); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }   /* vec_math.h:411 */
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }   /* vec_math.h:432 */
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }   /* vec_math.h:453 */
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }    /* vec_math.h:457 */
/* vec_math.h:527  a.x*b.x + a.y*b.y + a.z*b.z, left to right, no fma */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* vec_math.h:533 */
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec_math.h:539 */
static inline float v3_length(v3 v) { return sqrtf(v3_dot(v, v)); }
/* vec_math.h:545  v * (1/sqrt(dot)) */
static inline v3 v3_normalize(v3 v)
{
    float inv = 1.0f / sqrtf(v3_dot(v, v));
    return v3_scale(v, inv);
}
/* NaN-ignoring min/max, same result as C fminf/fmaxf (vec_math.h:119 clamp) */
static inline float f_min(float a, float b) { return (a < b || b != b) ? a : b; }
static inline float f_max(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float f_clamp(float f, float a, float b) { return f_max(a, f_min(f, b)); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* KAT hook for the helpers above: out = dot, length, normalize[3], cross[3],
 * faceforward(normalize(a), -b, normalize(a))[3], lerp(a,b,t)[3], a/t[3] */
void orc_vec_kat(const float a[3], const float b[3], float t, float out[17])
{
    v3 A = v3_ld(a), B = v3_ld(b);
    out[0] = v3_dot(A, B);
    out[1] = v3_length(A);
    v3 n = v3_normalize(A);
    out[2] = n.x; out[3] = n.y; out[4] = n.z;
    v3 c = v3_cross(A, B);
    out[5] = c.x; out[6] = c.y; out[7] = c.z;
    v3 nb = v3_make(-B.x, -B.y, -B.z);
    v3 ff = v3_scale(n, copysignf(1.0f, v3_dot(nb, n)));            /* vec_math.h:567 */
    out[8] = ff.x; out[9] = ff.y; out[10] = ff.z;
    v3 l = v3_add(A, v3_scale(v3_sub(B, A), t));                      /* vec_math.h:500 */
    out[11] = l.x; out[12] = l.y; out[13] = l.z;
    float inv = 1.0f / t;                                             /* vec_math.h:483 */
    v3 d = v3_scale(A, inv);
    out[14] = d.x; out[15] = d.y; out[16] = d.z;
}

/* ------------------------------------------------------------------------- */
/* RNG -- SDK/cuda/random.h:31-67                                             */
/* ------------------------------------------------------------------------- */
uint32_t orc_tea4(uint32_t val0, uint32_t val1)
{
    uint32_t a = val0, b = val1, sum = 0;
    for (int round = 0; round < 4; ++round) {
        sum += 0x9e3779b9u;
        a += ((b << 4) + 0xa341316cu) ^ (b + sum) ^ ((b >> 5) + 0xc8013ea4u);
        b += ((a << 4) + 0xad90777du) ^ (a + sum) ^ ((a >> 5) + 0x7e95761eu);
    }
    return a;
}

uint32_t orc_lcg(uint32_t* prev)
{
    *prev = 1664525u * (*prev) + 1013904223u;
    return *prev & 0x00FFFFFFu;
}

float orc_rnd(uint32_t* prev)
{
    return (float)orc_lcg(prev) / (float)0x01000000;
}

/* ------------------------------------------------------------------------- */
/* Deterministic transcendental math (build-defined; DESIGN.md "detmath").    */
/* The reference calls cosf/sinf/powf under nvcc --use_fast_math              */
/* (SDK/CMakeLists.txt:267); those are not reproducible across devices, so    */
/* CPU and GPU both evaluate these fixed polynomials with explicit fmaf.      */
/* ------------------------------------------------------------------------- */
void orc_det_sincos(float x, float* s_out, float* c_out)
{
    /* quadrant reduction, Cody-Waite 3-term pi/2 */
    float kf = rintf(x * 0.636619772f);
    int   k  = (int)kf;
    float r  = fmaf(kf, -1.5703125f, x);
    r        = fmaf(kf, -4.837512969970703125e-4f, r);
    r        = fmaf(kf, -7.54978995489188216e-8f, r);
    float z  = r * r;
    /* sin(r), |r| <= pi/4 */
    float sp = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp       = fmaf(sp, z, -1.6666654611e-1f);
    sp       = fmaf(sp, z * r, r);
    /* cos(r) */
    float cp = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp       = fmaf(cp, z, 4.166664568298827e-2f);
    cp       = fmaf(cp, z * z, fmaf(-0.5f, z, 1.0f));
    switch (k & 3) {
    case 0:  *s_out = sp;  *c_out = cp;  break;
    case 1:  *s_out = cp;  *c_out = -sp; break;
    case 2:  *s_out = -sp; *c_out = -cp; break;
    default: *s_out = -cp; *c_out = sp;  break;
    }
}

/* x^k (k = 1/2.4, 2.2) for normal positive x <= ~2; log2 via fdlibm-style series, exp2 via
 * degree-7 polynomial.  ~2 ulp. */
static float det_powk(float x, float k)
{
    uint32_t ix = f2u(x);
    int      e  = (int)(ix >> 23) - 127;
    float    m  = u2f((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m *= 0.5f; e += 1; }
    float f    = m - 1.0f;
    float s    = f / (2.0f + f);
    float z    = s * s;
    float w    = z * z;
    float t1   = w * fmaf(w, 0.24279078841f, 0.40000972152f);
    float t2   = z * fmaf(w, 0.28498786688f, 0.66666662693f);
    float R    = t2 + t1;
    float hfsq = 0.5f * f * f;
    float lnm  = f - (hfsq - s * (hfsq + R));
    float lg2  = fmaf(lnm, 1.44269502f, (float)e);
    float y    = lg2 * k;
    float n    = rintf(y);
    float g    = y - n;
    float p    = fmaf(1.52527338e-5f, g, 1.54035304e-4f);
    p          = fmaf(p, g, 1.33335581e-3f);
    p          = fmaf(p, g, 9.61812911e-3f);
    p          = fmaf(p, g, 5.55041087e-2f);
    p          = fmaf(p, g, 2.40226507e-1f);
    p          = fmaf(p, g, 6.93147182e-1f);
    p          = fmaf(p, g, 1.0f);
    return u2f(f2u(p) + ((uint32_t)(int)n << 23));
}

float orc_det_pow_inv24(float x) { return det_powk(x, 0.416666657f); }
/* powf(c, 2.2f) of whitted::linearize (cuda/whitted_cuda.h:73-80), texture values in [0, 1] */
float orc_det_pow22(float x) { return x < 1e-10f ? 0.0f : det_powk(x, 2.2f); }

/* ------------------------------------------------------------------------- */
/* Camera -- SDK/sutil/Camera.cpp:34-46 (fod = author's m_fod, Camera.h:64,75) */
/* ------------------------------------------------------------------------- */
void orc_camera_uvw(const float eye[3], const float lookat[3], const float up[3],
                    float fovY_deg, float aspect, float fod, float U[3], float V[3], float W[3])
{
    v3    w    = v3_scale(v3_sub(v3_ld(lookat), v3_ld(eye)), fod);
    float wlen = v3_length(w);
    v3    u    = v3_normalize(v3_cross(w, v3_ld(up)));
    v3    v    = v3_normalize(v3_cross(u, w));
    float vlen = wlen * tanf(0.5f * fovY_deg * 3.14159265358979323846f / 180.0f);
    v          = v3_scale(v, vlen);
    float ulen = vlen * aspect;
    u          = v3_scale(u, ulen);
    U[0] = u.x; U[1] = u.y; U[2] = u.z;
    V[0] = v.x; V[1] = v.y; V[2] = v.z;
    W[0] = w.x; W[1] = w.y; W[2] = w.z;
}

/* ------------------------------------------------------------------------- */
/* sRGB + 8-bit quantisation -- SDK/cuda/helpers.h:35-66                      */
/* ------------------------------------------------------------------------- */
static inline float srgb_channel(float c)
{
    return c < 0.0031308f ? 12.92f * c : 1.055f * orc_det_pow_inv24(c) - 0.055f;
}

static inline uint8_t quantize8(float x)
{
    x = f_clamp(x, 0.0f, 1.0f);
    uint32_t q = (uint32_t)(x * 256.0f);
    return (uint8_t)(q < 255u ? q : 255u);
}

void orc_make_color(const float c[3], uint8_t out[4])
{
    for (int k = 0; k < 3; ++k)
        out[k] = quantize8(srgb_channel(f_clamp(c[k], 0.0f, 1.0f)));
    out[3] = 255u;
}

/* ------------------------------------------------------------------------- */
/* Hemisphere sampling + ONB -- optixPathTracer.cu:47-78, 149-159             */
/* ------------------------------------------------------------------------- */
void orc_cosine_sample_hemisphere(float u1, float u2, float p[3])
{
    float r   = sqrtf(u1);
    float phi = (2.0f * 3.14159265358979323846f) * u2;
    float sn, cs;
    orc_det_sincos(phi, &sn, &cs);
    p[0] = r * cs;
    p[1] = r * sn;
    p[2] = sqrtf(f_max(0.0f, 1.0f - p[0] * p[0] - p[1] * p[1]));
}

void orc_onb_inverse_transform(const float nrm[3], float p[3])
{
    v3 n = v3_ld(nrm), b;
    if (fabsf(n.x) > fabsf(n.z)) b = v3_make(-n.y, n.x, 0.0f);
    else                         b = v3_make(0.0f, -n.z, n.y);
    b    = v3_normalize(b);
    v3 t = v3_cross(b, n);
    v3 r = v3_add(v3_add(v3_scale(t, p[0]), v3_scale(b, p[1])), v3_scale(n, p[2]));
    p[0] = r.x; p[1] = r.y; p[2] = r.z;
}

/* ------------------------------------------------------------------------- */
/* Multi-GPU tile map -- SDK/sutil/WorkDistribution.h:50-81 (8x4 tiles)       */
/* ------------------------------------------------------------------------- */
enum { TILE_W = 8, TILE_H = 4 };

int orc_work_num_samples(int width, int height, int num_gpus)
{
    int strip_w = TILE_W * num_gpus;
    int cols    = (width + strip_w - 1) / strip_w;
    int rows    = (height + TILE_H - 1) / TILE_H;
    return rows * cols * TILE_W * TILE_H;
}

void orc_work_pixel(int width, int height, int num_gpus, int gpu, int sample_idx, int* px, int* py)
{
    (void)height;
    int strip_w = TILE_W * num_gpus;
    int cols    = (width + strip_w - 1) / strip_w;
    int tile    = sample_idx / (TILE_W * TILE_H);
    int in_tile = sample_idx % (TILE_W * TILE_H);
    int srow    = tile / cols;
    int scol    = tile % cols;
    int rot     = (gpu + srow % num_gpus) % num_gpus;
    *px         = scol * strip_w + rot * TILE_W + in_tile % TILE_W;
    *py         = srow * TILE_H + in_tile / TILE_W;
}

/* ------------------------------------------------------------------------- */
/* Scene + BVH                                                                */
/* ------------------------------------------------------------------------- */
/* 64-byte node, two children with their boxes (layout = include/mipt.h)      */
typedef struct {
    float   c0x[2], c0y[2];     /* child0 lo.x hi.x lo.y hi.y */
    float   c1x[2], c1y[2];     /* child1 lo.x hi.x lo.y hi.y */
    float   c0z[2], c1z[2];     /* child0 lo.z hi.z, child1 lo.z hi.z */
    int32_t c0, c1;             /* >=0 inner node index; <0 leaf: ~((first<<4)|count) */
    uint32_t pad0, pad1;
} Node64;

/* 48-byte triangle in leaf order */
typedef struct {
    float    v0[3]; uint32_t prim;  /* original primitive index */
    float    e1[3]; uint32_t pad1;  /* v1 - v0 */
    float    e2[3]; uint32_t pad2;  /* v2 - v0 */
} Tri48;

struct OrcScene {
    uint32_t  n_tris, n_mat;
    float*    verts;       /* n_tris*3 float4 (caller layout, optixPathTracer.cpp:96-99) */
    uint32_t* mat;         /* per-prim material (optixPathTracer.cpp:285-294) */
    float*    emission;    /* n_mat*3 */
    float*    diffuse;     /* n_mat*3 */
    float     bg[4];       /* MissData (optixPathTracer.h:115-118) */
    Tri48*    brute;       /* all triangles, prim order, for brute force */
    Node64*   nodes; uint32_t n_nodes;
    Tri48*    tris;  uint32_t n_leaf_tris;
    int       depth;
    uint32_t* prim_flags;  /* OptixGeometryFlags per ORIGINAL primitive (NULL = none): 1 DISABLE_ANYHIT, 4 DISABLE_TRIANGLE_FACE_CULLING */
    /* two-level (instance) acceleration structure adopted from the product (orc_scene_set_instances, ORC_MODE_INST):
     * a BVH over the instances' world boxes whose leaves list instances; per instance the world->object 3x4 matrix, the
     * object-space BVH of its mesh and the flattened index of the mesh's primitive 0 (Scene.cpp:1134-1212 builds the IAS the
     * driver walks; the walk itself is the driver's) */
    Node64*   tlas; uint32_t n_tlas;
    uint32_t* inst_order; uint32_t n_inst;
    struct OrcInst { float inv[12]; uint32_t prim_base; Node64* nodes; uint32_t n_nodes; Tri48* tris; uint32_t n_tris; }* inst;
};

#define LEAF_REF(first, count) (~(int32_t)(((uint32_t)(first) << 4) | (uint32_t)(count)))
#define STACK_MAX 64
#define SENTINEL 0x7fffffff

static void make_tri(const float* verts, uint32_t prim, Tri48* t)
{
    const float* a = verts + (size_t)prim * 12;
    const float* b = a + 4;
    const float* c = a + 8;
    for (int k = 0; k < 3; ++k) {
        t->v0[k] = a[k];
        t->e1[k] = b[k] - a[k];
        t->e2[k] = c[k] - a[k];
    }
    t->prim = prim; t->pad1 = 0; t->pad2 = 0;
}

OrcScene* orc_scene_create(const float* vertices4, uint32_t n_tris, const uint32_t* mat_index,
                           const float* emission3, const float* diffuse3, uint32_t n_mat,
                           const float bg[4])
{
    OrcScene* s = (OrcScene*)calloc(1, sizeof(OrcScene));
    s->n_tris = n_tris; s->n_mat = n_mat;
    s->verts = (float*)malloc((size_t)n_tris * 12 * sizeof(float));
    memcpy(s->verts, vertices4, (size_t)n_tris * 12 * sizeof(float));
    s->mat = (uint32_t*)calloc(n_tris ? n_tris : 1, sizeof(uint32_t));
    if (mat_index) memcpy(s->mat, mat_index, (size_t)n_tris * sizeof(uint32_t));
    s->emission = (float*)malloc((size_t)n_mat * 3 * sizeof(float));
    s->diffuse  = (float*)malloc((size_t)n_mat * 3 * sizeof(float));
    memcpy(s->emission, emission3, (size_t)n_mat * 3 * sizeof(float));
    memcpy(s->diffuse, diffuse3, (size_t)n_mat * 3 * sizeof(float));
    memcpy(s->bg, bg, 4 * sizeof(float));
    s->brute = (Tri48*)malloc((size_t)(n_tris ? n_tris : 1) * sizeof(Tri48));
    for (uint32_t i = 0; i < n_tris; ++i) make_tri(s->verts, i, &s->brute[i]);
    return s;
}

void orc_scene_destroy(OrcScene* s)
{
    if (!s) return;
    free(s->verts); free(s->mat); free(s->emission); free(s->diffuse);
    free(s->brute); free(s->nodes); free(s->tris); free(s->prim_flags);
    if (s->inst) for (uint32_t i = 0; i < s->n_inst; ++i) { free(s->inst[i].nodes); free(s->inst[i].tris); }
    free(s->inst); free(s->tlas); free(s->inst_order);
    free(s);
}

/* adopts the product's two-level structure (mipt_accel_export_instances + mipt_accel_export of every instance's mesh BVH) */
int orc_scene_set_instances(OrcScene* s, const void* tlas_nodes, uint32_t n_tlas, const uint32_t* order, uint32_t n_inst,
                            const float* inv12, const uint32_t* prim_base, const void* const* gas_nodes, const uint32_t* gas_n_nodes,
                            const void* const* gas_tris, const uint32_t* gas_n_tris)
{
    if (!s || !tlas_nodes || n_tlas == 0 || !order || n_inst == 0 || !inv12 || !prim_base || !gas_nodes || !gas_tris) return 1;
    s->tlas = (Node64*)malloc((size_t)n_tlas * sizeof(Node64));
    memcpy(s->tlas, tlas_nodes, (size_t)n_tlas * sizeof(Node64));
    s->n_tlas = n_tlas;
    s->inst_order = (uint32_t*)malloc((size_t)n_inst * 4);
    memcpy(s->inst_order, order, (size_t)n_inst * 4);
    s->n_inst = n_inst;
    s->inst = (struct OrcInst*)calloc(n_inst, sizeof *s->inst);
    for (uint32_t i = 0; i < n_inst; ++i) {
        struct OrcInst* in = &s->inst[i];
        memcpy(in->inv, inv12 + (size_t)i * 12, 48);
        in->prim_base = prim_base[i];
        in->n_nodes = gas_n_nodes[i]; in->n_tris = gas_n_tris[i];
        in->nodes = (Node64*)malloc((size_t)in->n_nodes * sizeof(Node64));
        in->tris = (Tri48*)malloc((size_t)(in->n_tris ? in->n_tris : 1) * sizeof(Tri48));
        memcpy(in->nodes, gas_nodes[i], (size_t)in->n_nodes * sizeof(Node64));
        memcpy(in->tris, gas_tris[i], (size_t)in->n_tris * sizeof(Tri48));
        /* flattened ids must index the scene's arrays (a mesh built with spatial splits holds more records than primitives: the
         * largest primitive number counts, not the number of records) */
        for (uint32_t k = 0; k < in->n_tris; ++k) if ((uint64_t)in->prim_base + in->tris[k].prim >= s->n_tris) return 2;
    }
    for (uint32_t i = 0; i < n_inst; ++i) if (order[i] >= n_inst) return 3;
    return 0;
}

/* OptixGeometryFlags per original primitive (copied); NULL clears them */
void orc_scene_set_prim_flags(OrcScene* s, const uint32_t* flags)
{
    free(s->prim_flags); s->prim_flags = NULL;
    if (!flags) return;
    s->prim_flags = (uint32_t*)malloc((size_t)(s->n_tris ? s->n_tris : 1) * sizeof(uint32_t));
    memcpy(s->prim_flags, flags, (size_t)s->n_tris * sizeof(uint32_t));
}

uint32_t orc_scene_num_nodes(const OrcScene* s) { return s->n_nodes; }
int      orc_scene_bvh_depth(const OrcScene* s) { return s->depth; }

/* ---- binned SAH builder (own definition; OptiX's builder is closed) ---- */
typedef struct { float lo[3], hi[3]; } Box;

static inline void box_empty(Box* b)
{
    for (int k = 0; k < 3; ++k) { b->lo[k] = INFINITY; b->hi[k] = -INFINITY; }
}
static inline void box_grow_pt(Box* b, const float* p)
{
    for (int k = 0; k < 3; ++k) { if (p[k] < b->lo[k]) b->lo[k] = p[k]; if (p[k] > b->hi[k]) b->hi[k] = p[k]; }
}
static inline void box_grow(Box* b, const Box* o)
{
    for (int k = 0; k < 3; ++k) { if (o->lo[k] < b->lo[k]) b->lo[k] = o->lo[k]; if (o->hi[k] > b->hi[k]) b->hi[k] = o->hi[k]; }
}
static inline float box_half_area(const Box* b)
{
    float dx = b->hi[0] - b->lo[0], dy = b->hi[1] - b->lo[1], dz = b->hi[2] - b->lo[2];
    if (!(dx >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}

typedef struct {
    const OrcScene* s;
    Box*      tbox;     /* per prim */
    float*    cent;     /* per prim * 3 */
    uint32_t* idx;      /* permutation */
    Node64*   nodes; uint32_t n_nodes, cap_nodes;
    Tri48*    tris;  uint32_t n_out;
    int       leaf_max, max_depth, depth_seen;
} Builder;

static void store_child(Node64* n, int which, const Box* b, int32_t ref)
{
    if (which == 0) {
        n->c0x[0] = b->lo[0]; n->c0x[1] = b->hi[0]; n->c0y[0] = b->lo[1]; n->c0y[1] = b->hi[1];
        n->c0z[0] = b->lo[2]; n->c0z[1] = b->hi[2]; n->c0 = ref;
    } else {
        n->c1x[0] = b->lo[0]; n->c1x[1] = b->hi[0]; n->c1y[0] = b->lo[1]; n->c1y[1] = b->hi[1];
        n->c1z[0] = b->lo[2]; n->c1z[1] = b->hi[2]; n->c1 = ref;
    }
}

static Box range_box(const Builder* B, uint32_t lo, uint32_t hi)
{
    Box b; box_empty(&b);
    for (uint32_t i = lo; i < hi; ++i) box_grow(&b, &B->tbox[B->idx[i]]);
    return b;
}

static int32_t emit_leaf(Builder* B, uint32_t lo, uint32_t hi)
{
    uint32_t first = B->n_out;
    for (uint32_t i = lo; i < hi; ++i) make_tri(B->s->verts, B->idx[i], &B->tris[B->n_out++]);
    return LEAF_REF(first, hi - lo);
}

static int cmp_axis; static const float* cmp_cent;
static int cmp_idx(const void* a, const void* b)
{
    uint32_t ia = *(const uint32_t*)a, ib = *(const uint32_t*)b;
    float ca = cmp_cent[ia * 3 + cmp_axis], cb = cmp_cent[ib * 3 + cmp_axis];
    if (ca < cb) return -1;
    if (ca > cb) return 1;
    return ia < ib ? -1 : (ia > ib ? 1 : 0);
}

/* returns split position mid in (lo,hi) after partitioning idx[lo,hi) */
static uint32_t split_range(Builder* B, uint32_t lo, uint32_t hi, int rem_depth)
{
    enum { NB = 16 };
    uint32_t n = hi - lo;
    Box cb; box_empty(&cb);
    for (uint32_t i = lo; i < hi; ++i) box_grow_pt(&cb, &B->cent[B->idx[i] * 3]);
    uint64_t cap = (uint64_t)B->leaf_max << (rem_depth - 1 > 40 ? 40 : rem_depth - 1);
    float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
    for (int ax = 0; ax < 3; ++ax) {
        float ext = cb.hi[ax] - cb.lo[ax];
        if (!(ext > 0.0f)) continue;
        Box bb[NB]; uint32_t cnt[NB];
        for (int b = 0; b < NB; ++b) { box_empty(&bb[b]); cnt[b] = 0; }
        float scale = (float)NB / ext;
        for (uint32_t i = lo; i < hi; ++i) {
            uint32_t p = B->idx[i];
            int b = (int)((B->cent[p * 3 + ax] - cb.lo[ax]) * scale);
            if (b < 0) b = 0; if (b > NB - 1) b = NB - 1;
            cnt[b]++; box_grow(&bb[b], &B->tbox[p]);
        }
        float rarea[NB]; uint32_t rcnt[NB];
        Box acc; box_empty(&acc); uint32_t c = 0;
        for (int b = NB - 1; b > 0; --b) { box_grow(&acc, &bb[b]); c += cnt[b]; rarea[b] = box_half_area(&acc); rcnt[b] = c; }
        box_empty(&acc); c = 0;
        for (int b = 0; b < NB - 1; ++b) {
            box_grow(&acc, &bb[b]); c += cnt[b];
            uint32_t nl = c, nr = rcnt[b + 1];
            if (nl == 0 || nr == 0) continue;
            if (nl > cap || nr > cap) continue;
            float cost = box_half_area(&acc) * (float)nl + rarea[b + 1] * (float)nr;
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
        }
    }
    if (best_axis >= 0) {
        float ext = cb.hi[best_axis] - cb.lo[best_axis];
        float scale = (float)NB / ext;
        uint32_t i = lo, j = hi;
        while (i < j) {
            uint32_t p = B->idx[i];
            int b = (int)((B->cent[p * 3 + best_axis] - cb.lo[best_axis]) * scale);
            if (b < 0) b = 0; if (b > NB - 1) b = NB - 1;
            if (b <= best_bin) ++i;
            else { --j; B->idx[i] = B->idx[j]; B->idx[j] = p; }
        }
        if (i > lo && i < hi) return i;
    }
    /* fallback: object median on the widest centroid axis (also enforces depth cap) */
    int ax = 0; float w = -1.0f;
    for (int k = 0; k < 3; ++k) { float e = cb.hi[k] - cb.lo[k]; if (e > w) { w = e; ax = k; } }
    cmp_axis = ax; cmp_cent = B->cent;
    qsort(B->idx + lo, n, sizeof(uint32_t), cmp_idx);
    return lo + n / 2;
}

/* builds the subtree for idx[lo,hi), returns its reference */
static int32_t build_range(Builder* B, uint32_t lo, uint32_t hi, int depth)
{
    if (depth > B->depth_seen) B->depth_seen = depth;
    uint32_t n = hi - lo;
    if (n <= (uint32_t)B->leaf_max) return emit_leaf(B, lo, hi);
    uint32_t me = B->n_nodes++;
    uint32_t mid = split_range(B, lo, hi, B->max_depth - depth);
    Box b0 = range_box(B, lo, mid), b1 = range_box(B, mid, hi);
    int32_t r0 = build_range(B, lo, mid, depth + 1);
    int32_t r1 = build_range(B, mid, hi, depth + 1);
    store_child(&B->nodes[me], 0, &b0, r0);
    store_child(&B->nodes[me], 1, &b1, r1);
    return (int32_t)me;
}

int orc_scene_build_bvh(OrcScene* s, int leaf_max, int max_depth)
{
    if (leaf_max < 1 || leaf_max > 15 || max_depth < 2 || max_depth > STACK_MAX - 2) return -1;
    { uint64_t cap = (uint64_t)leaf_max << (max_depth > 40 ? 40 : max_depth); if (s->n_tris > cap) return -2; }
    free(s->nodes); free(s->tris); s->nodes = NULL; s->tris = NULL;
    Builder B; memset(&B, 0, sizeof B);
    B.s = s; B.leaf_max = leaf_max; B.max_depth = max_depth;
    uint32_t n = s->n_tris;
    B.tbox = (Box*)malloc((size_t)(n ? n : 1) * sizeof(Box));
    B.cent = (float*)malloc((size_t)(n ? n : 1) * 3 * sizeof(float));
    B.idx  = (uint32_t*)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) {
        const float* v = s->verts + (size_t)i * 12;
        box_empty(&B.tbox[i]);
        box_grow_pt(&B.tbox[i], v); box_grow_pt(&B.tbox[i], v + 4); box_grow_pt(&B.tbox[i], v + 8);
        for (int k = 0; k < 3; ++k) B.cent[i * 3 + k] = 0.5f * (B.tbox[i].lo[k] + B.tbox[i].hi[k]);
        B.idx[i] = i;
    }
    B.cap_nodes = n + 2;
    B.nodes = (Node64*)calloc(B.cap_nodes, sizeof(Node64));
    B.tris  = (Tri48*)malloc((size_t)(n ? n : 1) * sizeof(Tri48));
    if (n <= (uint32_t)leaf_max) {
        /* root must be an inner node: child0 = everything, child1 = empty leaf */
        Box b0 = range_box(&B, 0, n), bz; memset(&bz, 0, sizeof bz);
        B.n_nodes = 1;
        int32_t r0 = emit_leaf(&B, 0, n);
        store_child(&B.nodes[0], 0, &b0, r0);
        store_child(&B.nodes[0], 1, &bz, LEAF_REF(n, 0));
        B.depth_seen = 1;
    } else {
        build_range(&B, 0, n, 0);
    }
    s->nodes = B.nodes; s->n_nodes = B.n_nodes;
    s->tris = B.tris;   s->n_leaf_tris = B.n_out;
    s->depth = B.depth_seen;
    free(B.tbox); free(B.cent); free(B.idx);
    return 0;
}

int orc_scene_set_bvh(OrcScene* s, const void* nodes, uint32_t n_nodes, const void* tris, uint32_t n_tris)
{
    free(s->nodes); free(s->tris);
    s->nodes = (Node64*)malloc((size_t)n_nodes * sizeof(Node64));
    s->tris  = (Tri48*)malloc((size_t)(n_tris ? n_tris : 1) * sizeof(Tri48));
    memcpy(s->nodes, nodes, (size_t)n_nodes * sizeof(Node64));
    memcpy(s->tris, tris, (size_t)n_tris * sizeof(Tri48));
    s->n_nodes = n_nodes; s->n_leaf_tris = n_tris;
    s->depth = -1;
    return orc_scene_validate_bvh(s);
}

/* every prim appears exactly once, child boxes contain their triangles, refs in range,
 * depth < STACK_MAX.  Returns 0 when sound.
 * A tree with MORE triangle records than the scene has triangles was built with spatial splits (MIPT_OPT_BUILD = 3): a triangle may then
 * be referenced by several leaves, each bounding only the part of it that lies in the leaf's region -- there every primitive must
 * appear at least once, every reference's leaf box must overlap its triangle's bounds, and the leaf boxes of a triangle's references
 * must COVER the triangle where it is sampled: its three corners (exactly), its edge midpoints and its centroid (computed in double,
 * compared with a relative slack of 1e-12) must each lie in the leaf box of one of its references -- what can be said without redoing
 * the builder's clipping; that such a tree loses no hit is what the tests check against the brute-force intersector. */
static int validate_rec(const OrcScene* s, int32_t ref, const Box* bound, uint8_t* seen, uint8_t* cover, int depth, int* maxd)
{
    const int spatial = s->n_leaf_tris > s->n_tris;
    if (depth > *maxd) *maxd = depth;
    if (depth >= STACK_MAX - 1) return -10;
    if (ref < 0) {
        uint32_t d = (uint32_t)~ref, first = d >> 4, cnt = d & 15u;
        if (first + cnt > s->n_leaf_tris) return -11;
        for (uint32_t i = first; i < first + cnt; ++i) {
            const Tri48* t = &s->tris[i];
            if (t->prim >= s->n_tris || (seen[t->prim] && !spatial)) return -12;
            seen[t->prim] = 1;
            const float* v = s->verts + (size_t)t->prim * 12;
            if (!spatial) {
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < 3; ++k) {
                        float x = v[c * 4 + k];
                        if (x < bound->lo[k] || x > bound->hi[k]) return -13;
                    }
            } else {
                for (int k = 0; k < 3; ++k) {
                    const float tlo = fminf(v[k], fminf(v[4 + k], v[8 + k])), thi = fmaxf(v[k], fmaxf(v[4 + k], v[8 + k]));
                    if (thi < bound->lo[k] || tlo > bound->hi[k]) return -13;
                }
                if (cover) {
                    /* sample points: corners 0..2, edge midpoints 3..5, centroid 6 */
                    double pt[7][3];
                    for (int k = 0; k < 3; ++k) {
                        const double a = v[k], b = v[4 + k], c = v[8 + k];
                        pt[0][k] = a; pt[1][k] = b; pt[2][k] = c;
                        pt[3][k] = 0.5 * (a + b); pt[4][k] = 0.5 * (b + c); pt[5][k] = 0.5 * (c + a); pt[6][k] = (a + b + c) / 3.0;
                    }
                    for (int q = 0; q < 7; ++q) {
                        int in = 1;
                        for (int k = 0; k < 3; ++k) {
                            const double slack = q < 3 ? 0.0 : 1e-12 * (fabs(pt[q][k]) + 1e-30);
                            if (pt[q][k] < (double)bound->lo[k] - slack || pt[q][k] > (double)bound->hi[k] + slack) in = 0;
                        }
                        if (in) cover[t->prim] |= (uint8_t)(1u << q);
                    }
                }
            }
            for (int k = 0; k < 3; ++k) {
                if (t->v0[k] != v[k] || t->e1[k] != v[4 + k] - v[k] || t->e2[k] != v[8 + k] - v[k]) return -14;
            }
        }
        return 0;
    }
    if ((uint32_t)ref >= s->n_nodes) return -15;
    const Node64* n = &s->nodes[ref];
    Box b0 = { { n->c0x[0], n->c0y[0], n->c0z[0] }, { n->c0x[1], n->c0y[1], n->c0z[1] } };
    Box b1 = { { n->c1x[0], n->c1y[0], n->c1z[0] }, { n->c1x[1], n->c1y[1], n->c1z[1] } };
    int r = validate_rec(s, n->c0, &b0, seen, cover, depth + 1, maxd);
    if (r) return r;
    return validate_rec(s, n->c1, &b1, seen, cover, depth + 1, maxd);
}

int orc_scene_validate_bvh(const OrcScene* cs)
{
    OrcScene* s = (OrcScene*)cs;
    if (!s->nodes || s->n_nodes == 0) return -1;
    uint8_t* seen = (uint8_t*)calloc(s->n_tris ? s->n_tris : 1, 1);
    uint8_t* cover = s->n_leaf_tris > s->n_tris ? (uint8_t*)calloc(s->n_tris ? s->n_tris : 1, 1) : NULL;      /* spatial splits: sample coverage */
    Box all = { { -INFINITY, -INFINITY, -INFINITY }, { INFINITY, INFINITY, INFINITY } };
    int maxd = 0;
    int r = validate_rec(s, 0, &all, seen, cover, 0, &maxd);
    if (!r) for (uint32_t i = 0; i < s->n_tris; ++i) if (!seen[i]) { r = -16; break; }
    if (!r && cover) for (uint32_t i = 0; i < s->n_tris; ++i) if (cover[i] != 0x7fu) { r = -17; break; }
    free(seen); free(cover);
    if (!r) s->depth = maxd;
    return r;
}

/* ------------------------------------------------------------------------- */
/* Ray / triangle and ray / box -- BUILD-DEFINED conventions:   */
/*   hit iff tmin < t < tmax; double sided; det == 0 rejected; u>=0, v>=0,    */
/*   u+v<=1 inclusive; closest-hit tie -> lowest primitive index.             */
/* Moeller-Trumbore on (v0, e1, e2) with explicit fmaf (same sequence in the  */
/* HIP kernel).  Replaces optixTraverse (optixPathTracer.cu:184-197,227-238). */
/* ------------------------------------------------------------------------- */
typedef struct { v3 o, d; float tmin, tmax; v3 idir, ood; } Ray;

static inline float fdot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 fcross(v3 a, v3 b)
{
    return v3_make(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

static inline void ray_prepare(Ray* r)
{
    const float eps = 8.27180613e-25f; /* 2^-80 */
    float dx = fabsf(r->d.x) > eps ? r->d.x : copysignf(eps, r->d.x);
    float dy = fabsf(r->d.y) > eps ? r->d.y : copysignf(eps, r->d.y);
    float dz = fabsf(r->d.z) > eps ? r->d.z : copysignf(eps, r->d.z);
    r->idir = v3_make(1.0f / dx, 1.0f / dy, 1.0f / dz);
    r->ood  = v3_make(r->o.x * r->idir.x, r->o.y * r->idir.y, r->o.z * r->idir.z);
}

/* returns 1 and writes t,u,v when the triangle is hit inside (tmin, tmax) */
static inline int tri_test_det(const Tri48* T, const Ray* r, float tmax, float* t_out, float* u_out, float* v_out, float* det_out)
{
    v3    e1 = v3_ld(T->e1), e2 = v3_ld(T->e2);
    v3    p   = fcross(r->d, e2);
    float det = fdot(e1, p);
    if (det == 0.0f) return 0;
    float inv = 1.0f / det;
    v3    tv  = v3_sub(r->o, v3_ld(T->v0));
    float u   = fdot(tv, p) * inv;
    v3    q   = fcross(tv, e1);
    float v   = fdot(r->d, q) * inv;
    float t   = fdot(e2, q) * inv;
    if (!(u >= 0.0f && v >= 0.0f && u + v <= 1.0f)) return 0;
    if (!(t > r->tmin && t < tmax)) return 0;
    *t_out = t; *u_out = u; *v_out = v; *det_out = det;
    return 1;
}
static inline int tri_test(const Tri48* T, const Ray* r, float tmax, float* t_out, float* u_out, float* v_out)
{
    float det;
    return tri_test_det(T, r, tmax, t_out, u_out, v_out, &det);
}

/* slab test of one child box; returns entry distance, hit flag through *hit */
static inline float box_test(const float x[2], const float y[2], const float z[2], const Ray* r, float tfar_limit, int* hit)
{
    float x0 = fmaf(x[0], r->idir.x, -r->ood.x), x1 = fmaf(x[1], r->idir.x, -r->ood.x);
    float y0 = fmaf(y[0], r->idir.y, -r->ood.y), y1 = fmaf(y[1], r->idir.y, -r->ood.y);
    float z0 = fmaf(z[0], r->idir.z, -r->ood.z), z1 = fmaf(z[1], r->idir.z, -r->ood.z);
    float tn = f_max(f_max(f_min(x0, x1), f_min(y0, y1)), f_max(f_min(z0, z1), r->tmin));
    float tf = f_min(f_min(f_max(x0, x1), f_max(y0, y1)), f_min(f_max(z0, z1), tfar_limit));
    *hit = tn <= tf;
    return tn;
}

typedef struct { int hit; float t, u, v; uint32_t prim; } HitRec;

/* what optixTrace's ray flags and the any-hit programs add to a plain traversal (NULL = nothing) */
typedef struct TraceFilter {
    uint32_t ray_flags;                                   /* 16 = OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES */
    int    (*anyhit)(void* ctx, uint32_t prim, float t, float u, float v);   /* 0 = optixIgnoreIntersection() */
    void*    ctx;
} TraceFilter;

/* back = the ray sees (v1-v0) x (v2-v0) from behind: det = e1 . (d x e2) = -d . Ng < 0 */
static inline int filter_rejects(const OrcScene* s, const TraceFilter* f, uint32_t prim, float det, float t, float u, float v)
{
    if (!f) return 0;
    uint32_t gf = s->prim_flags ? s->prim_flags[prim] : 0u;
    if ((f->ray_flags & 16u) && !(gf & 4u) && det < 0.0f) return 1;
    if (f->anyhit && !(gf & 1u) && !f->anyhit(f->ctx, prim, t, u, v)) return 1;
    return 0;
}

static inline void consider(HitRec* h, float t, float u, float v, uint32_t prim)
{
    if (!h->hit || t < h->t || (t == h->t && prim < h->prim)) {
        h->hit = 1; h->t = t; h->u = u; h->v = v; h->prim = prim;
    }
}

static void brute_force(const OrcScene* s, Ray* r, int any, HitRec* h, OrcStats* st, const TraceFilter* flt)
{
    h->hit = 0; h->t = r->tmax; h->prim = 0xffffffffu; h->u = h->v = 0.0f;
    for (uint32_t i = 0; i < s->n_tris; ++i) {
        float t, u, v, det;
        if (st) { st->tri_tests++; if (any) st->shadow_tri_tests++; }
        if (tri_test_det(&s->brute[i], r, r->tmax, &t, &u, &v, &det)) {
            if (filter_rejects(s, flt, i, det, t, u, v)) continue;
            if (any) { h->hit = 1; h->t = t; h->prim = i; return; }
            consider(h, t, u, v, i);
        }
    }
}

/* per-ray stack traversal, near child first; identical visiting order to the HIP kernel */
static void bvh_traverse(const OrcScene* s, Ray* r, int any, HitRec* h, OrcStats* st, const TraceFilter* flt)
{
    int32_t stack[STACK_MAX];
    int     sp = 0;
    h->hit = 0; h->t = r->tmax; h->prim = 0xffffffffu; h->u = h->v = 0.0f;
    stack[sp++] = SENTINEL;
    if (!(r->tmax > r->tmin)) return;          /* empty interval: a miss without touching the BVH */
    int32_t node = 0;
    while (node != SENTINEL) {
        if (node >= 0) {
            const Node64* n = &s->nodes[node];
            if (st) { st->node_visits++; if (any) st->shadow_node_visits++; }
            int   h0, h1;
            /* prune with the current best t (inclusive, so equal-t candidates are still visited) */
            float lim = h->t;
            float t0 = box_test(n->c0x, n->c0y, n->c0z, r, lim, &h0);
            float t1 = box_test(n->c1x, n->c1y, n->c1z, r, lim, &h1);
            if (!h0 && !h1) { node = stack[--sp]; continue; }
            int32_t a = n->c0, b = n->c1;
            if (h0 && h1) {
                if (t1 < t0) { int32_t tmp = a; a = b; b = tmp; }
                stack[sp++] = b;
                node = a;
            } else {
                node = h0 ? a : b;
            }
        } else {
            uint32_t d = (uint32_t)~node, first = d >> 4, cnt = d & 15u;
            for (uint32_t i = first; i < first + cnt; ++i) {
                float t, u, v, det;
                if (st) { st->tri_tests++; if (any) st->shadow_tri_tests++; }
                if (tri_test_det(&s->tris[i], r, r->tmax, &t, &u, &v, &det)) {
                    if (filter_rejects(s, flt, s->tris[i].prim, det, t, u, v)) continue;
                    if (any) { h->hit = 1; h->t = t; h->prim = s->tris[i].prim; return; }
                    consider(h, t, u, v, s->tris[i].prim);
                }
            }
            node = stack[--sp];
        }
    }
}

/* ---- two-level traversal (ORC_MODE_INST) -----------------------------------------------------------------------
 * The world ray walks the BVH over the instance boxes; at a leaf, for each listed instance in turn, the ray is taken
 * into the instance's object space -- o' = M o + t, d' = M d with M = the exported world->object matrix, each row
 * fmaf(m2, z, fmaf(m1, y, m0 * x)) (+ m3 for points); d' is NOT re-normalised, so a distance t means the same point in
 * both spaces and tmin / tmax / the running best carry over -- and the mesh's own BVH is walked to the end.  A hit
 * reports the flattened primitive index prim_base + local index; ties between equal t go to the lower flattened index,
 * as in the one-level walk.  Same sequence in csrc/instances.hip. */
static inline v3 xform_vec(const float* m, v3 d)
{
    return v3_make(fmaf(m[2], d.z, fmaf(m[1], d.y, m[0] * d.x)), fmaf(m[6], d.z, fmaf(m[5], d.y, m[4] * d.x)),
                   fmaf(m[10], d.z, fmaf(m[9], d.y, m[8] * d.x)));
}
static inline v3 xform_pt(const float* m, v3 p)
{
    v3 r = xform_vec(m, p);
    return v3_make(r.x + m[3], r.y + m[7], r.z + m[11]);
}

/* one mesh BVH, object-space ray; h is the WORLD hit record: candidates enter it with their flattened index */
static void gas_traverse(const OrcScene* s, const struct OrcInst* in, Ray* r, int any, HitRec* h, OrcStats* st, const TraceFilter* flt)
{
    int32_t stack[STACK_MAX];
    int     sp = 0;
    stack[sp++] = SENTINEL;
    int32_t node = 0;
    while (node != SENTINEL) {
        if (node >= 0) {
            const Node64* n = &in->nodes[node];
            if (st) { st->node_visits++; if (any) st->shadow_node_visits++; }
            int   h0, h1;
            float lim = h->t;
            float t0 = box_test(n->c0x, n->c0y, n->c0z, r, lim, &h0);
            float t1 = box_test(n->c1x, n->c1y, n->c1z, r, lim, &h1);
            if (!h0 && !h1) { node = stack[--sp]; continue; }
            int32_t a = n->c0, b = n->c1;
            if (h0 && h1) {
                if (t1 < t0) { int32_t tmp = a; a = b; b = tmp; }
                stack[sp++] = b;
                node = a;
            } else {
                node = h0 ? a : b;
            }
        } else {
            uint32_t d = (uint32_t)~node, first = d >> 4, cnt = d & 15u;
            for (uint32_t i = first; i < first + cnt; ++i) {
                float t, u, v, det;
                if (st) { st->tri_tests++; if (any) st->shadow_tri_tests++; }
                if (tri_test_det(&in->tris[i], r, r->tmax, &t, &u, &v, &det)) {
                    if (filter_rejects(s, flt, in->prim_base + in->tris[i].prim, det, t, u, v)) continue;
                    if (any) { h->hit = 1; h->t = t; h->prim = in->prim_base + in->tris[i].prim; return; }
                    consider(h, t, u, v, in->prim_base + in->tris[i].prim);
                }
            }
            node = stack[--sp];
        }
    }
}

static void inst_traverse(const OrcScene* s, Ray* r, int any, HitRec* h, OrcStats* st, const TraceFilter* flt)
{
    int32_t stack[STACK_MAX];
    int     sp = 0;
    h->hit = 0; h->t = r->tmax; h->prim = 0xffffffffu; h->u = h->v = 0.0f;
    stack[sp++] = SENTINEL;
    if (!(r->tmax > r->tmin)) return;
    int32_t node = 0;
    while (node != SENTINEL) {
        if (node >= 0) {
            const Node64* n = &s->tlas[node];
            if (st) { st->node_visits++; if (any) st->shadow_node_visits++; }
            int   h0, h1;
            float lim = h->t;
            float t0 = box_test(n->c0x, n->c0y, n->c0z, r, lim, &h0);
            float t1 = box_test(n->c1x, n->c1y, n->c1z, r, lim, &h1);
            if (!h0 && !h1) { node = stack[--sp]; continue; }
            int32_t a = n->c0, b = n->c1;
            if (h0 && h1) {
                if (t1 < t0) { int32_t tmp = a; a = b; b = tmp; }
                stack[sp++] = b;
                node = a;
            } else {
                node = h0 ? a : b;
            }
        } else {
            uint32_t d = (uint32_t)~node, first = d >> 4, cnt = d & 15u;
            for (uint32_t i = first; i < first + cnt; ++i) {
                const struct OrcInst* in = &s->inst[s->inst_order[i]];
                Ray ro;
                ro.o = xform_pt(in->inv, r->o); ro.d = xform_vec(in->inv, r->d); ro.tmin = r->tmin; ro.tmax = r->tmax;
                ray_prepare(&ro);
                gas_traverse(s, in, &ro, any, h, st, flt);
                if (any && h->hit) return;
            }
            node = stack[--sp];
        }
    }
}

static inline void trace_ex(const OrcScene* s, int mode, v3 o, v3 d, float tmin, float tmax, int any,
                            HitRec* h, OrcStats* st, const TraceFilter* flt)
{
    Ray r; r.o = o; r.d = d; r.tmin = tmin; r.tmax = tmax;
    ray_prepare(&r);
    if (st) { if (any) st->shadow_rays++; else st->closest_rays++; }
    if (mode == ORC_MODE_INST && s->tlas) inst_traverse(s, &r, any, h, st, flt);
    else if (mode == ORC_MODE_BVH)        bvh_traverse(s, &r, any, h, st, flt);
    else                                  brute_force(s, &r, any, h, st, flt);
}
static inline void trace(const OrcScene* s, int mode, v3 o, v3 d, float tmin, float tmax, int any,
                         HitRec* h, OrcStats* st)
{
    trace_ex(s, mode, o, d, tmin, tmax, any, h, st, NULL);
}

void orc_trace_closest(const OrcScene* s, int mode, const float* rays, uint32_t n,
                       float* hit_t, uint32_t* hit_prim, float* hit_u, float* hit_v, OrcStats* st)
{
    for (uint32_t i = 0; i < n; ++i) {
        const float* q = rays + (size_t)i * 8;
        HitRec h;
        trace(s, mode, v3_ld(q), v3_ld(q + 4), q[3], q[7], 0, &h, st);
        hit_t[i]    = h.hit ? h.t : -1.0f;
        hit_prim[i] = h.hit ? h.prim : 0xffffffffu;
        if (hit_u) hit_u[i] = h.u;
        if (hit_v) hit_v[i] = h.v;
    }
}

/* step profile of one closest-hit traversal: pairs (inner steps before the leaf, triangles in the leaf);
 * a final pair (k, 0xffff) holds the inner steps after the last leaf.  For scheduling studies only. */
uint32_t orc_trace_profile(const OrcScene* s, const float* ray8, int any, uint16_t* pairs, uint32_t max_pairs)
{
    Ray r; r.o = v3_ld(ray8); r.d = v3_ld(ray8 + 4); r.tmin = ray8[3]; r.tmax = ray8[7];
    ray_prepare(&r);
    int32_t stack[STACK_MAX]; int sp = 0;
    HitRec h; h.hit = 0; h.t = r.tmax; h.prim = 0xffffffffu; h.u = h.v = 0.0f;
    uint32_t np = 0; uint16_t inner = 0;
    stack[sp++] = SENTINEL;
    int32_t node = (r.tmax > r.tmin) ? 0 : SENTINEL;
    while (node != SENTINEL) {
        if (node >= 0) {
            const Node64* n = &s->nodes[node];
            inner++;
            int h0, h1;
            float t0 = box_test(n->c0x, n->c0y, n->c0z, &r, h.t, &h0);
            float t1 = box_test(n->c1x, n->c1y, n->c1z, &r, h.t, &h1);
            if (!h0 && !h1) { node = stack[--sp]; continue; }
            int32_t a = n->c0, b = n->c1;
            if (h0 && h1) { if (t1 < t0) { int32_t tmp = a; a = b; b = tmp; } stack[sp++] = b; node = a; }
            else node = h0 ? a : b;
        } else {
            uint32_t d = (uint32_t)~node, first = d >> 4, cnt = d & 15u;
            uint32_t tested = 0; int stop = 0;
            for (uint32_t i = first; i < first + cnt; ++i) {
                float t, u, v;
                tested++;
                if (tri_test(&s->tris[i], &r, r.tmax, &t, &u, &v)) {
                    if (any) { stop = 1; break; }
                    consider(&h, t, u, v, s->tris[i].prim);
                }
            }
            if (np < max_pairs) { pairs[2 * np] = inner; pairs[2 * np + 1] = (uint16_t)tested; }
            np++; inner = 0;
            if (stop) return np;
            node = stack[--sp];
        }
    }
    if (np < max_pairs) { pairs[2 * np] = inner; pairs[2 * np + 1] = 0xffff; }
    return np + 1;
}

void orc_trace_any(const OrcScene* s, int mode, const float* rays, uint32_t n, uint8_t* occluded, OrcStats* st)
{
    for (uint32_t i = 0; i < n; ++i) {
        const float* q = rays + (size_t)i * 8;
        HitRec h;
        trace(s, mode, v3_ld(q), v3_ld(q + 4), q[3], q[7], 1, &h, st);
        occluded[i] = (uint8_t)h.hit;
    }
}

/* ------------------------------------------------------------------------- */
/* Path tracer -- optixPathTracer.cu:249-413                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    v3 attenuation; uint32_t seed; int depth;       /* optixPathTracer.h:35-49 RadiancePRD */
    v3 emitted, radiance, origin, direction; int done;
} PRD;

/* __miss__radiance, optixPathTracer.cu:323-335 */
static void miss_radiance(const OrcScene* s, PRD* prd)
{
    prd->radiance = v3_make(s->bg[0], s->bg[1], s->bg[2]);
    prd->emitted  = v3_make(0.0f, 0.0f, 0.0f);
    prd->done     = 1;
}

/* __closesthit__radiance, optixPathTracer.cu:338-413 */
static void closesthit_radiance(const OrcScene* s, int mode, const OrcLight* light, v3 ray_o, v3 ray_d,
                                const HitRec* h, PRD* prd, OrcStats* st)
{
    const float* vb = s->verts + (size_t)h->prim * 12;
    v3 v0 = v3_ld(vb), v1 = v3_ld(vb + 4), v2 = v3_ld(vb + 8);
    v3 N0 = v3_normalize(v3_cross(v3_sub(v1, v0), v3_sub(v2, v0)));
    v3 neg_d = v3_make(-ray_d.x, -ray_d.y, -ray_d.z);
    v3 N  = v3_scale(N0, copysignf(1.0f, v3_dot(neg_d, N0)));        /* faceforward, vec_math.h:567 */
    v3 P  = v3_add(ray_o, v3_scale(ray_d, h->t));
    uint32_t m = s->mat[h->prim];
    const float* em = s->emission + 3 * m;
    const float* df = s->diffuse + 3 * m;

    prd->emitted = prd->depth == 0 ? v3_ld(em) : v3_make(0.0f, 0.0f, 0.0f);

    uint32_t seed = prd->seed;
    {
        float z1 = orc_rnd(&seed);
        float z2 = orc_rnd(&seed);
        float w_in[3];
        orc_cosine_sample_hemisphere(z1, z2, w_in);
        float nn[3] = { N.x, N.y, N.z };
        orc_onb_inverse_transform(nn, w_in);
        prd->direction   = v3_ld(w_in);
        prd->origin      = P;
        prd->attenuation = v3_mul(prd->attenuation, v3_ld(df));
    }
    float z1 = orc_rnd(&seed);
    float z2 = orc_rnd(&seed);
    prd->seed = seed;

    v3 lv1 = v3_ld(light->v1), lv2 = v3_ld(light->v2);
    v3 light_pos = v3_add(v3_add(v3_ld(light->corner), v3_scale(lv1, z1)), v3_scale(lv2, z2));
    v3    toL   = v3_sub(light_pos, P);
    float Ldist = v3_length(toL);
    v3    L     = v3_normalize(toL);
    float nDl   = v3_dot(N, L);
    float LnDl  = -v3_dot(v3_ld(light->normal), L);

    float weight = 0.0f;
    if (nDl > 0.0f && LnDl > 0.0f) {
        HitRec sh;
        trace(s, mode, P, L, 0.01f, Ldist - 0.01f, 1, &sh, st);       /* traceOcclusion, :218-240 */
        if (!sh.hit) {
            float A = v3_length(v3_cross(lv1, lv2));
            weight  = nDl * LnDl * A / (3.14159265358979323846f * Ldist * Ldist);
        }
    }
    prd->radiance = v3_scale(v3_ld(light->emission), weight);
    prd->done     = 0;
}

/* one pixel of __raygen__rg (optixPathTracer.cu:249-320); returns the spp-averaged colour */
static v3 raygen_pixel(const OrcScene* s, int mode, const OrcParams* p, uint32_t max_depth,
                       uint32_t px, uint32_t py, OrcStats* st)
{
    const int w = (int)p->width, hgt = (int)p->height;
    v3 eye = v3_ld(p->eye), U = v3_ld(p->U), V = v3_ld(p->V), W = v3_ld(p->W);
    uint32_t seed = orc_tea4(py * (uint32_t)w + px, p->subframe_index);
    v3 result = v3_make(0.0f, 0.0f, 0.0f);
    int i = (int)p->samples_per_launch;
    do {
        float jx = orc_rnd(&seed);
        float jy = orc_rnd(&seed);
        float dx = 2.0f * (((float)px + jx) / (float)w) - 1.0f;
        float dy = 2.0f * (((float)py + jy) / (float)hgt) - 1.0f;
        v3 ray_d = v3_normalize(v3_add(v3_add(v3_scale(U, dx), v3_scale(V, dy)), W));
        v3 ray_o = eye;

        PRD prd;
        prd.attenuation = v3_make(1.0f, 1.0f, 1.0f);
        prd.seed        = seed;
        prd.depth       = 0;
        for (;;) {
            HitRec h;
            trace(s, mode, ray_o, ray_d, 0.01f, 1e16f, 0, &h, st);
            if (h.hit) closesthit_radiance(s, mode, &p->light, ray_o, ray_d, &h, &prd, st);
            else       miss_radiance(s, &prd);
            if (st && (uint64_t)prd.depth > st->max_depth_seen) st->max_depth_seen = (uint64_t)prd.depth;

            result = v3_add(result, prd.emitted);
            result = v3_add(result, v3_mul(prd.radiance, prd.attenuation));

            float lum = v3_dot(prd.attenuation, v3_make(0.30f, 0.59f, 0.11f));
            int done = prd.done || orc_rnd(&prd.seed) > lum;
            if (done) break;
            /* build extension: bounded path length (0 = reference behaviour) */
            if (max_depth && (uint32_t)(prd.depth + 1) >= max_depth) break;
            float inv = 1.0f / lum;
            prd.attenuation = v3_scale(prd.attenuation, inv);
            ray_o = prd.origin;
            ray_d = prd.direction;
            ++prd.depth;
        }
    } while (--i);
    float inv_spp = 1.0f / (float)p->samples_per_launch;
    return v3_scale(result, inv_spp);
}

static void resolve_pixel(const OrcParams* p, v3 c, const float* prev4, float* out4, uint8_t* out_rgba)
{
    if (p->subframe_index > 0) {
        float a = 1.0f / (float)(p->subframe_index + 1);
        v3 pr = v3_ld(prev4);
        c = v3_add(pr, v3_scale(v3_sub(c, pr), a));          /* lerp, vec_math.h:500 */
    }
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = 1.0f;
    if (out_rgba) { float cc[3] = { c.x, c.y, c.z }; orc_make_color(cc, out_rgba); }
}

int orc_render_rows(const OrcScene* s, int mode, const OrcParams* p, uint32_t max_depth,
                    uint32_t y0, uint32_t y1, OrcStats* st)
{
    if (mode == ORC_MODE_BVH && !s->nodes) return -1;
    if (!p->samples_per_launch) return -2;
    for (uint32_t y = y0; y < y1 && y < p->height; ++y)
        for (uint32_t x = 0; x < p->width; ++x) {
            size_t idx = (size_t)y * p->width + x;
            v3 c = raygen_pixel(s, mode, p, max_depth, x, y, st);
            resolve_pixel(p, c, p->accum_buffer + idx * 4, p->accum_buffer + idx * 4,
                          p->frame_buffer ? p->frame_buffer + idx * 4 : NULL);
        }
    return 0;
}

int orc_render_pixels(const OrcScene* s, int mode, const OrcParams* p, uint32_t max_depth,
                      const uint32_t* pixel_index, uint32_t n, float* compact_accum4,
                      const float* compact_prev4, OrcStats* st)
{
    if (mode == ORC_MODE_BVH && !s->nodes) return -1;
    if (!p->samples_per_launch) return -2;
    float zero[4] = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t idx = pixel_index[i];
        float* out = compact_accum4 + (size_t)i * 4;
        if (idx >= p->width * p->height) { out[0] = out[1] = out[2] = out[3] = 0.0f; continue; }
        v3 c = raygen_pixel(s, mode, p, max_depth, idx % p->width, idx / p->width, st);
        resolve_pixel(p, c, compact_prev4 ? compact_prev4 + (size_t)i * 4 : zero, out, NULL);
    }
    return 0;
}

void orc_primary_rays(const OrcParams* p, uint32_t y0, uint32_t y1, float* rays8)
{
    v3 eye = v3_ld(p->eye), U = v3_ld(p->U), V = v3_ld(p->V), W = v3_ld(p->W);
    size_t k = 0;
    for (uint32_t y = y0; y < y1; ++y)
        for (uint32_t x = 0; x < p->width; ++x, ++k) {
            uint32_t seed = orc_tea4(y * p->width + x, p->subframe_index);
            float jx = orc_rnd(&seed), jy = orc_rnd(&seed);
            float dx = 2.0f * (((float)x + jx) / (float)p->width) - 1.0f;
            float dy = 2.0f * (((float)y + jy) / (float)p->height) - 1.0f;
            v3 d = v3_normalize(v3_add(v3_add(v3_scale(U, dx), v3_scale(V, dy)), W));
            float* q = rays8 + k * 8;
            q[0] = eye.x; q[1] = eye.y; q[2] = eye.z; q[3] = 0.01f;
            q[4] = d.x;   q[5] = d.y;   q[6] = d.z;   q[7] = 1e16f;
        }
}


/* ------------------------------------------------------------------------- */
/* SDK/imgui_test (the author's sample) -- optixTriangle.cu:103-268           */
/* ------------------------------------------------------------------------- */
void orc_imgui_camera_set_fd(OrcImguiCamera* c, float new_fd)
{
    c->fd = new_fd / v3_length(v3_sub(v3_ld(c->lookat), v3_ld(c->eye)));      /* camera.h:36 */
}

void orc_imgui_camera_compute_uvw(OrcImguiCamera* c)                           /* camera.h:110-123 */
{
    v3 w = v3_scale(v3_sub(v3_ld(c->lookat), v3_ld(c->eye)), c->fd);
    float wlen = v3_length(w);
    v3 u = v3_normalize(v3_cross(w, v3_ld(c->up)));
    v3 v = v3_normalize(v3_cross(u, w));
    float vlen = wlen * tanf(0.5f * c->fov * 3.14159265358979323846f / 180.0f);
    v = v3_scale(v, vlen);
    u = v3_scale(u, vlen);                                                     /* ulen = vlen: square pixels, no aspect */
    c->u[0] = u.x; c->u[1] = u.y; c->u[2] = u.z;
    c->v[0] = v.x; c->v[1] = v.y; c->v[2] = v.z;
    c->w[0] = w.x; c->w[1] = w.y; c->w[2] = w.z;
}

/* Camera::compute_ray, camera.h:127-144.  The two aperture draws are written as arguments of one
 * make_float2 call in the reference; evaluated left to right here (as nvcc/clang do). */
static void imgui_compute_ray(const OrcImguiCamera* c, uint32_t ix, uint32_t iy, uint32_t dimx, uint32_t dimy,
                              v3* origin, v3* direction, uint32_t* seed)
{
    float dx = 2.0f * ((float)ix / (float)dimx) - 1.0f;
    float dy = 2.0f * ((float)iy / (float)dimy) - 1.0f;
    v3 u = v3_ld(c->u), v = v3_ld(c->v), w = v3_ld(c->w), eye = v3_ld(c->eye);
    if (c->ortho) {
        *direction = v3_normalize(v3_add(v3_add(v3_scale(u, dx), v3_scale(v, dy)), w));
        *origin    = v3_add(v3_add(eye, v3_scale(u, dx)), v3_scale(v, dy));
    } else {
        float ax = (orc_rnd(seed) - 0.5f) * c->aperture;
        float ay = (orc_rnd(seed) - 0.5f) * c->aperture;
        dx = dx - ax; dy = dy - ay;
        *direction = v3_normalize(v3_add(v3_add(v3_scale(u, dx), v3_scale(v, dy)), w));
        *origin    = v3_add(v3_add(eye, v3_scale(u, ax)), v3_scale(v, ay));
    }
}

/* LightVariant::wi / lumi, light.h:19-40 (point_light.h:20-25, directional_light.h:19-24, volumetric_light.h:22-27) */
static v3 imgui_light_wi(const OrcImguiLight* l, v3 p, uint32_t* seed)
{
    if (l->tag == 0) return v3_sub(v3_ld(l->a), p);
    if (l->tag == 1) {
        float j = l->b[3];
        float jx = j * orc_rnd(seed), jy = j * orc_rnd(seed), jz = j * orc_rnd(seed);
        return v3_add(v3_ld(l->a), v3_make(jx, jy, jz));
    }
    {
        float r = l->b[0];
        float jx = r * orc_rnd(seed), jy = r * orc_rnd(seed), jz = r * orc_rnd(seed);
        return v3_sub(v3_add(v3_ld(l->a), v3_make(jx, jy, jz)), p);
    }
}
static v3 imgui_light_lumi(const OrcImguiLight* l)
{
    if (l->tag == 2) return v3_make(l->b[1], l->b[2], l->b[3]);
    return v3_make(l->b[0], l->b[1], l->b[2]);
}

/* exported for the known-answer tests (tests/golden/ref_kat_imgui.json, made by oracle/ref_kat_imgui.cpp from the reference's
 * own camera.h / light.h / diffuse.h) */
void orc_imgui_compute_ray(const OrcImguiCamera* c, uint32_t ix, uint32_t iy, uint32_t dimx, uint32_t dimy, uint32_t* seed,
                           float origin[3], float direction[3])
{
    v3 o, d;
    imgui_compute_ray(c, ix, iy, dimx, dimy, &o, &d, seed);
    origin[0] = o.x; origin[1] = o.y; origin[2] = o.z; direction[0] = d.x; direction[1] = d.y; direction[2] = d.z;
}
void orc_imgui_light_wi(const OrcImguiLight* l, const float p[3], uint32_t* seed, float wi[3], float lumi[3])
{
    v3 r = imgui_light_wi(l, v3_ld(p), seed), e = imgui_light_lumi(l);
    wi[0] = r.x; wi[1] = r.y; wi[2] = r.z; lumi[0] = e.x; lumi[1] = e.y; lumi[2] = e.z;
}

/* __closesthit__ch, optixTriangle.cu:174-268 */
static v3 imgui_closesthit(const OrcScene* s, int mode, const OrcImguiFrame* f, v3 ray_o, v3 ray_d, const HitRec* h,
                           uint32_t ix, uint32_t iy, OrcStats* st)
{
    const float* nb = f->normals + (size_t)h->prim * 9;
    v3 n0 = v3_ld(nb), n1 = v3_ld(nb + 3), n2 = v3_ld(nb + 6);
    v3 normal = v3_add(v3_add(v3_scale(n1, h->u), v3_scale(n2, h->v)), v3_scale(n0, 1.0f - h->u - h->v));
    v3 P = v3_add(v3_add(ray_o, v3_scale(ray_d, h->t)), v3_scale(normal, 0.0001f));
    uint32_t seed = orc_tea4(ix + f->image_width * iy, f->dt);          /* re-seeded identically for every sample (:201) */
    v3 result = v3_make(0.0f, 0.0f, 0.0f);
    int mi = f->mat_indices[h->prim];
    if (mi < 0 || mi >= f->nmaterials) mi = 0;
    v3 color = v3_ld(f->materials + 3 * mi);
    for (int li = 0; li < f->nlights; ++li) {
        v3 wi = imgui_light_wi(&f->lights[li], P, &seed);
        float ndotwi = v3_dot(normal, wi);
        HitRec sh;
        trace(s, mode, P, wi, 0.01f, 1.0f, 1, &sh, st);                   /* tmax 1.0 in units of the un-normalised wi */
        if (!(sh.hit || ndotwi < 0.0f))
            result = v3_add(result, v3_scale(v3_mul(color, imgui_light_lumi(&f->lights[li])), ndotwi));
        else
            result = v3_add(result, v3_make(0.0f, 0.0f, 0.0f));
    }
    float nrm[3] = { normal.x, normal.y, normal.z };
    float u1 = orc_rnd(&seed), u2 = orc_rnd(&seed);
    float out[3];
    orc_cosine_sample_hemisphere(u1, u2, out);
    orc_onb_inverse_transform(nrm, out);                                   /* Onb keeps the un-normalised normal as its z axis */
    HitRec ao;
    trace(s, mode, P, v3_ld(out), 0.01f, 1e16f, 1, &ao, st);               /* only hit / no hit is used (:243-267) */
    v3 ambient = v3_make(0.01f, 0.01f, 0.01f);
    result = v3_add(result, ao.hit ? ambient : v3_mul(ambient, color));
    return result;
}

int orc_imgui_render_rows(const OrcScene* s, int mode, const OrcImguiFrame* f, uint32_t y0, uint32_t y1, OrcStats* st)
{
    if (mode == ORC_MODE_BVH && !s->nodes) return -1;
    const uint32_t W = f->image_width, H = f->image_height;
    for (uint32_t y = y0; y < y1 && y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            const size_t index = (size_t)y * W + x;
            v3 result = v3_make(0.0f, 0.0f, 0.0f);
            uint32_t seed = orc_tea4(x + W * y, f->dt);
            for (uint32_t i = 0; i < f->samples_per_frame; ++i) {
                v3 ro, rd;
                imgui_compute_ray(f->camera, x, y, W, H, &ro, &rd, &seed);
                HitRec h;
                trace(s, mode, ro, rd, 0.0f, 1e16f, 0, &h, st);
                v3 payload = h.hit ? imgui_closesthit(s, mode, f, ro, rd, &h, x, y, st)
                                   : v3_add(v3_scale(rd, 0.5f), v3_make(0.5f, 0.5f, 0.5f));     /* __miss__ms :153-171 */
                result = v3_add(result, payload);
            }
            float* film = f->film + index * 3;
            v3 fl = f->dirty ? result : v3_add(v3_ld(film), result);
            film[0] = fl.x; film[1] = fl.y; film[2] = fl.z;
            if (f->image) {
                float inv = 1.0f / (float)f->dt;
                float c[3] = { fl.x * inv, fl.y * inv, fl.z * inv };
                orc_make_color(c, f->image + index * 4);
            }
        }
    return 0;
}


/* ------------------------------------------------------------------------- */
/* SDK/cuda/whitted.cu -- glTF PBR direct lighting (opaque, untextured)       */
/* ------------------------------------------------------------------------- */
static inline float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }   /* powf(x, 5) as products (detmath) */

/* whitted_cuda.h:48-72 */
static v3 w_schlick(v3 spec, float VdH)
{
    float p = pow5(1.0f - VdH);
    return v3_add(spec, v3_scale(v3_sub(v3_make(1.0f, 1.0f, 1.0f), spec), p));
}
static float w_vis(float NdL, float NdV, float alpha)
{
    float a2 = alpha * alpha;
    float g0 = NdL * sqrtf(NdV * NdV * (1.0f - a2) + a2);
    float g1 = NdV * sqrtf(NdL * NdL * (1.0f - a2) + a2);
    return 2.0f * NdL * NdV / (g0 + g1);
}
static float w_ggx(float NdH, float alpha)
{
    float a2 = alpha * alpha;
    float x = NdH * NdH * (a2 - 1.0f) + 1.0f;
    return a2 / (3.14159265358979323846f * x * x);
}

/* ---- texture fetch: what tex2D<float4> does for the objects of Scene::addSampler (Scene.cpp:624-652) ---- */
static int wrap_index(int i, int n, uint32_t mode)
{
    if (mode == 1u) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    if (mode == 2u) { int p = i % (2 * n); if (p < 0) p += 2 * n; return p < n ? p : 2 * n - 1 - p; }
    int p = i % n; return p < 0 ? p + n : p;
}
static void fetch_texel(const OrcTextureW* t, int x, int y, float o[4])
{
    size_t at = ((size_t)y * t->width + (size_t)x) * 4;
    if (t->bits == 16) { const uint16_t* p = (const uint16_t*)t->texels + at; for (int k = 0; k < 4; ++k) o[k] = (float)p[k] / 65535.0f; }
    else               { const uint8_t*  p = (const uint8_t*)t->texels + at;  for (int k = 0; k < 4; ++k) o[k] = (float)p[k] / 255.0f; }
}
void orc_sample_texture(const OrcTextureW* t, float u, float v, float out[4])
{
    const int W = (int)t->width, H = (int)t->height;
    if (t->filter == 0u) {
        int x = wrap_index((int)floorf(u * (float)W), W, t->address_s), y = wrap_index((int)floorf(v * (float)H), H, t->address_t);
        fetch_texel(t, x, y, out);
        return;
    }
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float a = fx - x0f, b = fy - y0f;
    int x0 = wrap_index((int)x0f, W, t->address_s), x1 = wrap_index((int)x0f + 1, W, t->address_s);
    int y0 = wrap_index((int)y0f, H, t->address_t), y1 = wrap_index((int)y0f + 1, H, t->address_t);
    float t00[4], t10[4], t01[4], t11[4];
    fetch_texel(t, x0, y0, t00); fetch_texel(t, x1, y0, t10); fetch_texel(t, x0, y1, t01); fetch_texel(t, x1, y1, t11);
    for (int k = 0; k < 4; ++k) {
        float top = t00[k] + a * (t10[k] - t00[k]);
        float bot = t01[k] + a * (t11[k] - t01[k]);
        out[k] = top + b * (bot - top);
    }
}

typedef struct { float u, v; v3 dpdu, dpdv; } TexcoordW;
typedef struct { v3 P, N, Ng; TexcoordW tc[2]; float color[4]; } LocalGeomW;

/* getLocalGeometry (cuda/LocalGeometry.h:59-176) on world-space triangles.  dpdu / dpdv come from the world-space
 * corners (the reference takes object-space ones even under an instance transform). */
static void local_geometry(const OrcScene* s, const OrcWhittedFrame* f, const HitRec* h, LocalGeomW* g)
{
    const float* vb = s->verts + (size_t)h->prim * 12;
    v3 P0 = v3_ld(vb), P1 = v3_ld(vb + 4), P2 = v3_ld(vb + 8);
    float b0 = 1.0f - h->u - h->v;
    g->P = v3_add(v3_add(v3_scale(P0, b0), v3_scale(P1, h->u)), v3_scale(P2, h->v));
    for (int k = 0; k < 4; ++k) g->color[k] = 1.0f;
    if (f->colors4 && (!f->color_present || f->color_present[h->prim])) {
        const float* c = f->colors4 + (size_t)h->prim * 12;
        for (int k = 0; k < 4; ++k) g->color[k] = (b0 * c[k] + h->u * c[4 + k]) + h->v * c[8 + k];
    }
    g->Ng = v3_normalize(v3_cross(v3_sub(P1, P0), v3_sub(P2, P0)));
    g->N = g->Ng;
    if (f->normals4) {
        const float* nb = f->normals4 + (size_t)h->prim * 12;
        if (nb[3] != 0.0f) {
            v3 n = v3_add(v3_add(v3_scale(v3_ld(nb), b0), v3_scale(v3_ld(nb + 4), h->u)), v3_scale(v3_ld(nb + 8), h->v));
            g->N = v3_normalize(n);
        }
    }
    v3 dp1 = v3_sub(P0, P2), dp2 = v3_sub(P1, P2);
    for (int j = 0; j < 2; ++j) {
        TexcoordW* tc = &g->tc[j];
        if (f->texcoords[j] && (!f->texcoord_present[j] || f->texcoord_present[j][h->prim])) {
            const float* uv = f->texcoords[j] + (size_t)h->prim * 6;
            tc->u = (b0 * uv[0] + h->u * uv[2]) + h->v * uv[4];
            tc->v = (b0 * uv[1] + h->u * uv[3]) + h->v * uv[5];
            float du1 = uv[0] - uv[4], du2 = uv[2] - uv[4], dv1 = uv[1] - uv[5], dv2 = uv[3] - uv[5];
            float det = du1 * dv2 - dv1 * du2;
            float invdet = 1.0f / det;
            tc->dpdu = v3_scale(v3_sub(v3_scale(dp1, dv2), v3_scale(dp2, dv1)), invdet);
            tc->dpdv = v3_scale(v3_add(v3_scale(dp1, -du2), v3_scale(dp2, du1)), invdet);
        } else {
            tc->u = h->u; tc->v = h->v;                                  /* UV = barys */
            tc->dpdu = v3_make(-dp1.x, -dp1.y, -dp1.z);
            tc->dpdv = v3_add(v3_make(-dp1.x, -dp1.y, -dp1.z), dp2);
        }
    }
}

/* sampleTexture<float4> (cuda/LocalShading.h:37-53); returns 0 when the material has no such texture (or its
 * image could not be decoded), 1 + the sample otherwise */
static int sample_material_texture(const OrcWhittedFrame* f, const OrcTexInfoW* ti, const LocalGeomW* g, float out[4])
{
    if (ti->texture < 0 || (uint32_t)ti->texture >= f->ntextures) return 0;
    const OrcTextureW* t = &f->textures[ti->texture];
    if (t->width == 0 || !t->texels) return 0;
    const TexcoordW* tc = &g->tc[ti->texcoord == 1 ? 1 : 0];
    float ux = tc->u * ti->scale[0], uy = tc->v * ti->scale[1];
    float tu = (ux * ti->rotation[1] + uy * ti->rotation[0]) + ti->offset[0];
    float tv = (ux * -ti->rotation[0] + uy * ti->rotation[1]) + ti->offset[1];
    orc_sample_texture(t, tu, tv, out);
    return 1;
}

#define WH_MAX_TRACE_DEPTH 8u                     /* whitted::MAX_TRACE_DEPTH, cuda/whitted.h:42 */

typedef struct { const OrcScene* s; const OrcWhittedFrame* f; float occlusion; } AnyhitCtx;

static const OrcMaterialW* prim_material(const OrcWhittedFrame* f, uint32_t prim)
{
    uint32_t mi = f->material_index ? f->material_index[prim] : 0u;
    if (mi >= f->nmaterials) mi = f->nmaterials - 1u;
    return &f->materials[mi];
}

/* alpha of the base-colour texture at a candidate hit; 0 = the material has no (decodable) base-colour texture */
static int candidate_alpha(const AnyhitCtx* c, uint32_t prim, float t, float u, float v, const OrcMaterialW** m_out, float* alpha)
{
    const OrcMaterialW* m = prim_material(c->f, prim);
    *m_out = m;
    HitRec h; h.hit = 1; h.t = t; h.u = u; h.v = v; h.prim = prim;
    LocalGeomW g;
    local_geometry(c->s, c->f, &h, &g);
    float tx[4];
    if (!sample_material_texture(c->f, &m->base_color_tex, &g, tx)) return 0;
    *alpha = tx[3];
    return 1;
}

/* __anyhit__radiance, cuda/whitted.cu:100-111: "force mask mode, even for blend mode" */
static int anyhit_radiance(void* ctx, uint32_t prim, float t, float u, float v)
{
    const AnyhitCtx* c = (const AnyhitCtx*)ctx;
    const OrcMaterialW* m; float alpha;
    if (candidate_alpha(c, prim, t, u, v, &m, &alpha) && alpha < m->alpha_cutoff) return 0;
    return 1;
}

/* __anyhit__occlusion, cuda/whitted.cu:113-136, at ONE candidate: 1 = the ray terminates there (occluded), 0 = ignored (below the
 * cutoff, or attenuating: c->occlusion is then the product so far) */
static int anyhit_occlusion(AnyhitCtx* c, uint32_t prim, float t, float u, float v)
{
    const OrcMaterialW* m; float alpha;
    if (candidate_alpha(c, prim, t, u, v, &m, &alpha) && m->alpha_mode != 0) {
        if (m->alpha_mode == 1 && alpha < m->alpha_cutoff) return 0;
        float attenuation = c->occlusion * (1.0f - alpha);
        if (attenuation > 0.0f) { c->occlusion = attenuation; return 0; }
    }
    return 1;
}

/* OptiX leaves the order in which a ray's candidates meet the any-hit program open, and a product of floats depends on it.  The
 * build's definition (csrc/whitted_kernels.hip: the deferred any-hit programs): the candidates of the ray's interval in the order
 * of (t, primitive index).  Restated here the direct way: the traversal only COLLECTS the candidates whose any-hit program is
 * enabled (it stops at one whose program is disabled: opaque), then they are sorted and the program runs over them in order. */
typedef struct { float t, u, v; uint32_t prim; } OccCand;
typedef struct { OccCand* v; size_t n, cap; } OccList;

static int occ_collect(void* ctx, uint32_t prim, float t, float u, float v)
{
    OccList* l = (OccList*)ctx;
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 16;
        l->v = (OccCand*)realloc(l->v, l->cap * sizeof(OccCand));
        if (!l->v) abort();
    }
    OccCand c = { t, u, v, prim };
    l->v[l->n++] = c;
    return 0;                                              /* never accepted during the traversal */
}

static int occ_order(const void* a, const void* b)
{
    const OccCand* x = (const OccCand*)a; const OccCand* y = (const OccCand*)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->prim < y->prim ? -1 : (x->prim > y->prim ? 1 : 0);
}

/* whitted::traceOcclusion, cuda/whitted_cuda.h:127-160: pending attenuation, committed by the miss program only */
static float whitted_trace_occlusion(const OrcScene* s, int mode, const OrcWhittedFrame* f, v3 o, v3 d, float tmin, float tmax, OrcStats* st)
{
    OccList l = { NULL, 0, 0 };
    TraceFilter flt = { 0u, occ_collect, &l };
    HitRec sh;
    trace_ex(s, mode, o, d, tmin, tmax, 1, &sh, st, &flt);
    AnyhitCtx c = { s, f, 1.0f };
    int occluded = sh.hit;                                 /* a candidate without any-hit program */
    if (!occluded && l.n) {
        qsort(l.v, l.n, sizeof(OccCand), occ_order);
        for (size_t i = 0; i < l.n && !occluded; ++i) {
            /* a tree with spatial splits (MIPT_OPT_BUILD = 3) references a triangle from several leaves: the same candidate, met twice, counts once */
            if (i && l.v[i].prim == l.v[i - 1].prim && l.v[i].t == l.v[i - 1].t) continue;
            occluded = anyhit_occlusion(&c, l.v[i].prim, l.v[i].t, l.v[i].u, l.v[i].v);
        }
    }
    free(l.v);
    return occluded ? 0.0f : c.occlusion;
}

static v3 whitted_closesthit(const OrcScene* s, int mode, const OrcWhittedFrame* f, v3 ray_o, v3 ray_d, const HitRec* h,
                             uint32_t payload_depth, OrcStats* st);

/* whitted::traceRadiance, cuda/whitted_cuda.h:88-119 (OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) + the miss program */
static v3 whitted_trace_radiance(const OrcScene* s, int mode, const OrcWhittedFrame* f, v3 o, v3 d, float tmin, float tmax,
                                 uint32_t payload_depth, OrcStats* st)
{
    AnyhitCtx c = { s, f, 1.0f };
    TraceFilter flt = { 16u, anyhit_radiance, &c };
    HitRec h;
    trace_ex(s, mode, o, d, tmin, tmax, 0, &h, st, &flt);
    return h.hit ? whitted_closesthit(s, mode, f, o, d, &h, payload_depth, st) : v3_ld(f->miss_color);
}

/* __closesthit__radiance, cuda/whitted.cu:149-289 */
static v3 whitted_closesthit(const OrcScene* s, int mode, const OrcWhittedFrame* f, v3 ray_o, v3 ray_d, const HitRec* h,
                             uint32_t payload_depth, OrcStats* st)
{
    LocalGeomW g;
    local_geometry(s, f, h, &g);
    const OrcMaterialW* m = prim_material(f, h->prim);
    float bc[4];
    for (int k = 0; k < 4; ++k) bc[k] = m->base_color[k] * g.color[k];
    float tx[4];
    if (sample_material_texture(f, &m->base_color_tex, &g, tx)) {
        bc[0] *= orc_det_pow22(tx[0]); bc[1] *= orc_det_pow22(tx[1]); bc[2] *= orc_det_pow22(tx[2]);   /* linearize, not the alpha */
        bc[3] *= tx[3];
    }
    v3 base = v3_make(bc[0], bc[1], bc[2]);
    float metallic = m->metallic, roughness = m->roughness;
    float mr[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    sample_material_texture(f, &m->metallic_roughness_tex, &g, mr);          /* (occlusion, roughness, metallic) */
    roughness *= mr[1];
    metallic *= mr[2];
    const float F0 = 0.04f;
    v3 diff_color = v3_scale(v3_scale(base, 1.0f - F0), 1.0f - metallic);
    v3 f0v = v3_make(F0, F0, F0);
    v3 spec_color = v3_add(f0v, v3_scale(v3_sub(base, f0v), metallic));           /* lerp, vec_math.h:500 */
    float alpha = roughness * roughness;
    v3 result = v3_make(0.0f, 0.0f, 0.0f);
    float em[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    sample_material_texture(f, &m->emissive_tex, &g, em);
    result = v3_add(result, v3_mul(v3_ld(m->emissive_factor), v3_make(em[0], em[1], em[2])));
    v3 N = g.N;
    float nt[4];
    if (sample_material_texture(f, &m->normal_tex, &g, nt)) {
        float nx = 2.0f * nt[0] - 1.0f, ny = 2.0f * nt[1] - 1.0f, nz = 2.0f * nt[2] - 1.0f;
        const float* rot = m->normal_tex.rotation;
        float tx_ = nx * rot[1] + ny * -rot[0];
        float ty_ = nx * rot[0] + ny * rot[1];
        const TexcoordW* tc = &g.tc[m->normal_tex.texcoord == 1 ? 1 : 0];
        N = v3_normalize(v3_add(v3_add(v3_scale(v3_normalize(tc->dpdu), tx_), v3_scale(v3_normalize(tc->dpdv), ty_)), v3_scale(g.N, nz)));
    }
    if (v3_dot(N, ray_d) > 0.0f) N = v3_make(-N.x, -N.y, -N.z);
    const uint32_t depth = payload_depth + 1u;
    const v3 P = g.P;
    for (uint32_t i = 0; i < f->nlights; ++i) {
        const OrcLightW* l = &f->lights[i];
        if (l->type == 0) {
            if (depth < WH_MAX_TRACE_DEPTH) {
                v3 toL = v3_sub(v3_ld(l->position), P);
                float Ld = v3_length(toL);
                float invd = 1.0f / Ld;
                v3 L = v3_scale(toL, invd);
                v3 nd = v3_normalize(ray_d);
                v3 Vv = v3_make(-nd.x, -nd.y, -nd.z);
                v3 H = v3_normalize(v3_add(L, Vv));
                float NdL = v3_dot(N, L), NdV = v3_dot(N, Vv), NdH = v3_dot(N, H), VdH = v3_dot(Vv, H);
                if (NdL > 0.0f && NdV > 0.0f) {
                    float attenuation = whitted_trace_occlusion(s, mode, f, P, L, 0.001f, Ld - 0.001f, st);
                    if (attenuation > 0.0f) {
                        v3 F = w_schlick(spec_color, VdH);
                        float G = w_vis(NdL, NdV, alpha);
                        float D = w_ggx(NdH, alpha);
                        v3 one = v3_make(1.0f, 1.0f, 1.0f);
                        v3 diff = v3_scale(v3_mul(v3_sub(one, F), diff_color), 1.0f / 3.14159265358979323846f);
                        v3 spec = v3_scale(v3_scale(F, G), D);
                        v3 lc = v3_scale(v3_scale(v3_scale(v3_ld(l->color), attenuation), l->intensity), NdL);
                        result = v3_add(result, v3_mul(lc, v3_add(diff, spec)));
                    }
                }
            }
        } else if (l->type == 1) {
            result = v3_add(result, v3_mul(v3_ld(l->color), base));
        }
    }
    if (m->alpha_mode == 2) {                                                    /* ALPHA_MODE_BLEND, :268-286 */
        result = v3_scale(result, bc[3]);
        if (depth < WH_MAX_TRACE_DEPTH) {
            v3 next = whitted_trace_radiance(s, mode, f, ray_o, ray_d, h->t, 1e16f, depth, st);
            float w = 1.0f - bc[3];
            result = v3_add(result, v3_mul(next, v3_make(w, w, w)));
        }
    }
    return result;
}

int orc_whitted_render_rows(const OrcScene* s, int mode, const OrcWhittedFrame* f, uint32_t y0, uint32_t y1, OrcStats* st)
{
    if (mode == ORC_MODE_BVH && !s->nodes) return -1;
    v3 eye = v3_ld(f->eye), U = v3_ld(f->U), V = v3_ld(f->V), W = v3_ld(f->W);
    for (uint32_t y = y0; y < y1 && y < f->height; ++y)
        for (uint32_t x = 0; x < f->width; ++x) {
            size_t idx = (size_t)y * f->width + x;
            uint32_t seed = orc_tea4(y * f->width + x, f->subframe_index);
            float jx = 0.5f, jy = 0.5f;                                          /* pixel centre at subframe 0 (:60-61) */
            if (f->subframe_index != 0) { jx = orc_rnd(&seed); jy = orc_rnd(&seed); }
            float dx = 2.0f * (((float)x + jx) / (float)f->width) - 1.0f;
            float dy = 2.0f * (((float)y + jy) / (float)f->height) - 1.0f;
            v3 rd = v3_normalize(v3_add(v3_add(v3_scale(U, dx), v3_scale(V, dy)), W));
            v3 c = whitted_trace_radiance(s, mode, f, eye, rd, 0.0f, 1e16f, 0u, st);
            float* acc = f->accum_buffer + idx * 4;
            if (f->subframe_index > 0) {
                float a = 1.0f / (float)(f->subframe_index + 1);
                v3 pr = v3_ld(acc);
                c = v3_add(pr, v3_scale(v3_sub(c, pr), a));
            }
            acc[0] = c.x; acc[1] = c.y; acc[2] = c.z; acc[3] = 1.0f;
            if (f->frame_buffer) { float cc[3] = { c.x, c.y, c.z }; orc_make_color(cc, f->frame_buffer + idx * 4); }
        }
    return 0;
}
