// lbvh_build.hip -- device-side BVH2 build (the counterpart of OPTIX_BUILD_FLAG_PREFER_FAST_BUILD).
// Linear BVH: 63-bit Morton codes of the triangle centroids, radix sort (rocPRIM through hipCUB), the binary radix
// tree of Karras 2012 built fully in parallel, bottom-up box fitting, then subtrees of <= leaf_max triangles are
// folded into leaves and the surviving inner nodes are renumbered (scan) and written in the 64-byte two-child layout
// the traversal kernels read.  Same node / triangle layout as the host's binned-SAH builder (bvh_build.cpp), lower
// tree quality, a few milliseconds instead of ~0.2 s per million triangles.  Results of a render do not depend on
// which builder made the tree (closest hit: min t, tie -> lowest primitive).
#include <hipcub/hipcub.hpp>

#include "kernels.h"
#include "detmath.h"

namespace mipt {

namespace {

__device__ __forceinline__ uint32_t f_ord(float f) { const uint32_t b = f2u(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float    f_unord(uint32_t k) { return u2f(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

struct Box6 { float lo[3], hi[3]; };

__global__ __launch_bounds__(256) void k_lbvh_prims(const float* __restrict__ world, uint32_t n, Box6* boxes, uint32_t* scene /* 6 ordered words */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* v = world + (size_t)i * 9;
    Box6 b;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = fminf(v[k], fminf(v[3 + k], v[6 + k]));
        b.hi[k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
    }
    boxes[i] = b;
    for (int k = 0; k < 3; ++k) {                       // centroid bounds are enough for the Morton grid
        const float c = 0.5f * (b.lo[k] + b.hi[k]);
        atomicMin(&scene[k], f_ord(c));
        atomicMax(&scene[3 + k], f_ord(c));
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long x)
{
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void k_lbvh_morton(const Box6* boxes, uint32_t n, const uint32_t* scene, unsigned long long* keys, uint32_t* vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long code = 0;
    for (int k = 0; k < 3; ++k) {
        const float lo = f_unord(scene[k]), hi = f_unord(scene[3 + k]);
        const float c = 0.5f * (boxes[i].lo[k] + boxes[i].hi[k]);
        const float ext = hi - lo;
        float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const unsigned long long q = (unsigned long long)fminf(t * 2097152.0f, 2097151.0f);
        code |= spread21(q) << (2 - k);
    }
    keys[i] = code;
    vals[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j; duplicates are told apart by their position
__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned int)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Karras 2012, Algorithm "Construct binary radix tree": internal node i covers sorted range [lo, hi]
__global__ __launch_bounds__(256) void k_lbvh_hierarchy(const unsigned long long* keys, int n, int* left, int* right, int* parent_inner, int* parent_leaf,
                                                       int* range_lo, int* range_hi)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    // children: a leaf (encoded ~position) when the sub-range is one element
    const int lc = lo == gamma ? ~gamma : gamma;
    const int rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    left[i] = lc; right[i] = rc; range_lo[i] = lo; range_hi[i] = hi;
    if (lc >= 0) parent_inner[lc] = i; else parent_leaf[~lc] = i;
    if (rc >= 0) parent_inner[rc] = i; else parent_leaf[~rc] = i;
    if (i == 0) parent_inner[0] = -1;
}

__device__ __forceinline__ Box6 box_union(const Box6& a, const Box6& b)
{
    Box6 r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fminf(a.lo[k], b.lo[k]); r.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    return r;
}

// bottom-up: the second thread to arrive at a node computes its box (its children's are final by then)
__global__ __launch_bounds__(256) void k_lbvh_fit(const Box6* prim_boxes, const uint32_t* sorted_prims, int n, const int* left, const int* right,
                                                 const int* parent_inner, const int* parent_leaf, Box6* node_boxes, uint32_t* arrived)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n) return;
    int node = parent_leaf[k];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;
        const int lc = left[node], rc = right[node];
        // the sibling's box was written before its __threadfence + atomicAdd: read it past the L1 (agent-scope loads)
        Box6 a, b;
        for (int c = 0; c < 2; ++c) {
            const int ch = c == 0 ? lc : rc;
            Box6& dst = c == 0 ? a : b;
            if (ch >= 0) {
                const float* src = &node_boxes[ch].lo[0];
                for (int k = 0; k < 6; ++k) (&dst.lo[0])[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else dst = prim_boxes[sorted_prims[~ch]];
        }
        node_boxes[node] = box_union(a, b);
        node = parent_inner[node];
    }
}

__global__ __launch_bounds__(256) void k_lbvh_flags(const int* range_lo, const int* range_hi, int n_inner, int leaf_max, uint32_t* emit)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner) return;
    emit[i] = (i == 0 || range_hi[i] - range_lo[i] + 1 > leaf_max) ? 1u : 0u;
}

__device__ __forceinline__ void put_child(MiptBvhNode& nd, int c, const Box6& b, int ref)
{
    if (c == 0) { nd.c0x[0] = b.lo[0]; nd.c0x[1] = b.hi[0]; nd.c0y[0] = b.lo[1]; nd.c0y[1] = b.hi[1]; nd.c0z[0] = b.lo[2]; nd.c0z[1] = b.hi[2]; nd.c0 = ref; }
    else        { nd.c1x[0] = b.lo[0]; nd.c1x[1] = b.hi[0]; nd.c1y[0] = b.lo[1]; nd.c1y[1] = b.hi[1]; nd.c1z[0] = b.lo[2]; nd.c1z[1] = b.hi[2]; nd.c1 = ref; }
}

__global__ __launch_bounds__(256) void k_lbvh_emit(const int* left, const int* right, const int* range_lo, const int* range_hi, int n_inner, int leaf_max,
                                                  const uint32_t* emit, const uint32_t* new_index, const Box6* prim_boxes, const uint32_t* sorted_prims,
                                                  const Box6* node_boxes, MiptBvhNode* out)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner || !emit[i]) return;
    MiptBvhNode nd;
    nd.pad0 = 0u; nd.pad1 = 0u;
    for (int c = 0; c < 2; ++c) {
        const int ch = c == 0 ? left[i] : right[i];
        if (ch < 0) {                                                           // a single triangle
            put_child(nd, c, prim_boxes[sorted_prims[~ch]], ~(int)(((uint32_t)(~ch) << 4) | 1u));
        } else if (!emit[ch]) {                                                 // a subtree folded into one leaf
            const int lo = range_lo[ch], cnt = range_hi[ch] - lo + 1;
            put_child(nd, c, node_boxes[ch], ~(int)(((uint32_t)lo << 4) | (uint32_t)cnt));
        } else {
            put_child(nd, c, node_boxes[ch], (int)new_index[ch]);
        }
    }
    out[new_index[i]] = nd;
}

__global__ __launch_bounds__(256) void k_lbvh_tris(const float* __restrict__ world, const uint32_t* sorted_prims, const uint32_t* gflags, uint32_t n, MiptBvhTri* out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = sorted_prims[k];
    const float* v = world + (size_t)p * 9;
    MiptBvhTri t;
    for (int c = 0; c < 3; ++c) { t.v0[c] = v[c]; t.e1[c] = v[3 + c] - v[c]; t.e2[c] = v[6 + c] - v[c]; }
    t.prim = p; t.pad1 = gflags ? gflags[p] : 0u; t.pad2 = 0u;
    out[k] = t;
}

// inner nodes on the path from the root to each triangle's leaf (= traversal stack entries a ray can need)
__global__ __launch_bounds__(256) void k_lbvh_depth(const int* parent_inner, const int* parent_leaf, const uint32_t* emit, int n, uint32_t* depth)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n) return;
    uint32_t d = 0;
    for (int node = parent_leaf[k]; node >= 0; node = parent_inner[node]) d += emit[node];
    atomicMax(depth, d);
}

// ---- refit (mipt_accel_update): same topology, new vertex positions -------------------------------------------------------
// triangle records of the existing leaf order from the new world-space vertices
__global__ __launch_bounds__(256) void k_refit_tris(const float* __restrict__ world, uint32_t n, MiptBvhTri* tris)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = tris[k].prim;
    if (p >= n) return;
    const float* v = world + (size_t)p * 9;
    for (int c = 0; c < 3; ++c) { tris[k].v0[c] = v[c]; tris[k].e1[c] = v[3 + c] - v[c]; tris[k].e2[c] = v[6 + c] - v[c]; }
}

// parent[child] = node * 2 + slot for every inner child; the root keeps -1
__global__ __launch_bounds__(256) void k_refit_parents(const MiptBvhNode* nodes, uint32_t nn, int* parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    const int c0 = nodes[i].c0, c1 = nodes[i].c1;
    if (c0 >= 0 && (uint32_t)c0 < nn) parent[c0] = (int)(i * 2u);
    if (c1 >= 0 && (uint32_t)c1 < nn) parent[c1] = (int)(i * 2u + 1u);
}

__device__ __forceinline__ void refit_store_slot(MiptBvhNode* nd, int slot, const float lo[3], const float hi[3])
{
    float* x = slot == 0 ? nd->c0x : nd->c1x;
    float* y = slot == 0 ? nd->c0y : nd->c1y;
    float* z = slot == 0 ? nd->c0z : nd->c1z;
    x[0] = lo[0]; x[1] = hi[0]; y[0] = lo[1]; y[1] = hi[1]; z[0] = lo[2]; z[1] = hi[2];
}

// Every thread fits the leaf children of its node from the new vertices and then reports the node "done"; a node is complete
// when its own thread and every inner child have reported, and whoever reports last carries the node's box (the union of
// its two child boxes) into the parent's slot and reports there.  No thread ever waits for another.
__global__ __launch_bounds__(256) void k_refit_nodes(MiptBvhNode* nodes, uint32_t nn, const MiptBvhTri* tris, uint32_t nt,
                                                     const float* __restrict__ world, const int* parent, uint32_t* arrived)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    {
        MiptBvhNode* nd = &nodes[i];
        for (int slot = 0; slot < 2; ++slot) {
            const int ref = slot == 0 ? nd->c0 : nd->c1;
            if (ref >= 0) continue;
            const uint32_t desc = ~(uint32_t)ref, first = desc >> 4, cnt = desc & 15u;
            float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
            for (uint32_t j = 0; j < cnt && first + j < nt; ++j) {
                const uint32_t p = tris[first + j].prim;
                if (p >= nt) continue;
                const float* v = world + (size_t)p * 9;
                for (int c = 0; c < 9; ++c) { const float f = v[c]; lo[c % 3] = fminf(lo[c % 3], f); hi[c % 3] = fmaxf(hi[c % 3], f); }
            }
            refit_store_slot(nd, slot, lo, hi);
        }
    }
    for (;;) {
        MiptBvhNode* nd = &nodes[i];
        const uint32_t need = 1u + (nd->c0 >= 0 ? 1u : 0u) + (nd->c1 >= 0 ? 1u : 0u);
        __threadfence();
        if (atomicAdd(&arrived[i], 1u) + 1u < need) return;
        __threadfence();
        const int pr = parent[i];
        if (pr < 0) return;                                   // the root is complete
        // the boxes of node i were written by other threads before their fence + atomic: read them past the L1
        float b[12];
        const float* src = nd->c0x;
        for (int k = 0; k < 12; ++k) b[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // b: c0x[2] c0y[2] c1x[2] c1y[2] c0z[2] c1z[2]
        const float lo[3] = { fminf(b[0], b[4]), fminf(b[2], b[6]), fminf(b[8], b[10]) };
        const float hi[3] = { fmaxf(b[1], b[5]), fmaxf(b[3], b[7]), fmaxf(b[9], b[11]) };
        refit_store_slot(&nodes[pr >> 1], pr & 1, lo, hi);
        i = (uint32_t)(pr >> 1);
    }
}

// build-input assembly on the device: one OptixBuildInputTriangleArray-like input -> world-space triangles, SBT index
// and geometry flags per triangle, appended at `base`
__global__ __launch_bounds__(256) void k_gather_input(GatherInput in, float* world, uint32_t* mats, uint32_t* gflags, uint32_t base, uint32_t* error)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= in.num_triangles) return;
    uint32_t vi[3];
    if (in.index_format == 2u) {
        const unsigned short* ip = (const unsigned short*)((const char*)in.indices + (size_t)t * in.index_stride);
        vi[0] = ip[0]; vi[1] = ip[1]; vi[2] = ip[2];
    } else if (in.index_format == 4u) {
        const uint32_t* ip = (const uint32_t*)((const char*)in.indices + (size_t)t * in.index_stride);
        vi[0] = ip[0]; vi[1] = ip[1]; vi[2] = ip[2];
    } else { vi[0] = 3u * t; vi[1] = 3u * t + 1u; vi[2] = 3u * t + 2u; }
    float* w = world + (size_t)(base + t) * 9;
    for (int c = 0; c < 3; ++c) {
        if (vi[c] >= in.num_vertices) { atomicMax(error, 1u); vi[c] = 0u; }
        const float* v = (const float*)((const char*)in.vertices + (size_t)vi[c] * in.vertex_stride);
        const float x = v[0], y = v[1], z = v[2];
        if (in.has_transform) {
            for (int r = 0; r < 3; ++r) w[3 * c + r] = ((in.m[4 * r] * x + in.m[4 * r + 1] * y) + in.m[4 * r + 2] * z) + in.m[4 * r + 3];
        } else { w[3 * c] = x; w[3 * c + 1] = y; w[3 * c + 2] = z; }
    }
    uint32_t m = in.sbt_index_offsets ? in.sbt_index_offsets[t] : 0u;
    uint32_t gf = 0u;
    if (in.flags) { if (m >= in.num_sbt_records) { atomicMax(error, 2u); m = 0u; } gf = in.flags[m]; }
    mats[base + t] = m; gflags[base + t] = gf;
}

// ---- node format 1: fp16 boxes -----------------------------------------------------------------------------------------------
// fp16 bit patterns of the largest representable value <= x (down) / the smallest >= x (up).  Subnormal halves are avoided
// (flushed outward to 0 / +-2^-14) so that no reader depends on the fp16 denormal mode; +-inf stays +-inf (empty boxes).
__device__ __forceinline__ uint32_t half_bits_dir(float x, bool up)
{
    const _Float16 h = (_Float16)x;                       // round to nearest even; overflow -> +-inf
    const float back = (float)h;
    unsigned short b = __builtin_bit_cast(unsigned short, h);
    if (up ? back < x : back > x) {                       // one step in the wanted direction
        const bool neg = (b & 0x8000u) != 0u;
        if ((b & 0x7fffu) == 0u) b = up ? 0x0001u : 0x8001u;          // +-0 -> the smallest subnormal of the wanted sign
        else if (neg == up) b = (unsigned short)(b - 1u);              // towards zero: magnitude down
        else b = (unsigned short)(b + 1u);                             // away from zero: magnitude up (max finite -> inf)
    }
    if ((b & 0x7c00u) == 0u && (b & 0x03ffu) != 0u) {                  // subnormal: flush outward
        const bool neg = (b & 0x8000u) != 0u;
        b = neg ? (up ? 0x8000u : 0x8400u) : (up ? 0x0400u : 0x0000u);
    }
    return b;
}
__device__ __forceinline__ float half_bits_to_float(uint32_t b) { return (float)__builtin_bit_cast(_Float16, (unsigned short)b); }

__global__ __launch_bounds__(256) void k_pack_nodes32(MiptBvhNode* nodes, uint32_t nn, uint4* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    MiptBvhNode nd = nodes[i];
    uint32_t w[2][3];
    for (int c = 0; c < 2; ++c) {
        float* x = c == 0 ? nd.c0x : nd.c1x;
        float* y = c == 0 ? nd.c0y : nd.c1y;
        float* z = c == 0 ? nd.c0z : nd.c1z;
        float* ax[3] = { x, y, z };
        for (int k = 0; k < 3; ++k) {
            const uint32_t lo = half_bits_dir(ax[k][0], false), hi = half_bits_dir(ax[k][1], true);
            ax[k][0] = half_bits_to_float(lo); ax[k][1] = half_bits_to_float(hi);
            w[c][k] = lo | (hi << 16);
        }
    }
    nodes[i] = nd;
    out[2u * i]      = make_uint4(w[0][0], w[0][1], w[0][2], (uint32_t)nd.c0);
    out[2u * i + 1u] = make_uint4(w[1][0], w[1][1], w[1][2], (uint32_t)nd.c1);
}


// ---- node format 2 on the device: the BVH4 mirror -------------------------------------------------------------------------------
// The collapse of the (fp16-rounded) BVH2 into four-child nodes that optixAccelBuild's counterpart here used to do on one host core
// (round 2: a D2H copy of every node, tens of ms of host time, an H2D copy -- on every build AND every refit).  Same rule as there:
// a node takes the two children of one of its inner children in place of that child, largest box first, while the worst-case
// per-lane stack of the walk stays within the budget (a node with n children holds n - 1 entries while its deepest subtree is
// walked; a BVH2 subtree of height h always fits h entries).  Level by level, top down: pass L turns the BVH2 nodes that became
// the BVH4 nodes of level L into 64-byte nodes and allots the indices of level L + 1 (one atomic per workgroup; siblings get
// consecutive indices, so the four children of a node share two 128-byte lines).  The boxes are read from the PACKED BVH2 mirror
// (they are the fp16 values already), so the mirror holds the same boxes bit for bit and nothing is converted.
// Heights (bottom-up, no thread waits: the last reporter of a node carries on to its parent) come first.
__global__ __launch_bounds__(256) void k_b4_parents(const uint4* n32, uint32_t nn, int* parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    const int c0 = (int)n32[2u * i].w, c1 = (int)n32[2u * i + 1u].w;
    if (c0 >= 0 && (uint32_t)c0 < nn) parent[c0] = (int)i;
    if (c1 >= 0 && (uint32_t)c1 < nn) parent[c1] = (int)i;
}

__global__ __launch_bounds__(256) void k_b4_heights(const uint4* n32, uint32_t nn, const int* parent, uint32_t* arrived, uint32_t* height)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    for (;;) {
        const int c0 = (int)n32[2u * i].w, c1 = (int)n32[2u * i + 1u].w;
        const bool in0 = c0 >= 0 && (uint32_t)c0 < nn, in1 = c1 >= 0 && (uint32_t)c1 < nn;
        const uint32_t need = 1u + (in0 ? 1u : 0u) + (in1 ? 1u : 0u);
        __threadfence();
        if (atomicAdd(&arrived[i], 1u) + 1u < need) return;
        __threadfence();
        const uint32_t h0 = in0 ? __hip_atomic_load(&height[c0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t h1 = in1 ? __hip_atomic_load(&height[c1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        __hip_atomic_store(&height[i], 1u + (h0 > h1 ? h0 : h1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int pr = parent[i];
        if (pr < 0) return;
        i = (uint32_t)pr;
    }
}

struct B4Work { uint32_t x, budget, used, pad; };      // BVH2 node x is BVH4 node <its position in the work array>

__device__ __forceinline__ float b4_area(const uint4 q)
{
    const float dx = half_bits_to_float(q.x >> 16) - half_bits_to_float(q.x & 0xffffu);
    const float dy = half_bits_to_float(q.y >> 16) - half_bits_to_float(q.y & 0xffffu);
    const float dz = half_bits_to_float(q.z >> 16) - half_bits_to_float(q.z & 0xffffu);
    return dx * dy + dy * dz + dz * dx;
}

// cnt[k]: BVH4 nodes of level k (cnt[0] = 1, the rest zero at the start); misc[0]: the most stack entries a lane can need
__global__ __launch_bounds__(256) void k_b4_level(const uint4* __restrict__ n32, uint32_t nn, const uint32_t* __restrict__ height, uint32_t level,
                                                  uint32_t* cnt, B4Work* work, uint32_t cap, uint4* out, uint32_t* misc)
{
    __shared__ uint32_t s_wave[4], s_base;
    uint32_t begin = 0u;
    for (uint32_t k = 0; k < level; ++k) begin += cnt[k];
    const uint32_t end = begin + cnt[level];
    for (uint32_t i0 = begin + blockIdx.x * blockDim.x; i0 < end; i0 += gridDim.x * blockDim.x) {      // (uniform per workgroup)
        const uint32_t me = i0 + threadIdx.x;
        uint4 q[4];
        int n = 0;
        uint32_t budget = 0u, used = 0u, n_inner = 0u;
        if (me < end) {
            const B4Work w = work[me];
            budget = w.budget; used = w.used;
            q[0] = n32[2u * w.x]; q[1] = n32[2u * w.x + 1u];
            q[2] = q[3] = make_uint4(0u, 0u, 0u, 0x7fffffffu);
            n = 2;
            auto h_of = [&](uint32_t ref) { return (int)ref >= 0 && ref < nn ? height[ref] : 0u; };
            for (;;) {
                int best = -1; float best_area = -1.0f;
                if (n < 4) for (int i = 0; i < n; ++i) if ((int)q[i].w >= 0) {
                    const float a = b4_area(q[i]);
                    if (a > best_area) { best_area = a; best = i; }
                }
                if (best < 0) break;
                // expanding child c leaves n + 1 children: n entries are held while the deepest remaining subtree is walked
                auto fits = [&](int c) {
                    uint32_t hm = 0u;
                    for (int j = 0; j < n; ++j) if (j != c) { const uint32_t h = h_of(q[j].w); hm = h > hm ? h : hm; }
                    const uint32_t g0 = h_of(n32[2u * q[c].w].w), g1 = h_of(n32[2u * q[c].w + 1u].w);
                    hm = g0 > hm ? g0 : hm; hm = g1 > hm ? g1 : hm;
                    return (uint32_t)n + hm <= budget;
                };
                if (!fits(best)) {
                    int other = -1;
                    for (int i = 0; i < n && other < 0; ++i) if (i != best && (int)q[i].w >= 0 && fits(i)) other = i;
                    if (other < 0) break;
                    best = other;
                }
                const uint32_t c = q[best].w;
                q[best] = n32[2u * c]; q[n] = n32[2u * c + 1u];
                ++n;
            }
            for (int i = 0; i < n; ++i) n_inner += (int)q[i].w >= 0 ? 1u : 0u;
            atomicMax(&misc[0], used + (uint32_t)(n - 1));
        }
        // indices of this workgroup's children in the next level: wave scan, workgroup scan, one atomic
        uint32_t incl = n_inner;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64); if ((int)(threadIdx.x & 63u) >= off) incl += v; }
        if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = incl;
        __syncthreads();
        if (threadIdx.x == 0u) {
            const uint32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            s_base = tot ? atomicAdd(&cnt[level + 1u], tot) : 0u;
        }
        __syncthreads();
        uint32_t at = end + s_base + (incl - n_inner);
        for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv) at += s_wave[wv];
        __syncthreads();                                                     // s_wave / s_base are reused by the next round
        if (me < end) {
            const uint32_t nb = budget - (uint32_t)(n - 1), nu = used + (uint32_t)(n - 1);
            for (int i = 0; i < 4; ++i) {
                uint4 o = q[i];
                if (i >= n) o = make_uint4(0u, 0u, 0u, 0x7fffffffu);
                else if ((int)o.w >= 0) {
                    if (at < cap) work[at] = B4Work{ o.w, nb, nu, 0u };
                    else atomicMax(&misc[1], 1u);                            // cannot happen (a BVH4 has fewer nodes than its BVH2): reported
                    o.w = at++;
                }
                if (me < cap) out[4u * me + (uint32_t)i] = o;
            }
        }
    }
}

}  // namespace

hipError_t pack_nodes32(hipStream_t st, float4* d_nodes, uint32_t nn, void* d_nodes32, bool* ok)
{
    *ok = false;
    if (nn == 0u) return hipSuccess;
    // the root's two child boxes bound the scene
    MiptBvhNode root;
    hipError_t e = hipMemcpyAsync(&root, d_nodes, sizeof root, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    float lo[3], hi[3], mag = 0.0f, ext = 0.0f;
    lo[0] = std::fmin(root.c0x[0], root.c1x[0]); hi[0] = std::fmax(root.c0x[1], root.c1x[1]);
    lo[1] = std::fmin(root.c0y[0], root.c1y[0]); hi[1] = std::fmax(root.c0y[1], root.c1y[1]);
    lo[2] = std::fmin(root.c0z[0], root.c1z[0]); hi[2] = std::fmax(root.c0z[1], root.c1z[1]);
    for (int k = 0; k < 3; ++k) {
        if (!(lo[k] <= hi[k])) return hipSuccess;                                  // empty or NaN scene box
        mag = std::fmax(mag, std::fmax(std::fabs(lo[k]), std::fabs(hi[k])));
        ext = std::fmax(ext, hi[k] - lo[k]);
    }
    // an fp16 ulp is at most |x| / 1024: the largest extent must span >= 256 of the coarsest steps, and nothing may overflow
    if (!(mag <= 32768.0f) || !(ext >= 0.25f * mag)) return hipSuccess;
    hipLaunchKernelGGL(k_pack_nodes32, dim3((nn + 255u) / 256u), dim3(256), 0, st, reinterpret_cast<MiptBvhNode*>(d_nodes), nn,
                       reinterpret_cast<uint4*>(d_nodes32));
    e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) *ok = true;
    return e;
}

// The BVH4 mirror of the packed BVH2 at d_nodes32, written to d_nodes4 (room for nn nodes of 64 B).  *ok = false (and nothing to
// use) when the tree is higher than 32 or the mirror would not fit; *n4 nodes, *stack_need entries per lane at most.
hipError_t collapse_bvh4(hipStream_t st, const void* d_nodes32, uint32_t nn, uint32_t depth, void* d_nodes4, uint32_t* n4, uint32_t* stack_need, bool* ok)
{
    *ok = false; *n4 = 0u; *stack_need = 0u;
    if (nn == 0u || depth == 0u || depth > 32u) return hipSuccess;
    char* tmp = nullptr;
    const size_t o_parent = 0, o_arr = o_parent + (((size_t)nn * 4u + 255) & ~(size_t)255), o_h = o_arr + (((size_t)nn * 4u + 255) & ~(size_t)255),
                 o_work = o_h + (((size_t)nn * 4u + 255) & ~(size_t)255), o_cnt = o_work + (((size_t)nn * sizeof(B4Work) + 255) & ~(size_t)255), total = o_cnt + 512;
    hipError_t e = hipMalloc((void**)&tmp, total);
    if (e != hipSuccess) return e;
    int* parent = (int*)(tmp + o_parent); uint32_t* arrived = (uint32_t*)(tmp + o_arr); uint32_t* height = (uint32_t*)(tmp + o_h);
    B4Work* work = (B4Work*)(tmp + o_work); uint32_t* cnt = (uint32_t*)(tmp + o_cnt); uint32_t* misc = cnt + 64;
    auto fail = [&](hipError_t err) { (void)hipFree(tmp); return err; };
    if ((e = hipMemsetAsync(parent, 0xff, (size_t)nn * 4u, st)) != hipSuccess) return fail(e);
    if ((e = hipMemsetAsync(arrived, 0, (size_t)nn * 4u, st)) != hipSuccess) return fail(e);
    if ((e = hipMemsetAsync(cnt, 0, 512, st)) != hipSuccess) return fail(e);
    const uint32_t one = 1u;
    const B4Work root{ 0u, 32u, 0u, 0u };
    if ((e = hipMemcpyAsync(cnt, &one, 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(work, &root, sizeof root, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    const uint4* n32 = reinterpret_cast<const uint4*>(d_nodes32);
    const uint32_t blocks = (nn + 255u) / 256u;
    hipLaunchKernelGGL(k_b4_parents, dim3(blocks), dim3(256), 0, st, n32, nn, parent);
    hipLaunchKernelGGL(k_b4_heights, dim3(blocks), dim3(256), 0, st, n32, nn, parent, arrived, height);
    // a BVH4 level is at most a BVH2 level deep, so `depth` passes reach every node; a pass whose level is empty returns at once
    const uint32_t grid = blocks < 2048u ? blocks : 2048u;
    for (uint32_t level = 0; level < depth; ++level)
        hipLaunchKernelGGL(k_b4_level, dim3(grid), dim3(256), 0, st, n32, nn, height, level, cnt, work, nn, reinterpret_cast<uint4*>(d_nodes4), misc);
    uint32_t h_cnt[66];
    if ((e = hipMemcpyAsync(h_cnt, cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    (void)hipFree(tmp);
    uint64_t tot = 0;
    for (uint32_t k = 0; k <= depth && k < 64u; ++k) tot += h_cnt[k];
    if (h_cnt[65] != 0u || tot > nn || h_cnt[depth < 64u ? depth : 63u] != 0u || h_cnt[64] > 32u) return hipSuccess;      // (none of these can happen)
    *n4 = (uint32_t)tot; *stack_need = h_cnt[64]; *ok = true;
    return hipSuccess;
}

hipError_t lbvh_refit(hipStream_t st, const float* d_world, uint32_t n, float4* d_nodes, uint32_t nn, float4* d_tris)
{
    int* parent = nullptr; uint32_t* arrived = nullptr;
    hipError_t e = hipMalloc((void**)&parent, (size_t)nn * 4u);
    if (e == hipSuccess) e = hipMalloc((void**)&arrived, (size_t)nn * 4u);
    if (e == hipSuccess) e = hipMemsetAsync(parent, 0xff, (size_t)nn * 4u, st);
    if (e == hipSuccess) e = hipMemsetAsync(arrived, 0, (size_t)nn * 4u, st);
    if (e == hipSuccess) {
        MiptBvhNode* nodes = reinterpret_cast<MiptBvhNode*>(d_nodes);
        MiptBvhTri* tris = reinterpret_cast<MiptBvhTri*>(d_tris);
        hipLaunchKernelGGL(k_refit_tris, dim3((n + 255u) / 256u), dim3(256), 0, st, d_world, n, tris);
        hipLaunchKernelGGL(k_refit_parents, dim3((nn + 255u) / 256u), dim3(256), 0, st, nodes, nn, parent);
        hipLaunchKernelGGL(k_refit_nodes, dim3((nn + 255u) / 256u), dim3(256), 0, st, nodes, nn, tris, n, d_world, parent, arrived);
        e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (parent) (void)hipFree(parent);
    if (arrived) (void)hipFree(arrived);
    return e;
}

void launch_gather_input(hipStream_t st, const GatherInput& in, float* world, uint32_t* mats, uint32_t* gflags, uint32_t base, uint32_t* error)
{
    if (in.num_triangles == 0u) return;
    hipLaunchKernelGGL(k_gather_input, dim3((in.num_triangles + 255u) / 256u), dim3(256), 0, st, in, world, mats, gflags, base, error);
}

// d_world: 9 floats per triangle (device); d_gflags: per triangle or null.  On success *d_blob is ONE allocation
// nodes | triangles | 64 B pad (as mipt_accel_build makes for the host builder).  Returns hipSuccess and ok = false
// when the tree is deeper than depth_cap (the caller falls back to the host builder).
hipError_t lbvh_build(hipStream_t st, const float* d_world, const uint32_t* d_gflags, uint32_t n, int leaf_max, int depth_cap,
                      float4** d_blob, uint32_t* num_nodes, uint32_t* depth, bool* ok)
{
    *ok = false; *d_blob = nullptr;
    const int n_inner = (int)n - 1;
    const uint32_t blocks = (n + 255u) / 256u;
    char* tmp = nullptr;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_boxes = carve(sizeof(Box6) * n), o_nboxes = carve(sizeof(Box6) * n), o_keys = carve(8ull * n), o_keys2 = carve(8ull * n),
                 o_vals = carve(4ull * n), o_vals2 = carve(4ull * n), o_left = carve(4ull * n), o_right = carve(4ull * n), o_pin = carve(4ull * n),
                 o_plf = carve(4ull * n), o_lo = carve(4ull * n), o_hi = carve(4ull * n), o_arr = carve(4ull * n), o_emit = carve(4ull * n),
                 o_idx = carve(4ull * n), o_misc = carve(256);
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                      (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 63, st);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, n_inner, st);
    if (e != hipSuccess) return e;
    const size_t o_cub = carve(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
    e = hipMalloc((void**)&tmp, off);
    if (e != hipSuccess) return e;
    Box6* boxes = (Box6*)(tmp + o_boxes); Box6* nboxes = (Box6*)(tmp + o_nboxes);
    unsigned long long* keys = (unsigned long long*)(tmp + o_keys); unsigned long long* keys2 = (unsigned long long*)(tmp + o_keys2);
    uint32_t* vals = (uint32_t*)(tmp + o_vals); uint32_t* vals2 = (uint32_t*)(tmp + o_vals2);
    int* left = (int*)(tmp + o_left); int* right = (int*)(tmp + o_right); int* pin = (int*)(tmp + o_pin); int* plf = (int*)(tmp + o_plf);
    int* rlo = (int*)(tmp + o_lo); int* rhi = (int*)(tmp + o_hi);
    uint32_t* arrived = (uint32_t*)(tmp + o_arr); uint32_t* emit = (uint32_t*)(tmp + o_emit); uint32_t* idx = (uint32_t*)(tmp + o_idx);
    uint32_t* misc = (uint32_t*)(tmp + o_misc);          // [0..5] scene bounds (ordered), [8] depth
    const uint32_t init[9] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u, 0u };
    auto fail = [&](hipError_t err) { (void)hipFree(tmp); return err; };
    if ((e = hipMemcpyAsync(misc, init, sizeof init, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    if ((e = hipMemsetAsync(arrived, 0, 4ull * n, st)) != hipSuccess) return fail(e);
    hipLaunchKernelGGL(k_lbvh_prims, dim3(blocks), dim3(256), 0, st, d_world, n, boxes, misc);
    hipLaunchKernelGGL(k_lbvh_morton, dim3(blocks), dim3(256), 0, st, boxes, n, misc, keys, vals);
    if ((e = hipcub::DeviceRadixSort::SortPairs(tmp + o_cub, sort_bytes, keys, keys2, vals, vals2, (int)n, 0, 63, st)) != hipSuccess) return fail(e);
    hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(blocks), dim3(256), 0, st, keys2, (int)n, left, right, pin, plf, rlo, rhi);
    hipLaunchKernelGGL(k_lbvh_fit, dim3(blocks), dim3(256), 0, st, boxes, vals2, (int)n, left, right, pin, plf, nboxes, arrived);
    hipLaunchKernelGGL(k_lbvh_flags, dim3(blocks), dim3(256), 0, st, rlo, rhi, n_inner, leaf_max, emit);
    if ((e = hipcub::DeviceScan::ExclusiveSum(tmp + o_cub, scan_bytes, emit, idx, n_inner, st)) != hipSuccess) return fail(e);
    hipLaunchKernelGGL(k_lbvh_depth, dim3(blocks), dim3(256), 0, st, pin, plf, emit, (int)n, misc + 8);
    uint32_t last[2] = { 0, 0 }, dep = 0;
    if ((e = hipMemcpyAsync(&last[0], idx + (n_inner - 1), 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(&last[1], emit + (n_inner - 1), 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(&dep, misc + 8, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e);
    const uint32_t nn = last[0] + last[1];
    *num_nodes = nn; *depth = dep;
    if ((int)dep > depth_cap) { (void)hipFree(tmp); return hipSuccess; }                 // ok stays false
    const size_t nb = (size_t)nn * sizeof(MiptBvhNode), tb = (size_t)n * sizeof(MiptBvhTri);
    float4* blob = nullptr;
    if ((e = hipMalloc((void**)&blob, nb + tb + 64 + 256 + (size_t)nn * 96u)) != hipSuccess) return fail(e);     // + room for the packed nodes
    hipLaunchKernelGGL(k_lbvh_emit, dim3(blocks), dim3(256), 0, st, left, right, rlo, rhi, n_inner, leaf_max, emit, idx, boxes, vals2, nboxes, (MiptBvhNode*)blob);
    hipLaunchKernelGGL(k_lbvh_tris, dim3(blocks), dim3(256), 0, st, d_world, vals2, d_gflags, n, (MiptBvhTri*)((char*)blob + nb));
    if ((e = hipMemsetAsync((char*)blob + nb + tb, 0, 64, st)) != hipSuccess) { (void)hipFree(blob); return fail(e); }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) { (void)hipFree(blob); return fail(e); }
    (void)hipFree(tmp);
    *d_blob = blob; *ok = true;
    return hipSuccess;
}

}  // namespace mipt
