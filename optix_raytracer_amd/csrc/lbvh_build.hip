// lbvh_build.hip -- device-side BVH2 build (the counterpart of OPTIX_BUILD_FLAG_PREFER_FAST_BUILD).
// Linear BVH: 63-bit Morton codes of the triangle centroids, radix sort (rocPRIM through hipCUB), the binary radix
// tree of Karras 2012 built fully in parallel, bottom-up box fitting, then subtrees of <= leaf_max triangles are
// folded into leaves and the surviving inner nodes are renumbered (scan) and written in the 64-byte two-child layout
// the traversal kernels read.  Same node / triangle layout as the host's binned-SAH builder (bvh_build.cpp), lower
// tree quality, a few milliseconds instead of ~0.2 s per million triangles.  Results of a render do not depend on
// which builder made the tree (closest hit: min t, tie -> lowest primitive).
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"
#include "detmath.h"

namespace mipt {

namespace {

__device__ __forceinline__ uint32_t f_ord(float f) { const uint32_t b = f2u(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float    f_unord(uint32_t k) { return u2f(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

struct Box6 { float lo[3], hi[3]; };

__global__ __launch_bounds__(256) void k_lbvh_prims(const float* __restrict__ world, uint32_t n, Box6* boxes, uint32_t* scene /* 6 ordered words */)
{
    // centroid bounds are enough for the Morton grid (and the root's bins).  Atomics on ONE address from all over the chip take ~70 ns
    // each: a million of them on six words were 1.07 ms of a 4 ms build, one per wave still 1.0 ms -- so the grid is at most 1024
    // workgroups, each reduces its triangles in LDS and adds six atomics (round 4)
    __shared__ uint32_t s_b[6];
    if (threadIdx.x < 6u) s_b[threadIdx.x] = threadIdx.x < 3u ? 0xffffffffu : 0u;
    __syncthreads();
    uint32_t lo[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, hi[3] = { 0u, 0u, 0u };
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* v = world + (size_t)i * 9;
        Box6 b;
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = fminf(v[k], fminf(v[3 + k], v[6 + k]));
            b.hi[k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
        }
        boxes[i] = b;
        for (int k = 0; k < 3; ++k) { const uint32_t c = f_ord(0.5f * (b.lo[k] + b.hi[k])); lo[k] = c < lo[k] ? c : lo[k]; hi[k] = c > hi[k] ? c : hi[k]; }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; ++k) {
            const uint32_t a = (uint32_t)__shfl_xor((int)lo[k], off, 64), c = (uint32_t)__shfl_xor((int)hi[k], off, 64);
            lo[k] = a < lo[k] ? a : lo[k]; hi[k] = c > hi[k] ? c : hi[k];
        }
    if ((threadIdx.x & 63u) == 0u) for (int k = 0; k < 3; ++k) { atomicMin(&s_b[k], lo[k]); atomicMax(&s_b[3 + k], hi[k]); }
    __syncthreads();
    if (threadIdx.x < 3u) atomicMin(&scene[threadIdx.x], s_b[threadIdx.x]);
    else if (threadIdx.x < 6u) atomicMax(&scene[threadIdx.x], s_b[threadIdx.x]);
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
    // CUBIC grid cells (round 3): every axis is quantised by the LARGEST extent of the centroid bounds.  With each axis stretched to its
    // own extent (round 2) a flat scene -- a height field, a hall's floor -- gets cells that are slabs, the high bits of its thin axis
    // cut the sheet by height inside every xz-cell, and the two halves of such a cut overlap completely (imgui_test's height field:
    // frames 58 % slower than over the host SAH tree).
    float ext_max = 0.0f;
    for (int k = 0; k < 3; ++k) ext_max = fmaxf(ext_max, f_unord(scene[3 + k]) - f_unord(scene[k]));
    for (int k = 0; k < 3; ++k) {
        const float lo = f_unord(scene[k]);
        const float c = 0.5f * (boxes[i].lo[k] + boxes[i].hi[k]);
        const float ext = ext_max;
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

// ---- SAH treelets (round 3): the bottom of the radix tree rebuilt by surface-area heuristic ----------------------------------------------
// The radix tree splits space at Morton-code bits: near the root of a scene of evenly spread triangles that is what a SAH builder
// does too, but the last levels -- which triangles share a leaf, which leaves share a node -- are where it loses (LBVH frames 7 %
// behind the host SAH tree's).  Every subtree of at most 64 triangles whose parent holds more is a TREELET: one wave rebuilds it top
// down with a full-sweep SAH (one triangle per lane; per level and axis: rank the segment's centroids, scan the sorted boxes from both
// ends, cost = area(left) * count + area(right) * count, best split per segment; all segments of a level at once).  A treelet over the
// sorted positions [lo, hi] owns hi - lo inner-node slots of Karras' numbering: lo .. hi - 1 when its root is node lo, lo + 1 .. hi
// when its root is node hi (a radix-tree node's index is one end of its range; the other end's index belongs to an ancestor).  The
// node that splits a segment after position m takes the m-th of them (each slot once), the triangles are permuted inside [lo, hi]
// only, so ranges stay contiguous and the leaf folding, emission, depth and height passes that follow see an ordinary tree.
constexpr int kTreelet = 64;

// order2 != null (the top-down front end, whose partition leaves the triangles of a range in the order the atomics fell): a node over
// two triangles gets them in the order of their primitive numbers, so that the tree does not depend on the run
__global__ __launch_bounds__(256) void k_treelet_roots(const int* range_lo, const int* range_hi, const int* parent_inner, int n_inner, uint32_t* roots, uint32_t* n_roots,
                                                       uint32_t* order2)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner) return;
    const int sz = range_hi[i] - range_lo[i] + 1;
    if (order2 != nullptr && sz == 2) {
        const uint32_t a = order2[range_lo[i]], b = order2[range_hi[i]];
        if (a > b) { order2[range_lo[i]] = b; order2[range_hi[i]] = a; }
    }
    const int p = parent_inner[i];
    if (sz > kTreelet || sz < 3 || p < 0) return;                     // (the root keeps slot 0; two triangles have one shape)
    if (range_hi[p] - range_lo[p] + 1 <= kTreelet) return;            // inside a treelet, not its root
    roots[atomicAdd(n_roots, 1u)] = (uint32_t)i;
}

__device__ __forceinline__ float box_area6(const float* b) { const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return dx * dy + dy * dz + dz * dx; }

__global__ __launch_bounds__(64) void k_lbvh_treelets(const uint32_t* __restrict__ roots, const uint32_t* __restrict__ n_roots, const Box6* __restrict__ prim_boxes,
                                                      uint32_t* sorted_prims, int* left, int* right, int* parent_inner, int* parent_leaf, int* range_lo, int* range_hi,
                                                      int by_prim)
{
    __shared__ float s_box[64][7];                                    // a record per position: box + prim (as bits)
    const int lane = (int)threadIdx.x;
    const uint32_t n_t = *n_roots;
    for (uint32_t t = blockIdx.x; t < n_t; t += gridDim.x) {
        const int root = (int)roots[t];
        const int lo = range_lo[root], hi = range_hi[root], n = hi - lo + 1;
        const int slot0 = root == lo ? lo : lo + 1;                      // the first of the treelet's own slots
        const int P = parent_inner[root];
        const bool valid = lane < n;
        uint32_t prim = valid ? sorted_prims[lo + lane] : 0u;
        if (by_prim) {
            // the top-down front end leaves a range's triangles in the order its atomics fell: start from the order of the primitive
            // numbers, so that ties between equal costs fall the same way in every run (the tree is then a function of the input)
            int rank = 0;
            for (int k = 0; k < kTreelet; ++k) { const uint32_t pk = (uint32_t)__shfl((int)prim, k, 64); rank += (k < n && pk < prim) ? 1 : 0; }
            if (valid) s_box[rank][6] = __uint_as_float(prim);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (valid) prim = __float_as_uint(s_box[lane][6]);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        float b[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        if (valid) { const Box6 pb = prim_boxes[prim]; for (int k = 0; k < 3; ++k) { b[k] = pb.lo[k]; b[3 + k] = pb.hi[k]; } }
        int sa = 0, sb = n - 1, sp = P, ss = left[P] == root ? 0 : 1;  // my segment (positions = lanes), the node above it and the side it hangs on
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int level = 0; level < kTreelet; ++level) {
            const bool active = valid && sb > sa;
            if (__ballot(active) == 0ull) break;
            float best_cost = __builtin_inff(); int best_m = sa, best_axis = 0;
            int pos_axis[3] = { lane, lane, lane };
            for (int axis = 0; axis < 3; ++axis) {
                const float key = b[axis] + b[3 + axis];
                int rank = 0;
                for (int k = 0; k < kTreelet; ++k) {
                    const float kk = __shfl(key, k, 64);
                    rank += (k >= sa && k <= sb && (kk < key || (kk == key && k < lane))) ? 1 : 0;
                }
                const int pos = sa + rank;
                pos_axis[axis] = pos;
                // the segment's boxes in this axis' order: lane j gets the box of sorted position j
                if (valid) for (int c = 0; c < 6; ++c) s_box[pos][c] = b[c];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                float L[6], R[6];
                for (int c = 0; c < 6; ++c) { L[c] = valid ? s_box[lane][c] : 0.f; R[c] = L[c]; }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                for (int d = 1; d < kTreelet; d <<= 1) {              // union of positions sa .. j (L) and j .. sb (R)
                    for (int c = 0; c < 6; ++c) {
                        const float u = __shfl_up(L[c], d, 64), w = __shfl_down(R[c], d, 64);
                        if (lane - d >= sa) L[c] = c < 3 ? fminf(L[c], u) : fmaxf(L[c], u);
                        if (lane + d <= sb) R[c] = c < 3 ? fminf(R[c], w) : fmaxf(R[c], w);
                    }
                }
                float Rn[6];
                for (int c = 0; c < 6; ++c) Rn[c] = __shfl_down(R[c], 1, 64);      // union of positions j + 1 .. sb
                float cost = __builtin_inff();
                if (active && lane < sb) cost = box_area6(L) * (float)(lane - sa + 1) + box_area6(Rn) * (float)(sb - lane);
                // the cheapest split of my segment along this axis (ties: the lowest position)
                float seg_cost = __builtin_inff(); int seg_m = sa;
                for (int k = 0; k < kTreelet; ++k) {
                    const float ck = __shfl(cost, k, 64);
                    if (k >= sa && k < sb && ck < seg_cost) { seg_cost = ck; seg_m = k; }
                }
                if (seg_cost < best_cost) { best_cost = seg_cost; best_m = seg_m; best_axis = axis; }
            }
            if (active && !(best_cost < __builtin_inff())) { best_m = (sa + sb) >> 1; best_axis = 0; }       // (NaN boxes: any split)
            // move every record to its place in its segment's best order (an inactive lane keeps its position)
            const int to = active ? (best_axis == 0 ? pos_axis[0] : (best_axis == 1 ? pos_axis[1] : pos_axis[2])) : lane;
            if (valid) { for (int c = 0; c < 6; ++c) s_box[to][c] = b[c]; s_box[to][6] = __uint_as_float(prim); }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (valid) { for (int c = 0; c < 6; ++c) b[c] = s_box[lane][c]; prim = __float_as_uint(s_box[lane][6]); }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (active) {
                const int N = slot0 + best_m;                           // the new node's slot
                if (lane == best_m) {
                    range_lo[N] = lo + sa; range_hi[N] = lo + sb; parent_inner[N] = sp;
                    if (ss == 0) left[sp] = N; else right[sp] = N;
                }
                if (lane <= best_m) { sb = best_m; ss = 0; } else { sa = best_m + 1; ss = 1; }
                sp = N;
                if (sa == sb) {                                         // a single triangle: the leaf link
                    if (ss == 0) left[sp] = ~(lo + sa); else right[sp] = ~(lo + sa);
                    parent_leaf[lo + sa] = sp;
                }
            }
        }
        if (valid) sorted_prims[lo + lane] = prim;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// ---- SAH clusters (round 3, experiment MIPT_LBVH_CLUSTERS): the treelet pass one level up -----------------------------------------------
// The ITEMS of this pass are the subtrees the treelet pass has rebuilt (and single triangles hanging off bigger nodes): children of
// nodes over more than kTreelet triangles that are not such nodes themselves.  A radix-tree node with at most 64 items below it whose
// parent has more is a CLUSTER: one wave collects its items (a depth-first walk over the cluster's own inner nodes, whose slots it then
// reuses), takes each item's box as the union of its triangles' boxes and its weight as its triangle count, and rebuilds the tree
// over the items top down with the same full-sweep SAH (cost = area(left) * triangles(left) + area(right) * triangles(right)).  Items
// keep their own subtrees and their place in the triangle order; above them ranges are no longer contiguous, which nothing after this
// pass needs (leaf folding looks at ranges of at most leaf_max triangles; everything else follows the child / parent links) -- a
// cluster node's range holds its triangle COUNT.  The cluster's root keeps its slot, so the link from above stays.
// `big`: one byte per radix-tree node, 1 = above the item layer of this pass (first pass: over more than kTreelet triangles; each further
// pass: the big nodes with more than 64 items below them -- what lies above the clusters the pass before has made)
__global__ __launch_bounds__(256) void k_cluster_first_big(const int* range_lo, const int* range_hi, int n_inner, uint8_t* big)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n_inner) big[i] = range_hi[i] - range_lo[i] + 1 > kTreelet ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_cluster_next_big(const uint32_t* cnt, int n_inner, uint8_t* big)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n_inner) big[i] = big[i] != 0u && cnt[i] > (uint32_t)kTreelet ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_cluster_counts(const int* left, const int* right, const uint8_t* big, const int* parent_inner, int n_inner, uint32_t* cnt)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner || big[i] == 0u) return;
    uint32_t items = 0u;
    const int ch[2] = { left[i], right[i] };
    for (int k = 0; k < 2; ++k) if (ch[k] < 0 || big[ch[k]] == 0u) ++items;
    if (items == 0u) return;
    int a = i;
    for (int guard = 0; a >= 0 && guard < 4096; ++guard) { atomicAdd(&cnt[a], items); a = parent_inner[a]; }
}

__global__ __launch_bounds__(256) void k_cluster_roots(const uint8_t* big, const int* parent_inner, const uint32_t* cnt, int n_inner, uint32_t* roots, uint32_t* n_roots)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner || big[i] == 0u) return;
    const uint32_t c = cnt[i];
    if (c < 3u || c > (uint32_t)kTreelet) return;
    const int p = parent_inner[i];
    if (p >= 0 && cnt[p] <= (uint32_t)kTreelet) return;              // inside a cluster, not its root
    roots[atomicAdd(n_roots, 1u)] = (uint32_t)i;
}

__global__ __launch_bounds__(64) void k_lbvh_clusters(const uint32_t* __restrict__ roots, const uint32_t* __restrict__ n_roots, const uint8_t* __restrict__ big,
                                                      const Box6* __restrict__ prim_boxes, const uint32_t* __restrict__ sorted_prims, const Box6* __restrict__ node_boxes,
                                                      int* left, int* right, int* parent_inner, int* parent_leaf, int* range_lo, int* range_hi)
{
    __shared__ float s_box[64][8];                                    // a record per position: box, item reference (bits), triangle count
    __shared__ int s_items[64], s_slots[64], s_stack[64], s_n[2];
    const int lane = (int)threadIdx.x;
    const uint32_t n_t = *n_roots;
    for (uint32_t t = blockIdx.x; t < n_t; t += gridDim.x) {
        const int R = (int)roots[t];
        __syncthreads();
        if (lane == 0) {
            int ni = 0, ns = 0, sp = 0;
            s_stack[sp++] = R;
            while (sp > 0 && ns < 64) {
                const int x = s_stack[--sp];
                s_slots[ns++] = x;
                const int ch[2] = { right[x], left[x] };
                for (int k = 0; k < 2; ++k) {
                    const int c = ch[k];
                    const bool item = c < 0 || big[c] == 0u;
                    if (item) { if (ni < 64) s_items[ni] = c; ++ni; } else if (sp < 64) s_stack[sp++] = c; else ni = 1 << 20;
                }
            }
            s_n[0] = sp == 0 ? ni : 1 << 20; s_n[1] = ns;
        }
        __syncthreads();
        const int n = s_n[0];
        if (n < 3 || n > 64 || s_n[1] != n - 1) continue;              // (never by construction: leave the cluster as it is)
        const bool valid = lane < n;
        int ref = valid ? s_items[lane] : 0;
        float b[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        float wt = 0.0f;
        if (valid) {
            const Box6 u = ref < 0 ? prim_boxes[sorted_prims[~ref]] : node_boxes[ref];      // (an item's subtree is not touched by this pass: its fitted box stands)
            for (int k = 0; k < 3; ++k) { b[k] = u.lo[k]; b[3 + k] = u.hi[k]; }
            wt = ref < 0 ? 1.0f : (float)(range_hi[ref] - range_lo[ref] + 1);
        }
        int sa = 0, sb = n - 1, sp = -1, ss = 0;                       // my segment; the node above it (-1: the cluster's root is made first) and the side
        int m_root = 0;
        __syncthreads();
        for (int level = 0; level < kTreelet; ++level) {
            const bool active = valid && sb > sa;
            if (__ballot(active) == 0ull) break;
            float best_cost = __builtin_inff(); int best_m = sa, best_axis = 0;
            int pos_axis[3] = { lane, lane, lane };
            for (int axis = 0; axis < 3; ++axis) {
                const float key = b[axis] + b[3 + axis];
                int rank = 0;
                for (int k = 0; k < kTreelet; ++k) {
                    const float kk = __shfl(key, k, 64);
                    rank += (k >= sa && k <= sb && (kk < key || (kk == key && k < lane))) ? 1 : 0;
                }
                const int pos = sa + rank;
                pos_axis[axis] = pos;
                if (valid) { for (int c = 0; c < 6; ++c) s_box[pos][c] = b[c]; s_box[pos][7] = wt; }
                __syncthreads();
                float L[6], Rr[6], wl = valid ? s_box[lane][7] : 0.0f, wr = wl;
                for (int c = 0; c < 6; ++c) { L[c] = valid ? s_box[lane][c] : 0.f; Rr[c] = L[c]; }
                __syncthreads();
                for (int d = 1; d < kTreelet; d <<= 1) {              // union / weight of positions sa .. j (L, wl) and j .. sb (R, wr)
                    for (int c = 0; c < 6; ++c) {
                        const float u = __shfl_up(L[c], d, 64), w = __shfl_down(Rr[c], d, 64);
                        if (lane - d >= sa) L[c] = c < 3 ? fminf(L[c], u) : fmaxf(L[c], u);
                        if (lane + d <= sb) Rr[c] = c < 3 ? fminf(Rr[c], w) : fmaxf(Rr[c], w);
                    }
                    const float ul = __shfl_up(wl, d, 64), ur = __shfl_down(wr, d, 64);
                    if (lane - d >= sa) wl += ul;
                    if (lane + d <= sb) wr += ur;
                }
                float Rn[6];
                for (int c = 0; c < 6; ++c) Rn[c] = __shfl_down(Rr[c], 1, 64);      // positions j + 1 .. sb
                const float wrn = __shfl_down(wr, 1, 64);
                float cost = __builtin_inff();
                if (active && lane < sb) cost = box_area6(L) * wl + box_area6(Rn) * wrn;
                float seg_cost = __builtin_inff(); int seg_m = sa;
                for (int k = 0; k < kTreelet; ++k) {
                    const float ck = __shfl(cost, k, 64);
                    if (k >= sa && k < sb && ck < seg_cost) { seg_cost = ck; seg_m = k; }
                }
                if (seg_cost < best_cost) { best_cost = seg_cost; best_m = seg_m; best_axis = axis; }
            }
            if (active && !(best_cost < __builtin_inff())) { best_m = (sa + sb) >> 1; best_axis = 0; }
            const int to = active ? (best_axis == 0 ? pos_axis[0] : (best_axis == 1 ? pos_axis[1] : pos_axis[2])) : lane;
            if (valid) { for (int c = 0; c < 6; ++c) s_box[to][c] = b[c]; s_box[to][6] = __int_as_float(ref); s_box[to][7] = wt; }
            __syncthreads();
            if (valid) { for (int c = 0; c < 6; ++c) b[c] = s_box[lane][c]; ref = __float_as_int(s_box[lane][6]); wt = s_box[lane][7]; }
            __syncthreads();
            float seg_w = 0.0f;                                          // the segment's triangle count (the new node's range words)
            for (int k = 0; k < kTreelet; ++k) { const float wk = __shfl(wt, k, 64); if (k >= sa && k <= sb && k < n) seg_w += wk; }
            if (level == 0) m_root = __shfl(best_m, 0, 64);            // one segment at level 0: every lane has the same split
            if (active) {
                const int mi = level == 0 ? 0 : (best_m == m_root ? 0 : (best_m == 0 ? m_root : best_m));   // slot of the split after position best_m (the root's is slot 0)
                const int N = s_slots[mi];
                if (lane == best_m && sp >= 0) {
                    // (the cluster's root keeps its range -- still the contiguous run of its triangles -- and its place under its parent; a node
                    //  inside the cluster covers no contiguous run any more: its range holds a COUNT, never small enough to be folded into a leaf)
                    range_lo[N] = 0; range_hi[N] = ((int)seg_w > kTreelet ? (int)seg_w : kTreelet + 1) - 1;
                    parent_inner[N] = sp;
                    if (ss == 0) left[sp] = N; else right[sp] = N;
                }
                if (lane <= best_m) { sb = best_m; ss = 0; } else { sa = best_m + 1; ss = 1; }
                sp = N;
                if (sa == sb) {                                         // a single item: its link
                    if (ss == 0) left[sp] = ref; else right[sp] = ref;
                    if (ref < 0) parent_leaf[~ref] = sp; else parent_inner[ref] = sp;
                }
            }
        }
    }
}

// ---- top-down binned SAH on the device (round 4, MIPT_OPT_BUILD = 2) -----------------------------------------------------------------
// The radix tree cuts space at Morton-code bits; a scene of large triangles among small ones (a hall's walls around its furniture), a
// height field or a wall is cut where no SAH builder would cut it, and the treelets below repair the last six levels only: frames
// 10-60 % behind the host tree's (round 3).  This front end replaces {Morton codes, sort, radix tree} by what the host builder does
// (bvh_build.cpp: binned SAH over the centroid bounds, 16 bins per axis here), level by level over ALL open ranges at once:
//   k_sah_bin        every triangle of an open range adds its box, its centroid and 1 to its bin on each axis -- in LDS for the range
//                    the workgroup's chunk starts in (ranges are contiguous, so that is nearly all of a chunk), flushed with one global
//                    atomic per non-empty word; stragglers of other ranges go to global memory directly
//   k_sah_split      one thread per open range: sweep the 3 x 16 bins, cheapest area(L) n(L) + area(R) n(R) whose sides still fit under
//                    the depth cap (the host's rule), else the median; the children's centroid bounds are the union of the bins'
//                    centroid bounds on their side (no pass of their own); the node takes Karras' numbering -- a node's index is one
//                    end of its range: the root 0, a left child its LAST position, a right child its FIRST -- so everything after this
//                    front end (treelets, box fitting, leaf folding, emission) sees an ordinary radix tree
//   k_sah_partition  every triangle moves to its side of its range (out of place; one atomic per wave, range and side)
// until no range holds more than 64 triangles; those become TREELETS (the pass above: full-sweep SAH, one wave each).  No sort at all.
#ifndef MIPT_SAH_BINS
#define MIPT_SAH_BINS 32          // the host builder's count (16: the headline tree one level deeper, the 4.2 k-triangle glTF 4 % behind the host tree)
#endif
constexpr int kSahBins = MIPT_SAH_BINS, kSahBinWords = 7, kSahSegWords = 3 * kSahBins * kSahBinWords;      // per bin: count, box lo / hi (ordered words)
constexpr uint32_t kSahNone = 0xffffffffu;
struct SahSeg { int lo, hi, node, depth; uint32_t cb[6]; };                                   // an open range, its node, its centroid bounds (ordered)
struct SahSplit { int lo, left_count, axis, bin, median; float cb_lo, scale; uint32_t child[2], small[2]; };   // child: slot of the side's range in the next level, small: in the
                                                                                                            // list of ranges one workgroup finishes (k_sah_small); else kSahNone
constexpr int kSahSmall = 2048;                                  // a range of at most this many triangles leaves the level-by-level loop

__device__ __forceinline__ int sah_bin_of(float c, float cb_lo, float scale)
{
    const float t = (c - cb_lo) * scale;
    int b = t >= 0.0f ? (int)fminf(t, (float)(kSahBins - 1)) : 0;          // (a NaN centroid lands in bin 0)
    return b < kSahBins ? b : kSahBins - 1;
}
__device__ __forceinline__ uint32_t sah_word_identity(int w) { const int k = w % kSahBinWords; return k >= 1 && k <= 3 ? 0xffffffffu : 0u; }

__global__ __launch_bounds__(256) void k_sah_init(uint32_t* bins, size_t words, uint32_t* perm, uint32_t* seg_of, uint32_t n, int* left, int* right, int* rlo, int* rhi)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) bins[i] = sah_word_identity((int)(i % kSahSegWords));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        perm[i] = (uint32_t)i; seg_of[i] = 0u;
        left[i] = 0x7ffffff0; right[i] = 0x7ffffff0; rlo[i] = 0; rhi[i] = -1;               // an unused slot: no range, not a bottom node
    }
}

constexpr int kSahChunk = 2048, kSahTables = 8;
__global__ __launch_bounds__(256) void k_sah_bin(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ seg_of, const SahSeg* __restrict__ segs, uint32_t n,
                                                 const Box6* __restrict__ boxes, uint32_t* bins)
{
    // LDS bins for up to eight of the open ranges a chunk of 2048 positions touches (near the root that is one range; deep down ranges
    // are shorter than a chunk): a range claims the table its slot number hashes to, a range that finds its table taken adds to global
    // memory directly.  One global atomic per non-empty word and table at the end.
    __shared__ uint32_t s_dir[kSahTables];
    __shared__ uint32_t s_bins[kSahTables][kSahSegWords];
    for (uint32_t chunk = blockIdx.x; (size_t)chunk * kSahChunk < n; chunk += gridDim.x) {
        const uint32_t first = chunk * (uint32_t)kSahChunk;
        if (threadIdx.x < (uint32_t)kSahTables) s_dir[threadIdx.x] = kSahNone;
        for (int w = (int)threadIdx.x; w < kSahTables * kSahSegWords; w += 256) (&s_bins[0][0])[w] = sah_word_identity(w % kSahSegWords);
        __syncthreads();
        for (uint32_t p = first + threadIdx.x; p < first + (uint32_t)kSahChunk && p < n; p += 256u) {
            const uint32_t s = seg_of[p];
            if (s == kSahNone) continue;
            const SahSeg sg = segs[s];
            const Box6 b = boxes[perm[p]];
            const uint32_t h = s & (uint32_t)(kSahTables - 1);
            uint32_t owner = s_dir[h];
            if (owner == kSahNone) { owner = atomicCAS(&s_dir[h], kSahNone, s); if (owner == kSahNone) owner = s; }
            uint32_t* dst = owner == s ? s_bins[h] : bins + (size_t)s * kSahSegWords;
            uint32_t ob[6];
            for (int k = 0; k < 3; ++k) { ob[k] = f_ord(b.lo[k]); ob[3 + k] = f_ord(b.hi[k]); }
            for (int k = 0; k < 3; ++k) {
                const float cl = f_unord(sg.cb[k]), ext = f_unord(sg.cb[3 + k]) - cl;
                if (!(ext > 0.0f) || !(ext < 3.0e38f)) continue;              // an axis without extent is not a split axis (bvh_build.cpp: usable[])
                const int bi = sah_bin_of(0.5f * (b.lo[k] + b.hi[k]), cl, (float)kSahBins / ext);
                uint32_t* w = dst + (k * kSahBins + bi) * kSahBinWords;
                atomicAdd(&w[0], 1u);
                for (int c = 0; c < 3; ++c) { atomicMin(&w[1 + c], ob[c]); atomicMax(&w[4 + c], ob[3 + c]); }
            }
        }
        __syncthreads();
        for (int t = 0; t < kSahTables; ++t) {
            const uint32_t sid = s_dir[t];
            if (sid == kSahNone) continue;                                     // (uniform)
            uint32_t* g = bins + (size_t)sid * kSahSegWords;
            for (int w = (int)threadIdx.x; w < kSahSegWords; w += 256) {
                const uint32_t v = s_bins[t][w];
                if (v == sah_word_identity(w)) continue;
                const int k = w % kSahBinWords;
                if (k == 0) atomicAdd(&g[w], v);
                else if (k <= 3) atomicMin(&g[w], v);
                else atomicMax(&g[w], v);
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float sah_area(const uint32_t* lo, const uint32_t* hi)
{
    const float dx = f_unord(hi[0]) - f_unord(lo[0]), dy = f_unord(hi[1]) - f_unord(lo[1]), dz = f_unord(hi[2]) - f_unord(lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

// The sweep of one axis' bins by one wave (lane = bin): inclusive unions of bins 0 .. lane and lane .. last by shuffles, the cost of the
// split after every bin, the cheapest by a wave minimum (ties to the lower bin).  `bins`: the range's table (global memory or LDS).
// Returns (in every lane) the cost -- inf when the axis has no extent or no admissible split --, the bin and the triangles on its left.
__device__ __forceinline__ void sah_sweep_axis(const uint32_t* bins, int k, const SahSeg& sg, int lane, int leaf_max, int depth_cap, float& cost_out, int& bin_out, int& nl_out)
{
    const float cl = f_unord(sg.cb[k]), ext = f_unord(sg.cb[3 + k]) - cl;
    const bool usable = ext > 0.0f && ext < 3.0e38f;
    const bool has = lane < kSahBins;
    const uint32_t* w = bins + (k * kSahBins + (has ? lane : 0)) * kSahBinWords;
    uint32_t v[kSahBinWords];
    for (int c = 0; c < kSahBinWords; ++c) v[c] = w[c];                  // all seven words at once: a load that waits for the count costs a round trip per axis
    uint32_t pc = has ? v[0] : 0u, plo[3], phi[3];
    for (int c = 0; c < 3; ++c) { plo[c] = has && pc ? v[1 + c] : 0xffffffffu; phi[c] = has && pc ? v[4 + c] : 0u; }
    uint32_t sc = pc, slo[3] = { plo[0], plo[1], plo[2] }, shi[3] = { phi[0], phi[1], phi[2] };
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t c_up = (uint32_t)__shfl_up((int)pc, off, 64), c_dn = (uint32_t)__shfl_down((int)sc, off, 64);
        uint32_t lu[3], hu[3], ld[3], hd[3];
        for (int c = 0; c < 3; ++c) {
            lu[c] = (uint32_t)__shfl_up((int)plo[c], off, 64); hu[c] = (uint32_t)__shfl_up((int)phi[c], off, 64);
            ld[c] = (uint32_t)__shfl_down((int)slo[c], off, 64); hd[c] = (uint32_t)__shfl_down((int)shi[c], off, 64);
        }
        if (lane >= off) { pc += c_up; for (int c = 0; c < 3; ++c) { plo[c] = lu[c] < plo[c] ? lu[c] : plo[c]; phi[c] = hu[c] > phi[c] ? hu[c] : phi[c]; } }
        if (lane + off < 64) { sc += c_dn; for (int c = 0; c < 3; ++c) { slo[c] = ld[c] < slo[c] ? ld[c] : slo[c]; shi[c] = hd[c] > shi[c] ? hd[c] : shi[c]; } }
    }
    // the split after bin `lane`: left = bins 0 .. lane, right = bins lane + 1 .. last
    const uint32_t nr = (uint32_t)__shfl_down((int)sc, 1, 64);
    uint32_t rl[3], rh[3];
    for (int c = 0; c < 3; ++c) { rl[c] = (uint32_t)__shfl_down((int)slo[c], 1, 64); rh[c] = (uint32_t)__shfl_down((int)shi[c], 1, 64); }
    // sides must still fit below the depth cap when split evenly from here on (bvh_build.cpp: capacity()); four levels are left to
    // the treelets, which split by SAH alone
    int levels = depth_cap - sg.depth - 1 - 4;
    levels = levels < 0 ? 0 : (levels > 40 ? 40 : levels);
    const unsigned long long side_cap = (unsigned long long)leaf_max << levels;
    float cost = __builtin_inff();
    if (usable && lane < kSahBins - 1 && pc != 0u && nr != 0u && pc <= side_cap && nr <= side_cap)
        cost = sah_area(plo, phi) * (float)pc + sah_area(rl, rh) * (float)nr;
    if (!(cost < __builtin_inff())) cost = __builtin_inff();          // (a NaN area is no candidate)
    float bc = cost; int bb = lane, bn = (int)pc;
    for (int off = 32; off > 0; off >>= 1) {
        const float oc = __shfl_xor(bc, off, 64); const int ob = __shfl_xor(bb, off, 64), on = __shfl_xor(bn, off, 64);
        if (oc < bc || (oc == bc && ob < bb)) { bc = oc; bb = ob; bn = on; }
    }
    cost_out = bc; bin_out = bb; nl_out = bn;
}

__global__ __launch_bounds__(64) void k_sah_split(const SahSeg* __restrict__ cur, const uint32_t* __restrict__ n_cur_p, uint32_t* bins, SahSplit* splits, uint32_t* fill,
                                                  SahSeg* next, uint32_t* n_next, SahSeg* smalls, uint32_t* n_small, int leaf_max, int depth_cap,
                                                  int* left, int* right, int* pin, int* plf, int* rlo, int* rhi)
{
    const uint32_t s = blockIdx.x;                                             // one wave per open range (the sweep: lane = bin)
    if (s >= *n_cur_p) return;
    const int lane = (int)threadIdx.x;
    const SahSeg sg = cur[s];
    uint32_t* B = bins + (size_t)s * kSahSegWords;
    const int cnt = sg.hi - sg.lo + 1;
    float best = __builtin_inff(); int best_axis = -1, best_bin = -1, best_nl = 0;
    for (int k = 0; k < 3; ++k) {
        float c; int b, nl;
        sah_sweep_axis(B, k, sg, lane, leaf_max, depth_cap, c, b, nl);
        if (c < best) { best = c; best_axis = k; best_bin = b; best_nl = nl; }
    }
    for (int w = lane; w < kSahSegWords; w += 64) B[w] = sah_word_identity(w);           // the slot's bins, ready for the next level
    if (lane != 0) return;
    SahSplit sp;
    sp.lo = sg.lo; sp.median = best_axis < 0 ? 1 : 0;
    sp.left_count = sp.median ? cnt / 2 : best_nl;
    sp.axis = sp.median ? 0 : best_axis; sp.bin = best_bin;
    sp.cb_lo = 0.0f; sp.scale = 0.0f;
    // (the children's centroid bounds: empty here, filled by the partition pass as the triangles take their sides; after a median
    //  split -- no usable axis -- the parent's)
    uint32_t cbs[2][6];
    for (int side = 0; side < 2; ++side) for (int a = 0; a < 6; ++a) cbs[side][a] = sp.median ? sg.cb[a] : (a < 3 ? 0xffffffffu : 0u);
    if (!sp.median) {
        const float cl = f_unord(sg.cb[best_axis]), ext = f_unord(sg.cb[3 + best_axis]) - cl;
        sp.cb_lo = cl; sp.scale = (float)kSahBins / ext;
    }
    // the node's children.  A side of one triangle is a leaf link; of two, a node of two leaves; of 3 .. 64, a treelet (its root's slot
    // and range are set here, the treelet pass builds it); of more, an open range of the next level.
    const int i = sg.node;
    for (int side = 0; side < 2; ++side) {
        const int a = side == 0 ? sg.lo : sg.lo + sp.left_count, b = side == 0 ? sg.lo + sp.left_count - 1 : sg.hi;
        const int sz = b - a + 1;
        sp.child[side] = kSahNone; sp.small[side] = kSahNone;
        int ref;
        if (sz == 1) { ref = ~a; plf[a] = i; }
        else {
            const int c = side == 0 ? b : a;                                   // a left child is named by its last position, a right child by its first
            ref = c; pin[c] = i; rlo[c] = a; rhi[c] = b;
            if (sz == 2) { left[c] = ~a; right[c] = ~b; plf[a] = c; plf[b] = c; }
            else if (sz > kTreelet) {
                SahSeg ns; ns.lo = a; ns.hi = b; ns.node = c; ns.depth = sg.depth + 1;
                for (int q = 0; q < 6; ++q) ns.cb[q] = cbs[side][q];
                if (sz > kSahSmall) { const uint32_t slot = atomicAdd(n_next, 1u); next[slot] = ns; sp.child[side] = slot; }
                else                { const uint32_t slot = atomicAdd(n_small, 1u); smalls[slot] = ns; sp.small[side] = slot; }
            }
        }
        if (side == 0) left[i] = ref; else right[i] = ref;
    }
    splits[s] = sp;
    fill[2u * s] = 0u; fill[2u * s + 1u] = 0u;
}

// A workgroup moves a chunk of 2048 positions.  The ranges a chunk touches claim LDS tables as in k_sah_bin; a triangle takes its rank
// on its side from an LDS counter and adds its centroid to the side's bounds there; one thread per table then reserves the chunk's
// places in the range with ONE global atomic per side (and adds the bounds of the sides that stay open), and the triangles move.  A
// range without a table does all that with global atomics of its own (deep levels: many ranges, little contention).  Atomics on one
// address cost ~70 ns each whoever issues them: per wave (16 k waves at the root) the fill counters alone took a millisecond.
constexpr int kSahPerThread = kSahChunk / 256;
__global__ __launch_bounds__(256) void k_sah_partition(const uint32_t* __restrict__ perm_in, const uint32_t* __restrict__ seg_in, uint32_t* perm_out, uint32_t* seg_out,
                                                       const SahSplit* __restrict__ splits, uint32_t* fill, const Box6* __restrict__ boxes, uint32_t n, SahSeg* next, SahSeg* smalls)
{
    __shared__ uint32_t s_dir[kSahTables], s_cnt[kSahTables][2], s_base[kSahTables][2], s_cb[kSahTables][2][6];
    for (uint32_t chunk = blockIdx.x; (size_t)chunk * kSahChunk < n; chunk += gridDim.x) {
        const uint32_t first = chunk * (uint32_t)kSahChunk;
        if (threadIdx.x < (uint32_t)kSahTables) { s_dir[threadIdx.x] = kSahNone; s_cnt[threadIdx.x][0] = 0u; s_cnt[threadIdx.x][1] = 0u; }
        if (threadIdx.x < (uint32_t)kSahTables * 12u) (&s_cb[0][0][0])[threadIdx.x] = (threadIdx.x % 6u) < 3u ? 0xffffffffu : 0u;
        __syncthreads();
        uint32_t prim[kSahPerThread], dest[kSahPerThread], out_seg[kSahPerThread], tab[kSahPerThread];      // tab: table << 1 | side, or kSahNone
        for (int e = 0; e < kSahPerThread; ++e) {
            const uint32_t p = first + (uint32_t)e * 256u + threadIdx.x;
            prim[e] = 0u; dest[e] = kSahNone; out_seg[e] = kSahNone; tab[e] = kSahNone;
            if (p >= n) continue;
            const uint32_t s = seg_in[p];
            prim[e] = perm_in[p]; dest[e] = p;
            if (s == kSahNone) continue;
            const SahSplit sp = splits[s];
            bool goes_left;
            uint32_t oc[3] = { 0u, 0u, 0u };
            if (sp.median) goes_left = (int)p - sp.lo < sp.left_count;             // (no usable axis: the order stays, the bounds are the parent's)
            else {
                const Box6 b = boxes[prim[e]];
                goes_left = sah_bin_of(0.5f * (b.lo[sp.axis] + b.hi[sp.axis]), sp.cb_lo, sp.scale) <= sp.bin;
                for (int k = 0; k < 3; ++k) oc[k] = f_ord(0.5f * (b.lo[k] + b.hi[k]));
            }
            const uint32_t side = goes_left ? 0u : 1u, child = sp.child[side];
            SahSeg* const cbd = child != kSahNone ? &next[child] : (sp.small[side] != kSahNone ? &smalls[sp.small[side]] : nullptr);   // whose centroid bounds this triangle widens
            out_seg[e] = child;
            if (sp.median) continue;
            const uint32_t h = s & (uint32_t)(kSahTables - 1);
            uint32_t owner = s_dir[h];
            if (owner == kSahNone) { owner = atomicCAS(&s_dir[h], kSahNone, s); if (owner == kSahNone) owner = s; }
            if (owner == s) {
                dest[e] = atomicAdd(&s_cnt[h][side], 1u);                          // the rank in this chunk's share of the side: the base comes below
                tab[e] = (h << 1) | side;
                if (cbd != nullptr) for (int k = 0; k < 3; ++k) { atomicMin(&s_cb[h][side][k], oc[k]); atomicMax(&s_cb[h][side][3 + k], oc[k]); }
            } else {
                dest[e] = (uint32_t)(sp.lo + (goes_left ? 0 : sp.left_count)) + atomicAdd(&fill[2u * s + side], 1u);
                if (cbd != nullptr) for (int k = 0; k < 3; ++k) { atomicMin(&cbd->cb[k], oc[k]); atomicMax(&cbd->cb[3 + k], oc[k]); }
            }
        }
        __syncthreads();
        if (threadIdx.x < (uint32_t)kSahTables * 2u) {
            const uint32_t h = threadIdx.x >> 1, side = threadIdx.x & 1u, sid = s_dir[h], c = s_cnt[h][side];
            if (sid != kSahNone && c != 0u) {
                const SahSplit sp = splits[sid];
                s_base[h][side] = (uint32_t)(sp.lo + (side == 0u ? 0 : sp.left_count)) + atomicAdd(&fill[2u * sid + side], c);
                SahSeg* const cbd = sp.child[side] != kSahNone ? &next[sp.child[side]] : (sp.small[side] != kSahNone ? &smalls[sp.small[side]] : nullptr);
                if (cbd != nullptr) for (int k = 0; k < 3; ++k) { atomicMin(&cbd->cb[k], s_cb[h][side][k]); atomicMax(&cbd->cb[3 + k], s_cb[h][side][3 + k]); }
            }
        }
        __syncthreads();
        for (int e = 0; e < kSahPerThread; ++e) {
            if (dest[e] == kSahNone) continue;
            const uint32_t d = tab[e] != kSahNone ? s_base[tab[e] >> 1][tab[e] & 1u] + dest[e] : dest[e];
            perm_out[d] = prim[e]; seg_out[d] = out_seg[e];
        }
        __syncthreads();
    }
}

// ---- ranges of at most 2048 triangles: ONE workgroup takes a range down to its treelets, in LDS -----------------------------------------
// Level by level the deep levels were the expensive ones (1 M triangles: 0.3-0.8 ms of binning and 0.3-0.4 ms of partition per level from
// level 8 on, against 0.03 / 0.05 ms at the top): a chunk of 2048 positions then spans dozens of ranges, more than it has LDS tables, and
// the rest falls back to global atomics.  A range that fits a chunk leaves that loop (k_sah_split puts it on a list) and is finished here:
// its triangles' numbers in LDS (two copies: the partition is out of place), a stack of open sub-ranges, and per sub-range the same three
// steps -- bins by LDS atomics, the sweep by one WAVE per axis (lane = bin: prefix and suffix unions by shuffles, the cheapest split by a
// wave minimum), the partition with LDS counters -- with the same arithmetic and the same tie rules as k_sah_split, so the tree is the one
// the level-by-level loop would have made.
__global__ __launch_bounds__(256) void k_sah_small(const SahSeg* __restrict__ list, const uint32_t* __restrict__ n_list, uint32_t* perm, const Box6* __restrict__ boxes,
                                                   int leaf_max, int depth_cap, int* left, int* right, int* pin, int* plf, int* rlo, int* rhi)
{
    constexpr int kStack = 64;
    __shared__ uint32_t s_perm[2][kSahSmall];
    __shared__ uint32_t s_bins[kSahSegWords];
    __shared__ SahSeg s_stack[kStack];                  // lo / hi are positions inside the range here
    __shared__ uint8_t s_buf[kStack];
    __shared__ int s_sp;
    __shared__ SahSeg s_cur; __shared__ int s_cur_buf;
    __shared__ float s_cost[3]; __shared__ int s_bin[3], s_nl[3];
    __shared__ int s_axis, s_split_bin, s_left, s_median; __shared__ float s_cb_lo, s_scale;
    __shared__ uint32_t s_cnt[2], s_cb[2][6];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t item = blockIdx.x; item < *n_list; item += gridDim.x) {
        const SahSeg root = list[item];
        const int n = root.hi - root.lo + 1;
        for (int i = tid; i < n; i += 256) s_perm[0][i] = perm[root.lo + i];
        if (tid == 0) { SahSeg r = root; r.lo = 0; r.hi = n - 1; s_stack[0] = r; s_buf[0] = 0u; s_sp = 1; }
        for (;;) {
            __syncthreads();
            const int sp = s_sp;                                               // (every thread reads the height before thread 0 changes it)
            __syncthreads();
            if (sp == 0) break;
            if (tid == 0) { s_sp = sp - 1; s_cur = s_stack[sp - 1]; s_cur_buf = s_buf[sp - 1]; }
            for (int w = tid; w < kSahSegWords; w += 256) s_bins[w] = sah_word_identity(w);
            __syncthreads();
            const SahSeg sg = s_cur;
            const int buf = s_cur_buf, a0 = sg.lo, cnt = sg.hi - sg.lo + 1;
            // ---- bins
            for (int i = tid; i < cnt; i += 256) {
                const Box6 b = boxes[s_perm[buf][a0 + i]];
                uint32_t ob[6];
                for (int k = 0; k < 3; ++k) { ob[k] = f_ord(b.lo[k]); ob[3 + k] = f_ord(b.hi[k]); }
                for (int k = 0; k < 3; ++k) {
                    const float cl = f_unord(sg.cb[k]), ext = f_unord(sg.cb[3 + k]) - cl;
                    if (!(ext > 0.0f) || !(ext < 3.0e38f)) continue;
                    const int bi = sah_bin_of(0.5f * (b.lo[k] + b.hi[k]), cl, (float)kSahBins / ext);
                    uint32_t* w = s_bins + (k * kSahBins + bi) * kSahBinWords;
                    atomicAdd(&w[0], 1u);
                    for (int c = 0; c < 3; ++c) { atomicMin(&w[1 + c], ob[c]); atomicMax(&w[4 + c], ob[3 + c]); }
                }
            }
            __syncthreads();
            // ---- the sweep: wave k takes axis k, lane b bin b
            if (wave < 3) {
                float c; int b, nl;
                sah_sweep_axis(s_bins, wave, sg, lane, leaf_max, depth_cap, c, b, nl);
                if (lane == 0) { s_cost[wave] = c; s_bin[wave] = b; s_nl[wave] = nl; }
            }
            __syncthreads();
            if (tid == 0) {
                float best = __builtin_inff(); int ax = -1;
                for (int k = 0; k < 3; ++k) if (s_cost[k] < best) { best = s_cost[k]; ax = k; }
                s_median = ax < 0 ? 1 : 0;
                s_axis = ax < 0 ? 0 : ax; s_split_bin = ax < 0 ? 0 : s_bin[ax];
                s_left = ax < 0 ? cnt / 2 : s_nl[ax];
                if (ax >= 0) { const float cl = f_unord(sg.cb[ax]); s_cb_lo = cl; s_scale = (float)kSahBins / (f_unord(sg.cb[3 + ax]) - cl); }
                s_cnt[0] = 0u; s_cnt[1] = 0u;
                for (int side = 0; side < 2; ++side) for (int q = 0; q < 6; ++q) s_cb[side][q] = ax < 0 ? sg.cb[q] : (q < 3 ? 0xffffffffu : 0u);
            }
            __syncthreads();
            // ---- the partition (out of place: the other copy)
            const int left_count = s_left, median = s_median, axis = s_axis, split_bin = s_split_bin;
            for (int i = tid; i < cnt; i += 256) {
                const uint32_t prim = s_perm[buf][a0 + i];
                int dest;
                if (median) dest = a0 + i;
                else {
                    const Box6 b = boxes[prim];
                    const uint32_t side = sah_bin_of(0.5f * (b.lo[axis] + b.hi[axis]), s_cb_lo, s_scale) <= split_bin ? 0u : 1u;
                    dest = a0 + (side ? left_count : 0) + (int)atomicAdd(&s_cnt[side], 1u);
                    for (int k = 0; k < 3; ++k) { const uint32_t oc = f_ord(0.5f * (b.lo[k] + b.hi[k])); atomicMin(&s_cb[side][k], oc); atomicMax(&s_cb[side][3 + k], oc); }
                }
                s_perm[buf ^ 1][dest] = prim;
            }
            __syncthreads();
            // ---- the node's children (k_sah_split's cases; positions are the range's, the links the whole array's)
            if (tid == 0) {
                const int i = sg.node;
                for (int side = 0; side < 2; ++side) {
                    const int a = side == 0 ? a0 : a0 + left_count, b = side == 0 ? a0 + left_count - 1 : sg.hi;
                    const int sz = b - a + 1, ga = root.lo + a, gb = root.lo + b;
                    int ref;
                    if (sz == 1) { ref = ~ga; plf[ga] = i; }
                    else {
                        const int c = side == 0 ? gb : ga;
                        ref = c; pin[c] = i; rlo[c] = ga; rhi[c] = gb;
                        if (sz == 2) { left[c] = ~ga; right[c] = ~gb; plf[ga] = c; plf[gb] = c; }
                        else if (sz > kTreelet && s_sp < kStack) {
                            SahSeg ns; ns.lo = a; ns.hi = b; ns.node = c; ns.depth = sg.depth + 1;
                            for (int q = 0; q < 6; ++q) ns.cb[q] = s_cb[side][q];
                            s_stack[s_sp] = ns; s_buf[s_sp] = (uint8_t)(buf ^ 1); ++s_sp;
                        }
                        // (a stack that is full -- 64 open sub-ranges of one 2048-triangle range: not with two non-empty sides per split -- would leave
                        //  the side to the treelet pass as it is, which takes ranges of at most 64 only: it cannot happen, and the depth check after
                        //  the build would send the input to the host builder if it did)
                    }
                    if (side == 0) left[i] = ref; else right[i] = ref;
                }
            }
            // the sides that are finished go home: every position of this sub-range that does not stay open
            {
                const int lsz = left_count, rsz = cnt - left_count;
                for (int i = tid; i < cnt; i += 256) {
                    const bool in_left = i < lsz;
                    const int sz = in_left ? lsz : rsz;
                    if (sz <= kTreelet) perm[root.lo + a0 + i] = s_perm[buf ^ 1][a0 + i];
                }
            }
            __syncthreads();
        }
    }
}

// the root range: the first open range of the level loop, or -- a whole input of at most kSahSmall triangles -- the one range of k_sah_small
__global__ void k_sah_root(SahSeg* segs, SahSeg* smalls, const uint32_t* scene, uint32_t n, int* pin, int* rlo, int* rhi, uint32_t* counts)
{
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    SahSeg r; r.lo = 0; r.hi = (int)n - 1; r.node = 0; r.depth = 0;
    for (int k = 0; k < 6; ++k) r.cb[k] = scene[k];
    const bool small = n <= (uint32_t)kSahSmall;
    if (small) smalls[0] = r; else segs[0] = r;
    pin[0] = -1; rlo[0] = 0; rhi[0] = (int)n - 1;
    counts[0] = small ? 0u : 1u; counts[1] = 0u; counts[2] = small ? 1u : 0u;
}

// ---- spatial splits in the level loop (round 4, MIPT_OPT_BUILD = 4) ---------------------------------------------------------------------
// What bvh_build.cpp's build_sbvh buys (references that straddle a split plane are cut in two, so the children do not overlap along
// that axis; 1 M-triangle soup: SAH cost 537 -> 500, the headline frame 0.555 -> 0.52 ms) in its cheapest form, which measured as good as
// the full algorithm there: the plane is the OBJECT split's own plane (no spatial binning, no chopped clipping of every reference into
// every bin), and every reference whose box straddles it is cut whenever the split's two sides overlap noticeably and the range has
// references to spare (`budget`, handed down in proportion to the sides' sizes, as the host builder does).
// The ITEMS of the builder become references: an id into a pool of (box, primitive) pairs -- ids 0 .. n-1 are the triangles, a cut
// keeps the id for its left part (the box shrinks in place) and takes a new id for the right part.  Ranges GROW when cut, so the level
// loop no longer works in place: a range's two sides are counted first (k_sp_count), then given room -- a side that stays open in the next
// level's buffer, a side of at most `ret_below` references in the buffer of RETIRED ranges -- by a bump allocator (k_sp_alloc), then
// written (k_sp_partition).  Where ranges lie in those buffers depends on the run; the tree of the ranges (SpNode) does not, and the host
// lays the retired ranges out in its depth-first order afterwards (a few thousand nodes): positions, Karras' node numbers and links
// follow from that (k_sp_layout sorts each retired range by primitive number, so the order inside is a function of the input as well).
// Everything after -- k_sah_small over the retired ranges, treelets, box fitting, emission -- sees n' items with given boxes.
// A reference's box is the bounds of the part of its triangle inside the cuts so far: the clip is computed in double and rounded
// outward, then cut by the box it had (bvh_build.cpp: clip_tri), so a point of the triangle inside a node's region lies in the box of the
// reference that region holds, and hit records stay a brute-force intersector's.
struct SpSplit {
    int lo, cnt, axis, bin, median, cut, nl_obj, tnode, depth, budget;
    float cb_lo, scale, plane;
    uint32_t child[2], base[2], ret[2];           // child: the side's slot in the next level or kSahNone; base: offset of its room; ret: its retired record or kSahNone
    uint32_t pcb[6];                              // the range's centroid bounds (a median split hands them to both sides)
};
struct SpNode { int parent, child[2]; };          // child >= 0: a node of this list; < 0: ~(index of a retired range)
struct SpRet { uint32_t off, size, final_off; int parent, side, depth; uint32_t cb[6]; };
constexpr int kSpTables = 16;

__device__ __forceinline__ float sp_down(double x)
{
    float f = (float)x;
    if ((double)f > x) { uint32_t u = f2u(f); u = f > 0.0f ? u - 1u : (f < 0.0f ? u + 1u : 0x80000001u); f = u2f(u); }
    return f;
}
__device__ __forceinline__ float sp_up(double x)
{
    float f = (float)x;
    if ((double)f < x) { uint32_t u = f2u(f); u = f > 0.0f ? u + 1u : (f < 0.0f ? u - 1u : 0x00000001u); f = u2f(u); }
    return f;
}
// bounds of { p in triangle : p[k] <= pl } (upper = false) or { p[k] >= pl } (upper = true), cut by `within`; false when empty
__device__ __forceinline__ bool sp_clip_half(const float* __restrict__ v, int k, double pl, bool upper, const Box6& within, Box6& out)
{
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    bool any = false;
    for (int i = 0; i < 3; ++i) {
        const int j = i == 2 ? 0 : i + 1;
        const double p[3] = { (double)v[3 * i], (double)v[3 * i + 1], (double)v[3 * i + 2] }, q[3] = { (double)v[3 * j], (double)v[3 * j + 1], (double)v[3 * j + 2] };
        const double pk = k == 0 ? p[0] : (k == 1 ? p[1] : p[2]), qk = k == 0 ? q[0] : (k == 1 ? q[1] : q[2]);
        if (upper ? pk >= pl : pk <= pl) { any = true; for (int c = 0; c < 3; ++c) { lo[c] = fmin(lo[c], p[c]); hi[c] = fmax(hi[c], p[c]); } }
        if ((pk < pl && qk > pl) || (pk > pl && qk < pl)) {
            const double t = (pl - pk) / (qk - pk);
            double x[3] = { p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1]), p[2] + t * (q[2] - p[2]) };
            if (k == 0) x[0] = pl; else if (k == 1) x[1] = pl; else x[2] = pl;
            any = true;
            for (int c = 0; c < 3; ++c) { lo[c] = fmin(lo[c], x[c]); hi[c] = fmax(hi[c], x[c]); }
        }
    }
    if (!any) return false;
    bool ok = true;
    for (int c = 0; c < 3; ++c) {
        out.lo[c] = fmaxf(within.lo[c], sp_down(lo[c]));
        out.hi[c] = fminf(within.hi[c], sp_up(hi[c]));
        ok = ok && out.lo[c] <= out.hi[c];
    }
    return ok;
}
// where a reference goes when its range is cut at `plane` along `k`: 0 left, 1 right (its box as it is); 2 both (bl, br: the parts);
// 3 left / 4 right with the clipped box (the box straddles the plane, the triangle inside it does not)
__device__ __forceinline__ int sp_classify(const Box6& b, int k, float plane, const float* __restrict__ v, Box6& bl, Box6& br)
{
    const float lo = k == 0 ? b.lo[0] : (k == 1 ? b.lo[1] : b.lo[2]), hi = k == 0 ? b.hi[0] : (k == 1 ? b.hi[1] : b.hi[2]);
    if (hi <= plane) return 0;
    if (lo >= plane) return 1;
    const bool hl = sp_clip_half(v, k, (double)plane, false, b, bl), hr = sp_clip_half(v, k, (double)plane, true, b, br);
    if (hl && hr) return 2;
    if (hl) return 3;
    if (hr) return 4;
    return 0;
}

// the sweep of k_sah_split + the decision to cut.  One wave per open range.
__global__ __launch_bounds__(64) void k_sp_split(const SahSeg* __restrict__ cur, const int* __restrict__ budgets, const uint32_t* __restrict__ n_cur_p, uint32_t* bins,
                                                 SpSplit* splits, uint32_t* cnt2, int leaf_max, int depth_cap, float* root_area_p, float min_overlap)
{
    const uint32_t s = blockIdx.x;
    if (s >= *n_cur_p) return;
    const int lane = (int)threadIdx.x;
    const SahSeg sg = cur[s];
    uint32_t* B = bins + (size_t)s * kSahSegWords;
    const int cnt = sg.hi - sg.lo + 1;
    float best = __builtin_inff(); int best_axis = -1, best_bin = -1, best_nl = 0;
    for (int k = 0; k < 3; ++k) {
        float c; int b, nl;
        sah_sweep_axis(B, k, sg, lane, leaf_max, depth_cap, c, b, nl);
        if (c < best) { best = c; best_axis = k; best_bin = b; best_nl = nl; }
    }
    // the two sides' boxes at the chosen split: unions of the bins' boxes (lane = bin again)
    float ov_area = 0.0f, all_area = 0.0f;
    if (best_axis >= 0) {
        const bool has = lane < kSahBins;
        const uint32_t* w = B + (best_axis * kSahBins + (has ? lane : 0)) * kSahBinWords;
        uint32_t v[kSahBinWords];
        for (int c = 0; c < kSahBinWords; ++c) v[c] = w[c];
        const bool in = has && v[0] != 0u;
        uint32_t l_lo[3], l_hi[3], r_lo[3], r_hi[3];
        for (int c = 0; c < 3; ++c) {
            l_lo[c] = in && lane <= best_bin ? v[1 + c] : 0xffffffffu; l_hi[c] = in && lane <= best_bin ? v[4 + c] : 0u;
            r_lo[c] = in && lane > best_bin ? v[1 + c] : 0xffffffffu;  r_hi[c] = in && lane > best_bin ? v[4 + c] : 0u;
        }
        for (int off = 32; off > 0; off >>= 1)
            for (int c = 0; c < 3; ++c) {
                const uint32_t a0 = (uint32_t)__shfl_xor((int)l_lo[c], off, 64), a1 = (uint32_t)__shfl_xor((int)l_hi[c], off, 64);
                const uint32_t a2 = (uint32_t)__shfl_xor((int)r_lo[c], off, 64), a3 = (uint32_t)__shfl_xor((int)r_hi[c], off, 64);
                l_lo[c] = a0 < l_lo[c] ? a0 : l_lo[c]; l_hi[c] = a1 > l_hi[c] ? a1 : l_hi[c];
                r_lo[c] = a2 < r_lo[c] ? a2 : r_lo[c]; r_hi[c] = a3 > r_hi[c] ? a3 : r_hi[c];
            }
        uint32_t o_lo[3], o_hi[3], u_lo[3], u_hi[3];
        bool overlap = true;
        for (int c = 0; c < 3; ++c) {
            o_lo[c] = l_lo[c] > r_lo[c] ? l_lo[c] : r_lo[c]; o_hi[c] = l_hi[c] < r_hi[c] ? l_hi[c] : r_hi[c];
            u_lo[c] = l_lo[c] < r_lo[c] ? l_lo[c] : r_lo[c]; u_hi[c] = l_hi[c] > r_hi[c] ? l_hi[c] : r_hi[c];
            overlap = overlap && o_lo[c] <= o_hi[c];
        }
        ov_area = overlap ? sah_area(o_lo, o_hi) : 0.0f;
        all_area = sah_area(u_lo, u_hi);
    }
    for (int w = lane; w < kSahSegWords; w += 64) B[w] = sah_word_identity(w);           // the slot's bins, ready for the next level
    if (lane != 0) return;
    if (sg.depth == 0) *root_area_p = all_area;                                          // (one range at the first level: later launches read it)
    const float root_area = sg.depth == 0 ? all_area : *root_area_p;
    SpSplit sp;
    sp.lo = sg.lo; sp.cnt = cnt; sp.median = best_axis < 0 ? 1 : 0;
    sp.axis = sp.median ? 0 : best_axis; sp.bin = best_bin; sp.nl_obj = best_nl;
    sp.tnode = sg.node; sp.depth = sg.depth; sp.budget = budgets[s];
    sp.cb_lo = 0.0f; sp.scale = 0.0f; sp.plane = 0.0f; sp.cut = 0;
    for (int q = 0; q < 6; ++q) sp.pcb[q] = sg.cb[q];
    if (!sp.median) {
        const float cl = f_unord(sg.cb[best_axis]), ext = f_unord(sg.cb[3 + best_axis]) - cl;
        sp.cb_lo = cl; sp.scale = (float)kSahBins / ext;
        sp.plane = cl + (float)(best_bin + 1) / sp.scale;
        // as build_sbvh: only where the object split leaves its sides overlapping, with references and depth to spare
        int levels = depth_cap - sg.depth - 1 - 4 - 3;
        levels = levels < 0 ? 0 : (levels > 40 ? 40 : levels);
        const unsigned long long sp_cap = (unsigned long long)leaf_max << levels;
        sp.cut = sp.budget > 0 && cnt > leaf_max && (unsigned long long)cnt <= sp_cap && ov_area > min_overlap * root_area && sp.plane == sp.plane ? 1 : 0;
    }
    for (int side = 0; side < 2; ++side) { sp.child[side] = kSahNone; sp.base[side] = 0u; sp.ret[side] = kSahNone; }
    splits[s] = sp;
    cnt2[2u * s] = 0u; cnt2[2u * s + 1u] = 0u;
}

// how many references each side of a cut range gets (a chunk of 2048 positions per workgroup, LDS counters per range, one global atomic
// per chunk, range and side)
__global__ __launch_bounds__(256) void k_sp_count(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ seg_of, const SpSplit* __restrict__ splits, uint32_t n_pos,
                                                  const Box6* __restrict__ pbox, const uint32_t* __restrict__ pprim, const float* __restrict__ world, uint32_t* cnt2)
{
    __shared__ uint32_t s_dir[kSpTables], s_cnt[kSpTables][2];
    for (uint32_t chunk = blockIdx.x; (size_t)chunk * kSahChunk < n_pos; chunk += gridDim.x) {
        const uint32_t first = chunk * (uint32_t)kSahChunk;
        if (threadIdx.x < (uint32_t)kSpTables) { s_dir[threadIdx.x] = kSahNone; s_cnt[threadIdx.x][0] = 0u; s_cnt[threadIdx.x][1] = 0u; }
        __syncthreads();
        for (uint32_t p = first + threadIdx.x; p < first + (uint32_t)kSahChunk && p < n_pos; p += 256u) {
            const uint32_t s = seg_of[p];
            const SpSplit sp = splits[s];
            if (!sp.cut) continue;
            const uint32_t id = ids[p];
            Box6 bl, br;
            const int cls = sp_classify(pbox[id], sp.axis, sp.plane, world + (size_t)pprim[id] * 9, bl, br);
            const bool tl = cls == 0 || cls == 2 || cls == 3, tr = cls == 1 || cls == 2 || cls == 4;
            const uint32_t h = s & (uint32_t)(kSpTables - 1);
            uint32_t owner = s_dir[h];
            if (owner == kSahNone) { owner = atomicCAS(&s_dir[h], kSahNone, s); if (owner == kSahNone) owner = s; }
            if (owner == s) { if (tl) atomicAdd(&s_cnt[h][0], 1u); if (tr) atomicAdd(&s_cnt[h][1], 1u); }
            else            { if (tl) atomicAdd(&cnt2[2u * s], 1u); if (tr) atomicAdd(&cnt2[2u * s + 1u], 1u); }
        }
        __syncthreads();
        if (threadIdx.x < (uint32_t)kSpTables * 2u) {
            const uint32_t h = threadIdx.x >> 1, side = threadIdx.x & 1u, sid = s_dir[h], c = s_cnt[h][side];
            if (sid != kSahNone && c != 0u) atomicAdd(&cnt2[2u * sid + side], c);
        }
        __syncthreads();
    }
}

// room for every range's two sides.  cur[0 .. 3]: the bump allocators -- next level's buffer, retired buffer, nodes, retired records
__global__ __launch_bounds__(256) void k_sp_alloc(SpSplit* splits, const uint32_t* __restrict__ n_cur_p, const uint32_t* __restrict__ cnt2, uint32_t* fill, SahSeg* next,
                                                  int* next_budget, uint32_t* n_next, uint32_t* cursors, SpNode* tnodes, SpRet* rets, int leaf_max, int depth_cap,
                                                  int ret_below, uint32_t max_segs, uint32_t max_rets)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= *n_cur_p) return;
    SpSplit sp = splits[s];
    int nl, nr, added = 0;
    if (sp.cut) {
        nl = (int)cnt2[2u * s]; nr = (int)cnt2[2u * s + 1u]; added = nl + nr - sp.cnt;
        int levels = depth_cap - sp.depth - 1 - 4;
        levels = levels < 0 ? 0 : (levels > 40 ? 40 : levels);
        const long long side_cap = (long long)leaf_max << levels;
        if (!(nl > 0 && nr > 0 && nl < sp.cnt && nr < sp.cnt && nl <= side_cap && nr <= side_cap && added >= 0 && added <= sp.budget)) sp.cut = 0;
    }
    if (!sp.cut) { nl = sp.median ? sp.cnt / 2 : sp.nl_obj; nr = sp.cnt - nl; added = 0; }
    const int rem = sp.budget - added;
    const int bud_l = (int)((double)rem * (double)nl / (double)(nl + nr)), bud_r = rem - bud_l;
    for (int side = 0; side < 2; ++side) {
        const int sz = side == 0 ? nl : nr;
        uint32_t cb[6];
        for (int q = 0; q < 6; ++q) cb[q] = sp.median ? sp.pcb[q] : (q < 3 ? 0xffffffffu : 0u);
        if (sz > ret_below) {
            uint32_t slot = atomicAdd(n_next, 1u), tn = atomicAdd(&cursors[2], 1u);
            const uint32_t base = atomicAdd(&cursors[0], (uint32_t)sz);
            // (a list that is full: the counters tell the host, which gives the input up; until then everything stays inside the arrays)
            if (slot >= max_segs) slot = max_segs - 1u;
            if (tn >= max_rets) tn = max_rets - 1u;
            sp.child[side] = slot; sp.base[side] = base; sp.ret[side] = kSahNone;
            SahSeg ns; ns.lo = (int)base; ns.hi = (int)base + sz - 1; ns.node = (int)tn; ns.depth = sp.depth + 1;
            for (int q = 0; q < 6; ++q) ns.cb[q] = cb[q];
            next[slot] = ns; next_budget[slot] = side == 0 ? bud_l : bud_r;
            tnodes[tn].parent = sp.tnode; tnodes[sp.tnode].child[side] = (int)tn;
        } else {
            uint32_t r = atomicAdd(&cursors[3], 1u);
            const uint32_t base = atomicAdd(&cursors[1], (uint32_t)sz);
            if (r > max_rets) r = max_rets;                                            // (the list holds max_rets + 1 records)
            sp.child[side] = kSahNone; sp.base[side] = base; sp.ret[side] = r;
            SpRet rr; rr.off = base; rr.size = (uint32_t)sz; rr.final_off = 0u; rr.parent = sp.tnode; rr.side = side; rr.depth = sp.depth + 1;
            for (int q = 0; q < 6; ++q) rr.cb[q] = cb[q];
            rets[r] = rr;
            tnodes[sp.tnode].child[side] = ~(int)r;
        }
    }
    sp.nl_obj = nl;                                     // (from here on: the left side's size, whichever way the range is split)
    splits[s] = sp;
    fill[2u * s] = 0u; fill[2u * s + 1u] = 0u;
}

// every reference moves to its side(s).  Two passes over a chunk: ranks and bounds first (LDS counters per range and side, the chunk's
// new ids from one LDS counter), then -- bases known -- the writes; a reference that is cut is clipped in both (no boxes held across).
__global__ __launch_bounds__(256) void k_sp_partition(const uint32_t* __restrict__ ids_in, const uint32_t* __restrict__ seg_in, uint32_t n_pos, const SpSplit* __restrict__ splits,
                                                      uint32_t* ids_out, uint32_t* seg_out, uint32_t* retired, uint32_t* fill, Box6* pbox, uint32_t* pprim, Box6* pcell,
                                                      const float* __restrict__ world, uint32_t* pool_cursor, uint32_t pool_cap, SahSeg* next, SpRet* rets)
{
    __shared__ uint32_t s_dir[kSpTables], s_cnt[kSpTables][2], s_base[kSpTables][2], s_cb[kSpTables][2][6], s_new, s_newbase;
    for (uint32_t chunk = blockIdx.x; (size_t)chunk * kSahChunk < n_pos; chunk += gridDim.x) {
        const uint32_t first = chunk * (uint32_t)kSahChunk;
        if (threadIdx.x < (uint32_t)kSpTables) { s_dir[threadIdx.x] = kSahNone; s_cnt[threadIdx.x][0] = 0u; s_cnt[threadIdx.x][1] = 0u; }
        if (threadIdx.x < (uint32_t)kSpTables * 12u) (&s_cb[0][0][0])[threadIdx.x] = (threadIdx.x % 6u) < 3u ? 0xffffffffu : 0u;
        if (threadIdx.x == 0u) s_new = 0u;
        __syncthreads();
        uint32_t rank[kSahPerThread][2], newrank[kSahPerThread], tab[kSahPerThread]; int cls[kSahPerThread];
        for (int e = 0; e < kSahPerThread; ++e) {
            const uint32_t p = first + (uint32_t)e * 256u + threadIdx.x;
            cls[e] = -1; rank[e][0] = kSahNone; rank[e][1] = kSahNone; newrank[e] = kSahNone; tab[e] = kSahNone;
            if (p >= n_pos) continue;
            const uint32_t s = seg_in[p];
            const SpSplit sp = splits[s];
            if (sp.median) { cls[e] = (int)p - sp.lo < sp.nl_obj ? 0 : 1; continue; }       // (no usable axis: the order stays, the bounds are the parent's)
            const uint32_t id = ids_in[p];
            const Box6 b = pbox[id];
            Box6 bl = b, br = b;
            int c;
            if (sp.cut) c = sp_classify(b, sp.axis, sp.plane, world + (size_t)pprim[id] * 9, bl, br);
            else        c = sah_bin_of(0.5f * (b.lo[sp.axis] + b.hi[sp.axis]), sp.cb_lo, sp.scale) <= sp.bin ? 0 : 1;
            cls[e] = c;
            const bool tl = c == 0 || c == 2 || c == 3, tr = c == 1 || c == 2 || c == 4;
            if (c == 2) newrank[e] = atomicAdd(&s_new, 1u);
            const uint32_t h = s & (uint32_t)(kSpTables - 1);
            uint32_t owner = s_dir[h];
            if (owner == kSahNone) { owner = atomicCAS(&s_dir[h], kSahNone, s); if (owner == kSahNone) owner = s; }
            const bool mine = owner == s;
            if (mine) tab[e] = h;
            for (int side = 0; side < 2; ++side) {
                if (!(side == 0 ? tl : tr)) continue;
                const Box6& pb = side == 0 ? bl : br;
                uint32_t oc[3];
                for (int k = 0; k < 3; ++k) oc[k] = f_ord(0.5f * (pb.lo[k] + pb.hi[k]));
                uint32_t* cbd = sp.child[side] != kSahNone ? next[sp.child[side]].cb : rets[sp.ret[side]].cb;
                if (mine) {
                    rank[e][side] = atomicAdd(&s_cnt[h][side], 1u);
                    for (int k = 0; k < 3; ++k) { atomicMin(&s_cb[h][side][k], oc[k]); atomicMax(&s_cb[h][side][3 + k], oc[k]); }
                } else {
                    rank[e][side] = atomicAdd(&fill[2u * s + side], 1u);
                    for (int k = 0; k < 3; ++k) { atomicMin(&cbd[k], oc[k]); atomicMax(&cbd[3 + k], oc[k]); }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < (uint32_t)kSpTables * 2u) {
            const uint32_t h = threadIdx.x >> 1, side = threadIdx.x & 1u, sid = s_dir[h], c = s_cnt[h][side];
            if (sid != kSahNone && c != 0u) {
                const SpSplit sp = splits[sid];
                s_base[h][side] = atomicAdd(&fill[2u * sid + side], c);
                uint32_t* cbd = sp.child[side] != kSahNone ? next[sp.child[side]].cb : rets[sp.ret[side]].cb;
                for (int k = 0; k < 3; ++k) { atomicMin(&cbd[k], s_cb[h][side][k]); atomicMax(&cbd[3 + k], s_cb[h][side][3 + k]); }
            }
        }
        if (threadIdx.x == 64u && s_new != 0u) s_newbase = atomicAdd(pool_cursor, s_new);
        __syncthreads();
        for (int e = 0; e < kSahPerThread; ++e) {
            if (cls[e] < 0) continue;
            const uint32_t p = first + (uint32_t)e * 256u + threadIdx.x;
            const uint32_t s = seg_in[p];
            const SpSplit sp = splits[s];
            const uint32_t id = ids_in[p];
            int c = cls[e];
            uint32_t id_r = id;
            if (c >= 2) {
                const Box6 b = pbox[id];
                Box6 bl = b, br = b;
                sp_classify(b, sp.axis, sp.plane, world + (size_t)pprim[id] * 9, bl, br);
                if (c == 2) {
                    // the CELLS of the two parts: the half-spaces of the cuts this triangle has been through.  The cells of a triangle's records
                    // tile space, so after the triangle has moved (mipt_accel_update) "box of the triangle, cut by the cell" still bounds the part
                    // of it each record answers for
                    Box6 cl = pcell[id], cr = cl;
                    if (sp.axis == 0) { cl.hi[0] = fminf(cl.hi[0], sp.plane); cr.lo[0] = fmaxf(cr.lo[0], sp.plane); }
                    else if (sp.axis == 1) { cl.hi[1] = fminf(cl.hi[1], sp.plane); cr.lo[1] = fmaxf(cr.lo[1], sp.plane); }
                    else { cl.hi[2] = fminf(cl.hi[2], sp.plane); cr.lo[2] = fmaxf(cr.lo[2], sp.plane); }
                    id_r = s_newbase + newrank[e];
                    if (id_r < pool_cap) { pbox[id_r] = br; pprim[id_r] = pprim[id]; pcell[id_r] = cr; }
                    pbox[id] = bl; pcell[id] = cl;
                } else pbox[id] = c == 3 ? bl : br;             // (the triangle does not reach across the plane inside its cell TODAY: the cell stays)
            }
            for (int side = 0; side < 2; ++side) {
                const bool takes = side == 0 ? (c == 0 || c == 2 || c == 3) : (c == 1 || c == 2 || c == 4);
                if (!takes) continue;
                uint32_t d;
                if (sp.median) d = (uint32_t)((int)p - sp.lo - (side ? sp.nl_obj : 0));
                else d = (tab[e] != kSahNone ? s_base[tab[e]][side] : 0u) + rank[e][side];
                d += sp.base[side];
                if (d >= pool_cap) continue;                                     // (cannot be: the budget keeps every buffer inside its room)
                const uint32_t v = side == 1 && c == 2 ? id_r : id;
                if (sp.child[side] != kSahNone) { ids_out[d] = v; seg_out[d] = sp.child[side]; }
                else retired[d] = v;
            }
        }
        __syncthreads();
    }
}

// a retired range in its final place: sorted by primitive number (the order its references arrived in depends on the run), boxes and
// primitive numbers gathered, the items numbered by their positions
__global__ __launch_bounds__(256) void k_sp_layout(const SpRet* __restrict__ rets, uint32_t n_rets, const uint32_t* __restrict__ retired, const Box6* __restrict__ pbox,
                                                   const uint32_t* __restrict__ pprim, const Box6* __restrict__ pcell, Box6* fbox, uint32_t* fprim, Box6* fcell, uint32_t* perm)
{
    __shared__ unsigned long long s_key[kSahSmall];
    for (uint32_t r = blockIdx.x; r < n_rets; r += gridDim.x) {
        const SpRet rr = rets[r];
        int width = 2;
        while (width < (int)rr.size) width <<= 1;
        for (int i = (int)threadIdx.x; i < width; i += 256) {
            unsigned long long key = ~0ull;
            if (i < (int)rr.size) { const uint32_t id = retired[rr.off + (uint32_t)i]; key = ((unsigned long long)pprim[id] << 32) | id; }
            s_key[i] = key;
        }
        __syncthreads();
        for (int k = 2; k <= width; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = (int)threadIdx.x; i < width; i += 256) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a = s_key[i], b = s_key[l];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { s_key[i] = b; s_key[l] = a; }
                    }
                }
                __syncthreads();
            }
        for (int i = (int)threadIdx.x; i < (int)rr.size; i += 256) {
            const uint32_t id = (uint32_t)s_key[i], at = rr.final_off + (uint32_t)i;
            fbox[at] = pbox[id]; fprim[at] = pprim[id]; fcell[at] = pcell[id]; perm[at] = at;
        }
        __syncthreads();
    }
}

// the links the host made from the tree of the ranges (Karras' numbering over the final positions)
struct SpLink { int node, left, right, parent, lo, hi; };      // left / right = kSpKeep: set by a later pass (k_sah_small, treelets)
struct SpLeaf { int pos, parent; };
constexpr int kSpKeep = 0x7ffffff1;
__global__ __launch_bounds__(256) void k_sp_links(const SpLink* __restrict__ links, uint32_t n_links, const SpLeaf* __restrict__ leaves, uint32_t n_leaves,
                                                  int* left, int* right, int* pin, int* plf, int* rlo, int* rhi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_links) {
        const SpLink l = links[i];
        if (l.left != kSpKeep) left[l.node] = l.left;
        if (l.right != kSpKeep) right[l.node] = l.right;
        pin[l.node] = l.parent; rlo[l.node] = l.lo; rhi[l.node] = l.hi;
    }
    if (i < n_leaves) plf[leaves[i].pos] = leaves[i].parent;
}

__global__ __launch_bounds__(256) void k_sp_init(uint32_t* ids, uint32_t* seg_of, uint32_t* pprim, Box6* pcell, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ids[i] = i; seg_of[i] = 0u; pprim[i] = i;
    Box6 c;
    for (int k = 0; k < 3; ++k) { c.lo[k] = -__builtin_inff(); c.hi[k] = __builtin_inff(); }
    pcell[i] = c;
}

// ---- bottom-up passes without fences ---------------------------------------------------------------------------------------------
// Round 2 fitted boxes with "the second thread to arrive at a node computes it and climbs on": every step of the climb needs an
// agent-scope release / acquire pair (__threadfence = L2 write-back + invalidate, microseconds each on a chip of eight L2s), and
// 3.7 ms of a 6 ms LBVH build and 3.3 ms of a refit went there.  Now: (1) the HEIGHT of every inner node (1 + the larger height of
// its inner children) by a climb that exchanges everything through ONE returning atomic per step -- a child adds its height into
// its slot's byte of the parent's word together with an arrival count, and whoever completes the count has both heights in the
// value the atomic returned: no fence, no second access; (2) the boxes level by level, one launch per height: a node reads what
// the launch before wrote.  Heights depend on the topology only, so a refit reuses them.
// word: bits 0-7 / 8-15 = height of the inner child in slot 0 / 1, bits 24+ = inner children that have reported
__global__ __launch_bounds__(256) void k_lbvh_heights(const int* left, const int* right, const int* parent_inner, int n_inner, uint32_t* word, uint8_t* height)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner) return;
    if (left[i] >= 0 || right[i] >= 0) return;                      // not a bottom node: its last child to report carries on from here
    uint32_t h = 1u;
    int node = i;
    height[node] = 1u;
    for (;;) {
        const int p = parent_inner[node];
        if (p < 0) return;
        const int lc = left[p], rc = right[p];
        const uint32_t slot = lc == node ? 0u : 1u, need = (lc >= 0 ? 1u : 0u) + (rc >= 0 ? 1u : 0u);
        const uint32_t old = atomicAdd(&word[p], (h << (8u * slot)) + (1u << 24));
        if ((old >> 24) + 1u < need) return;
        const uint32_t other = (old >> (8u * (1u - slot))) & 0xffu;
        h = 1u + (h > other ? h : other);
        h = h < 255u ? h : 255u;
        height[p] = (uint8_t)h;
        node = p;
    }
}

__global__ __launch_bounds__(256) void k_lbvh_fit_level(const Box6* prim_boxes, const uint32_t* sorted_prims, int n_inner, const int* left, const int* right,
                                                       const uint8_t* height, uint32_t level, Box6* node_boxes)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner || height[i] != level) return;
    const int lc = left[i], rc = right[i];
    const Box6 a = lc >= 0 ? node_boxes[lc] : prim_boxes[sorted_prims[~lc]];
    const Box6 b = rc >= 0 ? node_boxes[rc] : prim_boxes[sorted_prims[~rc]];
    node_boxes[i] = box_union(a, b);
}

// Which radix-tree nodes become nodes of the BVH.  A node over more than leaf_max triangles always does.  A smaller one does when
// splitting pays for the extra node by the surface-area heuristic -- the host builder's rule (bvh_build.cpp: keep the leaf unless
// (area(L) nL + area(R) nR) / area + node cost < n, node cost 1.2 triangle tests) -- and every node above it up to the first big one
// decided the same (a folded node takes its whole subtree into the leaf).  sah = false: every subtree of at most leaf_max triangles
// is a leaf (round 2).
__device__ __forceinline__ float half_area(const Box6& b) { const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2]; return dx * dy + dy * dz + dz * dx; }

__device__ __forceinline__ bool split_pays(int i, const int* left, const int* right, const int* range_lo, const int* range_hi, const Box6* prim_boxes,
                                           const uint32_t* sorted_prims, const Box6* node_boxes, float node_cost)
{
    const int lc = left[i], rc = right[i];
    const float al = half_area(lc >= 0 ? node_boxes[lc] : prim_boxes[sorted_prims[~lc]]), ar = half_area(rc >= 0 ? node_boxes[rc] : prim_boxes[sorted_prims[~rc]]);
    const float nl = lc >= 0 ? (float)(range_hi[lc] - range_lo[lc] + 1) : 1.0f, nr = rc >= 0 ? (float)(range_hi[rc] - range_lo[rc] + 1) : 1.0f;
    const float area = half_area(node_boxes[i]);
    return area > 0.0f && (al * nl + ar * nr) / area + node_cost < nl + nr;
}

__global__ __launch_bounds__(256) void k_lbvh_flags(const int* range_lo, const int* range_hi, int n_inner, int leaf_max, uint32_t* emit, bool sah, const int* left,
                                                    const int* right, const int* parent_inner, const Box6* prim_boxes, const uint32_t* sorted_prims, const Box6* node_boxes,
                                                    float node_cost)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner) return;
    bool on = i == 0 || range_hi[i] - range_lo[i] + 1 > leaf_max;
    if (!on && sah) {
        on = true;
        for (int a = i; on && a > 0 && range_hi[a] - range_lo[a] + 1 <= leaf_max; a = parent_inner[a])
            on = split_pays(a, left, right, range_lo, range_hi, prim_boxes, sorted_prims, node_boxes, node_cost);
    }
    emit[i] = on ? 1u : 0u;
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

__global__ __launch_bounds__(256) void k_lbvh_tris(const float* __restrict__ world, const uint32_t* sorted_prims, const uint32_t* gflags, uint32_t n, MiptBvhTri* out,
                                                  uint32_t* prim_slot, const uint32_t* __restrict__ ref_prim, const Box6* __restrict__ ref_cell, Box6* cells_out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t p = sorted_prims[k];
    if (cells_out) cells_out[k] = ref_cell[p];          // the records' cells, in record order: what a refit cuts the moved triangles' boxes by
    if (ref_prim) p = ref_prim[p];                      // (spatial splits: the items are references; one of a triangle's records wins its prim_slot)
    if (prim_slot) prim_slot[p] = k;
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
    int guard = 0;
    for (int node = parent_leaf[k]; node >= 0 && guard < 65536; node = parent_inner[node], ++guard) d += emit[node];
    if (guard >= 65536) d = 0xffffu;                               // (a broken parent chain: reported as too deep, the host builder takes over)
    atomicMax(depth, d);
}

// ---- refit (mipt_accel_update): same topology, new vertex positions -------------------------------------------------------
// triangle records of the existing leaf order from the new world-space vertices (n: the records -- more than there are triangles when the
// tree was built with spatial splits; a record's primitive number is below the number of triangles, hence below n)
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

// heights of the traversal-layout tree (same scheme as k_lbvh_heights; parent[c] = node * 2 + slot)
__global__ __launch_bounds__(256) void k_topo_heights(const MiptBvhNode* nodes, uint32_t nn, const int* parent, uint32_t* word, uint8_t* height)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    if (nodes[i].c0 >= 0 || nodes[i].c1 >= 0) return;
    uint32_t h = 1u, node = i;
    height[node] = 1u;
    for (;;) {
        const int pr = parent[node];
        if (pr < 0) return;
        const uint32_t p = (uint32_t)pr >> 1, slot = (uint32_t)pr & 1u;
        const uint32_t need = (nodes[p].c0 >= 0 ? 1u : 0u) + (nodes[p].c1 >= 0 ? 1u : 0u);
        const uint32_t old = atomicAdd(&word[p], (h << (8u * slot)) + (1u << 24));
        if ((old >> 24) + 1u < need) return;
        const uint32_t other = (old >> (8u * (1u - slot))) & 0xffu;
        h = 1u + (h > other ? h : other);
        h = h < 255u ? h : 255u;
        height[p] = (uint8_t)h;
        node = p;
    }
}

// refit, step 1: the boxes of every LEAF child from the new vertices (all nodes at once)
__global__ __launch_bounds__(256) void k_refit_leaf_slots(MiptBvhNode* nodes, uint32_t nn, const MiptBvhTri* tris, uint32_t nt, const float* __restrict__ world,
                                                          const Box6* __restrict__ cells)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
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
            float tl[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, th[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
            for (int c = 0; c < 9; ++c) { const float f = v[c]; tl[c % 3] = fminf(tl[c % 3], f); th[c % 3] = fmaxf(th[c % 3], f); }
            if (cells) {
                // a record of a tree with spatial splits answers for the part of its triangle inside its cell.  As the builder made its box:
                // the triangle's box, cut in turn by the bounds of the triangle's part on the record's side of every plane of the cell (the
                // bounds of an intersection lie inside the intersection of the bounds) -- for vertices that have not moved, the box the
                // builder gave it.  A triangle that has left the cell altogether keeps its whole box (a sound, if loose, choice)
                const Box6 cell = cells[first + j];
                Box6 b; bool some = true;
                for (int k = 0; k < 3; ++k) { b.lo[k] = tl[k]; b.hi[k] = th[k]; }
                for (int k = 0; k < 3 && some; ++k) {
                    const float pl_hi = k == 0 ? cell.hi[0] : (k == 1 ? cell.hi[1] : cell.hi[2]), pl_lo = k == 0 ? cell.lo[0] : (k == 1 ? cell.lo[1] : cell.lo[2]);
                    if (pl_hi < 3.0e38f) { Box6 o; some = sp_clip_half(v, k, (double)pl_hi, false, b, o); if (some) b = o; }
                    if (some && pl_lo > -3.0e38f) { Box6 o; some = sp_clip_half(v, k, (double)pl_lo, true, b, o); if (some) b = o; }
                }
                if (some) for (int k = 0; k < 3; ++k) { tl[k] = b.lo[k]; th[k] = b.hi[k]; }
            }
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], tl[k]); hi[k] = fmaxf(hi[k], th[k]); }
        }
        refit_store_slot(nd, slot, lo, hi);
    }
}

// refit, step 2, one launch per height h = 2 .. depth: the boxes of the INNER children of the nodes of that height (a child is
// lower, so its own two boxes are final: written by step 1 or an earlier launch)
__global__ __launch_bounds__(256) void k_refit_level(MiptBvhNode* nodes, uint32_t nn, const uint8_t* height, uint32_t level)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn || height[i] != level) return;
    MiptBvhNode* nd = &nodes[i];
    for (int slot = 0; slot < 2; ++slot) {
        const int ref = slot == 0 ? nd->c0 : nd->c1;
        if (ref < 0 || (uint32_t)ref >= nn) continue;
        const MiptBvhNode* ch = &nodes[ref];
        const float lo[3] = { fminf(ch->c0x[0], ch->c1x[0]), fminf(ch->c0y[0], ch->c1y[0]), fminf(ch->c0z[0], ch->c1z[0]) };
        const float hi[3] = { fmaxf(ch->c0x[1], ch->c1x[1]), fmaxf(ch->c0y[1], ch->c1y[1]), fmaxf(ch->c0z[1], ch->c1z[1]) };
        refit_store_slot(nd, slot, lo, hi);
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
// The heights come from topo_heights() (once per build: a refit keeps the topology).
struct B4Work { uint32_t x, budget, used, pad; };      // BVH2 node x is BVH4 node <its position in the work array>

__device__ __forceinline__ float b4_area(const uint4 q)
{
    const float dx = half_bits_to_float(q.x >> 16) - half_bits_to_float(q.x & 0xffffu);
    const float dy = half_bits_to_float(q.y >> 16) - half_bits_to_float(q.y & 0xffffu);
    const float dz = half_bits_to_float(q.z >> 16) - half_bits_to_float(q.z & 0xffffu);
    return dx * dy + dy * dz + dz * dx;
}

// cnt[k]: BVH4 nodes of level k (cnt[0] = 1, the rest zero at the start); misc[0]: the most stack entries a lane can need
__global__ __launch_bounds__(256) void k_b4_level(const uint4* __restrict__ n32, uint32_t nn, const uint8_t* __restrict__ height, uint32_t level,
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
            auto h_of = [&](uint32_t ref) { return (int)ref >= 0 && ref < nn ? (uint32_t)height[ref] : 0u; };
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

// ---- two-level structures on the entry pool: the pieces of the build that touch BVH4 nodes -------------------------------------------
__global__ __launch_bounds__(256) void k_rebase_nodes4(const uint4* __restrict__ src, uint4* dst, uint32_t n4, uint32_t node_base, uint32_t tri_base)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece (one child) per thread
    if (i >= n4 * 4u) return;
    uint4 q = src[i];
    const int ref = (int)q.w;
    if (ref >= 0) { if (ref != 0x7fffffff) q.w = (uint32_t)ref + node_base; }
    else { const uint32_t desc = ~(uint32_t)ref; q.w = ~((((desc >> 4) + tri_base) << 4) | (desc & 15u)); }
    dst[i] = q;
}

__global__ __launch_bounds__(256) void k_tlas4_leaves(uint4* nodes4, uint32_t n4, const uint32_t* __restrict__ order, const DeviceInstance* __restrict__ inst,
                                                      uint32_t* root4_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4 * 4u) return;
    const int ref = (int)nodes4[i].w;
    uint32_t r4 = 0u;
    if (ref < 0) {                                                              // leaves hold ONE instance each
        const uint32_t desc = ~(uint32_t)ref;
        if ((desc & 15u) == 0u) nodes4[i].w = 0x7fffffffu;                      // the empty second child of a structure of ONE instance: no child (kNoChild)
        else {
            const uint32_t k = order[desc >> 4];
            nodes4[i].w = ~((k << 4) | 1u);
            r4 = inst[k].root4;
        }
    }
    root4_out[i] = r4;
}

}  // namespace

void launch_rebase_nodes4(hipStream_t st, const void* src, void* dst, uint32_t n4, uint32_t node_base, uint32_t tri_base)
{
    if (n4 == 0u) return;
    hipLaunchKernelGGL(k_rebase_nodes4, dim3((n4 * 4u + 255u) / 256u), dim3(256), 0, st, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n4, node_base, tri_base);
}

void launch_tlas4_leaves(hipStream_t st, void* nodes4, uint32_t n4, const uint32_t* order, const DeviceInstance* inst, uint32_t* root4_out)
{
    if (n4 == 0u) return;
    hipLaunchKernelGGL(k_tlas4_leaves, dim3((n4 * 4u + 255u) / 256u), dim3(256), 0, st, reinterpret_cast<uint4*>(nodes4), n4, order, inst, root4_out);
}

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
// height of every node of the traversal-layout tree (1 = both children are leaves), into d_height[nn]; scratch: 2 * nn words
size_t topo_heights_scratch_bytes(uint32_t nn) { return 2 * (((size_t)nn * 4u + 255) & ~(size_t)255); }

hipError_t topo_heights(hipStream_t st, const float4* d_nodes, uint32_t nn, uint8_t* d_height, void* scratch, size_t scratch_bytes)
{
    const size_t r = ((size_t)nn * 4u + 255) & ~(size_t)255;
    char* tmp = nullptr; bool own = false;
    hipError_t e = hipSuccess;
    if (scratch != nullptr && scratch_bytes >= 2 * r) tmp = (char*)scratch;
    else { e = hipMalloc((void**)&tmp, 2 * r); own = true; }
    if (e != hipSuccess) return e;
    int* parent = (int*)tmp; uint32_t* word = (uint32_t*)(tmp + r);
    e = hipMemsetAsync(parent, 0xff, (size_t)nn * 4u, st);
    if (e == hipSuccess) e = hipMemsetAsync(word, 0, (size_t)nn * 4u, st);
    if (e == hipSuccess) {
        const MiptBvhNode* nodes = reinterpret_cast<const MiptBvhNode*>(d_nodes);
        hipLaunchKernelGGL(k_refit_parents, dim3((nn + 255u) / 256u), dim3(256), 0, st, nodes, nn, parent);
        hipLaunchKernelGGL(k_topo_heights, dim3((nn + 255u) / 256u), dim3(256), 0, st, nodes, nn, parent, word, d_height);
        e = hipGetLastError();
        if (own && e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (own) (void)hipFree(tmp);
    return e;
}

size_t collapse_bvh4_scratch_bytes(uint32_t nn) { return (((size_t)nn * sizeof(B4Work) + 255) & ~(size_t)255) + 512; }

// The BVH4 mirror of the packed BVH2 at d_nodes32, written to d_nodes4 (room for nn nodes of 64 B); d_height: topo_heights() of the
// same tree.  *ok = false (and nothing to use) when the tree is higher than 32 or the mirror would not fit; *n4 nodes, *stack_need
// entries per lane at most.
hipError_t collapse_bvh4(hipStream_t st, const void* d_nodes32, const uint8_t* d_height, uint32_t nn, uint32_t depth, void* d_nodes4, uint32_t* n4,
                         uint32_t* stack_need, bool* ok, void* scratch, size_t scratch_bytes, uint32_t budget)
{
    *ok = false; *n4 = 0u; *stack_need = 0u;
    if (nn == 0u || depth == 0u || depth > 32u || depth > budget) return hipSuccess;          // (a tree of height h always fits h entries: the budget must cover that)
    const size_t o_work = 0, o_cnt = o_work + (((size_t)nn * sizeof(B4Work) + 255) & ~(size_t)255), total = o_cnt + 512;
    char* tmp = nullptr; bool own = false;
    hipError_t e = hipSuccess;
    if (scratch != nullptr && scratch_bytes >= total) tmp = (char*)scratch;
    else { e = hipMalloc((void**)&tmp, total); own = true; }
    if (e != hipSuccess) return e;
    B4Work* work = (B4Work*)(tmp + o_work); uint32_t* cnt = (uint32_t*)(tmp + o_cnt); uint32_t* misc = cnt + 64;
    auto fail = [&](hipError_t err) { if (own) (void)hipFree(tmp); return err; };
    if ((e = hipMemsetAsync(cnt, 0, 512, st)) != hipSuccess) return fail(e);
    const uint32_t one = 1u;
    const B4Work root{ 0u, budget < 32u ? budget : 32u, 0u, 0u };
    if ((e = hipMemcpyAsync(cnt, &one, 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(work, &root, sizeof root, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    const uint4* n32 = reinterpret_cast<const uint4*>(d_nodes32);
    const uint32_t blocks = (nn + 255u) / 256u;
    // a BVH4 level is at most a BVH2 level deep, so `depth` passes reach every node; a pass whose level is empty returns at once
    const uint32_t grid = blocks < 2048u ? blocks : 2048u;
    for (uint32_t level = 0; level < depth; ++level)
        hipLaunchKernelGGL(k_b4_level, dim3(grid), dim3(256), 0, st, n32, nn, d_height, level, cnt, work, nn, reinterpret_cast<uint4*>(d_nodes4), misc);
    uint32_t h_cnt[66];
    if ((e = hipMemcpyAsync(h_cnt, cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    if (own) (void)hipFree(tmp);
    uint64_t tot = 0;
    for (uint32_t k = 0; k <= depth && k < 64u; ++k) tot += h_cnt[k];
    if (h_cnt[65] != 0u || tot > nn || h_cnt[depth < 64u ? depth : 63u] != 0u || h_cnt[64] > 32u) return hipSuccess;      // (none of these can happen)
    *n4 = (uint32_t)tot; *stack_need = h_cnt[64]; *ok = true;
    return hipSuccess;
}

// Refit of an existing tree (same topology) to new world-space vertices: triangle records, the boxes of the leaf children, then the
// inner children's boxes one height at a time (d_height: topo_heights() of the tree; depth = the root's height).  Asynchronous on st.
hipError_t lbvh_refit(hipStream_t st, const float* d_world, uint32_t n, float4* d_nodes, uint32_t nn, float4* d_tris, const uint8_t* d_height, uint32_t depth,
                      const float* d_cells)
{
    MiptBvhNode* nodes = reinterpret_cast<MiptBvhNode*>(d_nodes);
    MiptBvhTri* tris = reinterpret_cast<MiptBvhTri*>(d_tris);
    const uint32_t blocks = (nn + 255u) / 256u;
    hipLaunchKernelGGL(k_refit_tris, dim3((n + 255u) / 256u), dim3(256), 0, st, d_world, n, tris);
    hipLaunchKernelGGL(k_refit_leaf_slots, dim3(blocks), dim3(256), 0, st, nodes, nn, tris, n, d_world, reinterpret_cast<const Box6*>(d_cells));
    for (uint32_t level = 2; level <= depth; ++level)
        hipLaunchKernelGGL(k_refit_level, dim3(blocks), dim3(256), 0, st, nodes, nn, d_height, level);
    return hipGetLastError();
}

void launch_gather_input(hipStream_t st, const GatherInput& in, float* world, uint32_t* mats, uint32_t* gflags, uint32_t base, uint32_t* error)
{
    if (in.num_triangles == 0u) return;
    hipLaunchKernelGGL(k_gather_input, dim3((in.num_triangles + 255u) / 256u), dim3(256), 0, st, in, world, mats, gflags, base, error);
}

// d_world: 9 floats per triangle (device); d_gflags: per triangle or null.  On success *d_blob is ONE allocation
// nodes | triangles | 64 B pad (as mipt_accel_build makes for the host builder).  Returns hipSuccess and ok = false
// when the tree is deeper than depth_cap (the caller falls back to the host builder).
namespace {
#define SP_TRY(x) do { if ((e = (x)) != hipSuccess) { if (std::getenv("MIPT_BUILD_DEBUG")) std::fprintf(stderr, "lbvh_build.hip:%d: %s\n", __LINE__, hipGetErrorString(e)); return fail(e); } } while (0)
hipError_t lbvh_build_impl(hipStream_t st, const float* d_world, const uint32_t* d_gflags, const uint32_t n_in, int leaf_max, int depth_cap,
                           float4** d_blob, uint32_t* num_nodes, uint32_t* depth, bool* ok, bool sah_treelets, uint32_t* d_prim_slot, bool sah_top,
                           float spatial_extra, int ret_below, uint32_t* num_records, float** d_cells_out)
{
    *ok = false; *d_blob = nullptr; *num_records = n_in; *d_cells_out = nullptr;
    // spatial splits (the comment at struct SpSplit): the level loop may add up to `sp_budget` references, every array below has room for them
    const bool top_down = sah_top && sah_treelets && n_in > 2u * (uint32_t)kTreelet;
    if (ret_below > kSahSmall) ret_below = kSahSmall;
    if (ret_below < kTreelet) ret_below = kTreelet;
    // (the entry pool's words hold a 23-bit record index: the records stay below 2^23, whatever the budget asks for)
    uint32_t sp_budget = top_down && spatial_extra > 0.0f && n_in > (uint32_t)ret_below && n_in < (1u << 23) - 1u ? (uint32_t)fmin((double)n_in * (double)spatial_extra, 4.0e9) : 0u;
    if (sp_budget > (1u << 23) - 1u - n_in) sp_budget = (1u << 23) - 1u - n_in;
    const bool spatial = sp_budget != 0u;
    const uint32_t cap = n_in + sp_budget;
    uint32_t n = n_in;                                   // the items of the passes after the front end: triangles, or references
    int n_inner = (int)n - 1;
    uint32_t blocks = (n + 255u) / 256u;
    char* tmp = nullptr;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_boxes = carve(sizeof(Box6) * cap), o_nboxes = carve(sizeof(Box6) * cap), o_keys = carve(8ull * cap), o_keys2 = carve(8ull * cap),
                 o_vals = carve(4ull * cap), o_vals2 = carve(4ull * cap), o_left = carve(4ull * cap), o_right = carve(4ull * cap), o_pin = carve(4ull * cap),
                 o_plf = carve(4ull * cap), o_lo = carve(4ull * cap), o_hi = carve(4ull * cap), o_arr = carve(4ull * cap), o_emit = carve(4ull * cap),
                 o_idx = carve(4ull * cap), o_kh = carve(cap), o_big = carve(cap), o_misc = carve(256);
    // the top-down SAH front end (k_sah_*): open ranges of two levels, their splits, fill counters and bins
    const size_t max_segs = top_down ? (size_t)cap / (size_t)kTreelet + 4u : 0u;
    const size_t o_seg0 = carve(max_segs * sizeof(SahSeg)), o_seg1 = carve(max_segs * sizeof(SahSeg)), o_spl = carve(max_segs * (spatial ? sizeof(SpSplit) : sizeof(SahSplit))),
                 o_fill = carve(max_segs * 8u), o_bins = carve(max_segs * (size_t)kSahSegWords * 4u), o_small = carve(max_segs * sizeof(SahSeg));
    const size_t max_tn = spatial ? 8u * ((size_t)cap / (size_t)ret_below) + 4096u : 0u;
    const size_t o_fbox = carve(spatial ? sizeof(Box6) * cap : 0u), o_fprim = carve(spatial ? 4ull * cap : 0u), o_pprim = carve(spatial ? 4ull * cap : 0u),
                 o_pcell = carve(spatial ? sizeof(Box6) * cap : 0u), o_fcell = carve(spatial ? sizeof(Box6) * cap : 0u),
                 o_bud0 = carve(max_segs * 4u * (spatial ? 1u : 0u)), o_bud1 = carve(max_segs * 4u * (spatial ? 1u : 0u)), o_cnt2 = carve(max_segs * 8u * (spatial ? 1u : 0u)),
                 o_tn = carve(max_tn * sizeof(SpNode)), o_ret = carve((max_tn + 1u) * sizeof(SpRet)), o_lnk = carve((2u * max_tn + 2u) * sizeof(SpLink)),
                 o_lf = carve((max_tn + 1u) * sizeof(SpLeaf));
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                      (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 63, st);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)cap - 1, st);
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
    hipLaunchKernelGGL(k_lbvh_prims, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, st, d_world, n, boxes, misc);
    const Box6* item_boxes = boxes;                      // the items' boxes (spatial splits: the references' boxes in their final order)
    const uint32_t* item_prim = nullptr;                 // ... and their triangles
    const Box6* item_cell = nullptr;                     // ... and the cells their boxes were clipped to
    if (spatial) {
        SahSeg* segs[2] = { (SahSeg*)(tmp + o_seg0), (SahSeg*)(tmp + o_seg1) };
        int* budgets[2] = { (int*)(tmp + o_bud0), (int*)(tmp + o_bud1) };
        SpSplit* splits = (SpSplit*)(tmp + o_spl);
        uint32_t* fill = (uint32_t*)(tmp + o_fill); uint32_t* bins = (uint32_t*)(tmp + o_bins); uint32_t* cnt2 = (uint32_t*)(tmp + o_cnt2);
        uint32_t* ids[2] = { vals, vals2 };
        uint32_t* seg_of[2] = { (uint32_t*)keys, (uint32_t*)keys2 };
        uint32_t* retired = idx;                          // (the root list / scan output later)
        uint32_t* pprim = (uint32_t*)(tmp + o_pprim);
        Box6* fbox = (Box6*)(tmp + o_fbox); uint32_t* fprim = (uint32_t*)(tmp + o_fprim);
        Box6* pcell = (Box6*)(tmp + o_pcell); Box6* fcell = (Box6*)(tmp + o_fcell);
        SpNode* tnodes = (SpNode*)(tmp + o_tn); SpRet* rets = (SpRet*)(tmp + o_ret);
        SahSeg* smalls = (SahSeg*)(tmp + o_small);
        uint32_t* counts = misc + 20;                     // [0] / [1]: open ranges of the current / next level; [2]: ranges for k_sah_small
        uint32_t* cursors = misc + 24;                    // [0] next level's buffer, [1] retired buffer, [2] nodes, [3] retired records, [4] the pool of references
        float* root_area = reinterpret_cast<float*>(misc + 30);
        hipLaunchKernelGGL(k_sah_init, dim3(2048), dim3(256), 0, st, bins, max_segs * (size_t)kSahSegWords, ids[0], seg_of[0], cap, left, right, rlo, rhi);
        hipLaunchKernelGGL(k_sp_init, dim3(blocks), dim3(256), 0, st, ids[0], seg_of[0], pprim, pcell, n);
        {
            // the root: the one open range, node 0 of the tree of ranges
            struct { uint32_t counts[4]; uint32_t cursors[6]; } h0 = { { 1u, 0u, 0u, 0u }, { 0u, 0u, 1u, 0u, n, 0u } };
            SP_TRY(hipMemcpyAsync(counts, &h0, sizeof h0, hipMemcpyHostToDevice, st));
            SahSeg r; r.lo = 0; r.hi = (int)n - 1; r.node = 0; r.depth = 0;
            SP_TRY(hipMemcpyAsync(&segs[0][0], &r, 16, hipMemcpyHostToDevice, st));
            SP_TRY(hipMemcpyAsync(segs[0][0].cb, misc, 24, hipMemcpyDeviceToDevice, st));
            const int b0 = (int)sp_budget;
            SP_TRY(hipMemcpyAsync(budgets[0], &b0, 4, hipMemcpyHostToDevice, st));
            const SpNode root_node = { -1, { 0, 0 } };
            SP_TRY(hipMemcpyAsync(tnodes, &root_node, sizeof root_node, hipMemcpyHostToDevice, st));
        }
        uint32_t n_cur = 1u, n_pos = n;
        int cur = 0;
        auto soft_fail = [&]() { (void)hipFree(tmp); *num_nodes = 0; *depth = 255u; return hipSuccess; };      // (ok stays false: the caller builds without spatial splits)
        for (int level = 0; n_cur != 0u; ++level) {
            if (level > 96 || n_cur > max_segs) return soft_fail();
            const uint32_t chunks = (n_pos + (uint32_t)kSahChunk - 1u) / (uint32_t)kSahChunk, cg = chunks < 4096u ? chunks : 4096u;
            hipLaunchKernelGGL(k_sah_bin, dim3(cg), dim3(256), 0, st, ids[cur], seg_of[cur], segs[cur], n_pos, boxes, bins);
            hipLaunchKernelGGL(k_sp_split, dim3(n_cur), dim3(64), 0, st, segs[cur], budgets[cur], counts + cur, bins, splits, cnt2, leaf_max, depth_cap, root_area,
                               [] { const char* c = std::getenv("MIPT_SPLIT_OVERLAP"); return c ? (float)std::atof(c) : 1e-5f; }());     // (experiments: build_sbvh's 1e-5 of the root's area)
            hipLaunchKernelGGL(k_sp_count, dim3(cg), dim3(256), 0, st, ids[cur], seg_of[cur], splits, n_pos, boxes, pprim, d_world, cnt2);
            SP_TRY(hipMemsetAsync(cursors, 0, 4, st));
            hipLaunchKernelGGL(k_sp_alloc, dim3((n_cur + 255u) / 256u), dim3(256), 0, st, splits, counts + cur, cnt2, fill, segs[cur ^ 1], budgets[cur ^ 1], counts + (cur ^ 1),
                               cursors, tnodes, rets, leaf_max, depth_cap, ret_below, (uint32_t)max_segs, (uint32_t)max_tn);
            hipLaunchKernelGGL(k_sp_partition, dim3(cg), dim3(256), 0, st, ids[cur], seg_of[cur], n_pos, splits, ids[cur ^ 1], seg_of[cur ^ 1], retired, fill, boxes, pprim, pcell,
                               d_world, cursors + 4, cap, segs[cur ^ 1], rets);
            uint32_t hc[10];
            SP_TRY(hipMemcpyAsync(hc, counts, sizeof hc, hipMemcpyDeviceToHost, st));
            SP_TRY(hipStreamSynchronize(st));
            SP_TRY(hipMemsetAsync(counts + cur, 0, 4, st));
            n_cur = hc[cur ^ 1]; n_pos = hc[4];
            if (hc[4] > cap || hc[5] > cap || hc[6] > max_tn || hc[7] > max_tn + 1u || hc[8] > cap || n_cur > max_segs) return soft_fail();      // (the budget keeps the first two and the pool inside; the lists: by construction of max_tn, unless the ranges peel off one by one)
            cur ^= 1;
        }
        // the tree of the ranges -> final positions, Karras' numbers, links
        uint32_t hc[10];
        SP_TRY(hipMemcpyAsync(hc, counts, sizeof hc, hipMemcpyDeviceToHost, st));
        SP_TRY(hipStreamSynchronize(st));
        const uint32_t n_tn = hc[6], n_ret = hc[7], n_refs = hc[8];
        if (hc[5] != n_refs || n_ret != n_tn + 1u) return soft_fail();                   // (every reference retired once; a binary tree)
        std::vector<SpNode> tn(n_tn); std::vector<SpRet> rt(n_ret);
        SP_TRY(hipMemcpy(tn.data(), tnodes, (size_t)n_tn * sizeof(SpNode), hipMemcpyDeviceToHost));
        SP_TRY(hipMemcpy(rt.data(), rets, (size_t)n_ret * sizeof(SpRet), hipMemcpyDeviceToHost));
        std::vector<uint32_t> size(n_tn, 0u), lo(n_tn, 0u); std::vector<int> name(n_tn, 0), pname(n_tn, -1), dep(n_tn, 0);
        for (uint32_t i = n_tn; i-- > 0u;)                                                // children have larger numbers than their parents
            for (int side = 0; side < 2; ++side) {
                const int c = tn[i].child[side];
                if (c >= 0 ? ((uint32_t)c >= n_tn || (uint32_t)c <= i) : (uint32_t)~c >= n_ret) return soft_fail();
                size[i] += c >= 0 ? size[(uint32_t)c] : rt[(uint32_t)~c].size;
            }
        if (size[0] != n_refs) return soft_fail();
        std::vector<SpLink> links; std::vector<SpLeaf> leaves; std::vector<SahSeg> small_list;
        links.reserve(2u * n_tn + 2u); leaves.reserve(n_ret);
        for (uint32_t i = 0; i < n_tn; ++i) {
            const int my_lo = (int)lo[i], my_hi = (int)(lo[i] + size[i] - 1u);
            SpLink L; L.node = name[i]; L.parent = pname[i]; L.lo = my_lo; L.hi = my_hi; L.left = 0; L.right = 0;
            uint32_t at = lo[i];
            for (int side = 0; side < 2; ++side) {
                const int c = tn[i].child[side];
                const uint32_t sz = c >= 0 ? size[(uint32_t)c] : rt[(uint32_t)~c].size;
                const int c_lo = (int)at, c_hi = (int)(at + sz - 1u);
                const int c_name = side == 0 ? c_hi : c_lo;                               // a left child is named by its last position, a right child by its first
                int ref = c_name;
                if (c >= 0) { lo[(uint32_t)c] = at; name[(uint32_t)c] = c_name; pname[(uint32_t)c] = name[i]; dep[(uint32_t)c] = dep[i] + 1; }
                else {
                    SpRet& r = rt[(uint32_t)~c];
                    r.final_off = at;
                    if (sz == 1u) { ref = ~c_lo; leaves.push_back({ c_lo, name[i] }); }
                    else {
                        SpLink R; R.node = c_name; R.parent = name[i]; R.lo = c_lo; R.hi = c_hi; R.left = kSpKeep; R.right = kSpKeep;
                        if (sz == 2u) { R.left = ~c_lo; R.right = ~c_hi; leaves.push_back({ c_lo, c_name }); leaves.push_back({ c_hi, c_name }); }
                        links.push_back(R);
                        if (sz > (uint32_t)kTreelet) {
                            SahSeg sg; sg.lo = c_lo; sg.hi = c_hi; sg.node = c_name; sg.depth = dep[i] + 1;
                            for (int q = 0; q < 6; ++q) sg.cb[q] = r.cb[q];
                            small_list.push_back(sg);
                        }
                    }
                }
                if (side == 0) L.left = ref; else L.right = ref;
                at += sz;
            }
            links.push_back(L);
        }
        if (small_list.size() > max_segs || links.size() > 2u * max_tn + 2u || leaves.size() > max_tn + 1u) return soft_fail();
        SpLink* d_links = (SpLink*)(tmp + o_lnk); SpLeaf* d_leaves = (SpLeaf*)(tmp + o_lf);
        SP_TRY(hipMemcpy(rets, rt.data(), (size_t)n_ret * sizeof(SpRet), hipMemcpyHostToDevice));
        SP_TRY(hipMemcpy(d_links, links.data(), links.size() * sizeof(SpLink), hipMemcpyHostToDevice));
        if (!leaves.empty()) SP_TRY(hipMemcpy(d_leaves, leaves.data(), leaves.size() * sizeof(SpLeaf), hipMemcpyHostToDevice));
        if (!small_list.empty()) SP_TRY(hipMemcpy(smalls, small_list.data(), small_list.size() * sizeof(SahSeg), hipMemcpyHostToDevice));
        const uint32_t n_small = (uint32_t)small_list.size();
        SP_TRY(hipMemcpy(counts + 2, &n_small, 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_sp_layout, dim3(n_ret < 4096u ? n_ret : 4096u), dim3(256), 0, st, rets, n_ret, retired, boxes, pprim, pcell, fbox, fprim, fcell, vals2);
        const uint32_t n_l = (uint32_t)(links.size() > leaves.size() ? links.size() : leaves.size());
        hipLaunchKernelGGL(k_sp_links, dim3((n_l + 255u) / 256u), dim3(256), 0, st, d_links, (uint32_t)links.size(), d_leaves, (uint32_t)leaves.size(), left, right, pin, plf, rlo, rhi);
        n = n_refs; n_inner = (int)n - 1; blocks = (n + 255u) / 256u;
        *num_records = n;
        item_boxes = fbox; item_prim = fprim; item_cell = n_refs > n_in ? fcell : nullptr;
        if (n_small != 0u)
            hipLaunchKernelGGL(k_sah_small, dim3(n_small < 4096u ? n_small : 4096u), dim3(256), 0, st, smalls, counts + 2, vals2, item_boxes, leaf_max, depth_cap,
                               left, right, pin, plf, rlo, rhi);
    } else if (top_down) {
        // binned SAH, level by level over all open ranges, until none holds more than kTreelet triangles (the comment at k_sah_bin)
        SahSeg* segs[2] = { (SahSeg*)(tmp + o_seg0), (SahSeg*)(tmp + o_seg1) };
        SahSplit* splits = (SahSplit*)(tmp + o_spl);
        uint32_t* fill = (uint32_t*)(tmp + o_fill); uint32_t* bins = (uint32_t*)(tmp + o_bins);
        uint32_t* perm[2] = { vals, vals2 };
        uint32_t* seg_of[2] = { (uint32_t*)keys, (uint32_t*)keys2 };
        uint32_t* counts = misc + 20;                                   // [0] / [1]: open ranges of the current / next level (alternating); [2]: ranges for k_sah_small
        SahSeg* smalls = (SahSeg*)(tmp + o_small);
        hipLaunchKernelGGL(k_sah_init, dim3(2048), dim3(256), 0, st, bins, max_segs * (size_t)kSahSegWords, perm[0], seg_of[0], n, left, right, rlo, rhi);
        hipLaunchKernelGGL(k_sah_root, dim3(1), dim3(64), 0, st, segs[0], smalls, misc, n, pin, rlo, rhi, counts);
        uint32_t n_cur = n <= (uint32_t)kSahSmall ? 0u : 1u;              // (a small input: no level loop, one workgroup builds it)
        int cur = 0;
        const uint32_t chunks = (n + (uint32_t)kSahChunk - 1u) / (uint32_t)kSahChunk;
        for (int level = 0; n_cur != 0u; ++level) {
            if (level > 96 || n_cur > max_segs) { (void)hipFree(tmp); *num_nodes = 0; *depth = 255u; return hipSuccess; }   // (every level splits every open range in two non-empty
                                                                                                                            //  sides, so this does not happen; if it did: ok stays false, the host builder takes the input)
            hipLaunchKernelGGL(k_sah_bin, dim3(chunks < 4096u ? chunks : 4096u), dim3(256), 0, st, perm[cur], seg_of[cur], segs[cur], n, boxes, bins);
            hipLaunchKernelGGL(k_sah_split, dim3(n_cur), dim3(64), 0, st, segs[cur], counts + cur, bins, splits, fill, segs[cur ^ 1], counts + (cur ^ 1),
                               smalls, counts + 2, leaf_max, depth_cap, left, right, pin, plf, rlo, rhi);
            hipLaunchKernelGGL(k_sah_partition, dim3(chunks < 4096u ? chunks : 4096u), dim3(256), 0, st, perm[cur], seg_of[cur], perm[cur ^ 1], seg_of[cur ^ 1], splits, fill, boxes, n, segs[cur ^ 1], smalls);
            if ((e = hipMemcpyAsync(&n_cur, counts + (cur ^ 1), 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e);
            if ((e = hipMemsetAsync(counts + cur, 0, 4, st)) != hipSuccess) return fail(e);
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e);
            cur ^= 1;
        }
        if (perm[cur] != vals2 && (e = hipMemcpyAsync(vals2, perm[cur], 4ull * n, hipMemcpyDeviceToDevice, st)) != hipSuccess) return fail(e);
        uint32_t n_small = 0u;
        if ((e = hipMemcpyAsync(&n_small, counts + 2, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e);
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e);
        if (n_small > max_segs) return fail(hipErrorUnknown);
        if (n_small != 0u)
            hipLaunchKernelGGL(k_sah_small, dim3(n_small < 4096u ? n_small : 4096u), dim3(256), 0, st, smalls, counts + 2, vals2, boxes, leaf_max, depth_cap,
                               left, right, pin, plf, rlo, rhi);
    } else {
    hipLaunchKernelGGL(k_lbvh_morton, dim3(blocks), dim3(256), 0, st, boxes, n, misc, keys, vals);
    if ((e = hipcub::DeviceRadixSort::SortPairs(tmp + o_cub, sort_bytes, keys, keys2, vals, vals2, (int)n, 0, 63, st)) != hipSuccess) return fail(e);
    hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(blocks), dim3(256), 0, st, keys2, (int)n, left, right, pin, plf, rlo, rhi);
    }
    if (sah_treelets && n > (uint32_t)kTreelet) {
        // the subtrees of at most 64 triangles, rebuilt by SAH (reuses `idx` as the list of their roots until the scan needs it)
        if ((e = hipMemsetAsync(misc + 16, 0, 4, st)) != hipSuccess) return fail(e);
        hipLaunchKernelGGL(k_treelet_roots, dim3(blocks), dim3(256), 0, st, rlo, rhi, pin, n_inner, idx, misc + 16, top_down ? vals2 : nullptr);
        const uint32_t tg = blocks * 4u < 16384u ? blocks * 4u : 16384u;
        hipLaunchKernelGGL(k_lbvh_treelets, dim3(tg), dim3(64), 0, st, idx, misc + 16, item_boxes, vals2, left, right, pin, plf, rlo, rhi, top_down ? 1 : 0);
    }
    const int cluster_levels = sah_treelets && n > (uint32_t)kTreelet * 4u ? [] { const char* e = std::getenv("MIPT_LBVH_CLUSTERS"); return e ? std::atoi(e) : 0; }() : 0;
    auto fit_boxes = [&](uint8_t* kh, uint8_t* root_h) -> hipError_t {
        // boxes bottom-up: heights first (one atomic per step), then one launch per height
        hipError_t fe;
        if ((fe = hipMemsetAsync(arrived, 0, 4ull * n, st)) != hipSuccess) return fe;
        hipLaunchKernelGGL(k_lbvh_heights, dim3(blocks), dim3(256), 0, st, left, right, pin, n_inner, arrived, kh);
        if ((fe = hipMemcpyAsync(root_h, kh, 1, hipMemcpyDeviceToHost, st)) != hipSuccess) return fe;
        if ((fe = hipStreamSynchronize(st)) != hipSuccess) return fe;
        if (*root_h == 0u || *root_h >= 255u) return hipSuccess;
        for (uint32_t level = 1; level <= *root_h; ++level)
            hipLaunchKernelGGL(k_lbvh_fit_level, dim3(blocks), dim3(256), 0, st, item_boxes, vals2, n_inner, left, right, kh, level, nboxes);
        return hipSuccess;
    };
    if (cluster_levels > 0) {
        // experiment MIPT_LBVH_CLUSTERS = passes: the same SAH over clusters of at most 64 items, one pass per level above the treelets
        // (reuses `emit` as the item counters, `idx` as the root list, the height bytes as the `big` flags until the final fit)
        uint8_t* kh = reinterpret_cast<uint8_t*>(tmp + o_kh);
        uint8_t* big = reinterpret_cast<uint8_t*>(tmp + o_big);
        uint8_t root_h = 0;
        if ((e = fit_boxes(kh, &root_h)) != hipSuccess) return fail(e);
        if (root_h != 0u && root_h < 255u) {
            hipLaunchKernelGGL(k_cluster_first_big, dim3(blocks), dim3(256), 0, st, rlo, rhi, n_inner, big);
            for (int pass = 0; pass < cluster_levels && pass < 4; ++pass) {
                if ((e = hipMemsetAsync(emit, 0, 4ull * n, st)) != hipSuccess) return fail(e);
                if ((e = hipMemsetAsync(misc + 17, 0, 4, st)) != hipSuccess) return fail(e);
                hipLaunchKernelGGL(k_cluster_counts, dim3(blocks), dim3(256), 0, st, left, right, big, pin, n_inner, emit);
                hipLaunchKernelGGL(k_cluster_roots, dim3(blocks), dim3(256), 0, st, big, pin, emit, n_inner, idx, misc + 17);
                const uint32_t cg = blocks < 16384u ? blocks : 16384u;
                hipLaunchKernelGGL(k_lbvh_clusters, dim3(cg), dim3(64), 0, st, idx, misc + 17, big, item_boxes, vals2, nboxes, left, right, pin, plf, rlo, rhi);
                hipLaunchKernelGGL(k_cluster_next_big, dim3(blocks), dim3(256), 0, st, emit, n_inner, big);
            }
        }
    }
    {
        uint8_t* kh = reinterpret_cast<uint8_t*>(tmp + o_kh);
        uint8_t root_h = 0;
        if ((e = fit_boxes(kh, &root_h)) != hipSuccess) return fail(e);
        if (root_h == 0u || root_h >= 255u) { (void)hipFree(tmp); *num_nodes = 0; *depth = 255u; return hipSuccess; }      // (degenerate input: the host builder takes it)
    }
    hipLaunchKernelGGL(k_lbvh_flags, dim3(blocks), dim3(256), 0, st, rlo, rhi, n_inner, leaf_max, emit, sah_treelets, left, right, pin, item_boxes, vals2, nboxes,
                       [] { const char* c = std::getenv("MIPT_BVH_NODE_COST"); return c ? (float)std::atof(c) : 1.2f; }());     // (experiments only, as in bvh_build.cpp)
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
    if ((e = hipMalloc((void**)&blob, nb + tb + 64 + 256 + 8192 + (size_t)nn * 96u)) != hipSuccess) return fail(e);     // + room for the packed nodes
    hipLaunchKernelGGL(k_lbvh_emit, dim3(blocks), dim3(256), 0, st, left, right, rlo, rhi, n_inner, leaf_max, emit, idx, item_boxes, vals2, nboxes, (MiptBvhNode*)blob);
    Box6* cells = nullptr;                               // (a tree whose triangles were cut: its records' cells, kept for mipt_accel_update)
    if (item_cell != nullptr && (e = hipMalloc((void**)&cells, (size_t)n * sizeof(Box6))) != hipSuccess) { (void)hipFree(blob); return fail(e); }
    hipLaunchKernelGGL(k_lbvh_tris, dim3(blocks), dim3(256), 0, st, d_world, vals2, d_gflags, n, (MiptBvhTri*)((char*)blob + nb), d_prim_slot, item_prim, item_cell, cells);
    if ((e = hipMemsetAsync((char*)blob + nb + tb, 0, 64, st)) != hipSuccess) { (void)hipFree(blob); if (cells) (void)hipFree(cells); return fail(e); }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) { (void)hipFree(blob); if (cells) (void)hipFree(cells); return fail(e); }
    (void)hipFree(tmp);
    *d_blob = blob; *ok = true; *d_cells_out = reinterpret_cast<float*>(cells);
    return hipSuccess;
}
#undef SP_TRY
}  // namespace

// spatial_extra > 0 (MIPT_OPT_BUILD = 4): references the level loop may add by cutting triangles at split planes, as a fraction of the
// triangle count; ret_below: ranges of at most that many references leave the loop (and are split by objects only from there on).
// *num_records: the triangle records of the blob (> n when triangles were cut); *d_cells: then 6 floats per record (its own allocation, the
// caller's to free): the cell each record's box was clipped to, which lbvh_refit cuts the moved triangles' boxes by.  An input the spatial
// front end gives up on is built without it.
hipError_t lbvh_build(hipStream_t st, const float* d_world, const uint32_t* d_gflags, uint32_t n, int leaf_max, int depth_cap,
                      float4** d_blob, uint32_t* num_nodes, uint32_t* depth, bool* ok, bool sah_treelets, uint32_t* d_prim_slot, bool sah_top,
                      float spatial_extra, int ret_below, uint32_t* num_records, float** d_cells)
{
    uint32_t records_local = n; float* cells_local = nullptr;
    if (num_records == nullptr) num_records = &records_local;
    const bool keep_cells = d_cells != nullptr;
    if (!keep_cells) d_cells = &cells_local;
    hipError_t e = hipSuccess;
    bool done = false;
    if (spatial_extra > 0.0f) {
        e = lbvh_build_impl(st, d_world, d_gflags, n, leaf_max, depth_cap, d_blob, num_nodes, depth, ok, sah_treelets, d_prim_slot, sah_top, spatial_extra, ret_below, num_records, d_cells);
        done = e != hipSuccess || *ok;
    }
    if (!done) e = lbvh_build_impl(st, d_world, d_gflags, n, leaf_max, depth_cap, d_blob, num_nodes, depth, ok, sah_treelets, d_prim_slot, sah_top, 0.0f, ret_below, num_records, d_cells);
    if (!keep_cells && cells_local) (void)hipFree(cells_local);
    return e;
}

}  // namespace mipt
