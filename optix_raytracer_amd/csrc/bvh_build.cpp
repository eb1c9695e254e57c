// bvh_build.cpp -- binned-SAH BVH2 builder (host, task-parallel), output in the linear layout the
// gfx950 traversal kernels read: 64-byte two-child nodes in DFS order + 48-byte triangles in leaf order.
//
// Replaces the closed-source optixAccelBuild/optixAccelCompact step of buildMeshAccel
// (SDK/optixPathTracer/optixPathTracer.cpp:576-684).  Constraints that come from the kernel:
//   * every root->leaf path crosses at most `depth_cap` inner nodes (the per-lane LDS stack size),
//     enforced while splitting (a split is only taken if both sides still fit below the cap);
//   * leaves hold at most `leaf_max` (<= 15) triangles, reference = ~((first << 4) | count).
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>

namespace mipt {
namespace {

constexpr int   kBins     = 32;
float kNodeCost = 1.2f;             // cost of one more inner node relative to one triangle test

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const Box& o)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); }
    }
    void grow(const float* p)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    }
    float half_area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TempNode {
    Box      box[2];
    uint32_t lo[2], hi[2];      // prim ranges of the two children
    uint32_t slot[2];           // temp slot of an inner child, or UINT32_MAX for a leaf child
};

struct Builder {
    const float*          verts;
    uint32_t              n;
    int                   leaf_max, depth_cap;
    std::vector<Box>      pbox;
    std::vector<float>    cent;       // 3 per prim
    std::vector<uint32_t> idx;
    std::vector<TempNode> temp;       // sparse: subtree over n prims owns n-1 consecutive slots
    std::atomic<int>      tasks{ 0 };
    int                   max_tasks = 1;

    uint64_t capacity(int levels) const   // triangles representable with `levels` more inner levels
    {
        if (levels < 0) return 0;
        if (levels > 40) levels = 40;
        return (uint64_t)leaf_max << levels;
    }

    Box range_box(uint32_t lo, uint32_t hi) const
    {
        Box b; b.reset();
        for (uint32_t i = lo; i < hi; ++i) b.grow(pbox[idx[i]]);
        return b;
    }

    // Finds the best binned-SAH split of [lo,hi) whose two sides both fit under the depth cap.
    // Returns the partition point (lo < mid < hi) or 0 if no admissible binned split exists.
    uint32_t sah_split(uint32_t lo, uint32_t hi, int depth, float& cost_out)
    {
        const uint32_t cnt = hi - lo;
        Box cb; cb.reset();
        for (uint32_t i = lo; i < hi; ++i) cb.grow(&cent[(size_t)idx[i] * 3]);
        float scale[3]; bool usable[3];
        for (int k = 0; k < 3; ++k) {
            const float ext = cb.hi[k] - cb.lo[k];
            usable[k] = ext > 0.0f && std::isfinite(ext);
            scale[k]  = usable[k] ? (float)kBins / ext : 0.0f;
        }
        Box      bbox[3][kBins];
        uint32_t bcnt[3][kBins];
        for (int k = 0; k < 3; ++k) for (int b = 0; b < kBins; ++b) { bbox[k][b].reset(); bcnt[k][b] = 0; }
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t p = idx[i];
            for (int k = 0; k < 3; ++k) {
                if (!usable[k]) continue;
                int b = (int)((cent[(size_t)p * 3 + k] - cb.lo[k]) * scale[k]);
                b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
                bcnt[k][b]++; bbox[k][b].grow(pbox[p]);
            }
        }
        const uint64_t side_cap = capacity(depth_cap - depth - 1);
        float best = std::numeric_limits<float>::infinity(); int best_axis = -1, best_bin = -1;
        for (int k = 0; k < 3; ++k) {
            if (!usable[k]) continue;
            float rarea[kBins]; uint32_t rcnt[kBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bbox[k][b]); c += bcnt[k][b]; rarea[b] = acc.half_area(); rcnt[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bbox[k][b]); c += bcnt[k][b];
                const uint32_t nl = c, nr = rcnt[b + 1];
                if (nl == 0 || nr == 0) continue;
                if (nl > side_cap || nr > side_cap) continue;
                const float cost = acc.half_area() * (float)nl + rarea[b + 1] * (float)nr;
                if (cost < best) { best = cost; best_axis = k; best_bin = b; }
            }
        }
        cost_out = best;
        if (best_axis < 0) return 0;
        const float cl = cb.lo[best_axis], sc = scale[best_axis];
        uint32_t* first = idx.data() + lo;
        uint32_t* last  = idx.data() + hi;
        uint32_t* mid = std::partition(first, last, [&](uint32_t p) {
            int b = (int)((cent[(size_t)p * 3 + best_axis] - cl) * sc);
            b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
            return b <= best_bin;
        });
        const uint32_t m = (uint32_t)(mid - idx.data());
        (void)cnt;
        return (m > lo && m < hi) ? m : 0;
    }

    uint32_t median_split(uint32_t lo, uint32_t hi)
    {
        Box cb; cb.reset();
        for (uint32_t i = lo; i < hi; ++i) cb.grow(&cent[(size_t)idx[i] * 3]);
        int ax = 0; float w = -1.0f;
        for (int k = 0; k < 3; ++k) { const float e = cb.hi[k] - cb.lo[k]; if (e > w) { w = e; ax = k; } }
        const uint32_t mid = lo + (hi - lo) / 2;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](uint32_t a, uint32_t b) {
            const float ca = cent[(size_t)a * 3 + ax], cb2 = cent[(size_t)b * 3 + ax];
            return ca < cb2 || (ca == cb2 && a < b);
        });
        return mid;
    }

    // Builds the subtree of [lo,hi) rooted at temp slot `slot`; returns false when the range should
    // be a leaf instead (then nothing was written).
    bool build(uint32_t lo, uint32_t hi, int depth, uint32_t slot, bool force_inner)
    {
        const uint32_t cnt = hi - lo;
        if (!force_inner && cnt <= 1) return false;
        float cost = 0.0f;
        uint32_t mid = cnt >= 2 ? sah_split(lo, hi, depth, cost) : 0;
        if (!force_inner && cnt <= (uint32_t)leaf_max) {
            if (mid == 0) return false;
            const float area = range_box(lo, hi).half_area();
            // keep the leaf unless splitting pays for the extra node
            if (!(area > 0.0f) || cost / area + kNodeCost >= (float)cnt) return false;
        }
        if (mid == 0) {
            if (cnt >= 2) mid = median_split(lo, hi);
            else          mid = hi;            // forced root over 0/1 triangles: child1 empty
        }
        TempNode& tn = temp[slot];
        tn.lo[0] = lo;  tn.hi[0] = mid;
        tn.lo[1] = mid; tn.hi[1] = hi;
        for (int c = 0; c < 2; ++c) {
            if (tn.hi[c] > tn.lo[c]) tn.box[c] = range_box(tn.lo[c], tn.hi[c]);
            else                     std::memset(&tn.box[c], 0, sizeof(Box));
        }
        const uint32_t nl = mid - lo;
        const uint32_t slot_l = slot + 1, slot_r = slot + (nl > 0 ? nl : 1);
        // left subtree owns slots [slot+1, slot+nl), right [slot+nl, slot+cnt-1)
        std::future<bool> fut;
        bool spawned = false;
        if (cnt > 32768 && tasks.load(std::memory_order_relaxed) < max_tasks) {
            tasks.fetch_add(1);
            spawned = true;
            fut = std::async(std::launch::async, [this, lo, mid, depth, slot_l] {
                const bool r = build(lo, mid, depth + 1, slot_l, false);
                tasks.fetch_sub(1);
                return r;
            });
        }
        const bool inner_r = build(mid, hi, depth + 1, slot_r, false);
        const bool inner_l = spawned ? fut.get() : build(lo, mid, depth + 1, slot_l, false);
        tn.slot[0] = inner_l ? slot_l : UINT32_MAX;
        tn.slot[1] = inner_r ? slot_r : UINT32_MAX;
        return true;
    }
};

void put_child(MiptBvhNode& n, int which, const Box& b, int32_t ref)
{
    if (which == 0) {
        n.c0x[0] = b.lo[0]; n.c0x[1] = b.hi[0]; n.c0y[0] = b.lo[1]; n.c0y[1] = b.hi[1];
        n.c0z[0] = b.lo[2]; n.c0z[1] = b.hi[2]; n.c0 = ref;
    } else {
        n.c1x[0] = b.lo[0]; n.c1x[1] = b.hi[0]; n.c1y[0] = b.lo[1]; n.c1y[1] = b.hi[1];
        n.c1z[0] = b.lo[2]; n.c1z[1] = b.hi[2]; n.c1 = ref;
    }
}

}  // namespace

bool build_bvh(const float* world_verts, uint32_t n_tris, int leaf_max, int depth_cap, int n_threads, HostBvh& out)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (const char* e = std::getenv("MIPT_BVH_NODE_COST")) kNodeCost = (float)std::atof(e);   // experiments only
    if (const char* e = std::getenv("MIPT_BVH_LEAF_MAX")) leaf_max = std::atoi(e);
    if (leaf_max < 1 || leaf_max > 15 || depth_cap < 1) return false;
    if (n_tris >= (1u << 27)) return false;
    Builder B;
    B.verts = world_verts; B.n = n_tris; B.leaf_max = leaf_max; B.depth_cap = depth_cap;
    B.max_tasks = n_threads > 1 ? n_threads - 1 : 0;
    if ((uint64_t)n_tris > B.capacity(depth_cap)) return false;
    B.pbox.resize(n_tris); B.cent.resize((size_t)n_tris * 3); B.idx.resize(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) {
        const float* v = world_verts + (size_t)i * 9;
        Box b; b.reset(); b.grow(v); b.grow(v + 3); b.grow(v + 6);
        B.pbox[i] = b;
        for (int k = 0; k < 3; ++k) B.cent[(size_t)i * 3 + k] = 0.5f * (b.lo[k] + b.hi[k]);
        B.idx[i] = i;
    }
    B.temp.resize(n_tris > 1 ? n_tris : 1);
    B.build(0, n_tris, 0, 0, /*force_inner=*/true);

    // compaction: DFS preorder renumbering, triangles emitted in leaf order
    out.nodes.clear(); out.tris.clear();
    out.nodes.reserve(n_tris / 2 + 1); out.tris.reserve(n_tris);
    out.depth = 0; out.leaf_max = (uint32_t)leaf_max;
    struct Item { uint32_t temp_slot, out_index; uint32_t depth; };
    std::vector<Item> stack;
    out.nodes.emplace_back();
    stack.push_back({ 0u, 0u, 1u });
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const TempNode& tn = B.temp[it.temp_slot];
        if (it.depth > out.depth) out.depth = it.depth;
        int32_t refs[2];
        uint32_t child_out[2] = { 0, 0 };
        for (int c = 0; c < 2; ++c) {
            if (tn.slot[c] != UINT32_MAX) {
                child_out[c] = (uint32_t)out.nodes.size();
                out.nodes.emplace_back();
                refs[c] = (int32_t)child_out[c];
            } else {
                const uint32_t first = (uint32_t)out.tris.size();
                for (uint32_t i = tn.lo[c]; i < tn.hi[c]; ++i) {
                    const uint32_t p = B.idx[i];
                    const float* v = world_verts + (size_t)p * 9;
                    MiptBvhTri t{};
                    for (int k = 0; k < 3; ++k) { t.v0[k] = v[k]; t.e1[k] = v[3 + k] - v[k]; t.e2[k] = v[6 + k] - v[k]; }
                    t.prim = p;
                    out.tris.push_back(t);
                }
                refs[c] = ~(int32_t)((first << 4) | (tn.hi[c] - tn.lo[c]));
            }
        }
        MiptBvhNode& n = out.nodes[it.out_index];
        std::memset(&n, 0, sizeof n);
        put_child(n, 0, tn.box[0], refs[0]);
        put_child(n, 1, tn.box[1], refs[1]);
        // push child 1 first so child 0's subtree follows its parent in memory
        if (tn.slot[1] != UINT32_MAX) stack.push_back({ tn.slot[1], child_out[1], it.depth + 1 });
        if (tn.slot[0] != UINT32_MAX) stack.push_back({ tn.slot[0], child_out[0], it.depth + 1 });
    }
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return out.depth <= (uint32_t)depth_cap;
}

}  // namespace mipt
