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
#include <deque>
#include <future>
#include <mutex>
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

// ---- spatial splits (SBVH; Stich, Friedrich, Dietrich 2009): MIPT_OPT_BUILD = 3 -----------------------------------------------------
// The object-split builder above puts every triangle into exactly one leaf, so where triangles are as large as the gaps between them
// (the headline soup: edges of ~0.01 at a spacing of 0.01) sibling boxes overlap all the way down and a ray that enters the cloud walks
// every one of the overlapping subtrees around its first hit -- ~150 node visits for a ray that hits, where the tree is 20 levels deep.
// A SPATIAL split cuts the node's box by a plane instead: a triangle that straddles the plane is referenced on BOTH sides, each side
// with the bounds of the part of the triangle that lies there.  Children then do not overlap (along that axis), at the price of
// duplicated references (capped: `extra` references per triangle in all).  Per node both kinds of split are priced by the same
// surface-area heuristic and the cheaper one is taken; spatial splits are only tried where the object split's children overlap
// noticeably (area of the overlap > 1e-5 of the root's).
// Reference boxes are CONSERVATIVE: the clip of a triangle against a slab is computed in double and rounded outward to float, then
// cut by the reference's own box (which bounds the part of the triangle the ancestors' planes left) -- a point of the triangle inside
// the node's region is inside the box of the reference that region holds, so a ray that meets the triangle there passes that leaf's
// boxes, and hit records stay what a brute-force intersector gives (closest hit: min t, tie -> lowest primitive).
struct Ref { uint32_t prim; Box box; };

inline float f_down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; }
inline float f_up(double x)   { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; }

// bounds of { p in triangle : a <= p[k] <= b }, cut by `within`; false when empty
bool clip_tri(const float* v, int k, double a, double b, const Box& within, Box& out)
{
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    bool any = false;
    auto add = [&](const double* p) { any = true; for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], p[c]); hi[c] = std::max(hi[c], p[c]); } };
    for (int i = 0; i < 3; ++i) {
        const float* pf = v + 3 * i; const float* qf = v + 3 * ((i + 1) % 3);
        const double p[3] = { pf[0], pf[1], pf[2] }, q[3] = { qf[0], qf[1], qf[2] };
        if (p[k] >= a && p[k] <= b) add(p);
        for (int side = 0; side < 2; ++side) {
            const double pl = side == 0 ? a : b;
            if ((p[k] < pl && q[k] > pl) || (p[k] > pl && q[k] < pl)) {
                const double t = (pl - p[k]) / (q[k] - p[k]);
                double x[3] = { p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1]), p[2] + t * (q[2] - p[2]) };
                x[k] = pl;
                add(x);
            }
        }
    }
    if (!any) return false;
    for (int c = 0; c < 3; ++c) {
        out.lo[c] = std::max(within.lo[c], f_down(lo[c]));
        out.hi[c] = std::min(within.hi[c], f_up(hi[c]));
        if (!(out.lo[c] <= out.hi[c])) return false;
    }
    return true;
}

struct SNode { Box box[2]; int32_t child[2]; std::vector<uint32_t> leaf[2]; };      // child >= 0: node index; -1: a leaf of leaf[c]

struct SbvhBuilder {
    const float* verts = nullptr;
    int leaf_max = 4, depth_cap = 32;
    float root_area = 0.0f;
    std::atomic<int> tasks{ 0 };
    int max_tasks = 0;
    std::deque<SNode> nodes;
    std::mutex mu;
    // test / experiment knobs, read from the environment by build_sbvh at every build
    bool force = false, unsplit = true, simple = false;
    uint32_t sp_max_n = 0xffffffffu, sp_min_n = 0u;

    uint64_t capacity(int levels) const { if (levels < 0) return 0; if (levels > 40) levels = 40; return (uint64_t)leaf_max << levels; }
    int alloc() { std::lock_guard<std::mutex> g(mu); nodes.emplace_back(); return (int)nodes.size() - 1; }
    SNode& node(int i) { std::lock_guard<std::mutex> g(mu); return nodes[(size_t)i]; }      // (deque: references stay valid while others append)

    // builds the subtree over `refs` (consumed); returns its node index, or -1 when the references should be one leaf.  `budget`: references
    // the spatial splits of this subtree may add -- handed down in proportion to the children's sizes, so that the tree does not depend on
    // which thread built what
    int build(std::vector<Ref>& refs, int depth, bool force_inner, long long budget)
    {
        const uint32_t n = (uint32_t)refs.size();
        if (!force_inner && n <= 1) return -1;
        Box bounds; bounds.reset(); Box cb; cb.reset();
        for (const Ref& r : refs) {
            bounds.grow(r.box);
            const float c[3] = { 0.5f * (r.box.lo[0] + r.box.hi[0]), 0.5f * (r.box.lo[1] + r.box.hi[1]), 0.5f * (r.box.lo[2] + r.box.hi[2]) };
            cb.grow(c);
        }
        const uint64_t side_cap = capacity(depth_cap - depth - 1);
        // ---- object split: binned SAH over the references' centres (as Builder::sah_split)
        float obj_cost = std::numeric_limits<float>::infinity(); int obj_axis = -1, obj_bin = -1;
        Box obj_l, obj_r; obj_l.reset(); obj_r.reset();
        float scale[3]; bool usable[3];
        for (int k = 0; k < 3; ++k) { const float ext = cb.hi[k] - cb.lo[k]; usable[k] = ext > 0.0f && std::isfinite(ext); scale[k] = usable[k] ? (float)kBins / ext : 0.0f; }
        auto bin_of = [&](const Ref& r, int k) { int b = (int)((0.5f * (r.box.lo[k] + r.box.hi[k]) - cb.lo[k]) * scale[k]); return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b); };
        if (n >= 2) {
            Box bbox[3][kBins]; uint32_t bcnt[3][kBins];
            for (int k = 0; k < 3; ++k) for (int b = 0; b < kBins; ++b) { bbox[k][b].reset(); bcnt[k][b] = 0; }
            for (const Ref& r : refs) for (int k = 0; k < 3; ++k) if (usable[k]) { const int b = bin_of(r, k); bcnt[k][b]++; bbox[k][b].grow(r.box); }
            for (int k = 0; k < 3; ++k) {
                if (!usable[k]) continue;
                Box rbox[kBins]; uint32_t rcnt[kBins];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bbox[k][b]); c += bcnt[k][b]; rbox[b] = acc; rcnt[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bbox[k][b]); c += bcnt[k][b];
                    const uint32_t nl = c, nr = rcnt[b + 1];
                    if (nl == 0 || nr == 0 || nl > side_cap || nr > side_cap) continue;
                    const float cost = acc.half_area() * (float)nl + rbox[b + 1].half_area() * (float)nr;
                    if (cost < obj_cost) { obj_cost = cost; obj_axis = k; obj_bin = b; obj_l = acc; obj_r = rbox[b + 1]; }
                }
            }
        }
        const float area = bounds.half_area();
        if (!force_inner && n <= (uint32_t)leaf_max) {
            if (obj_axis < 0) return -1;
            if (!(area > 0.0f) || obj_cost / area + kNodeCost >= (float)n) return -1;        // keep the leaf unless splitting pays for the extra node
        }
        // ---- spatial split: only where the object split leaves its children overlapping, with references to spare, and depth to spare
        float sp_cost = std::numeric_limits<float>::infinity(); int sp_axis = -1; float sp_plane = 0.0f;
        const uint64_t sp_cap = capacity(depth_cap - depth - 1 - 3);
        // (env MIPT_SBVH_FORCE=1, tests: a valid spatial split is taken whether it is cheaper or not -- trees full of shared references)
        // (env MIPT_SBVH_MAX_N / MIPT_SBVH_MIN_N, experiments: spatial splits only at nodes of that many references)
        bool try_spatial = n > (uint32_t)leaf_max && budget > 0 && (uint64_t)n <= sp_cap && n <= sp_max_n && n >= sp_min_n;
        if (try_spatial && obj_axis >= 0 && !force) {
            Box ov;
            for (int c = 0; c < 3; ++c) { ov.lo[c] = std::max(obj_l.lo[c], obj_r.lo[c]); ov.hi[c] = std::min(obj_l.hi[c], obj_r.hi[c]); }
            const bool overlap = ov.lo[0] <= ov.hi[0] && ov.lo[1] <= ov.hi[1] && ov.lo[2] <= ov.hi[2];
            try_spatial = overlap && ov.half_area() > 1e-5f * root_area;
        }
        // (env MIPT_SBVH_SIMPLE=1, experiment: the only spatial candidate is the object split's own plane, taken whenever it is tried)
        if (try_spatial && simple) {
            if (obj_axis >= 0) { sp_axis = obj_axis; sp_plane = cb.lo[obj_axis] + (float)(obj_bin + 1) / scale[obj_axis]; sp_cost = -1.0f; }
        } else if (try_spatial) {
            for (int k = 0; k < 3; ++k) {
                const double lo = bounds.lo[k], ext = (double)bounds.hi[k] - lo;
                if (!(ext > 0.0) || !std::isfinite(ext)) continue;
                const double width = ext / kBins;
                Box sbox[kBins]; uint32_t enter[kBins], leave[kBins];
                for (int b = 0; b < kBins; ++b) { sbox[b].reset(); enter[b] = 0; leave[b] = 0; }
                auto sbin = [&](float x) { int b = (int)(((double)x - lo) / width); return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b); };
                for (const Ref& r : refs) {
                    const int b0 = sbin(r.box.lo[k]), b1 = sbin(r.box.hi[k]);
                    enter[b0]++; leave[b1]++;
                    if (b0 == b1) { sbox[b0].grow(r.box); continue; }
                    const float* v = verts + (size_t)r.prim * 9;
                    for (int b = b0; b <= b1; ++b) {
                        Box c;
                        if (clip_tri(v, k, lo + width * b, lo + width * (b + 1), r.box, c)) sbox[b].grow(c);
                    }
                }
                Box rbox[kBins]; uint32_t rcnt[kBins];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(sbox[b]); c += leave[b]; rbox[b] = acc; rcnt[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(sbox[b]); c += enter[b];
                    const uint32_t nl = c, nr = rcnt[b + 1];
                    if (nl == 0 || nr == 0 || nl >= n || nr >= n) continue;                 // (a split that leaves a side with everything makes no progress)
                    const float cost = acc.half_area() * (float)nl + rbox[b + 1].half_area() * (float)nr;
                    if (cost < sp_cost) { sp_cost = cost; sp_axis = k; sp_plane = (float)(lo + width * (b + 1)); }
                }
            }
        }
        std::vector<Ref> L, R;
        bool done = false;
        if (sp_axis >= 0 && (sp_cost < obj_cost || force)) {
            const int k = sp_axis;
            long long added = 0;
            // the references that lie on one side go there; the straddlers are clipped, and then each is looked at once more ("reference
            // unsplitting", Stich et al. 4.4): whole on the left, whole on the right, or split -- whichever is cheapest with the sides as they
            // stand (env MIPT_SBVH_UNSPLIT=0: always split)
            struct Straddler { Ref whole, a, b; bool hl, hr; };
            std::vector<Straddler> st;
            Box BL, BR; BL.reset(); BR.reset();
            for (const Ref& r : refs) {
                if (r.box.hi[k] <= sp_plane) { L.push_back(r); BL.grow(r.box); }
                else if (r.box.lo[k] >= sp_plane) { R.push_back(r); BR.grow(r.box); }
                else {
                    const float* v = verts + (size_t)r.prim * 9;
                    Straddler s2; s2.whole = r; s2.a = r; s2.b = r;
                    s2.hl = clip_tri(v, k, -1e300, (double)sp_plane, r.box, s2.a.box); s2.hr = clip_tri(v, k, (double)sp_plane, 1e300, r.box, s2.b.box);
                    if (s2.hl) BL.grow(s2.a.box);
                    if (s2.hr) BR.grow(s2.b.box);
                    st.push_back(s2);
                }
            }
            size_t NL = L.size(), NR = R.size();
            for (const Straddler& s2 : st) { NL += s2.hl ? 1u : 0u; NR += s2.hr ? 1u : 0u; }
            for (const Straddler& s2 : st) {
                if (!s2.hl && !s2.hr) { L.push_back(s2.whole); continue; }                  // (cannot be: the reference's box straddles the plane)
                if (!s2.hl) { R.push_back(s2.b); continue; }
                if (!s2.hr) { L.push_back(s2.a); continue; }
                int choice = 0;                                                             // 0 split, 1 all left, 2 all right
                if (unsplit) {
                    Box bl1 = BL, br2 = BR; bl1.grow(s2.whole.box); br2.grow(s2.whole.box);
                    const float c_split = BL.half_area() * (float)NL + BR.half_area() * (float)NR;
                    const float c_left = bl1.half_area() * (float)NL + BR.half_area() * (float)(NR - 1);
                    const float c_right = BL.half_area() * (float)(NL - 1) + br2.half_area() * (float)NR;
                    if (c_left < c_split && c_left <= c_right && NR > 1) choice = 1;
                    else if (c_right < c_split && NL > 1) choice = 2;
                }
                if (choice == 1)      { L.push_back(s2.whole); BL.grow(s2.whole.box); --NR; }
                else if (choice == 2) { R.push_back(s2.whole); BR.grow(s2.whole.box); --NL; }
                else                  { L.push_back(s2.a); R.push_back(s2.b); ++added; }
            }
            if (!L.empty() && !R.empty() && L.size() < n && R.size() < n && L.size() <= side_cap && R.size() <= side_cap && added <= budget) {
                budget -= added;
                done = true;
            } else { L.clear(); R.clear(); }
        }
        if (!done && obj_axis >= 0) {
            for (const Ref& r : refs) (bin_of(r, obj_axis) <= obj_bin ? L : R).push_back(r);
            done = !L.empty() && !R.empty();
            if (!done) { L.clear(); R.clear(); }
        }
        if (!done) {
            if (n < 2) { L = refs; }                                                        // forced root over 0 / 1 triangles: child 1 empty
            else {
                int ax = 0; float w = -1.0f;
                for (int k = 0; k < 3; ++k) { const float e = cb.hi[k] - cb.lo[k]; if (e > w) { w = e; ax = k; } }
                const size_t mid = n / 2;
                std::nth_element(refs.begin(), refs.begin() + (long)mid, refs.end(), [&](const Ref& a, const Ref& b) {
                    const float ca = a.box.lo[ax] + a.box.hi[ax], cb2 = b.box.lo[ax] + b.box.hi[ax];
                    return ca < cb2 || (ca == cb2 && a.prim < b.prim);
                });
                L.assign(refs.begin(), refs.begin() + (long)mid); R.assign(refs.begin() + (long)mid, refs.end());
            }
        }
        std::vector<Ref>().swap(refs);
        const int me = alloc();
        Box bl, br; bl.reset(); br.reset();
        for (const Ref& r : L) bl.grow(r.box);
        for (const Ref& r : R) br.grow(r.box);
        if (R.empty()) std::memset(&br, 0, sizeof br);
        if (L.empty()) std::memset(&bl, 0, sizeof bl);
        const long long bl_budget = L.size() + R.size() ? (long long)((double)budget * (double)L.size() / (double)(L.size() + R.size())) : 0, br_budget = budget - bl_budget;
        std::future<int> fut; bool spawned = false;
        if (n > 32768 && tasks.load(std::memory_order_relaxed) < max_tasks) {
            tasks.fetch_add(1); spawned = true;
            fut = std::async(std::launch::async, [this, &L, depth, bl_budget] { const int r = build(L, depth + 1, false, bl_budget); tasks.fetch_sub(1); return r; });
        }
        const int cr = build(R, depth + 1, false, br_budget);
        const int cl = spawned ? fut.get() : build(L, depth + 1, false, bl_budget);
        SNode& nd = node(me);
        nd.box[0] = bl; nd.box[1] = br; nd.child[0] = cl; nd.child[1] = cr;
        if (cl < 0) for (const Ref& r : L) nd.leaf[0].push_back(r.prim);                   // (a side that stays a leaf: build() left its references alone)
        if (cr < 0) for (const Ref& r : R) nd.leaf[1].push_back(r.prim);
        return me;
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

// MIPT_OPT_BUILD = 3: binned SAH with spatial splits (the comment at struct Ref).  `extra`: references the spatial splits may add, as a
// fraction of the triangle count.  Leaves of up to 15 references would be possible; the traversal kernels' pool wants at most 4.
bool build_sbvh(const float* world_verts, uint32_t n_tris, int leaf_max, int depth_cap, int n_threads, float extra, HostBvh& out)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (const char* e = std::getenv("MIPT_BVH_NODE_COST")) kNodeCost = (float)std::atof(e);
    if (leaf_max < 1 || leaf_max > 15 || depth_cap < 1 || n_tris >= (1u << 26)) return false;
    SbvhBuilder B;
    B.verts = world_verts; B.leaf_max = leaf_max; B.depth_cap = depth_cap; B.max_tasks = n_threads > 1 ? n_threads - 1 : 0;
    if (const char* e = std::getenv("MIPT_SBVH_FORCE")) B.force = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_SBVH_UNSPLIT")) B.unsplit = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_SBVH_SIMPLE")) B.simple = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIPT_SBVH_MAX_N")) B.sp_max_n = (uint32_t)std::atoll(e);
    if (const char* e = std::getenv("MIPT_SBVH_MIN_N")) B.sp_min_n = (uint32_t)std::atoll(e);
    if ((uint64_t)n_tris > B.capacity(depth_cap)) return false;
    std::vector<Ref> refs(n_tris);
    Box all; all.reset();
    for (uint32_t i = 0; i < n_tris; ++i) {
        const float* v = world_verts + (size_t)i * 9;
        Box b; b.reset(); b.grow(v); b.grow(v + 3); b.grow(v + 6);
        refs[i].prim = i; refs[i].box = b; all.grow(b);
    }
    B.root_area = all.half_area();
    const int root = B.build(refs, 0, /*force_inner=*/true, (long long)((double)n_tris * (double)extra));
    if (root < 0) return false;
    // emission: DFS preorder (a node's first child follows it), triangles in leaf order, duplicates and all
    out.nodes.clear(); out.tris.clear();
    out.nodes.reserve(B.nodes.size()); out.tris.reserve((size_t)n_tris + (size_t)((double)n_tris * extra));
    out.depth = 0; out.leaf_max = (uint32_t)leaf_max;
    struct Item { int src; uint32_t out_index, depth; };
    std::vector<Item> stack;
    out.nodes.emplace_back();
    stack.push_back({ root, 0u, 1u });
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const SNode& sn = B.nodes[(size_t)it.src];
        if (it.depth > out.depth) out.depth = it.depth;
        int32_t refs2[2]; uint32_t child_out[2] = { 0, 0 };
        for (int c = 0; c < 2; ++c) {
            if (sn.child[c] >= 0) { child_out[c] = (uint32_t)out.nodes.size(); out.nodes.emplace_back(); refs2[c] = (int32_t)child_out[c]; }
            else {
                const uint32_t first = (uint32_t)out.tris.size();
                if (sn.leaf[c].size() > 15u || first >= (1u << 27)) return false;
                for (uint32_t p : sn.leaf[c]) {
                    const float* v = world_verts + (size_t)p * 9;
                    MiptBvhTri t{};
                    for (int k = 0; k < 3; ++k) { t.v0[k] = v[k]; t.e1[k] = v[3 + k] - v[k]; t.e2[k] = v[6 + k] - v[k]; }
                    t.prim = p;
                    out.tris.push_back(t);
                }
                refs2[c] = ~(int32_t)((first << 4) | (uint32_t)sn.leaf[c].size());
            }
        }
        MiptBvhNode& nd = out.nodes[it.out_index];
        std::memset(&nd, 0, sizeof nd);
        put_child(nd, 0, sn.box[0], refs2[0]);
        put_child(nd, 1, sn.box[1], refs2[1]);
        if (sn.child[1] >= 0) stack.push_back({ sn.child[1], child_out[1], it.depth + 1 });
        if (sn.child[0] >= 0) stack.push_back({ sn.child[0], child_out[0], it.depth + 1 });
    }
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return out.depth <= (uint32_t)depth_cap;
}

}  // namespace mipt
