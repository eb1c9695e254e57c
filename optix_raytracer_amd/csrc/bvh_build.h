// bvh_build.h -- host BVH2 builder of libmipt (replaces the driver-internal optixAccelBuild /
// optixAccelCompact used by buildMeshAccel, SDK/optixPathTracer/optixPathTracer.cpp:576-684).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/mipt.h"

namespace mipt {

struct HostBvh {
    std::vector<MiptBvhNode> nodes;      // DFS order, node 0 = root (always an inner node)
    std::vector<MiptBvhTri>  tris;       // leaf order
    uint32_t depth = 0;                  // max number of inner nodes on a root->leaf path
    uint32_t leaf_max = 0;
    double   seconds = 0.0;
};

// world_verts: n_tris * 9 floats (v0 v1 v2 per triangle, already in world space).
// depth_cap: traversal stack entries available per ray (kStackEntries).
// Returns false if n_tris cannot be represented within depth_cap.
bool build_bvh(const float* world_verts, uint32_t n_tris, int leaf_max, int depth_cap, int n_threads, HostBvh& out);

// The same with spatial splits (SBVH): triangles that straddle a split plane are referenced on both sides, so `tris` holds more records than
// there are triangles (at most n_tris * (1 + extra)); false also when the tree is deeper than depth_cap.
bool build_sbvh(const float* world_verts, uint32_t n_tris, int leaf_max, int depth_cap, int n_threads, float extra, HostBvh& out);

}  // namespace mipt
