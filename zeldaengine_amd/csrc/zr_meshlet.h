// zr_meshlet.h — meshlet clusteriser + bounds (host side).
#pragma once

#include <stdint.h>
#include <vector>
#include "../../include/zelda_abi.h"

struct ZrMeshletSet {
    std::vector<XkMeshlet> meshlets;     // BindlessContext = number of triangles in earlier meshlets (tri_base)
    std::vector<uint32_t>  mverts;
    std::vector<uint8_t>   mtris;        // 3 B per triangle, tightly packed
    std::vector<uint32_t>  tri_order;    // [tri_base + t] -> triangle index in the mesh's draw-order index buffer
};

// Bounding sphere + normal cone of one meshlet (meshopt_computeMeshletBounds' published definition, ZM:151).
void zr_meshlet_bounds(const XkVertex* verts, const uint32_t* mv, uint32_t nv, const uint8_t* mt, uint32_t nt, XkMeshlet* out);
// Greedy adjacency clusteriser (BuildMeshlets, ZM:132-172: 64 vertices / 124 triangles / cone weight 0.2).
void zr_build_meshlets(const XkVertex* verts, uint32_t nv, const uint32_t* idx, uint32_t ni,
                       uint32_t max_vertices, uint32_t max_triangles, float cone_weight, ZrMeshletSet* out);
