// zr_types.h — device-resident scene and per-pass parameter blocks (HBM layout).
#pragma once

#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/zelda_abi.h"
#include "../../include/zelda_render.h"

#define ZR_EMPTY_PRIM 0xFFFFFFFFu
#define ZR_GUARD 4.0f                       // guard band, multiples of w
#define ZR_RECT_CULLED 0xFFFFFFFFu
#ifndef ZR_CHUNK
#define ZR_CHUNK 64u                         // bin entries (meshlet-instances) per raster work unit
#endif

enum { ZR_MODE_GBUFFER = 0, ZR_MODE_SHADOW = 1 };
#define ZR_OBJ_SKY 1u                        // the skydome draw: camera pass only, unlit, written to the overlay plane

// Per-instance transform, prepared once at zr_object_add from XkInstanceData (32 B -> 64 B):
// R = mat3(MakeRotMatrix(InstanceRotation)) column-major, t = InstancePosition, s = InstancePScale.
struct ZrInstance {
    float R[9];
    float t[3];
    float s;
    float _pad[3];
};
static_assert(sizeof(ZrInstance) == 64, "ZrInstance");

// Vertex record of the GBuffer resolve: two aligned 16-byte loads per corner instead of the eleven dwords of an XkVertex.
// nx..nz hold normalize(XkVertex::Normal), formed once on the host with the kernels' own zr_normalize (Base.vert:29 normalises the
// attribute before anything else, so the value is a per-vertex constant).
struct ZrRVertex { float px, py, pz, u; float nx, ny, nz, v; };
static_assert(sizeof(ZrRVertex) == 32, "ZrRVertex");

// One material texture: RGBA8 mip chain, level l = max(1, w >> l) x max(1, h >> l) texels, levels concatenated.
struct ZrTex { const uint8_t* data; uint32_t w, h, levels, _pad; };

// One draw (object) of the scene, in the reference's draw order (non-instanced draws first, ZE:3445-3476).
struct ZrObject {
    const XkVertex*   verts;
    const ZrRVertex*  rverts;        // the same vertices repacked for the resolve (see ZrRVertex)
    const ZrRVertex*  rtris;         // ... and once more per triangle corner in draw order: rtris[3 * tri + k] = rverts[indices[3 * tri + k]]
    const uint32_t*   indices;       // draw-order index buffer (3 per triangle)
    const XkMeshlet*  meshlets;      // device copy; BindlessContext = tri_base (triangles in earlier meshlets)
    const float4*     mpos;          // flattened meshlet vertices: mpos[VertexOffset + k] = position of meshlet vertex k
                                     // (CreateMeshVertexBuffers<XkMeshIndirect> flattens the same way, ZE:4733-4756)
    const float4*     mbox;          // per meshlet: object-space box of its vertices, [2 m] = least, [2 m + 1] = greatest corner
    const uint32_t*   tri_meshlet;   // draw-order triangle -> meshlet index (visibility history for the Hi-Z pass)
    const uint2*      mtri;          // per meshlet triangle slot (tri_base + t): x = corners i0 | i1 << 8 | i2 << 16,
                                     // y = draw-order triangle index (primitive id within the instance)
    const ZrInstance* inst;
    uint32_t n_meshlets, n_tris, n_inst, instanced;
    uint32_t work_base;              // first meshlet-instance id of this draw
    uint32_t prim_base;              // first primitive id of this draw
    uint32_t inst_base;              // instances in earlier draws (global instance ordinal of instance 0)
    ZrTex    tex[7];                 // sampled material slots (data == nullptr: the slot is the constant `texel`)
    ZrTex    packed;                 // the image slots of ONE size interleaved, 16 B per texel (see ZR_PK_*): what BaseScene.frag reads of
                                     // a texel of all seven slots comes with one load.  nullptr: no image, or images of different sizes
    uint32_t texel[7];               // constant material: RGBA8 per PBR slot (bc, m, r, n, ao, ev, ms)
    float    bc_linear[3];           // sRGB-decoded base colour (slot 0 is R8G8B8A8_SRGB, ZE:5878)
    float    texc[7][4];             // the constant texels decoded on the host (slot 0 rgb through the sRGB table, the rest c / 255)
    float    mesh_center[3];         // object-space bounding sphere of the whole mesh
    float    mesh_radius;
    uint32_t flags;                  // ZR_OBJ_*
    uint32_t const_slots;            // bit t: material slot t is a constant texel (no image)
    // What BaseScene.frag makes of constant slots, formed once on the host with the kernels' own arithmetic:
    float    ts_const[3];            // normalize(2 * normalize(texNormal) - 1) when slot 3 is constant (ComputeNormal, SH/Common.glsl:125-126)
    uint32_t c_scene_color, c_gB, c_gC;   // the packed SceneColor / GBufferB / GBufferC words when their slots (5,6 / 1,2 / 0,4) are constant
};

// byte of a packed material texel that holds a channel BaseScene.frag uses (:30-48): base colour rgb, metallic r, roughness r, normal rgb,
// ambient occlusion r, emissive rgb, mask r; bytes 13..15 are zero
#define ZR_PK_BC 0
#define ZR_PK_ME 3
#define ZR_PK_RO 4
#define ZR_PK_NO 5
#define ZR_PK_AO 8
#define ZR_PK_EM 9
#define ZR_PK_MS 12
#define ZR_PK_CHANNELS 13

// One (tile, meshlet-instance) entry of a bin list, self-contained: the rasteriser starts every load of a meshlet from
// this record alone (one dependent round trip instead of bins -> draw table -> meshlet -> vertices).
struct ZrBinEntry {
    const float4*     mpos;          // first flattened vertex of the meshlet
    const uint2*      mtri;          // first triangle word of the meshlet
    const ZrInstance* inst;          // the instance record
    uint32_t          counts;        // VertexCount | TriangleCount << 8 | instanced << 16
    uint32_t          prim_base;     // primitive id of the instance's first triangle
};
static_assert(sizeof(ZrBinEntry) == 32, "ZrBinEntry");

// Parameters of one geometry pass (camera or shadow), passed by value in the kernarg segment.
struct ZrPass {
    float PVM[16];                   // proj * view * model
    float M[16];                     // model (world) matrix
    float planes[6][4];              // world-space frustum planes, inward, normalised xyz
    float cam_pos[3];                // world-space eye (cone test); valid when cone_ok
    float m_scale;                   // upper bound of |M x| / |x| (Frobenius norm of mat3(M))
    float hw, hh;                    // half extent of the target in pixels
    uint32_t W, H;                   // target extent
    uint32_t tiles_x, tiles_y;
    uint32_t tile_rank, tile_world;  // this device owns the tiles whose zr_tile_owner(tx, ty, tile_world) == tile_rank (super-tiles of
                                     // (1 << ZR_SUPERTILE_SHIFT)^2 tiles, skewed round-robin: see zelda_render.h)
    uint32_t rect_cull;              // camera pass, tile_world > 1: reject meshlets (instances) whose bounding sphere cannot reach an owned
                                     // tile, before any vertex is transformed (needs the engine's centred perspective and rigid view * model)
    float    VM[16];                 // view * model (rect_cull)
    float    p00, p11;               // Proj[0][0], Proj[1][1] (rect_cull)
    uint32_t inst_rank, inst_world;  // shadow pass only: this device draws instances with i % inst_world == inst_rank
    uint32_t n_objects, n_work;
    uint32_t n_inst_total;           // instances over all draws
    uint32_t use_worklist;           // 1: instance-level pre-cull compacts the work items into work[]; 0: work item k = k
    uint32_t mode;                   // ZR_MODE_*
    uint32_t frustum_ok, cone_ok;    // culling enabled (cone_ok also needs a standard perspective eye)
    uint32_t bin_capacity;
    uint32_t images;                 // some material slot (or the skydome) holds an image: the resolve needs the texture filter
    uint32_t debug_skip;             // diagnostics only (env ZR_DEBUG_SKIP): 1 skip pixel walk, 2 skip triangle phase too
    uint32_t sphere_ok;              // VM / p00 / p11 / pz_* hold a centred perspective with rigid model and view: sphere_bounds() applies
    float    pz_a, pz_b;             //   ndc depth of a point d in front of the eye = pz_a + pz_b / d
    uint32_t m_identity;             // M is bit for bit the identity: M * vec4(p, 1) == p + 0.0f for finite p
    uint32_t write_overlay;          // the resolve must write the overlay plane (a skydome is drawn, or stale sky pixels must go)
    const unsigned long long* sky_keys;   // the skydome's own key plane (k_sky_tiles) or nullptr: a dome pixel shows where its depth is LESS
    uint32_t sky_object;             //   than the scene's; index of the dome's record in the draw table
};

// Frame statistics block in device memory (one per pass slot: [shadow, camera]).
// what ran full (ZrDevStats::overflow_sticky; zr_finish names it)
#define ZR_OVF_BINS 1u            // the shadow pass's (tile, meshlet) bin entries
#define ZR_OVF_SLOW 2u            // a list of clipped / long triangles
#define ZR_OVF_UNITS 3u           // the camera pass's work-unit table
#define ZR_OVF_RECORDS 4u         // the camera pass's triangle-record arrays (a section of the overflow region)
#define ZR_OVF_LATE 5u            // the shadow pass's late bin entries
struct ZrDevStats {
    uint32_t survivors[3];           // slots: 0 shadow pass, 1 camera pass (round 1), 2 camera pass round 2 (after Hi-Z)
    uint32_t bin_entries[3];
    uint32_t covered_part[32];       // covered pixels, in 32 partial sums (the resolve adds to word blockIdx & 31)
    uint32_t covered_shadow;
    uint32_t overflow;
    uint32_t n_chunks[3];
    uint32_t chunk_counter[3];
    uint32_t hiz_culled;             // meshlet-instances rejected by the Hi-Z test
    uint32_t n_sel[3];               // triangle-binned camera pass: meshlet-instances selected for a round (slots as above)
    uint32_t n_slow[3];              //   triangles of the round that need the clipper / the 64-bit walk
    uint32_t pool_next[3], pool_used[3];   // pool_used[slot]: records of the round that did not fit their tile's bucket (overflow region); pool_next: unused
    uint32_t shadow_occluded;        // shadow pass: meshlet-instances left out because the map's depths already hide them (host copy: the sum of
                                     //   the 32 partial sums k_shadow_occlusion keeps in the shadow pipeline's covered_part)
    uint32_t shadow_late;            //   ... and drawn in the late launch (not drawn last frame, not hidden this frame)
    uint32_t hiz_culled_geom;        // of hiz_culled: rejected by k_geom (exact vertex box / every triangle hidden), i.e. AFTER a wave transformed
                                     //   the meshlet's vertices; the rest fell to k_select's bounds before any vertex work
    uint32_t overflow_sticky;        // from here on: NOT cleared at frame begin.  Set with `overflow` (to the ZR_OVF_* code of what ran full), cleared by
                                     // zr_finish when it reports it
    uint32_t n_vis_work[2];          // meshlet-instances on the pass's work list (k_cull_instances); the list and its length stand while the
                                     // pass's matrices and the scene do - k_frame_begin zeroes a slot when the host is about to rebuild it
};

// Two-pass Hi-Z occlusion culling of the camera pass (config 5; conservative).
// Hi-Z level l holds, per (8 << l) x (8 << l) pixel block, the MAX depth currently in the key buffer (1.0 where empty).
struct ZrHiz {
    float*    lvl[4];                // device arrays, level l is hw[l] x hh[l]: max depth per (8 << l)^2 pixel block
    uint32_t  hw[4], hh[4];
    float*    fine; uint32_t fw, fh; // one level below: max depth per 4 x 4 pixel block (small meshlets and single triangles are tested here)
    uint2*    pxrect;                // per work item: snapped pixel bbox (x0 | y0 << 16, x1 | y1 << 16)
    float*    zmin;                  // per work item: least NDC depth of the meshlet's vertices, < 0: do not occlusion-test
    const uint8_t* vis_prev;         // per meshlet-instance: == vis_stamp: owned a pixel of the previous frame
    uint8_t*  vis_now;               // marked by the resolve with THIS frame's stamp (1 + frame % 255: the marks of older frames need no clearing;
    uint32_t  vis_stamp;             //   one 510 frames old reads as "visible" once, which only moves a meshlet between the rounds)
    uint32_t  phase;                 // 0: no Hi-Z (one round); 1: round 1 = last frame's visible set; 2: round 2 = the rest, Hi-Z tested
    uint32_t  tiles_x, tile_rank, tile_world;   // pyramid texels over another rank's tiles read 0 ("hidden"): nothing is drawn there
};

// Uniforms of the lighting pass that are not in XkView.
struct ZrLightParams {
    float SB[16];                    // BiasMat * shadowmapSpace (SH/Common.glsl:294-304), folded on the host
    uint32_t W, H, SD;
    uint32_t tiles_x;
    uint32_t debug_view;
    uint32_t cube_dim, cube_levels;
    uint32_t packed_out;             // 1: write tile-major packed output (multi-GPU), 0: row-major frame
    uint32_t tile_world;
    uint32_t debug_skip;             // diagnostics only (env ZR_DEBUG_SKIP_LIGHT bits: 1 PCF, 2 lights, 4 reflection)
    uint32_t bg_enabled;             // background quad (Background.vert/.frag) on
    uint32_t light_list;             // 1: per-tile point-light lists
    uint32_t has_overlay;            // the overlay plane may hold skydome pixels (else it is all zero and is not read)
    const uint32_t* empty_rgba;      // the lit colour of a pixel holding the GBuffer's clear values (same for all of them), or null
    uint32_t* clear_next;            // the NEXT frame's shadow map (idle while this pass runs): cleared to depth 1.0 here, or null
    uint32_t clear_n, _pad0;
    ZrTex    bg;                     // its sRGB texture
};

// SoA GBuffer planes in HBM, row-major W x H each (formats ZE:2807-2843): D32F, RGBA8, A2R10G10B10, RGBA8, RGBA8, RGBA16F.
struct GBufferPtrs {
    float* depth; uint32_t* scene_color; uint32_t* gA; uint32_t* gB; uint32_t* gC; uint2* gD;
    uint32_t* overlay;               // RGBA8 of the skydome pass (0 = nothing drawn); not a GBuffer attachment
    uint32_t* prim;                  // forward variant only (else nullptr): the primitive id that won the pixel, ZR_EMPTY_PRIM where nothing was drawn
};
// Cubemap mip chain, level l = 6 faces of (dim >> l)^2 RGBA8 sRGB texels, face-major.
struct CubeDesc { const uint8_t* levels[16]; };

// launchers: each defined in the .hip of its pass (zr_cull / zr_shadow / zr_camera / zr_resolve / zr_lighting / zr_forward / zr_frame)
void zr_launch_instance_prep(const XkInstanceData* in, ZrInstance* out, uint32_t n, uint32_t instanced, hipStream_t s);
void zr_launch_cull(const ZrPass& P, const ZrObject* objs, uint32_t* work, uint32_t* rects, const ZrHiz& Z, ZrDevStats* stats,
                    int slot, uint32_t n_waves, hipStream_t s);
void zr_launch_bin_count(const ZrPass& P, const uint32_t* work, uint32_t* rects, uint32_t* tile_count, const ZrHiz& Z, ZrDevStats* stats,
                         int slot, hipStream_t s);
void zr_launch_hiz_build(const unsigned long long* vis64, uint32_t W, uint32_t H, const ZrHiz& Z, const uint32_t* regions, uint32_t n_regions, hipStream_t s);
void zr_launch_scan(uint32_t* tile_count, uint32_t* tile_offset, uint32_t* tile_cursor, uint32_t* chunk_offset, uint4* chunk_tab,
                    uint32_t chunk_cap, uint32_t n, uint32_t capacity, ZrDevStats* stats, int slot, hipStream_t s,
                    uint32_t chunk = ZR_CHUNK);
// triangle-binned camera pass, per round: [k_select ->] k_geom -> k_tile; once per frame, after the resolve: k_plan (the next frame's buckets)
struct ZrTriBins {
    ZrBinEntry* sel;                 // meshlet-instances of this round, as self-contained 32-byte records
    // 32-byte triangle records (see zr_camera.hip, "triangle records") in two 16-byte planes:
    uint4*    recA; uint4* recB;     //   (X0|Y0, z0, X1|Y1, z1)  (X2|Y2, z2, prim, 0), tile-relative int16 coordinates
    uint32_t* tile_base;             // per tile: its bucket [tile_base, tile_base + tile_cap) of the planes, planned by k_plan from the previous
    uint32_t* tile_cap;              //   frame's counts (the buckets take at most bucket_max of the planes' n_rec records)
    uint32_t* cursor;                // [2][n_tiles * ZR_TSTRIDE]: records appended to each tile in round 1 / round 2 of the frame (k_geom; zeroed by k_plan)
    uint32_t  n_rec, bucket_max;
    uint32_t* plan;                  // [2] (k_plan): where this frame's buckets end = where its overflow region begins; records per section of it
    uint32_t* over_tile;             // [n_rec] the tile of each record of the overflow region (what did not fit its bucket, any tile's)
    uint32_t* over_cursor;           // [2][ZR_OVER_SECTIONS]: overflow records of round 1 / round 2, per section of the region (section = tile % ZR_OVER_SECTIONS)
    uint4*    unit_tab;              // k_tile's work units (tile, part, parts of the tile, 0), planned with the buckets
    uint32_t  unit_cap; uint32_t* n_units;
    uint32_t  n_tiles;
    uint32_t  n_waves;               // waves of the k_geom grid
    uint32_t* wave_culled;           // per wave: meshlets it dropped behind the Hi-Z pyramid (round 2)
    uint4*    slow; uint32_t slow_cap;      // 4 x uint4 per slow triangle: three clip-space vertices, (prim, tile rect, 0, 0)
};
#define ZR_OVER_SECTIONS 64u          // sections of the overflow region (a power of two)
#ifndef ZR_TSTRIDE
#define ZR_TSTRIDE 4u                 // words between the per-tile record cursors of neighbouring tiles
#endif
#ifdef ZR_TCHUNK_AB
#define ZR_TCHUNK ZR_TCHUNK_AB
#else
#define ZR_TCHUNK 512u               // triangle records per BATCH of the tile kernel (gathered, sorted and walked together)
#endif
#ifndef ZR_TBATCHES
#define ZR_TBATCHES 4u               // batches per work unit: a unit is <= ZR_TCHUNK * ZR_TBATCHES records of ONE tile, walked into the same LDS keys,
#endif                               // which are cleared and merged into the key buffer once per unit (1 / 2 / 4 / 8: 5 399 / 5 414 / 5 455 / 5 442 Mpixel/s)
void zr_launch_select(const ZrPass& P, const ZrObject* objs, const uint32_t* work, const uint32_t* rects, const ZrHiz& Z, const ZrTriBins& B, ZrDevStats* stats,
                      int slot, hipStream_t s);
void zr_launch_geom(const ZrPass& P, const ZrHiz& Z, const ZrTriBins& B, ZrDevStats* stats, int slot, bool count_only, hipStream_t s);
void zr_launch_plan(const ZrTriBins& B, const uint32_t* owned_tiles, uint32_t n_owned, ZrDevStats* stats, bool exact, uint32_t bucket_pct, hipStream_t s);
void zr_launch_tile(const ZrPass& P, const ZrTriBins& B, ZrDevStats* stats, int slot, unsigned long long* vis64, uint32_t n_blocks, hipStream_t s, bool last,
                    const uint32_t* owned_tiles, uint32_t n_owned);
void zr_launch_cull_box(const ZrPass& P, const ZrObject* objs, uint32_t* work, uint32_t* rects, const ZrHiz& Z, ZrDevStats* stats,
                        int slot, hipStream_t s, ZrBinEntry* sel = nullptr, const uint8_t* vis_prev = nullptr, bool reuse_list = false);      // sel: round 1's list (camera)
void zr_launch_bin_fill(const ZrPass& P, const ZrObject* objs, const uint32_t* work, const uint32_t* rects, const uint32_t* tile_offset,
                        uint32_t* tile_cursor, ZrBinEntry* bins, const ZrHiz& Z, ZrDevStats* stats, int slot, hipStream_t s);
void zr_launch_frame_begin(ZrDevStats* stats, const XkView* view_src_pinned, XkView* view_dst, uint32_t rebuild_lists, hipStream_t s);
void zr_launch_fill32(uint32_t* p, uint32_t v, size_t n, hipStream_t s);
void zr_launch_fill64(unsigned long long* p, unsigned long long v, size_t n, hipStream_t s);
// shadow pass with occlusion culling: after the rasteriser has drawn the meshlet-instances flagged in `flags`, test every survivor of the
// cull against the map, rewrite the flags, list the unflagged ones that are not hidden (from the top of `bins` downwards)
void zr_launch_shadow_occlusion(const ZrPass& P, const ZrObject* objs, const uint32_t* work, const uint32_t* rects, const uint2* pxrect,
                                const float* zmin, uint8_t* flags, const uint32_t* shadow_bits, ZrBinEntry* bins, ZrDevStats* stats,
                                uint32_t n_blocks, uint32_t retest, hipStream_t s);
void zr_launch_raster_chunks(const ZrPass& P, const ZrObject* objs, const uint4* chunk_tab,
                             const ZrBinEntry* bins, ZrDevStats* stats, int slot, unsigned long long* vis64, uint32_t* shadow_bits,
                             uint32_t n_blocks, const ZrHiz& Z, hipStream_t s, uint4* slow = nullptr, uint32_t slow_cap = 0, const uint32_t* tiles = nullptr,
                             uint32_t n_tiles = 0, int stage = 0);      // slow != nullptr (shadow pass): clipped triangles via the list + k_tile_slow
                                                                        // stage 1: the flagged share only (k_tile_slow waits); 2: the late list, then k_tile_slow
void zr_launch_resolve_gbuffer(const ZrPass& P, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned,
                               unsigned long long* vis64, const GBufferPtrs& G, const float* srgb_lut, const float* unorm_lut, uint8_t* vis_now,
                               ZrDevStats* stats, hipStream_t s, uint32_t vis_mark = 1u);
void zr_launch_sky_tiles(const ZrPass& P, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned, unsigned long long* sky64, hipStream_t s);
void zr_launch_count_shadow(const uint32_t* bits, size_t n, ZrDevStats* stats, hipStream_t s);
void zr_launch_lighting(const ZrLightParams& L, const XkView* view, const uint32_t* owned_tiles, uint32_t n_owned,
                        const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C, const float* lut, const float* unorm_lut,
                        uint32_t* out, hipStream_t s);
// forward variant (SH/Base.frag): shades the winners recorded in G.prim with the camera pass's own parameter block
void zr_launch_forward(const ZrPass& P, const ZrLightParams& L, const XkView* view, const ZrObject* objs, const uint32_t* owned_tiles, uint32_t n_owned,
                       const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C, const float* lut, const float* unorm_lut, uint32_t* out, hipStream_t s);
void zr_launch_gbuffer_vis(const ZrLightParams& L, const XkView* view, const GBufferPtrs& G, const float* shadowmap, const CubeDesc& C,
                           const float* lut, uint32_t* out, hipStream_t s);
void zr_launch_untile(const uint32_t* gathered, const uint32_t* tile_map, uint32_t* frame, uint32_t W, uint32_t H, uint32_t tiles_x,
                      uint32_t n_tiles, hipStream_t s);
void zr_launch_pack_tiles(const uint32_t* plane, const uint32_t* tiles, uint32_t n_tiles, uint32_t* packed, uint32_t W, uint32_t H, uint32_t tiles_x,
                          uint32_t pad, hipStream_t s);
