/*
 * zelda_render.h — C-ABI of the MI355X-native deferred renderer (libzelda_render.so).
 *
 * The reference (iceprincefounder/ZeldaEngine) is a monolith with no plugin/FFI
 * boundary: its renderer is the private body of XkZeldaEngineApp.  The entry
 * points below are therefore the seam a maintainer would cut: each one replaces
 * the engine-internal function cited next to it and takes the engine's own byte
 * layouts (zelda_abi.h).  See INTEGRATION.md for the binding stubs.
 *
 * Conventions: 0 = OK, negative = error (message via zr_last_error); no C++
 * exception crosses the ABI; the caller owns every input pointer (contents are
 * copied before return); outputs are written into caller buffers.  A context is
 * bound to one HIP device and is NOT thread-safe, except zr_livelink_* which
 * hand a parsed world to the render thread under a mutex.
 *
 * There is no CPU fallback: every entry point that computes fails with
 * ZR_ERR_DEVICE when no HIP device is usable.
 */
#ifndef ZELDA_RENDER_H
#define ZELDA_RENDER_H

#include "zelda_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ZR_OK            0
#define ZR_ERR_ARG      -1
#define ZR_ERR_DEVICE   -2
#define ZR_ERR_OOM      -3
#define ZR_ERR_PARSE    -4
#define ZR_ERR_IO       -5
#define ZR_ERR_STATE    -6
#define ZR_ERR_OVERFLOW -7
#define ZR_ERR_UNSUPPORTED -8

/* Version of this header's structs and entry points.  A host checks zr_abi_version() == ZR_ABI_VERSION after loading the library
 * (INTEGRATION.md); structs that may grow (zr_stats) are passed with their size as the caller knows it and only ever grow at the end. */
#define ZR_ABI_VERSION 6u

#ifndef ZR_TILE
#define ZR_TILE 32            /* screen tile edge in pixels (raster + multi-GPU partition unit); 32 or 64 */
#endif

/* Multi-GPU screen partition: tiles are owned in super-tiles of (1 << ZR_SUPERTILE_SHIFT)^2 tiles (128 x 128 pixels), dealt to the
 * ranks round-robin along x with a skew per super-tile row, so that a meshlet (tens of pixels) nearly always falls to ONE rank and
 * that rank alone transforms it, while neighbouring super-tiles still go to different ranks. */
#ifndef ZR_SUPERTILE_SHIFT
#define ZR_SUPERTILE_SHIFT 2
#endif
#define ZR_SUPERTILE_SKEW  3

typedef struct zr_ctx zr_ctx;

/* Flags for zr_config.flags */
#define ZR_FLAG_NO_FRUSTUM_CULL 1u  /* disable meshlet frustum culling (parity A/B) */
#define ZR_FLAG_NO_CONE_CULL    2u  /* disable meshlet cone culling   (parity A/B) */
#define ZR_FLAG_SKIP_COMPOSITE  4u  /* tile_world>1: caller gathers packed tiles itself */
#define ZR_FLAG_NO_HIZ          8u  /* disable two-pass Hi-Z occlusion culling of the camera pass (parity A/B) */
#define ZR_FLAG_SERIAL_PASSES   16u /* zr_render: shadow and camera pipelines on the one stream instead of side by side */
#define ZR_FLAG_PACKED_TILES    32u /* tile_world == 1: still light into the packed tile buffer (the multi-GPU data path on one GPU) */
#define ZR_FLAG_MESHLET_BINS   128u /* -DZR_DIAG builds only (zr_create: ZR_ERR_UNSUPPORTED otherwise): camera pass through the meshlet-binned
                                     * rasteriser the shadow pass uses, instead of the triangle-binned one (A/B measurements) */
#define ZR_FLAG_NO_RECT_CULL    64u /* tile_world > 1: do not reject meshlets by the rank's owned screen region before stage B (parity A/B) */
#define ZR_FLAG_NO_LIST_REUSE  256u /* rebuild the passes' instance-level work lists every frame instead of only when camera / light matrices or the
                                     * scene change (parity A/B; the lists are an acceleration structure, never pixels) */
#define ZR_FLAG_NO_SHADOW_OCCLUSION 512u /* shadow pass: draw every survivor of the cull instead of leaving out what the map's own depths already hide
                                     * (parity A/B: the map is the same bit for bit either way) */
#define ZR_FLAG_SHADOW_OCCLUSION 1024u  /* ... and the opposite: occlusion-cull the shadow pass of a scene of any size (by default only from one
                                     * meshlet-instance per five texels of the map on, where it pays) */

typedef struct zr_config {
    uint32_t width, height;   /* swapchain extent, ZE:78-79 (default 1920x1080) */
    uint32_t shadow_dim;      /* SHADOWMAP_DIM, ZE:87 (0 -> 1024) */
    uint32_t debug_view;      /* GlobalConstants.SpecConstants 0..9, ZE:1803-1842 */
    int32_t  device;          /* HIP device ordinal */
    uint32_t tile_rank;       /* screen-tile partition: this context renders tiles t with */
    uint32_t tile_world;      /*   t % tile_world == tile_rank (0 -> 1) */
    uint32_t flags;
} zr_config;

/* One RGBA8 image; rgba8 == NULL selects the slot's engine default (ZE:4951-4978). */
typedef struct zr_image { const uint8_t* rgba8; uint32_t width, height; } zr_image;
/* Material = the 7 PBR samplers of BaseScene.frag:7-13 in order bc, m, r, n, ao, ev, ms. */
typedef struct zr_material { zr_image tex[XK_PBR_SAMPLER_NUMBER]; } zr_material;

/* XkCameraDesc (ZE:619-669) as plain floats; input of zr_update_uniforms. */
typedef struct zr_camera { float Position[3]; float Lookat[3]; float Speed, FOV, zNear, zFar; } zr_camera;

/* Per-pass GPU timings of the last zr_render, milliseconds (hipEvents on the render stream). */
enum { ZR_PASS_CULL_SHADOW = 0,   /* k_cull_box<SHADOW> + k_bin_count + k_scan + k_bin_fill (the map's clear rides in the previous lighting pass) */
       ZR_PASS_SHADOW,            /* k_raster_chunks<SHADOW> + k_tile_slow<SHADOW> (clipped triangles) */
       ZR_PASS_CULL_CAMERA,       /* k_cull_box<GBUFFER> (also compacts round 1's list: last frame's visible set)
                                     [ZR_FLAG_MESHLET_BINS: k_cull<GBUFFER> + k_bin_count + k_scan + k_bin_fill] */
       ZR_PASS_GBUFFER,           /* round 1 (the whole pass when the frame runs in one round): k_geom + k_scan_tri + k_index + k_tile
                                     [ZR_FLAG_MESHLET_BINS: k_raster_chunks<GBUFFER>] */
       ZR_PASS_HIZ,               /* k_hiz_build + round 2's k_select [.. k_bin_count + k_scan + k_bin_fill] (0 in a one-round frame) */
       ZR_PASS_GBUFFER2,          /* round 2: k_geom<Hi-Z> + k_scan_tri + k_index + k_tile (the last round also draws both rounds' clipped triangles) [.. k_raster_chunks<GBUFFER, Hi-Z>] */
       ZR_PASS_RESOLVE,           /* k_resolve_gbuffer: the GBuffer write */
       ZR_PASS_LIGHTING,          /* k_lighting */
       ZR_PASS_COMPOSITE, ZR_PASS_TOTAL, ZR_PASS_COUNT };

/* Frame statistics of the last zr_render (read back lazily by zr_get_stats). */
typedef struct zr_stats {
    uint64_t work_items[2];     /* meshlet-instances tested   [shadow, camera] */
    uint64_t survivors[2];      /* meshlet-instances binned   [shadow, camera] */
    uint64_t bin_entries[2];    /* (tile, meshlet-instance) pairs */
    uint64_t covered_pixels;    /* GBuffer pixels with geometry */
    uint64_t covered_shadow_texels; /* shadow-map texels with depth < 1 */
    uint32_t overflow;          /* nonzero: a bin list overflowed; frame invalid */
    uint32_t hiz_culled;        /* camera meshlet-instances rejected by the Hi-Z occlusion test (0 on a scene's first frame) */
    uint64_t round1_survivors;  /* of survivors[1]: drawn in round 1 (visible last frame); 0 when the frame ran in one round */
    uint32_t shadow_occluded;   /* of survivors[0]: not drawn - every texel their box reaches already held a nearer depth (the map is the same) */
    uint32_t shadow_late;       /* of survivors[0]: hidden last frame, not hidden now: drawn by the shadow pass's late launch */
    uint32_t hiz_culled_geom;   /* of hiz_culled: rejected only after a wave had transformed the meshlet's vertices (exact vertex box, every triangle
                                 * hidden); hiz_culled - hiz_culled_geom fell to the box bounds before any vertex work */
    uint32_t struct_bytes;      /* sizeof(zr_stats) as the LIBRARY knows it (ABI version 5: 96) */
} zr_stats;

/* --- lifetime (replaces InitVulkan/Cleanup, ZE:1714, 3747) --- */
int  zr_create(const zr_config* cfg, zr_ctx** out);
void zr_destroy(zr_ctx* ctx);
const char* zr_last_error(const zr_ctx* ctx);
/* Render on a caller-owned hipStream_t (NULL = the context's own stream). */
int  zr_set_stream(zr_ctx* ctx, void* hip_stream);
int  zr_tile_size(void);                            /* the ZR_TILE this library was built with */
uint32_t zr_abi_version(void);                      /* the ZR_ABI_VERSION this library was built with */

/* --- scene submission (replaces CreateRenderObjectsFromProfabs ZE:4922-5000, CreateMeshVertexBuffers
 *     ZE:4725-4770, CreateInstancedBuffer ZE:4795-4824) --- */
int  zr_mesh_create(zr_ctx* ctx, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, uint32_t* mesh_id);
/* Attach caller-built meshlets (the `.meshlet` payload, ZE:7089-7127).  The mesh's draw order becomes
 * meshlet order exactly as CreateMeshVertexBuffers<XkMeshIndirect> flattens it (ZE:4733-4756). */
int  zr_mesh_set_meshlets(zr_ctx* ctx, uint32_t mesh_id, const XkMeshlet* m, uint32_t nm,
                          const uint32_t* meshlet_vertices, size_t nmv,
                          const uint8_t* meshlet_triangles, size_t nmt);
/* Build meshlets with the library's own clusteriser (BuildMeshlets, ZM:132-172: 64 v / 124 t / cone 0.2).
 * Called implicitly by zr_object_add for meshes that have none.  Draw order stays index order. */
int  zr_mesh_build_meshlets(zr_ctx* ctx, uint32_t mesh_id, uint32_t max_vertices, uint32_t max_triangles,
                            float cone_weight);
/* The clusteriser without a context (pure host code, no GPU needed): the ZeldaMeshlet tool's job.  NULL outputs = size query. */
int  zr_meshlets_build(const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, uint32_t max_vertices,
                       uint32_t max_triangles, float cone_weight, XkMeshlet* m, uint32_t* nm, uint32_t* meshlet_vertices,
                       size_t* nmv, uint8_t* meshlet_triangles, size_t* nmt, uint32_t* tri_order);
/* Copy the mesh's meshlets out (any pointer may be NULL; counts always returned). */
int  zr_mesh_get_meshlets(zr_ctx* ctx, uint32_t mesh_id, XkMeshlet* m, uint32_t* nm,
                          uint32_t* meshlet_vertices, size_t* nmv, uint8_t* meshlet_triangles, size_t* nmt);
/* n_inst == 0: non-instanced draw (Base.vert); n_inst >= 1: instanced draw (BaseInstanced.vert). */
int  zr_object_add(zr_ctx* ctx, uint32_t mesh_id, const zr_material* mat,
                   const XkInstanceData* inst, uint32_t n_inst);
/* Capacities of the camera pass's triangle-record arrays (record_chunks x 256 records of 32 bytes, + 4 bytes of tile id) and of its
 * clipped-triangle list; 0 = defaults (16 records per meshlet-instance of the scene, at least 32 Mi; 2^18 triangles).  A frame that
 * outgrows either reports ZR_ERR_OVERFLOW at zr_finish (zr_last_error names which).  Takes effect at the next frame (the arrays are re-made). */
int  zr_set_limits(zr_ctx* ctx, uint32_t record_chunks, uint32_t slow_triangles);
/* The record arrays are laid out per frame as one bucket per screen tile, sized from the previous frame's count (+ 25 % + 128), and
 * an overflow region behind them for what a tile gets beyond its bucket.  percent (1..100, default 100) plans every bucket at that
 * share of its size: more of the arrays left to the overflow region, more records taking the slower route through it - same
 * frame, bit for bit (the overflow tests pin that).  Takes effect with the next plan (the next frame's end). */
int  zr_set_bucket_share(zr_ctx* ctx, uint32_t percent);
int  zr_scene_clear(zr_ctx* ctx);                   /* CleanupBasePass, ZE:4142 (also drops meshes and Profabs) */
int  zr_object_count(zr_ctx* ctx, uint32_t* n);
/* Copy out the instance array of object `index` (add order); *n = 0 for a non-instanced draw.  dst may be NULL. */
int  zr_object_get_instances(zr_ctx* ctx, uint32_t index, uint32_t* mesh_id, XkInstanceData* dst, uint32_t* n);
/* 6 RGBA8 sRGB faces in Vulkan layer order +X,-X,+Y,-Y,+Z,-Z (RHICreateTextureCubeResource ZE:5908-6150);
 * mips are generated like RHIGenerateMipmaps (ZE:6348-6433).  faces == NULL: built-in 1x1 grey. */
int  zr_set_cubemap(zr_ctx* ctx, const uint8_t* const faces[6], uint32_t dim);

/* Skydome pass (CreateSkydomePass ZE:2690-2744, draw ZE:3681-3691): the sky mesh (Content/Models/skydome.obj in the engine)
 * with its sRGB texture, drawn unlit after the lighting quad with depth LESS against the deferred depth.  tex == NULL removes it.
 * Background pass (ZE:2657-2688, 3693-3699): full-screen quad at z = 1, LESS_OR_EQUAL, sRGB texture.  Both only in debug view 0. */
int  zr_set_skydome(zr_ctx* ctx, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const zr_image* tex);
int  zr_set_background(zr_ctx* ctx, const zr_image* tex);
/* XkWorld::EnableSkydome / EnableBackground (zr_world_load_json sets them from the JSON). */
int  zr_set_sky_flags(zr_ctx* ctx, int enable_skydome, int enable_background);

/* --- per-frame uniforms (replaces UpdateWorld ZE:4294-4308 + UpdateUniformBuffer ZE:4585-4664) --- */
int  zr_update_uniforms(zr_ctx* ctx, const zr_camera* cam,
                        const XkLight* dir, uint32_t n_dir, const XkLight* point, uint32_t n_point,
                        const XkLight* spot, uint32_t n_spot, float roll_stage, float roll_light, float time);
/* Raw form: the three UBOs the engine memcpy's each frame (ZE:4626-4664). */
int  zr_set_frame(zr_ctx* ctx, const XkUniformBufferMVP* camera, const XkUniformBufferMVP* shadow, const XkView* view);
int  zr_get_frame(zr_ctx* ctx, XkUniformBufferMVP* camera, XkUniformBufferMVP* shadow, XkView* view);
int  zr_set_debug_view(zr_ctx* ctx, uint32_t spec_constants);
/* Which of the engine's two scene pipelines shades the frame (ZE:93 ENABLE_DEFERRED_SHADING picks one at compile time):
 *   ZR_SHADING_DEFERRED (default)  BaseScene.frag -> GBuffer -> BaseLighting.frag (ZE:2803-2997, 3417-3480, 3531-3540);
 *   ZR_SHADING_FORWARD             Base.frag:46-144 straight into the frame (pipelines ZE:2749-2801, draws ZE:3544-3680): unquantised
 *                                  inputs, no Mask, FinalColor * ShadowFactor, Base.frag's own debug table (:123-143; view 9 = view 0).
 * Takes effect with the next frame; between the stages of a frame: ZR_ERR_STATE.  The GBuffer targets are still written (zr_read_gbuffer). */
#define ZR_SHADING_DEFERRED 0u
#define ZR_SHADING_FORWARD  1u
int  zr_set_shading(zr_ctx* ctx, uint32_t mode);

/* --- the frame (replaces RecordCommandBuffer ZE:3160-3744 + vkQueueSubmit ZE:2014) ---
 * cull -> shadow -> cull -> gbuffer -> lighting [-> composite].  Asynchronous: the call only enqueues.  The shadow pipeline and
 * the lighting pass go to the render stream (zr_set_stream), the camera pipeline to a stream of the library's own that the
 * lighting pass waits for; so work the host enqueues on the render stream afterwards is ordered after the finished frame, and
 * up to two frames are in flight (the next frame's camera pipeline runs next to this frame's lighting), as in the reference
 * (MAX_FRAMES_IN_FLIGHT, ZE:77).  zr_finish and the read-back entry points wait for everything. */
int  zr_render(zr_ctx* ctx);
/* The same frame in three stages (zr_render = all three, in this order), so that a multi-GPU host can place its
 * collectives between them: shadow pass | deferred-scene pass (cull, raster, GBuffer write) | deferred-lighting pass. */
int  zr_render_shadow(zr_ctx* ctx);
int  zr_render_gbuffer(zr_ctx* ctx);
int  zr_render_lighting(zr_ctx* ctx);
/* Both geometry passes at once, side by side as zr_render runs them: the shadow pass on the render stream, the deferred-scene
 * pass on the library's own high-priority stream (zr_render_lighting waits for it).  A host that post-processes the shadow map
 * on another stream (the min all-reduce over ranks) makes that stream wait with zr_stream_wait_shadow, and the render stream
 * wait for its result, before zr_render_lighting.  zr_render = zr_render_geometry + zr_render_lighting.  Replaces the pair
 * zr_render_shadow + zr_render_gbuffer. */
int  zr_render_geometry(zr_ctx* ctx);
int  zr_stream_wait_shadow(zr_ctx* ctx, void* hip_stream);   /* hip_stream waits for the last enqueued shadow pass */
int  zr_finish(zr_ctx* ctx);                        /* stream sync + overflow check */
int  zr_get_pass_times(zr_ctx* ctx, float ms[ZR_PASS_COUNT]);              /* last frame */
int  zr_get_pass_times_avg(zr_ctx* ctx, uint32_t last_n, float ms[ZR_PASS_COUNT]);  /* mean of the last n <= 64 timed frames */
int  zr_get_frame_latencies(zr_ctx* ctx, uint32_t n, float* ms);   /* begin-to-end GPU ms of the last n timed frames, newest first; returns the count */
int  zr_get_frame_periods(zr_ctx* ctx, uint32_t n, float* ms);     /* GPU ms between the ends of consecutive frames, last n <= 511 frames, newest first; returns the count */
int  zr_set_timing_interval(zr_ctx* ctx, uint32_t interval);  /* pass events on every interval-th frame (default 1, 0 = never) */
/* bytes = sizeof(zr_stats) of the caller's header: the library writes min(bytes, its own size), so an older host is never overrun. */
int  zr_get_stats(zr_ctx* ctx, zr_stats* out, size_t bytes);

/* --- read-back (there is no swapchain; replaces vkQueuePresentKHR ZE:2030) --- */
int  zr_read_color(zr_ctx* ctx, uint8_t* rgba8, size_t bytes);         /* W*H*4 row-major, top-left origin */
/* target 0 depth D32F, 1 SceneColor RGBA8, 2 GBufferA A2R10G10B10, 3 GBufferB RGBA8, 4 GBufferC RGBA8,
 * 5 GBufferD RGBA16F (ZE:2807-2843); W*H*{4,4,4,4,4,8} bytes. */
int  zr_read_gbuffer(zr_ctx* ctx, int target, void* dst, size_t bytes);
int  zr_read_shadowmap(zr_ctx* ctx, float* dst, size_t bytes);        /* dim*dim*4 */
/* The frame enqueued last, copied into caller-owned DEVICE buffers (W*H*4 bytes of RGBA8; dim*dim*4 of depth; either may be NULL) in
 * stream order, without synchronising the host: the copies run behind that frame's lighting pass and ahead of anything the next
 * zr_render puts on the render stream - what a presenting host with two frames in flight uses instead of zr_read_color. */
int  zr_copy_frame_async(zr_ctx* ctx, void* color_dev, void* shadow_dev);

/* --- multi-GPU screen-tile partition --- */
/* Owner of tile (tx, ty) in a world of `world` ranks, and the list of tiles a rank owns in increasing tile index (= its slot order
 * in the packed buffer); slots_per_rank = the largest count over all ranks (every rank contributes that many slots to the all-gather).
 * Context-free; owned may be NULL. */
uint32_t zr_tile_owner(uint32_t tx, uint32_t ty, uint32_t world);
int  zr_tile_partition(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, uint32_t* owned, uint32_t* n_owned,
                       uint32_t* slots_per_rank);
/* Shadow pass split: this context draws instances i % world == rank into its shadow map; the caller min-reduces the maps
 * (depth test LESS_OR_EQUAL = min) between zr_render_shadow and zr_render_lighting.  Default 0 / 1 = everything. */
int  zr_set_shadow_partition(zr_ctx* ctx, uint32_t rank, uint32_t world);
/* Shadow pass split, second form: the MAP is owned by light-space super-tiles (zr_tile_owner on the map's 32 x 32-texel tiles) the way
 * the frame is owned by screen super-tiles.  This context then draws only the casters whose texel box can reach a tile it owns - whole,
 * so its owned tiles equal the single-GPU map's bit for bit; the ranks exchange tiles with ONE all-gather and nothing is reduced:
 *   zr_render_geometry | zr_shadow_pack -> all-gather of world x zr_shadow_tiles_bytes -> zr_shadow_unpack | zr_render_lighting
 * (zr_tile_partition(shadow_dim, shadow_dim, world, rank, ...) lists the owned tiles = the slots of the packed buffer; unused slots hold
 * depth 1.0.)  Replaces nothing in the reference (one device, ZE:2241); the pass it shards is ZE:3239-3393.  Default 0 / 1 = the whole map.
 * hip_stream NULL = the render stream. */
int  zr_set_shadow_tiles(zr_ctx* ctx, uint32_t rank, uint32_t world);
int  zr_shadow_tiles_bytes(zr_ctx* ctx, size_t* bytes_per_rank);
int  zr_shadow_pack(zr_ctx* ctx, void* packed_dev, void* hip_stream);
int  zr_shadow_unpack(zr_ctx* ctx, const void* gathered_dev, void* hip_stream);
/* Caller-owned shadow map, float[shadow_dim^2] device memory (NULL = internal). */
int  zr_set_shadow_buffer(zr_ctx* ctx, void* dev_ptr);
/* Packed tile-major RGBA8 of the tiles this rank owns (device pointer, stable until zr_destroy). */
int  zr_tiles_device_buffer(zr_ctx* ctx, void** dev_ptr, size_t* bytes_per_rank);
/* Caller-owned packed buffer for the following frames (same size; NULL = the internal one).  Two alternating buffers let
 * frame k's all-gather overlap frame k+1's rendering. */
int  zr_set_tiles_buffer(zr_ctx* ctx, void* dev_ptr);
int  zr_read_tiles(zr_ctx* ctx, uint8_t* dst, size_t bytes);             /* host copy of that buffer (tests) */
/* Scatter the all-gathered buffer (tile_world * bytes_per_rank, rank-major, device pointer) into the frame. */
int  zr_composite(zr_ctx* ctx, const void* gathered_dev);
int  zr_color_device_ptr(zr_ctx* ctx, void** dev_ptr);

/* --- native multi-GPU host: one process per GPU, RCCL over xGMI called by the library itself (no Python, no torch in the frame) ---
 * rank 0 makes the id (zr_dist_unique_id: 128 bytes, ncclGetUniqueId) and hands it to the other ranks by any means; every rank then
 * calls zr_dist_init on a context created with the matching tile_rank / tile_world.  zr_dist_frame enqueues one frame:
 *   render stream + camera lane: the frame of this rank's tiles into packed buffer k & 1
 *   collective stream: ncclAllGather of the packed RGBA8 tiles (4 B per pixel of the frame in total) -> untile into the frame,
 *   overlapped with the rendering of frame k + 1 (buffers are double-buffered, ordering is by events).
 * The all-gather of the composite is the ONLY collective, unless ZR_DIST_SPLIT_SHADOW is given: then each rank rasterises the shadow
 * casters i % world == rank and the 1024^2 maps are reduced with ncclAllReduce(min) next to the camera passes.
 * zr_finish / the read-back entry points wait for the collective stream too.  librccl is loaded on first use (dlopen). */
#define ZR_DIST_SPLIT_SHADOW 1u
/* ... or ZR_DIST_SHADOW_TILES: the shadow MAP is owned by light-space super-tiles (zr_set_shadow_tiles) and the second collective is an
 * ncclAllGather of the packed shadow tiles (4 MiB in total for a 1024^2 map, no reduction), between the shadow pass and the lighting pass. */
#define ZR_DIST_SHADOW_TILES 2u
int  zr_dist_unique_id(void* id, size_t bytes);                       /* bytes must be 128 */
int  zr_dist_init(zr_ctx* ctx, const void* id, size_t bytes, uint32_t rank, uint32_t world, uint32_t dist_flags);
/* The same bring-up in two steps, for hosts that can agree between them: zr_dist_prepare is LOCAL (librccl, collective stream, packed /
 * gathered buffers) and may fail on one rank alone; zr_dist_connect calls ncclCommInitRank, which is itself a collective - a rank that
 * never reaches it leaves the others blocked inside it - so call it only after every rank reported a successful prepare.
 * zr_dist_init = prepare + connect.  A failed connect leaves a plain single-context renderer behind (whole shadow map). */
int  zr_dist_prepare(zr_ctx* ctx, uint32_t rank, uint32_t world, uint32_t dist_flags);
int  zr_dist_connect(zr_ctx* ctx, const void* id, size_t bytes);
int  zr_dist_frame(zr_ctx* ctx);
/* The composite of the frame zr_dist_frame enqueued last, copied into a caller-owned DEVICE buffer (W*H*4 bytes of RGBA8) in the order of
 * the collective stream: behind that frame's all-gather + untile, ahead of the next frame's - zr_copy_frame_async for a multi-GPU host
 * that keeps frames in flight (a presenting rank, a test that checks EVERY frame of a pipelined sequence). */
int  zr_dist_copy_frame_async(zr_ctx* ctx, void* color_dev);

/* --- world JSON + livelink (replaces XkWorld::Load ZE:1051-1147, socket thread ZE:1617-1710) --- */
/* A Profab is the engine's asset bundle `Profabs/<name>/{models, textures}` (ZE:4922-5000).  Either the caller registers each
 * model of a Profab (mesh + 7-texture material) under its name, or - after zr_set_asset_root - zr_world_load_json finds
 * `Profabs/<name>` on disk itself.  Unknown names draw nothing, like a missing directory. */
int  zr_profab_register(zr_ctx* ctx, const char* name, uint32_t mesh_id, const zr_material* mat);

/* --- the content tree (replaces ASSETS()/AssetPathSearch ZE:7173-7263, LoadMeshAsset ZE:6899-6948, LoadTextureAsset ZE:6882-6896,
 *     LoadMeshletAsset ZE:7046-7169, the Profab walk ZE:4922-5000 and the world's sky / cubemap / background overrides ZE:4147-4183) --- */
/* dir = the engine's working directory (holds Profabs/ and Content/); NULL switches file-system lookups off again.  With a root
 * set, zr_world_load_json also applies OverrideCubemap / OverrideSkydome (mesh: Content/Models/skydome.obj) / OverrideBackground. */
int  zr_set_asset_root(zr_ctx* ctx, const char* dir);
/* literal path -> Profabs/<set>/{models,textures} -> Content/<set>/...; returns the input (rooted) when nothing is found */
int  zr_asset_path_search(zr_ctx* ctx, const char* name, char* dst, size_t cap, size_t* len);
/* context-free loaders (host code only).  NULL outputs = size query (*nv / *ni, or *w / *h, are set). */
int  zr_load_obj(const char* path, XkVertex* v, uint32_t* nv, uint32_t* idx, uint32_t* ni);
int  zr_load_png_rgba8(const char* path, uint8_t* dst, size_t cap, uint32_t* w, uint32_t* h);
/* `.meshlet` file (ZM:52-75) -> mesh with the file's meshlets attached (zr_mesh_create + zr_mesh_set_meshlets) */
int  zr_load_meshlet_file(zr_ctx* ctx, const char* path, uint32_t* mesh_id);
/* XkWorld::Load() / Save() on a file (NULL = "Content/World.json", ZE:1027), relative to the asset root */
int  zr_world_load_file(zr_ctx* ctx, const char* path);
int  zr_world_save_file(zr_ctx* ctx, const char* path);
/* per-frame UpdateWorld + UpdateUniformBuffer from the loaded world's camera and lights (ZE:4294-4308, 4585-4664) */
int  zr_world_update_uniforms(zr_ctx* ctx, float roll_stage, float roll_light, float time);
/* XkWorld::Load + CreateEngineScene + the first UpdateUniformBuffer: replaces the scene's objects (InstanceCount > 1 ->
 * GenerateInstance, seeded PCG32(1234 + object index)) and sets camera + lights. */
int  zr_world_load_json(zr_ctx* ctx, const char* utf8, size_t len);
int  zr_world_get_camera(zr_ctx* ctx, zr_camera* out);
int  zr_world_save_json(zr_ctx* ctx, char* dst, size_t cap, size_t* len);
/* Context-free Load -> Save (pure host code, no GPU): validates a payload; on error dst receives the message. */
int  zr_world_json_normalize(const char* utf8, size_t len_in, char* dst, size_t cap, size_t* len);
/* Listens on loopback by default (the payload is unauthenticated); any != 0 selects the engine's wildcard bind (AI_PASSIVE,
 * ZE:1630-1636).  Call before zr_livelink_serve.  A client that connects and stays silent is dropped after 2 s. */
int  zr_livelink_bind_any(zr_ctx* ctx, int any);
int  zr_livelink_serve(zr_ctx* ctx, uint16_t port);   /* port 0 = ephemeral (tests); the engine's port is 8080 */
int  zr_livelink_port(zr_ctx* ctx, uint16_t* port);
int  zr_livelink_poll(zr_ctx* ctx, int* reloaded);  /* DrawFrame's bReloadScene pickup, ZE:1943-1951 */
int  zr_livelink_stop(zr_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* ZELDA_RENDER_H */
