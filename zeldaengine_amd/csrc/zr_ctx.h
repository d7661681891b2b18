// zr_ctx.h — internal context shared by zr_host.cpp (renderer) and zr_world.cpp (JSON world + livelink).
#pragma once

#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "zr_meshlet.h"
#include "zr_types.h"

struct ZrMesh {
    std::vector<XkVertex> v;
    std::vector<uint32_t> idx;           // draw-order index buffer
    ZrMeshletSet ms;
    bool has_meshlets = false, uploaded = false;
    float center[3] = { 0, 0, 0 }; float radius = 0;
    XkVertex* d_v = nullptr; ZrRVertex* d_rv = nullptr; ZrRVertex* d_rt = nullptr; uint32_t* d_idx = nullptr; XkMeshlet* d_meshlets = nullptr;
    float4* d_mpos = nullptr; float4* d_mbox = nullptr; uint2* d_mtri = nullptr; uint32_t* d_tri_meshlet = nullptr;
};

// Host form of one material: per slot either a constant texel or an RGBA8 image (mips are built at zr_object_add).
struct ZrMaterialHost {
    uint32_t texel[7]; float bc_linear[3];
    std::vector<uint8_t> image[7];       // empty: the slot is constant
    uint32_t w[7], h[7];
};

struct ZrSceneObject {
    uint32_t mesh = 0, n_inst = 1; bool instanced = false;
    std::vector<XkInstanceData> inst;    // host copy (zr_object_get_instances)
    ZrInstance* d_inst = nullptr;
    uint32_t texel[7]; float bc_linear[3];
    uint8_t* d_tex[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };   // [7]: the packed material (ZrObject::packed)
    uint32_t tex_w[8] = { 0 }, tex_h[8] = { 0 }, tex_levels[8] = { 0 };
    bool mixed_sizes = false;            // image slots of different sizes: no packed form
};

// XkWorld (ZE:1025-1291) as parsed from JSON
struct ZrLightDesc { float Position[3]; uint32_t Type; float Color[3]; float Intensity; float Direction[3]; float Radius; float ExtraData[4]; };
struct ZrObjectDesc {
    uint32_t RenderFlags = 0; std::string ProfabName; uint32_t InstanceCount = 0;
    float MinRadius = 0, MaxRadius = 0, MinRotYaw = 0, MaxRotYaw = 0, MinRotRoll = 0, MaxRotRoll = 0,
          MinRotPitch = 0, MaxRotPitch = 0, MinPScale = 0, MaxPScale = 0;
};
struct ZrWorld {
    bool EnableSkydome = true, OverrideSkydome = false; std::string SkydomeFileName;
    bool OverrideCubemap = false; std::string CubemapFileNames[6];
    bool EnableBackground = false, OverrideBackground = false; std::string BackgroundFileName;
    zr_camera MainCamera;
    std::vector<ZrLightDesc> DirectionalLights, PointLights, SpotLights;
    std::vector<ZrObjectDesc> ObjectDescs;
    bool loaded = false;
};
struct ZrProfab { uint32_t mesh; ZrMaterialHost mat; };

struct zr_ctx {
    zr_config cfg;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;

    std::vector<ZrMesh> meshes;
    std::vector<ZrSceneObject> objects;
    bool scene_dirty = true;
    ZrObject* d_objs = nullptr; uint32_t n_objs = 0, n_work = 0;
    // this frame's two geometry passes (0 shadow, 1 camera), built at frame begin; the passes' work lists (k_cull_instances) are kept
    // while the pass block and the scene stand still: list_key = the block the list on the device was built from
    ZrPass pass[2]; bool pass_live[2] = { false, false }, list_reuse[2] = { false, false }, list_valid[2] = { false, false };
    ZrPass list_key[2];

    XkUniformBufferMVP cam, shadow; XkView view; XkView* d_view = nullptr; bool frame_valid = false;
    uint32_t debug_view = 0;
    uint32_t shading = 0;                           // ZR_SHADING_*: which scene pipeline shades the frame (zr_set_shading)

    uint32_t W = 0, H = 0, SD = 0;
    uint32_t tiles_x = 0, tiles_y = 0, n_tiles = 0, n_owned = 0, slots_per_rank = 0;
    uint32_t stiles_x = 0, stiles_y = 0, sn_tiles = 0;
    uint32_t *d_owned = nullptr, *d_sowned = nullptr;
    uint32_t* d_tile_map = nullptr;      // tile -> owner * slots_per_rank + slot (k_untile)
    struct ZrDist* dist = nullptr;       // native multi-GPU host (zr_dist.cpp), or null
    GBufferPtrs G = {};
    float* d_shadow = nullptr; uint32_t* d_color = nullptr; uint32_t* d_tiles = nullptr;
    float* d_shadow_ext = nullptr;       // caller-owned shadow map (zr_set_shadow_buffer), or null
    uint32_t shadow_rank = 0, shadow_world = 1; int stage = 0;   // stage: 0 idle, 1 shadow done, 2 gbuffer done
    // The shadow MAP owned by light-space super-tiles (zr_set_shadow_tiles): this context draws the casters that can reach a tile of the map
    // it owns; its owned tiles are exact, the others hold leftovers until zr_shadow_unpack scatters every rank's tiles in.
    uint32_t stile_rank = 0, stile_world = 1, s_slots_per_rank = 0, n_sowned_rank = 0;
    uint32_t *d_sowned_rank = nullptr, *d_stile_map = nullptr;
    uint32_t* d_tiles_ext = nullptr;     // caller-owned packed tile buffer for the next frames (zr_set_tiles_buffer), or null

    // cull / bin scratch, one set per geometry pass (0 shadow, 1 camera) so that the two pipelines can run on two streams
    struct Scratch { uint32_t *rects = nullptr, *tile_count = nullptr, *tile_offset = nullptr, *tile_cursor = nullptr, *chunk_offset = nullptr,
                     *work = nullptr; ZrBinEntry* bins = nullptr; uint4* chunk_tab = nullptr; } sc[2];
    uint32_t bucket_pct = 100;                              // zr_set_bucket_share: every planned bucket at that share of its size
    bool plan_valid = false, plan_two_round = false;      // the record buckets' plan (k_plan): made at all / by a frame that drew two rounds
    ZrTriBins tb = {};                    // triangle-binned camera pass: selection list, records (as emitted / in tile order), slow list
    uint32_t chunk_capacity = 0;         // raster work units the chunk table holds: bin_capacity / ZR_CHUNK + tiles
    uint32_t n_inst_total = 0;
    // one pixel holding the clear value of every GBuffer target, and the colour the lighting shader gives it this frame
    uint8_t* d_clear_px = nullptr; GBufferPtrs Gclear = {}; uint32_t* d_empty_rgba = nullptr; bool empty_ready = false;
    // diagnostics / A-B switches read from the environment once, at zr_create (never needed for a correct frame)
    uint32_t env_skip = 0, env_skip_light = 0; int32_t env_light_list_min = 4; bool env_no_empty_px = false, env_serial = false;
    // XkView upload: a pageable-memory hipMemcpyAsync blocks the host until the stream has drained (~0.3 ms per frame here),
    // so the uniforms go through a small ring of pinned copies, and only when they changed
    static constexpr int VIEW_RING = 4;
    XkView* h_view_ring = nullptr; hipEvent_t view_ev[VIEW_RING] = {}; uint32_t view_slot = 0; bool view_dirty = true;
    uint64_t view_version = 1, view_uploaded[2] = { 0, 0 };      // which version of the uniforms each device copy holds
    // Two frames in flight (the reference does: MAX_FRAMES_IN_FLIGHT, ZE:77): the camera pipeline runs on `cam_s`, the shadow
    // pipeline and the lighting pass on the host's `stream`; frame N + 1's camera pipeline overlaps frame N's lighting.
    // What a lighting pass reads is therefore double-buffered (GBuffer, shadow map, XkView, the empty-pixel colour); G, d_shadow,
    // d_view, d_empty_rgba are aliases of the current frame's copies (set at frame begin, so the read-back entry points see the
    // frame rendered last).
    hipStream_t cam_s = nullptr; bool camera_on_lane = false;
    // Experiment kept behind ZR_LANES=3 (zr_render only, not the staged entry points): the shadow pipeline and the lighting pass on
    // streams of their own as well, so that frame N's lighting, frame N + 1's shadow pipeline and frame N + 1's camera pipeline all run
    // side by side and the host's stream only joins the finished frame.  Measured SLOWER than two lanes: the
    // camera pipeline - a chain of short kernels - is then starved by two heavy neighbours instead of one.
    bool in_render = false;
    hipEvent_t ev_join = nullptr, ev_cam = nullptr;
    unsigned long long* d_sky_keys = nullptr; uint32_t sky_object = 0;      // the skydome's key plane (k_sky_tiles) and its draw record
    // End of every frame's lighting pass, one (timing-enabled) event per frame in a ring: the next-but-one frame waits for it before
    // it reuses the double-buffered copies, and consecutive ones give the per-frame GPU period (zr_get_frame_periods) for free.
    static constexpr int END_RING = 512;
    hipEvent_t ev_end[END_RING] = {};
    uint32_t* d_prim_b[2] = { nullptr, nullptr };   // forward variant: winner ids per GBuffer copy (zr_set_shading)
    GBufferPtrs Gb[2] = {}; float* d_shadow_b[2] = { nullptr, nullptr }; XkView* d_view_b[2] = { nullptr, nullptr };
    uint32_t* d_empty_b[2] = { nullptr, nullptr };
    bool overlay_dirty[2] = { false, false };       // Gb[i].overlay may hold skydome pixels of an earlier frame
    bool shadow_cleared[2] = { false, false };      // d_shadow_b[i] already holds depth 1.0 (cleared by the previous lighting pass)
    unsigned long long* d_vis = nullptr; uint32_t raster_blocks = 2048, shadow_blocks = 2048; bool env_shadow_box = true, env_shadow_defer = true;
    uint4* d_slow0 = nullptr; uint32_t slow0_cap = 1u << 18;      // shadow pass: triangles for the clipper (k_tile_slow)
    uint32_t work_capacity = 0, bin_capacity = 0; bool any_images = false, mixed_images = false;
    uint32_t limit_record_chunks = 0, limit_slow_triangles = 0;      // zr_set_limits (0 = defaults)
    // two-pass Hi-Z occlusion culling of the camera pass: per work item pixel bbox + least depth (written by the cull),
    // visibility of the previous / current frame (one byte per meshlet-instance, marked by the resolve), the pyramid
    uint2* d_pxrect = nullptr; float* d_zmin = nullptr; uint8_t* d_visflag[2] = { nullptr, nullptr };
    // shadow pass occlusion culling (k_shadow_occlusion): the cull's box + least depth per work item, "not hidden last frame" per meshlet-instance
    uint2* d_spxrect = nullptr; float* d_szmin = nullptr; uint8_t* d_sflag = nullptr;
    bool sflag_history = false;          // the flags come from a frame of this scene (else: all set, and the first test takes every item)
    float* d_hiz = nullptr; ZrHiz hiz = {}; int vis_cur = 0; bool vis_history = false, last_two_round = false;
    uint32_t vis_mark_prev = 0;          // the stamp the resolve wrote into last frame's visibility marks (ZrHiz::vis_stamp)
    uint32_t* d_hiz_regions = nullptr; uint32_t n_hiz_regions = 0;      // the 64 x 64 pixel regions over owned tiles (k_hiz_build)
    ZrDevStats* d_stats = nullptr; ZrDevStats h_stats = {};
    // The shadow pipeline's statistics / work counters (slot 0) live in a block of their own: the pipeline resets what it counts itself
    // (k_scan), so it does not wait for the camera lane's k_frame_begin, and the camera lane does not wait for it.
    ZrDevStats* d_sstats = nullptr; uint32_t list_rebuild_mask = 0;
    uint64_t last_work[2] = { 0, 0 };

    std::vector<uint8_t*> d_cube; CubeDesc cube = {}; uint32_t cube_dim = 0, cube_levels = 0;
    float lut[256]; float* d_lut = nullptr;
    float* d_unorm_lut = nullptr;        // [0..255] = c / 255, [256..1279] = c / 1023 (IEEE quotients, computed on the host)

    static constexpr int EV_RING = 64;     // per-pass hipEvents of the last EV_RING timed frames (bench averages over them)
    // skydome + background passes (ZE:2657-2744, 3681-3699)
    ZrMesh sky_mesh; ZrSceneObject sky_obj; bool sky_set = false, sky_enabled = true;
    uint8_t* d_bg = nullptr; uint32_t bg_w = 0, bg_h = 0, bg_levels = 0; bool bg_set = false, bg_enabled = true;

    hipEvent_t evr[EV_RING][10] = {}; uint64_t frame_no = 0; bool rendered = false;
    uint32_t timing_interval = 1; bool timing_now = true; uint64_t sample_no = 0;    // pass events every interval-th frame

    // world + livelink + the content tree (zr_assets.cpp)
    std::string asset_root; bool assets_on = false;     // directory holding Profabs/ and Content/ (the engine's working directory)
    ZrWorld world;
    std::map<std::string, std::vector<ZrProfab>> profabs;
    std::mutex ll_mutex; std::thread ll_thread; std::atomic<bool> ll_run{ false };
    int ll_listen_fd = -1; bool ll_pending = false, ll_bind_any = false; ZrWorld ll_world; uint16_t ll_port = 0;
};

int zr_fail(zr_ctx* c, int code, const std::string& msg);      // records the message (never throws), returns code
// No exception crosses the C-ABI: entry points that build host-side containers run their body through this.
template <typename F> static inline int zr_guard(zr_ctx* c, F&& body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc&) { return zr_fail(c, ZR_ERR_OOM, "out of host memory"); }
    catch (const std::exception& e) { return zr_fail(c, ZR_ERR_IO, e.what()); }
    catch (...) { return zr_fail(c, ZR_ERR_IO, "unexpected exception"); }
}
hipError_t zr_sync_all(zr_ctx* c);     // every stream the library enqueues on
// helpers implemented in zr_host.cpp and used by zr_world.cpp
float zr_srgb_decode8(uint32_t c);
int zr_material_prepare(zr_ctx* c, const zr_material* mat, ZrMaterialHost* out);
int zr_object_add_internal(zr_ctx* c, uint32_t mesh_id, const ZrMaterialHost& mat, const XkInstanceData* inst, uint32_t n_inst);
// zr_dist.cpp
void zr_dist_destroy(zr_ctx* c);
hipError_t zr_dist_sync(zr_ctx* c);
// zr_assets.cpp
std::string zr_asset_search(const zr_ctx* c, const std::string& name);
int zr_profab_from_disk(zr_ctx* c, const std::string& name, int* found);
int zr_world_apply_overrides(zr_ctx* c, const ZrWorld& w);
