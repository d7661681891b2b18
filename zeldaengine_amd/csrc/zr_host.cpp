// zr_host.cpp — context, scene upload, per-frame uniforms and the frame graph behind the C-ABI (zelda_render.h).
//
// Host counterpart of XkZeldaEngineApp's CreateEngineScene / UpdateUniformBuffer / RecordCommandBuffer / DrawFrame
// (ZE:4140, 4585, 3160, 1940).  GPU work is enqueued on two HIP streams (the host's render stream and the library's camera
// lane, see geometry_passes); nothing here computes a pixel on the CPU and there is no fallback: without a usable HIP device
// zr_create fails.
#include "zr_ctx.h"
#include "zr_math.h"

#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <cstdlib>

#define HIPCHK(c, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
    return zr_fail((c), ZR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
#define ARGCHK(c, cond) do { if (!(cond)) return zr_fail((c), ZR_ERR_ARG, "bad argument: " #cond); } while (0)

int zr_fail(zr_ctx* c, int code, const std::string& msg)
{
    if (c) { try { c->err = msg; } catch (...) { c->err.clear(); } }      // (called from catch blocks: must not throw itself)
    return code;
}

// Everything the library has enqueued: the host's stream (shadow pipeline, lighting) and its own camera lane.
hipError_t zr_sync_all(zr_ctx* c)
{
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && c->cam_s) e = hipStreamSynchronize(c->cam_s);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // (the host's stream may have been made to wait for the lanes)
    if (e == hipSuccess) e = zr_dist_sync(c);           // the native multi-GPU host's collective stream, if any
    return e;
}

static const uint8_t kDefaultTexel[7][4] = {    // ZE:4951-4978: default_{grey,black,white,normal,white,black,white}.png
    {127,127,127,255}, {0,0,0,255}, {255,255,255,255}, {127,127,255,255}, {255,255,255,255}, {0,0,0,255}, {255,255,255,255}
};

float zr_srgb_decode8(uint32_t c)
{
    double x = (double)c / 255.0;
    double l = (x <= 0.04045) ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
    return (float)l;
}
static uint8_t srgb_encode8(float l)
{
    double x = (double)l;
    if (!(x > 0.0)) x = 0.0;
    if (x > 1.0) x = 1.0;
    double s = (x <= 0.0031308) ? 12.92 * x : 1.055 * pow(x, 1.0 / 2.4) - 0.055;
    return (uint8_t)floor(s * 255.0 + 0.5);
}

template <typename T> static hipError_t dev_alloc(T** p, size_t n) { return hipMalloc((void**)p, (n ? n : 1) * sizeof(T)); }
template <typename T> static void dev_free(T*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } }
// Images that kernels gather from at random (material textures, skydome, background): allocated in whole 2 MiB units, so that the
// driver maps them with large page fragments whatever the allocator's pools look like at the time - a 1.4 MiB texture that lands
// in 4 KiB-mapped memory costs the sampled resolve a third of its speed (seen as two modes of `value_textured`, run to run).
static hipError_t dev_alloc_image(uint8_t** p, size_t bytes)
{
    const size_t unit = (size_t)2 << 20;
    return hipMalloc((void**)p, (bytes + unit - 1) / unit * unit);
}

// ------------------------------------------------------------------------------------------------ lifetime

static void default_lights(XkView* v)
{   // XkLight() default constructor, ZE:779
    XkLight d; memset(&d, 0, sizeof d);
    d.Color[0] = d.Color[1] = d.Color[2] = d.Color[3] = 1.0f; d.Direction[2] = 1.0f; d.Direction[3] = 1.0f;
    for (auto& l : v->DirectionalLights) l = d;
    for (auto& l : v->PointLights) l = d;
    for (auto& l : v->SpotLights) l = d;
}

extern "C" int zr_create(const zr_config* cfg, zr_ctx** out)
{
    if (!cfg || !out) return ZR_ERR_ARG;
    *out = nullptr;
    if (cfg->width == 0 || cfg->height == 0 || cfg->width > 255u * ZR_TILE || cfg->height > 255u * ZR_TILE) return ZR_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZR_ERR_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return ZR_ERR_DEVICE;
    if (hipSetDevice(cfg->device) != hipSuccess) return ZR_ERR_DEVICE;
    zr_ctx* c = new zr_ctx();
    c->cfg = *cfg;
    if (c->cfg.tile_world == 0) c->cfg.tile_world = 1;
    if (c->cfg.tile_rank >= c->cfg.tile_world) { delete c; return ZR_ERR_ARG; }
    c->device = cfg->device;
    c->W = cfg->width; c->H = cfg->height; c->SD = cfg->shadow_dim ? cfg->shadow_dim : XK_SHADOWMAP_DIM;
    if (c->SD > 255u * ZR_TILE) { delete c; return ZR_ERR_ARG; }
#ifndef ZR_DIAG
    if (c->cfg.flags & ZR_FLAG_MESHLET_BINS) { delete c; return ZR_ERR_UNSUPPORTED; }      // the A/B rasteriser exists in -DZR_DIAG builds only (refused before anything is allocated)
#endif
    c->debug_view = cfg->debug_view;
    memset(&c->cam, 0, sizeof c->cam); memset(&c->shadow, 0, sizeof c->shadow); memset(&c->view, 0, sizeof c->view);
    default_lights(&c->view);
    for (int i = 0; i < 256; ++i) c->lut[i] = zr_srgb_decode8((uint32_t)i);

    bool ok = true;
    ok &= hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess;
    c->stream = c->own_stream;
    for (auto& fr : c->evr) for (auto& e : fr) ok &= hipEventCreate(&e) == hipSuccess;
    for (auto& e : c->ev_end) ok &= hipEventCreate(&e) == hipSuccess;
    const size_t n = (size_t)c->W * c->H;
    for (int b = 0; b < 2; ++b) {       // two frames in flight: see zr_ctx.h
        GBufferPtrs& G = c->Gb[b];
        ok &= dev_alloc(&G.depth, n) == hipSuccess;
        ok &= dev_alloc(&G.scene_color, n) == hipSuccess;
        ok &= dev_alloc(&G.gA, n) == hipSuccess;
        ok &= dev_alloc(&G.gB, n) == hipSuccess;
        ok &= dev_alloc(&G.gC, n) == hipSuccess;
        ok &= dev_alloc(&G.gD, n) == hipSuccess;
        ok &= dev_alloc(&G.overlay, n) == hipSuccess;
        if (ok) ok &= hipMemset(G.overlay, 0, n * 4) == hipSuccess;
        ok &= dev_alloc(&c->d_shadow_b[b], (size_t)c->SD * c->SD) == hipSuccess;
        ok &= dev_alloc(&c->d_view_b[b], 1) == hipSuccess;
        ok &= dev_alloc(&c->d_empty_b[b], 1) == hipSuccess;
    }
    c->G = c->Gb[0]; c->d_shadow = c->d_shadow_b[0]; c->d_view = c->d_view_b[0]; c->d_empty_rgba = c->d_empty_b[0];
    ok &= dev_alloc(&c->d_color, n) == hipSuccess;
    ok &= dev_alloc(&c->d_stats, 1) == hipSuccess;
    if (ok) ok &= hipMemset(c->d_stats, 0, sizeof(ZrDevStats)) == hipSuccess;
    ok &= dev_alloc(&c->d_sstats, 1) == hipSuccess;       // the shadow pipeline's own block (see zr_ctx.h)
    if (ok) ok &= hipMemset(c->d_sstats, 0, sizeof(ZrDevStats)) == hipSuccess;
    ok &= dev_alloc(&c->d_lut, 256) == hipSuccess;
    if (ok) ok &= hipMemcpy(c->d_lut, c->lut, sizeof c->lut, hipMemcpyHostToDevice) == hipSuccess;
    {
        std::vector<float> ul(1280);
        for (int i = 0; i < 256; ++i) {
            ul[(size_t)i] = (float)i / 255.0f;
            ok &= fmaf((float)i, ZR_UNORM8_HI, (float)i * ZR_UNORM8_LO) == ul[(size_t)i];        // the packed sampler's division-free decode
        }
        for (int i = 0; i < 1024; ++i) ul[256 + (size_t)i] = (float)i / 1023.0f;
        ok &= dev_alloc(&c->d_unorm_lut, ul.size()) == hipSuccess;
        if (ok) ok &= hipMemcpy(c->d_unorm_lut, ul.data(), ul.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) ok &= hipMemset(c->d_color, 0, n * 4) == hipSuccess;

    // screen tiles: camera target partitioned t % world == rank; the shadow map is rendered whole on every rank
    c->tiles_x = (c->W + ZR_TILE - 1) / ZR_TILE; c->tiles_y = (c->H + ZR_TILE - 1) / ZR_TILE; c->n_tiles = c->tiles_x * c->tiles_y;
    c->stiles_x = (c->SD + ZR_TILE - 1) / ZR_TILE; c->stiles_y = c->stiles_x; c->sn_tiles = c->stiles_x * c->stiles_y;
    if (c->n_tiles > 16000u || c->sn_tiles > 16000u) { zr_destroy(c); return ZR_ERR_ARG; }   // binning histograms (4 B per tile, dynamic) + a few static words must fit the default 64 KB of LDS per workgroup
    // tile ownership (zr_tile_owner): per rank the owned tiles in increasing index = its slots in the packed buffer
    std::vector<uint32_t> owned, sowned(c->sn_tiles), tile_map(c->n_tiles), counts(c->cfg.tile_world, 0u);
    for (uint32_t t = 0; t < c->n_tiles; ++t) {
        const uint32_t o = zr_tile_owner(t % c->tiles_x, t / c->tiles_x, c->cfg.tile_world);
        tile_map[t] = counts[o]++;                       // slot within its owner, for now
        if (o == c->cfg.tile_rank) owned.push_back(t);
    }
    c->slots_per_rank = 0;
    for (uint32_t n : counts) c->slots_per_rank = std::max(c->slots_per_rank, n);
    for (uint32_t t = 0; t < c->n_tiles; ++t)
        tile_map[t] += zr_tile_owner(t % c->tiles_x, t / c->tiles_x, c->cfg.tile_world) * c->slots_per_rank;
    ok &= dev_alloc(&c->d_tile_map, tile_map.size()) == hipSuccess;
    if (ok) ok &= hipMemcpy(c->d_tile_map, tile_map.data(), tile_map.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    for (uint32_t t = 0; t < c->sn_tiles; ++t) sowned[t] = t;
    c->n_owned = (uint32_t)owned.size();
    ok &= dev_alloc(&c->d_owned, owned.size()) == hipSuccess;
    ok &= dev_alloc(&c->d_sowned, sowned.size()) == hipSuccess;
    ok &= dev_alloc(&c->d_tiles, (size_t)c->slots_per_rank * ZR_TILE * ZR_TILE) == hipSuccess;
    if (ok && !owned.empty()) ok &= hipMemcpy(c->d_owned, owned.data(), owned.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok &= hipMemcpy(c->d_sowned, sowned.data(), sowned.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok &= hipMemset(c->d_tiles, 0, (size_t)c->slots_per_rank * ZR_TILE * ZR_TILE * 4) == hipSuccess;
    const uint32_t mt = (c->n_tiles > c->sn_tiles ? c->n_tiles : c->sn_tiles) * ZR_TSTRIDE + 1;     // (the triangle-binned pass spreads its counters)
    for (auto& sc : c->sc) {
        ok &= dev_alloc(&sc.tile_count, mt) == hipSuccess;
        ok &= dev_alloc(&sc.tile_offset, mt) == hipSuccess;
        ok &= dev_alloc(&sc.tile_cursor, mt) == hipSuccess;
        ok &= dev_alloc(&sc.chunk_offset, mt) == hipSuccess;
        if (ok) ok &= hipMemset(sc.tile_count, 0, mt * 4) == hipSuccess;       // k_geom counts into zeroes, k_index advances cursors from zero:
        if (ok) ok &= hipMemset(sc.tile_cursor, 0, mt * 4) == hipSuccess;      // k_tile leaves both that way for the next round
    }
    {   // the clear values of ZE:3427-3433, as resolve_pixel writes them for an empty pixel
        ok &= dev_alloc(&c->d_clear_px, 64) == hipSuccess;
        uint32_t px[16] = { 0 };
        px[0] = 0x3F800000u;                    // depth 1.0
        px[1] = 0xFF000000u; px[2] = 0u; px[3] = 0xFF000000u; px[4] = 0xFF000000u;   // SceneColor, A, B, C
        px[6] = 0u; px[7] = 0x3C000000u;        // D = (0, 0, 0, 1) as fp16
        px[8] = 0u;                             // overlay
        if (ok) ok &= hipMemcpy(c->d_clear_px, px, sizeof px, hipMemcpyHostToDevice) == hipSuccess;
        uint32_t* w = (uint32_t*)c->d_clear_px;
        c->Gclear.depth = (float*)w; c->Gclear.scene_color = w + 1; c->Gclear.gA = w + 2; c->Gclear.gB = w + 3; c->Gclear.gC = w + 4;
        c->Gclear.gD = (uint2*)(w + 6); c->Gclear.overlay = w + 8;
    }
    {   // environment switches, read once
#ifdef ZR_DIAG       // work-skipping / A-B switches: diagnostic builds only (zeldaengine_amd.build.build(extra_flags=["-DZR_DIAG"]))
        const char* e;
        if ((e = getenv("ZR_DEBUG_SKIP"))) c->env_skip = (uint32_t)atoi(e);                 // 1: no pixel walk, 2: no triangle phase
        if ((e = getenv("ZR_DEBUG_SKIP_LIGHT"))) c->env_skip_light = (uint32_t)atoi(e);     // bits: 1 PCF, 2 lights, 4 reflection
        if ((e = getenv("ZR_LIGHT_LIST_MIN"))) c->env_light_list_min = atoi(e);
        c->env_no_empty_px = getenv("ZR_NO_EMPTY_PIXEL") != nullptr;
        c->env_serial = getenv("ZR_SERIAL_PASSES") != nullptr;      // same frame, one stream (= ZR_FLAG_SERIAL_PASSES)
        if ((e = getenv("ZR_SHADOW_BOX_CULL"))) c->env_shadow_box = atoi(e) != 0;
        if ((e = getenv("ZR_SHADOW_DEFER"))) c->env_shadow_defer = atoi(e) != 0;
#endif
    }
    ok &= hipHostMalloc((void**)&c->h_view_ring, sizeof(XkView) * zr_ctx::VIEW_RING, hipHostMallocDefault) == hipSuccess;
    for (auto& e : c->view_ev) ok &= hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    {   // The camera lane must not share a hardware queue with the host's stream (HIP multiplexes streams onto a few of them and
        // two streams on one queue run strictly one after the other).  Streams of different priority come from different queue
        // pools.  Which priority: the frame's period is the HOST's lane (lighting -> shadow pipeline), and since the camera lane lost its
        // two scans and two index passes per frame (round 6: tile buckets) it no longer fills the period - at the highest priority it
        // took from the host lane what it saved itself (5 300 Mpixel/s), at the lowest the host lane keeps its share (5 540; normal: 5 470).
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
#ifdef ZR_DIAG
        // experiment: CU partition between the lanes.  ZR_CU_MASK_CAM / ZR_CU_MASK_HOST = "lo-hi" CU ranges (of 256), or "xN:k" = the first k
        // CUs of every group of N; the host's stream is then the library's own masked stream (only when the caller sets none)
        auto make_masked = [&](const char* spec, hipStream_t* out) -> bool {
            uint32_t mask[8] = { 0 };
            int a = 0, b = 0;
            if (sscanf(spec, "x%d:%d", &a, &b) == 2 && a > 0) { for (int i = 0; i < 256; ++i) if (i % a < b) mask[i >> 5] |= 1u << (i & 31); }
            else if (sscanf(spec, "%d-%d", &a, &b) == 2) { for (int i = a; i <= b && i < 256; ++i) if (i >= 0) mask[i >> 5] |= 1u << (i & 31); }
            else return false;
            return hipExtStreamCreateWithCUMask(out, 8, mask) == hipSuccess;
        };
        const char* mc = getenv("ZR_CU_MASK_CAM"); const char* mh = getenv("ZR_CU_MASK_HOST");
        if (mh) { hipStream_t hs = nullptr; if (make_masked(mh, &hs)) { (void)hipStreamDestroy(c->own_stream); c->own_stream = hs; c->stream = hs; } }
        if (mc && make_masked(mc, &c->cam_s)) { }
        else
#endif
        ok &= hipStreamCreateWithPriority(&c->cam_s, hipStreamNonBlocking, least) == hipSuccess;
    }
    ok &= hipEventCreateWithFlags(&c->ev_cam, hipEventDisableTiming) == hipSuccess;
    ok &= hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    ok &= dev_alloc(&c->d_vis, n) == hipSuccess;
    if (ok) { zr_launch_fill64(c->d_vis, (unsigned long long)0x3F800000u << 32 | ZR_EMPTY_PRIM, n, c->stream); ok &= hipStreamSynchronize(c->stream) == hipSuccess; }
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->raster_blocks = (uint32_t)prop.multiProcessorCount * 12u; }      // k_tile's persistent grid (6 workgroups fit a CU: two rounds of them; A/B 4 / 6 / 8 / 12 / 16 / 32 per CU -> 5 133 / 5 250 / 5 294 / 5 344 / 5 327 / 5 277 Mpixel/s)
    c->shadow_blocks = c->raster_blocks / 12u * 8u;      // the shadow rasteriser's persistent grid stays at 8 per CU
    c->slow0_cap = std::max<uint32_t>(c->slow0_cap, 128u * c->sn_tiles);      // (a clipped triangle is listed once per tile of its meshlet)
    if (ok) ok &= dev_alloc(&c->d_slow0, 4ull * c->slow0_cap) == hipSuccess;
#ifdef ZR_DIAG
    if (const char* e = getenv("ZR_RASTER_BLOCKS")) c->raster_blocks = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("ZR_SHADOW_BLOCKS")) c->shadow_blocks = (uint32_t)std::max(1, atoi(e));
#endif
    {   // Hi-Z pyramid: level l = max depth per (8 << l)^2 pixel block
        size_t tot = 0;
        for (int l = 0; l < 4; ++l) { c->hiz.hw[l] = (c->W + (8u << l) - 1) / (8u << l); c->hiz.hh[l] = (c->H + (8u << l) - 1) / (8u << l); tot += (size_t)c->hiz.hw[l] * c->hiz.hh[l]; }
        c->hiz.fw = (c->W + 3u) / 4u; c->hiz.fh = (c->H + 3u) / 4u;
        tot += (size_t)c->hiz.fw * c->hiz.fh;
        ok &= dev_alloc(&c->d_hiz, tot) == hipSuccess;
        if (ok) ok &= hipMemset(c->d_hiz, 0, tot * sizeof(float)) == hipSuccess;      // texels over other ranks' regions stay 0 ("hidden")
        static_assert(ZR_TILE == 32 && ZR_SUPERTILE_SHIFT >= 1, "a 64 x 64 region of the pyramid must lie inside one super-tile");
        std::vector<uint32_t> regions;
        for (uint32_t ry = 0; ry < (c->H + 63u) / 64u; ++ry)
            for (uint32_t rx = 0; rx < (c->W + 63u) / 64u; ++rx)
                if (zr_tile_owner(rx * 2u, ry * 2u, c->cfg.tile_world) == c->cfg.tile_rank) regions.push_back(rx | ry << 16);
        c->n_hiz_regions = (uint32_t)regions.size();
        ok &= dev_alloc(&c->d_hiz_regions, regions.size()) == hipSuccess;
        if (ok && !regions.empty()) ok &= hipMemcpy(c->d_hiz_regions, regions.data(), regions.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        float* p = c->d_hiz;
        for (int l = 0; l < 4; ++l) { c->hiz.lvl[l] = p; p += (size_t)c->hiz.hw[l] * c->hiz.hh[l]; }
        c->hiz.fine = p;
    }
    if (ok) {
        // The runtime backs an event with a signal on its FIRST record and grows that pool in batches, which blocks the host for
        // milliseconds at unpredictable frames of a short run: record every event once now.
        for (auto& fr : c->evr) for (auto& e : fr) ok &= hipEventRecord(e, c->stream) == hipSuccess;
        for (auto& e : c->ev_end) ok &= hipEventRecord(e, c->stream) == hipSuccess;
        for (auto& e : c->view_ev) ok &= hipEventRecord(e, c->stream) == hipSuccess;
        // hipMemset of device memory is ordered on the NULL stream and need not be complete when it returns; the library's streams are
        // non-blocking ones (no implicit ordering with the null stream): nothing may be enqueued on them before the fills above are through
        ok &= hipDeviceSynchronize() == hipSuccess;
        ok &= hipStreamSynchronize(c->stream) == hipSuccess;
    }
    if (!ok) { zr_destroy(c); return ZR_ERR_DEVICE; }
    if (zr_set_cubemap(c, nullptr, 0) != ZR_OK) { zr_destroy(c); return ZR_ERR_DEVICE; }
    *out = c;
    return ZR_OK;
}

static void free_mesh_buffers(ZrMesh& m)
{
    dev_free(m.d_v); dev_free(m.d_rv); dev_free(m.d_rt); dev_free(m.d_idx); dev_free(m.d_meshlets); dev_free(m.d_mpos); dev_free(m.d_mbox); dev_free(m.d_mtri); dev_free(m.d_tri_meshlet);
    m.uploaded = false;
}

static void free_tri_bins(zr_ctx* c)
{
    dev_free(c->tb.sel); dev_free(c->tb.recA); dev_free(c->tb.recB); dev_free(c->tb.tile_base); dev_free(c->tb.tile_cap); dev_free(c->tb.cursor);
    dev_free(c->tb.over_tile); dev_free(c->tb.plan); dev_free(c->tb.over_cursor); dev_free(c->tb.unit_tab); dev_free(c->tb.n_units);
    dev_free(c->tb.slow); dev_free(c->tb.wave_culled);
    c->plan_valid = false;
}

static void free_scene(zr_ctx* c)
{
    for (auto& o : c->objects) { dev_free(o.d_inst); for (auto& t : o.d_tex) dev_free(t); }
    c->objects.clear();
    for (auto& m : c->meshes) {
        dev_free(m.d_v); dev_free(m.d_rv); dev_free(m.d_rt); dev_free(m.d_idx); dev_free(m.d_meshlets); dev_free(m.d_mpos); dev_free(m.d_mbox); dev_free(m.d_mtri); dev_free(m.d_tri_meshlet);
    }
    c->meshes.clear();
    c->profabs.clear();
    dev_free(c->d_objs); c->n_objs = 0; c->n_work = 0; c->scene_dirty = true;
}

extern "C" void zr_destroy(zr_ctx* c)
{
    if (!c) return;
    zr_livelink_stop(c);
    (void)hipSetDevice(c->device);
    (void)zr_sync_all(c);                      // including a geometry stage whose lighting pass never came
    zr_dist_destroy(c);
    free_scene(c);
    free_mesh_buffers(c->sky_mesh); dev_free(c->sky_obj.d_inst); for (auto& t : c->sky_obj.d_tex) dev_free(t); dev_free(c->d_bg);
    for (auto p : c->d_cube) if (p) (void)hipFree(p);
    for (int b = 0; b < 2; ++b) {
        GBufferPtrs& G = c->Gb[b];
        dev_free(G.depth); dev_free(G.scene_color); dev_free(G.gA); dev_free(G.gB); dev_free(G.gC); dev_free(G.gD); dev_free(G.overlay);
        dev_free(c->d_shadow_b[b]); dev_free(c->d_view_b[b]); dev_free(c->d_empty_b[b]); dev_free(c->d_prim_b[b]);
    }
    dev_free(c->d_color); dev_free(c->d_stats); dev_free(c->d_sstats); dev_free(c->d_lut); dev_free(c->d_unorm_lut); dev_free(c->d_sky_keys);
    dev_free(c->d_owned); dev_free(c->d_sowned); dev_free(c->d_tiles); dev_free(c->d_tile_map); dev_free(c->d_sowned_rank); dev_free(c->d_stile_map);
    for (auto& sc : c->sc) {
        dev_free(sc.tile_count); dev_free(sc.tile_offset); dev_free(sc.tile_cursor); dev_free(sc.chunk_offset);
        dev_free(sc.rects); dev_free(sc.bins); dev_free(sc.work); dev_free(sc.chunk_tab);
    }
    dev_free(c->d_clear_px);
    if (c->h_view_ring) (void)hipHostFree(c->h_view_ring);
    for (auto& e : c->view_ev) if (e) (void)hipEventDestroy(e);
    if (c->cam_s) (void)hipStreamDestroy(c->cam_s);
    if (c->ev_cam) (void)hipEventDestroy(c->ev_cam);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    dev_free(c->d_vis); dev_free(c->d_slow0);
    dev_free(c->d_pxrect); dev_free(c->d_zmin); dev_free(c->d_visflag[0]); dev_free(c->d_visflag[1]); dev_free(c->d_hiz); dev_free(c->d_hiz_regions);
    dev_free(c->d_spxrect); dev_free(c->d_szmin); dev_free(c->d_sflag);
    free_tri_bins(c);
    for (auto& fr : c->evr) for (auto& e : fr) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_end) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" const char* zr_last_error(const zr_ctx* c) { return c ? c->err.c_str() : "no context (no usable HIP device?)"; }

extern "C" int zr_tile_size(void) { return ZR_TILE; }

extern "C" uint32_t zr_tile_owner(uint32_t tx, uint32_t ty, uint32_t world)
{
    return world <= 1 ? 0u : ((tx >> ZR_SUPERTILE_SHIFT) + (ty >> ZR_SUPERTILE_SHIFT) * ZR_SUPERTILE_SKEW) % world;
}

static int zr_tile_partition_impl(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, uint32_t* owned, uint32_t* n_owned,
                                 uint32_t* slots_per_rank)
{
    if (!width || !height || !world || rank >= world || !n_owned || !slots_per_rank) return ZR_ERR_ARG;
    const uint32_t tx = (width + ZR_TILE - 1) / ZR_TILE, ty = (height + ZR_TILE - 1) / ZR_TILE;
    std::vector<uint32_t> counts(world, 0u);
    uint32_t n = 0;
    for (uint32_t t = 0; t < tx * ty; ++t) {
        const uint32_t o = zr_tile_owner(t % tx, t / tx, world);
        counts[o]++;
        if (o == rank) { if (owned) owned[n] = t; ++n; }
    }
    *n_owned = n; *slots_per_rank = 0;
    for (uint32_t k : counts) *slots_per_rank = std::max(*slots_per_rank, k);
    return ZR_OK;
}
extern "C" int zr_tile_partition(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, uint32_t* owned, uint32_t* n_owned, uint32_t* slots_per_rank)
{
    return zr_guard(nullptr, [&]() { return zr_tile_partition_impl(width, height, world, rank, owned, n_owned, slots_per_rank); });
}

extern "C" int zr_set_stream(zr_ctx* c, void* s)
{
    if (!c) return ZR_ERR_ARG;
    hipStream_t ns = s ? (hipStream_t)s : c->own_stream;
    // frames in flight are ordered by their place on the host's stream (frame_begin relies on it): a change of stream drains them
    if (ns != c->stream && c->rendered) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, zr_sync_all(c)); }
    c->stream = ns;
    return ZR_OK;
}

// ------------------------------------------------------------------------------------------------ scene

static int zr_mesh_create_impl(zr_ctx* c, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, uint32_t* mesh_id)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, v && idx && mesh_id && nv > 0 && ni > 0 && ni % 3 == 0);
    for (uint32_t i = 0; i < ni; ++i) if (idx[i] >= nv) return zr_fail(c, ZR_ERR_ARG, "index out of range");
    ZrMesh m;
    m.v.assign(v, v + nv); m.idx.assign(idx, idx + ni);
    c->meshes.push_back(std::move(m));
    *mesh_id = (uint32_t)c->meshes.size() - 1;
    return ZR_OK;
}
extern "C" int zr_mesh_create(zr_ctx* c, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, uint32_t* mesh_id)
{
    return zr_guard(c, [&]() { return zr_mesh_create_impl(c, v, nv, idx, ni, mesh_id); });
}

static int validate_meshlets(zr_ctx* c, const ZrMesh& m, const XkMeshlet* ml, uint32_t nm, size_t nmv, const uint32_t* mv,
                             size_t nmt, const uint8_t* mt)
{
    for (uint32_t i = 0; i < nm; ++i) {
        if (ml[i].VertexCount == 0 || ml[i].VertexCount > 64 || ml[i].TriangleCount == 0 || ml[i].TriangleCount > 128)
            return zr_fail(c, ZR_ERR_ARG, "meshlet exceeds 64 vertices / 128 triangles");
        if ((size_t)ml[i].VertexOffset + ml[i].VertexCount > nmv || (size_t)ml[i].TriangleOffset + 3u * ml[i].TriangleCount > nmt)
            return zr_fail(c, ZR_ERR_ARG, "meshlet range out of bounds");
        for (uint32_t k = 0; k < ml[i].VertexCount; ++k)
            if (mv[ml[i].VertexOffset + k] >= m.v.size()) return zr_fail(c, ZR_ERR_ARG, "meshlet vertex index out of range");
        for (uint32_t k = 0; k < 3u * ml[i].TriangleCount; ++k)
            if (mt[ml[i].TriangleOffset + k] >= ml[i].VertexCount) return zr_fail(c, ZR_ERR_ARG, "meshlet triangle corner out of range");
    }
    return ZR_OK;
}

static int zr_mesh_set_meshlets_impl(zr_ctx* c, uint32_t mesh_id, const XkMeshlet* ml, uint32_t nm,
                                    const uint32_t* mv, size_t nmv, const uint8_t* mt, size_t nmt)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, mesh_id < c->meshes.size() && ml && nm && mv && mt);
    ZrMesh& m = c->meshes[mesh_id];
    if (m.uploaded) return zr_fail(c, ZR_ERR_STATE, "mesh already in use by a rendered scene");
    int rc = validate_meshlets(c, m, ml, nm, nmv, mv, nmt, mt);
    if (rc) return rc;
    // CreateMeshVertexBuffers<XkMeshIndirect> (ZE:4733-4756): the draw becomes "meshlet by meshlet"; rebuild the
    // draw-order index buffer accordingly so primitive ids follow meshlet order.
    m.ms.meshlets.assign(ml, ml + nm); m.ms.mverts.assign(mv, mv + nmv); m.ms.mtris.assign(mt, mt + nmt);
    // Every cull trusts the bounding sphere to enclose the meshlet's vertices (and the cone to describe its triangles): a record
    // whose sphere does not is recomputed (ZM:149-166 fills them from meshopt_computeMeshletBounds, so a sound file never is).
    for (uint32_t i = 0; i < nm; ++i) {
        XkMeshlet& d = m.ms.meshlets[i];
        bool ok = std::isfinite(d.BoundsRadius) && d.BoundsRadius >= 0.0f;
        for (uint32_t k = 0; ok && k < d.VertexCount; ++k) {
            const float* q = m.v[mv[d.VertexOffset + k]].Position;
            const double dx = (double)q[0] - d.BoundsCenter[0], dy = (double)q[1] - d.BoundsCenter[1], dz = (double)q[2] - d.BoundsCenter[2];
            if (!(std::sqrt(dx * dx + dy * dy + dz * dz) <= (double)d.BoundsRadius * (1.0 + 1e-5) + 1e-30)) ok = false;
        }
        if (!ok) {
            XkMeshlet b = d;
            zr_meshlet_bounds(m.v.data(), mv + d.VertexOffset, d.VertexCount, mt + d.TriangleOffset, d.TriangleCount, &b);
            d = b;
        }
    }
    m.ms.tri_order.clear(); m.idx.clear();
    uint32_t base = 0;
    for (uint32_t i = 0; i < nm; ++i) {
        XkMeshlet& d = m.ms.meshlets[i];
        d.BindlessContext = base;
        for (uint32_t t = 0; t < d.TriangleCount; ++t) {
            for (int k = 0; k < 3; ++k) m.idx.push_back(mv[d.VertexOffset + mt[d.TriangleOffset + 3u * t + (uint32_t)k]]);
            m.ms.tri_order.push_back(base + t);
        }
        base += d.TriangleCount;
    }
    m.has_meshlets = true;
    return ZR_OK;
}
extern "C" int zr_mesh_set_meshlets(zr_ctx* c, uint32_t mesh_id, const XkMeshlet* ml, uint32_t nm, const uint32_t* mv, size_t nmv, const uint8_t* mt, size_t nmt)
{
    return zr_guard(c, [&]() { return zr_mesh_set_meshlets_impl(c, mesh_id, ml, nm, mv, nmv, mt, nmt); });
}

static int zr_mesh_build_meshlets_impl(zr_ctx* c, uint32_t mesh_id, uint32_t max_v, uint32_t max_t, float cone_weight)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, mesh_id < c->meshes.size());
    if (max_v == 0) max_v = 64;
    if (max_t == 0) max_t = 124;
    ARGCHK(c, max_v >= 3 && max_v <= 64 && max_t >= 1 && max_t <= 128);
    ZrMesh& m = c->meshes[mesh_id];
    if (m.uploaded) return zr_fail(c, ZR_ERR_STATE, "mesh already in use by a rendered scene");
    zr_build_meshlets(m.v.data(), (uint32_t)m.v.size(), m.idx.data(), (uint32_t)m.idx.size(), max_v, max_t, cone_weight, &m.ms);
    m.has_meshlets = true;
    return ZR_OK;
}
extern "C" int zr_mesh_build_meshlets(zr_ctx* c, uint32_t mesh_id, uint32_t max_v, uint32_t max_t, float cone_weight)
{
    return zr_guard(c, [&]() { return zr_mesh_build_meshlets_impl(c, mesh_id, max_v, max_t, cone_weight); });
}

// Context-free form of the clusteriser: the ZeldaMeshlet tool's BuildMeshlets (ZM:132-172) as a library call.  Pure host
// code (runs without a GPU).  Pass NULL outputs to query the sizes.  tri_order[k] = index-buffer triangle of slot k.
static int zr_meshlets_build_impl(const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, uint32_t max_v, uint32_t max_t,
                                 float cone_weight, XkMeshlet* ml, uint32_t* nm, uint32_t* mv, size_t* nmv, uint8_t* mt, size_t* nmt,
                                 uint32_t* tri_order)
{
    if (!v || !idx || !nm || !nmv || !nmt || nv == 0 || ni == 0 || ni % 3) return ZR_ERR_ARG;
    if (max_v == 0) max_v = 64;
    if (max_t == 0) max_t = 124;
    if (max_v < 3 || max_v > 64 || max_t < 1 || max_t > 128) return ZR_ERR_ARG;
    for (uint32_t i = 0; i < ni; ++i) if (idx[i] >= nv) return ZR_ERR_ARG;
    ZrMeshletSet ms;
    zr_build_meshlets(v, nv, idx, ni, max_v, max_t, cone_weight, &ms);
    *nm = (uint32_t)ms.meshlets.size(); *nmv = ms.mverts.size(); *nmt = ms.mtris.size();
    if (ml) { memcpy(ml, ms.meshlets.data(), ms.meshlets.size() * sizeof(XkMeshlet)); for (uint32_t i = 0; i < *nm; ++i) ml[i].BindlessContext = 0; }
    if (mv) memcpy(mv, ms.mverts.data(), ms.mverts.size() * 4);
    if (mt) memcpy(mt, ms.mtris.data(), ms.mtris.size());
    if (tri_order) memcpy(tri_order, ms.tri_order.data(), ms.tri_order.size() * 4);
    return ZR_OK;
}
extern "C" int zr_meshlets_build(const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, uint32_t max_v, uint32_t max_t, float cone_weight, XkMeshlet* ml, uint32_t* nm, uint32_t* mv, size_t* nmv, uint8_t* mt, size_t* nmt, uint32_t* tri_order)
{
    return zr_guard(nullptr, [&]() { return zr_meshlets_build_impl(v, nv, idx, ni, max_v, max_t, cone_weight, ml, nm, mv, nmv, mt, nmt, tri_order); });
}

static int zr_mesh_get_meshlets_impl(zr_ctx* c, uint32_t mesh_id, XkMeshlet* ml, uint32_t* nm, uint32_t* mv, size_t* nmv,
                                    uint8_t* mt, size_t* nmt)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, mesh_id < c->meshes.size());
    const ZrMesh& m = c->meshes[mesh_id];
    if (nm) *nm = (uint32_t)m.ms.meshlets.size();
    if (nmv) *nmv = m.ms.mverts.size();
    if (nmt) *nmt = m.ms.mtris.size();
    if (ml) { memcpy(ml, m.ms.meshlets.data(), m.ms.meshlets.size() * sizeof(XkMeshlet));
              for (size_t i = 0; i < m.ms.meshlets.size(); ++i) ml[i].BindlessContext = 0; }
    if (mv) memcpy(mv, m.ms.mverts.data(), m.ms.mverts.size() * 4);
    if (mt) memcpy(mt, m.ms.mtris.data(), m.ms.mtris.size());
    return ZR_OK;
}
extern "C" int zr_mesh_get_meshlets(zr_ctx* c, uint32_t mesh_id, XkMeshlet* ml, uint32_t* nm, uint32_t* mv, size_t* nmv, uint8_t* mt, size_t* nmt)
{
    return zr_guard(c, [&]() { return zr_mesh_get_meshlets_impl(c, mesh_id, ml, nm, mv, nmv, mt, nmt); });
}

int zr_material_prepare(zr_ctx* c, const zr_material* mat, ZrMaterialHost* out)
{
    for (int t = 0; t < 7; ++t) {
        const uint8_t* px = kDefaultTexel[t];
        out->image[t].clear(); out->w[t] = out->h[t] = 1;
        if (mat && mat->tex[t].rgba8) {
            const zr_image& im = mat->tex[t];
            if (im.width == 0 || im.height == 0 || im.width > 16384 || im.height > 16384) return zr_fail(c, ZR_ERR_ARG, "bad material image size");
            const size_t n = (size_t)im.width * im.height * 4;
            bool constant = true;
            for (size_t i = 4; i < n; ++i) if (im.rgba8[i] != im.rgba8[i & 3]) { constant = false; break; }
            px = im.rgba8;
            if (!constant) { out->image[t].assign(im.rgba8, im.rgba8 + n); out->w[t] = im.width; out->h[t] = im.height; }
        }
        out->texel[t] = (uint32_t)px[0] | (uint32_t)px[1] << 8 | (uint32_t)px[2] << 16 | (uint32_t)px[3] << 24;
    }
    for (int k = 0; k < 3; ++k) out->bc_linear[k] = zr_srgb_decode8((out->texel[0] >> (8 * k)) & 255u);
    return ZR_OK;
}

static int idx_clamp_h(float f, int hi) { f = fminf(fmaxf(f, 0.0f), (float)hi); return (int)f; }

// RHIGenerateMipmaps (ZE:6348-6433): level l+1 = vkCmdBlitImage(LINEAR) of level l at half size; mipLevels =
// floor(log2(max(w, h))) + 1 (ZE:6887).  Filtered on decoded values (sRGB for the base-colour slot, ZE:5878), re-encoded.
static void build_mip_chain(const zr_ctx* c, const std::vector<uint8_t>& img, uint32_t w, uint32_t h, bool srgb,
                            std::vector<uint8_t>* chain, uint32_t* levels)
{
    uint32_t m = w > h ? w : h;
    uint32_t nl = 1; while (m > 1) { m >>= 1; nl++; }
    *levels = nl;
    *chain = img;
    size_t src_off = 0;
    uint32_t sw = w, sh = h;
    for (uint32_t l = 1; l < nl; ++l) {
        const uint32_t dw = sw > 1 ? sw >> 1 : 1, dh = sh > 1 ? sh >> 1 : 1;
        const size_t dst_off = chain->size();
        chain->resize(dst_off + (size_t)dw * dh * 4);
        const uint8_t* src = chain->data() + src_off;
        uint8_t* dst = chain->data() + dst_off;
        const float kx = (float)sw / (float)dw, ky = (float)sh / (float)dh;
        for (uint32_t y = 0; y < dh; ++y) for (uint32_t x = 0; x < dw; ++x) {
            const float fu = fmaf((float)x + 0.5f, kx, -0.5f), fv = fmaf((float)y + 0.5f, ky, -0.5f);
            const float fx = floorf(fu), fy = floorf(fv), a = fu - fx, b = fv - fy;
            const int x0 = idx_clamp_h(fx, (int)sw - 1), x1 = idx_clamp_h(fx + 1.0f, (int)sw - 1);
            const int y0 = idx_clamp_h(fy, (int)sh - 1), y1 = idx_clamp_h(fy + 1.0f, (int)sh - 1);
            const uint8_t* p00 = src + ((size_t)y0 * sw + x0) * 4; const uint8_t* p10 = src + ((size_t)y0 * sw + x1) * 4;
            const uint8_t* p01 = src + ((size_t)y1 * sw + x0) * 4; const uint8_t* p11 = src + ((size_t)y1 * sw + x1) * 4;
            for (int ch = 0; ch < 4; ++ch) {
                const bool sr = srgb && ch < 3;
                const float t00 = sr ? c->lut[p00[ch]] : (float)p00[ch] / 255.0f, t10 = sr ? c->lut[p10[ch]] : (float)p10[ch] / 255.0f;
                const float t01 = sr ? c->lut[p01[ch]] : (float)p01[ch] / 255.0f, t11 = sr ? c->lut[p11[ch]] : (float)p11[ch] / 255.0f;
                const float top = fmaf(a, t10 - t00, t00), bot = fmaf(a, t11 - t01, t01);
                const float v = fmaf(b, bot - top, top);
                dst[((size_t)y * dw + x) * 4 + ch] = sr ? srgb_encode8(v) : (uint8_t)zr_unorm(v, 255.0f);
            }
        }
        src_off = dst_off; sw = dw; sh = dh;
    }
}

int zr_object_add_internal(zr_ctx* c, uint32_t mesh_id, const ZrMaterialHost& mat, const XkInstanceData* inst, uint32_t n_inst)
{
    ZrSceneObject o;
    o.mesh = mesh_id; o.instanced = n_inst > 0; o.n_inst = n_inst ? n_inst : 1;
    memcpy(o.texel, mat.texel, sizeof o.texel); memcpy(o.bc_linear, mat.bc_linear, sizeof o.bc_linear);
    if (n_inst) o.inst.assign(inst, inst + n_inst);
    HIPCHK(c, hipSetDevice(c->device));
    auto cleanup = [&]() { dev_free(o.d_inst); for (auto& t : o.d_tex) dev_free(t); };
    std::vector<uint8_t> chains[7];
    int lead = -1;                                          // first slot that holds an image
    for (int t = 0; t < 7; ++t) {
        if (mat.image[t].empty()) continue;
        std::vector<uint8_t>& chain = chains[t]; uint32_t levels = 1;
        build_mip_chain(c, mat.image[t], mat.w[t], mat.h[t], t == 0, &chain, &levels);
        hipError_t e = dev_alloc_image(&o.d_tex[t], chain.size());
        if (e == hipSuccess) e = hipMemcpy(o.d_tex[t], chain.data(), chain.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { cleanup(); return zr_fail(c, ZR_ERR_DEVICE, hipGetErrorString(e)); }
        o.tex_w[t] = mat.w[t]; o.tex_h[t] = mat.h[t]; o.tex_levels[t] = levels;
        if (lead < 0) lead = t;
        else if (mat.w[t] != mat.w[lead] || mat.h[t] != mat.h[lead]) o.mixed_sizes = true;
    }
    if (lead >= 0 && !o.mixed_sizes) {
        // The packed material: per texel of the (common) mip chain the 13 channels BaseScene.frag reads, 16 B (ZR_PK_*); constant slots
        // put their constant there (the resolve takes those from the draw record, not from here).
        static const struct { int slot, ch, n; } kPack[7] = { {0, ZR_PK_BC, 3}, {1, ZR_PK_ME, 1}, {2, ZR_PK_RO, 1}, {3, ZR_PK_NO, 3}, {4, ZR_PK_AO, 1}, {5, ZR_PK_EM, 3}, {6, ZR_PK_MS, 1} };
        const size_t n_texels = chains[lead].size() / 4;
        std::vector<uint8_t> pk(n_texels * 16, 0);
        for (const auto& k : kPack) {
            const bool image = !chains[k.slot].empty();
            const uint8_t* src = image ? chains[k.slot].data() : nullptr;
            for (size_t i = 0; i < n_texels; ++i)
                for (int ch = 0; ch < k.n; ++ch)
                    pk[i * 16 + (size_t)k.ch + (size_t)ch] = image ? src[i * 4 + (size_t)ch] : (uint8_t)(mat.texel[k.slot] >> (8 * ch));
        }
        hipError_t e = dev_alloc_image(&o.d_tex[7], pk.size());
        if (e == hipSuccess) e = hipMemcpy(o.d_tex[7], pk.data(), pk.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { cleanup(); return zr_fail(c, ZR_ERR_DEVICE, hipGetErrorString(e)); }
        o.tex_w[7] = mat.w[lead]; o.tex_h[7] = mat.h[lead]; o.tex_levels[7] = o.tex_levels[lead];
    }
    { hipError_t e = dev_alloc(&o.d_inst, o.n_inst); if (e != hipSuccess) { cleanup(); return zr_fail(c, ZR_ERR_DEVICE, hipGetErrorString(e)); } }
    XkInstanceData* d_raw = nullptr;
    if (n_inst) {
        hipError_t e = dev_alloc(&d_raw, n_inst);
        if (e == hipSuccess) e = hipMemcpyAsync(d_raw, inst, sizeof(XkInstanceData) * n_inst, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { dev_free(d_raw); cleanup(); return zr_fail(c, ZR_ERR_DEVICE, hipGetErrorString(e)); }
    }
    zr_launch_instance_prep(d_raw, o.d_inst, o.n_inst, o.instanced ? 1u : 0u, c->stream);
    hipError_t e = zr_sync_all(c);
    dev_free(d_raw);
    if (e != hipSuccess) { cleanup(); return zr_fail(c, ZR_ERR_DEVICE, hipGetErrorString(e)); }
    c->objects.push_back(std::move(o));
    c->scene_dirty = true;
    return ZR_OK;
}

static int zr_object_add_impl(zr_ctx* c, uint32_t mesh_id, const zr_material* mat, const XkInstanceData* inst, uint32_t n_inst)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, mesh_id < c->meshes.size() && (n_inst == 0 || inst));
    ZrMaterialHost m;
    int rc = zr_material_prepare(c, mat, &m);
    if (rc) return rc;
    return zr_object_add_internal(c, mesh_id, m, inst, n_inst);
}
extern "C" int zr_object_add(zr_ctx* c, uint32_t mesh_id, const zr_material* mat, const XkInstanceData* inst, uint32_t n_inst)
{
    return zr_guard(c, [&]() { return zr_object_add_impl(c, mesh_id, mat, inst, n_inst); });
}

// Capacities of the triangle-record arrays (chunks of 256 records) and of the clipped-triangle list, for hosts that size them themselves
// (0 = the default: 16 records per meshlet-instance, at least 32 Mi; 2^18 triangles).  Takes effect at the next frame.
extern "C" int zr_set_limits(zr_ctx* c, uint32_t record_chunks, uint32_t slow_triangles)
{
    if (!c) return ZR_ERR_ARG;
    c->limit_record_chunks = record_chunks; c->limit_slow_triangles = slow_triangles;
    c->work_capacity = 0; c->scene_dirty = true;          // the pools are re-made by the next frame
    return ZR_OK;
}

extern "C" int zr_set_bucket_share(zr_ctx* c, uint32_t percent)
{
    if (!c || percent < 1u || percent > 100u) return ZR_ERR_ARG;
    c->bucket_pct = percent;          // (k_plan's argument from the next plan on; a frame that overflows its buckets is the same frame)
    return ZR_OK;
}

extern "C" int zr_scene_clear(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, zr_sync_all(c));
    free_scene(c);
    return ZR_OK;
}

template <typename T> static hipError_t upload(T** d, const std::vector<T>& h)
{
    hipError_t e = dev_alloc(d, h.size());
    if (e != hipSuccess) return e;
    return h.empty() ? hipSuccess : hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

static int upload_mesh(zr_ctx* c, ZrMesh& m)
{
    if (m.uploaded) return ZR_OK;
    if (!m.has_meshlets) {
        zr_build_meshlets(m.v.data(), (uint32_t)m.v.size(), m.idx.data(), (uint32_t)m.idx.size(), 64, 124, 0.2f, &m.ms);
        m.has_meshlets = true;
    }
    // whole-mesh bounding sphere (centroid + max distance)
    double cx = 0, cy = 0, cz = 0;
    for (auto& v : m.v) { cx += v.Position[0]; cy += v.Position[1]; cz += v.Position[2]; }
    cx /= (double)m.v.size(); cy /= (double)m.v.size(); cz /= (double)m.v.size();
    double r = 0;
    for (auto& v : m.v) { double dx = v.Position[0] - cx, dy = v.Position[1] - cy, dz = v.Position[2] - cz; r = std::max(r, std::sqrt(dx * dx + dy * dy + dz * dz)); }
    m.center[0] = (float)cx; m.center[1] = (float)cy; m.center[2] = (float)cz; m.radius = (float)(r * 1.0001) + 1e-30f;
    // flatten for the kernels: one coalesced 16 B load per meshlet vertex, one 8 B load per meshlet triangle
    std::vector<float4> mpos(m.ms.mverts.size());
    for (size_t i = 0; i < mpos.size(); ++i) {
        const float* p = m.v[m.ms.mverts[i]].Position;
        mpos[i] = make_float4(p[0], p[1], p[2], 1.0f);
    }
    std::vector<uint2> mtri(m.ms.tri_order.size());
    for (const XkMeshlet& ml : m.ms.meshlets)
        for (uint32_t t = 0; t < ml.TriangleCount; ++t) {
            const uint8_t* tp = m.ms.mtris.data() + ml.TriangleOffset + 3u * t;
            mtri[ml.BindlessContext + t] = make_uint2((uint32_t)tp[0] | (uint32_t)tp[1] << 8 | (uint32_t)tp[2] << 16,
                                                      m.ms.tri_order[ml.BindlessContext + t]);
        }
    std::vector<ZrRVertex> rv(m.v.size());             // the resolve's vertex record: position + uv + the normalised normal
    for (size_t i = 0; i < rv.size(); ++i) {
        const XkVertex& x = m.v[i];
        const zf3 n = zr_normalize(zr3(x.Normal[0], x.Normal[1], x.Normal[2]));
        rv[i] = ZrRVertex{ x.Position[0], x.Position[1], x.Position[2], x.TexCoord[0], n.x, n.y, n.z, x.TexCoord[1] };
    }
    // ... and the same records per TRIANGLE CORNER in draw order (96 bytes a triangle): the resolve reaches a pixel's three corners from the
    // primitive id in one round trip instead of two (index, then vertex) - the kernel waits for its chain of dependent loads, not for arithmetic
    std::vector<ZrRVertex> rt(std::max<size_t>(1, m.idx.size()));
    for (size_t i = 0; i < m.idx.size(); ++i) rt[i] = rv[m.idx[i]];
    HIPCHK(c, upload(&m.d_rt, rt));
    HIPCHK(c, upload(&m.d_v, m.v)); HIPCHK(c, upload(&m.d_rv, rv)); HIPCHK(c, upload(&m.d_idx, m.idx)); HIPCHK(c, upload(&m.d_meshlets, m.ms.meshlets));
    // draw-order triangle -> meshlet (the resolve marks the meshlet-instances that own a pixel)
    std::vector<uint32_t> tri_meshlet(std::max<size_t>(1, m.idx.size() / 3), 0u);
    for (size_t mi = 0; mi < m.ms.meshlets.size(); ++mi) {
        const XkMeshlet& ml = m.ms.meshlets[mi];
        for (uint32_t t = 0; t < ml.TriangleCount; ++t) {
            const uint32_t tri = m.ms.tri_order[ml.BindlessContext + t];
            if (tri < tri_meshlet.size()) tri_meshlet[tri] = (uint32_t)mi;
        }
    }
    std::vector<float4> mbox(2 * std::max<size_t>(1, m.ms.meshlets.size()), make_float4(0.0f, 0.0f, 0.0f, 0.0f));   // object-space box per meshlet
    for (size_t mi = 0; mi < m.ms.meshlets.size(); ++mi) {
        const XkMeshlet& ml = m.ms.meshlets[mi];
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t v = 0; v < ml.VertexCount; ++v) {
            const float4& q = mpos[ml.VertexOffset + v];
            const float e[3] = { q.x, q.y, q.z };
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], e[a]); hi[a] = std::max(hi[a], e[a]); }      // (a NaN coordinate drops out: k_geom sees it)
        }
        mbox[2 * mi] = make_float4(lo[0], lo[1], lo[2], 0.0f); mbox[2 * mi + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    HIPCHK(c, upload(&m.d_mpos, mpos)); HIPCHK(c, upload(&m.d_mbox, mbox)); HIPCHK(c, upload(&m.d_mtri, mtri)); HIPCHK(c, upload(&m.d_tri_meshlet, tri_meshlet));
    m.uploaded = true;
    return ZR_OK;
}

// CreateEngineScene's GPU half (ZE:4140-4284): meshlets, buffers, draw table in the reference's draw order
static int finalize_scene(zr_ctx* c)
{
    if (!c->scene_dirty) return ZR_OK;
    HIPCHK(c, zr_sync_all(c));
    for (auto& o : c->objects) { int rc = upload_mesh(c, c->meshes[o.mesh]); if (rc) return rc; }
    const bool sky = c->sky_set && c->sky_enabled;
    if (sky) { int rc = upload_mesh(c, c->sky_mesh); if (rc) return rc; }
    std::vector<ZrObject> tab;
    uint64_t work = 0, prim = 0, inst_total = 0;
    auto emit = [&](const ZrSceneObject& o, const ZrMesh& m, uint32_t flags) {
        ZrObject d; memset(&d, 0, sizeof d);
        d.verts = m.d_v; d.rverts = m.d_rv; d.rtris = m.d_rt; d.indices = m.d_idx; d.meshlets = m.d_meshlets; d.mpos = m.d_mpos; d.mbox = m.d_mbox; d.mtri = m.d_mtri; d.tri_meshlet = m.d_tri_meshlet;
        d.inst = o.d_inst;
        d.n_meshlets = (uint32_t)m.ms.meshlets.size(); d.n_tris = (uint32_t)(m.idx.size() / 3);
        d.n_inst = o.n_inst; d.instanced = o.instanced; d.flags = flags;
        d.work_base = (uint32_t)work; d.prim_base = (uint32_t)prim; d.inst_base = (uint32_t)inst_total;
        inst_total += d.n_inst;
        memcpy(d.texel, o.texel, sizeof d.texel); memcpy(d.bc_linear, o.bc_linear, sizeof d.bc_linear);
        for (int t = 0; t < 7; ++t)
            for (int ch = 0; ch < 4; ++ch) {
                const uint32_t v8 = (o.texel[t] >> (8 * ch)) & 255u;
                d.texc[t][ch] = (t == 0 && ch < 3) ? c->lut[v8] : (float)v8 / 255.0f;
            }
        for (int t = 0; t < 7; ++t) { d.tex[t].data = o.d_tex[t]; d.tex[t].w = o.tex_w[t]; d.tex[t].h = o.tex_h[t]; d.tex[t].levels = o.tex_levels[t]; d.tex[t]._pad = 0; }
        d.packed.data = o.d_tex[7]; d.packed.w = o.tex_w[7]; d.packed.h = o.tex_h[7]; d.packed.levels = o.tex_levels[7]; d.packed._pad = 0;
        memcpy(d.mesh_center, m.center, sizeof d.mesh_center); d.mesh_radius = m.radius;
        // BaseScene.frag on constant slots, once per draw instead of once per pixel (the kernels' own arithmetic: zr_math.h)
        for (int t = 0; t < 7; ++t) if (!o.d_tex[t]) d.const_slots |= 1u << t;
        const zf3 ts = zr_tangent_space_normal(zr3(d.texc[3][0], d.texc[3][1], d.texc[3][2]));
        d.ts_const[0] = ts.x; d.ts_const[1] = ts.y; d.ts_const[2] = ts.z;
        d.c_scene_color = zr_unorm(d.texc[5][0], 255.0f) | zr_unorm(d.texc[5][1], 255.0f) << 8 | zr_unorm(d.texc[5][2], 255.0f) << 16 | zr_unorm(d.texc[6][0], 255.0f) << 24;
        d.c_gB = zr_unorm(d.texc[1][0], 255.0f) | zr_unorm(1.0f, 255.0f) << 8 | zr_unorm(fmaxf(0.01f, d.texc[2][0]), 255.0f) << 16 | 255u << 24;
        d.c_gC = zr_unorm(d.texc[0][0], 255.0f) | zr_unorm(d.texc[0][1], 255.0f) << 8 | zr_unorm(d.texc[0][2], 255.0f) << 16 | zr_unorm(d.texc[4][0], 255.0f) << 24;
        work += (uint64_t)d.n_meshlets * d.n_inst; prim += (uint64_t)d.n_tris * d.n_inst;
        tab.push_back(d);
    };
    for (int pass = 0; pass < 2; ++pass)                 // non-instanced draws, then instanced draws (ZE:3445-3476)
        for (auto& o : c->objects)
            if ((int)o.instanced == pass) emit(o, c->meshes[o.mesh], 0u);
    // The skydome is the table's last record but no work item of the shadow or the deferred-scene pass: it is drawn after the lighting
    // quad (ZE:3681-3691), depth-tested against the scene and colour only - k_sky_tiles + the resolve.
    const uint64_t scene_work = work, scene_inst = inst_total;
    if (sky) emit(c->sky_obj, c->sky_mesh, ZR_OBJ_SKY);
    if (work >= 0xFFFFFFFFull || prim >= 0xFFFFFFFFull) return zr_fail(c, ZR_ERR_OVERFLOW, "scene exceeds 2^32 meshlet-instances or primitives");
    dev_free(c->d_objs);
    HIPCHK(c, upload(&c->d_objs, tab));
    c->n_objs = (uint32_t)tab.size(); c->n_work = (uint32_t)scene_work; c->n_inst_total = (uint32_t)scene_inst;
    c->sky_object = sky ? (uint32_t)tab.size() - 1u : 0u;
    if (sky && !c->d_sky_keys) HIPCHK(c, dev_alloc(&c->d_sky_keys, (size_t)c->W * c->H));
    if (c->n_work > c->work_capacity) {
        for (auto& sc : c->sc) { dev_free(sc.rects); dev_free(sc.bins); dev_free(sc.work); dev_free(sc.chunk_tab); }
        dev_free(c->d_pxrect); dev_free(c->d_zmin); dev_free(c->d_visflag[0]); dev_free(c->d_visflag[1]);
        dev_free(c->d_spxrect); dev_free(c->d_szmin); dev_free(c->d_sflag);
        // (a failed allocation below returns with scene_dirty still set and work_capacity 0: the next frame tries again instead of
        // launching on freed buffers)
        const uint32_t cap_w = c->n_work;
        c->work_capacity = 0;
        const uint64_t cap = std::max<uint64_t>(1u << 20, 8ull * c->n_work);
        c->bin_capacity = (uint32_t)std::min<uint64_t>(cap, 0x3FFFFFFFull);
        c->chunk_capacity = c->bin_capacity / ZR_CHUNK + std::max(c->n_tiles, c->sn_tiles) + 1u;
        for (auto& sc : c->sc) {
            HIPCHK(c, dev_alloc(&sc.rects, cap_w));
            HIPCHK(c, dev_alloc(&sc.work, cap_w));
            HIPCHK(c, dev_alloc(&sc.bins, c->bin_capacity));
            HIPCHK(c, dev_alloc(&sc.chunk_tab, c->chunk_capacity));
        }
        // triangle-binned camera pass: triangle records (32 B) live in per-tile BUCKETS of two 16-byte planes, laid out every frame by
        // k_plan from the previous frame's per-tile counts; what lies behind the last bucket is the frame's overflow region (what a tile gets
        // beyond its bucket).  Sized from the scene: 16 records per meshlet-instance, at least 32 Mi - 1 GB of 288 reserved, touched as far as a frame
        // needs.  Planes that run full are reported like a bin overflow (zr_set_limits sizes them: 256 records per "chunk").
        free_tri_bins(c);
        c->tb.n_waves = 8192; c->tb.slow_cap = 1u << 18;
        // (16 per meshlet-instance: the frame after a camera cut at config 4 puts ~ 60 M records - 5 per meshlet-instance - into the 64
        // sections of the overflow region, unevenly; with 8 the fullest section ran over.  6 GB of 288 at 1 M instances.)
        uint64_t n_rec = std::min<uint64_t>(std::max<uint64_t>(32ull << 20, 16ull * c->n_work), 0x3FFFFFFFull);
        if (c->limit_record_chunks) n_rec = 256ull * c->limit_record_chunks;      // zr_set_limits (a host sizing the planes; the overflow tests)
        if (c->limit_slow_triangles) c->tb.slow_cap = std::max(2u, c->limit_slow_triangles);
        c->tb.n_rec = (uint32_t)n_rec; c->tb.bucket_max = (uint32_t)(n_rec - n_rec / 8u);
        c->tb.n_tiles = c->n_tiles;
        c->tb.unit_cap = c->tb.bucket_max / (ZR_TCHUNK * ZR_TBATCHES) + 2u * c->n_tiles + 1u;
        HIPCHK(c, dev_alloc(&c->tb.sel, cap_w));
        HIPCHK(c, dev_alloc(&c->tb.recA, (size_t)n_rec));
        HIPCHK(c, dev_alloc(&c->tb.recB, (size_t)n_rec));
        HIPCHK(c, dev_alloc(&c->tb.over_tile, (size_t)n_rec)); HIPCHK(c, dev_alloc(&c->tb.plan, 2));
        HIPCHK(c, dev_alloc(&c->tb.tile_base, c->n_tiles)); HIPCHK(c, dev_alloc(&c->tb.tile_cap, c->n_tiles));
        HIPCHK(c, dev_alloc(&c->tb.cursor, (size_t)2 * c->n_tiles * ZR_TSTRIDE));
        HIPCHK(c, dev_alloc(&c->tb.over_cursor, 2 * ZR_OVER_SECTIONS)); HIPCHK(c, dev_alloc(&c->tb.n_units, 1));
        HIPCHK(c, dev_alloc(&c->tb.unit_tab, c->tb.unit_cap));
        // (no plan yet: every bucket is empty - the first frame counts before it draws, see gbuffer_pass)
        HIPCHK(c, hipMemset(c->tb.tile_base, 0, (size_t)c->n_tiles * 4)); HIPCHK(c, hipMemset(c->tb.tile_cap, 0, (size_t)c->n_tiles * 4));
        HIPCHK(c, hipMemset(c->tb.cursor, 0, (size_t)2 * c->n_tiles * ZR_TSTRIDE * 4));
        HIPCHK(c, hipMemset(c->tb.over_cursor, 0, 2 * ZR_OVER_SECTIONS * 4)); HIPCHK(c, hipMemset(c->tb.n_units, 0, 4)); HIPCHK(c, hipMemset(c->tb.plan, 0, 8));
        HIPCHK(c, dev_alloc(&c->tb.wave_culled, c->tb.n_waves));
        HIPCHK(c, dev_alloc(&c->tb.slow, 4ull * c->tb.slow_cap));
        HIPCHK(c, dev_alloc(&c->d_pxrect, cap_w)); HIPCHK(c, dev_alloc(&c->d_zmin, cap_w));
        HIPCHK(c, dev_alloc(&c->d_visflag[0], cap_w)); HIPCHK(c, dev_alloc(&c->d_visflag[1], cap_w));
        HIPCHK(c, hipMemset(c->d_visflag[0], 0, cap_w)); HIPCHK(c, hipMemset(c->d_visflag[1], 0, cap_w));      // (no frame's stamp is 0)
        HIPCHK(c, dev_alloc(&c->d_spxrect, cap_w)); HIPCHK(c, dev_alloc(&c->d_szmin, cap_w)); HIPCHK(c, dev_alloc(&c->d_sflag, cap_w));
        // (the fills above sit on the null stream; the frame that follows runs on non-blocking streams: a fill that landed after that
        // frame's k_plan would wipe the plan - every record then overflows into sections of capacity 0)
        HIPCHK(c, hipDeviceSynchronize());
        c->work_capacity = cap_w;              // every buffer is there
    }
    c->any_images = c->mixed_images = false;
    for (const ZrObject& d : tab) for (int t = 0; t < 7; ++t) if (d.tex[t].data) c->any_images = true;
    for (const auto& o : c->objects) if (o.mixed_sizes) c->mixed_images = true;      // (the skydome's one image is sampled by itself)
    c->vis_history = false;         // work item numbering changed: last frame's visibility says nothing about this scene
    c->plan_valid = false;          // ... and neither do its per-tile record counts: the next frame counts before it draws (tri_raster)
    if (c->n_work) HIPCHK(c, hipMemsetAsync(c->d_sflag, 1, c->n_work, c->stream));      // shadow pass: everything is drawn in the first launch
    c->sflag_history = false;
    c->list_valid[0] = c->list_valid[1] = false;      // ... and neither do the passes' work lists
    c->scene_dirty = false;
    return ZR_OK;
}

// ------------------------------------------------------------------------------------------------ skydome + background

static int upload_texture(zr_ctx* c, const zr_image* tex, bool srgb, uint8_t** d, uint32_t* w, uint32_t* h, uint32_t* levels)
{
    if (tex->width == 0 || tex->height == 0 || tex->width > 16384 || tex->height > 16384) return zr_fail(c, ZR_ERR_ARG, "bad image size");
    std::vector<uint8_t> img(tex->rgba8, tex->rgba8 + (size_t)tex->width * tex->height * 4), chain;
    build_mip_chain(c, img, tex->width, tex->height, srgb, &chain, levels);
    HIPCHK(c, dev_alloc_image(d, chain.size()));
    HIPCHK(c, hipMemcpy(*d, chain.data(), chain.size(), hipMemcpyHostToDevice));
    *w = tex->width; *h = tex->height;
    return ZR_OK;
}

static int zr_set_skydome_impl(zr_ctx* c, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const zr_image* tex)
{
    if (!c) return ZR_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, zr_sync_all(c));
    free_mesh_buffers(c->sky_mesh); c->sky_mesh = ZrMesh();
    dev_free(c->sky_obj.d_inst); for (auto& t : c->sky_obj.d_tex) dev_free(t);
    c->sky_obj = ZrSceneObject(); c->sky_set = false; c->scene_dirty = true;
    if (!tex || !tex->rgba8) return ZR_OK;
    ARGCHK(c, v && idx && nv > 0 && ni > 0 && ni % 3 == 0);
    for (uint32_t i = 0; i < ni; ++i) if (idx[i] >= nv) return zr_fail(c, ZR_ERR_ARG, "index out of range");
    c->sky_mesh.v.assign(v, v + nv); c->sky_mesh.idx.assign(idx, idx + ni);
    ZrSceneObject& o = c->sky_obj;
    o.mesh = 0; o.instanced = false; o.n_inst = 1;
    for (int t = 0; t < 7; ++t) o.texel[t] = 0xFFFFFFFFu;
    o.bc_linear[0] = o.bc_linear[1] = o.bc_linear[2] = 1.0f;
    int rc = upload_texture(c, tex, true, &o.d_tex[0], &o.tex_w[0], &o.tex_h[0], &o.tex_levels[0]);   // sRGB by default, ZE:5860
    if (rc) return rc;
    HIPCHK(c, dev_alloc(&o.d_inst, 1));
    zr_launch_instance_prep(nullptr, o.d_inst, 1, 0u, c->stream);
    HIPCHK(c, zr_sync_all(c));
    c->sky_set = true;
    return ZR_OK;
}
extern "C" int zr_set_skydome(zr_ctx* c, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const zr_image* tex)
{
    return zr_guard(c, [&]() { return zr_set_skydome_impl(c, v, nv, idx, ni, tex); });
}

static int zr_set_background_impl(zr_ctx* c, const zr_image* tex)
{
    if (!c) return ZR_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, zr_sync_all(c));
    dev_free(c->d_bg); c->bg_set = false;
    if (!tex || !tex->rgba8) return ZR_OK;
    int rc = upload_texture(c, tex, true, &c->d_bg, &c->bg_w, &c->bg_h, &c->bg_levels);
    if (rc) return rc;
    c->bg_set = true;
    return ZR_OK;
}
extern "C" int zr_set_background(zr_ctx* c, const zr_image* tex)
{
    return zr_guard(c, [&]() { return zr_set_background_impl(c, tex); });
}

extern "C" int zr_set_sky_flags(zr_ctx* c, int sky, int bg)
{
    if (!c) return ZR_ERR_ARG;
    if ((sky != 0) != c->sky_enabled) c->scene_dirty = true;
    c->sky_enabled = sky != 0; c->bg_enabled = bg != 0;
    return ZR_OK;
}

// ------------------------------------------------------------------------------------------------ cubemap

static int zr_set_cubemap_impl(zr_ctx* c, const uint8_t* const faces[6], uint32_t dim)
{
    if (!c) return ZR_ERR_ARG;
    static const uint8_t grey[4] = { 127, 127, 127, 255 };
    if (!faces) dim = 1;
    ARGCHK(c, dim > 0 && dim <= 16384);
    if (faces) for (int f = 0; f < 6; ++f) ARGCHK(c, faces[f] != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, zr_sync_all(c));
    for (auto p : c->d_cube) if (p) (void)hipFree(p);
    c->d_cube.clear();
    uint32_t levels = 1; for (uint32_t d = dim; d > 1; d >>= 1) levels++;      // floor(log2(dim)) + 1, ZE:6887
    if (levels > 16) return zr_fail(c, ZR_ERR_ARG, "cubemap too large");
    std::vector<std::vector<uint8_t>> lv(levels);
    const size_t fsz = (size_t)dim * dim * 4;
    lv[0].resize(fsz * 6);
    for (int f = 0; f < 6; ++f) { if (faces) memcpy(lv[0].data() + fsz * f, faces[f], fsz); else memcpy(lv[0].data() + fsz * f, grey, 4); }
    uint32_t d = dim;
    for (uint32_t l = 1; l < levels; ++l) {        // RHIGenerateMipmaps: vkCmdBlitImage LINEAR from level l-1 (2x2 box, linear light)
        const uint32_t nd = d > 1 ? d >> 1 : 1;
        lv[l].resize((size_t)nd * nd * 4 * 6);
        for (int f = 0; f < 6; ++f) {
            const uint8_t* src = lv[l - 1].data() + (size_t)d * d * 4 * f;
            uint8_t* dst = lv[l].data() + (size_t)nd * nd * 4 * f;
            for (uint32_t y = 0; y < nd; ++y) for (uint32_t x = 0; x < nd; ++x) {
                const uint32_t x0 = 2 * x, x1 = (2 * x + 1 < d) ? 2 * x + 1 : d - 1, y0 = 2 * y, y1 = (2 * y + 1 < d) ? 2 * y + 1 : d - 1;
                const uint8_t* p00 = src + ((size_t)y0 * d + x0) * 4; const uint8_t* p10 = src + ((size_t)y0 * d + x1) * 4;
                const uint8_t* p01 = src + ((size_t)y1 * d + x0) * 4; const uint8_t* p11 = src + ((size_t)y1 * d + x1) * 4;
                for (int ch = 0; ch < 3; ++ch) {
                    const float a = (c->lut[p00[ch]] + c->lut[p10[ch]]) + (c->lut[p01[ch]] + c->lut[p11[ch]]);
                    dst[((size_t)y * nd + x) * 4 + ch] = srgb_encode8(a * 0.25f);
                }
                const uint32_t al = (uint32_t)p00[3] + p10[3] + p01[3] + p11[3];
                dst[((size_t)y * nd + x) * 4 + 3] = (uint8_t)((al + 2) >> 2);
            }
        }
        d = nd;
    }
    memset(&c->cube, 0, sizeof c->cube);
    for (uint32_t l = 0; l < levels; ++l) {
        uint8_t* p = nullptr;
        HIPCHK(c, upload(&p, lv[l]));
        c->d_cube.push_back(p); c->cube.levels[l] = p;
    }
    c->cube_dim = dim; c->cube_levels = levels;
    c->view.LightsCount[3] = (int32_t)levels;       // CubemapMaxMips, ZE:4308
    c->view_dirty = true;
    return ZR_OK;
}
extern "C" int zr_set_cubemap(zr_ctx* c, const uint8_t* const faces[6], uint32_t dim)
{
    return zr_guard(c, [&]() { return zr_set_cubemap_impl(c, faces, dim); });
}

// ------------------------------------------------------------------------------------------------ uniforms

static float radiansf(float deg) { return deg * 0.01745329251994329576923690768489f; }
static void perspective_rh_zo(float fovy, float aspect, float zn, float zf, float* m)
{
    const float t = tanf(fovy / 2.0f);
    memset(m, 0, 64);
    m[0] = 1.0f / (aspect * t); m[5] = 1.0f / t; m[10] = zf / (zn - zf); m[11] = -1.0f; m[14] = -(zf * zn) / (zf - zn);
}
static void look_at_rh(zf3 eye, zf3 center, zf3 up, float* m)
{
    const zf3 f = zr_normalize_ieee(center - eye);          // (glm on the host: IEEE, not the shaders' inversesqrt)
    const zf3 s = zr_normalize_ieee(zr_cross(f, up));
    const zf3 u = zr_cross(s, f);
    m[0] = s.x; m[4] = s.y; m[8] = s.z; m[1] = u.x; m[5] = u.y; m[9] = u.z; m[2] = -f.x; m[6] = -f.y; m[10] = -f.z;
    m[3] = 0; m[7] = 0; m[11] = 0; m[12] = -zr_dot(s, eye); m[13] = -zr_dot(u, eye); m[14] = zr_dot(f, eye); m[15] = 1.0f;
}
static void rotate_z(float angle, float* m)
{
    const float cs = cosf(angle), sn = sinf(angle);
    memset(m, 0, 64);
    m[0] = cs; m[1] = sn; m[4] = -sn; m[5] = cs; m[10] = cs + (1.0f - cs); m[15] = 1.0f;
}

// UpdateWorld (ZE:4294-4308) + UpdateUniformBuffer (ZE:4585-4664) in game mode (editor bars = 0, ZE:4575-4579)
extern "C" int zr_update_uniforms(zr_ctx* c, const zr_camera* cam, const XkLight* dir, uint32_t n_dir, const XkLight* point,
                                  uint32_t n_point, const XkLight* spot, uint32_t n_spot, float roll_stage, float roll_light, float time)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, cam && n_dir <= XK_MAX_DIRECTIONAL_LIGHTS_NUM && n_point <= XK_MAX_POINT_LIGHTS_NUM && n_spot <= XK_MAX_SPOT_LIGHTS_NUM);
    ARGCHK(c, (n_dir == 0 || dir) && (n_point == 0 || point) && (n_spot == 0 || spot));
    // both geometry passes' kernel arguments are fixed when a frame begins (zr_render_shadow / zr_render_geometry): uniforms set between the
    // stages of a frame would reach its lighting pass only
    if (c->stage != 0) return zr_fail(c, ZR_ERR_STATE, "zr_update_uniforms between the stages of a frame (finish it with zr_render_lighting first)");
    XkView* V = &c->view;
    c->view_dirty = true;
    for (uint32_t i = 0; i < n_dir; ++i) V->DirectionalLights[i] = dir[i];
    for (uint32_t i = 0; i < n_point; ++i) V->PointLights[i] = point[i];
    for (uint32_t i = 0; i < n_spot; ++i) V->SpotLights[i] = spot[i];
    V->LightsCount[0] = (int32_t)n_dir; V->LightsCount[1] = (int32_t)n_point; V->LightsCount[2] = (int32_t)n_spot;
    V->LightsCount[3] = (int32_t)c->cube_levels;

    const zf3 pos = zr3(cam->Position[0], cam->Position[1], cam->Position[2]);
    const zf3 look = zr3(cam->Lookat[0], cam->Lookat[1], cam->Lookat[2]);
    const zf3 up = zr3(0.0f, 0.0f, 1.0f);
    const zf3 lightPos = zr3(V->DirectionalLights[0].Position[0], V->DirectionalLights[0].Position[1], V->DirectionalLights[0].Position[2]);
    float l2w[16], sview[16], sproj[16], cview[16], cproj[16];
    rotate_z(roll_stage, l2w);
    look_at_rh(lightPos, zr3(0.0f, 0.0f, 0.0f), up, sview);
    perspective_rh_zo(radiansf(cam->FOV), 1.0f, cam->zNear, cam->zFar, sproj);
    sproj[5] *= -1.0f;
    look_at_rh(pos, look, up, cview);
    perspective_rh_zo(radiansf(cam->FOV), (float)c->W / (float)c->H, cam->zNear, cam->zFar, cproj);
    memcpy(c->cam.Model, l2w, 64); memcpy(c->cam.View, cview, 64); memcpy(c->cam.Proj, cproj, 64);
    c->cam.Proj[5] *= -1.0f;
    zr_mat4_mul(cproj, cview, V->ViewProjSpace);
    zr_mat4_mul(sproj, sview, V->ShadowmapSpace);
    memcpy(V->LocalToWorld, l2w, 64);
    V->CameraInfo[0] = pos.x; V->CameraInfo[1] = pos.y; V->CameraInfo[2] = pos.z; V->CameraInfo[3] = cam->FOV;
    V->ViewportInfo[0] = (float)c->W; V->ViewportInfo[1] = (float)c->H; V->ViewportInfo[2] = 0.0f; V->ViewportInfo[3] = 0.0f;
    const uint32_t N = n_point;
    for (uint32_t i = 0; i < N; ++i) {           // point lights ride a spiral, JSON positions are overwritten (ZE:4637-4646)
        const float deg = ((float)i / (float)N) * 360.0f - roll_light * 100.0f;
        const float distance = ((float)i / (float)N) * 5.0f + 2.5f;
        V->PointLights[i].Position[0] = sinf(radiansf(deg)) * distance;
        V->PointLights[i].Position[1] = cosf(radiansf(deg)) * distance;
        V->PointLights[i].Position[2] = 1.5f;
        V->PointLights[i].Position[3] = 1.0f;
    }
    V->Time = time; V->zNear = cam->zNear; V->zFar = cam->zFar;
    memcpy(c->shadow.Model, l2w, 64); memcpy(c->shadow.View, sview, 64); memcpy(c->shadow.Proj, sproj, 64);
    c->frame_valid = true;
    return ZR_OK;
}

extern "C" int zr_set_frame(zr_ctx* c, const XkUniformBufferMVP* cam, const XkUniformBufferMVP* sh, const XkView* v)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, cam && sh && v);
    ARGCHK(c, v->LightsCount[0] >= 0 && v->LightsCount[0] <= XK_MAX_DIRECTIONAL_LIGHTS_NUM && v->LightsCount[1] >= 0 &&
              v->LightsCount[1] <= XK_MAX_POINT_LIGHTS_NUM);
    if (c->stage != 0) return zr_fail(c, ZR_ERR_STATE, "zr_set_frame between the stages of a frame (finish it with zr_render_lighting first)");
    c->cam = *cam; c->shadow = *sh; c->view = *v; c->view_dirty = true;
    c->frame_valid = true;
    return ZR_OK;
}
extern "C" int zr_get_frame(zr_ctx* c, XkUniformBufferMVP* cam, XkUniformBufferMVP* sh, XkView* v)
{
    if (!c) return ZR_ERR_ARG;
    if (cam) *cam = c->cam;
    if (sh) *sh = c->shadow;
    if (v) *v = c->view;
    return ZR_OK;
}
extern "C" int zr_set_debug_view(zr_ctx* c, uint32_t s) { if (!c) return ZR_ERR_ARG; c->debug_view = s; return ZR_OK; }

// Forward variant (SH/Base.frag): the resolve additionally keeps each pixel's winning primitive id (one more plane per GBuffer copy,
// allocated on first use), and the lighting step runs k_forward on those instead of k_lighting on the GBuffer.
static int zr_set_shading_impl(zr_ctx* c, uint32_t mode)
{
    if (!c) return ZR_ERR_ARG;
    if (mode != ZR_SHADING_DEFERRED && mode != ZR_SHADING_FORWARD) return zr_fail(c, ZR_ERR_ARG, "zr_set_shading: unknown mode");
    if (c->stage != 0) return zr_fail(c, ZR_ERR_STATE, "zr_set_shading between the stages of a frame");
    if (mode == c->shading) return ZR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // frames in flight read / write the planes this call swaps in or out
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->cam_s) HIPCHK(c, hipStreamSynchronize(c->cam_s));
    const size_t n = (size_t)c->W * c->H;
    for (int b = 0; b < 2; ++b) {
        if (mode == ZR_SHADING_FORWARD && !c->d_prim_b[b]) {
            if (dev_alloc(&c->d_prim_b[b], n) != hipSuccess) return zr_fail(c, ZR_ERR_DEVICE, "zr_set_shading: out of device memory");
            HIPCHK(c, hipMemset(c->d_prim_b[b], 0xFF, n * 4));
            HIPCHK(c, hipDeviceSynchronize());      // (a null-stream fill; the frames run on non-blocking streams)
        }
        c->Gb[b].prim = mode == ZR_SHADING_FORWARD ? c->d_prim_b[b] : nullptr;
    }
    c->shading = mode;
    return ZR_OK;
}
extern "C" int zr_set_shading(zr_ctx* c, uint32_t mode)
{
    return zr_guard(c, [&]() { return zr_set_shading_impl(c, mode); });
}

static bool finite16(const float* m) { for (int i = 0; i < 16; ++i) if (!std::isfinite(m[i])) return false; return true; }
static bool rigid3(const float* m)     // upper 3x3 orthonormal, det > 0, last row 0 0 0 1
{
    const zf3 a = zr3(m[0], m[1], m[2]), b = zr3(m[4], m[5], m[6]), cc = zr3(m[8], m[9], m[10]);
    const float e = 1e-3f;
    if (fabsf(zr_dot(a, a) - 1) > e || fabsf(zr_dot(b, b) - 1) > e || fabsf(zr_dot(cc, cc) - 1) > e) return false;
    if (fabsf(zr_dot(a, b)) > e || fabsf(zr_dot(a, cc)) > e || fabsf(zr_dot(b, cc)) > e) return false;
    if (zr_dot(zr_cross(a, b), cc) <= 0) return false;
    return m[3] == 0 && m[7] == 0 && m[11] == 0 && m[15] == 1;
}

// Builds the kernarg block of one geometry pass.  Returns false when the pass cannot produce a fragment
// (non-finite PVM: every vertex is non-finite and every triangle is discarded).
static bool build_pass(const zr_ctx* c, const XkUniformBufferMVP& u, int mode, ZrPass* P)
{
    memset(P, 0, sizeof *P);
    float pv[16];
    zr_mat4_mul(u.Proj, u.View, pv);
    zr_mat4_mul(pv, u.Model, P->PVM);           // proj * view * model, left to right
    memcpy(P->M, u.Model, 64);
    P->mode = (uint32_t)mode;
    P->W = mode == ZR_MODE_SHADOW ? c->SD : c->W; P->H = mode == ZR_MODE_SHADOW ? c->SD : c->H;
    P->hw = 0.5f * (float)P->W; P->hh = 0.5f * (float)P->H;
    P->tiles_x = mode == ZR_MODE_SHADOW ? c->stiles_x : c->tiles_x; P->tiles_y = mode == ZR_MODE_SHADOW ? c->stiles_y : c->tiles_y;
    P->tile_rank = mode == ZR_MODE_SHADOW ? c->stile_rank : c->cfg.tile_rank; P->tile_world = mode == ZR_MODE_SHADOW ? c->stile_world : c->cfg.tile_world;
    P->inst_rank = mode == ZR_MODE_SHADOW ? c->shadow_rank : 0; P->inst_world = mode == ZR_MODE_SHADOW ? c->shadow_world : 1;
    P->images = !c->any_images ? 0u : c->mixed_images ? 2u : 1u;     // 1: every material with images has the packed form
    P->n_objects = c->n_objs; P->n_work = c->n_work; P->n_inst_total = c->n_inst_total; P->bin_capacity = c->bin_capacity;
    // the instance-level pre-pass pays for itself on big scenes; small ones go straight to a lane per meshlet-instance - unless this
    // context owns a share of the tiles (below): then the pre-pass leaves a RANK-LOCAL list and the culls walk 1 / N of the scene
    P->use_worklist = c->n_inst_total >= 65536u ? 1u : 0u;
    P->debug_skip = c->env_skip;
    if (ZR_TILE == 32) {      // (both passes: the shadow pass uses it for the instance-level "no texel centre" reject)
        // sphere_bounds() needs clip.x = p00 * x_view, clip.y = p11 * y_view, clip.z = p10 * z_view + p14, clip.w = -z_view and
        // view-space radii = object radii
        const float* pr = u.Proj;
        const bool centred = pr[1] == 0 && pr[2] == 0 && pr[3] == 0 && pr[4] == 0 && pr[6] == 0 && pr[7] == 0 && pr[8] == 0 && pr[9] == 0 &&
                             pr[11] == -1.0f && pr[12] == 0 && pr[13] == 0 && pr[15] == 0 && std::isfinite(pr[0]) && std::isfinite(pr[5]) &&
                             pr[0] != 0 && pr[5] != 0 && std::isfinite(pr[10]) && std::isfinite(pr[14]) && pr[14] < 0;
        zr_mat4_mul(u.View, u.Model, P->VM);
        P->p00 = pr[0]; P->p11 = pr[5];
        P->pz_a = -pr[10]; P->pz_b = pr[14];       // z_view = -d: (p10 * -d + p14) / d
        P->sphere_ok = (centred && rigid3(u.Model) && rigid3(u.View) && finite16(P->VM)) ? 1u : 0u;
        // (both passes: the camera pass against the frame's tiles, the shadow pass against the map's when the map is owned by tiles)
        P->rect_cull = (P->sphere_ok && P->tile_world > 1 && !(c->cfg.flags & ZR_FLAG_NO_RECT_CULL)) ? 1u : 0u;
        if (P->rect_cull) P->use_worklist = 1u;
    }
    {
        static const float ident[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        P->m_identity = memcmp(u.Model, ident, 64) == 0 ? 1u : 0u;      // bitwise: a -0 entry would not do
    }
    if (!finite16(P->PVM)) return false;
    // frustum planes of proj*view in world space (sphere centres are taken to world space by M in the kernel)
    bool fr_ok = !(c->cfg.flags & ZR_FLAG_NO_FRUSTUM_CULL) && finite16(u.Model);
    if (fr_ok) {
        float rows[6][4];
        for (int k = 0; k < 4; ++k) {
            const float r0 = pv[k * 4 + 0], r1 = pv[k * 4 + 1], r2 = pv[k * 4 + 2], r3 = pv[k * 4 + 3];
            rows[0][k] = r3 + r0; rows[1][k] = r3 - r0; rows[2][k] = r3 + r1; rows[3][k] = r3 - r1; rows[4][k] = r2; rows[5][k] = r3 - r2;
        }
        for (int i = 0; i < 6 && fr_ok; ++i) {
            const float l = sqrtf(rows[i][0] * rows[i][0] + rows[i][1] * rows[i][1] + rows[i][2] * rows[i][2]);
            if (!(l > 1e-20f) || !std::isfinite(l)) { fr_ok = false; break; }
            for (int k = 0; k < 4; ++k) P->planes[i][k] = rows[i][k] / l;
        }
    }
    const bool m_rigid = rigid3(u.Model);
    if (m_rigid) P->m_scale = 1.002f;
    else { float s = 0; for (int cidx = 0; cidx < 3; ++cidx) for (int r = 0; r < 3; ++r) s += u.Model[cidx * 4 + r] * u.Model[cidx * 4 + r]; P->m_scale = sqrtf(s) * 1.002f; }
    if (!std::isfinite(P->m_scale)) fr_ok = false;
    P->frustum_ok = fr_ok ? 1u : 0u;
    // cone culling needs: rigid model and view, a perspective projection with its eye at the view origin, and the
    // engine's handedness (Proj[0][0] > 0, Proj[1][1] < 0 after the Vulkan y-flip, ZE:4624) so that CCW = front
    bool cone = mode == ZR_MODE_GBUFFER && !(c->cfg.flags & ZR_FLAG_NO_CONE_CULL) && m_rigid && rigid3(u.View);
    const float* pr = u.Proj;
    cone = cone && pr[3] == 0 && pr[7] == 0 && pr[11] == -1.0f && pr[15] == 0 && pr[1] == 0 && pr[2] == 0 && pr[4] == 0 && pr[6] == 0 &&
           pr[12] == 0 && pr[13] == 0 && pr[0] > 0 && pr[5] < 0;
    if (cone) {     // eye = -R^T t
        const float* v = u.View;
        P->cam_pos[0] = -(v[0] * v[12] + v[1] * v[13] + v[2] * v[14]);
        P->cam_pos[1] = -(v[4] * v[12] + v[5] * v[13] + v[6] * v[14]);
        P->cam_pos[2] = -(v[8] * v[12] + v[9] * v[13] + v[10] * v[14]);
    }
    P->cone_ok = cone ? 1u : 0u;
    return true;
}

// ------------------------------------------------------------------------------------------------ the frame

// cull -> count -> scan -> fill -> raster of one pass.  Z.phase selects the share of the camera pass drawn (0 = all of it).
static void bin_and_raster(zr_ctx* c, const ZrPass& P, const ZrHiz& Z, int slot, uint32_t n_tiles, hipStream_t s)
{
    const zr_ctx::Scratch& sc = c->sc[slot ? 1 : 0];
    ZrDevStats* st = slot ? c->d_stats : c->d_sstats;        // the shadow pipeline (slot 0) has a block of its own
    zr_launch_bin_count(P, sc.work, sc.rects, sc.tile_count, Z, st, slot, s);
    // A rank that owns a share of the shadow MAP (zr_set_shadow_tiles, four ranks or more) has a small pass beside a camera lane that is as
    // busy as ever: units of 128 entries on half the persistent grid leave that lane more of the machine (a rank of eight at config 4:
    // 0.779 -> 0.752 ms; 32 / 16 entries: 0.88 / 1.05 ms; 256: 0.754).  Units only get bigger here: the chunk table's capacity holds.
    const uint32_t chunk = (slot == 0 && c->stile_world >= 4u) ? 2u * ZR_CHUNK : ZR_CHUNK;
    zr_launch_scan(sc.tile_count, sc.tile_offset, sc.tile_cursor, sc.chunk_offset, sc.chunk_tab, c->chunk_capacity, n_tiles, c->bin_capacity, st, slot, s, chunk);
    zr_launch_bin_fill(P, c->d_objs, sc.work, sc.rects, sc.tile_offset, sc.tile_cursor, sc.bins, Z, st, slot, s);
}
// One round of the triangle-binned camera pass: which meshlet-instances (k_select: timed with the cull), then their triangles as
// records (k_geom), the records' places per tile (k_scan, k_index) and the tile kernel: those four are what the meshlet-binned
// path's one raster launch does, and are timed as the raster.
static void tri_select(zr_ctx* c, const ZrPass& P, const ZrHiz& Z, int slot, hipStream_t s)
{
    zr_launch_select(P, c->d_objs, c->sc[1].work, c->sc[1].rects, Z, c->tb, c->d_stats, slot, s);
}
// One round of the triangle-binned camera pass: triangles -> records in their tiles' buckets (k_geom), tile raster (k_tile).  The buckets were
// laid out by the previous frame's k_plan; `count_first`: there is no usable plan (first frame of a scene, or the last plan was made by a
// two-round frame and this round draws everything) - k_geom runs once more ahead of the round, counting only, and k_plan lays the buckets
// out from that.
static void tri_raster(zr_ctx* c, const ZrPass& P, const ZrHiz& Z, int slot, hipStream_t s, bool last, bool count_first)
{
    if (P.n_work == 0) return;          // nothing to draw: the pass is its clear
    if (count_first) {
        zr_launch_geom(P, Z, c->tb, c->d_stats, slot, true, s);
        zr_launch_plan(c->tb, c->d_owned, c->n_owned, c->d_stats, true, c->bucket_pct, s);      // (exact: the round that follows appends what was just counted)
    }
    zr_launch_geom(P, Z, c->tb, c->d_stats, slot, false, s);
    // (the frame's last round also draws the slow triangles of both rounds: k_tile<LAST>)
    zr_launch_tile(P, c->tb, c->d_stats, slot, c->d_vis, c->raster_blocks, s, last, c->d_owned, c->n_owned);
}
static void raster(zr_ctx* c, const ZrPass& P, const ZrHiz& Z, int slot, hipStream_t s, int stage = 0)
{
    const bool shadow = slot == 0;
    const zr_ctx::Scratch& sc = c->sc[shadow ? 0 : 1];
    const bool defer = shadow && c->env_shadow_defer && c->d_slow0 != nullptr;
    zr_launch_raster_chunks(P, c->d_objs, sc.chunk_tab, sc.bins, shadow ? c->d_sstats : c->d_stats, slot, c->d_vis,
                            (uint32_t*)(c->d_shadow_ext ? c->d_shadow_ext : c->d_shadow),
                            shadow ? (c->stile_world >= 4u ? c->shadow_blocks / 2u : c->shadow_blocks) : c->raster_blocks, Z, s,
                            defer ? c->d_slow0 : nullptr, c->slow0_cap, c->d_sowned, c->sn_tiles, stage);
}

static inline float* shadow_buf(zr_ctx* c) { return c->d_shadow_ext ? c->d_shadow_ext : c->d_shadow; }

// The frame in three stages so that a multi-GPU host can put collectives between them (zeldaengine_amd/dist.py):
//   zr_render_shadow    shadow pass (ZE:3239-3393) of this rank's share of the instances
//   zr_render_gbuffer   deferred-scene pass (ZE:3417-3480): cull + bin + raster + resolve of the owned tiles
//   zr_render_lighting  deferred-lighting pass (ZE:3531-3540) [+ skydome / background overlay]
// zr_render = all three.
// Start of a frame on stream s: pick this frame's copies of the double-buffered resources, make s wait until the lighting pass
// that last read them (two frames ago) and the previous frame's shadow pipeline (it shares d_stats) are done, reset the
// statistics, upload the uniforms if this copy does not hold them yet.
static int frame_begin(zr_ctx* c, hipStream_t s)
{
    if (!c->frame_valid) return zr_fail(c, ZR_ERR_STATE, "no frame uniforms: call zr_update_uniforms or zr_set_frame first");
    if (c->stage != 0) return zr_fail(c, ZR_ERR_STATE, "zr_render_shadow out of order");
    if (c->debug_view == 9u && c->cfg.tile_world > 1u)
        return zr_fail(c, ZR_ERR_STATE, "debug view 9 (GBufferVis) re-samples the whole GBuffer: not available on a tile-partitioned context");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = finalize_scene(c);
    if (rc) return rc;
    if (c->view.LightsCount[3] != (int32_t)c->cube_levels) { c->view.LightsCount[3] = (int32_t)c->cube_levels; c->view_dirty = true; }
    const int par = (int)(c->frame_no & 1u);
    c->G = c->Gb[par]; c->d_shadow = c->d_shadow_b[par]; c->d_view = c->d_view_b[par]; c->d_empty_rgba = c->d_empty_b[par];
    // (two lanes: this frame's copies of the double-buffered resources were last read by the lighting pass of two frames ago, on the
    // host's stream.  Nothing else ties the lanes together here: the shadow pipeline keeps statistics of its own)
    if (s != c->stream && c->frame_no >= 2) HIPCHK(c, hipStreamWaitEvent(s, c->ev_end[(c->frame_no - 2) % zr_ctx::END_RING], 0));
    c->timing_now = c->timing_interval != 0 && c->frame_no % c->timing_interval == 0;     // pass events cost ~6 us of stream bubble each
    hipEvent_t* ev = c->timing_now ? c->evr[c->sample_no % zr_ctx::EV_RING] : nullptr;
    if (ev) HIPCHK(c, hipEventRecord(ev[0], s));
    if (c->view_dirty) { c->view_version++; c->view_dirty = false; }
    // the frame's two geometry passes; a pass's work list on the device is rebuilt only when its block or the scene changed
    uint32_t rebuild = 0;
    for (int slot = 0; slot < 2; ++slot) {
        ZrPass& P = c->pass[slot];
        c->pass_live[slot] = build_pass(c, slot == 0 ? c->shadow : c->cam, slot == 0 ? ZR_MODE_SHADOW : ZR_MODE_GBUFFER, &P);
        if (!c->pass_live[slot]) P.n_work = 0;      // no finite vertex: the pass is its clear
        c->list_reuse[slot] = P.use_worklist && P.n_work != 0 && c->list_valid[slot] && memcmp(&c->list_key[slot], &P, sizeof P) == 0 &&
                              !(c->cfg.flags & ZR_FLAG_NO_LIST_REUSE);
        // (the list counts as standing only once its k_cull_instances has been enqueued: shadow_pass / gbuffer_pass set list_valid)
        if (P.use_worklist && P.n_work != 0 && !c->list_reuse[slot]) { rebuild |= 1u << slot; c->list_key[slot] = P; c->list_valid[slot] = false; }
    }
    const XkView* src = nullptr;
    uint32_t k = 0;
    if (c->view_uploaded[par] != c->view_version) {        // pinned ring slot: reused only after the kernel that read it last has run
        k = c->view_slot++ % zr_ctx::VIEW_RING;
        HIPCHK(c, hipEventSynchronize(c->view_ev[k]));
        memcpy(&c->h_view_ring[k], &c->view, sizeof(XkView));
        src = &c->h_view_ring[k];
    }
    c->list_rebuild_mask = rebuild;
    // zeroes the camera lane's statistics (the sticky overflow latch survives) and - when the camera list is rebuilt - its length; uploads
    // XkView.  The SHADOW list's length lives in the shadow pipeline's block and is reset on that pipeline's own stream (shadow_pass):
    // the previous frame's shadow pipeline may still be walking it while this kernel runs on the camera lane.
    zr_launch_frame_begin(c->d_stats, src, c->d_view, rebuild & 2u, s);
    if (src) { HIPCHK(c, hipEventRecord(c->view_ev[k], s)); c->view_uploaded[par] = c->view_version; }
    return ZR_OK;
}

// shadow pass (ZE:3239-3393) of this rank's share of the instances, on stream s
static int shadow_pass(zr_ctx* c, hipStream_t s)
{
    hipEvent_t* ev = c->timing_now ? c->evr[c->sample_no % zr_ctx::EV_RING] : nullptr;
    const ZrPass& P = c->pass[0];      // (built by frame_begin)
    c->last_work[0] = P.n_work;
    // clear depth 1.0 (ZE:3248): the previous frame's lighting pass already did it for the internal double-buffered map
    const int spar = (int)(c->frame_no & 1u);
    if (c->d_shadow_ext || !c->shadow_cleared[spar]) zr_launch_fill32((uint32_t*)shadow_buf(c), 0x3F800000u, (size_t)c->SD * c->SD, s);
    c->shadow_cleared[spar] = false;
    ZrHiz Z; memset(&Z, 0, sizeof Z);
    // occlusion culling (k_shadow_occlusion): the first launch draws what was not hidden last frame, the rest is tested against the map.
    // It pays when casters pile up behind each other: the test + the late launch cost what a quarter of config 3's rasteriser does
    // (0.1 meshlet-instances per texel: 25 % hidden, frame 2.7 % slower); the same spheres at 0.21 / 0.31 / 0.52 per texel: frame 2 /
    // 8 / 10.5 % faster (tools/occlusion_threshold.py); 1 M instances (10 per texel): 10 % - on by itself from one per five texels.
    bool occl = !(c->cfg.flags & ZR_FLAG_NO_SHADOW_OCCLUSION) && P.n_work != 0 && ZR_TILE == 32 && c->SD >= 4u &&
                ((c->cfg.flags & ZR_FLAG_SHADOW_OCCLUSION) || 5ull * P.n_work >= (uint64_t)c->SD * c->SD);
#ifdef ZR_DIAG
    if (!c->env_shadow_box) {      // (A/B only: the exact cull always rebuilds its list, in the camera lane's block)
        occl = false; c->list_valid[0] = false;
        if (P.use_worklist) zr_launch_fill32(&c->d_stats->n_vis_work[0], 0u, 1, s);
        zr_launch_cull(P, c->d_objs, c->sc[0].work, c->sc[0].rects, Z, c->d_stats, 0, c->raster_blocks * 4u, s);
    }
    else
#endif
    {
        if (occl) { Z.pxrect = c->d_spxrect; Z.zmin = c->d_szmin; Z.vis_prev = c->d_sflag; Z.vis_stamp = 1u; Z.phase = 1u; }      // (the pass's own flags are 0 / 1)
        // a rebuilt work list starts from length 0 - zeroed HERE, in stream order behind the previous frame's shadow pipeline (k_cull_instances
        // grows it, every later kernel of the pipeline reads it)
        if (c->list_rebuild_mask & 1u) zr_launch_fill32(&c->d_sstats->n_vis_work[0], 0u, 1, s);
        zr_launch_cull_box(P, c->d_objs, c->sc[0].work, c->sc[0].rects, Z, c->d_sstats, 0, s, nullptr, nullptr, c->list_reuse[0]);
        if (c->list_rebuild_mask & 1u) c->list_valid[0] = true;
    }
    bin_and_raster(c, P, Z, 0, c->sn_tiles, s);
    if (ev) HIPCHK(c, hipEventRecord(ev[1], s));
    raster(c, P, Z, 0, s, occl ? 1 : 0);
    if (occl) {
        zr_launch_shadow_occlusion(P, c->d_objs, c->sc[0].work, c->sc[0].rects, c->d_spxrect, c->d_szmin, c->d_sflag, (const uint32_t*)shadow_buf(c),
                                   c->sc[0].bins, c->d_sstats, c->shadow_blocks * 8u, c->sflag_history ? (uint32_t)(c->frame_no & 3u) : 4u, s);
        c->sflag_history = true;
        raster(c, P, Z, 0, s, 2);
    }
    if (ev) HIPCHK(c, hipEventRecord(ev[2], s));
    HIPCHK(c, hipGetLastError());
    return ZR_OK;
}

// deferred-scene pass (ZE:3417-3480): cull + bin + raster + resolve of the owned tiles, on stream s
static int gbuffer_pass(zr_ctx* c, hipStream_t s)
{
    hipEvent_t* ev = c->timing_now ? c->evr[c->sample_no % zr_ctx::EV_RING] : nullptr;
    if (ev) HIPCHK(c, hipEventRecord(ev[9], s));
    ZrPass P = c->pass[1];             // (built by frame_begin; the overlay fields are set below)
    c->last_work[1] = P.n_work;
    // Two-pass occlusion culling: round 1 draws the meshlet-instances that owned a pixel last frame, a Hi-Z pyramid of the
    // result rejects what it hides, round 2 draws the rest.  The depth test decides every pixel either way, so the frame does
    // not depend on the history; without one (first frame of a scene) or with ZR_FLAG_NO_HIZ everything is drawn at once.
    const bool hiz_on = !(c->cfg.flags & ZR_FLAG_NO_HIZ) && P.n_work != 0;
    ZrHiz Z = c->hiz;
    Z.tiles_x = c->tiles_x; Z.tile_rank = c->cfg.tile_rank; Z.tile_world = c->cfg.tile_world;
    Z.pxrect = hiz_on ? c->d_pxrect : nullptr; Z.zmin = hiz_on ? c->d_zmin : nullptr;
    Z.vis_prev = c->d_visflag[c->vis_cur ^ 1]; Z.vis_now = hiz_on ? c->d_visflag[c->vis_cur] : nullptr;
    // visibility marks are frame stamps (1 .. 255): the resolve writes this frame's, the culls compare with last frame's - nothing is cleared
    const uint32_t vis_mark = 1u + (uint32_t)(c->frame_no % 255u);
    Z.vis_stamp = c->vis_mark_prev;
    Z.phase = 0;
#ifdef ZR_DIAG
    const bool tri_bins = !(c->cfg.flags & ZR_FLAG_MESHLET_BINS) && ZR_TILE == 32;      // A/B: the meshlet-binned rasteriser for the camera pass too
#else
    constexpr bool tri_bins = true;
    static_assert(ZR_TILE == 32, "the triangle-binned camera pass is written for 32 x 32 tiles");
#endif
    c->last_two_round = hiz_on && c->vis_history;
    // (triangle-binned pass: the cull kernel also compacts round 1's list - the survivors that owned a pixel last frame, or all of them)
#ifdef ZR_DIAG
    if (!tri_bins) zr_launch_cull(P, c->d_objs, c->sc[1].work, c->sc[1].rects, Z, c->d_stats, 1, c->raster_blocks * 4u, s);
    else
#endif
    {
        zr_launch_cull_box(P, c->d_objs, c->sc[1].work, c->sc[1].rects, Z, c->d_stats, 1, s, c->tb.sel, c->last_two_round ? Z.vis_prev : nullptr, c->list_reuse[1]);
        if (c->list_rebuild_mask & 2u) c->list_valid[1] = true;
    }
    const bool two = c->last_two_round;
    auto bin = [&](int slot) { if (!tri_bins) bin_and_raster(c, P, Z, slot, c->n_tiles, s); else if (slot == 2) tri_select(c, P, Z, slot, s); };
    // (the record buckets are planned from the previous frame: see tri_raster)
    const bool count_first = !c->plan_valid || (!two && c->plan_two_round);
    auto rast = [&](int slot) { if (tri_bins) tri_raster(c, P, Z, slot, s, slot == 2 || !two, slot == 1 && count_first); else raster(c, P, Z, slot, s); };
    if (c->last_two_round) {
        Z.phase = 1;
        bin(1);
        if (ev) HIPCHK(c, hipEventRecord(ev[3], s));
        rast(1);
        if (ev) HIPCHK(c, hipEventRecord(ev[4], s));
        zr_launch_hiz_build(c->d_vis, c->W, c->H, Z, c->d_hiz_regions, c->n_hiz_regions, s);
        Z.phase = 2;
        bin(2);
        if (ev) HIPCHK(c, hipEventRecord(ev[5], s));
        rast(2);
    } else {
        bin(1);
        if (ev) HIPCHK(c, hipEventRecord(ev[3], s));
        rast(1);
        if (ev) { HIPCHK(c, hipEventRecord(ev[4], s)); HIPCHK(c, hipEventRecord(ev[5], s)); }
    }
    {   // the overlay plane (skydome pixels) is written only when a skydome is drawn, or once more to wipe one that was
        const int par = (int)(c->frame_no & 1u);
        const bool sky = c->sky_set && c->sky_enabled;
        P.write_overlay = (sky || c->overlay_dirty[par]) ? 1u : 0u;
        c->overlay_dirty[par] = sky;
        P.sky_keys = nullptr; P.sky_object = c->sky_object;
        if (sky && c->d_sky_keys) { zr_launch_sky_tiles(P, c->d_objs, c->d_owned, c->n_owned, c->d_sky_keys, s); P.sky_keys = c->d_sky_keys; }
    }
    if (ev) HIPCHK(c, hipEventRecord(ev[6], s));
    zr_launch_resolve_gbuffer(P, c->d_objs, c->d_owned, c->n_owned, c->d_vis, c->G, c->d_lut, c->d_unorm_lut, Z.vis_now, c->d_stats, s, vis_mark);
    c->vis_mark_prev = vis_mark;
    if (ev) HIPCHK(c, hipEventRecord(ev[7], s));
    if (tri_bins && P.n_work != 0) {     // the next frame's buckets, from this frame's counts: nothing on this lane waits for it
        zr_launch_plan(c->tb, c->d_owned, c->n_owned, c->d_stats, false, c->bucket_pct, s);
        c->plan_valid = true; c->plan_two_round = two;
    }
    if (hiz_on) { c->vis_history = true; c->vis_cur ^= 1; } else c->vis_history = false;
    HIPCHK(c, hipGetLastError());
    return ZR_OK;
}

static int zr_render_shadow_impl(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    c->camera_on_lane = false;
    int rc = frame_begin(c, c->stream);
    if (rc == ZR_OK) rc = shadow_pass(c, c->stream);
    if (rc == ZR_OK) HIPCHK(c, hipEventRecord(c->ev_join, c->stream));
    if (rc == ZR_OK) c->stage = 1;
    return rc;
}
extern "C" int zr_render_shadow(zr_ctx* c)
{
    return zr_guard(c, [&]() { return zr_render_shadow_impl(c); });
}

extern "C" int zr_render_gbuffer(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    if (c->stage != 1) return zr_fail(c, ZR_ERR_STATE, "zr_render_gbuffer out of order");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = gbuffer_pass(c, c->stream);
    if (rc == ZR_OK) c->stage = 2;
    return rc;
}

// Both geometry passes of a frame.  Two lanes (unless ZR_FLAG_SERIAL_PASSES): the camera pipeline on cam_s; the shadow pipeline on
// the host's stream, where the lighting pass will follow.  The next frame's camera pipeline starts as soon as this one's is
// through, next to this frame's lighting; its shadow pipeline follows the lighting.  Never more than two kernels side by side:
// a third only takes occupancy from the other two (measured).
static int geometry_passes(zr_ctx* c)
{
    const bool lanes = !(c->cfg.flags & ZR_FLAG_SERIAL_PASSES) && c->cam_s != nullptr && !c->env_serial;
    int rc;
    c->camera_on_lane = false;
    if (lanes) {
        // Every event record / wait is a barrier packet, worth 5-10 us of bubble on the stream it sits on, and the host's stream
        // (lighting -> shadow pipeline -> lighting ...) is the lane the frame rate hangs on: it waits for the camera lane once per frame
        // (before the lighting pass) and for nothing else.  The shadow pipeline needs nothing of frame_begin's - its matrices are kernel
        // arguments, its statistics a block of its own that it resets itself, work-list length included.
        rc = frame_begin(c, c->cam_s);
        if (rc != ZR_OK) return rc;
        rc = shadow_pass(c, c->stream);
        if (rc == ZR_OK && !c->in_render) HIPCHK(c, hipEventRecord(c->ev_join, c->stream));      // (zr_stream_wait_shadow: a host that puts a collective behind the shadow pass)
        if (rc == ZR_OK) rc = gbuffer_pass(c, c->cam_s);
        if (rc == ZR_OK) { HIPCHK(c, hipEventRecord(c->ev_cam, c->cam_s)); c->camera_on_lane = true; }
    } else {
        rc = frame_begin(c, c->stream);
        if (rc != ZR_OK) return rc;
        rc = shadow_pass(c, c->stream);
        if (rc == ZR_OK) HIPCHK(c, hipEventRecord(c->ev_join, c->stream));
        if (rc == ZR_OK) rc = gbuffer_pass(c, c->stream);
    }
    if (rc == ZR_OK) c->stage = 2;
    return rc;
}

static int zr_render_geometry_impl(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    return geometry_passes(c);
}
extern "C" int zr_render_geometry(zr_ctx* c)
{
    return zr_guard(c, [&]() { return zr_render_geometry_impl(c); });
}

extern "C" int zr_stream_wait_shadow(zr_ctx* c, void* hip_stream)
{
    if (!c) return ZR_ERR_ARG;
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_join, 0));
    return ZR_OK;
}

static void light_params(const zr_ctx* c, ZrLightParams* Lp)
{
    ZrLightParams& L = *Lp; memset(&L, 0, sizeof L);
    static const float Bias[16] = { 0.5f, 0, 0, 0, 0, 0.5f, 0, 0, 0, 0, 1, 0, 0.5f, 0.5f, 0, 1 };
    zr_mat4_mul(Bias, c->view.ShadowmapSpace, L.SB);
    L.W = c->W; L.H = c->H; L.SD = c->SD; L.tiles_x = c->tiles_x; L.debug_view = c->debug_view;
    L.cube_dim = c->cube_dim; L.cube_levels = c->cube_levels; L.tile_world = c->cfg.tile_world;
    L.packed_out = (c->cfg.tile_world > 1 || (c->cfg.flags & ZR_FLAG_PACKED_TILES)) ? 1u : 0u;
    L.debug_skip = c->env_skip_light;
    L.bg_enabled = (c->bg_set && c->bg_enabled) ? 1u : 0u;
    L.has_overlay = c->overlay_dirty[c->frame_no & 1u] ? 1u : 0u;      // set by this frame's gbuffer pass
    { const int32_t np = c->view.LightsCount[1]; L.light_list = (np >= c->env_light_list_min && np <= XK_MAX_POINT_LIGHTS_NUM) ? 1u : 0u; }
    L.bg.data = c->d_bg; L.bg.w = c->bg_w; L.bg.h = c->bg_h; L.bg.levels = c->bg_levels; L.bg._pad = 0;
}

// The lighting shader's colour for a pixel that still holds every target's clear value: one launch of the lighting kernel over a
// one-pixel GBuffer.  It needs the finished shadow map (PCF at world position 0) and the frame's uniforms, nothing else.  View 6
// (the quad's interpolated vertex colour) depends on the pixel position, so it goes without.
static int empty_pixel_pass(zr_ctx* c, hipStream_t s)
{
    c->empty_ready = false;
    if (c->debug_view == 6u || c->env_no_empty_px || c->shading == ZR_SHADING_FORWARD) return ZR_OK;      // (forward: an empty pixel is the clear colour)
    ZrLightParams L; light_params(c, &L);
    L.W = 1; L.H = 1; L.tiles_x = 1; L.packed_out = 0; L.bg_enabled = 0;
    zr_launch_lighting(L, c->d_view, c->d_sowned, 1, c->Gclear, shadow_buf(c), c->cube, c->d_lut, c->d_unorm_lut, c->d_empty_rgba, s);
    HIPCHK(c, hipGetLastError());
    c->empty_ready = true;
    return ZR_OK;
}

static int lighting_pass(zr_ctx* c, hipStream_t s)
{
    hipEvent_t* ev = c->timing_now ? c->evr[c->sample_no % zr_ctx::EV_RING] : nullptr;
    ZrLightParams L; light_params(c, &L);
    L.empty_rgba = c->empty_ready ? c->d_empty_rgba : nullptr;
    // The next frame's shadow pass follows on this stream and rasterises into the OTHER copy of the map, which nothing reads or
    // writes while this pass runs: clear it here.
    const int npar = (int)((c->frame_no + 1u) & 1u);
    if (c->n_owned && s == c->stream) { L.clear_next = (uint32_t*)c->d_shadow_b[npar]; L.clear_n = c->SD * c->SD; c->shadow_cleared[npar] = true; }
    uint32_t* const frame_out = L.packed_out ? (c->d_tiles_ext ? c->d_tiles_ext : c->d_tiles) : c->d_color;
    if (c->shading == ZR_SHADING_FORWARD) {
        // Base.frag over the winners the resolve recorded, with this frame's camera block (frame_begin built it; the overlay fields play no part)
        if (L.clear_next) { zr_launch_fill32(L.clear_next, 0x3F800000u, L.clear_n, s); L.clear_next = nullptr; }
        zr_launch_forward(c->pass[1], L, c->d_view, c->d_objs, c->d_owned, c->n_owned, c->G, shadow_buf(c), c->cube, c->d_lut, c->d_unorm_lut, frame_out, s);
    } else {
        zr_launch_lighting(L, c->d_view, c->d_owned, c->n_owned, c->G, shadow_buf(c), c->cube, c->d_lut, c->d_unorm_lut, frame_out, s);
        if (c->debug_view == 9u)        // GBufferVis mosaic over the lit frame (needs the whole GBuffer: single-rank contexts only)
            zr_launch_gbuffer_vis(L, c->d_view, c->G, shadow_buf(c), c->cube, c->d_lut, c->d_color, s);
    }
    if (ev) HIPCHK(c, hipEventRecord(ev[8], s));
    HIPCHK(c, hipEventRecord(c->ev_end[c->frame_no % zr_ctx::END_RING], s));      // this frame's GBuffer / shadow map / uniforms copies are free again
    HIPCHK(c, hipGetLastError());
    if (c->timing_now) c->sample_no++;
    c->rendered = true; c->frame_no++; c->stage = 0;
    return ZR_OK;
}

extern "C" int zr_render_lighting(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    if (c->stage != 2) return zr_fail(c, ZR_ERR_STATE, "zr_render_lighting out of order");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t ls = c->stream;
    // The one wait of the host's stream per frame: the camera lane's GBuffer - and, ahead of it on that lane, this frame's k_frame_begin,
    // whose uniforms the empty-pixel pass below reads (the shadow pipeline before it needed nothing of them and did not wait).
    if (c->camera_on_lane) HIPCHK(c, hipStreamWaitEvent(ls, c->ev_cam, 0));
    int rc = empty_pixel_pass(c, ls);              // the shadow map (possibly reduced over ranks by the host) is final only now
    if (rc == ZR_OK) rc = lighting_pass(c, ls);
    return rc;
}

// RecordCommandBuffer (ZE:3160-3744) + vkQueueSubmit (ZE:2014): shadow -> deferred scene -> deferred lighting, with two
// frames in flight as in the reference (MAX_FRAMES_IN_FLIGHT, ZE:77).
// The shadow pass and the deferred-scene pass do not depend on each other, and the next frame's geometry does not depend on this
// frame's lighting.  zr_render therefore runs two lanes: the camera pipeline on the library's high-priority stream cam_s, and
// shadow pipeline -> lighting on the host's stream.  Whatever the host enqueues on its stream after zr_render is ordered after
// the finished frame, as before.  ZR_FLAG_SERIAL_PASSES keeps everything on the one stream, as the staged entry points do.
static int zr_render_impl(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    c->in_render = true;
    int rc = geometry_passes(c);
    if (rc == ZR_OK) rc = zr_render_lighting(c);
    c->in_render = false;
    if (rc != ZR_OK) c->stage = 0;
    return rc;
}
extern "C" int zr_render(zr_ctx* c)
{
    return zr_guard(c, [&]() { return zr_render_impl(c); });
}

// Multi-GPU shadow pass: this context draws instances i with i % world == rank (non-instanced draws count as instance 0).
// The per-rank shadow maps must be min-reduced before zr_render_lighting.  rank 0 / world 1 = the whole scene (default).
extern "C" int zr_set_shadow_partition(zr_ctx* c, uint32_t rank, uint32_t world)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, world >= 1 && rank < world);
    if (world > 1 && c->stile_world > 1) return zr_fail(c, ZR_ERR_STATE, "zr_set_shadow_partition: the map is already owned by tiles (zr_set_shadow_tiles)");
    c->shadow_rank = rank; c->shadow_world = world;
    return ZR_OK;
}

// Multi-GPU shadow pass, second form: the MAP is owned by light-space super-tiles exactly as the frame is owned by screen super-tiles
// (zr_tile_owner on the map's 32 x 32-texel tiles).  This context then draws only the casters whose texel box can reach a tile it owns
// (rank-local work list, instance- and meshlet-level rejects before any vertex work) - drawn whole, so its owned tiles are bit for
// bit the single-GPU map's - and the ranks exchange their tiles with ONE all-gather: zr_shadow_pack -> all-gather -> zr_shadow_unpack.
// No reduction: every texel has one owner.  rank 0 / world 1 = the whole map (default).
static int zr_set_shadow_tiles_impl(zr_ctx* c, uint32_t rank, uint32_t world)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, world >= 1 && rank < world);
    if (c->stage != 0) return zr_fail(c, ZR_ERR_STATE, "zr_set_shadow_tiles between the stages of a frame");
    if (world > 1 && c->shadow_world > 1) return zr_fail(c, ZR_ERR_STATE, "zr_set_shadow_tiles: the casters are already split by instance (zr_set_shadow_partition)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, zr_sync_all(c));
    dev_free(c->d_sowned_rank); dev_free(c->d_stile_map);
    c->stile_rank = 0; c->stile_world = 1; c->s_slots_per_rank = c->sn_tiles; c->n_sowned_rank = 0;
    c->list_valid[0] = false;
    if (world == 1) return ZR_OK;
    std::vector<uint32_t> owned, map(c->sn_tiles), counts(world, 0u);
    for (uint32_t t = 0; t < c->sn_tiles; ++t) {
        const uint32_t o = zr_tile_owner(t % c->stiles_x, t / c->stiles_x, world);
        map[t] = counts[o]++;
        if (o == rank) owned.push_back(t);
    }
    uint32_t spr = 0;
    for (uint32_t n : counts) spr = std::max(spr, n);
    for (uint32_t t = 0; t < c->sn_tiles; ++t) map[t] += zr_tile_owner(t % c->stiles_x, t / c->stiles_x, world) * spr;
    HIPCHK(c, upload(&c->d_sowned_rank, owned)); HIPCHK(c, upload(&c->d_stile_map, map));
    c->stile_rank = rank; c->stile_world = world; c->s_slots_per_rank = spr; c->n_sowned_rank = (uint32_t)owned.size();
    return ZR_OK;
}
extern "C" int zr_set_shadow_tiles(zr_ctx* c, uint32_t rank, uint32_t world)
{
    return zr_guard(c, [&]() { return zr_set_shadow_tiles_impl(c, rank, world); });
}
// bytes of one rank's packed share (slots_per_rank tiles of 32 x 32 floats; the all-gathered buffer holds world times that)
extern "C" int zr_shadow_tiles_bytes(zr_ctx* c, size_t* bytes_per_rank)
{
    if (!c || !bytes_per_rank) return ZR_ERR_ARG;
    *bytes_per_rank = (size_t)(c->stile_world > 1 ? c->s_slots_per_rank : c->sn_tiles) * ZR_TILE * ZR_TILE * 4;
    return ZR_OK;
}
// The owned tiles of the map just rasterised -> packed_dev (slot k = the k-th owned tile, unused slots keep depth 1.0), on `hip_stream`
// (NULL = the render stream, behind the shadow pass).
extern "C" int zr_shadow_pack(zr_ctx* c, void* packed_dev, void* hip_stream)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, packed_dev != nullptr);
    if (c->stile_world <= 1) return zr_fail(c, ZR_ERR_STATE, "zr_shadow_pack: the shadow map is not owned by tiles (zr_set_shadow_tiles)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    zr_launch_pack_tiles((const uint32_t*)shadow_buf(c), c->d_sowned_rank, c->n_sowned_rank, (uint32_t*)packed_dev, c->SD, c->SD, c->stiles_x, 0x3F800000u, s);
    HIPCHK(c, hipGetLastError());
    return ZR_OK;
}
// The all-gathered buffer (world x bytes_per_rank, rank-major) -> this frame's shadow map, every tile from its owner, on `hip_stream`
// (NULL = the render stream: call it before zr_render_lighting).
extern "C" int zr_shadow_unpack(zr_ctx* c, const void* gathered_dev, void* hip_stream)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, gathered_dev != nullptr);
    if (c->stile_world <= 1) return zr_fail(c, ZR_ERR_STATE, "zr_shadow_unpack: the shadow map is not owned by tiles (zr_set_shadow_tiles)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    zr_launch_untile((const uint32_t*)gathered_dev, c->d_stile_map, (uint32_t*)shadow_buf(c), c->SD, c->SD, c->stiles_x, c->sn_tiles, s);
    HIPCHK(c, hipGetLastError());
    return ZR_OK;
}

// Caller-owned shadow map (float[shadow_dim^2], e.g. a torch tensor RCCL reduces in place); NULL = the internal one.
extern "C" int zr_set_shadow_buffer(zr_ctx* c, void* ptr)
{
    if (!c) return ZR_ERR_ARG;
    c->d_shadow_ext = (float*)ptr;
    return ZR_OK;
}

extern "C" int zr_finish(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, zr_sync_all(c));
    if (c->rendered) {
        HIPCHK(c, hipMemcpy(&c->h_stats, c->d_stats, sizeof(ZrDevStats), hipMemcpyDeviceToHost));
        {   // the shadow pipeline's block: its slot-0 counters and its overflow latch belong to the frame's statistics
            ZrDevStats sh;
            HIPCHK(c, hipMemcpy(&sh, c->d_sstats, sizeof(ZrDevStats), hipMemcpyDeviceToHost));
            c->h_stats.survivors[0] = sh.survivors[0]; c->h_stats.bin_entries[0] = sh.bin_entries[0]; c->h_stats.n_chunks[0] = sh.n_chunks[0];
            c->h_stats.n_slow[0] = sh.n_slow[0]; c->h_stats.n_vis_work[0] = sh.n_vis_work[0];
            // (k_shadow_occlusion tallies in 32 partial sums; survivors of the cull = drawn by the first launch + left out + drawn late)
            c->h_stats.shadow_occluded = 0; for (uint32_t v : sh.covered_part) c->h_stats.shadow_occluded += v;
            c->h_stats.shadow_late = sh.shadow_late;
            c->h_stats.survivors[0] += c->h_stats.shadow_occluded + c->h_stats.shadow_late;
            c->h_stats.overflow |= sh.overflow;
            if (!c->h_stats.overflow_sticky) c->h_stats.overflow_sticky = sh.overflow_sticky;      // (a ZR_OVF_* code: the camera lane's, else the pipeline's)
        }
        c->h_stats.covered_shadow = 0;
        if (c->h_stats.overflow_sticky) {     // latched by ANY frame since the last zr_finish, not only the newest one
            HIPCHK(c, hipMemset(&c->d_stats->overflow_sticky, 0, sizeof(uint32_t)));
            HIPCHK(c, hipMemset(&c->d_sstats->overflow_sticky, 0, sizeof(uint32_t)));
            HIPCHK(c, hipDeviceSynchronize());      // (null-stream fills: through before the next frame is enqueued on the library's streams)
            c->h_stats.overflow = 1u;
            static const char* const what[] = { "?", "shadow bin entries", "slow-triangle list (zr_set_limits)", "camera work-unit table", "triangle-record arrays (zr_set_limits)",
                                                "late shadow bin entries" };
            const uint32_t code = c->h_stats.overflow_sticky < 6u ? c->h_stats.overflow_sticky : 0u;
            char msg[160];
            snprintf(msg, sizeof msg, "tile bin list overflow (%s): a frame since the last zr_finish is incomplete", what[code]);
            return zr_fail(c, ZR_ERR_OVERFLOW, msg);
        }
    }
    return ZR_OK;
}

// Mean per-pass GPU time over the last `last_n` frames (<= EV_RING), from hipEvents recorded on the render stream.
extern "C" int zr_get_pass_times_avg(zr_ctx* c, uint32_t last_n, float ms[ZR_PASS_COUNT])
{
    if (!c || !ms) return ZR_ERR_ARG;
    if (!c->rendered) return zr_fail(c, ZR_ERR_STATE, "nothing rendered yet");
    int rc = zr_finish(c);
    if (rc && rc != ZR_ERR_OVERFLOW) return rc;
    if (last_n == 0) last_n = 1;
    if (last_n > (uint32_t)zr_ctx::EV_RING) last_n = zr_ctx::EV_RING;
    if (c->sample_no == 0) return zr_fail(c, ZR_ERR_STATE, "no timed frame yet (zr_set_timing_interval)");
    if ((uint64_t)last_n > c->sample_no) last_n = (uint32_t)c->sample_no;
    double acc[ZR_PASS_COUNT] = { 0 };
    for (uint32_t k = 0; k < last_n; ++k) {
        hipEvent_t* ev = c->evr[(c->sample_no - 1 - k) % zr_ctx::EV_RING];
        float t[ZR_PASS_COUNT] = { 0 };
        (void)hipEventElapsedTime(&t[ZR_PASS_CULL_SHADOW], ev[0], ev[1]);
        (void)hipEventElapsedTime(&t[ZR_PASS_SHADOW], ev[1], ev[2]);
        (void)hipEventElapsedTime(&t[ZR_PASS_CULL_CAMERA], ev[9], ev[3]);
        (void)hipEventElapsedTime(&t[ZR_PASS_GBUFFER], ev[3], ev[4]);
        (void)hipEventElapsedTime(&t[ZR_PASS_HIZ], ev[4], ev[5]);
        (void)hipEventElapsedTime(&t[ZR_PASS_GBUFFER2], ev[5], ev[6]);
        (void)hipEventElapsedTime(&t[ZR_PASS_RESOLVE], ev[6], ev[7]);
        (void)hipEventElapsedTime(&t[ZR_PASS_LIGHTING], ev[7], ev[8]);
        (void)hipEventElapsedTime(&t[ZR_PASS_TOTAL], ev[0], ev[8]);
        for (int i = 0; i < ZR_PASS_COUNT; ++i) acc[i] += t[i];
    }
    for (int i = 0; i < ZR_PASS_COUNT; ++i) ms[i] = (float)(acc[i] / last_n);
    return ZR_OK;
}
extern "C" int zr_get_pass_times(zr_ctx* c, float ms[ZR_PASS_COUNT]) { return zr_get_pass_times_avg(c, 1, ms); }

// Begin-to-end GPU time (first kernel of the camera lane to the end of the lighting pass) of each of the last `n` timed frames,
// newest first; returns how many were written.  With two frames in flight this latency is longer than the frame period.
extern "C" int zr_get_frame_latencies(zr_ctx* c, uint32_t n, float* ms)
{
    if (!c || !ms) return ZR_ERR_ARG;
    if (!c->rendered) return zr_fail(c, ZR_ERR_STATE, "nothing rendered yet");
    int rc = zr_finish(c);
    if (rc && rc != ZR_ERR_OVERFLOW) return rc;
    if (n > (uint32_t)zr_ctx::EV_RING) n = zr_ctx::EV_RING;
    if ((uint64_t)n > c->sample_no) n = (uint32_t)c->sample_no;
    for (uint32_t k = 0; k < n; ++k) {
        hipEvent_t* ev = c->evr[(c->sample_no - 1 - k) % zr_ctx::EV_RING];
        ms[k] = 0.0f;
        (void)hipEventElapsedTime(&ms[k], ev[0], ev[8]);
    }
    return (int)n;
}

// GPU time between the ends of consecutive frames (the frame period the GPU sustained) for the last `n` frames, newest first;
// returns how many were written (<= END_RING - 1).  Costs nothing extra: the end-of-frame event exists for the double buffering.
extern "C" int zr_get_frame_periods(zr_ctx* c, uint32_t n, float* ms)
{
    if (!c || !ms) return ZR_ERR_ARG;
    if (!c->rendered) return zr_fail(c, ZR_ERR_STATE, "nothing rendered yet");
    int rc = zr_finish(c);
    if (rc && rc != ZR_ERR_OVERFLOW) return rc;
    const uint64_t have = c->frame_no > 0 ? c->frame_no - 1 : 0;
    if (n > (uint32_t)zr_ctx::END_RING - 1u) n = zr_ctx::END_RING - 1;
    if ((uint64_t)n > have) n = (uint32_t)have;
    for (uint32_t k = 0; k < n; ++k) {
        const uint64_t f = c->frame_no - 1 - k;
        ms[k] = 0.0f;
        (void)hipEventElapsedTime(&ms[k], c->ev_end[(f - 1) % zr_ctx::END_RING], c->ev_end[f % zr_ctx::END_RING]);
    }
    return (int)n;
}

// Per-pass hipEvents are recorded on every interval-th frame (default 1 = every frame, 0 = never).  Each record is a small
// bubble on the render stream (~6 us on MI355X, six per frame), so a host that only wants throughput samples sparsely.
extern "C" int zr_set_timing_interval(zr_ctx* c, uint32_t interval)
{
    if (!c) return ZR_ERR_ARG;
    c->timing_interval = interval;
    return ZR_OK;
}

extern "C" uint32_t zr_abi_version(void) { return ZR_ABI_VERSION; }

// `bytes` = sizeof(zr_stats) as the CALLER was compiled: the struct only ever grows at its end, so a host built against an older header
// gets the fields it knows and is never written past.
extern "C" int zr_get_stats(zr_ctx* c, zr_stats* out_user, size_t bytes)
{
    if (!c || !out_user) return ZR_ERR_ARG;
    ARGCHK(c, bytes >= offsetof(zr_stats, round1_survivors) && bytes % 4 == 0);      // (the first release's struct ended there)
    zr_stats out_full; zr_stats* out = &out_full;
    int rc = zr_finish(c);
    if (c->rendered && (rc == ZR_OK || rc == ZR_ERR_OVERFLOW)) {      // shadow coverage is a statistic, counted on demand
        ZrDevStats z; (void)hipMemcpy(&z, c->d_stats, sizeof z, hipMemcpyDeviceToHost);
        uint32_t zero = 0;
        (void)hipMemcpy(&c->d_stats->covered_shadow, &zero, 4, hipMemcpyHostToDevice);
        zr_launch_count_shadow((const uint32_t*)(c->d_shadow_ext ? c->d_shadow_ext : c->d_shadow), (size_t)c->SD * c->SD, c->d_stats, c->stream);
        (void)hipStreamSynchronize(c->stream);
        const ZrDevStats keep = c->h_stats;      // (zr_finish merged the shadow pipeline's block into it)
        (void)hipMemcpy(&c->h_stats, c->d_stats, sizeof(ZrDevStats), hipMemcpyDeviceToHost);
        c->h_stats.survivors[0] = keep.survivors[0]; c->h_stats.bin_entries[0] = keep.bin_entries[0]; c->h_stats.n_chunks[0] = keep.n_chunks[0];
        c->h_stats.n_slow[0] = keep.n_slow[0]; c->h_stats.overflow |= keep.overflow;
        c->h_stats.shadow_occluded = keep.shadow_occluded; c->h_stats.shadow_late = keep.shadow_late;
    }
#ifdef ZR_DIAG
    if (getenv("ZR_DUMP_STATS")) {     // diagnostics: the raw device block
        const ZrDevStats& h = c->h_stats;
        fprintf(stderr, "zr stats: survivors %u %u %u  bin_entries %u %u %u  n_sel %u %u %u  n_slow %u %u %u  hiz_culled %u  n_chunks %u %u %u  overflow records %u %u\n",
                h.survivors[0], h.survivors[1], h.survivors[2], h.bin_entries[0], h.bin_entries[1], h.bin_entries[2], h.n_sel[0], h.n_sel[1], h.n_sel[2],
                h.n_slow[0], h.n_slow[1], h.n_slow[2], h.hiz_culled, h.n_chunks[0], h.n_chunks[1], h.n_chunks[2], h.pool_used[1], h.pool_used[2]);
    }
#endif
    memset(out, 0, sizeof *out);
    for (int i = 0; i < 2; ++i) {
        out->work_items[i] = c->last_work[i]; out->survivors[i] = c->h_stats.survivors[i]; out->bin_entries[i] = c->h_stats.bin_entries[i];
    }
    out->survivors[1] += c->h_stats.survivors[2]; out->bin_entries[1] += c->h_stats.bin_entries[2];    // both rounds of the camera pass
    out->hiz_culled = c->h_stats.hiz_culled; out->round1_survivors = c->last_two_round ? c->h_stats.survivors[1] : 0;
    out->covered_pixels = 0; for (uint32_t v : c->h_stats.covered_part) out->covered_pixels += v;
    out->covered_shadow_texels = c->h_stats.covered_shadow; out->overflow = c->h_stats.overflow;
    out->shadow_occluded = c->h_stats.shadow_occluded; out->shadow_late = c->h_stats.shadow_late;
    out->hiz_culled_geom = c->h_stats.hiz_culled_geom; out->struct_bytes = (uint32_t)sizeof(zr_stats);
    memcpy(out_user, out, std::min(bytes, sizeof(zr_stats)));
    return rc;
}

// ------------------------------------------------------------------------------------------------ read-back

extern "C" int zr_read_color(zr_ctx* c, uint8_t* dst, size_t bytes)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, dst && bytes == (size_t)c->W * c->H * 4);
    int rc = zr_finish(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(dst, c->d_color, bytes, hipMemcpyDeviceToHost));
    return ZR_OK;
}
extern "C" int zr_read_gbuffer(zr_ctx* c, int target, void* dst, size_t bytes)
{
    if (!c) return ZR_ERR_ARG;
    const void* src[6] = { c->G.depth, c->G.scene_color, c->G.gA, c->G.gB, c->G.gC, c->G.gD };
    ARGCHK(c, dst && target >= 0 && target < 6);
    ARGCHK(c, bytes == (size_t)c->W * c->H * (target == 5 ? 8 : 4));
    int rc = zr_finish(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(dst, src[target], bytes, hipMemcpyDeviceToHost));
    return ZR_OK;
}
extern "C" int zr_read_shadowmap(zr_ctx* c, float* dst, size_t bytes)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, dst && bytes == (size_t)c->SD * c->SD * 4);
    int rc = zr_finish(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(dst, c->d_shadow_ext ? c->d_shadow_ext : c->d_shadow, bytes, hipMemcpyDeviceToHost));
    return ZR_OK;
}

// The frame enqueued last, copied into caller-owned DEVICE buffers in stream order (no host synchronisation): what a host with two frames
// in flight uses instead of zr_read_color - the copies are ordered behind that frame's lighting pass and ahead of whatever the next
// zr_render enqueues on the render stream.  Either pointer may be NULL.  (The shadow map is double-buffered inside: the NEXT frame's
// shadow pipeline draws into the other copy, so the map copied here is this frame's whatever runs beside it.)
extern "C" int zr_copy_frame_async(zr_ctx* c, void* color_dev, void* shadow_dev)
{
    if (!c) return ZR_ERR_ARG;
    if (!c->rendered || c->stage != 0) return zr_fail(c, ZR_ERR_STATE, "zr_copy_frame_async: no finished frame enqueued");
    HIPCHK(c, hipSetDevice(c->device));
    if (color_dev) HIPCHK(c, hipMemcpyAsync(color_dev, c->d_color, (size_t)c->W * c->H * 4, hipMemcpyDeviceToDevice, c->stream));
    if (shadow_dev) HIPCHK(c, hipMemcpyAsync(shadow_dev, shadow_buf(c), (size_t)c->SD * c->SD * 4, hipMemcpyDeviceToDevice, c->stream));
    return ZR_OK;
}

// ------------------------------------------------------------------------------------------------ multi-GPU tiles

extern "C" int zr_tiles_device_buffer(zr_ctx* c, void** p, size_t* bytes)
{
    if (!c || !p || !bytes) return ZR_ERR_ARG;
    *p = c->d_tiles; *bytes = (size_t)c->slots_per_rank * ZR_TILE * ZR_TILE * 4;
    return ZR_OK;
}
// Lets the caller own the packed tile buffer (e.g. a torch tensor handed to RCCL; two of them alternate so that frame k's
// all-gather overlaps frame k+1's rendering).  ptr must hold zr_tiles_device_buffer's byte count; NULL = internal buffer.
extern "C" int zr_set_tiles_buffer(zr_ctx* c, void* ptr)
{
    if (!c) return ZR_ERR_ARG;
    c->d_tiles_ext = (uint32_t*)ptr;
    return ZR_OK;
}

extern "C" int zr_read_tiles(zr_ctx* c, uint8_t* dst, size_t bytes)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, dst && bytes == (size_t)c->slots_per_rank * ZR_TILE * ZR_TILE * 4);
    int rc = zr_finish(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(dst, c->d_tiles_ext ? c->d_tiles_ext : c->d_tiles, bytes, hipMemcpyDeviceToHost));
    return ZR_OK;
}
extern "C" int zr_composite(zr_ctx* c, const void* gathered)
{
    if (!c) return ZR_ERR_ARG;
    ARGCHK(c, gathered != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    zr_launch_untile((const uint32_t*)gathered, c->d_tile_map, c->d_color, c->W, c->H, c->tiles_x, c->n_tiles, c->stream);
    HIPCHK(c, hipGetLastError());
    return ZR_OK;
}
extern "C" int zr_color_device_ptr(zr_ctx* c, void** p) { if (!c || !p) return ZR_ERR_ARG; *p = c->d_color; return ZR_OK; }
