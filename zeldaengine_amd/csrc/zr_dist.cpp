// zr_dist.cpp — native multi-GPU host: one process per GPU, the frame's collectives issued by the library itself.
//
// The reference has no multi-GPU path (one VkDevice, one queue, ZE:2241); this is the MI355X-native addition of SURVEY 5 / 8(e): the
// frame is partitioned by screen super-tiles (zr_tile_owner), every rank holds the whole scene and renders its tiles into a packed,
// tile-major RGBA8 buffer, and ONE ncclAllGather over xGMI per frame (4 bytes per pixel of the frame in total) + an untile kernel
// gives every rank the composite.  Optionally the shadow pass shrinks with N too: ZR_DIST_SHADOW_TILES owns the MAP by light-space
// super-tiles (a rank draws the casters that reach its tiles; a second ncclAllGather of the packed tiles, 4 MiB in total, no reduction),
// or ZR_DIST_SPLIT_SHADOW splits the casters i % world and reduces the 1024^2 maps with ncclAllReduce(min) (the depth test LESS_OR_EQUAL
// is a min and the bias is per triangle); both are exact.
//
// Streams: the render stream + the camera lane produce frame k + 1 while the collective stream gathers and composites frame k; the
// packed / gathered buffers are double-buffered and ordered by events, so the xGMI latency hides behind rendering and the host only
// enqueues (no Python, no torch in the loop).  RCCL is loaded on first use with dlopen("librccl.so.1"): in a process that already holds
// PyTorch's copy that is the same library (same SONAME), otherwise ROCm's.
#include "zr_ctx.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <vector>

struct ZrDist {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t comm_s = nullptr;
    uint32_t rank = 0, world = 1; bool split_shadow = false, shadow_tiles = false;
    size_t stile_bytes = 0; uint32_t* spacked = nullptr; uint32_t* sgathered = nullptr; hipEvent_t shadow_packed = nullptr;
    size_t tile_bytes = 0;
    uint32_t* tiles[2] = { nullptr, nullptr }; uint32_t* gathered[2] = { nullptr, nullptr };
    hipEvent_t rendered[2] = { nullptr, nullptr }, consumed[2] = { nullptr, nullptr }, shadow_reduced = nullptr;
    float* shadow = nullptr;
    uint64_t k = 0;
};

static bool load_rccl(ZrDist* d, std::string* err)
{
    if (d->lib) return true;
    d->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!d->lib) d->lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!d->lib) { *err = std::string("cannot load librccl: ") + dlerror(); return false; }
    bool ok = true;
    auto sym = [&](const char* n) -> void* { void* p = dlsym(d->lib, n); if (!p) { ok = false; *err = std::string("librccl lacks ") + n; } return p; };
    d->GetUniqueId = (decltype(d->GetUniqueId))sym("ncclGetUniqueId");
    d->CommInitRank = (decltype(d->CommInitRank))sym("ncclCommInitRank");
    d->CommDestroy = (decltype(d->CommDestroy))sym("ncclCommDestroy");
    d->AllGather = (decltype(d->AllGather))sym("ncclAllGather");
    d->AllReduce = (decltype(d->AllReduce))sym("ncclAllReduce");
    d->GetErrorString = (decltype(d->GetErrorString))sym("ncclGetErrorString");
    return ok;
}

#define HIPCHK(c, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
    return zr_fail((c), ZR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
#define NCCLCHK(c, d, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) \
    return zr_fail((c), ZR_ERR_DEVICE, std::string(#expr) + ": " + (d)->GetErrorString(_r)); } while (0)

extern "C" int zr_dist_unique_id(void* id, size_t bytes)
{
    if (!id || bytes != sizeof(ncclUniqueId)) return ZR_ERR_ARG;
    ZrDist tmp; std::string err;
    if (!load_rccl(&tmp, &err)) { fprintf(stderr, "zr_dist_unique_id: %s\n", err.c_str()); return ZR_ERR_UNSUPPORTED; }
    ncclUniqueId u;
    if (tmp.GetUniqueId(&u) != ncclSuccess) return ZR_ERR_DEVICE;
    memcpy(id, &u, sizeof u);
    return ZR_OK;                     // (the library handle stays open: RCCL is not meant to be unloaded)
}

void zr_dist_destroy(zr_ctx* c)
{
    ZrDist* d = c->dist;
    if (!d) return;
    if (d->comm_s) (void)hipStreamSynchronize(d->comm_s);
    if (d->comm && d->CommDestroy) (void)d->CommDestroy(d->comm);
    for (int b = 0; b < 2; ++b) {
        if (d->tiles[b]) (void)hipFree(d->tiles[b]);
        if (d->gathered[b]) (void)hipFree(d->gathered[b]);
        if (d->rendered[b]) (void)hipEventDestroy(d->rendered[b]);
        if (d->consumed[b]) (void)hipEventDestroy(d->consumed[b]);
    }
    if (d->shadow) {
        if (c->d_shadow_ext == d->shadow) { c->d_shadow_ext = nullptr; c->shadow_rank = 0; c->shadow_world = 1; }      // back to the whole map
        (void)hipFree(d->shadow);
    }
    if (d->shadow_tiles && c->stile_world > 1) (void)zr_set_shadow_tiles(c, 0, 1);      // back to the whole map
    if (d->spacked) (void)hipFree(d->spacked);
    if (d->sgathered) (void)hipFree(d->sgathered);
    if (d->shadow_packed) (void)hipEventDestroy(d->shadow_packed);
    if (d->shadow_reduced) (void)hipEventDestroy(d->shadow_reduced);
    if (d->comm_s) (void)hipStreamDestroy(d->comm_s);
    if (c->d_tiles_ext == d->tiles[0] || c->d_tiles_ext == d->tiles[1]) c->d_tiles_ext = nullptr;
    delete d;
    c->dist = nullptr;
}

hipError_t zr_dist_sync(zr_ctx* c) { return (c->dist && c->dist->comm_s) ? hipStreamSynchronize(c->dist->comm_s) : hipSuccess; }

// Bring-up in two steps so that a host can AGREE between them: zr_dist_prepare is local (librccl, the collective stream, the packed /
// gathered buffers) and may fail on one rank alone; ncclCommInitRank inside zr_dist_connect is itself a collective - a rank that never
// reaches it leaves the others blocked inside it - so a host calls it only after every rank has reported a successful prepare.
extern "C" int zr_dist_prepare(zr_ctx* c, uint32_t rank, uint32_t world, uint32_t dist_flags)
{
    if (!c) return ZR_ERR_ARG;
    if (world == 0 || rank >= world) return zr_fail(c, ZR_ERR_ARG, "zr_dist_prepare: bad rank / world");
    if (c->dist) return zr_fail(c, ZR_ERR_STATE, "zr_dist_prepare: already initialised");
    if (rank != c->cfg.tile_rank || world != c->cfg.tile_world)
        return zr_fail(c, ZR_ERR_ARG, "zr_dist_prepare: rank / world differ from the context's tile_rank / tile_world");
    if (world == 1 && !(c->cfg.flags & ZR_FLAG_PACKED_TILES))
        return zr_fail(c, ZR_ERR_ARG, "zr_dist_prepare: a world of one needs ZR_FLAG_PACKED_TILES (the packed tile path)");
    HIPCHK(c, hipSetDevice(c->device));
    ZrDist* d = new (std::nothrow) ZrDist();
    if (!d) return zr_fail(c, ZR_ERR_OOM, "zr_dist_prepare: out of memory");
    c->dist = d;
    std::string err;
    if (!load_rccl(d, &err)) { zr_dist_destroy(c); return zr_fail(c, ZR_ERR_UNSUPPORTED, err); }
    if ((dist_flags & ZR_DIST_SPLIT_SHADOW) && (dist_flags & ZR_DIST_SHADOW_TILES)) { zr_dist_destroy(c); return zr_fail(c, ZR_ERR_ARG, "zr_dist_prepare: SPLIT_SHADOW and SHADOW_TILES exclude each other"); }
    d->rank = rank; d->world = world; d->split_shadow = (dist_flags & ZR_DIST_SPLIT_SHADOW) != 0 && world > 1;
    d->shadow_tiles = (dist_flags & ZR_DIST_SHADOW_TILES) != 0 && world > 1;
    d->tile_bytes = (size_t)c->slots_per_rank * ZR_TILE * ZR_TILE * 4;
    auto bail = [&](int code, const std::string& m) { zr_dist_destroy(c); return zr_fail(c, code, m); };
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&d->comm_s, hipStreamNonBlocking, least) != hipSuccess) return bail(ZR_ERR_DEVICE, "zr_dist_prepare: stream");
    for (int b = 0; b < 2; ++b) {
        if (hipMalloc((void**)&d->tiles[b], d->tile_bytes) != hipSuccess || hipMalloc((void**)&d->gathered[b], d->tile_bytes * world) != hipSuccess ||
            hipMemset(d->tiles[b], 0, d->tile_bytes) != hipSuccess ||
            hipEventCreateWithFlags(&d->rendered[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->consumed[b], hipEventDisableTiming) != hipSuccess) return bail(ZR_ERR_DEVICE, "zr_dist_prepare: buffers");
    }
    if (hipDeviceSynchronize() != hipSuccess) return bail(ZR_ERR_DEVICE, "zr_dist_prepare: fills");      // (null-stream fills; the streams are non-blocking ones)
    if (d->split_shadow) {
        if (hipMalloc((void**)&d->shadow, (size_t)c->SD * c->SD * 4) != hipSuccess ||
            hipEventCreateWithFlags(&d->shadow_reduced, hipEventDisableTiming) != hipSuccess) return bail(ZR_ERR_DEVICE, "zr_dist_prepare: shadow buffer");
    }
    if (d->shadow_tiles) {
        // the map's tile partition for `world` ranks: sizes only (the context keeps drawing the whole map until the communicator stands)
        uint32_t n_owned = 0, spr = 0;
        if (zr_tile_partition(c->SD, c->SD, world, rank, nullptr, &n_owned, &spr) != ZR_OK) return bail(ZR_ERR_ARG, "zr_dist_prepare: shadow tile partition");
        d->stile_bytes = (size_t)spr * ZR_TILE * ZR_TILE * 4;
        if (hipMalloc((void**)&d->spacked, d->stile_bytes) != hipSuccess || hipMalloc((void**)&d->sgathered, d->stile_bytes * world) != hipSuccess ||
            hipEventCreateWithFlags(&d->shadow_packed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&d->shadow_reduced, hipEventDisableTiming) != hipSuccess) return bail(ZR_ERR_DEVICE, "zr_dist_prepare: shadow tile buffers");
        {   // unused slots of a rank with fewer tiles than slots_per_rank are gathered too: depth 1.0, once
            std::vector<uint32_t> ones(d->stile_bytes / 4, 0x3F800000u);
            if (hipMemcpy(d->spacked, ones.data(), d->stile_bytes, hipMemcpyHostToDevice) != hipSuccess) return bail(ZR_ERR_DEVICE, "zr_dist_prepare: shadow tile buffers");
        }
    }
    return ZR_OK;
}

extern "C" int zr_dist_connect(zr_ctx* c, const void* id, size_t bytes)
{
    if (!c) return ZR_ERR_ARG;
    ZrDist* d = c->dist;
    if (!d) return zr_fail(c, ZR_ERR_STATE, "zr_dist_connect: zr_dist_prepare first");
    if (d->comm) return zr_fail(c, ZR_ERR_STATE, "zr_dist_connect: already connected");
    if (!id || bytes != sizeof(ncclUniqueId)) return zr_fail(c, ZR_ERR_ARG, "zr_dist_connect: bad id");
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId u; memcpy(&u, id, sizeof u);
    const ncclResult_t r = d->CommInitRank(&d->comm, (int)d->world, u, (int)d->rank);
    if (r != ncclSuccess) {
        const std::string m = std::string("ncclCommInitRank: ") + d->GetErrorString(r);
        d->comm = nullptr;
        zr_dist_destroy(c);
        return zr_fail(c, ZR_ERR_DEVICE, m);
    }
    // only a connected context draws a share of the shadow casters: a host that keeps using a context whose bring-up failed through
    // plain zr_render must get the whole map
    if (d->split_shadow) { c->d_shadow_ext = d->shadow; c->shadow_rank = d->rank; c->shadow_world = d->world; }
    if (d->shadow_tiles) { const int rc = zr_set_shadow_tiles(c, d->rank, d->world); if (rc != ZR_OK) return rc; }
    return ZR_OK;
}

extern "C" int zr_dist_init(zr_ctx* c, const void* id, size_t bytes, uint32_t rank, uint32_t world, uint32_t dist_flags)
{
    if (!c) return ZR_ERR_ARG;
    if (!id || bytes != sizeof(ncclUniqueId)) return zr_fail(c, ZR_ERR_ARG, "zr_dist_init: bad id");
    const int rc = zr_dist_prepare(c, rank, world, dist_flags);
    return rc != ZR_OK ? rc : zr_dist_connect(c, id, bytes);
}

// One frame of this rank: render -> (all-gather + untile on the collective stream, overlapped with the next frame's rendering).
extern "C" int zr_dist_frame(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    ZrDist* d = c->dist;
    if (!d || !d->comm) return zr_fail(c, ZR_ERR_STATE, "zr_dist_frame: zr_dist_init (or zr_dist_prepare + zr_dist_connect) first");
    HIPCHK(c, hipSetDevice(c->device));
    const int b = (int)(d->k & 1u);
    d->k++;
    // the render stream may overwrite packed buffer b only after its previous contents were gathered
    if (d->k > 2) HIPCHK(c, hipStreamWaitEvent(c->stream, d->consumed[b], 0));
    c->d_tiles_ext = d->tiles[b];
    int rc;
    if (d->split_shadow) {
        rc = zr_render_geometry(c);                       // this rank's share of the casters on the render stream, camera passes on the lane
        if (rc) return rc;
        rc = zr_stream_wait_shadow(c, d->comm_s);
        if (rc) return rc;
        NCCLCHK(c, d, d->AllReduce(d->shadow, d->shadow, (size_t)c->SD * c->SD, ncclFloat32, ncclMin, d->comm, d->comm_s));
        HIPCHK(c, hipEventRecord(d->shadow_reduced, d->comm_s));
        HIPCHK(c, hipStreamWaitEvent(c->stream, d->shadow_reduced, 0));
        rc = zr_render_lighting(c);
    } else if (d->shadow_tiles) {
        // this rank's share of the MAP on the render stream (camera passes on the lane); its owned tiles packed behind the pass, gathered on
        // the collective stream, scattered back into the map on the render stream ahead of the lighting pass.  One buffer pair does: the next
        // frame's pack follows this frame's unpack in render-stream order.
        rc = zr_render_geometry(c);
        if (rc) return rc;
        rc = zr_shadow_pack(c, d->spacked, nullptr);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(d->shadow_packed, c->stream));
        HIPCHK(c, hipStreamWaitEvent(d->comm_s, d->shadow_packed, 0));
        NCCLCHK(c, d, d->AllGather(d->spacked, d->sgathered, d->stile_bytes, ncclUint8, d->comm, d->comm_s));
        HIPCHK(c, hipEventRecord(d->shadow_reduced, d->comm_s));
        HIPCHK(c, hipStreamWaitEvent(c->stream, d->shadow_reduced, 0));
        rc = zr_shadow_unpack(c, d->sgathered, nullptr);
        if (rc) return rc;
        rc = zr_render_lighting(c);
    } else rc = zr_render(c);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(d->rendered[b], c->stream));
    HIPCHK(c, hipStreamWaitEvent(d->comm_s, d->rendered[b], 0));
    NCCLCHK(c, d, d->AllGather(d->tiles[b], d->gathered[b], d->tile_bytes, ncclUint8, d->comm, d->comm_s));
    HIPCHK(c, hipEventRecord(d->consumed[b], d->comm_s));
    zr_launch_untile(d->gathered[b], c->d_tile_map, c->d_color, c->W, c->H, c->tiles_x, c->n_tiles, d->comm_s);
    HIPCHK(c, hipGetLastError());
    return ZR_OK;
}

// The composite of the frame enqueued last, copied on the collective stream: in order behind that frame's untile, ahead of the next one's.
extern "C" int zr_dist_copy_frame_async(zr_ctx* c, void* color_dev)
{
    if (!c || !color_dev) return ZR_ERR_ARG;
    ZrDist* d = c->dist;
    if (!d || !d->comm || d->k == 0) return zr_fail(c, ZR_ERR_STATE, "zr_dist_copy_frame_async: no frame enqueued by zr_dist_frame");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(color_dev, c->d_color, (size_t)c->W * c->H * 4, hipMemcpyDeviceToDevice, d->comm_s));
    return ZR_OK;
}
