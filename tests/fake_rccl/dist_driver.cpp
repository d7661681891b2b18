// dist_driver — TEST-ONLY native host of the multi-GPU frame loop: the ranks of an N-rank job as processes (one rank per process) or as
// threads of one process, each with a context of its own through nothing but include/zelda_render.h:
//     zr_create(tile_rank, tile_world) -> zr_set_asset_root -> zr_world_load_file -> zr_dist_prepare -> [id file] -> zr_dist_connect
//     per frame: zr_world_update_uniforms -> zr_dist_frame -> zr_dist_copy_frame_async (EVERY frame's composite, in collective-stream order)
//     zr_finish -> the captured frames go to <out>_r<rank>.bin
// tests/test_gpu_dist_native.py runs it with tests/fake_rccl/_build (the stand-in librccl.so.1) on LD_LIBRARY_PATH, all ranks on device 0,
// and compares every frame of every rank with the frames `--plain` (one context, zr_render) produced.  On a real node the same program
// with --device-per-rank and ROCm's librccl is a native multi-GPU host.
//
//   dist_driver --root DIR --world-file F.json --size WxH --shadow N --frames K [--roll-stage-step F] [--roll-light-step F]
//               --out PREFIX  ( --plain | --world N --ranks A:B --id-file FILE [--mode replicated|tiles|split] [--device-per-rank] )
#include "../../include/zelda_render.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct Opts {
    std::string root = ".", world_file, out, id_file, mode = "replicated";
    uint32_t W = 640, H = 360, SD = 256, frames = 1, world = 1, r0 = 0, r1 = 1;
    float roll_stage = 0.0f, roll_light = 0.0f;
    bool plain = false, device_per_rank = false;
};

static int run_rank(const Opts& o, uint32_t rank)
{
    auto fail = [&](zr_ctx* c, const char* what, int rc) {
        fprintf(stderr, "dist_driver rank %u: %s failed (%d): %s\n", rank, what, rc, c ? zr_last_error(c) : "");
        if (c) zr_destroy(c);
        return 1;
    };
    zr_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.width = o.W; cfg.height = o.H; cfg.shadow_dim = o.SD; cfg.device = o.device_per_rank ? (int)rank : 0;
    cfg.tile_rank = o.plain ? 0 : rank; cfg.tile_world = o.plain ? 1 : o.world;
    zr_ctx* c = nullptr;
    int rc = zr_create(&cfg, &c);
    if (rc) return fail(nullptr, "zr_create", rc);
    if ((rc = zr_set_asset_root(c, o.root.c_str()))) return fail(c, "zr_set_asset_root", rc);
    if ((rc = zr_world_load_file(c, o.world_file.c_str()))) return fail(c, "zr_world_load_file", rc);
    if (!o.plain) {
        const uint32_t flags = o.mode == "tiles" ? ZR_DIST_SHADOW_TILES : o.mode == "split" ? ZR_DIST_SPLIT_SHADOW : 0u;
        if ((rc = zr_dist_prepare(c, rank, o.world, flags))) return fail(c, "zr_dist_prepare", rc);
        unsigned char id[128];
        if (rank == 0) {          // the rendezvous of a single node: rank 0's id through a file (write, then rename: readers never see half of it)
            if ((rc = zr_dist_unique_id(id, sizeof id))) return fail(c, "zr_dist_unique_id", rc);
            const std::string tmp = o.id_file + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return fail(c, "write id file", -1);
            fclose(f);
            if (rename(tmp.c_str(), o.id_file.c_str()) != 0) return fail(c, "rename id file", -1);
        } else {
            bool got = false;
            for (int i = 0; i < 6000 && !got; ++i) {
                FILE* f = fopen(o.id_file.c_str(), "rb");
                if (f) { got = fread(id, 1, sizeof id, f) == sizeof id; fclose(f); }
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
            if (!got) return fail(c, "read id file", -1);
        }
        if ((rc = zr_dist_connect(c, id, sizeof id))) return fail(c, "zr_dist_connect", rc);
    }
    const size_t frame_bytes = (size_t)o.W * o.H * 4;
    uint8_t* dev = nullptr;
    if (hipSetDevice(cfg.device) != hipSuccess || hipMalloc((void**)&dev, frame_bytes * o.frames) != hipSuccess) return fail(c, "hipMalloc", -1);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t f = 0; f < o.frames; ++f) {
        if ((rc = zr_world_update_uniforms(c, o.roll_stage * (float)f, o.roll_light * (float)f, 0.016f * (float)f))) return fail(c, "zr_world_update_uniforms", rc);
        if (o.plain) {
            if ((rc = zr_render(c))) return fail(c, "zr_render", rc);
            if ((rc = zr_copy_frame_async(c, dev + f * frame_bytes, nullptr))) return fail(c, "zr_copy_frame_async", rc);
        } else {
            if ((rc = zr_dist_frame(c))) return fail(c, "zr_dist_frame", rc);
            if ((rc = zr_dist_copy_frame_async(c, dev + f * frame_bytes))) return fail(c, "zr_dist_copy_frame_async", rc);
        }
    }
    const double enq_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if ((rc = zr_finish(c))) return fail(c, "zr_finish", rc);
    const double all_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<uint8_t> host(frame_bytes * o.frames);
    if (hipMemcpy(host.data(), dev, host.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail(c, "hipMemcpy", -1);
    (void)hipFree(dev);
    const std::string path = o.out + "_r" + std::to_string(rank) + ".bin";
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp || fwrite(host.data(), 1, host.size(), fp) != host.size()) return fail(c, "write frames", -1);
    fclose(fp);
    zr_stats st;
    if ((rc = zr_get_stats(c, &st, sizeof st))) return fail(c, "zr_get_stats", rc);
    printf("rank %u/%u %s: %u frames, host enqueued them in %.2f ms, finished after %.2f ms; camera survivors %llu, shadow survivors %llu\n", rank,
           o.plain ? 1u : o.world, o.plain ? "plain" : o.mode.c_str(), o.frames, enq_ms, all_ms, (unsigned long long)st.survivors[1], (unsigned long long)st.survivors[0]);
    zr_destroy(c);
    return 0;
}

int main(int argc, char** argv)
{
    Opts o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--root") o.root = next();
        else if (a == "--world-file") o.world_file = next();
        else if (a == "--out") o.out = next();
        else if (a == "--id-file") o.id_file = next();
        else if (a == "--mode") o.mode = next();
        else if (a == "--size") { if (sscanf(next(), "%ux%u", &o.W, &o.H) != 2) return 2; }
        else if (a == "--shadow") o.SD = (uint32_t)atoi(next());
        else if (a == "--frames") o.frames = (uint32_t)atoi(next());
        else if (a == "--world") o.world = (uint32_t)atoi(next());
        else if (a == "--ranks") { if (sscanf(next(), "%u:%u", &o.r0, &o.r1) != 2) return 2; }
        else if (a == "--roll-stage-step") o.roll_stage = (float)atof(next());
        else if (a == "--roll-light-step") o.roll_light = (float)atof(next());
        else if (a == "--plain") o.plain = true;
        else if (a == "--device-per-rank") o.device_per_rank = true;
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (o.world_file.empty() || o.out.empty() || (!o.plain && (o.id_file.empty() || o.r1 <= o.r0 || o.r1 > o.world))) { fprintf(stderr, "dist_driver: see the head of dist_driver.cpp\n"); return 2; }
    if (o.plain) return run_rank(o, 0);
    std::vector<std::thread> th;
    std::vector<int> rcs(o.r1 - o.r0, 0);
    for (uint32_t r = o.r0; r < o.r1; ++r) th.emplace_back([&, r]() { rcs[r - o.r0] = run_rank(o, r); });
    for (auto& t : th) t.join();
    for (int rc : rcs) if (rc) return rc;
    return 0;
}
