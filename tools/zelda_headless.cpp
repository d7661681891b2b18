// zelda_headless — a native caller of the C-ABI: the engine's main loop without a window.
//
// Replaces XkZeldaEngineApp::Run / MainTick / DrawFrame (ZE:1576, 1743, 1940) for a host that has no Vulkan surface: create the
// renderer, point it at the engine's content tree, load a world (file, or wait for one over the livelink), then per frame
//     zr_livelink_poll (bReloadScene pickup, ZE:1943-1951) -> zr_world_update_uniforms (UpdateWorld + UpdateUniformBuffer) -> zr_render
// and finally write the last frame as a PPM.  Uses nothing but include/zelda_render.h; links -lzelda_render.
//
//   zelda_headless --root DIR [--world FILE.json] [--livelink PORT [--wait-ms MS]] [--meshlet FILE.meshlet --profab NAME]
//                  [--size WxH] [--shadow N] [--frames N] [--roll-light-step F] [--debug-view V] [--out FRAME.ppm] [--device D]
#include "../include/zelda_render.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static int fail(zr_ctx* c, const char* what, int rc)
{
    fprintf(stderr, "zelda_headless: %s failed (%d): %s\n", what, rc, zr_last_error(c));
    if (c) zr_destroy(c);
    return 1;
}

int main(int argc, char** argv)
{
    std::string root = ".", world, meshlet, profab = "meshlet", out;
    uint32_t W = 1920, H = 1080, SD = 1024, frames = 1, debug_view = 0;
    int port = -1, wait_ms = 10000, device = 0;
    float roll_step = 0.0f;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--root") root = next();
        else if (a == "--world") world = next();
        else if (a == "--meshlet") meshlet = next();
        else if (a == "--profab") profab = next();
        else if (a == "--out") out = next();
        else if (a == "--size") { if (sscanf(next(), "%ux%u", &W, &H) != 2) { fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (a == "--shadow") SD = (uint32_t)atoi(next());
        else if (a == "--frames") frames = (uint32_t)atoi(next());
        else if (a == "--debug-view") debug_view = (uint32_t)atoi(next());
        else if (a == "--livelink") port = atoi(next());
        else if (a == "--wait-ms") wait_ms = atoi(next());
        else if (a == "--device") device = atoi(next());
        else if (a == "--roll-light-step") roll_step = (float)atof(next());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    zr_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.width = W; cfg.height = H; cfg.shadow_dim = SD; cfg.debug_view = debug_view; cfg.device = device; cfg.tile_world = 1;
    zr_ctx* c = nullptr;
    int rc = zr_create(&cfg, &c);
    if (rc) { fprintf(stderr, "zelda_headless: zr_create failed (%d): no usable HIP device? (there is no CPU fallback)\n", rc); return 1; }
    if ((rc = zr_set_asset_root(c, root.c_str()))) return fail(c, "zr_set_asset_root", rc);
    if (!meshlet.empty()) {          // the XkMeshIndirect path: a .meshlet file written by the ZeldaMeshlet tool, registered as a Profab
        uint32_t mesh = 0;
        if ((rc = zr_load_meshlet_file(c, meshlet.c_str(), &mesh))) return fail(c, "zr_load_meshlet_file", rc);
        if ((rc = zr_profab_register(c, profab.c_str(), mesh, nullptr))) return fail(c, "zr_profab_register", rc);
    }
    bool have_world = false;
    if (!world.empty()) {
        if ((rc = zr_world_load_file(c, world.c_str()))) return fail(c, "zr_world_load_file", rc);
        have_world = true;
    }
    if (port >= 0) {
        if ((rc = zr_livelink_serve(c, (uint16_t)port))) return fail(c, "zr_livelink_serve", rc);
        uint16_t p = 0; zr_livelink_port(c, &p);
        printf("[Socket] listening on port %u\n", (unsigned)p); fflush(stdout);
        for (int waited = 0; !have_world && waited < wait_ms; waited += 10) {
            int reloaded = 0;
            if ((rc = zr_livelink_poll(c, &reloaded))) return fail(c, "zr_livelink_poll", rc);
            if (reloaded) have_world = true; else std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    if (!have_world) { fprintf(stderr, "zelda_headless: no world (give --world FILE or send one to --livelink PORT)\n"); zr_destroy(c); return 1; }

    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t f = 0; f < frames; ++f) {
        int reloaded = 0;
        if (port >= 0 && (rc = zr_livelink_poll(c, &reloaded))) return fail(c, "zr_livelink_poll", rc);
        if ((rc = zr_world_update_uniforms(c, 0.0f, roll_step * (float)f, 0.016f * (float)f))) return fail(c, "zr_world_update_uniforms", rc);
        if ((rc = zr_render(c))) return fail(c, "zr_render", rc);
    }
    if ((rc = zr_finish(c))) return fail(c, "zr_finish", rc);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    zr_stats st;
    if ((rc = zr_get_stats(c, &st, sizeof st))) return fail(c, "zr_get_stats", rc);
    uint32_t n_obj = 0; zr_object_count(c, &n_obj);
    printf("frames %u  %.3f ms/frame  objects %u  meshlet-instances %llu  camera survivors %llu  covered pixels %llu\n", frames, ms / frames, n_obj,
           (unsigned long long)st.work_items[1], (unsigned long long)st.survivors[1], (unsigned long long)st.covered_pixels);
    if (!out.empty()) {
        std::vector<uint8_t> rgba((size_t)W * H * 4);
        if ((rc = zr_read_color(c, rgba.data(), rgba.size()))) return fail(c, "zr_read_color", rc);
        FILE* fp = fopen(out.c_str(), "wb");
        if (!fp) { fprintf(stderr, "zelda_headless: cannot write %s\n", out.c_str()); zr_destroy(c); return 1; }
        fprintf(fp, "P6\n%u %u\n255\n", W, H);
        std::vector<uint8_t> rgb((size_t)W * H * 3);
        for (size_t i = 0; i < (size_t)W * H; ++i) { rgb[3 * i] = rgba[4 * i]; rgb[3 * i + 1] = rgba[4 * i + 1]; rgb[3 * i + 2] = rgba[4 * i + 2]; }
        fwrite(rgb.data(), 1, rgb.size(), fp);
        fclose(fp);
    }
    zr_destroy(c);
    return 0;
}
