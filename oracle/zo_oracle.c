/*
 * zo_oracle.c — scalar CPU restatement of ZeldaEngine's deferred render path.
 * TEST INFRASTRUCTURE ONLY (see zo_math.h).  PARITY UNPINNED (no reference vectors exist).
 *
 * Citations: ZE = Engine/ZeldaEngine/ZeldaEngine.cpp, SH = Engine/ZeldaEngine/Shaders/.
 * Raster rules the reference relies on implicitly are Vulkan 1.3's (top-left fill rule,
 * pixel centres at +0.5, facing from the sign of the framebuffer-space area, D32 depth
 * bias r = 2^(e-23), UNORM / A2R10G10B10 / fp16 stores, 2x2-quad fine derivatives).
 * Where Vulkan leaves a choice to the implementation, this file picks ONE and says so:
 *   - sub-pixel precision: 8 bits (vertices snapped to 1/256 px), exact int64 edge functions;
 *   - depth and attributes are plane equations anchored at vertex 0;
 *   - clipping: near plane (z >= 0) + a 4x guard band in x,y; the far plane and z < 0 with
 *     w > 0 are per-fragment depth clips (equivalent for planar primitives);
 *   - derivatives: fine, with helper invocations extrapolating the same triangle;
 *   - bilinear weights in full fp32; cube faces are clamp-to-edge, not seamless.
 */
#include "zo_oracle.h"
#include "zo_math.h"

#include <omp.h>
#include <stdlib.h>
#include <stdio.h>

#define ZO_GUARD 4.0f
#define ZO_EMPTY 0xFFFFFFFFu

/* ------------------------------------------------------------------ scene */

typedef struct { XkVertex* v; uint32_t nv; uint32_t* idx; uint32_t ni; } zo_mesh;
typedef struct { uint8_t* px; uint32_t w, h; int constant; int levels; uint8_t* mip[16]; } zo_tex;   /* mip[0] == px */
typedef struct {
    int mesh; uint32_t n_inst; int instanced; XkInstanceData* inst;
    zo_tex tex[7]; uint32_t prim_base;
} zo_object;
typedef struct { float R[9]; float t[3]; float s; } zo_xform;   /* R column-major 3x3 */

struct zo_ctx {
    uint32_t W, H, SD;
    zo_mesh* meshes; int n_meshes;
    zo_object* objects; int n_objects;
    int* order; int order_valid;                 /* draw order: non-instanced then instanced */
    XkUniformBufferMVP cam, shadow; XkView view;
    /* cubemap: levels of 6 faces RGBA8 sRGB */
    uint8_t** cube; uint32_t cube_dim; int cube_levels;
    float srgb_lut[256];
    /* targets */
    float* depth; uint32_t* scene_color; uint32_t* gA; uint32_t* gB; uint32_t* gC; uint64_t* gD;
    uint32_t* vis; float* shadowmap; uint8_t* color;
    uint64_t covered;
    int threads;                /* OpenMP team of the per-pixel stages (resolve, lighting); 1 unless zo_set_threads */
    /* skydome + background passes (ZE:2657-2744, 3681-3699) */
    zo_mesh sky_mesh; zo_tex sky_tex; int sky_set, sky_enabled; zo_tex bg_tex; int bg_set, bg_enabled;
    uint32_t* overlay; float* main_depth; uint32_t* sky_vis;
    /* forward variant (SH/Base.frag; the engine built with ENABLE_DEFERRED_SHADING false, ZE:93): the fragment's unquantised inputs */
    int forward; struct zo_fsurf* fwd;
};
/* what Base.frag:48-60 holds before it lights the fragment: nothing has been through a render-target format */
typedef struct zo_fsurf { zo_v3 BaseColor; float Metallic, Roughness; zo_v3 Normal, AmbientOcclution, P, VertexColor; } zo_fsurf;

static int zo_idx_clamp(float f, int hi);
static uint8_t zo_srgb_encode(float l);

static const uint8_t zo_default_texel[7][4] = {   /* ZE:4951-4978: grey, black, white, normal, white, black, white */
    {127,127,127,255}, {0,0,0,255}, {255,255,255,255}, {127,127,255,255}, {255,255,255,255}, {0,0,0,255}, {255,255,255,255}
};

static void zo_default_lights(XkView* v)
{   /* XkLight() default constructor, ZE:779 */
    XkLight d; memset(&d, 0, sizeof d);
    d.Color[0] = d.Color[1] = d.Color[2] = d.Color[3] = 1.0f;
    d.Direction[2] = 1.0f; d.Direction[3] = 1.0f;
    for (int i = 0; i < XK_MAX_DIRECTIONAL_LIGHTS_NUM; ++i) v->DirectionalLights[i] = d;
    for (int i = 0; i < XK_MAX_POINT_LIGHTS_NUM; ++i) v->PointLights[i] = d;
    for (int i = 0; i < XK_MAX_SPOT_LIGHTS_NUM; ++i) v->SpotLights[i] = d;
}

static float zo_srgb_decode(uint32_t c)
{
    double x = (double)c / 255.0;
    double l = (x <= 0.04045) ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
    return (float)l;
}
static uint8_t zo_srgb_encode(float l)
{
    double x = (double)l;
    if (!(x > 0.0)) x = 0.0;
    if (x > 1.0) x = 1.0;
    double s = (x <= 0.0031308) ? 12.92 * x : 1.055 * pow(x, 1.0 / 2.4) - 0.055;
    return (uint8_t)floor(s * 255.0 + 0.5);
}

static void zo_free_cube(zo_ctx* c)
{
    if (c->cube) { for (int l = 0; l < c->cube_levels; ++l) free(c->cube[l]); free(c->cube); }
    c->cube = NULL; c->cube_levels = 0; c->cube_dim = 0;
}

zo_ctx* zo_create(uint32_t W, uint32_t H, uint32_t SD)
{
    zo_ctx* c = (zo_ctx*)calloc(1, sizeof *c);
    c->W = W; c->H = H; c->SD = SD ? SD : XK_SHADOWMAP_DIM;
    size_t n = (size_t)W * H;
    c->depth = (float*)malloc(n * 4); c->scene_color = (uint32_t*)malloc(n * 4);
    c->gA = (uint32_t*)malloc(n * 4); c->gB = (uint32_t*)malloc(n * 4); c->gC = (uint32_t*)malloc(n * 4);
    c->gD = (uint64_t*)malloc(n * 8); c->vis = (uint32_t*)malloc(n * 4); c->color = (uint8_t*)calloc(n, 4);
    c->shadowmap = (float*)malloc((size_t)c->SD * c->SD * 4);
    c->overlay = (uint32_t*)calloc(n, 4); c->main_depth = (float*)malloc(n * 4); c->sky_vis = (uint32_t*)malloc(n * 4);
    c->sky_enabled = c->bg_enabled = 1;
    c->threads = 1;
    for (int i = 0; i < 256; ++i) c->srgb_lut[i] = zo_srgb_decode((uint32_t)i);
    zo_default_lights(&c->view);
    zo_set_cubemap(c, NULL, 0);
    return c;
}

void zo_scene_clear(zo_ctx* c)
{
    for (int i = 0; i < c->n_objects; ++i) {
        free(c->objects[i].inst);
        for (int t = 0; t < 7; ++t) for (int l = 0; l < c->objects[i].tex[t].levels; ++l) free(c->objects[i].tex[t].mip[l]);
    }
    free(c->objects); c->objects = NULL; c->n_objects = 0;
    for (int i = 0; i < c->n_meshes; ++i) { free(c->meshes[i].v); free(c->meshes[i].idx); }
    free(c->meshes); c->meshes = NULL; c->n_meshes = 0;
    free(c->order); c->order = NULL; c->order_valid = 0;
}

void zo_destroy(zo_ctx* c)
{
    if (!c) return;
    zo_scene_clear(c); zo_free_cube(c);
    free(c->depth); free(c->scene_color); free(c->gA); free(c->gB); free(c->gC); free(c->gD);
    free(c->vis); free(c->color); free(c->shadowmap); free(c->overlay); free(c->main_depth); free(c->sky_vis); free(c->fwd);
    zo_set_skydome(c, NULL, 0, NULL, 0, NULL); zo_set_background(c, NULL);
    free(c);
}

int zo_mesh_create(zo_ctx* c, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni)
{
    if (!v || !idx || ni % 3) return -1;
    for (uint32_t i = 0; i < ni; ++i) if (idx[i] >= nv) return -1;
    c->meshes = (zo_mesh*)realloc(c->meshes, sizeof(zo_mesh) * (size_t)(c->n_meshes + 1));
    zo_mesh* m = &c->meshes[c->n_meshes];
    m->v = (XkVertex*)malloc(sizeof(XkVertex) * nv); memcpy(m->v, v, sizeof(XkVertex) * nv); m->nv = nv;
    m->idx = (uint32_t*)malloc(4u * ni); memcpy(m->idx, idx, 4u * ni); m->ni = ni;
    return c->n_meshes++;
}

/* RHIGenerateMipmaps (ZE:6348-6433): level l+1 = vkCmdBlitImage(LINEAR) of level l at half size (floor, min 1);
 * mipLevels = floor(log2(max(w, h))) + 1 (ZE:6887).  Filtering is done on decoded values (sRGB for the base-colour slot,
 * whose format is R8G8B8A8_SRGB, ZE:5878) and re-encoded. */
static float zo_decode8(const zo_ctx* c, uint8_t v, int srgb) { return srgb ? c->srgb_lut[v] : (float)v / 255.0f; }
static void zo_build_mips(zo_ctx* c, zo_tex* tx, int srgb)
{
    uint32_t m = tx->w > tx->h ? tx->w : tx->h;
    int levels = 1; while (m > 1) { m >>= 1; levels++; }
    tx->levels = levels; tx->mip[0] = tx->px;
    uint32_t sw = tx->w, sh = tx->h;
    for (int l = 1; l < levels; ++l) {
        uint32_t dw = sw > 1 ? sw >> 1 : 1, dh = sh > 1 ? sh >> 1 : 1;
        const uint8_t* src = tx->mip[l - 1];
        uint8_t* dst = (uint8_t*)malloc((size_t)dw * dh * 4);
        float kx = (float)sw / (float)dw, ky = (float)sh / (float)dh;
        for (uint32_t y = 0; y < dh; ++y) for (uint32_t x = 0; x < dw; ++x) {
            float fu = fmaf((float)x + 0.5f, kx, -0.5f), fv = fmaf((float)y + 0.5f, ky, -0.5f);
            float fx = floorf(fu), fy = floorf(fv), a = fu - fx, b = fv - fy;
            int x0 = zo_idx_clamp(fx, (int)sw - 1), x1 = zo_idx_clamp(fx + 1.0f, (int)sw - 1);
            int y0 = zo_idx_clamp(fy, (int)sh - 1), y1 = zo_idx_clamp(fy + 1.0f, (int)sh - 1);
            const uint8_t* p00 = src + ((size_t)y0 * sw + x0) * 4; const uint8_t* p10 = src + ((size_t)y0 * sw + x1) * 4;
            const uint8_t* p01 = src + ((size_t)y1 * sw + x0) * 4; const uint8_t* p11 = src + ((size_t)y1 * sw + x1) * 4;
            for (int ch = 0; ch < 4; ++ch) {
                int sr = srgb && ch < 3;
                float t00 = zo_decode8(c, p00[ch], sr), t10 = zo_decode8(c, p10[ch], sr), t01 = zo_decode8(c, p01[ch], sr), t11 = zo_decode8(c, p11[ch], sr);
                float top = fmaf(a, t10 - t00, t00), bot = fmaf(a, t11 - t01, t01);
                float v = fmaf(b, bot - top, top);
                dst[((size_t)y * dw + x) * 4 + ch] = sr ? zo_srgb_encode(v) : (uint8_t)zo_unorm(v, 255.0f);
            }
        }
        tx->mip[l] = dst; sw = dw; sh = dh;
    }
}

int zo_object_add(zo_ctx* c, int mesh, const zo_material* mat, const XkInstanceData* inst, uint32_t n_inst)
{
    if (mesh < 0 || mesh >= c->n_meshes) return -1;
    c->objects = (zo_object*)realloc(c->objects, sizeof(zo_object) * (size_t)(c->n_objects + 1));
    zo_object* o = &c->objects[c->n_objects];
    memset(o, 0, sizeof *o);
    o->mesh = mesh; o->instanced = n_inst > 0; o->n_inst = n_inst ? n_inst : 1;
    o->inst = (XkInstanceData*)calloc(o->n_inst, sizeof(XkInstanceData));
    if (n_inst) memcpy(o->inst, inst, sizeof(XkInstanceData) * n_inst);
    for (int t = 0; t < 7; ++t) {
        zo_tex* tx = &o->tex[t];
        if (mat && mat->tex[t].rgba8) {
            tx->w = mat->tex[t].width; tx->h = mat->tex[t].height;
            size_t n = (size_t)tx->w * tx->h * 4;
            tx->px = (uint8_t*)malloc(n); memcpy(tx->px, mat->tex[t].rgba8, n);
            tx->constant = 1;
            for (size_t i = 4; i < n; ++i) if (tx->px[i] != tx->px[i & 3]) { tx->constant = 0; break; }
            zo_build_mips(c, tx, t == 0);
        } else {
            tx->w = tx->h = 1; tx->px = (uint8_t*)malloc(4); memcpy(tx->px, zo_default_texel[t], 4); tx->constant = 1;
            tx->levels = 1; tx->mip[0] = tx->px;
        }
    }
    c->order_valid = 0;
    return c->n_objects++;
}

/* Draw order of the deferred-scene pass: all non-instanced draws, then all instanced draws (ZE:3445-3476). */
static void zo_finalize_order(zo_ctx* c)
{
    if (c->order_valid) return;
    free(c->order); c->order = (int*)malloc(sizeof(int) * (size_t)(c->n_objects + 1));
    int k = 0; uint32_t base = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < c->n_objects; ++i)
            if (c->objects[i].instanced == pass) {
                c->order[k++] = i; c->objects[i].prim_base = base;
                base += c->objects[i].n_inst * (c->meshes[c->objects[i].mesh].ni / 3);
            }
    c->order_valid = 1;
}

/* ------------------------------------------------------------------ cubemap (ZE:5908-6150, mips ZE:6348-6433) */

int zo_set_cubemap(zo_ctx* c, const uint8_t* const faces[6], uint32_t dim)
{
    static const uint8_t grey[4] = {127,127,127,255};
    zo_free_cube(c);
    if (!faces) dim = 1;
    if (dim == 0) return -1;
    int levels = 1; for (uint32_t d = dim; d > 1; d >>= 1) levels++;   /* floor(log2(dim)) + 1, ZE:6887 */
    c->cube = (uint8_t**)calloc((size_t)levels, sizeof(uint8_t*));
    c->cube_levels = levels; c->cube_dim = dim;
    size_t fsz = (size_t)dim * dim * 4;
    c->cube[0] = (uint8_t*)malloc(fsz * 6);
    for (int f = 0; f < 6; ++f) {
        if (faces) memcpy(c->cube[0] + fsz * f, faces[f], fsz);
        else memcpy(c->cube[0] + fsz * f, grey, 4);
    }
    uint32_t d = dim;
    for (int l = 1; l < levels; ++l) {            /* vkCmdBlitImage LINEAR from level l-1: 2x2 box in linear light */
        uint32_t nd = d > 1 ? d >> 1 : 1;
        c->cube[l] = (uint8_t*)malloc((size_t)nd * nd * 4 * 6);
        for (int f = 0; f < 6; ++f) {
            const uint8_t* src = c->cube[l - 1] + (size_t)d * d * 4 * f;
            uint8_t* dst = c->cube[l] + (size_t)nd * nd * 4 * f;
            for (uint32_t y = 0; y < nd; ++y) for (uint32_t x = 0; x < nd; ++x) {
                uint32_t x0 = 2 * x, x1 = (2 * x + 1 < d) ? 2 * x + 1 : d - 1;
                uint32_t y0 = 2 * y, y1 = (2 * y + 1 < d) ? 2 * y + 1 : d - 1;
                const uint8_t* p00 = src + ((size_t)y0 * d + x0) * 4; const uint8_t* p10 = src + ((size_t)y0 * d + x1) * 4;
                const uint8_t* p01 = src + ((size_t)y1 * d + x0) * 4; const uint8_t* p11 = src + ((size_t)y1 * d + x1) * 4;
                for (int ch = 0; ch < 3; ++ch) {
                    float a = (c->srgb_lut[p00[ch]] + c->srgb_lut[p10[ch]]) + (c->srgb_lut[p01[ch]] + c->srgb_lut[p11[ch]]);
                    dst[((size_t)y * nd + x) * 4 + ch] = zo_srgb_encode(a * 0.25f);
                }
                uint32_t al = (uint32_t)p00[3] + p10[3] + p01[3] + p11[3];
                dst[((size_t)y * nd + x) * 4 + 3] = (uint8_t)((al + 2) >> 2);
            }
        }
        d = nd;
    }
    c->view.LightsCount[3] = levels;              /* CubemapMaxMips, ZE:4308 */
    return 0;
}

static int zo_idx_clamp(float f, int hi)          /* NaN-safe float -> clamped int */
{
    f = fminf(fmaxf(f, 0.0f), (float)hi);
    return (int)f;
}

static zo_v3 zo_cube_fetch(const zo_ctx* c, int level, int face, int x, int y)
{
    uint32_t d = c->cube_dim >> level; if (d == 0) d = 1;
    const uint8_t* p = c->cube[level] + ((size_t)d * d * (size_t)face + (size_t)y * d + (size_t)x) * 4;
    return zo_v3make(c->srgb_lut[p[0]], c->srgb_lut[p[1]], c->srgb_lut[p[2]]);
}

static zo_v3 zo_cube_bilinear(const zo_ctx* c, int level, int face, float s, float t)
{
    uint32_t d = c->cube_dim >> level; if (d == 0) d = 1;
    float u = fmaf(s, (float)d, -0.5f), v = fmaf(t, (float)d, -0.5f);
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int x0 = zo_idx_clamp(fu, (int)d - 1), x1 = zo_idx_clamp(fu + 1.0f, (int)d - 1);
    int y0 = zo_idx_clamp(fv, (int)d - 1), y1 = zo_idx_clamp(fv + 1.0f, (int)d - 1);
    zo_v3 t00 = zo_cube_fetch(c, level, face, x0, y0), t10 = zo_cube_fetch(c, level, face, x1, y0);
    zo_v3 t01 = zo_cube_fetch(c, level, face, x0, y1), t11 = zo_cube_fetch(c, level, face, x1, y1);
    zo_v3 top = zo_v3make(fmaf(a, t10.x - t00.x, t00.x), fmaf(a, t10.y - t00.y, t00.y), fmaf(a, t10.z - t00.z, t00.z));
    zo_v3 bot = zo_v3make(fmaf(a, t11.x - t01.x, t01.x), fmaf(a, t11.y - t01.y, t01.y), fmaf(a, t11.z - t01.z, t01.z));
    return zo_v3make(fmaf(b, bot.x - top.x, top.x), fmaf(b, bot.y - top.y, top.y), fmaf(b, bot.z - top.z, top.z));
}

/* textureLod(samplerCube, R, lod): Vulkan face selection (z wins ties over y over x), trilinear. */
static zo_v3 zo_cube_sample(const zo_ctx* c, zo_v3 R, float lod)
{
    float ax = fabsf(R.x), ay = fabsf(R.y), az = fabsf(R.z);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) { ma = az; if (R.z >= 0.0f) { face = 4; sc = R.x; tc = -R.y; } else { face = 5; sc = -R.x; tc = -R.y; } }
    else if (ay >= ax)        { ma = ay; if (R.y >= 0.0f) { face = 2; sc = R.x; tc = R.z; }  else { face = 3; sc = R.x; tc = -R.z; } }
    else                      { ma = ax; if (R.x >= 0.0f) { face = 0; sc = -R.z; tc = -R.y; } else { face = 1; sc = R.z; tc = -R.y; } }
    zo_v3 st = zo_div_scalar(zo_v3make(sc, tc, 0.0f), ma);     /* (sc, tc) / ma */
    float s = fmaf(st.x, 0.5f, 0.5f), t = fmaf(st.y, 0.5f, 0.5f);
    float maxl = (float)(c->cube_levels - 1);
    float l = fminf(fmaxf(lod, 0.0f), maxl);
    float fl = floorf(l);
    int l0 = (int)fl, l1 = l0 + 1 < c->cube_levels ? l0 + 1 : c->cube_levels - 1;
    float w = l - fl;
    zo_v3 c0 = zo_cube_bilinear(c, l0, face, s, t), c1 = zo_cube_bilinear(c, l1, face, s, t);
    return zo_v3make(fmaf(w, c1.x - c0.x, c0.x), fmaf(w, c1.y - c0.y, c0.y), fmaf(w, c1.z - c0.z, c0.z));
}

/* ------------------------------------------------------------------ glm restatement + UpdateUniformBuffer */

static float zo_radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

static void zo_perspective(float fovy, float aspect, float zn, float zf, float* m) /* glm::perspectiveRH_ZO */
{
    float t = tanf(fovy / 2.0f);
    memset(m, 0, 64);
    m[0] = 1.0f / (aspect * t);
    m[5] = 1.0f / t;
    m[10] = zf / (zn - zf);
    m[11] = -1.0f;
    m[14] = -(zf * zn) / (zf - zn);
}

static void zo_lookat(zo_v3 eye, zo_v3 center, zo_v3 up, float* m)                /* glm::lookAtRH */
{
    zo_v3 f = zo_normalize_ieee(zo_sub(center, eye));       /* glm on the host */
    zo_v3 s = zo_normalize_ieee(zo_cross(f, up));
    zo_v3 u = zo_cross(s, f);
    m[0] = s.x; m[4] = s.y; m[8] = s.z;
    m[1] = u.x; m[5] = u.y; m[9] = u.z;
    m[2] = -f.x; m[6] = -f.y; m[10] = -f.z;
    m[3] = 0; m[7] = 0; m[11] = 0;
    m[12] = -zo_dot(s, eye); m[13] = -zo_dot(u, eye); m[14] = zo_dot(f, eye); m[15] = 1.0f;
}

static void zo_rotate_z(float angle, float* m)        /* glm::rotate(mat4(1), angle, vec3(0,0,1)) */
{
    float c = cosf(angle), s = sinf(angle);
    float t2 = 1.0f - c;
    memset(m, 0, 64);
    m[0] = c; m[1] = s; m[4] = -s; m[5] = c; m[10] = c + t2; m[15] = 1.0f;
}

/* UpdateWorld (ZE:4294-4308) + UpdateUniformBuffer (ZE:4585-4664), game mode (bars = 0, ZE:4575-4579) */
void zo_update_uniforms(zo_ctx* c, const zo_camera* cam, const XkLight* dir, uint32_t n_dir,
                        const XkLight* point, uint32_t n_point, const XkLight* spot, uint32_t n_spot,
                        float roll_stage, float roll_light, float time)
{
    XkView* V = &c->view;
    for (uint32_t i = 0; i < n_dir && i < XK_MAX_DIRECTIONAL_LIGHTS_NUM; ++i) V->DirectionalLights[i] = dir[i];
    for (uint32_t i = 0; i < n_point && i < XK_MAX_POINT_LIGHTS_NUM; ++i) V->PointLights[i] = point[i];
    for (uint32_t i = 0; i < n_spot && i < XK_MAX_SPOT_LIGHTS_NUM; ++i) V->SpotLights[i] = spot[i];
    V->LightsCount[0] = (int32_t)n_dir; V->LightsCount[1] = (int32_t)n_point; V->LightsCount[2] = (int32_t)n_spot;
    V->LightsCount[3] = c->cube_levels;

    zo_v3 pos = zo_v3make(cam->Position[0], cam->Position[1], cam->Position[2]);
    zo_v3 look = zo_v3make(cam->Lookat[0], cam->Lookat[1], cam->Lookat[2]);
    zo_v3 up = zo_v3make(0, 0, 1);
    zo_v3 lightPos = zo_v3make(V->DirectionalLights[0].Position[0], V->DirectionalLights[0].Position[1],
                               V->DirectionalLights[0].Position[2]);
    float l2w[16], sview[16], sproj[16], cview[16], cproj[16];
    zo_rotate_z(roll_stage, l2w);
    zo_lookat(lightPos, zo_v3make(0, 0, 0), up, sview);
    zo_perspective(zo_radians(cam->FOV), 1.0f, cam->zNear, cam->zFar, sproj);
    sproj[5] *= -1.0f;
    zo_lookat(pos, look, up, cview);
    zo_perspective(zo_radians(cam->FOV), (float)c->W / (float)c->H, cam->zNear, cam->zFar, cproj);

    memcpy(c->cam.Model, l2w, 64); memcpy(c->cam.View, cview, 64); memcpy(c->cam.Proj, cproj, 64);
    c->cam.Proj[5] *= -1.0f;
    zo_mat4_mul(cproj, cview, V->ViewProjSpace);      /* un-flipped proj, ZE:4632 */
    zo_mat4_mul(sproj, sview, V->ShadowmapSpace);     /* no model matrix, ZE:4633 */
    memcpy(V->LocalToWorld, l2w, 64);
    V->CameraInfo[0] = pos.x; V->CameraInfo[1] = pos.y; V->CameraInfo[2] = pos.z; V->CameraInfo[3] = cam->FOV;
    V->ViewportInfo[0] = (float)c->W; V->ViewportInfo[1] = (float)c->H; V->ViewportInfo[2] = 0; V->ViewportInfo[3] = 0;
    uint32_t N = (uint32_t)V->LightsCount[1];
    for (uint32_t i = 0; i < N && i < XK_MAX_POINT_LIGHTS_NUM; ++i) {  /* spiral, ZE:4637-4646 */
        float radians = ((float)i / (float)N) * 360.0f - roll_light * 100.0f;
        float distance = ((float)i / (float)N) * 5.0f + 2.5f;
        V->PointLights[i].Position[0] = sinf(zo_radians(radians)) * distance;
        V->PointLights[i].Position[1] = cosf(zo_radians(radians)) * distance;
        V->PointLights[i].Position[2] = 1.5f;
        V->PointLights[i].Position[3] = 1.0f;
    }
    V->Time = time; V->zNear = cam->zNear; V->zFar = cam->zFar;
    memcpy(c->shadow.Model, l2w, 64); memcpy(c->shadow.View, sview, 64); memcpy(c->shadow.Proj, sproj, 64);
}

void zo_set_frame(zo_ctx* c, const XkUniformBufferMVP* cam, const XkUniformBufferMVP* sh, const XkView* v)
{ c->cam = *cam; c->shadow = *sh; c->view = *v; }
void zo_get_frame(zo_ctx* c, XkUniformBufferMVP* cam, XkUniformBufferMVP* sh, XkView* v)
{ if (cam) *cam = c->cam; if (sh) *sh = c->shadow; if (v) *v = c->view; }

/* ------------------------------------------------------------------ vertex stage */

/* MakeRotMatrix (SH/Common.glsl:60-87): rotMat = mz*my*mx; note mx(R.x) turns about Y, my(R.y) about Z, mz(R.z) about X */
static void zo_mat3_mul(const float* A, const float* B, float* C)
{
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r)
        C[c * 3 + r] = fmaf(A[6 + r], B[c * 3 + 2], fmaf(A[3 + r], B[c * 3 + 1], A[r] * B[c * 3 + 0]));
}
static void zo_make_rot(const float e[3], float* R)
{
    float s, c, mx[9], my[9], mz[9], t[9];
    zo_sincosf(e[0], &s, &c);
    mx[0] = c; mx[1] = 0; mx[2] = s;  mx[3] = 0; mx[4] = 1; mx[5] = 0;  mx[6] = -s; mx[7] = 0; mx[8] = c;
    zo_sincosf(e[1], &s, &c);
    my[0] = c; my[1] = s; my[2] = 0;  my[3] = -s; my[4] = c; my[5] = 0;  my[6] = 0; my[7] = 0; my[8] = 1;
    zo_sincosf(e[2], &s, &c);
    mz[0] = 1; mz[1] = 0; mz[2] = 0;  mz[3] = 0; mz[4] = c; mz[5] = s;  mz[6] = 0; mz[7] = -s; mz[8] = c;
    zo_mat3_mul(mz, my, t);
    zo_mat3_mul(t, mx, R);
}
static void zo_xform_make(const zo_object* o, uint32_t i, zo_xform* x)
{
    const XkInstanceData* d = &o->inst[i];
    zo_make_rot(d->InstanceRotation, x->R);
    x->t[0] = d->InstancePosition[0]; x->t[1] = d->InstancePosition[1]; x->t[2] = d->InstancePosition[2];
    x->s = d->InstancePScale;
}
/* v * mat3(R): component j = dot(v, column j) */
static zo_v3 zo_rowvec_mat3(zo_v3 v, const float* R)
{
    return zo_v3make(fmaf(v.z, R[2], fmaf(v.y, R[1], v.x * R[0])),
                     fmaf(v.z, R[5], fmaf(v.y, R[4], v.x * R[3])),
                     fmaf(v.z, R[8], fmaf(v.y, R[7], v.x * R[6])));
}
/* Base.vert:26 / BaseInstanced.vert:70 / Shadowmap*.vert: object-space position fed to PVM */
static zo_v3 zo_vs_position(const XkVertex* v, const zo_xform* x, int instanced)
{
    zo_v3 p = zo_v3make(v->Position[0], v->Position[1], v->Position[2]);
    if (!instanced) return p;
    zo_v3 q = zo_rowvec_mat3(zo_scale(p, x->s), x->R);
    return zo_v3make(q.x + x->t[0], q.y + x->t[1], q.z + x->t[2]);
}
/* outNormal: (M * vec4(normalize(n), 1)).xyz [* mat3(rotMat)]  — w = 1 is the reference's quirk (Base.vert:29) */
static zo_v3 zo_vs_normal(const XkVertex* v, const zo_xform* x, int instanced, const float* M)
{
    zo_v3 n = zo_normalize(zo_v3make(v->Normal[0], v->Normal[1], v->Normal[2]));
    zo_v4 mn = zo_mat4_point(M, n);
    zo_v3 r = zo_v3make(mn.x, mn.y, mn.z);
    return instanced ? zo_rowvec_mat3(r, x->R) : r;
}

/* ------------------------------------------------------------------ rasteriser */

typedef struct { int32_t X, Y; float z, rw; } zo_sv;      /* snapped screen vertex */

static int zo_finite4(zo_v4 a)
{ return fabsf(a.x) <= 3.402823466e38f && fabsf(a.y) <= 3.402823466e38f && fabsf(a.z) <= 3.402823466e38f && fabsf(a.w) <= 3.402823466e38f; }

static zo_sv zo_project(zo_v4 c, float hw, float hh)
{
    zo_sv s;
    s.rw = 1.0f / c.w;                       /* perspective divide: one IEEE reciprocal, then multiplies */
    float nx = c.x * s.rw, ny = c.y * s.rw;
    float xs = fmaf(nx, hw, hw), ys = fmaf(ny, hh, hh);
    s.X = (int32_t)floorf(fmaf(xs, 256.0f, 0.5f));
    s.Y = (int32_t)floorf(fmaf(ys, 256.0f, 0.5f));
    s.z = c.z * s.rw;
    return s;
}

/* classification shared by the raster and the resolve: 0 discard, 1 fast path, 2 needs clipping */
static int zo_classify(const zo_v4 c[3])
{
    if (!zo_finite4(c[0]) || !zo_finite4(c[1]) || !zo_finite4(c[2])) return 0;
    if ((c[0].x < -c[0].w && c[1].x < -c[1].w && c[2].x < -c[2].w) || (c[0].x > c[0].w && c[1].x > c[1].w && c[2].x > c[2].w) ||
        (c[0].y < -c[0].w && c[1].y < -c[1].w && c[2].y < -c[2].w) || (c[0].y > c[0].w && c[1].y > c[1].w && c[2].y > c[2].w) ||
        (c[0].z < 0.0f && c[1].z < 0.0f && c[2].z < 0.0f) || (c[0].z > c[0].w && c[1].z > c[1].w && c[2].z > c[2].w))
        return 0;
    for (int i = 0; i < 3; ++i) {
        float g = ZO_GUARD * c[i].w;
        if (c[i].z < 0.0f || !(c[i].w > 0.0f) || fabsf(c[i].x) > g || fabsf(c[i].y) > g) return 2;
    }
    return 1;
}

typedef struct { zo_v4 c; } zo_cv;
/* Sutherland-Hodgman against one plane; d() >= 0 is inside.  Intersections always run inside -> outside. */
static float zo_plane_dist(zo_v4 c, int plane)
{
    switch (plane) {
    case 0: return c.z;                            /* near: z >= 0 */
    case 1: return fmaf(ZO_GUARD, c.w, c.x);       /* x >= -G w */
    case 2: return fmaf(ZO_GUARD, c.w, -c.x);      /* x <=  G w */
    case 3: return fmaf(ZO_GUARD, c.w, c.y);
    default: return fmaf(ZO_GUARD, c.w, -c.y);
    }
}
static zo_v4 zo_lerp4(zo_v4 in, zo_v4 out, float t)
{
    zo_v4 r;
    r.x = fmaf(t, out.x - in.x, in.x); r.y = fmaf(t, out.y - in.y, in.y);
    r.z = fmaf(t, out.z - in.z, in.z); r.w = fmaf(t, out.w - in.w, in.w);
    return r;
}
static int zo_clip_polygon(const zo_v4 tri[3], zo_v4 out[10])
{
    zo_v4 a[10], b[10]; int na = 3;
    a[0] = tri[0]; a[1] = tri[1]; a[2] = tri[2];
    for (int plane = 0; plane < 5; ++plane) {
        int nb = 0;
        for (int i = 0; i < na; ++i) {
            zo_v4 p = a[i], q = a[(i + 1) % na];
            float dp = zo_plane_dist(p, plane), dq = zo_plane_dist(q, plane);
            int ip = dp >= 0.0f, iq = dq >= 0.0f;
            if (ip) b[nb++] = p;
            if (ip != iq) {
                if (ip) b[nb++] = zo_lerp4(p, q, dp / (dp - dq));
                else    b[nb++] = zo_lerp4(q, p, dq / (dq - dp));
            }
        }
        na = nb; if (na < 3) return 0;
        memcpy(a, b, sizeof(zo_v4) * (size_t)na);
    }
    for (int i = 0; i < na; ++i) { if (!(a[i].w > 0.0f)) return 0; out[i] = a[i]; }
    return na;
}

typedef struct {
    int64_t A; int sgn;
    int32_t X0, Y0;
    int32_t ex[3], ey[3];       /* oriented edge vectors (inside positive) */
    int32_t ax[3], ay[3];       /* edge start vertices */
    int tl[3];
    float a1, b1, a2, b2;       /* barycentric gradients per sub-pixel unit */
    float z0, gx, gy;
    float zlo, zhi;             /* depth range of the three vertices */
    int32_t minX, maxX, minY, maxY;
} zo_setup;

/* returns 0 if degenerate (or back-facing when cull_back) */
static int zo_tri_setup(const zo_sv v[3], int cull_back, zo_setup* s)
{
    int32_t dX1 = v[1].X - v[0].X, dY1 = v[1].Y - v[0].Y, dX2 = v[2].X - v[0].X, dY2 = v[2].Y - v[0].Y;
    int64_t A = (int64_t)dX1 * dY2 - (int64_t)dX2 * dY1;
    if (A == 0) return 0;
    /* Vulkan facing: a = -1/2 sum(x_i y_i+1 - x_i+1 y_i) = -A/2 in framebuffer coords; CCW front <=> a > 0 <=> A < 0 */
    if (cull_back && A > 0) return 0;
    s->A = A; s->sgn = A > 0 ? 1 : -1;
    s->X0 = v[0].X; s->Y0 = v[0].Y;
    static const int ea[3] = {1, 2, 0}, eb[3] = {2, 0, 1};
    for (int e = 0; e < 3; ++e) {
        int32_t dx = v[eb[e]].X - v[ea[e]].X, dy = v[eb[e]].Y - v[ea[e]].Y;
        s->ex[e] = s->sgn * dx; s->ey[e] = s->sgn * dy;
        s->ax[e] = v[ea[e]].X; s->ay[e] = v[ea[e]].Y;
        s->tl[e] = (s->ey[e] < 0) || (s->ey[e] == 0 && s->ex[e] > 0);
    }
    float invA = 1.0f / (float)A;
    s->a1 = (float)(v[2].Y - v[0].Y) * invA; s->b1 = (float)(v[0].X - v[2].X) * invA;
    s->a2 = (float)(v[0].Y - v[1].Y) * invA; s->b2 = (float)(v[1].X - v[0].X) * invA;
    float dz1 = v[1].z - v[0].z, dz2 = v[2].z - v[0].z;
    s->z0 = v[0].z;
    s->gx = fmaf(s->a2, dz2, s->a1 * dz1);
    s->gy = fmaf(s->b2, dz2, s->b1 * dz1);
    s->zlo = fminf(fminf(v[0].z, v[1].z), v[2].z);
    s->zhi = fmaxf(fmaxf(v[0].z, v[1].z), v[2].z);
    s->minX = v[0].X < v[1].X ? (v[0].X < v[2].X ? v[0].X : v[2].X) : (v[1].X < v[2].X ? v[1].X : v[2].X);
    s->maxX = v[0].X > v[1].X ? (v[0].X > v[2].X ? v[0].X : v[2].X) : (v[1].X > v[2].X ? v[1].X : v[2].X);
    s->minY = v[0].Y < v[1].Y ? (v[0].Y < v[2].Y ? v[0].Y : v[2].Y) : (v[1].Y < v[2].Y ? v[1].Y : v[2].Y);
    s->maxY = v[0].Y > v[1].Y ? (v[0].Y > v[2].Y ? v[0].Y : v[2].Y) : (v[1].Y > v[2].Y ? v[1].Y : v[2].Y);
    return 1;
}

static int zo_covers(const zo_setup* s, int32_t Px, int32_t Py)
{
    for (int e = 0; e < 3; ++e) {
        int64_t E = (int64_t)s->ex[e] * (Py - s->ay[e]) - (int64_t)s->ey[e] * (Px - s->ax[e]);
        /* sgn was folded into ex/ey, so E here is the oriented edge function */
        if (E < 0 || (E == 0 && !s->tl[e])) return 0;
    }
    return 1;
}

static float zo_depth_at(const zo_setup* s, int32_t Px, int32_t Py)
{
    float fx = (float)(Px - s->X0), fy = (float)(Py - s->Y0);
    float z = fmaf(s->gy, fy, fmaf(s->gx, fx, s->z0));
    /* the exact interpolant lies within the vertex depths; keep the rounded plane equation there too (a stated rule of
       this restatement: it bounds every fragment of a triangle by its vertices, which occlusion culling relies on) */
    z = fminf(fmaxf(z, s->zlo), s->zhi);
    return z + 0.0f;
}

/* D32 depth bias, vkCmdSetDepthBias(1.25, 0, 7.5) (ZE:3280-3287): o = m*slope + r*constant */
static float zo_depth_bias(const zo_setup* s, const zo_sv v[3])
{
    float m = fmaxf(fabsf(s->gx), fabsf(s->gy)) * 256.0f;
    float zm = fmaxf(fmaxf(fabsf(v[0].z), fabsf(v[1].z)), fabsf(v[2].z));
    uint32_t e = zo_f2u(zm) & 0x7F800000u;
    float r = (e > (23u << 23) && e < 0x7F800000u) ? zo_u2f(e - (23u << 23)) : 0.0f;
    return fmaf(m, 7.5f, r * 1.25f);
}

typedef struct { int shadow; uint32_t W, H; float* depth; uint32_t* vis; } zo_target;

static void zo_raster_sub(const zo_target* T, const zo_sv v[3], uint32_t prim)
{
    zo_setup s;
    if (!zo_tri_setup(v, !T->shadow, &s)) return;
    int32_t x0 = (s.minX - 128 + 255) >> 8, x1 = (s.maxX - 128) >> 8;
    int32_t y0 = (s.minY - 128 + 255) >> 8, y1 = (s.maxY - 128) >> 8;
    if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0;
    if (x1 > (int32_t)T->W - 1) x1 = (int32_t)T->W - 1; if (y1 > (int32_t)T->H - 1) y1 = (int32_t)T->H - 1;
    float bias = T->shadow ? zo_depth_bias(&s, v) : 0.0f;
    for (int32_t y = y0; y <= y1; ++y) for (int32_t x = x0; x <= x1; ++x) {
        int32_t Px = x * 256 + 128, Py = y * 256 + 128;
        if (!zo_covers(&s, Px, Py)) continue;
        float z = zo_depth_at(&s, Px, Py);
        if (!(z >= 0.0f && z <= 1.0f)) continue;             /* depth clip (depthClampEnable = FALSE, ZE:5100) */
        size_t p = (size_t)y * T->W + (size_t)x;
        if (T->shadow) {
            float zb = fminf(fmaxf(z + bias, 0.0f), 1.0f);
            if (zb <= T->depth[p]) T->depth[p] = zb;          /* LESS_OR_EQUAL, ZE:5141-5145 */
        } else if (z < T->depth[p]) {                         /* LESS */
            T->depth[p] = z; T->vis[p] = prim;
        }
    }
}

static void zo_raster_tri(const zo_target* T, const zo_v4 c[3], uint32_t prim)
{
    int cls = zo_classify(c);
    if (cls == 0) return;
    float hw = 0.5f * (float)T->W, hh = 0.5f * (float)T->H;
    zo_sv v[3];
    if (cls == 1) {
        for (int i = 0; i < 3; ++i) v[i] = zo_project(c[i], hw, hh);
        zo_raster_sub(T, v, prim);
        return;
    }
    zo_v4 poly[10]; int n = zo_clip_polygon(c, poly);
    if (n < 3) return;
    zo_sv pv[10];
    for (int i = 0; i < n; ++i) pv[i] = zo_project(poly[i], hw, hh);
    for (int i = 1; i + 1 < n; ++i) { v[0] = pv[0]; v[1] = pv[i]; v[2] = pv[i + 1]; zo_raster_sub(T, v, prim); }
}

/* draws [g0, g1) of the flattened (object in reference order, instance) sequence */
static void zo_raster_range(zo_ctx* c, const zo_target* T, const float* PVM, uint64_t g0, uint64_t g1)
{
    uint64_t base = 0;
    for (int k = 0; k < c->n_objects && base < g1; ++k) {
        const zo_object* o = &c->objects[c->order[k]];
        const zo_mesh* m = &c->meshes[o->mesh];
        uint64_t lo = g0 > base ? g0 - base : 0, hi = g1 - base < o->n_inst ? g1 - base : o->n_inst;
        base += o->n_inst;
        if (lo >= hi) continue;
        uint32_t ntri = m->ni / 3;
        zo_v4* clip = (zo_v4*)malloc(sizeof(zo_v4) * m->nv);
        for (uint32_t i = (uint32_t)lo; i < (uint32_t)hi; ++i) {
            zo_xform x; zo_xform_make(o, i, &x);
            for (uint32_t vi = 0; vi < m->nv; ++vi)
                clip[vi] = zo_mat4_point(PVM, zo_vs_position(&m->v[vi], &x, o->instanced));
            for (uint32_t t = 0; t < ntri; ++t) {
                zo_v4 tc[3] = { clip[m->idx[3 * t]], clip[m->idx[3 * t + 1]], clip[m->idx[3 * t + 2]] };
                zo_raster_tri(T, tc, o->prim_base + i * ntri + t);
            }
        }
        free(clip);
    }
}

/* one pass over every draw in reference order.  With zo_set_threads(n > 1) (the all-cores CPU baseline) thread t rasterises the t-th
 * contiguous slice of the draw sequence into a target of its own and the targets are merged in thread order with the pass's own
 * depth test, which gives the serial result: LESS keeps the earlier draw on equal depth, LESS_OR_EQUAL of depths alone is a min. */
static void zo_raster_scene(zo_ctx* c, const zo_target* T, const float* PVM)
{
    zo_finalize_order(c);
    uint64_t total = 0;
    for (int k = 0; k < c->n_objects; ++k) total += c->objects[k].n_inst;
    int nt = c->threads;
    if ((uint64_t)nt > total) nt = (int)total;
    if (nt <= 1) { zo_raster_range(c, T, PVM, 0, total); return; }
    size_t n = (size_t)T->W * T->H;
    float* depth = (float*)malloc(sizeof(float) * n * (size_t)nt);
    uint32_t* vis = T->vis ? (uint32_t*)malloc(sizeof(uint32_t) * n * (size_t)nt) : NULL;
#pragma omp parallel num_threads(nt)
    {
        int t = omp_get_thread_num(), tn = omp_get_num_threads();
        zo_target P = *T;
        P.depth = depth + n * (size_t)t; memcpy(P.depth, T->depth, sizeof(float) * n);
        if (vis) { P.vis = vis + n * (size_t)t; memcpy(P.vis, T->vis, sizeof(uint32_t) * n); }
        if (t < tn) zo_raster_range(c, &P, PVM, total * (uint64_t)t / (uint64_t)tn, total * (uint64_t)(t + 1) / (uint64_t)tn);
#pragma omp barrier
#pragma omp for schedule(static)
        for (long long p = 0; p < (long long)n; ++p) {
            float z = T->depth[p]; uint32_t v = vis ? T->vis[p] : 0u;
            for (int q = 0; q < tn; ++q) {
                float zq = depth[n * (size_t)q + (size_t)p];
                if (zq < z) { z = zq; if (vis) v = vis[n * (size_t)q + (size_t)p]; }
            }
            T->depth[p] = z; if (vis) T->vis[p] = v;
        }
    }
    free(depth); free(vis);
}

static void zo_pvm(const XkUniformBufferMVP* u, float* PVM)   /* proj * view * model, left to right */
{
    float pv[16]; zo_mat4_mul(u->Proj, u->View, pv); zo_mat4_mul(pv, u->Model, PVM);
}

/* ------------------------------------------------------------------ deferred-scene fragment stage */

typedef struct { zo_v3 P, N; float u, v; } zo_varyings;

/* barycentrics (perspective-correct) of pixel (px,py) for a fast-path triangle */
static void zo_bary_screen(const zo_setup* s, const zo_sv sv[3], int32_t px, int32_t py, float b[3])
{
    float fx = (float)(px * 256 + 128 - s->X0), fy = (float)(py * 256 + 128 - s->Y0);
    float l1 = fmaf(s->b1, fy, s->a1 * fx), l2 = fmaf(s->b2, fy, s->a2 * fx);
    float l0 = (1.0f - l1) - l2;
    float q0 = l0 * sv[0].rw, q1 = l1 * sv[1].rw, q2 = l2 * sv[2].rw;
    float inv = 1.0f / ((q0 + q1) + q2);
    b[0] = q0 * inv; b[1] = q1 * inv; b[2] = q2 * inv;
}
/* same for a clipped triangle: 2D-homogeneous interpolation from the unclipped clip-space vertices */
static void zo_bary_homog(const zo_v4 c[3], float hw, float hh, int32_t px, int32_t py, float b[3])
{
    float u = (((float)px + 0.5f) - hw) / hw, v = (((float)py + 0.5f) - hh) / hh;
    float k[3];
    for (int i = 0; i < 3; ++i) {
        const zo_v4 *p = &c[(i + 1) % 3], *q = &c[(i + 2) % 3];
        float kx = fmaf(p->y, q->w, -(q->y * p->w));
        float ky = fmaf(q->x, p->w, -(p->x * q->w));
        float kz = fmaf(p->x, q->y, -(q->x * p->y));
        k[i] = fmaf(kx, u, fmaf(ky, v, kz));
    }
    float inv = 1.0f / ((k[0] + k[1]) + k[2]);
    b[0] = k[0] * inv; b[1] = k[1] * inv; b[2] = k[2] * inv;
}
static zo_varyings zo_interp(const float b[3], const zo_v3 P[3], const zo_v3 N[3], const float uv[3][2])
{
    zo_varyings r;
    r.P = zo_v3make(fmaf(b[2], P[2].x, fmaf(b[1], P[1].x, b[0] * P[0].x)),
                    fmaf(b[2], P[2].y, fmaf(b[1], P[1].y, b[0] * P[0].y)),
                    fmaf(b[2], P[2].z, fmaf(b[1], P[1].z, b[0] * P[0].z)));
    r.N = zo_v3make(fmaf(b[2], N[2].x, fmaf(b[1], N[1].x, b[0] * N[0].x)),
                    fmaf(b[2], N[2].y, fmaf(b[1], N[1].y, b[0] * N[0].y)),
                    fmaf(b[2], N[2].z, fmaf(b[1], N[1].z, b[0] * N[0].z)));
    r.u = fmaf(b[2], uv[2][0], fmaf(b[1], uv[1][0], b[0] * uv[0][0]));
    r.v = fmaf(b[2], uv[2][1], fmaf(b[1], uv[1][1], b[0] * uv[0][1]));
    return r;
}

/* A texel as the filter sees it: sRGB channels decoded to linear (the format conversion comes before filtering), UNORM channels as
 * their 8-bit CODE.  The filter (bilinear, trilinear, the anisotropic average) is linear, so the codes are filtered and the result is
 * scaled by 1 / 255 ONCE (zo_unorm8_scale, at the end of zo_tex_sample) instead of every texel being divided first: the same real
 * number, rounded once at the end - Vulkan leaves the precision of filtering to the implementation.  csrc states the same. */
static void zo_tex_fetch(const zo_ctx* c, const zo_tex* t, int srgb, int level, int x, int y, float out[4])
{
    uint32_t w = t->w >> level, h = t->h >> level; if (!w) w = 1; if (!h) h = 1;
    const uint8_t* p = t->mip[level] + ((size_t)y * w + (size_t)x) * 4;
#ifdef ZO_LITERAL
    for (int ch = 0; ch < 4; ++ch) out[ch] = zo_decode8(c, p[ch], srgb && ch < 3);      /* every texel converted before the filter */
#else
    for (int ch = 0; ch < 4; ++ch) out[ch] = (srgb && ch < 3) ? c->srgb_lut[p[ch]] : (float)p[ch];
#endif
}
/* x / 255 for a filtered code x: fma(x, k_hi, x * k_lo) with k_hi + k_lo = 1 / 255 to 48 bits (exactly c / 255 rounded for an integer c) */
static float zo_unorm8_scale(float x) { return fmaf(x, 0x1.010102p-8f, x * -0x1.fdfdfep-33f); }
static void zo_tex_finish(int srgb, float out[4])
{
    if (ZO_IS_LITERAL) return;                              /* (the texels were converted one by one) */
    for (int ch = 0; ch < 4; ++ch) if (!(srgb && ch < 3)) out[ch] = zo_unorm8_scale(out[ch]);
}
/* bilinear, REPEAT addressing (RHICreateSampler defaults, ZE:6523-6557) */
static void zo_tex_bilinear(const zo_ctx* c, const zo_tex* t, int srgb, int level, float u, float v, float out[4])
{
    uint32_t w = t->w >> level, h = t->h >> level; if (!w) w = 1; if (!h) h = 1;
    float ur = u - floorf(u), vr = v - floorf(v);
    float x = fmaf(ur, (float)w, -0.5f), y = fmaf(vr, (float)h, -0.5f);
    float fx = floorf(x), fy = floorf(y), a = x - fx, b = y - fy;
    int x0 = zo_idx_clamp(fx + 1.0f, (int)w) - 1, y0 = zo_idx_clamp(fy + 1.0f, (int)h) - 1;     /* -1 .. w-1, NaN -> -1 */
    int x1 = x0 + 1; if (x1 >= (int)w) x1 = 0; if (x0 < 0) x0 = (int)w - 1;
    int y1 = y0 + 1; if (y1 >= (int)h) y1 = 0; if (y0 < 0) y0 = (int)h - 1;
    float t00[4], t10[4], t01[4], t11[4];
    zo_tex_fetch(c, t, srgb, level, x0, y0, t00); zo_tex_fetch(c, t, srgb, level, x1, y0, t10);
    zo_tex_fetch(c, t, srgb, level, x0, y1, t01); zo_tex_fetch(c, t, srgb, level, x1, y1, t11);
    for (int ch = 0; ch < 4; ++ch) {
        float top = fmaf(a, t10[ch] - t00[ch], t00[ch]), bot = fmaf(a, t11[ch] - t01[ch], t01[ch]);
        out[ch] = fmaf(b, bot - top, top);
    }
}
/* texture(sampler2D, uv) in the fragment stage.  The samplers are LINEAR / LINEAR-mip / REPEAT with anisotropyEnable and
 * maxAnisotropy = the device limit (ZE:6523-6557; 16 on the target class of hardware).  Vulkan leaves the anisotropic scheme to
 * the implementation; this restatement takes the one the specification itself describes ("Texel Anisotropic Filtering"):
 *   Pmax, Pmin = longer / shorter of the two screen-axis footprints in texels,  N = min(ceil(Pmax / Pmin), maxAniso),
 *   lambda = log2(Pmax / N),  result = 1/N * sum_{i=1..N} trilinear(uv + (i / (N+1) - 1/2) * duv/d(major axis)).
 * N = 1 (isotropic footprint) is plain trilinear filtering with lambda = log2(Pmax). */
#define ZO_MAX_ANISO 16
static void zo_tex_trilinear(const zo_ctx* c, const zo_tex* t, int srgb, float lambda, float u, float v, float out[4])
{
    float fl = floorf(lambda);
    int l0 = (int)fl, l1 = l0 + 1 < t->levels ? l0 + 1 : t->levels - 1;
    float f = lambda - fl;
    float c0[4], c1[4];
    zo_tex_bilinear(c, t, srgb, l0, u, v, c0); zo_tex_bilinear(c, t, srgb, l1, u, v, c1);
    for (int ch = 0; ch < 4; ++ch) out[ch] = fmaf(f, c1[ch] - c0[ch], c0[ch]);
}
/* tap count, level of detail and major axis for the footprint (ax, ay) / (bx, by) in texels of level 0 */
static void zo_aniso_setup(float ax, float ay, float bx, float by, int levels, int* Nout, float* lambda_out, int* xmajor_out)
{
    float rx2 = fmaf(ax, ax, ay * ay), ry2 = fmaf(bx, bx, by * by);
    float rmax2 = fmaxf(rx2, ry2), rmin2 = fminf(rx2, ry2);
    int N = 1;                                            /* least N with N^2 * Pmin^2 >= Pmax^2, at most maxAniso */
    while (N < ZO_MAX_ANISO && (float)(N * N) * rmin2 < rmax2) ++N;
    float lambda = 0.5f * zo_log2f(rmax2);
    if (N > 1) lambda = lambda - zo_log2f((float)N);
    *lambda_out = fminf(fmaxf(lambda, 0.0f), (float)(levels - 1));
    *Nout = N; *xmajor_out = rx2 >= ry2;
}
void zo_kat_aniso(float ax, float ay, float bx, float by, int levels, float out[3])
{
    int N, xm; float l;
    zo_aniso_setup(ax, ay, bx, by, levels, &N, &l, &xm);
    out[0] = (float)N; out[1] = l; out[2] = (float)xm;
}
static void zo_tex_sample(const zo_ctx* c, const zo_tex* t, int srgb, float u, float v, float dudx, float dvdx, float dudy, float dvdy, float out[4])
{
    if (t->constant) { zo_tex_fetch(c, t, srgb, 0, 0, 0, out); zo_tex_finish(srgb, out); return; }
    float W = (float)t->w, H = (float)t->h;
    int N, xmajor; float lambda;
    zo_aniso_setup(dudx * W, dvdx * H, dudy * W, dvdy * H, t->levels, &N, &lambda, &xmajor);
    if (N == 1) { zo_tex_trilinear(c, t, srgb, lambda, u, v, out); zo_tex_finish(srgb, out); return; }
    float du = xmajor ? dudx : dudy, dv = xmajor ? dvdx : dvdy;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 1; i <= N; ++i) {
        float off = (float)i / (float)(N + 1) - 0.5f;
        float s[4];
        zo_tex_trilinear(c, t, srgb, lambda, fmaf(du, off, u), fmaf(dv, off, v), s);
        for (int ch = 0; ch < 4; ++ch) acc[ch] += s[ch];
    }
    for (int ch = 0; ch < 4; ++ch) out[ch] = acc[ch] / (float)N;
    zo_tex_finish(srgb, out);
}

/* ComputeNormal(fragPosition, fragTexCoord, fragNormal, texNormal), SH/Common.glsl:113-127 */
static zo_v3 zo_compute_normal(zo_v3 pos_dx, zo_v3 pos_dy, float s1, float t1, float s2, float t2, zo_v3 fragN, zo_v3 texN)
{
    float det = fmaf(s1, t2, -(s2 * t1));
    zo_v3 T = zo_div_scalar(zo_v3make(fmaf(t2, pos_dx.x, -(t1 * pos_dy.x)),     /* vec3 / scalar */
                                      fmaf(t2, pos_dx.y, -(t1 * pos_dy.y)),
                                      fmaf(t2, pos_dx.z, -(t1 * pos_dy.z))), det);
    zo_v3 N = zo_normalize(fragN);
    T = zo_normalize(zo_sub(T, zo_scale(N, zo_dot(N, T))));
    zo_v3 B = zo_normalize(zo_cross(N, T));
    zo_v3 n = zo_normalize(texN);
    zo_v3 ts = zo_normalize(zo_v3make(fmaf(2.0f, n.x, -1.0f), fmaf(2.0f, n.y, -1.0f), fmaf(2.0f, n.z, -1.0f)));
    zo_v3 w = zo_v3make(fmaf(N.x, ts.z, fmaf(B.x, ts.y, T.x * ts.x)),
                        fmaf(N.y, ts.z, fmaf(B.y, ts.y, T.y * ts.x)),
                        fmaf(N.z, ts.z, fmaf(B.z, ts.y, T.z * ts.x)));
    return zo_normalize(w);
}

static void zo_clear_gbuffer(zo_ctx* c)
{   /* clears ZE:3427-3433 */
    size_t n = (size_t)c->W * c->H;
    for (size_t i = 0; i < n; ++i) {
        c->depth[i] = 1.0f; c->vis[i] = ZO_EMPTY;
        c->scene_color[i] = 0xFF000000u; c->gA[i] = 0u; c->gB[i] = 0xFF000000u; c->gC[i] = 0xFF000000u;
        c->gD[i] = (uint64_t)0x3C00u << 48;
    }
}

static const zo_object* zo_find_object(const zo_ctx* c, uint32_t prim, uint32_t* inst, uint32_t* tri)
{
    for (int k = c->n_objects - 1; k >= 0; --k) {
        const zo_object* o = &c->objects[c->order[k]];
        if (prim >= o->prim_base) {
            uint32_t ntri = c->meshes[o->mesh].ni / 3, local = prim - o->prim_base;
            *inst = local / ntri; *tri = local % ntri;
            return o;
        }
    }
    return NULL;
}

/* BaseScene.frag:26-48 for every covered pixel */
static void zo_resolve_gbuffer(zo_ctx* c, const float* PVM)
{
    float hw = 0.5f * (float)c->W, hh = 0.5f * (float)c->H;
    uint64_t covered = 0;
    /* rows are independent: optional OpenMP team for the all-cores CPU baseline (zo_set_threads; default 1 thread) */
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : covered) num_threads(c->threads)
    for (uint32_t py = 0; py < c->H; ++py) for (uint32_t px = 0; px < c->W; ++px) {
        size_t p = (size_t)py * c->W + px;
        uint32_t prim = c->vis[p];
        if (prim == ZO_EMPTY) continue;
        covered++;
        uint32_t inst, tri;
        const zo_object* o = zo_find_object(c, prim, &inst, &tri);
        const zo_mesh* m = &c->meshes[o->mesh];
        zo_xform x; zo_xform_make(o, inst, &x);
        zo_v4 clip[3]; zo_v3 WP[3], WN[3]; float uv[3][2];
        for (int k = 0; k < 3; ++k) {
            const XkVertex* v = &m->v[m->idx[3 * tri + (uint32_t)k]];
            zo_v3 pos = zo_vs_position(v, &x, o->instanced);
            clip[k] = zo_mat4_point(PVM, pos);
            zo_v4 wp = zo_mat4_point(c->cam.Model, pos);
            WP[k] = zo_v3make(wp.x, wp.y, wp.z);
            WN[k] = zo_vs_normal(v, &x, o->instanced, c->cam.Model);
            uv[k][0] = v->TexCoord[0]; uv[k][1] = v->TexCoord[1];
        }
        int cls = zo_classify(clip);
        zo_sv sv[3]; zo_setup s;
        if (cls == 1) { for (int k = 0; k < 3; ++k) sv[k] = zo_project(clip[k], hw, hh); zo_tri_setup(sv, 0, &s); }
        /* 2x2 quad: own pixel, horizontal partner, vertical partner */
        int32_t qx = (int32_t)(px ^ 1u), qy = (int32_t)(py ^ 1u);
        float b0[3], bh[3], bv[3];
        if (cls == 1) {
            zo_bary_screen(&s, sv, (int32_t)px, (int32_t)py, b0);
            zo_bary_screen(&s, sv, qx, (int32_t)py, bh);
            zo_bary_screen(&s, sv, (int32_t)px, qy, bv);
        } else {
            zo_bary_homog(clip, hw, hh, (int32_t)px, (int32_t)py, b0);
            zo_bary_homog(clip, hw, hh, qx, (int32_t)py, bh);
            zo_bary_homog(clip, hw, hh, (int32_t)px, qy, bv);
        }
        zo_varyings f0 = zo_interp(b0, WP, WN, uv), fh = zo_interp(bh, WP, WN, uv), fv = zo_interp(bv, WP, WN, uv);
        /* dFdx = right - left, dFdy = lower - upper (fine derivatives inside the quad) */
        float sx = (px & 1u) ? 1.0f : -1.0f, sy = (py & 1u) ? 1.0f : -1.0f;
        zo_v3 pos_dx = zo_scale(zo_sub(f0.P, fh.P), sx), pos_dy = zo_scale(zo_sub(f0.P, fv.P), sy);
        float s1 = (f0.u - fh.u) * sx, t1 = (f0.v - fh.v) * sx, s2 = (f0.u - fv.u) * sy, t2 = (f0.v - fv.v) * sy;

        float bc[4], me[4], ro[4], nm[4], ao[4], em[4], ms[4];
        /* texture(samplerN, fragTexCoord), BaseScene.frag:30-36; s1,t1 = dFdx(uv), s2,t2 = dFdy(uv) */
        zo_tex_sample(c, &o->tex[0], 1, f0.u, f0.v, s1, t1, s2, t2, bc);         /* base colour is R8G8B8A8_SRGB, ZE:5878 */
        zo_tex_sample(c, &o->tex[1], 0, f0.u, f0.v, s1, t1, s2, t2, me); zo_tex_sample(c, &o->tex[2], 0, f0.u, f0.v, s1, t1, s2, t2, ro);
        zo_tex_sample(c, &o->tex[3], 0, f0.u, f0.v, s1, t1, s2, t2, nm); zo_tex_sample(c, &o->tex[4], 0, f0.u, f0.v, s1, t1, s2, t2, ao);
        zo_tex_sample(c, &o->tex[5], 0, f0.u, f0.v, s1, t1, s2, t2, em); zo_tex_sample(c, &o->tex[6], 0, f0.u, f0.v, s1, t1, s2, t2, ms);

        zo_v3 Nw = zo_compute_normal(pos_dx, pos_dy, s1, t1, s2, t2, f0.N, zo_v3make(nm[0], nm[1], nm[2]));
        float Rough = fmaxf(0.01f, ro[0]);
        if (c->forward) {      /* Base.frag:48-60: the same fetches and ComputeNormal, kept as floats; fragColor interpolated like the rest */
            zo_fsurf* F = &c->fwd[p];
            const XkVertex* v0 = &m->v[m->idx[3 * tri]]; const XkVertex* v1 = &m->v[m->idx[3 * tri + 1u]]; const XkVertex* v2 = &m->v[m->idx[3 * tri + 2u]];
            F->BaseColor = zo_v3make(bc[0], bc[1], bc[2]);
            F->Metallic = zo_saturate(me[0]);
            F->Roughness = fmaxf(0.01f, zo_saturate(ro[0]));
            F->Normal = Nw;
            F->AmbientOcclution = zo_v3make(ao[0], ao[1], ao[2]);
            F->P = f0.P;
            F->VertexColor = zo_v3make(fmaf(b0[2], v2->Color[0], fmaf(b0[1], v1->Color[0], b0[0] * v0->Color[0])),
                                       fmaf(b0[2], v2->Color[1], fmaf(b0[1], v1->Color[1], b0[0] * v0->Color[1])),
                                       fmaf(b0[2], v2->Color[2], fmaf(b0[1], v1->Color[2], b0[0] * v0->Color[2])));
        }
        zo_v3 Nn = zo_normalize(Nw);
        zo_v3 NP = zo_v3make((Nn.x + 1.0f) / 2.0f, (Nn.y + 1.0f) / 2.0f, (Nn.z + 1.0f) / 2.0f);
        c->scene_color[p] = zo_unorm(em[0], 255.0f) | zo_unorm(em[1], 255.0f) << 8 | zo_unorm(em[2], 255.0f) << 16 | zo_unorm(ms[0], 255.0f) << 24;
        c->gA[p] = zo_unorm(NP.z, 1023.0f) | zo_unorm(NP.y, 1023.0f) << 10 | zo_unorm(NP.x, 1023.0f) << 20 | 3u << 30;
        c->gB[p] = zo_unorm(me[0], 255.0f) | zo_unorm(1.0f, 255.0f) << 8 | zo_unorm(Rough, 255.0f) << 16 | 255u << 24;
        c->gC[p] = zo_unorm(bc[0], 255.0f) | zo_unorm(bc[1], 255.0f) << 8 | zo_unorm(bc[2], 255.0f) << 16 | zo_unorm(ao[0], 255.0f) << 24;
        c->gD[p] = (uint64_t)zo_f32_to_f16(f0.P.x) | (uint64_t)zo_f32_to_f16(f0.P.y) << 16 |
                   (uint64_t)zo_f32_to_f16(f0.P.z) << 32 | (uint64_t)0x3C00u << 48;
    }
    c->covered = covered;
}

/* ------------------------------------------------------------------ deferred lighting (SH/BaseLighting.frag:147-254) */

static float zo_F_Schlick(float f0, float f90, float u) { return fmaf(f90 - f0, zo_pow5f(1.0f - u), f0); }  /* :134 */
static float zo_Fr_DisneyDiffuse(float NdotV, float NdotL, float LdotH, float r)                            /* :148 */
{
    float E_bias = fmaf(0.5f, r, 0.0f * (1.0f - r));
    float E_factor = fmaf(1.0f / 1.51f, r, 1.0f * (1.0f - r));
    float fd90 = fmaf((2.0f * LdotH) * LdotH, r, E_bias);
    float ls = zo_F_Schlick(1.0f, fd90, NdotL), vs = zo_F_Schlick(1.0f, fd90, NdotV);
    return (ls * vs) * E_factor;
}
static float zo_V_SmithGGXCorrelated(float NdotV, float NdotL, float r)                                      /* :161 */
{
    float a2 = r * r;
    float GGXV = NdotL * sqrtf(fmaf(NdotV * NdotV, 1.0f - a2, a2));
    float GGXL = NdotV * sqrtf(fmaf(NdotL * NdotL, 1.0f - a2, a2));
    float GGX = GGXV + GGXL;
    return GGX > 0.0f ? 0.5f / GGX : 0.0f;
}
static float zo_D_GGX(float NdotH, float r)                                                                  /* :178 */
{
    float a2 = r * r;
    float f = fmaf(fmaf(NdotH, a2, -NdotH), NdotH, 1.0f);
    return a2 / ((3.14159265359f * f) * f);
}
static float zo_ReflectionMip(float r, float maxmip)                                                         /* :191 */
{
    float level_from_1x1 = fmaf(-1.2f, zo_log2f(fmaxf(r, 0.001f)), 1.0f);
    return (maxmip - 1.0f) - level_from_1x1;
}
static void zo_EnvBRDFApproxLazarov(float r, float NoV, float AB[2])                                         /* :201 */
{
    float rx = fmaf(r, -1.0f, 1.0f), ry = fmaf(r, -0.0275f, 0.0425f), rz = fmaf(r, -0.572f, 1.04f), rw = fmaf(r, 0.022f, -0.04f);
    float a004 = fmaf(fminf(rx * rx, zo_exp2f(-9.28f * NoV)), rx, ry);
    AB[0] = fmaf(-1.04f, a004, rz); AB[1] = fmaf(1.04f, a004, rw);
}

/* ShadowDepthProject + texture(LINEAR, clamp-to-edge) on the D32 shadow map, :307-319, sampler ZE:2532-2537 */
static float zo_shadow_tap(const zo_ctx* c, float sx, float sy, float sz, float sw, float ox, float oy)
{
    float f = 1.0f;
    if (sz > -1.0f && sz < 1.0f) {
        float dim = (float)c->SD;
        float u = fmaf(sx + ox, dim, -0.5f), v = fmaf(sy + oy, dim, -0.5f);
        float fu = floorf(u), fv = floorf(v), a = u - fu, b = v - fv;
        int x0 = zo_idx_clamp(fu, (int)c->SD - 1), x1 = zo_idx_clamp(fu + 1.0f, (int)c->SD - 1);
        int y0 = zo_idx_clamp(fv, (int)c->SD - 1), y1 = zo_idx_clamp(fv + 1.0f, (int)c->SD - 1);
        const float* S = c->shadowmap;
        float t00 = S[(size_t)y0 * c->SD + x0], t10 = S[(size_t)y0 * c->SD + x1];
        float t01 = S[(size_t)y1 * c->SD + x0], t11 = S[(size_t)y1 * c->SD + x1];
        float top = fmaf(a, t10 - t00, t00), bot = fmaf(a, t11 - t01, t01);
        float dist = fmaf(b, bot - top, top);
        if (sw > 0.0f && dist < sz) f = 0.1f;
    }
    return f;
}

/* ------------------------------------------------------------------ GBufferVis, BaseLighting.frag:42-145 (SPEC_CONSTANTS 9)
 *
 * The lighting quad samples every GBuffer target again at UV = fragTexCoord * 3 / (1 - EmptyRatio) through the LINEAR / REPEAT
 * samplers of ZE:2811-2847 and shows a 3 x 3 mosaic; the centre cell and everything outside the cells keep FinalColor.
 * Choice made here (Vulkan leaves the weight precision to the implementation): the bilinear weights are snapped to 8 fractional
 * bits, as sampling hardware does; with EmptyRatio = 0 every sample then lands exactly on texel (3x + 1, 3y + 1) mod (W, H).
 */
typedef struct { float sc[4], a[4], b[4], c[4], d[4]; } zo_gtexel;

static void zo_gbuffer_texel(const zo_ctx* c, int x, int y, zo_gtexel* t)
{
    size_t p = (size_t)y * c->W + (size_t)x;
    uint32_t sc = c->scene_color[p], A = c->gA[p], B = c->gB[p], C = c->gC[p]; uint64_t D = c->gD[p];
    for (int k = 0; k < 4; ++k) {
        t->sc[k] = (float)((sc >> (8 * k)) & 255u) / 255.0f;
        t->b[k] = (float)((B >> (8 * k)) & 255u) / 255.0f;
        t->c[k] = (float)((C >> (8 * k)) & 255u) / 255.0f;
        t->d[k] = zo_f16_to_f32((uint16_t)(D >> (16 * k)));
    }
    /* A2R10G10B10_UNORM_PACK32: r = bits 20-29, g = 10-19, b = 0-9, a = 30-31 */
    t->a[0] = (float)((A >> 20) & 1023u) / 1023.0f; t->a[1] = (float)((A >> 10) & 1023u) / 1023.0f;
    t->a[2] = (float)(A & 1023u) / 1023.0f; t->a[3] = (float)(A >> 30) / 3.0f;
}

static int zo_wrap(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

static void zo_gbuffer_sample(const zo_ctx* c, float u, float v, zo_gtexel* out)
{
    float x = fmaf(u, (float)c->W, -0.5f), y = fmaf(v, (float)c->H, -0.5f);
    if (!(fabsf(x) < 1.0e9f)) x = 0.0f;
    if (!(fabsf(y) < 1.0e9f)) y = 0.0f;
    float fx = floorf(x), fy = floorf(y);
    float ax = floorf(fmaf(x - fx, 256.0f, 0.5f)) / 256.0f, ay = floorf(fmaf(y - fy, 256.0f, 0.5f)) / 256.0f;
    int x0 = zo_wrap((int)fx, (int)c->W), x1 = zo_wrap((int)fx + 1, (int)c->W);
    int y0 = zo_wrap((int)fy, (int)c->H), y1 = zo_wrap((int)fy + 1, (int)c->H);
    zo_gtexel t00, t10, t01, t11;
    zo_gbuffer_texel(c, x0, y0, &t00); zo_gbuffer_texel(c, x1, y0, &t10);
    zo_gbuffer_texel(c, x0, y1, &t01); zo_gbuffer_texel(c, x1, y1, &t11);
    const float* s00 = t00.sc; const float* s10 = t10.sc; const float* s01 = t01.sc; const float* s11 = t11.sc;
    float* o = out->sc;
    for (int k = 0; k < 20; ++k) {          /* the five vec4s are contiguous */
        float top = fmaf(ax, s10[k] - s00[k], s00[k]), bot = fmaf(ax, s11[k] - s01[k], s01[k]);
        o[k] = fmaf(ay, bot - top, top);
    }
}

static float zo_pcf(const zo_ctx* c, const float* SB, zo_v3 P, float dxy)
{
    zo_v4 s4 = zo_mat4_point(SB, P);
    /* shadowCoord / shadowCoord.w (SH/Common.glsl:296): a correctly rounded division per component in BOTH builds since round 6 - as
       reciprocal-multiply it moved z by an ulp and with it the PCF comparison's ties (oracle/CONTRACT.md, the struck part of row 3) */
    float sx = s4.x / s4.w, sy = s4.y / s4.w, sz = s4.z / s4.w, sw = s4.w / s4.w;
    float sum = 0.0f;
    for (int x = -2; x <= 2; ++x) for (int y = -2; y <= 2; ++y)               /* ComputePCF r=2, :323-342 */
        sum += zo_shadow_tap(c, sx, sy, sz, sw, dxy * (float)x, dxy * (float)y);
    return zo_div_25(sum);                                  /* ShadowFactor / Count */
}

static zo_v3 zo_gbuffer_vis(const zo_ctx* c, uint32_t px, uint32_t py, zo_v3 FinalColor, const float* SB, zo_v3 cam, float dxy)
{
    const XkView* V = &c->view;
    float ERx = V->ViewportInfo[2] / V->ViewportInfo[0], ERy = V->ViewportInfo[3] / V->ViewportInfo[1];
    float tx = ((float)px + 0.5f) / (float)c->W, ty = ((float)py + 0.5f) / (float)c->H;       /* fragTexCoord */
    float UVx = (tx * 3.0f) / (1.0f - ERx), UVy = (ty * 3.0f) / (1.0f - ERy);
    float Sx = (1.0f - ERx) / 3.0f, Sy = (1.0f - ERy) / 3.0f;                                   /* Step */
    int cell = -1; float bx = 0.0f, by = 0.0f;      /* which branch of :74-143, and its white-border thresholds */
    if (tx < Sx && ty < Sy) { cell = 0; bx = 1.0f; by = 1.0f; }
    else if (tx < Sx * 2.0f && ty < Sy) { cell = 1; bx = 2.0f; by = 1.0f; }
    else if (tx < Sx * 3.0f && ty < Sy) { cell = 2; bx = 3.0f; by = 1.0f; }
    else if (tx < Sx && ty < Sy * 2.0f) { cell = 3; bx = 1.0f; by = 2.0f; }
    else if (tx < 1.0f && ty < Sy * 2.0f && tx > Sx * 2.0f) { cell = 4; bx = 3.0f; by = 2.0f; }
    else if (tx < Sx && ty < Sx * 3.0f) { cell = 5; bx = 1.0f; by = 3.0f; }                     /* Step.x * 3 is the shader's own */
    else if (tx < Sx * 2.0f && tx > Sx && ty < Sy * 3.0f && ty > Sy * 2.0f) { cell = 6; bx = 2.0f; by = 3.0f; }
    else if (tx < Sx * 3.0f && tx > Sx * 2.0f && ty < Sy * 3.0f && ty > Sy * 2.0f) { cell = 7; bx = 3.0f; by = 3.0f; }
    if (cell < 0) return FinalColor;
    if (tx > Sx * (bx - ERx) || ty > Sy * (by - ERy)) return zo_v3make(1.0f, 1.0f, 1.0f);
    zo_gtexel g; zo_gbuffer_sample(c, UVx, UVy, &g);
    zo_v3 BaseColor = zo_v3make(g.c[0], g.c[1], g.c[2]);
    float Metallic = zo_saturate(g.b[0]);
    float Roughness = fmaxf(0.01f, zo_saturate(g.b[2]));
    zo_v3 Normal = zo_v3make(fmaf(g.a[0], 2.0f, -1.0f), fmaf(g.a[1], 2.0f, -1.0f), fmaf(g.a[2], 2.0f, -1.0f));
    float AO = zo_saturate(g.c[3]);
    zo_v3 N = zo_normalize(Normal);
    zo_v3 P = zo_v3make(g.d[0], g.d[1], g.d[2]);
    switch (cell) {
    case 0: return zo_v3make(zo_powf(BaseColor.x, 0.4545f), zo_powf(BaseColor.y, 0.4545f), zo_powf(BaseColor.z, 0.4545f));
    case 1: return zo_v3make(Metallic, Metallic, Metallic);
    case 2: return zo_v3make(Roughness, Roughness, Roughness);
    case 3: return N;
    case 4: return zo_v3make(AO, AO, AO);
    case 5: return zo_v3make(0.0f, 0.0f, 0.0f);
    case 6: {
        zo_v3 Vv = zo_normalize(zo_sub(cam, P));
        zo_v3 Nn = zo_normalize(N);
        float eta = 1.00f / 1.52f;
        float dNI = zo_dot(Nn, Vv);
        float kk = fmaf(-(eta * eta), fmaf(-dNI, dNI, 1.0f), 1.0f);
        zo_v3 R;
        if (kk < 0.0f) R = zo_v3make(0, 0, 0);
        else { float q = fmaf(eta, dNI, sqrtf(kk)); R = zo_v3make(fmaf(eta, Vv.x, -(q * Nn.x)), fmaf(eta, Vv.y, -(q * Nn.y)), fmaf(eta, Vv.z, -(q * Nn.z))); }
        return zo_scale(zo_cube_sample(c, R, 0.0f), 10.0f);
    }
    default: { float sf = zo_pcf(c, SB, P, dxy); return zo_v3make(sf, sf, sf); }
    }
}

/* What BaseLighting.frag:174-221 and Base.frag:62-112 have in common, word for word: the PCF factor, (1) direct lighting over the
 * directional then the point lights, (2) the lambert indirect term, (3) the image-based reflection.  Inputs as the shader holds them at
 * that point (N: `normalize(Normal)` of the unpacked GBufferA in the deferred shader, ComputeNormal()'s result in the forward one). */
typedef struct { zo_v3 Direct, Indirect, RefC; float ShadowFactor; } zo_lit;
static zo_lit zo_shade(const zo_ctx* c, const float* SB, zo_v3 cam, float dxy, zo_v3 BaseColor, float Metallic, float Roughness,
                       zo_v3 N, float AO, zo_v3 P)
{
    const XkView* V = &c->view;
    uint32_t nDir = (uint32_t)V->LightsCount[0], nPoint = (uint32_t)V->LightsCount[1];
    float maxmips = (float)(uint32_t)V->LightsCount[3];
    zo_v3 Vv = zo_normalize(zo_sub(cam, P));
    float NdotV = zo_saturate(zo_dot(N, Vv));

    float ShadowFactor = zo_pcf(c, SB, P, dxy);

    zo_v3 Direct = zo_v3make(0, 0, 0);
    zo_v3 Nn = zo_normalize(N);                  /* Apply*Light and refract() re-normalise N */
    zo_v3 DiffuseColor = zo_scale(BaseColor, 1.0f - Metallic);
    for (uint32_t i = 0; i < nDir + nPoint; ++i) {
        int isdir = i < nDir;
        const XkLight* Lt = isdir ? &V->DirectionalLights[i] : &V->PointLights[i - nDir];
        zo_v3 lp = zo_v3make(Lt->Position[0], Lt->Position[1], Lt->Position[2]);
        /* point light: distance() and normalize() of light_pos - position share one inversesqrt: length = d2 * inversesqrt(d2) */
        zo_v3 dl = zo_sub(lp, P);
        float d2 = zo_dot(dl, dl), rd = zo_shader_rsqrt(d2);
        zo_v3 L = isdir ? zo_normalize(zo_v3make(Lt->Direction[0], Lt->Direction[1], Lt->Direction[2])) : zo_scale(dl, rd);
        zo_v3 Hh = zo_normalize(zo_add(Vv, L));
        float LdotH = zo_saturate(zo_dot(L, Hh)), NdotH = zo_saturate(zo_dot(N, Hh)), NdotL = zo_saturate(zo_dot(N, L));
        /* DefaultLitBxDF, Common.glsl:259-282: F0 = 0.04, F90 = saturate(50*0.04) = 1 */
        float F = zo_F_Schlick(0.04f, zo_saturate(50.0f * 0.04f), LdotH);
        float Vis = zo_V_SmithGGXCorrelated(NdotV, NdotL, Roughness);
        float Dg = zo_D_GGX(NdotH, Roughness);
        float Fr = (F * Dg) * Vis;
        float Fd = zo_Fr_DisneyDiffuse(NdotV, NdotL, LdotH, Roughness);
        zo_v3 bx = zo_v3make(fmaf(DiffuseColor.x * (1.0f - F), Fd, Fr), fmaf(DiffuseColor.y * (1.0f - F), Fd, Fr),
                             fmaf(DiffuseColor.z * (1.0f - F), Fd, Fr));
        /* Apply*Light, Common.glsl:364-372 / 399-416 */
        float ndotl = zo_clampf(zo_dot(Nn, L), 0.0f, 1.0f);
        float k = ndotl * Lt->Color[3];
        zo_v3 rad = zo_v3make(k * Lt->Color[0], k * Lt->Color[1], k * Lt->Color[2]);
        if (isdir) {
            Direct = zo_v3make(fmaf(rad.x * bx.x, ShadowFactor, Direct.x), fmaf(rad.y * bx.y, ShadowFactor, Direct.y),
                               fmaf(rad.z * bx.z, ShadowFactor, Direct.z));
        } else {
            float dist = ZO_IS_LITERAL ? sqrtf(d2) : (d2 > 0.0f ? d2 * rd : 0.0f);      /* distance(): literal = its own sqrt */
            float falloff = Lt->Direction[3];
            float att = 1.0f - zo_clampf(dist, 0.0f, falloff) / falloff;   /* remap(dist,0,falloff,0,1), :43-47 */
            rad = zo_scale(rad, att);
            Direct = zo_v3make(fmaf(rad.x, bx.x, Direct.x), fmaf(rad.y, bx.y, Direct.y), fmaf(rad.z, bx.z, Direct.z));
        }
    }
    /* (2) indirect, :210 */
    zo_v3 Indirect = zo_v3make(((zo_div_pi(DiffuseColor.x) * AO) * 0.3f) * ShadowFactor,
                               ((zo_div_pi(DiffuseColor.y) * AO) * 0.3f) * ShadowFactor,
                               ((zo_div_pi(DiffuseColor.z) * AO) * 0.3f) * ShadowFactor);
    /* (3) reflection, :213-221 */
    zo_v3 bcl = zo_v3make(zo_clampf(BaseColor.x, 0.04f, 1.0f), zo_clampf(BaseColor.y, 0.04f, 1.0f), zo_clampf(BaseColor.z, 0.04f, 1.0f));
    float dsf0 = (0.04f * 2.0f) * 0.5f;
    zo_v3 RSpec = zo_v3make(fmaf(Metallic, bcl.x, (1.0f - Metallic) * dsf0), fmaf(Metallic, bcl.y, (1.0f - Metallic) * dsf0),
                            fmaf(Metallic, bcl.z, (1.0f - Metallic) * dsf0));
    float AB[2]; zo_EnvBRDFApproxLazarov(Roughness, NdotV, AB);
    float F90 = zo_saturate(50.0f * RSpec.y);
    zo_v3 RBRDF = zo_v3make(fmaf(RSpec.x, AB[0], F90 * AB[1]), fmaf(RSpec.y, AB[0], F90 * AB[1]), fmaf(RSpec.z, AB[0], F90 * AB[1]));
    float eta = 1.00f / 1.52f;
    float dNI = zo_dot(Nn, Vv);
    float kk = fmaf(-(eta * eta), fmaf(-dNI, dNI, 1.0f), 1.0f);
    zo_v3 R;
    if (kk < 0.0f) R = zo_v3make(0, 0, 0);
    else { float q = fmaf(eta, dNI, sqrtf(kk)); R = zo_v3make(fmaf(eta, Vv.x, -(q * Nn.x)), fmaf(eta, Vv.y, -(q * Nn.y)), fmaf(eta, Vv.z, -(q * Nn.z))); }
    float MIPS = zo_ReflectionMip(Roughness, maxmips);
    zo_v3 RL = zo_scale(zo_cube_sample(c, R, MIPS), 10.0f);
    float RV = zo_saturate((zo_powf(NdotV + AO, Roughness * Roughness) - 1.0f) + AO);
    zo_v3 RefC = zo_v3make((RL.x * RV) * RBRDF.x, (RL.y * RV) * RBRDF.y, (RL.z * RV) * RBRDF.z);
    zo_lit r = { Direct, Indirect, RefC, ShadowFactor };
    return r;
}

static void zo_lighting(zo_ctx* c, uint32_t debug_view)
{
    const XkView* V = &c->view;
    /* BiasMat * shadowmapSpace (Common.glsl:294-304), folded on the host like a driver folds uniform*const */
    static const float Bias[16] = {0.5f,0,0,0, 0,0.5f,0,0, 0,0,1,0, 0.5f,0.5f,0,1};
    float SB[16]; zo_mat4_mul(Bias, V->ShadowmapSpace, SB);
    zo_v3 cam = zo_v3make(V->CameraInfo[0], V->CameraInfo[1], V->CameraInfo[2]);
    float dxy = 1.5f * 1.0f / (float)c->SD;
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->threads)
    for (uint32_t py = 0; py < c->H; ++py) for (uint32_t px = 0; px < c->W; ++px) {
        size_t p = (size_t)py * c->W + px;
        uint32_t sc = c->scene_color[p], A = c->gA[p], B = c->gB[p], C = c->gC[p]; uint64_t D = c->gD[p];
        zo_v3 BaseColor = zo_v3make((float)(C & 255u) / 255.0f, (float)((C >> 8) & 255u) / 255.0f, (float)((C >> 16) & 255u) / 255.0f);
        float Metallic = zo_saturate((float)(B & 255u) / 255.0f);
        float Roughness = zo_saturate((float)((B >> 16) & 255u) / 255.0f);
        zo_v3 Normal = zo_v3make(fmaf((float)((A >> 20) & 1023u) / 1023.0f, 2.0f, -1.0f),
                                 fmaf((float)((A >> 10) & 1023u) / 1023.0f, 2.0f, -1.0f),
                                 fmaf((float)(A & 1023u) / 1023.0f, 2.0f, -1.0f));
        float AOc = (float)(C >> 24) / 255.0f;
        float Mask = (float)(sc >> 24) / 255.0f;
        Roughness = fmaxf(0.01f, Roughness);
        float AO = zo_saturate(AOc);
        zo_v3 N = zo_normalize(Normal);
        zo_v3 P = zo_v3make(zo_f16_to_f32((uint16_t)D), zo_f16_to_f32((uint16_t)(D >> 16)), zo_f16_to_f32((uint16_t)(D >> 32)));
        zo_lit lit = zo_shade(c, SB, cam, dxy, BaseColor, Metallic, Roughness, N, AO, P);
        zo_v3 Direct = lit.Direct, Indirect = lit.Indirect, RefC = lit.RefC; float ShadowFactor = lit.ShadowFactor;

        zo_v3 Final = zo_add(zo_add(Direct, Indirect), RefC);
        Final = zo_scale(Final, Mask);
        Final = zo_v3make(zo_powf(Final.x, 0.4545f), zo_powf(Final.y, 0.4545f), zo_powf(Final.z, 0.4545f));
        zo_v3 out;
        switch (debug_view) {
        case 0: out = Final; break;
        case 1: out = zo_v3make(zo_powf(BaseColor.x, 0.4545f), zo_powf(BaseColor.y, 0.4545f), zo_powf(BaseColor.z, 0.4545f)); break;
        case 2: out = zo_v3make(Metallic, Metallic, Metallic); break;
        case 3: out = zo_v3make(Roughness, Roughness, Roughness); break;
        case 4: out = Normal; break;
        case 5: out = zo_v3make(AO, AO, AO); break;
        case 7: out = RefC; break;
        case 8: out = zo_v3make(ShadowFactor, ShadowFactor, ShadowFactor); break;
        case 6: {   /* fragColor of the full-screen quad: vertex colours of Background.vert:10-17 interpolated over its two triangles */
            float u = ((float)px + 0.5f) / (float)c->W, v = ((float)py + 0.5f) / (float)c->H;
            out = v >= u ? zo_v3make(1.0f - v, u, v - u) : zo_v3make(1.0f - u, v, u - v);
            break;
        }
        case 9: out = zo_gbuffer_vis(c, px, py, Final, SB, cam, dxy); break;
        default: out = zo_scale(Final, ShadowFactor); break;
        }
        uint8_t* o = c->color + p * 4;
        o[0] = (uint8_t)zo_unorm(out.x, 255.0f); o[1] = (uint8_t)zo_unorm(out.y, 255.0f);
        o[2] = (uint8_t)zo_unorm(out.z, 255.0f); o[3] = 255;
        if (debug_view == 0 && c->overlay[p]) memcpy(o, &c->overlay[p], 4);   /* skydome / background drawn over the lit quad */
    }
}

/* ------------------------------------------------------------------ forward variant (SH/Base.frag:46-144) */
/* The engine built with ENABLE_DEFERRED_SHADING false (ZE:93): the main render pass clears colour (0,0,0,1) and depth 1.0 (ZE:3517-3519,
 * 2366-2373) and Base.frag shades every fragment that passes LESS straight into the swapchain image (pipelines ZE:2749-2801, draws
 * ZE:3544-3680).  Against the deferred path: the inputs never pass through a render-target format, N is ComputeNormal()'s result as it
 * is, AO is not saturated, there is no Mask, case 0 shows FinalColor * ShadowFactor (gamma first, :114-121), and the debug table is
 * Base.frag's own (:123-143: base colour without gamma, AmbientOcclution.rgb, the interpolated vertex colour, no GBufferVis).  The
 * skydome and background follow in the same pass (view 0 only, ZE:3681-3699), depth-tested as in the deferred frame. */
void zo_set_shading(zo_ctx* c, int forward)
{
    c->forward = forward != 0;
    if (c->forward && !c->fwd) c->fwd = (zo_fsurf*)calloc((size_t)c->W * c->H, sizeof(zo_fsurf));
}

static void zo_forward_shading(zo_ctx* c, uint32_t debug_view)
{
    const XkView* V = &c->view;
    static const float Bias[16] = {0.5f,0,0,0, 0,0.5f,0,0, 0,0,1,0, 0.5f,0.5f,0,1};
    float SB[16]; zo_mat4_mul(Bias, V->ShadowmapSpace, SB);
    zo_v3 cam = zo_v3make(V->CameraInfo[0], V->CameraInfo[1], V->CameraInfo[2]);
    float dxy = 1.5f * 1.0f / (float)c->SD;
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->threads)
    for (uint32_t py = 0; py < c->H; ++py) for (uint32_t px = 0; px < c->W; ++px) {
        size_t p = (size_t)py * c->W + px;
        uint8_t* o = c->color + p * 4;
        o[0] = o[1] = o[2] = 0; o[3] = 255;                              /* clearValues[0].color, ZE:3517 */
        if (c->vis[p] != ZO_EMPTY) {
            const zo_fsurf* F = &c->fwd[p];
            float AO = F->AmbientOcclution.x;                            /* Base.frag:57: not saturated */
            zo_lit lit = zo_shade(c, SB, cam, dxy, F->BaseColor, F->Metallic, F->Roughness, F->Normal, AO, F->P);
            zo_v3 Final = zo_add(zo_add(lit.Direct, lit.Indirect), lit.RefC);
            Final = zo_v3make(zo_powf(Final.x, 0.4545f), zo_powf(Final.y, 0.4545f), zo_powf(Final.z, 0.4545f));
            zo_v3 out;
            switch (debug_view) {
            case 1: out = F->BaseColor; break;
            case 2: out = zo_v3make(F->Metallic, F->Metallic, F->Metallic); break;
            case 3: out = zo_v3make(F->Roughness, F->Roughness, F->Roughness); break;
            case 4: out = F->Normal; break;
            case 5: out = F->AmbientOcclution; break;
            case 6: out = F->VertexColor; break;
            case 7: out = lit.RefC; break;
            case 8: out = zo_v3make(lit.ShadowFactor, lit.ShadowFactor, lit.ShadowFactor); break;
            default: out = zo_scale(Final, lit.ShadowFactor); break;    /* cases 0, 9 and default */
            }
            o[0] = (uint8_t)zo_unorm(out.x, 255.0f); o[1] = (uint8_t)zo_unorm(out.y, 255.0f); o[2] = (uint8_t)zo_unorm(out.z, 255.0f);
        }
        if (debug_view == 0 && c->overlay[p]) memcpy(o, &c->overlay[p], 4);
    }
}

/* ------------------------------------------------------------------ skydome + background (ZE:3681-3699) */

static void zo_tex_free(zo_tex* t) { for (int l = 0; l < t->levels; ++l) free(t->mip[l]); memset(t, 0, sizeof *t); }
static void zo_tex_from_image(zo_ctx* c, zo_tex* tx, const zo_image* im, int srgb)
{
    size_t n = (size_t)im->width * im->height * 4;
    tx->w = im->width; tx->h = im->height;
    tx->px = (uint8_t*)malloc(n); memcpy(tx->px, im->rgba8, n);
    tx->constant = 1;
    for (size_t i = 4; i < n; ++i) if (tx->px[i] != tx->px[i & 3]) { tx->constant = 0; break; }
    zo_build_mips(c, tx, srgb);
}
int zo_set_skydome(zo_ctx* c, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const zo_image* tex)
{
    if (c->sky_set) { free(c->sky_mesh.v); free(c->sky_mesh.idx); zo_tex_free(&c->sky_tex); c->sky_set = 0; }
    if (!v || !idx || !tex || !tex->rgba8) return 0;
    c->sky_mesh.v = (XkVertex*)malloc(sizeof(XkVertex) * nv); memcpy(c->sky_mesh.v, v, sizeof(XkVertex) * nv); c->sky_mesh.nv = nv;
    c->sky_mesh.idx = (uint32_t*)malloc(4u * ni); memcpy(c->sky_mesh.idx, idx, 4u * ni); c->sky_mesh.ni = ni;
    zo_tex_from_image(c, &c->sky_tex, tex, 1);               /* RHICreateTextureResource defaults to sRGB, ZE:5860 */
    c->sky_set = 1;
    return 0;
}
int zo_set_background(zo_ctx* c, const zo_image* tex)
{
    if (c->bg_set) { zo_tex_free(&c->bg_tex); c->bg_set = 0; }
    if (!tex || !tex->rgba8) return 0;
    zo_tex_from_image(c, &c->bg_tex, tex, 1);
    c->bg_set = 1;
    return 0;
}
void zo_set_sky_flags(zo_ctx* c, int enable_skydome, int enable_background) { c->sky_enabled = enable_skydome; c->bg_enabled = enable_background; }

static uint32_t zo_pack_gamma(const float rgb[3])            /* pow(color, 0.4545), alpha 1 (Skydome.frag / Background.frag) */
{
    return zo_unorm(zo_powf(rgb[0], 0.4545f), 255.0f) | zo_unorm(zo_powf(rgb[1], 0.4545f), 255.0f) << 8 |
           zo_unorm(zo_powf(rgb[2], 0.4545f), 255.0f) << 16 | 255u << 24;
}

/* Main render pass after the lighting quad: skydome mesh (Skydome.vert/.frag, cull BACK, depth LESS against the copied
 * deferred depth, ZE:3482-3506), then the background quad at z = 1 with LESS_OR_EQUAL.  overlay[p] != 0 replaces the lit pixel. */
static void zo_sky_background(zo_ctx* c, const float* PVM)
{
    size_t n = (size_t)c->W * c->H;
    float hw = 0.5f * (float)c->W, hh = 0.5f * (float)c->H;
    memset(c->overlay, 0, n * 4);
    memcpy(c->main_depth, c->depth, n * 4);
    for (size_t i = 0; i < n; ++i) c->sky_vis[i] = ZO_EMPTY;
    if (c->sky_set && c->sky_enabled) {
        const zo_mesh* m = &c->sky_mesh;
        zo_target T = { 0, c->W, c->H, c->main_depth, c->sky_vis };
        zo_v4* clip = (zo_v4*)malloc(sizeof(zo_v4) * m->nv);
        for (uint32_t vi = 0; vi < m->nv; ++vi)
            clip[vi] = zo_mat4_point(PVM, zo_v3make(m->v[vi].Position[0], m->v[vi].Position[1], m->v[vi].Position[2]));
        for (uint32_t t = 0; t < m->ni / 3; ++t) {
            zo_v4 tc[3] = { clip[m->idx[3 * t]], clip[m->idx[3 * t + 1]], clip[m->idx[3 * t + 2]] };
            zo_raster_tri(&T, tc, t);
        }
        for (uint32_t py = 0; py < c->H; ++py) for (uint32_t px = 0; px < c->W; ++px) {
            size_t p = (size_t)py * c->W + px;
            uint32_t tri = c->sky_vis[p];
            if (tri == ZO_EMPTY) continue;
            zo_v4 tc[3]; float uv[3][2];
            for (int k = 0; k < 3; ++k) { uint32_t vi = m->idx[3 * tri + (uint32_t)k]; tc[k] = clip[vi]; uv[k][0] = m->v[vi].TexCoord[0]; uv[k][1] = m->v[vi].TexCoord[1]; }
            int cls = zo_classify(tc);
            zo_sv sv[3]; zo_setup s;
            if (cls == 1) { for (int k = 0; k < 3; ++k) sv[k] = zo_project(tc[k], hw, hh); zo_tri_setup(sv, 0, &s); }
            int32_t qx = (int32_t)(px ^ 1u), qy = (int32_t)(py ^ 1u);
            float b0[3], bh[3], bv[3];
            if (cls == 1) { zo_bary_screen(&s, sv, (int32_t)px, (int32_t)py, b0); zo_bary_screen(&s, sv, qx, (int32_t)py, bh); zo_bary_screen(&s, sv, (int32_t)px, qy, bv); }
            else { zo_bary_homog(tc, hw, hh, (int32_t)px, (int32_t)py, b0); zo_bary_homog(tc, hw, hh, qx, (int32_t)py, bh); zo_bary_homog(tc, hw, hh, (int32_t)px, qy, bv); }
            float u0 = fmaf(b0[2], uv[2][0], fmaf(b0[1], uv[1][0], b0[0] * uv[0][0])), v0 = fmaf(b0[2], uv[2][1], fmaf(b0[1], uv[1][1], b0[0] * uv[0][1]));
            float uh = fmaf(bh[2], uv[2][0], fmaf(bh[1], uv[1][0], bh[0] * uv[0][0])), vh = fmaf(bh[2], uv[2][1], fmaf(bh[1], uv[1][1], bh[0] * uv[0][1]));
            float uvv = fmaf(bv[2], uv[2][0], fmaf(bv[1], uv[1][0], bv[0] * uv[0][0])), vv = fmaf(bv[2], uv[2][1], fmaf(bv[1], uv[1][1], bv[0] * uv[0][1]));
            float sx = (px & 1u) ? 1.0f : -1.0f, sy = (py & 1u) ? 1.0f : -1.0f;
            float col[4];
            zo_tex_sample(c, &c->sky_tex, 1, u0, v0, (u0 - uh) * sx, (v0 - vh) * sx, (u0 - uvv) * sy, (v0 - vv) * sy, col);
            c->overlay[p] = zo_pack_gamma(col);
        }
        free(clip);
    }
    if (c->bg_set && c->bg_enabled) {      /* quad at z = 1.0, LESS_OR_EQUAL: only where nothing was drawn (Background.vert:33-37) */
        for (uint32_t py = 0; py < c->H; ++py) for (uint32_t px = 0; px < c->W; ++px) {
            size_t p = (size_t)py * c->W + px;
            if (!(1.0f <= c->main_depth[p])) continue;
            float u = ((float)px + 0.5f) / (float)c->W, v = ((float)py + 0.5f) / (float)c->H;
            float col[4];
            zo_tex_sample(c, &c->bg_tex, 1, u, v, 1.0f / (float)c->W, 0.0f, 0.0f, 1.0f / (float)c->H, col);
            c->overlay[p] = zo_pack_gamma(col);
        }
    }
}

/* ------------------------------------------------------------------ frame */

void zo_render(zo_ctx* c, uint32_t debug_view, uint32_t passes)
{
    float PVM[16];
    if (passes & 1u) {                                       /* shadow pass, ZE:3239-3393 */
        size_t n = (size_t)c->SD * c->SD;
        for (size_t i = 0; i < n; ++i) c->shadowmap[i] = 1.0f;
        zo_pvm(&c->shadow, PVM);
        zo_target T = { 1, c->SD, c->SD, c->shadowmap, NULL };
        zo_raster_scene(c, &T, PVM);
    }
    if (passes & 2u) {                                       /* deferred scene pass, ZE:3417-3480 */
        zo_clear_gbuffer(c);
        zo_pvm(&c->cam, PVM);
        zo_target T = { 0, c->W, c->H, c->depth, c->vis };
        zo_raster_scene(c, &T, PVM);
        zo_resolve_gbuffer(c, PVM);
        zo_sky_background(c, PVM);
    }
    if (passes & 4u) { if (c->forward) zo_forward_shading(c, debug_view); else zo_lighting(c, debug_view); }   /* lighting quad, ZE:3531-3540 */
}

const uint8_t* zo_color(zo_ctx* c) { return c->color; }
const float* zo_shadowmap(zo_ctx* c) { return c->shadowmap; }
const uint32_t* zo_visibility(zo_ctx* c) { return c->vis; }
uint64_t zo_covered_pixels(zo_ctx* c) { return c->covered; }
void zo_set_threads(zo_ctx* c, int n) { c->threads = n < 1 ? 1 : n; }
const void* zo_gbuffer(zo_ctx* c, int t)
{
    switch (t) { case 0: return c->depth; case 1: return c->scene_color; case 2: return c->gA; case 3: return c->gB;
                 case 4: return c->gC; case 5: return c->gD; default: return NULL; }
}

/* ------------------------------------------------------------------ meshlet bounds (meshopt_computeMeshletBounds, ZM:151) */
/*
 * meshoptimizer is absent from the reference tree (empty submodule, pin unrecoverable); this follows its
 * published definition: bounding sphere of the meshlet's vertices; cone axis = normalised mean of the
 * triangle normals; cone_cutoff = sqrt(1 - mindp^2) with mindp = min dot(normal_i, axis), or 1 (never
 * cull) when mindp <= 0.1; apex = centre - axis * max_t.  The oracle uses a Ritter sphere over points.
 */
int zo_meshlet_bounds(const XkVertex* v, const uint32_t* mv, const uint8_t* mt, uint32_t ntri, XkMeshlet* out)
{
    if (ntri == 0) return -1;
    /* points */
    uint32_t maxl = 0; for (uint32_t i = 0; i < ntri * 3; ++i) if (mt[i] > maxl) maxl = mt[i];
    uint32_t nvtx = maxl + 1;
    double cx = 0, cy = 0, cz = 0;
    for (uint32_t i = 0; i < nvtx; ++i) { const float* p = v[mv[i]].Position; cx += p[0]; cy += p[1]; cz += p[2]; }
    cx /= nvtx; cy /= nvtx; cz /= nvtx;
    double r = 0;
    for (uint32_t i = 0; i < nvtx; ++i) {
        const float* p = v[mv[i]].Position;
        double d = sqrt((p[0] - cx) * (p[0] - cx) + (p[1] - cy) * (p[1] - cy) + (p[2] - cz) * (p[2] - cz));
        if (d > r) r = d;
    }
    out->BoundsCenter[0] = (float)cx; out->BoundsCenter[1] = (float)cy; out->BoundsCenter[2] = (float)cz;
    out->BoundsRadius = (float)(r * (1.0 + 1e-6)) + 1e-30f;
    double ax = 0, ay = 0, az = 0;
    double* nrm = (double*)malloc(sizeof(double) * 3 * ntri);
    for (uint32_t t = 0; t < ntri; ++t) {
        const float* a = v[mv[mt[3 * t]]].Position; const float* b = v[mv[mt[3 * t + 1]]].Position; const float* c3 = v[mv[mt[3 * t + 2]]].Position;
        double e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, e2[3] = { c3[0] - a[0], c3[1] - a[1], c3[2] - a[2] };
        double n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
        double l = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (l > 0) { n[0] /= l; n[1] /= l; n[2] /= l; } else { n[0] = n[1] = n[2] = 0; }
        nrm[3 * t] = n[0]; nrm[3 * t + 1] = n[1]; nrm[3 * t + 2] = n[2];
        ax += n[0]; ay += n[1]; az += n[2];
    }
    double al = sqrt(ax * ax + ay * ay + az * az);
    if (al > 0) { ax /= al; ay /= al; az /= al; } else { ax = 1; ay = az = 0; }
    double mindp = 1;
    for (uint32_t t = 0; t < ntri; ++t) {
        double dp = nrm[3 * t] * ax + nrm[3 * t + 1] * ay + nrm[3 * t + 2] * az;
        if (dp < mindp) mindp = dp;
    }
    free(nrm);
    out->ConeAxis[0] = (float)ax; out->ConeAxis[1] = (float)ay; out->ConeAxis[2] = (float)az;
    out->ConeCutoff = (mindp <= 0.1 || al == 0) ? 1.0f : (float)sqrt(1.0 - mindp * mindp);
    out->ConeApex[0] = out->BoundsCenter[0]; out->ConeApex[1] = out->BoundsCenter[1]; out->ConeApex[2] = out->BoundsCenter[2];
    return 0;
}

/* ------------------------------------------------------------------ KAT exports */
float zo_kat_rsqrt(float x) { return zo_rsqrt(x); }
/* worst error of zo_rsqrt in ulps of the exact value over the floats with bit patterns lo, lo + step, ... < hi (normal, positive) */
double zo_kat_rsqrt_worst(uint32_t lo, uint32_t hi, uint32_t step)
{
    double worst = 0.0;
    for (uint64_t b = lo; b < hi; b += step) {
        float x = zo_u2f((uint32_t)b);
        double ref = 1.0 / sqrt((double)x);
        int ex; (void)frexp(ref, &ex);
        double err = fabs((double)zo_rsqrt(x) - ref) / ldexp(1.0, ex - 24);
        if (err > worst) worst = err;
    }
    return worst;
}
float zo_kat_D_GGX(float a, float r) { return zo_D_GGX(a, r); }
float zo_kat_V_SmithGGXCorrelated(float a, float b, float r) { return zo_V_SmithGGXCorrelated(a, b, r); }
float zo_kat_F_Schlick(float f0, float f90, float u) { return zo_F_Schlick(f0, f90, u); }
float zo_kat_Fr_DisneyDiffuse(float a, float b, float c, float r) { return zo_Fr_DisneyDiffuse(a, b, c, r); }
void  zo_kat_EnvBRDFApproxLazarov(float r, float NoV, float out[2]) { zo_EnvBRDFApproxLazarov(r, NoV, out); }
float zo_kat_ReflectionMip(float r, float m) { return zo_ReflectionMip(r, m); }
void  zo_kat_default_normal_ts(float out[3])
{
    zo_v3 n = zo_normalize(zo_v3make(127.0f / 255.0f, 127.0f / 255.0f, 255.0f / 255.0f));
    zo_v3 ts = zo_normalize(zo_v3make(fmaf(2.0f, n.x, -1.0f), fmaf(2.0f, n.y, -1.0f), fmaf(2.0f, n.z, -1.0f)));
    out[0] = ts.x; out[1] = ts.y; out[2] = ts.z;
}
float zo_kat_srgb8_to_linear(uint32_t c) { return zo_srgb_decode(c); }
void  zo_kat_perspective(float fov_deg, float aspect, float zn, float zf, float out[16]) { zo_perspective(zo_radians(fov_deg), aspect, zn, zf, out); }
void  zo_kat_lookat(const float e[3], const float c[3], const float u[3], float out[16])
{ zo_lookat(zo_v3make(e[0], e[1], e[2]), zo_v3make(c[0], c[1], c[2]), zo_v3make(u[0], u[1], u[2]), out); }
void  zo_kat_sincos(float x, float out[2]) { zo_sincosf(x, &out[0], &out[1]); }
float zo_kat_exp2(float x) { return zo_exp2f(x); }
float zo_kat_log2(float x) { return zo_log2f(x); }
float zo_kat_pow(float x, float y) { return zo_powf(x, y); }
uint16_t zo_kat_f32_to_f16(float x) { return zo_f32_to_f16(x); }
void  zo_kat_rotmat(const float e[3], float out9[9]) { zo_make_rot(e, out9); }

/* sampler KATs (tests/independent_sampler.py restates the Vulkan rules in float64 and is compared with these) */
int zo_kat_tex_sample(zo_ctx* c, const uint8_t* rgba8, uint32_t w, uint32_t h, int srgb, const float uv[2], const float duv[4], float out[4])
{
    zo_tex tx; memset(&tx, 0, sizeof tx);
    zo_image im; memset(&im, 0, sizeof im);
    im.rgba8 = rgba8; im.width = w; im.height = h;
    zo_tex_from_image(c, &tx, &im, srgb);
    zo_tex_sample(c, &tx, srgb, uv[0], uv[1], duv[0], duv[1], duv[2], duv[3], out);     /* duv = dudx, dvdx, dudy, dvdy */
    int levels = tx.levels;
    zo_tex_free(&tx);
    return levels;
}
int zo_kat_tex_mip(zo_ctx* c, const uint8_t* rgba8, uint32_t w, uint32_t h, int srgb, int level, uint8_t* out)   /* returns the level count */
{
    zo_tex tx; memset(&tx, 0, sizeof tx);
    zo_image im; memset(&im, 0, sizeof im);
    im.rgba8 = rgba8; im.width = w; im.height = h;
    zo_tex_from_image(c, &tx, &im, srgb);
    int levels = tx.levels;
    if (level >= 0 && level < levels) {
        uint32_t lw = w >> level, lh = h >> level; if (!lw) lw = 1; if (!lh) lh = 1;
        memcpy(out, tx.mip[level], (size_t)lw * lh * 4);
    }
    zo_tex_free(&tx);
    return levels;
}
int zo_kat_cube_sample(zo_ctx* c, const float dir[3], float lod, float out[3])            /* the context's cubemap (zo_set_cubemap) */
{
    zo_v3 r = zo_cube_sample(c, zo_v3make(dir[0], dir[1], dir[2]), lod);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return c->cube_levels;
}
int zo_kat_cube_mip(zo_ctx* c, int level, uint8_t* out)                                   /* 6 faces of the level, RGBA8; returns its edge */
{
    if (level < 0 || level >= c->cube_levels) return 0;
    uint32_t d = c->cube_dim >> level; if (!d) d = 1;
    memcpy(out, c->cube[level], (size_t)d * d * 4 * 6);
    return (int)d;
}
