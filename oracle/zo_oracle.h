/*
 * zo_oracle.h — API of the CPU oracle (liboracle.so).  TEST INFRASTRUCTURE ONLY.
 * See zo_math.h for the usage rule and the "parity unpinned" statement.
 *
 * The oracle is a scalar restatement of ZeldaEngine's deferred path:
 *   shadow pass -> deferred-scene (GBuffer) pass -> deferred-lighting pass
 * with NO meshlets, NO culling and NO tiling: every triangle of every draw is
 * rasterised in the reference's draw order (non-instanced draws, then instanced
 * draws; ZE:3445-3476).  The HIP path must reproduce its output.
 */
#ifndef ZO_ORACLE_H
#define ZO_ORACLE_H

#include "../include/zelda_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zo_ctx zo_ctx;

typedef struct zo_image { const uint8_t* rgba8; uint32_t width, height; } zo_image;
typedef struct zo_material { zo_image tex[7]; } zo_material;
typedef struct zo_camera { float Position[3]; float Lookat[3]; float Speed, FOV, zNear, zFar; } zo_camera;

zo_ctx* zo_create(uint32_t width, uint32_t height, uint32_t shadow_dim);
void    zo_destroy(zo_ctx*);
int     zo_mesh_create(zo_ctx*, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni);
int     zo_object_add(zo_ctx*, int mesh, const zo_material* mat, const XkInstanceData* inst, uint32_t n_inst);
void    zo_scene_clear(zo_ctx*);
int     zo_set_cubemap(zo_ctx*, const uint8_t* const faces[6], uint32_t dim);
int     zo_set_skydome(zo_ctx*, const XkVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, const zo_image* tex);
int     zo_set_background(zo_ctx*, const zo_image* tex);
void    zo_set_sky_flags(zo_ctx*, int enable_skydome, int enable_background);
void    zo_update_uniforms(zo_ctx*, const zo_camera* cam, const XkLight* dir, uint32_t n_dir,
                           const XkLight* point, uint32_t n_point, const XkLight* spot, uint32_t n_spot,
                           float roll_stage, float roll_light, float time);
void    zo_set_frame(zo_ctx*, const XkUniformBufferMVP* camera, const XkUniformBufferMVP* shadow, const XkView* view);
void    zo_get_frame(zo_ctx*, XkUniformBufferMVP* camera, XkUniformBufferMVP* shadow, XkView* view);
/* 0 (default): the deferred frame (BaseScene.frag + BaseLighting.frag); 1: the forward variant (Base.frag:46-144) - set before zo_render */
void    zo_set_shading(zo_ctx*, int forward);
/* passes bitmask: 1 shadow, 2 gbuffer, 4 lighting */
void    zo_render(zo_ctx*, uint32_t debug_view, uint32_t passes);
const uint8_t*  zo_color(zo_ctx*);              /* W*H RGBA8 */
const void*     zo_gbuffer(zo_ctx*, int target);/* 0 depth f32, 1..4 u32, 5 4xf16 */
const float*    zo_shadowmap(zo_ctx*);
const uint32_t* zo_visibility(zo_ctx*);         /* W*H winning primitive id, 0xFFFFFFFF = none */
uint64_t        zo_covered_pixels(zo_ctx*);
void            zo_set_threads(zo_ctx*, int n); /* OpenMP threads of the per-pixel stages (all-cores CPU baseline); default 1 */

/* meshlet builder + bounds (the library's clusteriser is checked against these invariants, not for equality) */
int  zo_meshlet_bounds(const XkVertex* v, const uint32_t* mv, const uint8_t* mt, uint32_t ntri, XkMeshlet* out);

/* scalar KAT entry points */
float zo_kat_rsqrt(float x);
double zo_kat_rsqrt_worst(uint32_t lo, uint32_t hi, uint32_t step);
float zo_kat_D_GGX(float NdotH, float r);
float zo_kat_V_SmithGGXCorrelated(float NdotV, float NdotL, float r);
float zo_kat_F_Schlick(float f0, float f90, float u);
float zo_kat_Fr_DisneyDiffuse(float NdotV, float NdotL, float LdotH, float r);
void  zo_kat_EnvBRDFApproxLazarov(float r, float NoV, float out[2]);
float zo_kat_ReflectionMip(float r, float maxmip);
void  zo_kat_default_normal_ts(float out[3]);
float zo_kat_srgb8_to_linear(uint32_t c);
void  zo_kat_perspective(float fov_deg, float aspect, float zn, float zf, float out[16]);
void  zo_kat_lookat(const float eye[3], const float center[3], const float up[3], float out[16]);
void  zo_kat_sincos(float x, float out[2]);
float zo_kat_exp2(float x);
float zo_kat_log2(float x);
float zo_kat_pow(float x, float y);
uint16_t zo_kat_f32_to_f16(float x);
void  zo_kat_rotmat(const float euler[3], float out9[9]);
void  zo_kat_aniso(float ax, float ay, float bx, float by, int levels, float out[3]);   /* N, lambda, x-major */
/* the samplers themselves (material textures: mips by LINEAR blits, REPEAT, trilinear, anisotropic; cubemap: textureLod) */
int   zo_kat_tex_sample(zo_ctx*, const uint8_t* rgba8, uint32_t w, uint32_t h, int srgb, const float uv[2], const float duv[4], float out[4]);
int   zo_kat_tex_mip(zo_ctx*, const uint8_t* rgba8, uint32_t w, uint32_t h, int srgb, int level, uint8_t* out);
int   zo_kat_cube_sample(zo_ctx*, const float dir[3], float lod, float out[3]);
int   zo_kat_cube_mip(zo_ctx*, int level, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
