/*
 * zelda_abi.h — frozen byte layouts of the ZeldaEngine scene-submission surface.
 *
 * These are the wire/memory layouts a ZeldaEngine scene is made of.  They are
 * restated here as plain C (no glm) so that existing producers (the C++ engine
 * structs, the `.meshlet` file, the livelink JSON) drop in unchanged.
 *
 * Reference (all under Engine/ZeldaEngine/ZeldaEngine.cpp unless noted):
 *   XkVertex            ZE:417-422   (attribute map ZE:433-457)
 *   XkInstanceData      ZE:409-414   (attribute map ZE:497-515)
 *   XkMeshlet           ZE:689-701   == Engine/ZeldaMeshlet/ZeldaMeshlet.cpp:39-49
 *   meshlet-file Vertex ZM:19-23
 *   XkLight             ZE:772-787   == Shaders/Common.glsl:3-13
 *   XkUniformBufferMVP  ZE:384-388
 *   XkView              ZE:922-940   == Shaders/BaseLighting.frag:5-20
 *   XkGlobalConstants   ZE:903-919
 *   limits              ZE:84-87 (16 dir / 512 point / 16 spot lights, 1024^2 shadow map)
 *
 * All matrices are column-major float[16] (glm::mat4): m[c*4 + r].
 */
#ifndef ZELDA_ABI_H
#define ZELDA_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XK_MAX_DIRECTIONAL_LIGHTS_NUM 16   /* ZE:84 */
#define XK_MAX_POINT_LIGHTS_NUM       512  /* ZE:85 */
#define XK_MAX_SPOT_LIGHTS_NUM        16   /* ZE:86 */
#define XK_SHADOWMAP_DIM              1024 /* ZE:87 */
#define XK_PBR_SAMPLER_NUMBER         7    /* ZE:82: bc, m, r, n, ao, ev, ms */
#define XK_LIVELINK_PORT              8080 /* ZE:1636 */
#define XK_LIVELINK_RECV_MAX          65720/* ZE:972-973 */

/* EXkRenderFlags, ZE:352-366.  operator== is "has all bits of b" (ZE:378-381). */
enum {
    XK_RF_NONE = 1 << 0, XK_RF_VERTEX_INDEXED = 1 << 1, XK_RF_INSTANCED = 1 << 2,
    XK_RF_SCREEN_RECT = 1 << 3, XK_RF_TWO_SIDED = 1 << 4, XK_RF_NO_DEPTH_TEST = 1 << 5,
    XK_RF_SHADOW = 1 << 6, XK_RF_SKYDOME = 1 << 7, XK_RF_BACKGROUND = 1 << 8,
    XK_RF_FORWARD_SHADING = 1 << 9, XK_RF_DEFERRED_SCENE = 1 << 10, XK_RF_DEFERRED_LIGHTING = 1 << 11
};

typedef struct XkVertex {          /* 44 B, ZE:417-422 */
    float Position[3];             /* @0  */
    float Normal[3];               /* @12 */
    float Color[3];                /* @24 (always 1,1,1 from the OBJ loader, ZE:6933) */
    float TexCoord[2];             /* @36 */
} XkVertex;

typedef struct XkInstanceData {    /* 32 B stride, ZE:409-414 */
    float   InstancePosition[3];   /* @0  */
    float   InstanceRotation[3];   /* @12 euler, radians */
    float   InstancePScale;        /* @24 */
    uint8_t InstanceTexIndex;      /* @28 */
    uint8_t _pad[3];
} XkInstanceData;

typedef struct XkMeshlet {         /* 64 B, ZE:689-701 */
    uint32_t VertexOffset;         /* @0  into meshletVertices[] */
    uint32_t VertexCount;          /* @4  <= 64 */
    uint32_t TriangleOffset;       /* @8  into meshletTriangles[] (bytes) */
    uint32_t TriangleCount;        /* @12 <= 124 */
    float    BoundsCenter[3];      /* @16 */
    float    BoundsRadius;         /* @28 */
    float    ConeApex[3];          /* @32 */
    float    ConeAxis[3];          /* @44 */
    float    ConeCutoff;           /* @56 */
    uint32_t BindlessContext;      /* @60 (pad in the .meshlet file, ZM:48) */
} XkMeshlet;

typedef struct XkMeshletFileVertex { /* 32 B, ZM:19-23 */
    float x, y, z, nx, ny, nz, u, v;
} XkMeshletFileVertex;

typedef struct XkLight {           /* 64 B, ZE:772-787 */
    float Position[4];             /* xyz + type */
    float Color[4];                /* rgb + intensity */
    float Direction[4];            /* xyz + radius */
    float LightInfo[4];
} XkLight;

typedef struct XkUniformBufferMVP { /* 192 B, ZE:384-388 */
    float Model[16];
    float View[16];
    float Proj[16];
} XkUniformBufferMVP;

typedef struct XkView {            /* 35068 B, ZE:922-940 */
    float   ViewProjSpace[16];     /* @0   */
    float   ShadowmapSpace[16];    /* @64  */
    float   LocalToWorld[16];      /* @128 */
    float   CameraInfo[4];         /* @192 camera pos, FOV */
    float   ViewportInfo[4];       /* @208 W, H, rightBar, bottomBar */
    XkLight DirectionalLights[XK_MAX_DIRECTIONAL_LIGHTS_NUM]; /* @224   */
    XkLight PointLights[XK_MAX_POINT_LIGHTS_NUM];             /* @1248  */
    XkLight SpotLights[XK_MAX_SPOT_LIGHTS_NUM];               /* @34016 */
    int32_t LightsCount[4];        /* @35040 nDir, nPoint, nSpot, cubemap max mips */
    float   Time;                  /* @35056 */
    float   zNear;                 /* @35060 */
    float   zFar;                  /* @35064 */
} XkView;

typedef struct XkGlobalConstants { /* 24 B, ZE:903-919 */
    float    BaseColorOverride, MetallicOverride, SpecularOverride, RoughnessOverride;
    uint32_t SpecConstants;        /* debug view 0..9 */
    uint32_t SpecConstantsCount;   /* = 10 */
} XkGlobalConstants;

typedef struct XkDrawIndexedIndirectCommand { /* 20 B == VkDrawIndexedIndirectCommand, ZE:4229-4236 */
    uint32_t indexCount, instanceCount, firstIndex;
    int32_t  vertexOffset;
    uint32_t firstInstance;
} XkDrawIndexedIndirectCommand;

#ifdef __cplusplus
}
#define XK_SA(c, m) static_assert(c, m)
#else
#define XK_SA(c, m) _Static_assert(c, m)
#endif

XK_SA(sizeof(XkVertex) == 44 && offsetof(XkVertex, Normal) == 12 && offsetof(XkVertex, Color) == 24 &&
      offsetof(XkVertex, TexCoord) == 36, "XkVertex layout");
XK_SA(sizeof(XkInstanceData) == 32 && offsetof(XkInstanceData, InstanceRotation) == 12 &&
      offsetof(XkInstanceData, InstancePScale) == 24 && offsetof(XkInstanceData, InstanceTexIndex) == 28,
      "XkInstanceData layout");
XK_SA(sizeof(XkMeshlet) == 64 && offsetof(XkMeshlet, BoundsCenter) == 16 && offsetof(XkMeshlet, BoundsRadius) == 28 &&
      offsetof(XkMeshlet, ConeApex) == 32 && offsetof(XkMeshlet, ConeAxis) == 44 &&
      offsetof(XkMeshlet, ConeCutoff) == 56 && offsetof(XkMeshlet, BindlessContext) == 60, "XkMeshlet layout");
XK_SA(sizeof(XkMeshletFileVertex) == 32, "meshlet file vertex layout");
XK_SA(sizeof(XkLight) == 64, "XkLight layout");
XK_SA(sizeof(XkUniformBufferMVP) == 192, "XkUniformBufferMVP layout");
XK_SA(sizeof(XkView) == 35068 && offsetof(XkView, CameraInfo) == 192 && offsetof(XkView, ViewportInfo) == 208 &&
      offsetof(XkView, DirectionalLights) == 224 && offsetof(XkView, PointLights) == 1248 &&
      offsetof(XkView, SpotLights) == 34016 && offsetof(XkView, LightsCount) == 35040 &&
      offsetof(XkView, Time) == 35056 && offsetof(XkView, zNear) == 35060 && offsetof(XkView, zFar) == 35064,
      "XkView layout");
XK_SA(sizeof(XkGlobalConstants) == 24, "XkGlobalConstants layout");
XK_SA(sizeof(XkDrawIndexedIndirectCommand) == 20, "indirect command layout");

#endif /* ZELDA_ABI_H */
