"""A second, independent evaluation of the engine's shaders — float64 numpy, libm transcendentals, the GLSL's own evaluation order.

TEST INFRASTRUCTURE.  Written from the reference's GLSL text alone (SH = Engine/ZeldaEngine/Shaders): SH/Base.vert:23-32,
SH/BaseInstanced.vert:38-76, SH/BaseScene.frag:26-48, SH/BaseLighting.frag:147-254 and the functions of SH/Common.glsl they call
— NOT from oracle/zo_oracle.c or the HIP kernels under csrc/, whose author might have misread a shader line identically in both.  It
shares no code with them: different language, different precision (float64), different transcendentals (libm), naive evaluation
order, exact sRGB / UNORM / fp16 conversions by definition instead of by table.

What it takes as GIVEN (and therefore does not check): which primitive owns each pixel (the oracle's visibility buffer: raster
rules, depth test, clipping) and, for the lighting pass, the GBuffer and the shadow map themselves.  What it recomputes: the vertex
stage, perspective-correct interpolation, the 2x2-quad derivatives, ComputeNormal, every output of BaseScene.frag with its format
conversion, and the whole lighting shader per pixel.  Material and cubemap textures are constant per slot / per face in the scenes
it is used on, so no texture-filtering choice enters.
"""
import numpy as np

F64 = np.float64


def mat(m16):
    """column-major float[16] (GLSL / glm layout) -> 4x4 matrix in the usual row, column indexing"""
    return np.asarray(m16, dtype=F64).reshape(4, 4).T


def normalize(v):
    with np.errstate(invalid="ignore", divide="ignore"):
        return v / np.sqrt(np.sum(v * v, axis=-1, keepdims=True))


def dot(a, b):
    return np.sum(a * b, axis=-1)


def saturate(x):
    return np.clip(x, 0.0, 1.0)


def srgb_to_linear(c8):
    """VK_FORMAT_R8G8B8A8_SRGB decode of an 8-bit channel (the sRGB EOTF)"""
    x = np.asarray(c8, dtype=F64) / 255.0
    return np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


# ----------------------------------------------------------------------------------------------- vertex stage

def make_rot_matrix(R):
    """MakeRotMatrix, SH/Common.glsl:60-87 / SH/BaseInstanced.vert:38-64.  GLSL `m[i] = vec4(...)` sets COLUMN i."""
    def cols(c0, c1, c2):
        return np.array([c0, c1, c2], dtype=F64).T          # columns -> matrix
    s, c = np.sin(R[0]), np.cos(R[0])
    mx = cols((c, 0.0, s), (0.0, 1.0, 0.0), (-s, 0.0, c))
    s, c = np.sin(R[1]), np.cos(R[1])
    my = cols((c, s, 0.0), (-s, c, 0.0), (0.0, 0.0, 1.0))
    s, c = np.sin(R[2]), np.cos(R[2])
    mz = cols((1.0, 0.0, 0.0), (0.0, c, s), (0.0, -s, c))
    return mz @ my @ mx                                      # mat3(rotMat)


def vertex_stage(verts, inst, model, view, proj):
    """Base.vert:23-32 (inst is None) / BaseInstanced.vert:66-76 for one draw instance -> (clip (n,4), world pos (n,3), normal (n,3), uv (n,2))"""
    pos = np.asarray(verts["Position"], dtype=F64)
    nrm = normalize(np.asarray(verts["Normal"], dtype=F64))
    ones = np.ones((len(pos), 1))
    if inst is not None:
        rot = make_rot_matrix(np.asarray(inst["InstanceRotation"], dtype=F64))
        pos = (pos * F64(inst["InstancePScale"])) @ rot + np.asarray(inst["InstancePosition"], dtype=F64)      # row vector * mat3
    wpos = (np.hstack([pos, ones]) @ model.T)
    clip = wpos @ view.T @ proj.T
    wn = (np.hstack([nrm, ones]) @ model.T)[:, :3]           # vec4(normalize(inNormal), 1.0): the w = 1 is the shader's own
    if inst is not None:
        wn = wn @ rot
    return clip, wpos[:, :3], wn, np.asarray(verts["TexCoord"], dtype=F64)


# ----------------------------------------------------------------------------------------------- BaseScene.frag

def compute_normal(pos_dx, pos_dy, st1, st2, frag_normal, tex_normal):
    """ComputeNormal(fragPosition, fragTexCoord, fragNormal, texNormal), SH/Common.glsl:113-127"""
    with np.errstate(invalid="ignore", divide="ignore"):
        T = (st2[..., 1:2] * pos_dx - st1[..., 1:2] * pos_dy) / (st1[..., 0:1] * st2[..., 1:2] - st2[..., 0:1] * st1[..., 1:2])
    N = normalize(frag_normal)
    T = normalize(T - N * dot(N, T)[..., None])
    B = normalize(np.cross(N, T))
    n = normalize(tex_normal)
    ts = normalize(2.0 * n - 1.0)
    return normalize(T * ts[..., 0:1] + B * ts[..., 1:2] + N * ts[..., 2:3])        # TBN * v, TBN = mat3(T, B, N) (columns)


def base_scene(draws, cam, prim_ids, W, H):
    """BaseScene.frag:26-48 for every covered pixel.

    draws: list of dicts in the engine's draw order {verts, idx, instances (or None), texel (7 RGBA8 tuples), prim_base}
    cam: XkUniformBufferMVP (numpy record); prim_ids: (H, W) winning primitive per pixel (0xFFFFFFFF: none)
    -> dict of float arrays over the covered pixels + their (y, x) coordinates
    """
    model, view, proj = mat(cam["Model"]), mat(cam["View"]), mat(cam["Proj"])
    ys, xs = np.nonzero(prim_ids != 0xFFFFFFFF)
    out = {k: np.zeros((len(ys), n)) for k, n in (("scene_color", 4), ("a", 4), ("b", 4), ("c", 4), ("d", 4), ("d_per_pixel", 3), ("normal", 3))}
    pid = prim_ids[ys, xs].astype(np.int64)
    for d in draws:
        n_tris = len(d["idx"]) // 3
        n_inst = 1 if d["instances"] is None else len(d["instances"])
        sel = np.nonzero((pid >= d["prim_base"]) & (pid < d["prim_base"] + n_tris * n_inst))[0]
        if not len(sel):
            continue
        local = pid[sel] - d["prim_base"]
        inst_i, tri = local // n_tris, local % n_tris
        tex = np.asarray(d["texel"], dtype=F64)              # 7 x RGBA8
        base_color = srgb_to_linear(tex[0, :3])              # sampler1 is R8G8B8A8_SRGB (ZE:5878); the others UNORM
        metallic, rough = tex[1, 0] / 255.0, tex[2, 0] / 255.0
        tex_n, ao, emissive, mask = tex[3, :3] / 255.0, tex[4, 0] / 255.0, tex[5, :3] / 255.0, tex[6, 0] / 255.0
        for ii in np.unique(inst_i):
            m = sel[inst_i == ii]
            t = tri[inst_i == ii]
            clip, wp, wn, uv = vertex_stage(d["verts"], None if d["instances"] is None else d["instances"][ii], model, view, proj)
            corner = np.asarray(d["idx"], dtype=np.int64).reshape(-1, 3)[t]                 # (k, 3)
            c3 = clip[corner]                                                                # (k, 3, 4)
            A = np.stack([c3[..., 0], c3[..., 1], c3[..., 3]], axis=1)                       # rows x, y, w; columns = corners
            Ainv = np.linalg.inv(A)

            def weights(px, py):
                # the pixel centre in NDC (viewport 0, 0, W, H): the point of the triangle's plane that projects there is
                # sum(l_i * clip_i) with A l ~ (u, v, 1); normalised weights are the perspective-correct barycentrics
                u, v = (px + 0.5) / W * 2.0 - 1.0, (py + 0.5) / H * 2.0 - 1.0
                l = np.einsum("kij,kj->ki", Ainv, np.stack([u, v, np.ones_like(u)], axis=1))
                return l / l.sum(axis=1, keepdims=True)

            def varyings(px, py):
                l = weights(px, py)
                return (np.einsum("ki,kij->kj", l, wp[corner]), np.einsum("ki,kij->kj", l, wn[corner]), np.einsum("ki,kij->kj", l, uv[corner]))
            px, py = xs[m].astype(F64), ys[m].astype(F64)
            P0, N0, UV0 = varyings(px, py)
            # dFdx / dFdy: differences inside the 2x2 quad, the partner invocation extrapolating THIS triangle (helper lane)
            sx, sy = np.where(xs[m] & 1, 1.0, -1.0), np.where(ys[m] & 1, 1.0, -1.0)
            Ph, _, UVh = varyings(px - sx, py)
            Pv, _, UVv = varyings(px, py - sy)
            pos_dx, pos_dy = (P0 - Ph) * sx[:, None], (P0 - Pv) * sy[:, None]
            st1, st2 = (UV0 - UVh) * sx[:, None], (UV0 - UVv) * sy[:, None]
            normal = compute_normal(pos_dx, pos_dy, st1, st2, N0, np.broadcast_to(tex_n, P0.shape))
            packed = (normalize(normal) + 1.0) / 2.0
            k = len(m)
            out["scene_color"][m] = np.hstack([np.broadcast_to(emissive, (k, 3)), np.full((k, 1), mask)])
            out["a"][m] = np.hstack([packed, np.ones((k, 1))])
            out["normal"][m] = normal                                       # ComputeNormal()'s result as Base.frag uses it (forward variant)
            out["b"][m] = np.broadcast_to([metallic, 1.0, max(0.01, rough), 1.0], (k, 4))
            out["c"][m] = np.hstack([np.broadcast_to(base_color, (k, 3)), np.full((k, 1), ao)])
            out["d"][m] = np.hstack([P0, np.ones((k, 1))])
            out["d_per_pixel"][m] = np.abs(pos_dx) + np.abs(pos_dy)       # how far the position moves per pixel step (x plus y)
    out["yx"] = (ys, xs)
    return out


# ----------------------------------------------------------------------------------------------- formats

def unorm(x, bits):
    """float -> UNORM code (round to nearest; ties are the implementation's business: compare within one code)"""
    mx = (1 << bits) - 1
    with np.errstate(invalid="ignore"):
        return np.floor(np.clip(np.nan_to_num(x, nan=0.0), 0.0, 1.0) * mx + 0.5).astype(np.int64)


def unpack_rgba8(w):
    w = np.asarray(w, dtype=np.uint32)
    return np.stack([(w >> s) & 255 for s in (0, 8, 16, 24)], axis=-1).astype(np.int64)


def unpack_a2r10g10b10(w):
    """VK_FORMAT_A2R10G10B10_UNORM_PACK32: B bits 0-9, G 10-19, R 20-29, A 30-31 -> (R, G, B, A) codes"""
    w = np.asarray(w, dtype=np.uint32)
    return np.stack([(w >> 20) & 1023, (w >> 10) & 1023, w & 1023, w >> 30], axis=-1).astype(np.int64)


def unpack_rgba16f(w):
    """(H, W) uint64 (four fp16) -> float64 (H, W, 4) and the raw codes"""
    codes = np.stack([(np.asarray(w, dtype=np.uint64) >> np.uint64(s)) & np.uint64(0xFFFF) for s in (0, 16, 32, 48)], axis=-1).astype(np.uint16)
    return codes.view(np.float16).astype(F64), codes


def f16_ordinal(codes):
    """fp16 bit patterns -> integers ordered like the values (so that |a - b| counts representable steps)"""
    c = np.asarray(codes, dtype=np.int64)
    return np.where(c & 0x8000, -(c & 0x7FFF), c & 0x7FFF)


# ----------------------------------------------------------------------------------------------- BaseLighting.frag

def texture_linear_clamp(img, u, v):
    """texture(sampler2D, uv).r of a one-channel image: LINEAR filter, CLAMP_TO_EDGE (the shadow map's sampler, ZE:2532-2537)"""
    H, W = img.shape
    x, y = u * W - 0.5, v * H - 0.5
    ok = np.isfinite(x) & np.isfinite(y)
    x, y = np.where(ok, x, 0.0), np.where(ok, y, 0.0)
    x0, y0 = np.floor(x), np.floor(y)
    a, b = x - x0, y - y0
    xi0, xi1 = np.clip(x0, 0, W - 1).astype(np.int64), np.clip(x0 + 1, 0, W - 1).astype(np.int64)
    yi0, yi1 = np.clip(y0, 0, H - 1).astype(np.int64), np.clip(y0 + 1, 0, H - 1).astype(np.int64)
    top = img[yi0, xi0] * (1 - a) + img[yi0, xi1] * a
    bot = img[yi1, xi0] * (1 - a) + img[yi1, xi1] * a
    return np.where(ok, top * (1 - b) + bot * b, np.nan)


def cube_face_constant(face_colors_srgb8, R):
    """textureLod(samplerCube, R, lod) for a cubemap whose six faces are each ONE colour (so no filter or mip choice matters):
    face selection by the major axis (Vulkan: +X, -X, +Y, -Y, +Z, -Z; z wins ties over y over x), R8G8B8A8_SRGB decode."""
    ax, ay, az = np.abs(R[..., 0]), np.abs(R[..., 1]), np.abs(R[..., 2])
    face = np.where((az >= ax) & (az >= ay), np.where(R[..., 2] >= 0, 4, 5),
                    np.where(ay >= ax, np.where(R[..., 1] >= 0, 2, 3), np.where(R[..., 0] >= 0, 0, 1)))
    lin = srgb_to_linear(np.asarray(face_colors_srgb8, dtype=F64)[:, :3])
    return lin[face]


def f_schlick(f0, f90, u):
    return f0 + (f90 - f0) * np.power(1.0 - u, 5.0)


def bxdf(diffuse_color, roughness, LoH, NoV, NoL, NoH):
    """DefaultLitBxDF, SH/Common.glsl:259-282 -> Diffuse + Specular"""
    F0 = 0.04
    F90 = saturate(50.0 * F0)
    F = f_schlick(F0, F90, LoH)
    a2 = roughness * roughness
    ggxv = NoL * np.sqrt(NoV * NoV * (1.0 - a2) + a2)
    ggxl = NoV * np.sqrt(NoL * NoL * (1.0 - a2) + a2)
    ggx = ggxv + ggxl
    with np.errstate(invalid="ignore", divide="ignore"):
        vis = np.where(ggx > 0.0, 0.5 / ggx, 0.0)
        f = (NoH * a2 - NoH) * NoH + 1.0
        D = a2 / (3.14159265359 * f * f)      # the shader's PI literal
    Fr = F * D * vis
    e_bias = 0.0 * (1.0 - roughness) + 0.5 * roughness
    e_factor = 1.0 * (1.0 - roughness) + (1.0 / 1.51) * roughness
    fd90 = e_bias + 2.0 * LoH * LoH * roughness
    Fd = f_schlick(1.0, fd90, NoL) * f_schlick(1.0, fd90, NoV) * e_factor
    return diffuse_color * (1.0 - F)[..., None] * Fd[..., None] + Fr[..., None]


def lighting(gb, shadow_map, view, cube_face_colors, W, H, pcf_eps=0.0, forward=False):
    """BaseLighting.frag:147-227 + case 0 of the switch for every pixel of the W x H quad.
    forward=True: Base.frag:46-123 instead - the same text except that N is used as ComputeNormal() returned it, AO is not saturated,
    there is no Mask, and case 0 shows FinalColor * ShadowFactor (after the gamma); gb then holds the fragment's unquantised inputs.

    gb: dict scene_color / a / b / c (float RGBA as texture() returns them) and d (fp16 values), each (H, W, 4); view: XkView record
    pcf_eps: added to the reference depth of the 25 shadow comparisons - the shader's one discontinuity: a caller that wants to know
    which pixels sit on it evaluates with +-eps and looks at the spread
    -> (H, W, 3) float colour before the UNORM store
    """
    PI = 3.14159265359
    base_color = gb["c"][..., :3]
    metallic = saturate(gb["b"][..., 0])
    roughness = np.maximum(0.01, saturate(gb["b"][..., 2]))
    normal = gb["a"][..., :3] * 2.0 - 1.0
    ao = gb["c"][..., 3] if forward else saturate(gb["c"][..., 3])
    mask = gb["scene_color"][..., 3]
    N = normal if forward else normalize(normal)
    P = gb["d"][..., :3]
    cam = np.asarray(view["CameraInfo"], dtype=F64)[:3]
    V = normalize(cam - P)
    NdotV = saturate(dot(N, V))

    # ComputeShadowCoord + ComputePCF(sampler, ShadowCoord / ShadowCoord.w, 2), SH/Common.glsl:294-342
    bias = np.array([[0.5, 0, 0, 0.5], [0, 0.5, 0, 0.5], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=F64)      # BiasMat (columns as written in the GLSL)
    SB = bias @ mat(view["ShadowmapSpace"])
    with np.errstate(invalid="ignore", divide="ignore"):
        sc = np.concatenate([P, np.ones(P.shape[:-1] + (1,))], axis=-1) @ SB.T
        sc = sc / sc[..., 3:4]
        SD = shadow_map.shape[0]
        dx = 1.5 * 1.0 / SD
        total = np.zeros(P.shape[:-1])
        for x in range(-2, 3):
            for y in range(-2, 3):
                f = np.ones(P.shape[:-1])
                inside = (sc[..., 2] > -1.0) & (sc[..., 2] < 1.0)
                dist = texture_linear_clamp(shadow_map.astype(F64), sc[..., 0] + dx * x, sc[..., 1] + dx * y)
                f = np.where(inside & (sc[..., 3] > 0.0) & (dist < sc[..., 2] + pcf_eps), 0.1, f)
                total += f
        shadow = total / 25.0

    direct = np.zeros(P.shape)
    diffuse_color = base_color * (1.0 - metallic)[..., None]
    n_dir, n_point = int(view["LightsCount"][0]), int(view["LightsCount"][1])
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(n_dir):
            Lt = view["DirectionalLights"][i]
            L = normalize(np.asarray(Lt["Direction"], dtype=F64)[:3]) * np.ones(P.shape)
            Hh = normalize(V + L)
            b = bxdf(diffuse_color, roughness, saturate(dot(L, Hh)), NdotV, saturate(dot(N, L)), saturate(dot(N, Hh)))
            ndotl = np.clip(dot(normalize(N), L), 0.0, 1.0)
            apply = ndotl[..., None] * F64(Lt["Color"][3]) * np.asarray(Lt["Color"], dtype=F64)[:3]
            direct = direct + apply * b * shadow[..., None]
        for i in range(n_point):
            Lt = view["PointLights"][i]
            lp = np.asarray(Lt["Position"], dtype=F64)[:3]
            L = normalize(lp - P)
            Hh = normalize(V + L)
            b = bxdf(diffuse_color, roughness, saturate(dot(L, Hh)), NdotV, saturate(dot(N, L)), saturate(dot(N, Hh)))
            ndotl = np.clip(dot(normalize(N), L), 0.0, 1.0)
            falloff = F64(Lt["Direction"][3])
            dist = np.sqrt(dot(lp - P, lp - P))
            att = 1.0 - (np.clip(dist, 0.0, falloff) - 0.0) / (falloff - 0.0) * (1.0 - 0.0) + 0.0      # 1 - remap(dist, 0, falloff, 0, 1)
            apply = (ndotl * F64(Lt["Color"][3]))[..., None] * np.asarray(Lt["Color"], dtype=F64)[:3] * att[..., None]
            direct = direct + apply * b

        indirect = diffuse_color / PI * ao[..., None] * 0.3 * shadow[..., None]

        # ComputeF0(0.5, BaseColor, Metallic); EnvBRDFApprox; refract; cubemap; GetSpecularOcclusion
        bc = np.clip(base_color, 0.04, 1.0)
        dsf0 = 0.04 * 2.0 * 0.5
        refl_spec = (1.0 - metallic)[..., None] * dsf0 + metallic[..., None] * bc
        r = roughness[..., None] * np.array([-1.0, -0.0275, -0.572, 0.022]) + np.array([1.0, 0.0425, 1.04, -0.04])
        a004 = np.minimum(r[..., 0] * r[..., 0], np.exp2(-9.28 * NdotV)) * r[..., 0] + r[..., 1]
        AB = np.stack([-1.04 * a004 + r[..., 2], 1.04 * a004 + r[..., 3]], axis=-1)
        F90 = saturate(50.0 * refl_spec[..., 1])
        refl_brdf = refl_spec * AB[..., 0:1] + (F90 * AB[..., 1])[..., None]
        eta = 1.00 / 1.52
        Nn = normalize(N)
        dNI = dot(Nn, V)
        k = 1.0 - eta * eta * (1.0 - dNI * dNI)
        R = np.where((k < 0.0)[..., None], 0.0, eta * V - (eta * dNI + np.sqrt(np.maximum(k, 0.0)))[..., None] * Nn)
        refl_l = cube_face_constant(cube_face_colors, R) * 10.0
        refl_v = saturate(np.power(NdotV + ao, roughness * roughness) - 1.0 + ao)
        refl = refl_l * refl_v[..., None] * refl_brdf

        if forward:
            final = np.power(direct + indirect + refl, 0.4545) * shadow[..., None]
        else:
            final = (direct + indirect + refl) * mask[..., None]
            final = np.power(final, 0.4545)
    return final
