// zr_world.cpp — XkWorld JSON schema (load/save), Profab registry and the TCP livelink server.
//
// Replaces XkWorld::Load / Save / Reset (ZE:1051-1290, rapidjson), the scene-building half of CreateEngineScene
// (ZE:4250-4267) and the socket listener thread (ZE:1617-1710, WinSock only in the reference: the #else branch is
// "@TODO: Implement for macOS and Linux").  The wire contract is kept: IPv4 TCP, one recv of <= 65720 bytes per
// connection holding the whole JSON, no reply bytes, shutdown(SHUT_WR).  The reference parses on the socket thread
// straight into the live World (a data race, ZE:1687 vs 4296); here the listener parses into a private ZrWorld and
// zr_livelink_poll swaps it in on the render thread, which is DrawFrame's bReloadScene pickup (ZE:1943-1951).
#include "zr_ctx.h"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>

#define XK_WORLD_MAX_INSTANCES_PER_OBJECT 4194304u     // 2^22 per ObjectDesc (the sample scene uses 10 000); bigger scenes use zr_object_add

// ------------------------------------------------------------------------------------------------ minimal JSON

namespace {

struct JValue;
using JPtr = std::unique_ptr<JValue>;
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false; double num = 0; bool is_int = false;
    std::string str;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;
    const JValue* get(const char* key) const { for (auto& kv : obj) if (kv.first == key) return kv.second.get(); return nullptr; }
};

struct JParser {
    const char* p; const char* end; std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool parse_string(std::string& out)
    {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u': {
                    if (end - p < 5) return fail("bad \\u escape");
                    unsigned cp = 0;
                    for (int i = 1; i <= 4; ++i) {
                        char ch = p[i]; cp <<= 4;
                        if (ch >= '0' && ch <= '9') cp |= (unsigned)(ch - '0');
                        else if (ch >= 'a' && ch <= 'f') cp |= (unsigned)(ch - 'a' + 10);
                        else if (ch >= 'A' && ch <= 'F') cp |= (unsigned)(ch - 'A' + 10);
                        else return fail("bad \\u escape");
                    }
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: return fail("bad escape");
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool parse(JValue& v, int depth = 0)
    {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            v.kind = JValue::Obj; ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws(); std::string k;
                if (!parse_string(k)) return false;
                ws(); if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                JPtr c(new JValue());
                if (!parse(*c, depth + 1)) return false;
                v.obj.emplace_back(std::move(k), std::move(c));
                ws(); if (p >= end) return fail("unexpected end");
                if (*p == ',') { ++p; continue; }
                if (*p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.kind = JValue::Arr; ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                JPtr c(new JValue());
                if (!parse(*c, depth + 1)) return false;
                v.arr.push_back(std::move(c));
                ws(); if (p >= end) return fail("unexpected end");
                if (*p == ',') { ++p; continue; }
                if (*p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { v.kind = JValue::Str; return parse_string(v.str); }
        if (end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JValue::Bool; v.b = true; p += 4; return true; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JValue::Bool; v.b = false; p += 5; return true; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { v.kind = JValue::Null; p += 4; return true; }
        const char* s = p;
        if (p < end && *p == '-') ++p;
        bool digits = false, isint = true;
        while (p < end && *p >= '0' && *p <= '9') { ++p; digits = true; }
        if (p < end && *p == '.') { isint = false; ++p; while (p < end && *p >= '0' && *p <= '9') { ++p; digits = true; } }
        if (p < end && (*p == 'e' || *p == 'E')) { isint = false; ++p; if (p < end && (*p == '+' || *p == '-')) ++p; while (p < end && *p >= '0' && *p <= '9') ++p; }
        if (!digits) return fail("invalid value");
        v.kind = JValue::Num; v.is_int = isint; v.num = strtod(std::string(s, p).c_str(), nullptr);
        return true;
    }
};

struct Reader {       // typed access with rapidjson-like strictness: a missing key or wrong type is an error, not a default
    std::string err;
    const JValue* need(const JValue* o, const char* k, JValue::Kind kind)
    {
        const JValue* v = (o && o->kind == JValue::Obj) ? o->get(k) : nullptr;
        if (!v) { if (err.empty()) err = std::string("missing key '") + k + "'"; return nullptr; }
        if (v->kind != kind) { if (err.empty()) err = std::string("wrong type for key '") + k + "'"; return nullptr; }
        return v;
    }
    float f(const JValue* o, const char* k) { const JValue* v = need(o, k, JValue::Num); return v ? (float)v->num : 0.0f; }
    uint32_t u(const JValue* o, const char* k)
    {
        const JValue* v = need(o, k, JValue::Num);
        if (v && (!v->is_int || v->num < 0 || v->num > 4294967295.0)) { if (err.empty()) err = std::string("key '") + k + "' is not a uint"; return 0; }
        return v ? (uint32_t)v->num : 0u;
    }
    bool b(const JValue* o, const char* k) { const JValue* v = need(o, k, JValue::Bool); return v ? v->b : false; }
    std::string s(const JValue* o, const char* k) { const JValue* v = need(o, k, JValue::Str); return v ? v->str : std::string(); }
    void fa(const JValue* o, const char* k, float* dst, size_t n)
    {
        const JValue* v = need(o, k, JValue::Arr);
        if (!v) return;
        if (v->arr.size() < n) { if (err.empty()) err = std::string("array '") + k + "' too short"; return; }
        for (size_t i = 0; i < n; ++i) {
            if (v->arr[i]->kind != JValue::Num) { if (err.empty()) err = std::string("array '") + k + "' holds a non-number"; return; }
            dst[i] = (float)v->arr[i]->num;
        }
    }
};

void world_reset(ZrWorld& w)     // XkWorld::Reset, ZE:1265-1290
{
    w = ZrWorld();
    w.EnableSkydome = true; w.OverrideSkydome = true; w.SkydomeFileName = "Content/Textures/skydome.png";
    w.OverrideCubemap = true;
    const char* cm[6] = { "Content/Textures/cubemap_X0.png", "Content/Textures/cubemap_X1.png", "Content/Textures/cubemap_Y2.png",
                          "Content/Textures/cubemap_Y3.png", "Content/Textures/cubemap_Z4.png", "Content/Textures/cubemap_Z5.png" };
    for (int i = 0; i < 6; ++i) w.CubemapFileNames[i] = cm[i];
    w.EnableBackground = true; w.OverrideBackground = true; w.BackgroundFileName = "Content/Textures/background.png";
    const zr_camera cam = { { 5.0f, 5.0f, 5.0f }, { 0.0f, 0.0f, 0.0f }, 2.5f, 45.0f, 0.1f, 45.0f };   // ResetToFocus, ZE:879-887
    w.MainCamera = cam;
}

// XkWorld::Load, ZE:1051-1147.  Returns false with `err` set on a parse or schema error.
bool world_parse(const char* utf8, size_t len, ZrWorld& w, std::string& err)
{
    world_reset(w);
    JParser jp{ utf8, utf8 + len, {} };
    JValue root;
    if (!jp.parse(root)) { err = "[WORLD] JSON parse error: " + jp.err; return false; }
    jp.ws();
    if (jp.p != jp.end) { err = "[WORLD] JSON parse error: trailing characters"; return false; }
    if (root.kind != JValue::Obj) { err = "[WORLD] JSON parse error: document is not an object"; return false; }
    Reader R;
    const JValue* cam = R.need(&root, "MainCamera", JValue::Obj);
    R.fa(cam, "Position", w.MainCamera.Position, 3); R.fa(cam, "Lookat", w.MainCamera.Lookat, 3);
    w.MainCamera.FOV = R.f(cam, "FOV"); w.MainCamera.Speed = R.f(cam, "Speed");
    w.MainCamera.zNear = R.f(cam, "zNear"); w.MainCamera.zFar = R.f(cam, "zFar");
    const JValue* sky = R.need(&root, "Skydome", JValue::Obj);
    w.EnableSkydome = R.b(sky, "EnableSkydome"); w.OverrideSkydome = R.b(sky, "OverrideSkydome");
    w.SkydomeFileName = R.s(sky, "SkydomeFileName"); w.OverrideCubemap = R.b(sky, "OverrideCubemap");
    if (const JValue* names = R.need(sky, "CubemapFileNames", JValue::Arr)) {
        if (names->arr.size() > 6) { if (R.err.empty()) R.err = "CubemapFileNames has more than 6 entries"; }
        else for (size_t i = 0; i < names->arr.size(); ++i) {
            if (names->arr[i]->kind != JValue::Str) { if (R.err.empty()) R.err = "CubemapFileNames holds a non-string"; break; }
            w.CubemapFileNames[i] = names->arr[i]->str;
        }
    }
    const JValue* bg = R.need(&root, "Background", JValue::Obj);
    w.EnableBackground = R.b(bg, "EnableBackground"); w.OverrideBackground = R.b(bg, "OverrideBackground");
    w.BackgroundFileName = R.s(bg, "BackgroundFileName");
    auto lights = [&](const char* key, std::vector<ZrLightDesc>& out) {
        const JValue* a = R.need(&root, key, JValue::Arr);
        if (!a) return;
        for (auto& e : a->arr) {
            ZrLightDesc l; memset(&l, 0, sizeof l);
            const JValue* o = e.get();
            if (o->kind != JValue::Obj) { if (R.err.empty()) R.err = std::string(key) + " holds a non-object"; return; }
            R.fa(o, "Position", l.Position, 3); l.Type = R.u(o, "Type"); R.fa(o, "Color", l.Color, 3);
            l.Intensity = R.f(o, "Intensity"); R.fa(o, "Direction", l.Direction, 3); l.Radius = R.f(o, "Radius");
            R.fa(o, "ExtraData", l.ExtraData, 4);
            out.push_back(l);
        }
    };
    lights("DirectionalLights", w.DirectionalLights); lights("PointLights", w.PointLights); lights("SpotLights", w.SpotLights);
    if (const JValue* objs = R.need(&root, "Objects", JValue::Arr))
        for (auto& e : objs->arr) {
            const JValue* o = e.get();
            if (o->kind != JValue::Obj) { if (R.err.empty()) R.err = "Objects holds a non-object"; break; }
            ZrObjectDesc d;
            d.RenderFlags = R.u(o, "RenderFlags"); d.ProfabName = R.s(o, "ProfabName"); d.InstanceCount = R.u(o, "InstanceCount");
            d.MinRadius = R.f(o, "MinRadius"); d.MaxRadius = R.f(o, "MaxRadius"); d.MinRotYaw = R.f(o, "MinRotYaw");
            d.MaxRotYaw = R.f(o, "MaxRotYaw"); d.MinRotRoll = R.f(o, "MinRotRoll"); d.MaxRotRoll = R.f(o, "MaxRotRoll");
            d.MinRotPitch = R.f(o, "MinRotPitch"); d.MaxRotPitch = R.f(o, "MaxRotPitch"); d.MinPScale = R.f(o, "MinPScale");
            d.MaxPScale = R.f(o, "MaxPScale");
            // the payload is unauthenticated: one 64 KiB packet must not be able to ask for gigabytes of instances
            if (d.InstanceCount > XK_WORLD_MAX_INSTANCES_PER_OBJECT && R.err.empty())
                R.err = "InstanceCount of '" + d.ProfabName + "' exceeds " + std::to_string(XK_WORLD_MAX_INSTANCES_PER_OBJECT);
            w.ObjectDescs.push_back(d);
        }
    if (!R.err.empty()) { err = "[WORLD] JSON schema error: " + R.err; return false; }
    if (w.DirectionalLights.size() > XK_MAX_DIRECTIONAL_LIGHTS_NUM || w.PointLights.size() > XK_MAX_POINT_LIGHTS_NUM ||
        w.SpotLights.size() > XK_MAX_SPOT_LIGHTS_NUM) { err = "[WORLD] too many lights (limits 16 / 512 / 16, ZE:84-86)"; return false; }
    w.loaded = true;
    return true;
}

// PCG-XSH-RR 64/32; stands in for mt19937(std::rand()) of XkObjectDesc::RandRange (ZE:592-603), which is libc-specific
struct Pcg32 {
    uint64_t state, inc;
    explicit Pcg32(uint64_t seed, uint64_t seq = 54) { state = 0; inc = (seq << 1) | 1; next(); state += seed; next(); }
    uint32_t next()
    {
        const uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    float unit() { return (float)(next() >> 8) * 5.9604644775390625e-8f; }
    float range(float a, float b) { return a + (b - a) * unit(); }
};

// XkObjectDesc::GenerateInstance, ZE:573-589
void generate_instances(const ZrObjectDesc& d, uint64_t seed, std::vector<XkInstanceData>& out)
{
    out.assign(d.InstanceCount, XkInstanceData());
    Pcg32 g(seed);
    for (uint32_t i = 0; i < d.InstanceCount; ++i) {
        XkInstanceData& I = out[i]; memset(&I, 0, sizeof I);
        const float deg = g.range(0.0f, 360.0f);
        const float dist = g.range(d.MinRadius, d.MaxRadius);
        const float rad = deg * 0.01745329251994329576923690768489f;
        I.InstancePosition[0] = sinf(rad) * dist; I.InstancePosition[1] = cosf(rad) * dist; I.InstancePosition[2] = 0.0f;
        I.InstanceRotation[0] = 0.0f; I.InstanceRotation[1] = 3.14159265358979323846f * g.range(0.0f, 180.0f); I.InstanceRotation[2] = 0.0f;
        I.InstancePScale = g.range(d.MinPScale, d.MaxPScale);
        I.InstanceTexIndex = (uint8_t)(g.next() >> 24);
    }
}

int world_uniforms(zr_ctx* c, const ZrWorld& w, float roll_stage, float roll_light, float time);

int apply_world(zr_ctx* c, const ZrWorld& w)
{
    // CreateEngineScene (ZE:4250-4267): drop the render objects, keep meshes and registered Profabs
    int rc = ZR_OK;
    {
        auto keep_m = std::move(c->meshes); auto keep_p = std::move(c->profabs);
        for (auto& o : c->objects) {
            if (o.d_inst) { (void)hipFree(o.d_inst); o.d_inst = nullptr; }
            for (auto& t : o.d_tex) if (t) { (void)hipFree(t); t = nullptr; }
        }
        c->objects.clear(); c->scene_dirty = true;
        c->meshes = std::move(keep_m); c->profabs = std::move(keep_p);
    }
    c->world = w;
    zr_set_sky_flags(c, w.EnableSkydome ? 1 : 0, w.EnableBackground ? 1 : 0);       // gates of ZE:3682 / ZE:3693
    rc = zr_world_apply_overrides(c, w);            // CreateEngineScene (1): cubemap / skydome / background named by the world, via ASSETS()
    for (size_t oi = 0; oi < w.ObjectDescs.size() && rc == ZR_OK; ++oi) {
        const ZrObjectDesc& d = w.ObjectDescs[oi];
        auto it = c->profabs.find(d.ProfabName);
        if (it == c->profabs.end() && c->assets_on) {          // not registered by the host: look for Profabs/<name> on disk (ZE:4922-5000)
            int found = 0;
            rc = zr_profab_from_disk(c, d.ProfabName, &found);
            if (rc) break;
            it = c->profabs.find(d.ProfabName);
        }
        if (it == c->profabs.end()) continue;      // no such Profab directory: the engine finds no models and draws nothing
        std::vector<XkInstanceData> inst;
        if (d.InstanceCount > 1) generate_instances(d, 1234u + oi, inst);
        for (const ZrProfab& pf : it->second) {
            rc = zr_object_add_internal(c, pf.mesh, pf.mat, inst.empty() ? nullptr : inst.data(), (uint32_t)inst.size());
            if (rc) break;
        }
    }
    if (rc) return rc;
    return world_uniforms(c, w, 0.0f, 0.0f, 0.0f);
}

int world_uniforms(zr_ctx* c, const ZrWorld& w, float roll_stage, float roll_light, float time)
{
    std::vector<XkLight> L[3];
    const std::vector<ZrLightDesc>* src[3] = { &w.DirectionalLights, &w.PointLights, &w.SpotLights };
    for (int k = 0; k < 3; ++k)
        for (const ZrLightDesc& l : *src[k]) {      // XkLight(const XkLightDesc&), ZE:781-787
            XkLight x;
            x.Position[0] = l.Position[0]; x.Position[1] = l.Position[1]; x.Position[2] = l.Position[2]; x.Position[3] = (float)(uint8_t)l.Type;
            x.Color[0] = l.Color[0]; x.Color[1] = l.Color[1]; x.Color[2] = l.Color[2]; x.Color[3] = l.Intensity;
            x.Direction[0] = l.Direction[0]; x.Direction[1] = l.Direction[1]; x.Direction[2] = l.Direction[2]; x.Direction[3] = l.Radius;
            memcpy(x.LightInfo, l.ExtraData, 16);
            L[k].push_back(x);
        }
    return zr_update_uniforms(c, &w.MainCamera, L[0].data(), (uint32_t)L[0].size(), L[1].data(), (uint32_t)L[1].size(),
                              L[2].data(), (uint32_t)L[2].size(), roll_stage, roll_light, time);
}

// No C++ exception crosses the ABI: an allocation failure while instantiating a world becomes an error code.
int apply_world_guarded(zr_ctx* c, const ZrWorld& w)
{
    try { return apply_world(c, w); }
    catch (const std::bad_alloc&) { return zr_fail(c, ZR_ERR_OOM, "[WORLD] out of host memory while building the scene"); }
    catch (const std::exception& e) { return zr_fail(c, ZR_ERR_STATE, std::string("[WORLD] ") + e.what()); }
}

void jnum(std::string& o, float v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.9g", (double)v);
    std::string s(buf);
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";   // rapidjson writes doubles with a fraction
    o += s;
}
void jstr(std::string& o, const std::string& s)
{
    o += '"';
    for (char ch : s) {
        switch (ch) {
        case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case '\n': o += "\\n"; break; case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        default: if ((unsigned char)ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04X", ch); o += b; } else o += ch;
        }
    }
    o += '"';
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C-ABI: world

extern "C" int zr_profab_register(zr_ctx* c, const char* name, uint32_t mesh_id, const zr_material* mat)
{
    if (!c || !name) return ZR_ERR_ARG;
    if (mesh_id >= c->meshes.size()) return zr_fail(c, ZR_ERR_ARG, "bad mesh id");
    ZrProfab pf; pf.mesh = mesh_id;
    int rc = zr_material_prepare(c, mat, &pf.mat);
    if (rc) return rc;
    c->profabs[name].push_back(pf);
    return ZR_OK;
}

extern "C" int zr_world_load_json(zr_ctx* c, const char* utf8, size_t len)
{
    if (!c || !utf8) return ZR_ERR_ARG;
    ZrWorld w; std::string err;
    if (!world_parse(utf8, len, w, err)) return zr_fail(c, ZR_ERR_PARSE, err);
    (void)hipSetDevice(c->device);
    if (zr_sync_all(c) != hipSuccess) return zr_fail(c, ZR_ERR_DEVICE, "world load: device synchronisation failed");
    return apply_world_guarded(c, w);
}

// XkWorld::Save (ZE:1149-1263), PrettyWriter layout (4-space indent).  The engine writes OverrideCubemap from
// EnableSkydome (ZE:1175); that bug is NOT reproduced, so Load(Save(w)) == w.
static std::string world_to_json(const ZrWorld& w)
{
    std::string o;
    auto arr3 = [&](const char* k, const float* v, int n, const char* ind) {
        o += ind; o += "\""; o += k; o += "\": [\n";
        for (int i = 0; i < n; ++i) { o += ind; o += "    "; jnum(o, v[i]); o += (i + 1 < n) ? ",\n" : "\n"; }
        o += ind; o += "]";
    };
    o += "{\n    \"MainCamera\": {\n";
    arr3("Position", w.MainCamera.Position, 3, "        "); o += ",\n";
    arr3("Lookat", w.MainCamera.Lookat, 3, "        "); o += ",\n";
    o += "        \"Speed\": "; jnum(o, w.MainCamera.Speed); o += ",\n        \"FOV\": "; jnum(o, w.MainCamera.FOV);
    o += ",\n        \"zNear\": "; jnum(o, w.MainCamera.zNear); o += ",\n        \"zFar\": "; jnum(o, w.MainCamera.zFar); o += "\n    },\n";
    o += "    \"Skydome\": {\n        \"EnableSkydome\": "; o += w.EnableSkydome ? "true" : "false";
    o += ",\n        \"OverrideSkydome\": "; o += w.OverrideSkydome ? "true" : "false";
    o += ",\n        \"SkydomeFileName\": "; jstr(o, w.SkydomeFileName);
    o += ",\n        \"OverrideCubemap\": "; o += w.OverrideCubemap ? "true" : "false";
    o += ",\n        \"CubemapFileNames\": [\n";
    for (int i = 0; i < 6; ++i) { o += "            "; jstr(o, w.CubemapFileNames[i]); o += i < 5 ? ",\n" : "\n"; }
    o += "        ]\n    },\n";
    o += "    \"Background\": {\n        \"EnableBackground\": "; o += w.EnableBackground ? "true" : "false";
    o += ",\n        \"OverrideBackground\": "; o += w.OverrideBackground ? "true" : "false";
    o += ",\n        \"BackgroundFileName\": "; jstr(o, w.BackgroundFileName); o += "\n    },\n";
    auto lights = [&](const char* key, const std::vector<ZrLightDesc>& ls) {
        o += "    \""; o += key; o += "\": [";
        for (size_t i = 0; i < ls.size(); ++i) {
            const ZrLightDesc& l = ls[i];
            o += i ? ",\n        {\n" : "\n        {\n";
            arr3("Position", l.Position, 3, "            "); o += ",\n            \"Type\": " + std::to_string(l.Type) + ",\n";
            arr3("Color", l.Color, 3, "            "); o += ",\n            \"Intensity\": "; jnum(o, l.Intensity); o += ",\n";
            arr3("Direction", l.Direction, 3, "            "); o += ",\n            \"Radius\": "; jnum(o, l.Radius); o += ",\n";
            arr3("ExtraData", l.ExtraData, 4, "            "); o += "\n        }";
        }
        o += ls.empty() ? "],\n" : "\n    ],\n";
    };
    lights("DirectionalLights", w.DirectionalLights); lights("PointLights", w.PointLights); lights("SpotLights", w.SpotLights);
    o += "    \"Objects\": [";
    for (size_t i = 0; i < w.ObjectDescs.size(); ++i) {
        const ZrObjectDesc& d = w.ObjectDescs[i];
        o += i ? ",\n        {\n" : "\n        {\n";
        o += "            \"RenderFlags\": " + std::to_string(d.RenderFlags & 0xFFFFu) + ",\n            \"ProfabName\": "; jstr(o, d.ProfabName);
        o += ",\n            \"InstanceCount\": " + std::to_string(d.InstanceCount);
        const char* keys[10] = { "MinRadius", "MaxRadius", "MinRotYaw", "MaxRotYaw", "MinRotRoll", "MaxRotRoll", "MinRotPitch", "MaxRotPitch", "MinPScale", "MaxPScale" };
        const float vals[10] = { d.MinRadius, d.MaxRadius, d.MinRotYaw, d.MaxRotYaw, d.MinRotRoll, d.MaxRotRoll, d.MinRotPitch, d.MaxRotPitch, d.MinPScale, d.MaxPScale };
        for (int k = 0; k < 10; ++k) { o += ",\n            \""; o += keys[k]; o += "\": "; jnum(o, vals[k]); }
        o += "\n        }";
    }
    o += w.ObjectDescs.empty() ? "]\n}" : "\n    ]\n}";
    return o;
}

extern "C" int zr_world_save_json(zr_ctx* c, char* dst, size_t cap, size_t* len)
{
    if (!c || !len) return ZR_ERR_ARG;
    const std::string o = world_to_json(c->world);
    *len = o.size();
    if (dst) {
        if (cap < o.size()) return zr_fail(c, ZR_ERR_ARG, "buffer too small");
        memcpy(dst, o.data(), o.size());
    }
    return ZR_OK;
}

// Context-free Load -> Save: parses a livelink payload / World.json with the same strictness as zr_world_load_json and
// writes it back in XkWorld::Save's layout.  Pure host code (no GPU): used to validate scenes before submission.
// On a parse/schema error returns ZR_ERR_PARSE and, if dst is given, the error text.
extern "C" int zr_world_json_normalize(const char* utf8, size_t len_in, char* dst, size_t cap, size_t* len)
{
    if (!utf8 || !len) return ZR_ERR_ARG;
    ZrWorld w; std::string err;
    const bool ok = world_parse(utf8, len_in, w, err);
    const std::string o = ok ? world_to_json(w) : err;
    *len = o.size();
    if (dst) {
        if (cap < o.size()) return ZR_ERR_ARG;
        memcpy(dst, o.data(), o.size());
    }
    return ok ? ZR_OK : ZR_ERR_PARSE;
}

// UpdateWorld + UpdateUniformBuffer (ZE:4294-4308, 4585-4664) from the loaded world's camera and lights: what the engine does every
// frame with its RollStage / RollLight / Time state.
extern "C" int zr_world_update_uniforms(zr_ctx* c, float roll_stage, float roll_light, float time)
{
    if (!c) return ZR_ERR_ARG;
    if (!c->world.loaded) return zr_fail(c, ZR_ERR_STATE, "no world loaded");
    return world_uniforms(c, c->world, roll_stage, roll_light, time);
}

extern "C" int zr_world_get_camera(zr_ctx* c, zr_camera* out)
{
    if (!c || !out) return ZR_ERR_ARG;
    *out = c->world.MainCamera;
    return ZR_OK;
}

extern "C" int zr_object_count(zr_ctx* c, uint32_t* n) { if (!c || !n) return ZR_ERR_ARG; *n = (uint32_t)c->objects.size(); return ZR_OK; }

extern "C" int zr_object_get_instances(zr_ctx* c, uint32_t index, uint32_t* mesh_id, XkInstanceData* dst, uint32_t* n)
{
    if (!c || !n) return ZR_ERR_ARG;
    if (index >= c->objects.size()) return zr_fail(c, ZR_ERR_ARG, "bad object index");
    const ZrSceneObject& o = c->objects[index];
    if (mesh_id) *mesh_id = o.mesh;
    *n = o.instanced ? o.n_inst : 0;
    if (dst && o.instanced) memcpy(dst, o.inst.data(), sizeof(XkInstanceData) * o.n_inst);
    return ZR_OK;
}

// ------------------------------------------------------------------------------------------------ C-ABI: livelink

// One connection: wait (in 100 ms slices, so that zr_livelink_stop is never held up by a silent client) for the first
// segment, then ONE recv (ZE:1683).  A client that sends nothing within XK_LIVELINK_IDLE_MS is dropped.
#define XK_LIVELINK_IDLE_MS 2000
static void livelink_thread(zr_ctx* c)
{
    std::vector<char> buf(XK_LIVELINK_RECV_MAX);
    while (c->ll_run.load()) {
        struct pollfd pfd = { c->ll_listen_fd, POLLIN, 0 };
        const int pr = poll(&pfd, 1, 100);
        if (pr <= 0 || !(pfd.revents & POLLIN)) continue;
        const int cs = accept(c->ll_listen_fd, nullptr, nullptr);
        if (cs < 0) { fprintf(stderr, "[Socket] accept failed\n"); continue; }      // keep listening, ZE:1676-1679
        ssize_t n = -2;                                                                // -2: timed out / stopping
        for (int waited = 0; waited < XK_LIVELINK_IDLE_MS && c->ll_run.load(); waited += 100) {
            struct pollfd cfd = { cs, POLLIN, 0 };
            const int cr = poll(&cfd, 1, 100);
            if (cr < 0) { n = -1; break; }
            if (cr > 0) { n = recv(cs, buf.data(), buf.size(), 0); break; }            // ONE recv per connection, ZE:1683
        }
        if (n > 0) {
            ZrWorld w; std::string err;
            if (world_parse(buf.data(), (size_t)n, w, err)) {
                std::lock_guard<std::mutex> g(c->ll_mutex);
                c->ll_world = std::move(w); c->ll_pending = true;                    // World.bReloadScene = true, ZE:4290
            } else {
                // the engine throws on the socket thread here (ZE:1071-1073) and dies; the library logs and keeps serving
                fprintf(stderr, "%s\n", err.c_str());
            }
        } else if (n == 0) fprintf(stdout, "[Socket] Connection closing...\n");
        else if (n == -2) fprintf(stderr, "[Socket] client sent nothing: dropped\n");
        else fprintf(stderr, "[Socket] recv failed\n");
        shutdown(cs, SHUT_WR);                                                        // no payload is ever sent back, ZE:1699
        close(cs);
    }
}

// The engine binds the wildcard address (AI_PASSIVE, ZE:1630-1636); the payload is unauthenticated, so the library listens on
// loopback unless the host asks for the engine's behaviour with zr_livelink_bind_any(ctx, 1) before zr_livelink_serve.
extern "C" int zr_livelink_bind_any(zr_ctx* c, int any)
{
    if (!c) return ZR_ERR_ARG;
    c->ll_bind_any = any != 0;
    return ZR_OK;
}

extern "C" int zr_livelink_serve(zr_ctx* c, uint16_t port)
{
    if (!c) return ZR_ERR_ARG;
    if (c->ll_run.load()) return zr_fail(c, ZR_ERR_STATE, "livelink already serving");
    const int fd = socket(AF_INET, SOCK_STREAM, IPPROTO_TCP);
    if (fd < 0) return zr_fail(c, ZR_ERR_IO, "[Socket] socket failed");
    int one = 1; setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    struct sockaddr_in a; memset(&a, 0, sizeof a);
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(c->ll_bind_any ? INADDR_ANY : INADDR_LOOPBACK); a.sin_port = htons(port);
    if (bind(fd, (struct sockaddr*)&a, sizeof a) < 0) { close(fd); return zr_fail(c, ZR_ERR_IO, "[Socket] bind failed"); }
    if (listen(fd, SOMAXCONN) < 0) { close(fd); return zr_fail(c, ZR_ERR_IO, "[Socket] listen failed"); }
    socklen_t al = sizeof a;
    if (getsockname(fd, (struct sockaddr*)&a, &al) == 0) c->ll_port = ntohs(a.sin_port);
    c->ll_listen_fd = fd; c->ll_run.store(true);
    c->ll_thread = std::thread(livelink_thread, c);
    return ZR_OK;
}

extern "C" int zr_livelink_port(zr_ctx* c, uint16_t* port) { if (!c || !port) return ZR_ERR_ARG; *port = c->ll_port; return ZR_OK; }

extern "C" int zr_livelink_poll(zr_ctx* c, int* reloaded)
{
    if (!c) return ZR_ERR_ARG;
    if (reloaded) *reloaded = 0;
    ZrWorld w; bool have = false;
    { std::lock_guard<std::mutex> g(c->ll_mutex); if (c->ll_pending) { w = std::move(c->ll_world); c->ll_pending = false; have = true; } }
    if (!have) return ZR_OK;
    (void)hipSetDevice(c->device);
    if (zr_sync_all(c) != hipSuccess) return zr_fail(c, ZR_ERR_DEVICE, "livelink: device synchronisation failed");   // "wait all fences" on both lanes before CreateEngineScene, ZE:1943-1951
    int rc = apply_world_guarded(c, w);
    if (rc == ZR_OK && reloaded) *reloaded = 1;
    return rc;
}

extern "C" int zr_livelink_stop(zr_ctx* c)
{
    if (!c) return ZR_ERR_ARG;
    if (!c->ll_run.load()) return ZR_OK;
    c->ll_run.store(false);
    if (c->ll_thread.joinable()) c->ll_thread.join();
    if (c->ll_listen_fd >= 0) { close(c->ll_listen_fd); c->ll_listen_fd = -1; }
    return ZR_OK;
}
