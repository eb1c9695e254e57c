// scene.cpp -- glTF 2.0 scene loader with the behaviour of sutil::loadScene / sutil::Scene
// (SDK/sutil/Scene.cpp:83-210 node + accessor rules, :267-550 loadScene, :673-688 finalize,
// :777-791 Scene::camera()), host side only.  The reference parses with the vendored tinygltf; that is
// third-party code inside /root/reference, so this file carries its own small JSON / base64 / GLB reader.
//
// What is kept: buffers, accessor -> buffer-view rules (u16/u32/f32 components, byteStride 0 -> tightly
// packed), materials (baseColorFactor, metallic, roughness, emissiveFactor, alphaMode, doubleSided),
// meshes as lists of TRIANGLES primitives (indices, POSITION, material, object aabb from accessor
// min/max), the node recursion `parent * matrix^T * T * R * S`, perspective camera nodes, instances with
// world aabbs, scene aabb, first-camera lookat = aabb centre, default camera.  Textures/images/samplers
// (PNG images are decoded here, png_decode.cpp; sampler state and KHR_texture_transform are kept).  GAS/IAS building is replaced by flattening instances
// to world-space triangles, which mipt_accel_build turns into one BVH.
#include "png_decode.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mipt.h"

namespace {

// ------------------------------------------------------------------------------------------------ JSON
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false; double num = 0.0; std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json* find(const std::string& k) const
    {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& k) const { return find(k) != nullptr; }
    const Json& at(const std::string& k) const
    {
        const Json* j = find(k);
        if (!j) throw std::runtime_error("glTF: missing key '" + k + "'");
        return *j;
    }
    double number(const std::string& k, double dflt) const { const Json* j = find(k); return j && j->kind == Num ? j->num : dflt; }
    int    integer(const std::string& k, int dflt) const { const Json* j = find(k); return j && j->kind == Num ? (int)j->num : dflt; }
    std::string string(const std::string& k, const std::string& dflt) const { const Json* j = find(k); return j && j->kind == Str ? j->str : dflt; }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
};

struct JsonParser {
    const char* p; const char* end;
    [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("glTF JSON: ") + what); }
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    Json parse()
    {
        ws();
        if (p >= end) fail("unexpected end");
        Json j;
        const char c = *p;
        if (c == '{') {
            j.kind = Json::Obj; ++p; ws();
            if (p < end && *p == '}') { ++p; return j; }
            for (;;) {
                ws(); if (p >= end || *p != '"') fail("expected string key");
                std::string k = parse_string();
                ws(); if (p >= end || *p != ':') fail("expected ':'"); ++p;
                Json v = parse();
                j.obj.emplace_back(std::move(k), std::move(v));
                ws(); if (p >= end) fail("unexpected end in object");
                if (*p == ',') { ++p; continue; }
                if (*p == '}') { ++p; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            j.kind = Json::Arr; ++p; ws();
            if (p < end && *p == ']') { ++p; return j; }
            for (;;) {
                j.arr.push_back(parse());
                ws(); if (p >= end) fail("unexpected end in array");
                if (*p == ',') { ++p; continue; }
                if (*p == ']') { ++p; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            j.kind = Json::Str; j.str = parse_string();
        } else if (c == 't' && end - p >= 4 && !std::strncmp(p, "true", 4)) { j.kind = Json::Bool; j.b = true; p += 4; }
        else if (c == 'f' && end - p >= 5 && !std::strncmp(p, "false", 5)) { j.kind = Json::Bool; j.b = false; p += 5; }
        else if (c == 'n' && end - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; }
        else {
            char* e = nullptr;
            j.kind = Json::Num; j.num = std::strtod(p, &e);
            if (e == p) fail("bad token");
            p = e;
        }
        return j;
    }
    std::string parse_string()
    {
        ++p; std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                ++p; if (p >= end) fail("bad escape");
                switch (*p) {
                case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break;
                case 'b': s += '\b'; break; case 'f': s += '\f'; break;
                case 'u': { if (end - p < 5) fail("bad \\u"); unsigned v = (unsigned)std::strtoul(std::string(p + 1, 4).c_str(), nullptr, 16);
                            if (v < 0x80) s += (char)v; else { s += '?'; } p += 4; break; }
                default: s += *p; break;
                }
                ++p;
            } else s += *p++;
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return s;
    }
};

std::vector<uint8_t> base64_decode(const std::string& in, size_t start)
{
    static int8_t tbl[256]; static bool init = false;
    if (!init) {
        std::memset(tbl, -1, sizeof tbl);
        const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) tbl[(unsigned char)a[i]] = (int8_t)i;
        init = true;
    }
    std::vector<uint8_t> out; out.reserve((in.size() - start) * 3 / 4);
    uint32_t acc = 0; int bits = 0;
    for (size_t i = start; i < in.size(); ++i) {
        const int8_t v = tbl[(unsigned char)in[i]];
        if (v < 0) continue;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xffu)); }
    }
    return out;
}

std::vector<uint8_t> read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open '" + path + "'");
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// ------------------------------------------------------------------------------------------------ math (sutil/Matrix.h, Aabb.h, Quaternion.h order of operations)
struct M4 { float m[16]; };
M4 m4_identity() { M4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
M4 m4_mul(const M4& a, const M4& b)                     // Matrix.h:339-355
{
    M4 t;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float sum = 0.0f;
        for (int k = 0; k < 4; ++k) sum += a.m[i * 4 + k] * b.m[k * 4 + j];
        t.m[i * 4 + j] = sum;
    }
    return t;
}
void m4_apply(const M4& a, const float v[4], float out[4])   // Matrix.h:369-384
{
    for (int i = 0; i < 4; ++i) {
        float sum = 0.0f;
        for (int j = 0; j < 4; ++j) sum += a.m[i * 4 + j] * v[j];
        out[i] = sum;
    }
}
M4 m4_transpose(const M4& a) { M4 t; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t.m[i * 4 + j] = a.m[j * 4 + i]; return t; }
M4 m4_translate(float x, float y, float z) { M4 r = m4_identity(); r.m[3] = x; r.m[7] = y; r.m[11] = z; return r; }
M4 m4_scale(float x, float y, float z) { M4 r = m4_identity(); r.m[0] = x; r.m[5] = y; r.m[10] = z; return r; }
M4 m4_rotation(float qw, float qx, float qy, float qz)   // Quaternion.h:239-268
{
    M4 r{};
    r.m[0] = 1.0f - 2.0f * qy * qy - 2.0f * qz * qz; r.m[1] = 2.0f * qx * qy - 2.0f * qz * qw; r.m[2] = 2.0f * qx * qz + 2.0f * qy * qw;
    r.m[4] = 2.0f * qx * qy + 2.0f * qz * qw; r.m[5] = 1.0f - 2.0f * qx * qx - 2.0f * qz * qz; r.m[6] = 2.0f * qy * qz - 2.0f * qx * qw;
    r.m[8] = 2.0f * qx * qz - 2.0f * qy * qw; r.m[9] = 2.0f * qy * qz + 2.0f * qx * qw; r.m[10] = 1.0f - 2.0f * qx * qx - 2.0f * qy * qy;
    r.m[15] = 1.0f;
    return r;
}

struct Aabb {
    float lo[3] = { 1e37f, 1e37f, 1e37f }, hi[3] = { -1e37f, -1e37f, -1e37f };      // Aabb::invalidate
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
    void include(const Aabb& o) { for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], o.lo[k]); hi[k] = std::fmax(hi[k], o.hi[k]); } }
    void transform(const M4& t)                            // Aabb.h:352-390
    {
        Aabb r;
        for (int row = 0; row < 3; ++row) {
            const float* m = t.m + row * 4;
            const float lox = m[0] * lo[0], hix = m[0] * hi[0], loy = m[1] * lo[1], hiy = m[1] * hi[1], loz = m[2] * lo[2], hiz = m[2] * hi[2];
            r.lo[row] = std::fmin(lox, hix) + std::fmin(loy, hiy) + std::fmin(loz, hiz) + m[3];
            r.hi[row] = std::fmax(lox, hix) + std::fmax(loy, hiy) + std::fmax(loz, hiz) + m[3];
        }
        *this = r;
    }
    void center(float c[3]) const { for (int k = 0; k < 3; ++k) c[k] = (lo[k] + hi[k]) * 0.5f; }
    float maxExtent() const { float e = hi[0] - lo[0]; e = std::fmax(e, hi[1] - lo[1]); return std::fmax(e, hi[2] - lo[2]); }
};

// ------------------------------------------------------------------------------------------------ scene
struct View {                     // cuda/BufferView.h:33-49 on host memory
    int buffer = -1; size_t offset = 0; uint32_t count = 0; uint16_t byte_stride = 0, elmt_byte_size = 0, ncomp = 0;
    bool valid() const { return buffer >= 0; }
};
struct Primitive { View indices, positions, normals, texcoords[2], colors; int material = -1; };
struct Texture { uint32_t address_s = 0, address_t = 0, filter = 1; int image = -1; };
struct Mesh { std::string name; std::vector<Primitive> prims; Aabb object_aabb; };
struct Instance { M4 transform; int mesh = -1; Aabb world_aabb; };
struct Cam { float eye[3], lookat[3], up[3], fovY, aspect; };

}  // namespace

struct MiptScene {
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<MiptMaterialData> materials;
    std::vector<Mesh> meshes;
    std::vector<Instance> instances;
    std::vector<Cam> cameras;
    Aabb scene_aabb;
    std::vector<mipt::DecodedImage> images;      // width 0: not decodable here
    std::vector<std::string> image_errors;
    std::vector<Texture> textures;
    uint32_t num_images = 0, num_textures = 0, num_triangles = 0;
};

namespace {

// A glTF number that is used as a byte offset, a count or a stride: finite, >= 0, an integer below 2^53 (what a double holds
// exactly); anything else is rejected before the cast (a negative or huge double -> size_t is undefined behaviour).
size_t checked_size(const Json& o, const char* key, const char* what)
{
    const double d = o.number(key, 0);
    if (!(d >= 0.0) || !(d <= 9007199254740992.0) || d != std::floor(d))
        throw std::runtime_error(std::string("glTF: ") + what + " is not a non-negative integer");
    return (size_t)d;
}

int accessor_components(const std::string& t)
{
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4" || t == "MAT2") return 4;
    if (t == "MAT3") return 9;
    if (t == "MAT4") return 16;
    throw std::runtime_error("glTF: unknown accessor type");
}

// want_comp != 0: the accessor must have exactly that many components (POSITION / NORMAL 3, TEXCOORD 2, COLOR_0 4, indices 1),
// because that is how many the flatten functions read per element.
View view_from_accessor(const Json& root, const MiptScene& sc, int accessor_idx, int want_comp = 0)     // Scene.cpp:83-123
{
    View v;
    if (accessor_idx < 0) return v;
    const Json& accs = root.at("accessors");
    if ((size_t)accessor_idx >= accs.size()) throw std::runtime_error("glTF: accessor index out of range");
    const Json& a = accs.arr[(size_t)accessor_idx];
    const int bv_idx = a.integer("bufferView", -1);
    if (bv_idx < 0 || (size_t)bv_idx >= root.at("bufferViews").size()) throw std::runtime_error("glTF: accessor without bufferView");
    const Json& bv = root.at("bufferViews").arr[(size_t)bv_idx];
    const int ct = a.integer("componentType", 0);
    const int esz = ct == 5123 ? 2 : ct == 5125 ? 4 : ct == 5126 ? 4 : 0;
    if (!esz) throw std::runtime_error("gltf accessor component type not supported");
    const int ncomp = accessor_components(a.string("type", "SCALAR"));
    if (want_comp != 0 && ncomp != want_comp) throw std::runtime_error("glTF: accessor type does not match the attribute it is used for");
    const size_t elem = (size_t)esz * (size_t)ncomp;                       // bytes read per element
    v.buffer = bv.integer("buffer", -1);
    if (v.buffer < 0 || (size_t)v.buffer >= sc.buffers.size()) throw std::runtime_error("glTF: bufferView.buffer out of range");
    const size_t buf_size = sc.buffers[(size_t)v.buffer].size();
    const size_t off_bv = checked_size(bv, "byteOffset", "bufferView.byteOffset"), off_a = checked_size(a, "byteOffset", "accessor.byteOffset");
    const size_t count = checked_size(a, "count", "accessor.count"), stride_in = checked_size(bv, "byteStride", "bufferView.byteStride");
    if (off_bv > buf_size || off_a > buf_size - off_bv) throw std::runtime_error("glTF: accessor starts past the end of its buffer");
    if (count > 0xffffffffull) throw std::runtime_error("glTF: accessor.count does not fit 32 bits");
    if (stride_in >= 65536 || (stride_in != 0 && stride_in < elem)) throw std::runtime_error("glTF: bufferView.byteStride is smaller than an element or >= 65536");
    v.offset = off_bv + off_a;
    v.byte_stride = (uint16_t)(stride_in ? stride_in : elem);
    v.count = (uint32_t)count;
    v.elmt_byte_size = (uint16_t)esz;
    v.ncomp = (uint16_t)ncomp;
    if (count) {
        // offset + (count - 1) * stride + elem <= buf_size, without overflow (count < 2^32, stride < 2^16, offset <= buf_size)
        const size_t room = buf_size - v.offset;
        if (elem > room || (count - 1) > (room - elem) / v.byte_stride) throw std::runtime_error("glTF: accessor reaches past the end of its buffer");
    }
    return v;
}

void process_node(MiptScene& sc, const Json& root, const Json& node, const M4& parent, int depth)     // Scene.cpp:125-210
{
    if (depth > 256) throw std::runtime_error("glTF: node hierarchy too deep (cycle?)");
    auto f = [](const Json& a, size_t i) { return (float)a.arr[i].num; };
    const Json* T = node.find("translation"); const Json* R = node.find("rotation");
    const Json* S = node.find("scale");       const Json* Mx = node.find("matrix");
    const M4 translation = (T && T->size() == 3) ? m4_translate(f(*T, 0), f(*T, 1), f(*T, 2)) : m4_identity();
    const M4 rotation = (R && R->size() == 4) ? m4_rotation(f(*R, 3), f(*R, 0), f(*R, 1), f(*R, 2)) : m4_identity();
    const M4 scale = (S && S->size() == 3) ? m4_scale(f(*S, 0), f(*S, 1), f(*S, 2)) : m4_identity();
    M4 matrix = m4_identity();
    if (Mx && Mx->size() == 16) { M4 raw; for (int i = 0; i < 16; ++i) raw.m[i] = f(*Mx, (size_t)i); matrix = m4_transpose(raw); }
    const M4 xf = m4_mul(m4_mul(m4_mul(m4_mul(parent, matrix), translation), rotation), scale);

    const int cam_idx = node.integer("camera", -1), mesh_idx = node.integer("mesh", -1);
    if (cam_idx >= 0) {
        const Json& cams = root.at("cameras");
        if ((size_t)cam_idx >= cams.size()) throw std::runtime_error("glTF: camera index out of range");
        const Json& gc = cams.arr[(size_t)cam_idx];
        if (gc.string("type", "") != "perspective") return;          // "skipping non-perspective camera": returns, children included (:169-172)
        const float e4[4] = { 0, 0, 0, 1 }, u4[4] = { 0, 1, 0, 0 };
        float eye[4], up[4];
        m4_apply(xf, e4, eye); m4_apply(xf, u4, up);
        Cam c{};
        std::memcpy(c.eye, eye, 12); std::memcpy(c.up, up, 12);
        c.lookat[0] = c.lookat[1] = c.lookat[2] = 0.0f;                // sutil::Camera default lookat (Camera.h:43)
        const Json& persp = gc.at("perspective");
        c.fovY = (float)persp.number("yfov", 0.0) * 180.0f / (float)M_PI;
        c.aspect = (float)persp.number("aspectRatio", 1.0);
        sc.cameras.push_back(c);
    } else if (mesh_idx >= 0) {
        if ((size_t)mesh_idx >= sc.meshes.size()) throw std::runtime_error("glTF: mesh index out of range");
        Instance in; in.transform = xf; in.mesh = mesh_idx; in.world_aabb = sc.meshes[(size_t)mesh_idx].object_aabb;
        in.world_aabb.transform(xf);
        sc.instances.push_back(in);
    }
    if (const Json* ch = node.find("children"))
        for (const Json& c : ch->arr) {
            const int ci = (int)c.num;
            if (ci < 0 || (size_t)ci >= root.at("nodes").size()) throw std::runtime_error("glTF: child index out of range");
            process_node(sc, root, root.at("nodes").arr[(size_t)ci], xf, depth + 1);
        }
}

void load_scene(const std::string& filename, MiptScene& sc)           // Scene.cpp:267-550 + finalize :683-688
{
    std::vector<uint8_t> file = read_file(filename);
    std::string json_text; std::vector<uint8_t> glb_bin;
    const bool is_glb = filename.size() >= 4 && filename.compare(filename.size() - 4, 4, ".glb") == 0;
    if (is_glb) {
        if (file.size() < 20 || std::memcmp(file.data(), "glTF", 4) != 0) throw std::runtime_error("not a GLB file");
        size_t pos = 12;
        while (pos + 8 <= file.size()) {
            uint32_t len, type; std::memcpy(&len, &file[pos], 4); std::memcpy(&type, &file[pos + 4], 4);
            pos += 8;
            if (pos + len > file.size()) throw std::runtime_error("GLB chunk reaches past the end of the file");
            if (type == 0x4E4F534Au) json_text.assign((const char*)&file[pos], len);
            else if (type == 0x004E4942u) glb_bin.assign(file.begin() + (long)pos, file.begin() + (long)(pos + len));
            pos += (len + 3u) & ~3u;
        }
    } else json_text.assign((const char*)file.data(), file.size());
    JsonParser jp{ json_text.data(), json_text.data() + json_text.size() };
    const Json root = jp.parse();
    if (root.kind != Json::Obj) throw std::runtime_error("glTF: top level is not an object");
    const std::string dir = filename.find_last_of("/\\") == std::string::npos ? std::string() : filename.substr(0, filename.find_last_of("/\\") + 1);

    if (const Json* bufs = root.find("buffers"))
        for (const Json& b : bufs->arr) {
            const std::string uri = b.string("uri", "");
            std::vector<uint8_t> data;
            if (uri.empty()) data = glb_bin;
            else if (uri.compare(0, 5, "data:") == 0) {
                const size_t comma = uri.find(',');
                if (comma == std::string::npos) throw std::runtime_error("glTF: bad data URI");
                data = base64_decode(uri, comma + 1);
            } else data = read_file(dir + uri);
            const size_t want = checked_size(b, "byteLength", "buffer.byteLength");
            if (data.size() < want) throw std::runtime_error("glTF: buffer shorter than byteLength");
            data.resize(want);
            sc.buffers.push_back(std::move(data));
        }
    // Images -- all loaded up front (Scene.cpp:303-320).  Undecodable images are kept as empty slots.
    if (const Json* imgs = root.find("images"))
        for (const Json& gi : imgs->arr) {
            std::vector<uint8_t> bytes;
            std::string why;
            try {
                const std::string uri = gi.string("uri", "");
                const int bv_idx = gi.integer("bufferView", -1);
                if (!uri.empty()) {
                    if (uri.compare(0, 5, "data:") == 0) {
                        const size_t comma = uri.find(',');
                        if (comma == std::string::npos) throw std::runtime_error("bad data URI");
                        bytes = base64_decode(uri, comma + 1);
                    } else bytes = read_file(dir + uri);
                } else if (bv_idx >= 0 && (size_t)bv_idx < root.at("bufferViews").size()) {
                    const Json& bv = root.at("bufferViews").arr[(size_t)bv_idx];
                    const int b = bv.integer("buffer", -1);
                    const size_t off = checked_size(bv, "byteOffset", "bufferView.byteOffset"), len = checked_size(bv, "byteLength", "bufferView.byteLength");
                    if (b < 0 || (size_t)b >= sc.buffers.size() || off > sc.buffers[(size_t)b].size() || len > sc.buffers[(size_t)b].size() - off) throw std::runtime_error("image bufferView out of range");
                    bytes.assign(sc.buffers[(size_t)b].begin() + (long)off, sc.buffers[(size_t)b].begin() + (long)(off + len));
                } else throw std::runtime_error("image without uri or bufferView");
            } catch (const std::exception& e) { why = e.what(); }
            mipt::DecodedImage img;
            if (why.empty()) {
                const bool jpeg = bytes.size() >= 2 && bytes[0] == 0xff && bytes[1] == 0xd8;
                const bool ok = jpeg ? mipt::decode_jpeg(bytes.data(), bytes.size(), img, why) : mipt::decode_png(bytes.data(), bytes.size(), img, why);
                if (!ok) img = mipt::DecodedImage();
            }
            sc.images.push_back(std::move(img));
            sc.image_errors.push_back(why);
        }
    sc.num_images = (uint32_t)sc.images.size();
    // Textures -- sampler state + image (Scene.cpp:325-348)
    if (const Json* texs = root.find("textures"))
        for (const Json& gt : texs->arr) {
            Texture t;
            t.image = gt.integer("source", -1);
            const int si = gt.integer("sampler", -1);
            const Json* samplers = root.find("samplers");
            if (si >= 0 && samplers && (size_t)si < samplers->size()) {
                const Json& gs = samplers->arr[(size_t)si];
                const int ws = gs.integer("wrapS", 10497), wt = gs.integer("wrapT", 10497);
                t.address_s = ws == 33071 ? 1u : ws == 33648 ? 2u : 0u;
                t.address_t = wt == 33071 ? 1u : wt == 33648 ? 2u : 0u;
                t.filter = gs.integer("minFilter", -1) == 9728 ? 0u : 1u;
            }
            sc.textures.push_back(t);
        }
    sc.num_textures = (uint32_t)sc.textures.size();
    auto parse_texture_info = [&](const Json* ti, MiptTextureInfo& out) {                  // parseTextureInfo, Scene.cpp:214-265
        out.texture = -1; out.texcoord = 0;
        out.offset[0] = out.offset[1] = 0.0f; out.scale[0] = out.scale[1] = 1.0f; out.rotation[0] = 0.0f; out.rotation[1] = 1.0f;
        if (!ti) return;
        const int idx = ti->integer("index", -1);
        if (idx < 0 || (size_t)idx >= sc.textures.size()) return;
        out.texture = idx;
        out.texcoord = ti->integer("texCoord", 0);
        float rotation = 0.0f;
        if (const Json* ext = ti->find("extensions")) if (const Json* tt = ext->find("KHR_texture_transform")) {
            if (const Json* o = tt->find("offset")) if (o->size() >= 2) { out.offset[0] = (float)o->arr[0].num; out.offset[1] = (float)o->arr[1].num; }
            if (tt->has("rotation")) rotation = (float)tt->number("rotation", 0.0);
            if (const Json* sc2 = tt->find("scale")) if (sc2->size() >= 2) { out.scale[0] = (float)sc2->arr[0].num; out.scale[1] = (float)sc2->arr[1].num; }
        }
        out.rotation[0] = std::sin(rotation); out.rotation[1] = std::cos(rotation);
        if (out.texcoord < 0 || out.texcoord >= 2) out.texcoord = 0;                       // "Maximum supported texcoords exceded"
    };

    if (const Json* mats = root.find("materials"))
        for (const Json& gm : mats->arr) {
            MiptMaterialData m{};                                    // defaults of MaterialData() (cuda/MaterialData.h:44-51)
            m.base_color[0] = m.base_color[1] = m.base_color[2] = m.base_color[3] = 1.0f;
            m.metallic = 1.0f; m.roughness = 1.0f;
            parse_texture_info(nullptr, m.base_color_tex); parse_texture_info(nullptr, m.metallic_roughness_tex);
            parse_texture_info(gm.find("normalTexture"), m.normal_tex); parse_texture_info(gm.find("emissiveTexture"), m.emissive_tex);
            m.double_sided = gm.find("doubleSided") && gm.find("doubleSided")->b ? 1 : 0;
            const std::string am = gm.string("alphaMode", "OPAQUE");
            m.alpha_mode = am == "MASK" ? 1 : am == "BLEND" ? 2 : 0;
            m.alpha_cutoff = am == "MASK" ? (float)gm.number("alphaCutoff", 0.5) : 0.0f;
            if (const Json* pbr = gm.find("pbrMetallicRoughness")) {
                if (const Json* c = pbr->find("baseColorFactor")) if (c->size() == 4) for (int k = 0; k < 4; ++k) m.base_color[k] = (float)c->arr[(size_t)k].num;
                if (pbr->has("roughnessFactor")) m.roughness = (float)pbr->number("roughnessFactor", 1.0);
                if (pbr->has("metallicFactor")) m.metallic = (float)pbr->number("metallicFactor", 1.0);
                parse_texture_info(pbr->find("baseColorTexture"), m.base_color_tex);
                parse_texture_info(pbr->find("metallicRoughnessTexture"), m.metallic_roughness_tex);
            }
            if (const Json* e = gm.find("emissiveFactor")) if (e->size() == 3) for (int k = 0; k < 3; ++k) m.emissive_factor[k] = (float)e->arr[(size_t)k].num;
            sc.materials.push_back(m);
        }

    if (const Json* meshes = root.find("meshes"))
        for (const Json& gmesh : meshes->arr) {
            Mesh mesh; mesh.name = gmesh.string("name", "");
            if (const Json* prims = gmesh.find("primitives"))
                for (const Json& gp : prims->arr) {
                    if (gp.integer("mode", 4) != 4) continue;        // ignore non-triangle primitives (:459-463)
                    Primitive p;
                    p.indices = view_from_accessor(root, sc, gp.integer("indices", -1), 1);
                    p.material = gp.integer("material", -1);
                    const Json& attrs = gp.at("attributes");
                    const int pos_idx = attrs.integer("POSITION", -1);
                    if (pos_idx < 0) throw std::runtime_error("glTF: primitive without POSITION");
                    if ((size_t)pos_idx >= root.at("accessors").size()) throw std::runtime_error("glTF: accessor index out of range");
                    if (root.at("accessors").arr[(size_t)pos_idx].integer("componentType", 0) != 5126) throw std::runtime_error("glTF: POSITION must be float");
                    p.positions = view_from_accessor(root, sc, pos_idx, 3);                          // float VEC3: flatten reads 12 bytes per vertex
                    auto float_view = [&](const char* name, const char* type) {                        // :502-531, float accessors only
                        const int ai = attrs.integer(name, -1);
                        if (ai < 0 || (size_t)ai >= root.at("accessors").size()) return View();
                        const Json& acc = root.at("accessors").arr[(size_t)ai];
                        if (acc.integer("componentType", 0) != 5126 || acc.string("type", "") != type) return View();
                        return view_from_accessor(root, sc, ai, accessor_components(type));
                    };
                    p.normals = float_view("NORMAL", "VEC3");                                          // Scene.cpp:490-498; read as 3 floats
                    p.texcoords[0] = float_view("TEXCOORD_0", "VEC2");
                    p.texcoords[1] = float_view("TEXCOORD_1", "VEC2");
                    p.colors = float_view("COLOR_0", "VEC4");
                    const Json& pa = root.at("accessors").arr[(size_t)pos_idx];
                    const Json* mn = pa.find("min"); const Json* mx = pa.find("max");
                    if (mn && mx && mn->size() >= 3 && mx->size() >= 3) {
                        Aabb bb; for (int k = 0; k < 3; ++k) { bb.lo[k] = (float)mn->arr[(size_t)k].num; bb.hi[k] = (float)mx->arr[(size_t)k].num; }
                        mesh.object_aabb.include(bb);
                    }
                    mesh.prims.push_back(p);
                }
            sc.meshes.push_back(std::move(mesh));
        }

    if (const Json* nodes = root.find("nodes")) {
        std::vector<int> is_root(nodes->size(), 1);
        for (const Json& n : nodes->arr) if (const Json* ch = n.find("children")) for (const Json& c : ch->arr) {
            const int ci = (int)c.num; if (ci >= 0 && (size_t)ci < is_root.size()) is_root[(size_t)ci] = 0;
        }
        for (size_t i = 0; i < nodes->size(); ++i) if (is_root[i]) process_node(sc, root, nodes->arr[i], m4_identity(), 0);
    }
    // Scene::finalize tail (:683-688)
    for (const Instance& in : sc.instances) sc.scene_aabb.include(in.world_aabb);
    if (!sc.cameras.empty() && sc.scene_aabb.valid()) sc.scene_aabb.center(sc.cameras.front().lookat);
    for (const Instance& in : sc.instances)
        for (const Primitive& p : sc.meshes[(size_t)in.mesh].prims)
            sc.num_triangles += (p.indices.valid() ? p.indices.count : p.positions.count) / 3u;
}

}  // namespace

namespace {
// per-corner attribute of the triangles of one mesh, in mipt_scene_mesh_triangles' order, appended at triangle t_out
template <int N>
int mesh_attribute(const MiptScene* sc, size_t mesh, int which, float fill, float* out, uint8_t* present, size_t& t_out)
{
    for (const Primitive& p : sc->meshes[mesh].prims) {
        const View& v = which < 2 ? p.texcoords[which] : p.colors;
        const uint32_t n_tris = (p.indices.valid() ? p.indices.count : p.positions.count) / 3u;
        for (uint32_t t = 0; t < n_tris; ++t) {
            for (int c = 0; c < 3; ++c) {
                float* o = out + (t_out * 3 + (size_t)c) * N;
                for (int k = 0; k < N; ++k) o[k] = fill;
                if (!v.valid()) continue;
                uint32_t vi = 3u * t + (uint32_t)c;
                if (p.indices.valid()) {
                    const uint8_t* ip = sc->buffers[(size_t)p.indices.buffer].data() + p.indices.offset + (size_t)vi * p.indices.byte_stride;
                    if (p.indices.elmt_byte_size == 2) { uint16_t x; std::memcpy(&x, ip, 2); vi = x; } else { std::memcpy(&vi, ip, 4); }
                }
                if (vi >= v.count) return MIPT_ERROR_INVALID_VALUE;
                std::memcpy(o, sc->buffers[(size_t)v.buffer].data() + v.offset + (size_t)vi * v.byte_stride, sizeof(float) * N);
            }
            if (present) present[t_out] = v.valid() ? 1 : 0;
            ++t_out;
        }
    }
    return MIPT_SUCCESS;
}

// ... of every flattened triangle, in mipt_scene_flatten's triangle order
template <int N>
int flatten_attribute(const MiptScene* sc, int which, float fill, float* out, uint8_t* present)
{
    if (!sc || (!out && sc->num_triangles)) return MIPT_ERROR_INVALID_VALUE;
    size_t t_out = 0;
    for (const Instance& in : sc->instances) {
        const int rc = mesh_attribute<N>(sc, (size_t)in.mesh, which, fill, out, present, t_out);
        if (rc) return rc;
    }
    return MIPT_SUCCESS;
}
}  // namespace

extern "C" {

int mipt_scene_load(const char* filename, MiptScene** out, char* errbuf, size_t errlen)
{
    if (!filename || !out) return MIPT_ERROR_INVALID_VALUE;
    *out = nullptr;
    std::unique_ptr<MiptScene> sc(new MiptScene());
    try {
        load_scene(filename, *sc);
    } catch (const std::exception& e) {
        if (errbuf && errlen) { std::snprintf(errbuf, errlen, "%s", e.what()); }
        return MIPT_ERROR_INVALID_VALUE;
    }
    *out = sc.release();
    return MIPT_SUCCESS;
}

int mipt_scene_destroy(MiptScene* sc) { delete sc; return MIPT_SUCCESS; }

int mipt_scene_info(const MiptScene* sc, MiptSceneInfo* info)
{
    if (!sc || !info) return MIPT_ERROR_INVALID_VALUE;
    std::memset(info, 0, sizeof *info);
    info->num_buffers = (uint32_t)sc->buffers.size(); info->num_materials = (uint32_t)sc->materials.size();
    info->num_meshes = (uint32_t)sc->meshes.size(); info->num_instances = (uint32_t)sc->instances.size();
    info->num_cameras = (uint32_t)sc->cameras.size(); info->num_images = sc->num_images; info->num_textures = sc->num_textures;
    info->num_triangles = sc->num_triangles;
    std::memcpy(info->aabb_min, sc->scene_aabb.lo, 12); std::memcpy(info->aabb_max, sc->scene_aabb.hi, 12);
    return MIPT_SUCCESS;
}

// Scene::camera() (Scene.cpp:777-791): the first camera, or the default one in front of the scene
int mipt_scene_camera(const MiptScene* sc, float eye[3], float lookat[3], float up[3], float* fovY_deg, float* aspect)
{
    if (!sc || !eye || !lookat || !up || !fovY_deg || !aspect) return MIPT_ERROR_INVALID_VALUE;
    if (!sc->cameras.empty()) {
        const Cam& c = sc->cameras.front();
        std::memcpy(eye, c.eye, 12); std::memcpy(lookat, c.lookat, 12); std::memcpy(up, c.up, 12);
        *fovY_deg = c.fovY; *aspect = c.aspect;
        return MIPT_SUCCESS;
    }
    float c[3] = { 0, 0, 0 };
    if (sc->scene_aabb.valid()) sc->scene_aabb.center(c);
    std::memcpy(lookat, c, 12);
    eye[0] = c[0] + 0.0f; eye[1] = c[1] + 0.0f; eye[2] = c[2] + 1.5f * (sc->scene_aabb.valid() ? sc->scene_aabb.maxExtent() : 1.0f);
    up[0] = 0.0f; up[1] = 1.0f; up[2] = 0.0f;                        // Camera() default (Camera.h:43)
    *fovY_deg = 45.0f; *aspect = 1.0f;
    return MIPT_SUCCESS;
}

int mipt_scene_material(const MiptScene* sc, uint32_t index, MiptMaterialData* out)
{
    if (!sc || !out || index >= sc->materials.size()) return MIPT_ERROR_INVALID_VALUE;
    *out = sc->materials[index];
    return MIPT_SUCCESS;
}

// world-space triangle soup: 3 float4 {x,y,z,0} per triangle + its material index (-1 -> 0xffffffff)
int mipt_scene_flatten(const MiptScene* sc, float* vertices4_out, uint32_t* material_index_out)
{
    if (!sc || (!vertices4_out && sc->num_triangles)) return MIPT_ERROR_INVALID_VALUE;
    size_t t_out = 0;
    for (const Instance& in : sc->instances)
        for (const Primitive& p : sc->meshes[(size_t)in.mesh].prims) {
            const std::vector<uint8_t>& pb = sc->buffers[(size_t)p.positions.buffer];
            const uint32_t n_tris = (p.indices.valid() ? p.indices.count : p.positions.count) / 3u;
            for (uint32_t t = 0; t < n_tris; ++t, ++t_out) {
                for (int c = 0; c < 3; ++c) {
                    uint32_t vi = 3u * t + (uint32_t)c;
                    if (p.indices.valid()) {
                        const uint8_t* ip = sc->buffers[(size_t)p.indices.buffer].data() + p.indices.offset + (size_t)vi * p.indices.byte_stride;
                        if (p.indices.elmt_byte_size == 2) { uint16_t v; std::memcpy(&v, ip, 2); vi = v; } else std::memcpy(&vi, ip, 4);
                    }
                    if (vi >= p.positions.count) return MIPT_ERROR_INVALID_VALUE;
                    float v[4] = { 0, 0, 0, 1.0f }, w[4];
                    std::memcpy(v, pb.data() + p.positions.offset + (size_t)vi * p.positions.byte_stride, 12);
                    m4_apply(in.transform, v, w);
                    float* o = vertices4_out + (t_out * 3 + (size_t)c) * 4;
                    o[0] = w[0]; o[1] = w[1]; o[2] = w[2]; o[3] = 0.0f;
                }
                if (material_index_out) material_index_out[t_out] = p.material < 0 ? 0xffffffffu : (uint32_t)p.material;
            }
        }
    return MIPT_SUCCESS;
}

// ---- per-mesh object-space geometry + instance records: what Scene::buildMeshAccels / buildInstanceAccel consume
// (Scene.cpp:817-1212).  A mesh's triangles come in the order mipt_scene_flatten emits them inside one instance, so
//   flattened index = (triangles of the instances before) + mesh-local index.
int mipt_scene_mesh_num_triangles(const MiptScene* sc, uint32_t mesh, uint32_t* n_out)
{
    if (!sc || !n_out || mesh >= sc->meshes.size()) return MIPT_ERROR_INVALID_VALUE;
    uint32_t n = 0;
    for (const Primitive& p : sc->meshes[mesh].prims) n += (p.indices.valid() ? p.indices.count : p.positions.count) / 3u;
    *n_out = n;
    return MIPT_SUCCESS;
}

int mipt_scene_mesh_triangles(const MiptScene* sc, uint32_t mesh, float* vertices4_out, uint32_t* material_index_out)
{
    if (!sc || mesh >= sc->meshes.size() || !vertices4_out) return MIPT_ERROR_INVALID_VALUE;
    size_t t_out = 0;
    for (const Primitive& p : sc->meshes[mesh].prims) {
        const std::vector<uint8_t>& pb = sc->buffers[(size_t)p.positions.buffer];
        const uint32_t n_tris = (p.indices.valid() ? p.indices.count : p.positions.count) / 3u;
        for (uint32_t t = 0; t < n_tris; ++t, ++t_out) {
            for (int c = 0; c < 3; ++c) {
                uint32_t vi = 3u * t + (uint32_t)c;
                if (p.indices.valid()) {
                    const uint8_t* ip = sc->buffers[(size_t)p.indices.buffer].data() + p.indices.offset + (size_t)vi * p.indices.byte_stride;
                    if (p.indices.elmt_byte_size == 2) { uint16_t v; std::memcpy(&v, ip, 2); vi = v; } else std::memcpy(&vi, ip, 4);
                }
                if (vi >= p.positions.count) return MIPT_ERROR_INVALID_VALUE;
                float* o = vertices4_out + (t_out * 3 + (size_t)c) * 4;
                std::memcpy(o, pb.data() + p.positions.offset + (size_t)vi * p.positions.byte_stride, 12);
                o[3] = 0.0f;
            }
            if (material_index_out) material_index_out[t_out] = p.material < 0 ? 0xffffffffu : (uint32_t)p.material;
        }
    }
    return MIPT_SUCCESS;
}

// instance `index` (the order of Scene::instances(), = the order mipt_scene_flatten walks): its mesh and the first three rows of
// its row-major object -> world matrix, as Scene::buildInstanceAccel copies them into OptixInstance::transform (Scene.cpp:1153)
int mipt_scene_instance(const MiptScene* sc, uint32_t index, uint32_t* mesh_out, float transform12_out[12])
{
    if (!sc || index >= sc->instances.size()) return MIPT_ERROR_INVALID_VALUE;
    if (mesh_out) *mesh_out = (uint32_t)sc->instances[index].mesh;
    if (transform12_out) std::memcpy(transform12_out, sc->instances[index].transform.m, 48);
    return MIPT_SUCCESS;
}

// World-space shading normals for the Whitted programs (cuda/LocalGeometry.h:114-126): per triangle 3 float4
// {n.xyz, 1} = (M^-T) * object normal, un-normalised; w = 0 and xyz = 0 where the primitive has no NORMAL
// attribute (then the geometric normal is used).  The reference interpolates in object space and transforms
// afterwards (optixTransformNormalFromObjectToWorldSpace); transforming the vertices first is linear-equivalent.
int mipt_scene_flatten_normals(const MiptScene* sc, float* normals4_out)
{
    if (!sc || (!normals4_out && sc->num_triangles)) return MIPT_ERROR_INVALID_VALUE;
    size_t t_out = 0;
    for (const Instance& in : sc->instances) {
        const float* m = in.transform.m;
        // cofactors of the upper 3x3 (= det * inverse transpose), divided by det
        const float c00 = m[5] * m[10] - m[6] * m[9], c01 = m[6] * m[8] - m[4] * m[10], c02 = m[4] * m[9] - m[5] * m[8];
        const float c10 = m[2] * m[9] - m[1] * m[10], c11 = m[0] * m[10] - m[2] * m[8], c12 = m[1] * m[8] - m[0] * m[9];
        const float c20 = m[1] * m[6] - m[2] * m[5], c21 = m[2] * m[4] - m[0] * m[6], c22 = m[0] * m[5] - m[1] * m[4];
        const float det = (m[0] * c00 + m[1] * c01) + m[2] * c02;
        const float inv = det != 0.0f ? 1.0f / det : 0.0f;
        for (const Primitive& p : sc->meshes[(size_t)in.mesh].prims) {
            const uint32_t n_tris = (p.indices.valid() ? p.indices.count : p.positions.count) / 3u;
            for (uint32_t t = 0; t < n_tris; ++t, ++t_out) {
                for (int c = 0; c < 3; ++c) {
                    float* o = normals4_out + (t_out * 3 + (size_t)c) * 4;
                    o[0] = o[1] = o[2] = o[3] = 0.0f;
                    if (!p.normals.valid() || p.normals.elmt_byte_size != 4) continue;
                    uint32_t vi = 3u * t + (uint32_t)c;
                    if (p.indices.valid()) {
                        const uint8_t* ip = sc->buffers[(size_t)p.indices.buffer].data() + p.indices.offset + (size_t)vi * p.indices.byte_stride;
                        if (p.indices.elmt_byte_size == 2) { uint16_t v; std::memcpy(&v, ip, 2); vi = v; } else std::memcpy(&vi, ip, 4);
                    }
                    if (vi >= p.normals.count) return MIPT_ERROR_INVALID_VALUE;
                    float n[3];
                    std::memcpy(n, sc->buffers[(size_t)p.normals.buffer].data() + p.normals.offset + (size_t)vi * p.normals.byte_stride, 12);
                    o[0] = ((c00 * n[0] + c01 * n[1]) + c02 * n[2]) * inv;
                    o[1] = ((c10 * n[0] + c11 * n[1]) + c12 * n[2]) * inv;
                    o[2] = ((c20 * n[0] + c21 * n[1]) + c22 * n[2]) * inv;
                    o[3] = 1.0f;
                }
            }
        }
    }
    return MIPT_SUCCESS;
}

int mipt_scene_texture(const MiptScene* sc, uint32_t index, MiptTextureData* out)
{
    if (!sc || !out || index >= sc->textures.size()) return MIPT_ERROR_INVALID_VALUE;
    const Texture& t = sc->textures[index];
    std::memset(out, 0, sizeof *out);
    out->address_s = t.address_s; out->address_t = t.address_t; out->filter = t.filter; out->image = t.image;
    if (t.image >= 0 && (size_t)t.image < sc->images.size() && sc->images[(size_t)t.image].width) {
        const mipt::DecodedImage& im = sc->images[(size_t)t.image];
        out->width = im.width; out->height = im.height; out->bits = im.bits; out->texels = im.rgba.data();
    }
    return MIPT_SUCCESS;
}

int mipt_scene_flatten_texcoords(const MiptScene* sc, uint32_t set, float* uv2_out, uint8_t* present_out)
{
    if (set >= 2) return MIPT_ERROR_INVALID_VALUE;
    return flatten_attribute<2>(sc, (int)set, 0.0f, uv2_out, present_out);
}

int mipt_scene_flatten_colors(const MiptScene* sc, float* rgba4_out, uint8_t* present_out)
{
    return flatten_attribute<4>(sc, 2, 1.0f, rgba4_out, present_out);
}

// ---- per-mesh OBJECT-space shading attributes (mipt_whitted_set_geometry_instanced): the same values the flatten_* functions hand
// out, once per mesh instead of once per instance -- the triangle order of mipt_scene_mesh_triangles
int mipt_scene_mesh_normals(const MiptScene* sc, uint32_t mesh, float* normals4_out)
{
    if (!sc || mesh >= sc->meshes.size() || !normals4_out) return MIPT_ERROR_INVALID_VALUE;
    size_t t_out = 0;
    for (const Primitive& p : sc->meshes[mesh].prims) {
        const uint32_t n_tris = (p.indices.valid() ? p.indices.count : p.positions.count) / 3u;
        for (uint32_t t = 0; t < n_tris; ++t, ++t_out)
            for (int c = 0; c < 3; ++c) {
                float* o = normals4_out + (t_out * 3 + (size_t)c) * 4;
                o[0] = o[1] = o[2] = o[3] = 0.0f;
                if (!p.normals.valid() || p.normals.elmt_byte_size != 4) continue;
                uint32_t vi = 3u * t + (uint32_t)c;
                if (p.indices.valid()) {
                    const uint8_t* ip = sc->buffers[(size_t)p.indices.buffer].data() + p.indices.offset + (size_t)vi * p.indices.byte_stride;
                    if (p.indices.elmt_byte_size == 2) { uint16_t v; std::memcpy(&v, ip, 2); vi = v; } else std::memcpy(&vi, ip, 4);
                }
                if (vi >= p.normals.count) return MIPT_ERROR_INVALID_VALUE;
                std::memcpy(o, sc->buffers[(size_t)p.normals.buffer].data() + p.normals.offset + (size_t)vi * p.normals.byte_stride, 12);
                o[3] = 1.0f;
            }
    }
    return MIPT_SUCCESS;
}

int mipt_scene_mesh_texcoords(const MiptScene* sc, uint32_t mesh, uint32_t set, float* uv2_out, uint8_t* present_out)
{
    if (!sc || mesh >= sc->meshes.size() || set >= 2 || !uv2_out) return MIPT_ERROR_INVALID_VALUE;
    size_t t_out = 0;
    return mesh_attribute<2>(sc, mesh, (int)set, 0.0f, uv2_out, present_out, t_out);
}

int mipt_scene_mesh_colors(const MiptScene* sc, uint32_t mesh, float* rgba4_out, uint8_t* present_out)
{
    if (!sc || mesh >= sc->meshes.size() || !rgba4_out) return MIPT_ERROR_INVALID_VALUE;
    size_t t_out = 0;
    return mesh_attribute<4>(sc, mesh, 2, 1.0f, rgba4_out, present_out, t_out);
}

}  // extern "C"
