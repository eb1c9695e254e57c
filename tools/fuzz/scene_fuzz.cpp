#include "../../include/mipt.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <string>
#include <fstream>
#include <iterator>
int main(int argc, char** argv) {
    std::mt19937 rng(11);
    int ok = 0, bad = 0;
    const char* tmp = argv[1];
    for (int f = 2; f < argc; ++f) {
        std::ifstream in(argv[f], std::ios::binary);
        std::vector<uint8_t> src((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        const bool glb = strstr(argv[f], ".glb") != nullptr;
        for (int it = 0; it < 3000; ++it) {
            std::vector<uint8_t> b = src;
            int mode = it % 5;
            if (mode == 0) b.resize(1 + rng() % b.size());
            else if (mode == 1) { int n = 1 + rng() % 4; for (int k = 0; k < n; ++k) b[rng() % b.size()] = (uint8_t)(glb ? rng() : 32 + rng() % 95); }
            else if (mode == 2) { size_t i = rng() % (b.size() - 10); b.erase(b.begin() + i, b.begin() + i + 1 + rng() % 8); }
            else if (mode == 4 && !glb) {
                // structure-aware: rewrite the VALUE of one accessor / bufferView / buffer field (the byte mutations above
                // almost never produce a well-formed file whose accessor type disagrees with how the attribute is read)
                static const char* keys[] = {"\"type\"", "\"count\"", "\"byteOffset\"", "\"byteStride\"", "\"byteLength\"", "\"componentType\"", "\"bufferView\"", "\"buffer\""};
                static const char* vals[] = {"\"SCALAR\"", "\"VEC2\"", "\"VEC3\"", "\"VEC4\"", "\"MAT4\"", "0", "1", "3", "-1", "-4", "2", "65535", "65536", "4294967295",
                                             "4294967296", "1e300", "-1e300", "1.5", "9007199254740993", "5126", "5123", "5125", "5121", "12", "16"};
                std::string t((const char*)b.data(), b.size());
                const std::string key = keys[rng() % 8];
                std::vector<size_t> at;
                for (size_t p = t.find(key); p != std::string::npos; p = t.find(key, p + 1)) at.push_back(p);
                if (!at.empty()) {
                    size_t p = at[rng() % at.size()] + key.size();
                    while (p < t.size() && (t[p] == ' ' || t[p] == ':')) ++p;
                    size_t e = p;
                    if (e < t.size() && t[e] == '"') { ++e; while (e < t.size() && t[e] != '"') ++e; if (e < t.size()) ++e; }
                    else while (e < t.size() && t[e] != ',' && t[e] != '}' && t[e] != ']' && t[e] != '\n') ++e;
                    t.replace(p, e - p, vals[rng() % 25]);
                }
                b.assign(t.begin(), t.end());
            }
            else { size_t i = rng() % b.size(); const char* ins[] = {"-1", "99999999", "[", "}", "\"", "1e999", "null", "{}", "4294967296"}; const char* t = ins[rng() % 9]; b.insert(b.begin() + i, t, t + strlen(t)); }
            std::string path = std::string(tmp) + (glb ? "/m.glb" : "/m.gltf");
            { std::ofstream o(path, std::ios::binary); o.write((const char*)b.data(), (std::streamsize)b.size()); }
            MiptScene* sc = nullptr; char err[256];
            if (mipt_scene_load(path.c_str(), &sc, err, sizeof err) == 0) {
                ++ok;
                // touch the flatten paths
                MiptSceneInfo info;
                if (mipt_scene_info(sc, &info) == 0 && info.num_triangles < 2000000u) {
                    std::vector<float> v((size_t)info.num_triangles * 12 + 4), nrm((size_t)info.num_triangles * 12 + 4), uv((size_t)info.num_triangles * 6 + 4), col((size_t)info.num_triangles * 12 + 4);
                    std::vector<uint32_t> mi(info.num_triangles + 1);
                    std::vector<uint8_t> pr(info.num_triangles + 1);
                    mipt_scene_flatten(sc, v.data(), mi.data());
                    mipt_scene_flatten_normals(sc, nrm.data());
                    mipt_scene_flatten_texcoords(sc, 0, uv.data(), pr.data());
                    mipt_scene_flatten_texcoords(sc, 1, uv.data(), pr.data());
                    mipt_scene_flatten_colors(sc, col.data(), pr.data());
                    for (uint32_t m = 0; m < info.num_materials; ++m) { MiptMaterialData md; mipt_scene_material(sc, m, &md); }
                    for (uint32_t t = 0; t < info.num_textures; ++t) { MiptTextureData td; mipt_scene_texture(sc, t, &td); }
                    float e[3], l[3], u[3], fov, asp; mipt_scene_camera(sc, e, l, u, &fov, &asp);
                }
                mipt_scene_destroy(sc);
            } else ++bad;
        }
    }
    printf("ok %d refused %d\n", ok, bad);
}
