// HostScene.h -- HostScene / SceneLoader with the reference's shape (HostScene.h:28-82),
// filled by the C ABI's GLB / OBJ readers instead of assimp + stb_image.
#pragma once

#include <cstddef>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/pt_amd.h"
#include "RendererConfig.h"

namespace ptamd {

// The reference's float2 is 8-byte aligned (CUDA: __builtin_align__(8); HIP: alignas(8)), so its Vertex is 24 bytes with
// texCoords at offset 16 and its Triangle 80 bytes (HostScene.h:28-31, :41-47).  The mirror keeps that TRUE layout, so
// code written against the reference's structs sees the same offsets, and goes to the C ABI through pt_triangle_layout.
struct alignas(8) Float2 { float x = 0.0f, y = 0.0f; };
struct Vertex { Float3 position; Float2 texCoords; };

struct HostMaterial {
    Float3 baseColor;
    std::optional<int> baseColorTextureIdx{};
    Float3 emissiveFactor;
    std::optional<int> emissiveTextureIdx{};
};

struct Triangle {
    Vertex v0, v1, v2;
    int textureIdx;
    int materialIdx;
};

struct HostTexture {
    int width;
    int height;
    std::vector<Float3> data;      // texel channels 0..255
};

struct HostScene {
    std::vector<Triangle> triangles{};
    std::vector<HostTexture> textures{};
    std::vector<HostMaterial> materials{};
    std::vector<std::string> materialNames{};
};

static_assert(sizeof(Vertex) == 24 && offsetof(Vertex, texCoords) == 16 && sizeof(Triangle) == 80 &&
              offsetof(Triangle, v1) == 24 && offsetof(Triangle, textureIdx) == 72,
              "Triangle must have the reference's layout (HostScene.h:28-47 with an 8-byte-aligned float2)");

// The layout record the C ABI reads a std::vector<Triangle> through (include/pt_amd.h: pt_triangle_layout).
inline pt_triangle_layout triangleLayout()
{
    pt_triangle_layout l = {sizeof(Triangle),
                            {offsetof(Triangle, v0) + offsetof(Vertex, position), offsetof(Triangle, v1) + offsetof(Vertex, position),
                             offsetof(Triangle, v2) + offsetof(Vertex, position)},
                            {offsetof(Triangle, v0) + offsetof(Vertex, texCoords), offsetof(Triangle, v1) + offsetof(Vertex, texCoords),
                             offsetof(Triangle, v2) + offsetof(Vertex, texCoords)},
                            offsetof(Triangle, textureIdx), offsetof(Triangle, materialIdx)};
    return l;
}

class SceneLoader {
public:
    // SceneLoader::load(std::string&), HostScene.cpp:98-139: throws std::runtime_error.
    // cornellConvention applies the benchmark convention of SURVEY.md 8d to the loaded scene.
    HostScene load(std::string &path, bool cornellConvention = false)
    {
        size_t dot = path.find_last_of('.');
        if (dot == std::string::npos) throw std::runtime_error("Wrong file name privided: " + path);
        std::string ext = path.substr(dot + 1);
        pt_host_scene *h = nullptr;
        pt_status st;
        if (ext == "glb" || ext == "gltf") st = pt_scene_load_glb(path.c_str(), &h);
        else if (ext == "obj") st = pt_scene_load_obj(path.c_str(), &h);
        else throw std::runtime_error("Unsupported file type" + path);
        if (st != PT_OK) throw std::runtime_error(pt_last_error_string());
        if (cornellConvention) pt_scene_apply_cornell_convention(h);
        HostScene out = fromHandle(h);
        pt_scene_free(h);
        return out;
    }

    static HostScene fromHandle(const pt_host_scene *h)
    {
        HostScene s;
        size_t nt = pt_scene_triangle_count(h), nm = pt_scene_material_count(h), nx = pt_scene_texture_count(h);
        const pt_triangle *t = pt_scene_triangles(h);
        const pt_material *m = pt_scene_materials(h);
        const pt_texture *x = pt_scene_textures(h);
        s.triangles.resize(nt);
        for (size_t i = 0; i < nt; i++) {
            auto cv = [](const pt_vertex &v) { return Vertex{{v.position[0], v.position[1], v.position[2]}, {v.texCoords[0], v.texCoords[1]}}; };
            s.triangles[i] = Triangle{cv(t[i].v0), cv(t[i].v1), cv(t[i].v2), t[i].textureIdx, t[i].materialIdx};
        }
        for (size_t i = 0; i < nm; i++) {
            HostMaterial hm;
            hm.baseColor = {m[i].baseColor[0], m[i].baseColor[1], m[i].baseColor[2]};
            hm.emissiveFactor = {m[i].emissiveFactor[0], m[i].emissiveFactor[1], m[i].emissiveFactor[2]};
            if (m[i].baseColorTextureIdx >= 0) hm.baseColorTextureIdx = m[i].baseColorTextureIdx;
            if (m[i].emissiveTextureIdx >= 0) hm.emissiveTextureIdx = m[i].emissiveTextureIdx;
            s.materials.push_back(hm);
            s.materialNames.emplace_back(pt_scene_material_name(h, i));
        }
        for (size_t i = 0; i < nx; i++) {
            HostTexture ht{x[i].width, x[i].height, {}};
            size_t n = (size_t)x[i].width * (size_t)x[i].height;
            ht.data.resize(n);
            for (size_t k = 0; k < n; k++) ht.data[k] = {x[i].data[3 * k], x[i].data[3 * k + 1], x[i].data[3 * k + 2]};
            s.textures.push_back(std::move(ht));
        }
        return s;
    }
};

} // namespace ptamd
