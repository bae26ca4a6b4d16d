// pt_internal.h -- host-side internals shared by the C-ABI translation units.
#pragma once

#include <cmath>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_math.h"

namespace pt {

// Thread-local error message behind pt_last_error_string().
pt_status set_error(pt_status code, const std::string &msg);
void clear_error();
// For the function-try-blocks of the C-ABI entry points: turns the exception in flight into a pt_status.
pt_status exception_status() noexcept;
#define PT_CATCH_ALL catch (...) { return pt::exception_status(); }

// HostScene.h:55-59 HostScene, owning its storage; exposed through the pt_scene_* C ABI.
struct HostSceneImpl {
    std::vector<pt_triangle> triangles;
    std::vector<pt_material> materials;
    std::vector<std::string> material_names;
    std::vector<std::vector<float>> texel_storage;   // float3 texels per texture
    std::vector<pt_texture> textures;                // views into texel_storage
    // what obj_loader.h:65-96 would have made of each material name (lambertian* / metal* / dielectric* /
    // diffuse_light*): records for the pt_set_legacy_scene extension; kind 0 for every other name
    std::vector<pt_legacy_material> legacy;
    void rebuild_texture_views();
};
} // namespace pt
struct pt_host_scene;
namespace pt {
const HostSceneImpl &host_scene_impl(const pt_host_scene *s);

// Reference-topology BVH (bvh.h:11-176) as built on the host, in the reference's node
// numbering, plus the statistics the C ABI reports.
struct RefNode {
    float xmin, xmax, ymin, ymax, zmin, zmax;
    uint32_t left, right, firstTri, triCount;
};
struct RefBvh {
    std::vector<RefNode> nodes;       // nodesUsed entries
    std::vector<int> indices;         // triangle permutation (first n entries of bvh.h:22)
    uint32_t leaves = 0, max_leaf = 0, depth = 0, max_stack = 0;
};
// positions: 9 floats per triangle (v0,v1,v2); centroid computed as triangle.h:27.
void build_reference_bvh(const float *positions9, size_t n_tris, RefBvh &out);

// ------------------------------------------------------------- device-side layout --
// All records are multiples of 16 B so a lane fetches them with dwordx4 loads.
//
// NodePair (64 B), one per INNER reference node, root = pair 0.  A visit of the reference
// traversal (bvh.h:190-243) reads exactly one pair: both children's boxes and what each
// child is.  For child c: cnt == 0 -> inner, link = its pair index; cnt > 0 -> leaf,
// link = first slot in the leaf-ordered triangle array.
struct NodePair {
    float lxmin, lxmax, lymin, lymax;
    float lzmin, lzmax, rxmin, rxmax;
    float rymin, rymax, rzmin, rzmax;
    uint32_t llink, lcnt, rlink, rcnt;
};
// TriIsect (48 B), in leaf order (slot k = reference indices[k]): everything
// triangle::hit needs, with e1, e2 and normalize(cross(e1,e2)) precomputed -- pure
// functions of the vertices, so bit-identical to evaluating them per test
// (triangle.h:66-67, :102).
struct TriIsect {
    float v0x, v0y, v0z, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z, nx, ny, nz;
};
// TriShade (48 B), leaf order: everything the integrator needs from the closest hit, fetched
// once per accepted hit (normal = normalize(cross(e1,e2)), triangle.h:102, precomputed).
struct TriShade {
    float u0, v0, u1, v1, u2, v2;
    int32_t material;
    int32_t face;                     // index in the uploaded triangle order
    float nx, ny, nz, pad;
};
// LightRec (80 B), light-list order (DevicePathTracer.h:304-307).
struct LightRec {
    float v0x, v0y, v0z, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z, nx, ny, nz;
    float v1x, v1y, v1z, v2x;
    float v2y, v2z, area, pad;
};
// MatRec (48 B).
struct MatRec {
    float br, bg, bb; int32_t baseTex;          // baseColorFactor_, baseColorTexture_ (-1 = null)
    float er, eg, eb; int32_t emisTex;          // emissiveFactor_, emissiveTexture_
    float e50r, e50g, e50b; int32_t is_light;   // emissiveFactor_*50 (material.h:85)
};
// Opt-in SAH tree (pt_bvh.cpp): NodePair records in DFS pre-order, leaves of <= 4 triangles (the product builds with 2),
// `order[slot]` = face index of the triangle stored at leaf-order slot `slot`.
struct SahBvh {
    std::vector<NodePair> pairs;
    std::vector<int> order;
    uint32_t max_stack = 0, max_leaf = 0;
};
// max_leaf: ranges of at most this many triangles become leaves (the device forms take up to 15 per leaf).
void build_sah_bvh(const float *positions9, size_t n_tris, SahBvh &out, uint32_t max_leaf = 4);
// The leaf size the product builds with: two.  A triangle trip tests a pair, so every leaf is one trip and one memory round
// trip; the tree has twice the leaves (the 8.4 M-triangle terrain: 180 MB of 64-byte nodes instead of 90, 1.3 s more to build).
// Leaves of 2 / 4 / 6 / 8: C4 33.3 / 34.5 / 36.4 / 40.6 ms, the duck through the SAH tree 96.7 / 100.5 / 105.3 / -- ms, the
// Cornell box through it 56.0 / 68.0 ms (profiles/r04_c4_visit_slope.json).
inline uint32_t sah_leaf_size_for(size_t /*n_tris*/) { return 2u; }

// The SAH tree in 4-wide form for the device (pt_walk.h, wide SAH walk): a QuadNode holds up to four child boxes, made
// from a pair by replacing its larger inner children with their own two children.  128 B = two NodePair-sized halves:
// floats [0..11] the boxes of children 0 and 1 (x/y/z min, max each, as in NodePair), [12..23] children 2 and 3 (second
// half's floats [0..11]), links in the FIRST half's link words [12..15]; the second half's link words are unused.
// (That is the builder's and the checks' form; the device reads it transposed: QuadF / sah_quads_plane_major below.)
// A link is a BYTE offset: of a QuadNode (multiple of 128: low four bits zero), or of a leaf's first TriIsect record
// (multiple of 48, so of 16) with the leaf's triangle count (1..15) in the low four bits.  An absent child has an inverted
// box and is never hit.  Returns false when the tree cannot take this form (a leaf of more than 15 triangles, which the
// builder only makes at its depth limit).  max_stack = the most entries a walk can have pushed at once.
bool collapse_sah_quads(const SahBvh &in, std::vector<NodePair> &quads2, uint32_t &max_stack);

// Moves the first n_top nodes of a breadth-first walk from the root to the front of the array (the rest keep their order)
// and rewrites the links: records [0, n_top) are then the TOP LEVELS of the tree, which the persistent SAH kernel can stage
// in LDS (PTAMD_SAH_TOP; north_star: "top BVH levels ... staged in LDS").  Returns how many it moved (fewer for small trees).
uint32_t sah_quads_top_levels_first(std::vector<NodePair> &quads2, uint32_t n_top);

// The 4-wide form at HALF the bytes, for trees that do not fit the caches (C4: the walk is then bound by the number of
// per-lane 16-byte requests the vector-memory path serves, profiles/r03_c4_memory_path.json): 64 B = four dwordx4 loads
// instead of seven.  The four child boxes are 8-bit offsets from the node's origin in power-of-two units per axis:
//   plane = fmaf((float)byte, 2^e_axis, origin_axis)
// and quantise_sah_quads rounds every low plane DOWN and every high plane UP against exactly that expression, so a
// decoded box always contains the box it stands for (and the device brackets the decoded planes' exact ray parameters from
// outside, quadq_ray below; tests/native/quad_collapse_check.cpp checks both steps): the walk may enter a subtree it could have skipped (+ <1 % triangle
// tests), never the other way round, and the closest hit -- min over (t, rank) of the triangles tested -- cannot change.
struct QuadQ {
    float ox, oy, oz;             // origin: the per-axis minimum over the node's children
    uint32_t exps;                // biased binary exponents of the x | y << 8 | z << 16 units
    uint32_t lo[4];               // per child: xlo | ylo << 8 | zlo << 16 | xhi << 24
    uint32_t hi[4];               // per child: yhi | zhi << 8
    uint32_t link[4];             // as in the 128-byte form with 64-byte node records (inner: byte offset, multiple of 64;
                                  // leaf: byte offset of its first TriIsect | count); 0xFFFFFFFF = no such child
};
static_assert(sizeof(QuadQ) == 64, "QuadQ is one 64-byte record");
// quads2 = collapse_sah_quads' output.  Returns false if some box cannot be represented (non-finite bounds).
bool quantise_sah_quads(const std::vector<NodePair> &quads2, std::vector<QuadQ> &out);
// The plane a byte stands for, one rounding away from origin + byte * unit: what the builder rounds outwards against and
// the CPU checks decode with.  (The device never forms it: quadq_ray below.)
inline float quadq_plane(uint32_t byte, uint32_t biased_exp, float origin)
{
    union { uint32_t u; float f; } s; s.u = biased_exp << 23;
    return __builtin_fmaf((float)byte, s.f, origin);
}

// The 4-wide form as the device reads it when the planes stay floats (RenderParams::sah == 2): plane-major, so that the row
// of the four children's planes a ray meets FIRST on an axis is one 16-byte piece at an offset the sign of its direction
// picks (row 2k for a rising ray, 2k + 1 for a falling one) -- the walk then needs no min / max and no select per child.
struct QuadF {
    float plane[6][4];            // [xlo, xhi, ylo, yhi, zlo, zhi][child]; absent children: +inf on all six
    uint32_t link[4];             // as in the NodePair form of the quads: byte offset of a 128-byte record, or of a TriIsect | count
    uint32_t pad[4];
};
static_assert(sizeof(QuadF) == 2 * sizeof(NodePair), "a QuadF record replaces the two NodePair records of a quad in place");
void sah_quads_plane_major(std::vector<NodePair> &quads2);      // in place, after sah_quads_top_levels_first
// The ray parameters the device computes for a QuadF child (pt_walk.h sahw_node_step): t = fma(plane, inv, offset) with
// offset = -(o * inv) lowered (near planes) / raised (far planes) by 2^-19 of its magnitude; inv clamped as for QuadQ.
inline void quadf_child_span(const QuadF &q, int c, const float o[3], const float inv[3], float &tn, float &tf)
{
    float t_near[3], t_far[3];
    for (int k = 0; k < 3; k++) {
        const float off = -(o[k] * inv[k]);
        const float near_off = __builtin_fmaf(-std::fabs(off), 1.9073486328125e-06f, off), far_off = __builtin_fmaf(std::fabs(off), 1.9073486328125e-06f, off);
        const bool neg = std::signbit(inv[k]);
        t_near[k] = __builtin_fmaf(q.plane[2 * k + (neg ? 1 : 0)][c], inv[k], near_off);
        t_far[k] = __builtin_fmaf(q.plane[2 * k + (neg ? 0 : 1)][c], inv[k], far_off);
    }
    tn = std::fmax(std::fmax(t_near[0], t_near[1]), t_near[2]);
    tf = std::fmin(std::fmin(t_far[0], t_far[1]), t_far[2]);
}

// The ray parameters the device computes for a QuadQ child (pt_walk.h sahq_node_eval / sahq_slab), restated for the CPU
// checks (tests/native/quad_collapse_check.cpp): per visit a slope a = unit * inv and an offset c = (origin - o) * inv per axis,
// the offset lowered for near planes and raised for far planes by 2^-19 of its magnitude (that bounds the two roundings of c);
// per plane t = fma(byte, a, offset).  inv must be clamped to +-2^100 (sahq_begin).
struct QuadqRay { float a[3], near_off[3], far_off[3]; bool neg[3]; };
inline float quadq_clamp_inv(float inv) { const float big = 1.2676506002282294e30f; return std::fmin(std::fmax(inv, -big), big); }
inline QuadqRay quadq_ray(const QuadQ &q, const float o[3], const float inv[3])
{
    QuadqRay r;
    const float org[3] = {q.ox, q.oy, q.oz};
    for (int k = 0; k < 3; k++) {
        union { uint32_t u; float f; } unit; unit.u = ((q.exps >> (8 * k)) & 0xffu) << 23;
        r.a[k] = unit.f * inv[k];
        const float c = (org[k] - o[k]) * inv[k];
        r.near_off[k] = __builtin_fmaf(-std::fabs(c), 1.9073486328125e-06f, c);
        r.far_off[k] = __builtin_fmaf(std::fabs(c), 1.9073486328125e-06f, c);
        r.neg[k] = std::signbit(inv[k]);
    }
    return r;
}
// entry / exit parameters of child `c` (the key of the visit is tn if tf >= max(tn, 0) and tn <= the closest hit so far)
inline void quadq_child_span(const QuadQ &q, int c, const QuadqRay &r, float &tn, float &tf)
{
    const uint32_t lo[3] = {q.lo[c] & 0xffu, (q.lo[c] >> 8) & 0xffu, (q.lo[c] >> 16) & 0xffu};
    const uint32_t hi[3] = {q.lo[c] >> 24, q.hi[c] & 0xffu, (q.hi[c] >> 8) & 0xffu};
    float t_near[3], t_far[3];
    for (int k = 0; k < 3; k++) {
        t_near[k] = __builtin_fmaf((float)(r.neg[k] ? hi[k] : lo[k]), r.a[k], r.near_off[k]);
        t_far[k] = __builtin_fmaf((float)(r.neg[k] ? lo[k] : hi[k]), r.a[k], r.far_off[k]);
    }
    tn = std::fmax(std::fmax(t_near[0], t_near[1]), t_near[2]);
    tf = std::fmin(std::fmin(t_far[0], t_far[1]), t_far[2]);
}

struct TexDesc { int32_t w, h; uint32_t offset /* in floats */, pad; };

} // namespace pt
