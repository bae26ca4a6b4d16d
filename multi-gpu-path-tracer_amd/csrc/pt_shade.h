// pt_shade.h -- device code of the integrator around one ray segment: LDS staging of the scene
// and shading tables, per-lane path state, sample start, shading of a hit (camera::ray_color,
// camera.h:49-83), pixel quantisation and the pixel <-> lane map.  Included by pt_device.hip only.
#pragma once

#include "pt_walk.h"

namespace pt {
namespace dev {

// BaseColorTexture::value + pixel_data, Texture.h:30-69.  The reference reads row `height`
// (one past the end) when j == 0; that row is clamped to height-1 here.
__device__ __forceinline__ v3 tex_value(const RenderParams &P, int tex, float u, float v)
{
    TexDesc td = P.texs[tex];
    if (td.h <= 0) return mk3((float)242 / 255, (float)45 / 255, (float)27 / 255);
    u = fmodf(u, 1.0f); v = fmodf(v, 1.0f);
    int i = (int)(u * (float)td.w);
    int j = (int)(v * (float)td.h);
    int x = i < 0 ? 0 : (i < td.w ? i : td.w - 1);
    int y = j < 0 ? 0 : (j < td.h ? j : td.h - 1);
    y = td.h - y;
    if (y > td.h - 1) y = td.h - 1;
    const float *px = P.texels + td.offset + 3 * ((size_t)y * (size_t)td.w + (size_t)x);
    double cs = 1.0 / 255.0;
    return mk3((float)(cs * (double)px[0]), (float)(cs * (double)px[1]), (float)(cs * (double)px[2]));
}

struct Onb { v3 a0, a1, a2; };
struct ShadeView;
struct ShadeViewLds;
// onb::onb, onb.h:8-13
__device__ __forceinline__ Onb onb_make(v3 n)
{
    Onb b;
    b.a2 = normalize(n);
    v3 a = fabsf(b.a2.x) > 0.9f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    b.a1 = normalize(cross(b.a2, a));
    b.a0 = cross(b.a2, b.a1);
    return b;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// 64-bit forms for the test counters: 64 lanes x spp x depth x (triangles of a big leaf) passes 2^32
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// A value every lane already agrees on, moved to SGPRs (the compiler cannot prove a shuffle reduction uniform).
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_max64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// Dynamic LDS layout of a workgroup: [staged NodePairs][staged TriIsects] (only when the scene
// is staged) [traversal stacks: stack_entries x blockDim dwords][leaf queues: queue_entries x
// blockDim uint2].  All sections are 16-byte aligned.
struct LdsLayout {
    const float4 *nodes;
    const float4 *tris;
    uint32_t *stack;
    void *queue;
    float4 *tail;             // first float4 after the traversal sections (shading tables)
};

template <bool kLds>
__device__ __forceinline__ LdsLayout lds_carve(const RenderParams &P, float4 *base, int block)
{
    LdsLayout L;
    size_t off = 0;                                  // in float4 units
    L.nodes = base;
    if (kLds) off += (size_t)P.n_pairs * 4;
    L.tris = base + off;
    if (kLds) off += (size_t)P.n_tris * 3;
    L.stack = reinterpret_cast<uint32_t *>(base + off);
    off += ((size_t)P.stack_entries * block + 3) / 4;
    L.queue = reinterpret_cast<void *>(base + off);
    off += ((size_t)P.queue_entries * block * P.queue_dwords + 3) / 4;
    L.tail = base + off;
    return L;
}

// Cooperative copy of the node and triangle arrays into LDS (scene fits: Cornell box 2.4 KB).
template <bool kLds>
__device__ __forceinline__ SceneView<kLds> stage_scene(const RenderParams &P, float4 *base, int tid, int block)
{
    SceneView<kLds> S;
    if (kLds) {
        const float4 *gn = reinterpret_cast<const float4 *>(P.nodes);
        const float4 *gt = reinterpret_cast<const float4 *>(P.tris);
        const int nn = P.n_pairs * 4, nt = P.n_tris * 3;
        for (int i = tid; i < nn; i += block) base[i] = gn[i];
        for (int i = tid; i < nt; i += block) base[nn + i] = gt[i];
        __syncthreads();
        S.nodes = base;
        S.tris = base + nn;
    } else {
        S.nodes = reinterpret_cast<const float4 *>(P.nodes);
        S.tris = reinterpret_cast<const float4 *>(P.tris);
    }
    return S;
}

// Shading tables (TriShade per slot, MatRec, LightRec).  Fetching them from HBM/L2 costs a chain
// of three dependent loads per hit; small scenes keep them in LDS after the traversal sections.
struct ShadeView {
    const float4 *shade;      // 3 x float4 per slot
    const float4 *mats;       // 3 x float4 per material
    const float4 *lights;     // 5 x float4 per light
    float4 *end;              // first float4 of the workgroup's LDS after the staged tables
};

// The same tables for a kernel that ALWAYS stages them (the dealt walk: its scenes are small by construction): the
// pointers carry the LDS address space, so every read is a ds_read_b128.  Through ShadeView's plain pointers -- which may
// point to LDS or to global memory, decided at launch -- the compiler has to emit FLAT loads, which take the vector-memory
// path to reach the LDS (that is why the mask-walk profile shows LDS bank-conflict cycles next to almost no LDS
// instructions: SQ_INSTS_LDS counts ds_* only, the conflicts are those of the flat loads).
typedef float pt_f4_lds __attribute__((ext_vector_type(4)));
struct LdsF4 {
    const __attribute__((address_space(3))) pt_f4_lds *p;
    __device__ __forceinline__ float4 operator[](size_t i) const { const pt_f4_lds v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ LdsF4 operator+(size_t k) const { LdsF4 r; r.p = p + k; return r; }
};
struct ShadeViewLds {
    LdsF4 shade, mats, lights;
    LdsF4 onb;                // 3 x float4 per slot: the orthonormal basis of the triangle's normal (onb.h:8-13), precomputed
    float4 *end;
};

__device__ __forceinline__ ShadeViewLds stage_shade_lds(const RenderParams &P, float4 *lds, int tid, int block)
{
    const float4 *gs = reinterpret_cast<const float4 *>(P.shade);
    const float4 *gm = reinterpret_cast<const float4 *>(P.mats);
    const float4 *gl = reinterpret_cast<const float4 *>(P.lights);
    const float4 *go = reinterpret_cast<const float4 *>(P.onb);
    const int ns = P.n_tris * 3, nm = P.n_mats * 3, nl = P.n_lights * 5;
    for (int i = tid; i < ns; i += block) lds[i] = gs[i];
    for (int i = tid; i < nm; i += block) lds[ns + i] = gm[i];
    for (int i = tid; i < nl; i += block) lds[ns + nm + i] = gl[i];
    for (int i = tid; i < ns; i += block) lds[ns + nm + nl + i] = go[i];
    __syncthreads();
    ShadeViewLds V;
    V.shade.p = (const __attribute__((address_space(3))) pt_f4_lds *)lds;
    V.mats = V.shade + (size_t)ns;
    V.lights = V.mats + (size_t)nm;
    V.onb = V.lights + (size_t)nl;
    V.end = lds + 2 * ns + nm + nl;
    return V;
}

__device__ __forceinline__ ShadeView stage_shade(const RenderParams &P, float4 *lds, int tid, int block)
{
    // shade_lds = 1: all three tables in LDS; 2: only the material and light tables (a scene of millions of triangles
    // still has a handful of materials and lights: the hit's TriShade record comes from memory, the two lookups that
    // depend on it do not pay a second and third round trip).
    ShadeView V;
    const float4 *gs = reinterpret_cast<const float4 *>(P.shade);
    const float4 *gm = reinterpret_cast<const float4 *>(P.mats);
    const float4 *gl = reinterpret_cast<const float4 *>(P.lights);
    if (P.shade_lds) {
        const int ns = P.shade_lds == 1 ? P.n_tris * 3 : 0, nm = P.n_mats * 3, nl = P.n_lights * 5;
        for (int i = tid; i < ns; i += block) lds[i] = gs[i];
        for (int i = tid; i < nm; i += block) lds[ns + i] = gm[i];
        for (int i = tid; i < nl; i += block) lds[ns + nm + i] = gl[i];
        __syncthreads();
        V.shade = ns ? lds : gs; V.mats = lds + ns; V.lights = lds + ns + nm;
        V.end = lds + ns + nm + nl;
    } else {
        V.end = lds;
        V.shade = gs;
        V.mats = gm;
        V.lights = gl;
    }
    return V;
}

// LDS of the dealt walk (kWalk 4, pt_walk.h): the workgroup's copy of the TriIsect table, then per wave
// {64 x float4 ray_a, 64 x u64 keys, 64 x float2 ray_b, deal_cap x u16 items}.  Mirrored by lds_bytes() on the host.
__device__ __forceinline__ DealView deal_carve(const RenderParams &P, float4 *lds, int tid, int block)
{
    DealView D;
    const float4 *gt = reinterpret_cast<const float4 *>(P.tris);
    const int nt = P.n_tris * 3;
    for (int i = tid; i < nt; i += block) lds[i] = gt[i];
    const int wave = tid / kWave;
    const size_t per_wave_f4 = (size_t)kWave + kWave / 2 + kWave / 2 + ((size_t)P.deal_cap * 2 + 15) / 16;
    D.boxes = nullptr; D.reach = nullptr; D.ray_c = nullptr;
    if (P.part_boxes) {
        // For the parts of split tiles (part_walk_by_lanes): behind the waves' areas the workgroup's child boxes, two float4
        // each, the reach table, and 32 reciprocal directions per wave.
        float4 *bx = lds + nt + (size_t)(block / kWave) * per_wave_f4;
        uint4 *rt = reinterpret_cast<uint4 *>(bx + 4 * (size_t)P.n_pairs);
        const float4 *gn = reinterpret_cast<const float4 *>(P.nodes);
        for (int p = tid; p < P.n_pairs; p += block) {
            const float4 a = gn[4 * p], b = gn[4 * p + 1], c = gn[4 * p + 2];
            bx[4 * p] = a;                                  bx[4 * p + 1] = make_float4(b.x, b.y, 0.0f, 0.0f);
            bx[4 * p + 2] = make_float4(b.z, b.w, c.x, c.y); bx[4 * p + 3] = make_float4(c.z, c.w, 0.0f, 0.0f);
        }
        if (tid == 0) {
            // reach[b].x = the boxes a ray must hit to enter child box b = 2 * pair + side: b itself and what its pair needs.
            // Pairs are numbered in visit order, parents first: a pair's own need is left in the .w of its left box by its parent.
            for (int p = 0; p < P.n_pairs; p++) {
                const uint32_t pair_need = p == 0 ? 0u : rt[2 * p].w;
                const uint32_t link[2] = {P.nodes[p].llink, P.nodes[p].rlink}, cnt[2] = {P.nodes[p].lcnt, P.nodes[p].rcnt};
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    const int b = 2 * p + side;
                    const uint32_t need = pair_need | (1u << b);
                    rt[b].x = need; rt[b].y = link[side]; rt[b].z = cnt[side];
                    if (cnt[side] == 0) rt[2 * link[side]].w = need;
                }
            }
        }
        D.boxes = bx;
        D.reach = rt;
        // (the wave index through readfirstlane: a scalar base, so that no per-lane address has to live across the trips)
        D.ray_c = reinterpret_cast<float4 *>(rt + 2 * (size_t)P.n_pairs) + (size_t)__builtin_amdgcn_readfirstlane(wave) * 32;
    }
    __syncthreads();
    D.tris = lds;
    float4 *w = lds + nt + (size_t)wave * per_wave_f4;
    D.ray_a = w;
    D.keys = reinterpret_cast<unsigned long long *>(w + kWave);
    D.ray_b = reinterpret_cast<float2 *>(w + kWave + kWave / 2);
    D.items = reinterpret_cast<uint16_t *>(w + kWave + kWave);
    D.cap = P.deal_cap;
    D.deal_min = P.deal_min;
    D.recompute_uv = P.any_tex;
    return D;
}

// LDS of the stack walk with dealt leaves (kWalk 5): per wave {64 x float4 ray_a, 64 x u64 keys, 64 x float2 ray_b,
// deal_cap x u32 items}; the triangle table stays in memory.  Mirrored by lds_bytes() on the host.
__device__ __forceinline__ DealView32 deal_carve32(const RenderParams &P, float4 *lds, int tid)
{
    DealView32 D;
    const int wave = tid / kWave;
    const size_t items_f4 = ((size_t)P.deal_cap * 4 + 15) / 16;
    const size_t per_wave_f4 = (size_t)kWave + kWave / 2 + kWave / 2 + items_f4 + 3 * (size_t)kStageTris;
    float4 *w = lds + (size_t)wave * per_wave_f4;
    D.ray_a = w;
    D.keys = reinterpret_cast<unsigned long long *>(w + kWave);
    D.ray_b = reinterpret_cast<float2 *>(w + kWave + kWave / 2);
    D.items = reinterpret_cast<uint32_t *>(w + kWave + kWave);
    D.stage = w + kWave + kWave + items_f4;
    D.cap = P.deal_cap;
    D.deal_min = P.deal_min;
    return D;
}

// render, DevicePathTracer.h:73-120, with render_init (:46-55) folded in and
// camera::ray_color (camera.h:49-83) flattened into a per-lane state machine.
// Per-lane integrator state of camera::ray_color (camera.h:49-83) and of render's sample loop
// (DevicePathTracer.h:80-87), plus this lane's exact work counters.
struct PathState {
    xorwow rng;
    v3 col, atten, ro, rd;
    int s;                 // samples started
    unsigned bounce;       // segments traced on the current path
    bool need_new;         // the next trip starts a new sample
    uint32_t n_paths, n_seg, n_box, n_tri, n_light;
    // per-lane walks only: how many times 2^fold_bits were taken out of n_box / n_tri.  A lane's tests grow as
    // spp x depth x tests per segment (reference tree on the 8 M-triangle terrain: 2.5e6 per segment), which
    // passes 2^32 at C5-class sample counts; the wave-uniform walks keep 64-bit wave totals instead.
    uint32_t box_hi, tri_hi;
};

__device__ __forceinline__ void fold_counters(PathState &L, int bits)
{
    if (L.n_box >> bits) { L.n_box -= 1u << bits; L.box_hi++; }
    if (L.n_tri >> bits) { L.n_tri -= 1u << bits; L.tri_hi++; }
}

__device__ __forceinline__ void path_init(PathState &L, const RenderParams &P, int pixel_index)
{
    xorwow_init(L.rng, (uint64_t)(long long)(1984 + pixel_index));   // :54 seed, :80 state copy
    L.col = mk3(0.0f, 0.0f, 0.0f);
    L.atten = mk3(1.0f, 1.0f, 1.0f);
    L.ro = P.cam.origin; L.rd = mk3(0.0f, 0.0f, 0.0f);
    L.s = 0; L.bounce = 0; L.need_new = true;
    L.n_paths = L.n_seg = L.n_box = L.n_tri = L.n_light = 0;
    L.box_hi = L.tri_hi = 0;
}

// Starts the next sample of pixel (x, y): DevicePathTracer.h:84-86 + camera::get_ray camera.h:95-97.
// Returns false when all samples are done.
__device__ __forceinline__ bool start_path(PathState &L, const RenderParams &P, int x, int y)
{
    if (L.s >= P.spp) return false;
    float u = ((float)x + xorwow_uniform(L.rng)) / (float)(unsigned)P.W;
    float v = ((float)y + xorwow_uniform(L.rng)) / (float)(unsigned)P.H;
    L.ro = P.cam.origin;
    L.rd = ((P.cam.llc + u * P.cam.horizontal) + v * P.cam.vertical) - P.cam.origin;
    L.atten = mk3(1.0f, 1.0f, 1.0f);
    L.bounce = 0;
    L.s++;
    L.n_paths++;
    L.need_new = false;
    return true;
}

// ---- extension: the reference's legacy material classes and sphere ----------------------
// Off unless pt_set_legacy_scene was called; compiled only into render_kernel<..., kLegacy = true>.  No
// integrator of the reference at HEAD calls these classes, so the loop around them (shade<true>) is this
// project's choice -- the classic scatter / emitted loop their signatures were written for -- and, like
// the rest, parity-unpinned; the kernel and oracle/pt_oracle.c agree bit for bit.

// sphere::hit, sphere.h:21-50 (the double promotions there do not change a bit, see the oracle).
__device__ __forceinline__ bool sphere_hit(float4 s, v3 o, v3 d, float tmin, float tmax, float &t_out)
{
    const v3 center = mk3(s.x, s.y, s.z);
    const v3 oc = o - center;
    const float a = dot(d, d);
    const float b = 2.0f * dot(oc, d);
    const float cc = dot(oc, oc) - s.w * s.w;
    const float discriminant = b * b - 4 * a * cc;
    if (discriminant > 0) {
        const float root = sqrtf(discriminant);
        float temp = (-b - root) / (2.0f * a);
        if (temp < tmax && temp > tmin) { t_out = temp; return true; }
        temp = (-b + root) / (2.0f * a);
        if (temp < tmax && temp > tmin) { t_out = temp; return true; }
    }
    return false;
}

// Spheres are tested after the BVH like members of a hitable_list: the closest hit so far shrinks the
// interval.  A sphere hit is reported as slot = -2 - index.
__device__ __forceinline__ void spheres_hit(const RenderParams &P, v3 o, v3 d, float tmin, Hit &h)
{
    float closest = h.slot >= 0 ? h.t : 3.402823466e+38f;
    for (int i = 0; i < P.n_spheres; i++) {
        float t;
        if (sphere_hit(P.spheres[2 * i], o, d, tmin, closest, t)) { h.t = t; h.u = h.v = 0.0f; h.slot = -2 - i; closest = t; }
    }
}

__device__ __forceinline__ v3 random_in_unit_sphere(xorwow &rng)            // helper_math.h:1511-1518
{
    v3 p;
    do {
        const float x = xorwow_uniform(rng), y = xorwow_uniform(rng), z = xorwow_uniform(rng);
        p = mk3(x, y, z) * 2.0f - mk3(1.0f, 1.0f, 1.0f);
    } while (dot(p, p) >= 1.0f);
    return p;
}
__device__ __forceinline__ v3 reflect(v3 i, v3 n) { return i - (n * 2.0f) * dot(n, i); }   // :1429-1432

// material::scatter of lambertian / metal / dielectric / diffuse_light (material.h:110-217).
// False = the path ends at this hit.
__device__ __forceinline__ bool legacy_scatter(PathState &L, int kind, v3 albedo, float param, v3 normal, v3 p,
                                               v3 &attenuation, v3 &out_dir)
{
    if (kind == 1) {                                                     // lambertian
        v3 dir = normal + random_in_unit_sphere(L.rng);
        if (fabsf(dir.x) <= 1e-8f && fabsf(dir.y) <= 1e-8f && fabsf(dir.z) <= 1e-8f) dir = normal;   // near_zero
        out_dir = dir; attenuation = albedo;
        return true;
    }
    if (kind == 2) {                                                     // metal, fuzz clamped by the ctor
        const float fuzz = param < 1 ? param : 1;
        const v3 reflected = reflect(normalize(L.rd), normal);
        out_dir = reflected + random_in_unit_sphere(L.rng) * fuzz;
        attenuation = albedo;
        return dot(out_dir, normal) > 0;
    }
    if (kind == 3) {                                                     // dielectric
        const float ir = param;
        const v3 reflected = reflect(L.rd, normal);
        v3 outward, refracted = mk3(0.0f, 0.0f, 0.0f);
        float ni_over_nt, reflect_prob, cosine;
        attenuation = mk3(1.0f, 1.0f, 1.0f);
        if (dot(L.rd, normal) > 0.0f) {
            outward = mk3(-normal.x, -normal.y, -normal.z);
            ni_over_nt = ir;
            cosine = dot(L.rd, normal) / length(L.rd);
            cosine = sqrtf(1.0f - ir * ir * (1 - cosine * cosine));
        } else {
            outward = normal;
            ni_over_nt = 1.0f / ir;
            cosine = -dot(L.rd, normal) / length(L.rd);
        }
        bool can_refract = false;
        {                                                                // refract, helper_math.cu:7-17
            const v3 uv = normalize(L.rd);
            const float dt = dot(uv, outward);
            const float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
            if (discriminant > 0) {
                refracted = (uv - outward * dt) * ni_over_nt - outward * sqrtf(discriminant);
                can_refract = true;
            }
        }
        if (can_refract) {                                               // schlick_approx, material.h:10-14; x^5 = ((x*x)*(x*x))*x
            float r0 = (1 - ir) / (1 + ir);
            r0 = r0 * r0;
            const float x = 1 - cosine, x2 = x * x;
            reflect_prob = r0 + (1 - r0) * ((x2 * x2) * x);
        } else reflect_prob = 1.0f;
        out_dir = (xorwow_uniform(L.rng) < reflect_prob) ? reflected : refracted;
        return true;
    }
    return false;                                                        // diffuse_light: material::scatter
}

// onb(rec.normal) of a hit: computed on the spot, or -- where the view carries the per-triangle table (the normal of a
// triangle hit is the triangle's precomputed normal, so its basis is a pure function of the triangle too; the host
// evaluates the same onb_make with the IEEE forms the device's short forms are proven equal to) -- read from LDS.
__device__ __forceinline__ Onb hit_onb(const ShadeView &, v3 normal, int) { return onb_make(normal); }
__device__ __forceinline__ Onb hit_onb(const ShadeViewLds &V, v3 normal, int slot)
{
    if (slot < 0) return onb_make(normal);                          // a sphere of the legacy extension (never in this view)
    const auto q = V.onb + 3 * (size_t)slot;
    const float4 a = q[0], b = q[1], c = q[2];
    Onb o;
    o.a0 = mk3(a.x, a.y, a.z); o.a1 = mk3(a.w, b.x, b.y); o.a2 = mk3(b.z, b.w, c.x);
    return o;
}

// What camera::ray_color does with the result of one BVH::hit (camera.h:55-80): miss, emissive
// hit, or a scattering bounce that sets up the next ray segment.
template <bool kLegacy = false, class View = ShadeView>
__device__ __forceinline__ void shade(PathState &L, const RenderParams &P, const View &V, const Hit &h)
{
    if (h.slot == -1) {
        L.col = L.col + mk3(0.0f, 0.0f, 0.0f) * L.atten;                // camera.h:78-79
        L.need_new = true;
        return;
    }
    // hit_record completion, triangle.h:100-108 (sphere.h:33-36 for a sphere of the extension)
    v3 normal;
    v3 p = L.ro + h.t * L.rd;
    float4 s0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), s1 = s0;
    int mat;
    if (kLegacy && h.slot < -1) {
        const int si = -2 - h.slot;
        const float4 sph = P.spheres[2 * si];
        normal = (p - mk3(sph.x, sph.y, sph.z)) / sph.w;
        mat = __float_as_int(P.spheres[2 * si + 1].x);
    } else {
        const float4 tc = V.shade[3 * (size_t)h.slot + 2];
        normal = mk3(tc.x, tc.y, tc.z);
        const auto sp4 = V.shade + 3 * (size_t)h.slot;
        s0 = sp4[0]; s1 = sp4[1];
        mat = __float_as_int(s1.z);
    }
    if (kLegacy) {
        const float4 l0 = P.legacy_mats[2 * (size_t)mat];
        const int kind = __float_as_int(P.legacy_mats[2 * (size_t)mat + 1].x);
        if (kind != 0) {
            // end on emitters and absorbed rays, otherwise multiply the throughput and follow the scattered ray
            v3 attenuation = mk3(0.0f, 0.0f, 0.0f), dir = L.rd;
            const v3 albedo = mk3(l0.x, l0.y, l0.z);
            if (!legacy_scatter(L, kind, albedo, l0.w, normal, p, attenuation, dir)) {
                const v3 emitted = kind == 4 ? albedo : mk3(0.0f, 0.0f, 0.0f);
                L.col = L.col + L.atten * emitted;
                L.need_new = true;
                return;
            }
            L.atten = L.atten * attenuation;
            L.ro = p; L.rd = dir;
            L.bounce++;
            if (L.bounce >= P.depth) {
                L.col = L.col + mk3(0.0f, 0.0f, 0.0f);
                L.need_new = true;
            }
            return;
        }
    }
    const auto mp = V.mats + 3 * (size_t)mat;
    float4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
    int baseTex = __float_as_int(m0.w), emisTex = __float_as_int(m1.w);
    float tu = 0.0f, tv = 0.0f;
    if (baseTex >= 0 || emisTex >= 0) {
        tu = (1 - h.u - h.v) * s0.x + h.u * s0.z + h.v * s1.x;
        tv = (1 - h.u - h.v) * s0.y + h.u * s0.w + h.v * s1.y;
    }
    // UniversalMaterial::emitted, material.h:80-86
    v3 emitted = mk3(m2.x, m2.y, m2.z);
    if (emisTex >= 0) emitted = (tex_value(P, emisTex, tu, tv) * mk3(m1.x, m1.y, m1.z)) * 50.0f;
    if (emitted.x > 0.0001f || emitted.y > 0.0001f || emitted.z > 0.0001f) {   // material.h:63
        L.col = L.col + L.atten * emitted;                               // camera.h:72-74
        L.need_new = true;
        return;
    }
    // material.h:67-68: a cosine sample that camera.h:65 overwrites; its 2 draws remain.
    (void)xorwow_next(L.rng); (void)xorwow_next(L.rng);
    v3 albedo = mk3(m0.x, m0.y, m0.z);
    if (baseTex >= 0) albedo = albedo * tex_value(P, baseTex, tu, tv);
    const Onb uvw = hit_onb(V, normal, h.slot);                    // cosine_pdf ctor, pdf.h:15-17
    v3 dir;
    if (xorwow_uniform(L.rng) < 0.5f) {                              // mixture_pdf::generate, pdf.h:66-72
        float uu = xorwow_uniform(L.rng);                            // hitable_list::random, hitable_list.h:23-26
        int idx = (int)truncf((float)((double)uu * P.light_index_scale));
        const auto lp = V.lights + 5 * (size_t)idx;
        float4 la = lp[0], ld = lp[3], le = lp[4];
        float r1 = xorwow_uniform(L.rng), r2 = xorwow_uniform(L.rng);  // triangle::random, triangle.h:41-47
        float sq = pt_sqrt(r1);
        v3 lv0 = mk3(la.x, la.y, la.z), lv1 = mk3(ld.x, ld.y, ld.z), lv2 = mk3(ld.w, le.x, le.y);
        v3 rp = ((1 - sq) * lv0 + (sq * (1 - r2)) * lv1) + (sq * r2) * lv2;
        dir = rp - p;
    } else {
        float r1 = xorwow_uniform(L.rng), r2 = xorwow_uniform(L.rng);  // random_cosine_direction, helper_math.h:1519-1527
        float z = pt_sqrt(1 - r2);
        float phi = (float)(2 * M_PI * (double)r1);
        float sn, cs;
        pt_sincosf(phi, sn, cs);
        float sq2 = pt_sqrt(r2);
        float lx = cs * 2 * sq2, ly = sn * 2 * sq2;
        dir = (lx * uvw.a0 + ly * uvw.a1) + z * uvw.a2;            // onb::local, onb.h:20-22
    }
    // hitable_list::pdf_value, hitable_list.h:16-22 -> triangle::pdf_value, triangle.h:32-40
    float lsum = 0.0f;
    for (int k = 0; k < P.n_lights; k++) {
        const auto lp = V.lights + 5 * (size_t)k;
        float4 la = lp[0], lb = lp[1], lc = lp[2], le = lp[4];
        v3 v0 = mk3(la.x, la.y, la.z), e1 = mk3(la.w, lb.x, lb.y), e2 = mk3(lb.z, lb.w, lc.x);
        v3 ln = mk3(lc.y, lc.z, lc.w);
        float t, u, v;
        L.n_light++;
        float pdfv = 0.0f;
        if (tri_hit(v0, e1, e2, p, dir, 0.001f, 3.402823466e+38f, t, u, v)) {
            float distance_squared = t * t * dot(dir, dir);
            float cosine = fabsf(dot(dir, ln) / length(dir));
            pdfv = distance_squared / (cosine * le.z);
        }
        lsum += P.light_weight * pdfv;
    }
    v3 ndir = normalize(dir);
    float cosv = dot(ndir, uvw.a2);                                // cosine_pdf::value, pdf.h:19-22
    float cpdf = (cosv <= 0) ? 0.0f : pt_div_pi(cosv);
    float pdf_value = 0.5f * lsum + 0.5f * cpdf;                   // mixture_pdf::value, pdf.h:63-65
    float cos2 = dot(normal, ndir);                                // scattering_pdf, material.h:88-91
    float spdf = cos2 < 0 ? 0.0f : pt_div_pi(cos2);
    L.atten = L.atten * ((albedo * spdf) / pdf_value);                 // camera.h:69
    L.ro = p; L.rd = dir;
    L.bounce++;
    if (L.bounce >= P.depth) {
        L.col = L.col + mk3(0.0f, 0.0f, 0.0f);                         // camera.h:82
        L.need_new = true;
    }
}

// Quantisation + RGB / YUV420 store of one pixel, DevicePathTracer.h:98-119.
template <class Params>
__device__ __forceinline__ void write_pixel(Params &P, v3 col, int pixel_index)
{
{
    // :98-119
    v3 q = (255.99f * col) / (float)P.spp;
    int cr = f2i_rz_sat(q.x * 1.0f), cg = f2i_rz_sat(q.y * 1.0f), cb = f2i_rz_sat(q.z * 1.0f);
    cr = min(255, cr); cg = min(255, cg); cb = min(255, cb);
    P.rgb[3 * (size_t)pixel_index] = (uint8_t)cr;
    P.rgb[3 * (size_t)pixel_index + 1] = (uint8_t)cg;
    P.rgb[3 * (size_t)pixel_index + 2] = (uint8_t)cb;
    P.yuv[pixel_index] = (uint8_t)(((66 * cr + 129 * cg + 25 * cb + 128) >> 8) + 16);
    int blockRow = pixel_index / P.W, blockCol = pixel_index % P.W;
    if (blockRow % 2 == 0 && blockCol % 2 == 0) {
        int total = P.W * P.H, uvSize = total / 4;
        int uvIndex = (blockRow / 2) * (P.W / 2) + (blockCol / 2);
        P.yuv[total + uvIndex] = (uint8_t)(((-38 * cr - 74 * cg + 112 * cb + 128) >> 8) + 128);
        P.yuv[total + uvSize + uvIndex] = (uint8_t)(((112 * cr - 94 * cg - 18 * cb + 128) >> 8) + 128);
    }
}
}

// Exact work counters: one atomic per wave per counter.
// wave_box / wave_tri: totals a wave-uniform walk (traverse_uniform, traverse_uniform_stack) kept for the whole
// wave.  A lane stops adding to its copy when its pixel is done, so the lane that ran longest holds the total.
template <class Params>
__device__ __forceinline__ void flush_counters(Params &P, const PathState &L, int lane,
                                               unsigned long long wave_box = 0, unsigned long long wave_tri = 0, uint32_t wave_seg = 0)
{
    const uint32_t c0 = wave_sum(L.n_paths), c1 = wave_sum(L.n_seg) + wave_seg, c4 = wave_sum(L.n_light);
    const unsigned long long lane_box = (unsigned long long)L.n_box + ((unsigned long long)L.box_hi << P.fold_bits);
    const unsigned long long lane_tri = (unsigned long long)L.n_tri + ((unsigned long long)L.tri_hi << P.fold_bits);
    const unsigned long long c2 = wave_sum64(lane_box) + wave_max64(wave_box), c3 = wave_sum64(lane_tri) + wave_max64(wave_tri);
    if (lane == 0 && ((c0 | c1 | c4) != 0 || (c2 | c3) != 0)) {
        atomicAdd(&P.counters[0], (unsigned long long)c0);
        atomicAdd(&P.counters[1], (unsigned long long)c1);
        atomicAdd(&P.counters[2], c2);
        atomicAdd(&P.counters[3], c3);
        atomicAdd(&P.counters[4], (unsigned long long)c4);
    }
}

// The launch parameters as they lie in the kernel-argument segment (constant address space: scalar loads).  The persistent
// kernels read what only a tile's prologue or epilogue needs through a pointer the compiler must take as new each time
// (the empty asm), so those fields are loaded where they are used instead of sitting in SGPRs through every trip of the
// tile's loop -- held live across the tile loop they push that loop's own scalars into spills.
typedef const __attribute__((address_space(4))) RenderParams KernargParams;
__device__ __forceinline__ KernargParams *fresh_kernarg()
{
    KernargParams *k = (KernargParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}

// What a persistent wave renders for the slot it drew from the launch's counter.  The first n_split entries of the tile
// order are handed out as `parts` slots each (part q = the 64 / parts consecutive lanes' pixels q * 64 / parts ... of the
// 8 x 8 tile, i.e. whole pixel rows or pieces of one); every later entry is a whole tile.
struct SlotWork {
    int tile;          // index into the launch's tile grid
    int tile_lane;     // which of the tile's 64 pixels this lane renders (lane & 7 = column, lane >> 3 = row)
    bool lane_on;      // false: this lane has no pixel (a part uses only the first 64 / parts lanes)
    bool is_part;
};
template <class Params>
__device__ __forceinline__ SlotWork slot_work(Params &K, uint32_t slot, uint32_t n_split, uint32_t parts, int lane)
{
    SlotWork w;
    w.is_part = slot < n_split * parts;
    const uint32_t entry = w.is_part ? slot / parts : slot - n_split * (parts - 1u);
    w.tile = K.tile_order ? (int)K.tile_order[entry] : (int)entry;
    w.tile_lane = lane;
    w.lane_on = true;
    if (w.is_part) {
        const uint32_t per = 64u / parts;
        w.lane_on = (uint32_t)lane < per;
        w.tile_lane = w.lane_on ? (int)((slot % parts) * per + (uint32_t)lane) : 0;
    }
    return w;
}

// Tile of this wave: its launch slot, or what the adaptive schedule assigns to that slot.
__device__ __forceinline__ int wave_tile(const RenderParams &P, int tid)
{
    const int slot = blockIdx.x * kWavesPerBlock + tid / kWave;
    return P.tile_order ? (int)P.tile_order[slot] : slot;
}

// Adaptive schedule: what this wave's tile cost, for next frame's order.  The cost is a trip count of
// the wave's own loop, not a clock: s_memtime would make every later load "may be clobbered" for
// the compiler and turn the walk's scalar loads into vector loads, and a count is reproducible.
__device__ __forceinline__ void record_tile_cost(const RenderParams &P, int tid, uint32_t cost)
{
    if (P.tile_cost && (tid & (kWave - 1)) == 0) P.tile_cost[wave_tile(P, tid)] = cost;
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// Pixel of this lane: 8x8 tile per wave; in row-band mode (pt_render_bands_async) local row j is row
// (j % band_rows) of this rank's (j / band_rows)-th band, band b of the frame (fb rows, 0 = top)
// belonging to rank b % band_ranks; with an explicit band list (pt_render_band_list_async) it is row (j % band_rows)
// of band band_list[j / band_rows].
template <class Params>
__device__ __forceinline__ bool tile_pixel(Params &P, int tile, int lane, int &x, int &y, int &pixel_index);

__device__ __forceinline__ bool lane_pixel(const RenderParams &P, int tid, int &x, int &y, int &pixel_index)
{
    return tile_pixel(P, wave_tile(P, tid), tid & (kWave - 1), x, y, pixel_index);
}

template <class Params>
__device__ __forceinline__ bool tile_pixel(Params &P, int tile, int lane, int &x, int &y, int &pixel_index)
{
    const int tile_x = tile % P.tiles_x, tile_y = tile / P.tiles_x;
    const int i = tile_x * 8 + (lane & 7), j = tile_y * 8 + (lane >> 3);
    const bool in_task = (tile_y < P.tiles_y) && (i < P.task_w) && (j < P.task_h);
    x = P.task_ox + i; y = P.task_oy + j;
    if (P.band_list) {
        const int k = in_task ? j / P.band_rows : 0;
        y = P.H - 1 - (P.band_list[k] * P.band_rows + (j % P.band_rows));
    } else if (P.band_ranks > 0) {
        const int fb_row = ((j / P.band_rows) * P.band_ranks + P.band_rank) * P.band_rows + (j % P.band_rows);
        y = P.H - 1 - fb_row;
    }
    pixel_index = (P.H - y - 1) * P.W + x;
    return in_task && x >= 0 && y >= 0 && x < P.W && y < P.H;
}

} // namespace dev
} // namespace pt
