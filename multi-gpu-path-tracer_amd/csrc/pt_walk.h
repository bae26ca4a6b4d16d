// pt_walk.h -- device code of the ray / scene intersection: slab and Moller-Trumbore tests and
// the four traversal schedules of the render kernel.  Included by
// pt_device.hip only.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_kernels.h"

namespace pt {
namespace dev {

constexpr int kWave = 64;
#ifndef PT_WAVES_PER_BLOCK
#define PT_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = PT_WAVES_PER_BLOCK;
constexpr int kBlock = kWave * kWavesPerBlock;

struct Hit {
    float t, u, v;
    int slot;          // leaf-order triangle slot of the closest hit, -1 = none
};

// aabb::hit, aabb.h:38-66 (ray_t / rec ignored there too).  inv = 1.0f / direction.
__device__ __forceinline__ bool box_hit(float xmin, float xmax, float ymin, float ymax, float zmin,
                                        float zmax, v3 o, v3 inv)
{
    float t1 = (xmin - o.x) * inv.x, t2 = (xmax - o.x) * inv.x;
    float t3 = (ymin - o.y) * inv.y, t4 = (ymax - o.y) * inv.y;
    float t5 = (zmin - o.z) * inv.z, t6 = (zmax - o.z) * inv.z;
    float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    return !(tmax < 0) && !(tmin > tmax);
}

// triangle::hit, triangle.h:63-113 (Moller-Trumbore) on a precomputed (v0, e1, e2) record.
// Returns true when the candidate is accepted for the interval (tmin, tmax); t/u/v out.
__device__ __forceinline__ bool tri_hit(v3 v0, v3 e1, v3 e2, v3 o, v3 d, float tmin, float tmax,
                                        float &t_out, float &u_out, float &v_out)
{
    v3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    // triangle.h:78 compares in double against +-1e-8; for a float det that is exactly
    // |det| <= 1e-8f (1e-8f < 1e-8 < nextafter(1e-8f)).
    bool ok = !(det <= 1e-8f && det >= -1e-8f);
    float inv_det = pt_rcp_not_tiny(det);             // :84 (double divide == IEEE 1.0f/det; pt_math.h)
    v3 tvec = o - v0;
    float u = dot(tvec, pvec) * inv_det;
    ok = ok && !(u < 0 || u > 1);
    v3 qvec = cross(tvec, e1);
    float v = dot(d, qvec) * inv_det;
    ok = ok && !(v < 0 || u + v > 1);
    float t = dot(e2, qvec) * inv_det;
    ok = ok && (t < tmax && t > tmin);
    t_out = t; u_out = u; v_out = v;
    return ok;
}

// Moller-Trumbore arithmetic of triangle.h:63-98 without the interval test: the u/v/t values and
// whether det, u and v pass (same operations, same order as tri_hit).
__device__ __forceinline__ bool mt_eval(v3 v0, v3 e1, v3 e2, v3 o, v3 d, float &t, float &u, float &v)
{
    v3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    bool ok = !((det <= 1e-8f) & (det >= -1e-8f));
    float inv_det = pt_rcp_not_tiny(det);
    v3 tvec = o - v0;
    u = dot(tvec, pvec) * inv_det;
    ok = ok & !((u < 0) | (u > 1));
    v3 qvec = cross(tvec, e1);
    v = dot(d, qvec) * inv_det;
    ok = ok & !((v < 0) | (u + v > 1));
    t = dot(e2, qvec) * inv_det;
    return ok;
}

__device__ __forceinline__ void load_tri(const TriIsect *tris, int slot, v3 &v0, v3 &e1, v3 &e2)
{
    const float4 *p = reinterpret_cast<const float4 *>(tris + slot);
    float4 a = p[0], b = p[1], c = p[2];
    v0 = mk3(a.x, a.y, a.z);
    e1 = mk3(a.w, b.x, b.y);
    e2 = mk3(b.z, b.w, c.x);
}

// Scene arrays as seen by the wave-uniform walks: never written while a render kernel runs, so they
// are read through the CONSTANT address space.  That is what keeps their (wave-uniform) loads on the
// scalar cache / in SGPRs: from a plain global pointer the compiler only emits s_load while it can
// prove that nothing in the kernel may have clobbered the memory; an atomic or a store inside the
// wave loop, or an s_memtime anywhere before, ends that proof and silently turns the walk's scalar
// loads into vector loads (measured: C2 77 -> 96 ms with a work-queue atomic in the loop, 84 -> 92 ms
// with one s_memtime at kernel entry).
typedef float pt_f4 __attribute__((ext_vector_type(4)));
struct ConstF4 {
    const __attribute__((address_space(4))) pt_f4 *p;
    __device__ __forceinline__ float4 operator[](int i) const { const pt_f4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ float scalar(int i, int c) const
    {
        return ((const __attribute__((address_space(4))) float *)(p + i))[c];
    }
};
__device__ __forceinline__ ConstF4 as_constant(const void *q)
{
    ConstF4 r; r.p = (const __attribute__((address_space(4))) pt_f4 *)(q); return r;
}

// Scene accessors: the node and triangle arrays are read either from HBM (through L1/L2) or,
// when the launch staged them, from LDS (ds_read_b128).
template <bool kLds>
struct SceneView {
    const float4 *nodes;      // NodePair records, 4 x float4 each
    const float4 *tris;       // TriIsect records, 3 x float4 each
};

// BVH::hit, bvh.h:178-246, split into two phases per ray segment:
//   phase A  walks the boxes in the reference's order (both children's boxes per visit,
//            left-first descent, right sibling pushed on the per-lane LDS stack) and only
//            RECORDS the leaves it would test, in visit order, in a per-lane LDS queue;
//   phase B  runs Moller-Trumbore over the queued leaves in that same order.
// This is the reference's exact sequence of triangle tests (aabb::hit ignores ray_t, so the
// boxes never depend on earlier triangle results); it only keeps the lanes of a wave in the
// same code at the same time.  When a lane's queue is full the wave drains (phase B) and
// resumes phase A where it stopped.
// `stack` / `queue` are this lane's LDS columns (element stride = `stride` lanes).
// Leaf-queue entry: {first slot, triangle count}.  Scenes with < 65536 triangles pack it into
// one dword (half the LDS), larger scenes use two.
template <bool kPacked> struct QueueEntry;
template <> struct QueueEntry<true> {
    typedef uint32_t type;
    static __device__ __forceinline__ type make(uint32_t first, uint32_t count) { return first | (count << 16); }
    static __device__ __forceinline__ void split(type e, uint32_t &first, uint32_t &count) { first = e & 0xFFFFu; count = e >> 16; }
};
template <> struct QueueEntry<false> {
    typedef uint2 type;
    static __device__ __forceinline__ type make(uint32_t first, uint32_t count) { return make_uint2(first, count); }
    static __device__ __forceinline__ void split(type e, uint32_t &first, uint32_t &count) { first = e.x; count = e.y; }
};

template <bool kLds, bool kPacked>
__device__ __forceinline__ Hit traverse(const SceneView<kLds> S, v3 o, v3 d, float tmin, float tmax0,
                                        uint32_t *stack, typename QueueEntry<kPacked>::type *queue, int stride,
                                        int qcap, uint32_t &n_box, uint32_t &n_tri)
{
    Hit h; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.slot = -1;
    float tmax = tmax0;
    v3 inv = mk3(pt_rcp(d.x), pt_rcp(d.y), pt_rcp(d.z));
    uint32_t pair = 0;
    int sp = 0;
    bool walking = true;
    while (walking) {
        // ---------------- phase A: boxes -> leaf queue
        int qn = 0;
        while (walking && qn + 2 <= qcap) {
            const float4 *np = S.nodes + 4 * (size_t)pair;
            float4 a = np[0], b = np[1], c = np[2];
            float4 lf = np[3];
            uint32_t llink = __float_as_uint(lf.x), lcnt = __float_as_uint(lf.y);
            uint32_t rlink = __float_as_uint(lf.z), rcnt = __float_as_uint(lf.w);
            bool ol = box_hit(a.x, a.y, a.z, a.w, b.x, b.y, o, inv);
            bool orr = box_hit(b.z, b.w, c.x, c.y, c.z, c.w, o, inv);
            n_box += 2;
            if (ol && lcnt) { queue[qn * stride] = QueueEntry<kPacked>::make(llink, lcnt); qn++; }
            if (orr && rcnt) { queue[qn * stride] = QueueEntry<kPacked>::make(rlink, rcnt); qn++; }
            bool tl = ol && lcnt == 0, tr = orr && rcnt == 0;
            if (tl && tr) {
                stack[sp * stride] = rlink;
                sp++;
                pair = llink;
            } else if (tl) {
                pair = llink;
            } else if (tr) {
                pair = rlink;
            } else if (sp == 0) {
                walking = false;
            } else {
                sp--;
                pair = stack[sp * stride];
            }
        }
        // ---------------- phase B: triangles of the queued leaves, in queue order
        int qi = 0;
        uint32_t k = 0, n = 0;
        while (true) {
            if (n == 0) {
                if (qi == qn) break;
                QueueEntry<kPacked>::split(queue[qi * stride], k, n);
                qi++;
            }
            const float4 *tp = S.tris + 3 * (size_t)k;
            float4 ta = tp[0], tb = tp[1];
            float e2z = reinterpret_cast<const float *>(tp + 2)[0];
            float t, u, v;
            n_tri++;
            if (tri_hit(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, e2z), o, d, tmin, tmax, t, u, v)) {
                tmax = t; h.t = t; h.u = u; h.v = v; h.slot = (int)k;
            }
            k++; n--;
        }
    }
    return h;
}

// Wave-uniform traversal for scenes whose whole tree is at most 64 inner nodes (Cornell box:
// 10).  All lanes of the wave walk the SAME node / triangle at the same time, in the
// reference's visit order (NodePairs are numbered in DFS pre-order = visit order), each lane
// carrying a 64-bit "this pair is on my path" mask instead of a stack.  Because the loop
// counters are wave-uniform, node and triangle records arrive through the scalar cache
// (s_load) and sit in SGPRs as instruction operands: no LDS, no per-lane queue, no divergent
// loop trip counts.  A pair (or a leaf) that no lane of the wave reaches is skipped with one
// scalar branch.  Per lane the sequence of slab and triangle tests is exactly bvh.h:178-246.
// n_box / n_tri are WAVE totals here (every lane ends up with the same value, kept in SGPRs): tests are
// counted as popcount(participating lanes) once per pair / leaf instead of one VALU add per lane and test.
__device__ __forceinline__ Hit traverse_uniform(const RenderParams &P, v3 o, v3 d, float tmin, float tmax0,
                                                unsigned long long &n_box, unsigned long long &n_tri)
{
    Hit h; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.slot = -1;
    float tmax = tmax0;
    const v3 inv = mk3(pt_rcp(d.x), pt_rcp(d.y), pt_rcp(d.z));
    const ConstF4 nodes = as_constant(P.nodes);
    const ConstF4 tris = as_constant(P.tris);
    unsigned long long act = 1ull;                       // bit p: pair p is on this lane's path
    const int n_pairs = P.n_pairs;
    for (int p = 0; p < n_pairs; p++) {
        const bool on = (act >> p) & 1ull;
        const unsigned long long on_mask = pt_ballot(on);
        if (on_mask == 0ull) continue;                   // nobody in the wave visits pair p
        const float4 a = nodes[4 * p], b = nodes[4 * p + 1], c = nodes[4 * p + 2], lf = nodes[4 * p + 3];
        const uint32_t llink = __float_as_uint(lf.x), lcnt = __float_as_uint(lf.y);
        const uint32_t rlink = __float_as_uint(lf.z), rcnt = __float_as_uint(lf.w);
        const bool hl = on && box_hit(a.x, a.y, a.z, a.w, b.x, b.y, o, inv);
        const bool hr = on && box_hit(b.z, b.w, c.x, c.y, c.z, c.w, o, inv);
        n_box += 2u * (uint32_t)__popcll(on_mask);       // wave-uniform count (see the comment above the function)
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const bool hit = side ? hr : hl;
            const uint32_t link = side ? rlink : llink, cnt = side ? rcnt : lcnt;
            if (cnt == 0) {                              // inner child: mark its pair
                if (hit) act |= 1ull << link;
            } else if (const unsigned long long hit_mask = pt_ballot(hit)) {   // leaf child somebody hits
                n_tri += (unsigned long long)cnt * (uint32_t)__popcll(hit_mask);
                for (uint32_t k = link; k < link + cnt; k++) {
                    const float4 ta = tris[3 * k], tb = tris[3 * k + 1];
                    const float e2z = tris.scalar(3 * k + 2, 0);
                    if (hit) {
                        float t, u, v;
                        if (tri_hit(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, e2z), o, d, tmin, tmax, t, u, v)) {
                            tmax = t; h.t = t; h.u = u; h.v = v; h.slot = (int)k;
                        }
                    }
                }
            }
        }
    }
    return h;
}

// ---- wave-uniform mask walk with DEALT sparse leaves (kWalk 4) ----------------------------------------------
// The mask walk above runs a leaf's triangle tests for the whole wave as soon as ONE lane's ray hit the leaf's
// (origin-containing) box: on the Cornell box three leaves are hit by ~90 % of the rays, the other eight by 15-50 %,
// so half of the wave sits idle through most leaf loops (PMC: 56 % VALU lane utilisation).  Here the box walk is the
// same, but a leaf that fewer than `deal_min` lanes hit is not tested in place: every (ray, triangle) pair it asks
// for becomes an ITEM in a per-wave LDS list, and after the walk the wave works the list off 64 items at a time,
// each lane testing whatever (ray, triangle) its item names -- ray and triangle records come from LDS, the closest
// hit of a ray is merged with one ds_min_u64 on the key (t bits, visit rank, slot).  min over (t, rank) is exactly
// what the reference's in-order `t < ray_t.max` update computes (triangle.h:100: strict, so the first visited of
// equal t wins), whatever order the tests run in.  Leaves most lanes hit (and any leaf whose items would not fit)
// keep the in-place loop with scalar operands.  Per ray the SET of slab and triangle tests is bvh.h:178-246's, and the
// counters count them per ray as before.
struct DealView {
    uint16_t *items;                 // this wave's item list: lane | slot << 6
    float4 *ray_a;                   // this wave's 64 x {o.x, o.y, o.z, d.x}
    float2 *ray_b;                   // this wave's 64 x {d.y, d.z}
    unsigned long long *keys;        // this wave's 64 closest-hit keys: t bits << 32 | rank << 16 | slot
    const float4 *tris;              // the workgroup's LDS copy of the TriIsect table (3 x float4 per slot)
    int cap;                         // item list capacity
    int deal_min;                    // a leaf hit by at least this many lanes is tested in place
    int recompute_uv;                // the scene has textures: barycentrics of a dealt winner are needed
    // parts of split tiles walk by lanes (part_walk_by_lanes); null when the launch has no parts or the tree does not qualify
    const float4 *boxes;             // the workgroup's 2 x float4 per child box b = 2 * pair + side: {xmin, xmax, ymin, ymax}, {zmin, zmax, -, -}
    const uint4 *reach;              // per child box: {boxes a ray must hit to get into it (its own bit included), link, cnt, -}
    float4 *ray_c;                   // this wave's 32 x {1 / d.x, 1 / d.y, 1 / d.z, -}
};

__device__ __forceinline__ void wave_lds_fence()
{
    // cross-lane hand-off inside ONE wave: the LDS serves a wave's requests in order, so only the compiler needs telling
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One pass over the wave's item list: 64 items at a time, each lane testing the (ray, triangle) pair its item names.
__device__ __forceinline__ void deal_rounds(const DealView &D, int lane, uint32_t n_items, float tmin, float tmax0)
{
    wave_lds_fence();
    for (uint32_t j0 = 0; j0 < n_items; j0 += kWave) {
        const uint32_t j = j0 + (uint32_t)lane;
        if (j < n_items) {
            const uint32_t it = D.items[j];
            const uint32_t r = it & 63u, k = it >> 6;
            const float4 ra = D.ray_a[r];
            const float2 rb = D.ray_b[r];
            const float4 ta = D.tris[3 * k], tb = D.tris[3 * k + 1];
            const float2 tc = reinterpret_cast<const float2 *>(D.tris + 3 * k + 2)[0];
            float t, u, v;
            const bool ok = mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x),
                                    mk3(ra.x, ra.y, ra.z), mk3(ra.w, rb.x, rb.y), t, u, v);
            if (ok & (t < tmax0) & (t > tmin))           // triangle.h:98 with the segment's initial interval
                atomicMin(&D.keys[r], ((unsigned long long)__float_as_uint(t) << 32) | ((unsigned long long)__float_as_uint(tc.y) << 16) | k);
        }
    }
    wave_lds_fence();
}

// ---- a PART of a split tile walks by lanes ---------------------------------------------------------------------
// A rank's share of a multi-GPU frame has fewer tiles than the device has wave slots, so its dearest tiles are handed out
// as parts of 64 / parts pixels each (slot_work, pt_shade.h): the launch then lasts as long as a part's chain of trips, and
// a trip of the walk above costs a part what it costs a whole tile -- ten pair visits one after the other, each waiting
// for its scalar loads, with 16 of 64 lanes at work (profiles/r03_lone_part_c2.json).  aabb::hit ignores the ray's
// interval (aabb.h:38-66), so which boxes a ray hits does not depend on the order of the visits: a part tests ALL child
// boxes of the tree for its few rays at once, g = 64 / rays lanes per ray, box records and rays from LDS.  From the hit bits
// a ray enters child box b exactly when it hit b and every box on the way down to it (`reach`, built at staging): the
// same set of visits as bvh.h:178-246's, hence the same counts; every leaf a ray enters is dealt onto the item list and
// worked off by deal_rounds as above (min over (t, visit rank) = the reference's in-order update).
// Needs: n_pairs <= 16 (the hit bits of all 2 n_pairs boxes in 32 bits) and an item list that takes every triangle for every
// ray of the part (rays x n_tris <= capacity: nothing is worked off in between, which would keep a dozen registers alive
// across the rounds); the host checks both (launch_render, P.part_boxes).
__device__ __forceinline__ unsigned long long part_walk_by_lanes(const RenderParams &P, const DealView &D, int lane, bool live, v3 inv, float tmin,
                                                                 float tmax0, unsigned long long &n_box, unsigned long long &n_tri, int part_lanes)
{
    // (the caller has published ray_a / ray_b / keys of every lane)
    const int g = kWave / part_lanes;                    // lanes per ray = boxes per round: 2 .. 16, wave-uniform
    const int gs = __builtin_ctz((unsigned)g);
    // Everything below that depends on the lane alone (ray, box of the round, their LDS addresses) is the same in every trip
    // of a part: left to itself the compiler computes it once per part and, out of registers, spills it around the trip loop
    // -- a scratch reload per trip in a chain where latency is everything.  An opaque copy of the lane id keeps it here.
    int ln = lane;
    asm volatile("; lane-derived values of part_walk_by_lanes stay in the trip" : "+v"(ln));
    const int r = ln >> gs, bl = ln & (g - 1);
    if (ln < part_lanes) D.ray_c[ln] = make_float4(inv.x, inv.y, inv.z, 0.0f);
    wave_lds_fence();
    const float4 ra = D.ray_a[r], rc = D.ray_c[r];
    const v3 ro = mk3(ra.x, ra.y, ra.z), rinv = mk3(rc.x, rc.y, rc.z);
    const unsigned long long live_mask = pt_ballot(live);
    const bool ray_live = (live_mask >> r) & 1ull;
    const int nb = 2 * P.n_pairs;
    uint32_t bh = 0;                                     // bit b: my ray hits child box b (all g lanes of a ray keep its bits)
    for (int b0 = 0; b0 < nb; b0 += g) {
        const int b = b0 + bl;
        bool hit = false;
        if (b < nb) {
            const float4 ba = D.boxes[2 * b], bb = D.boxes[2 * b + 1];
            hit = box_hit(ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, ro, rinv);
        }
        const unsigned long long m = pt_ballot(hit);
        bh |= ((uint32_t)(m >> (r << gs)) & ((1u << g) - 1u)) << b0;
    }
    n_box += 2u * (uint32_t)pt_popc(live_mask);          // the root pair: every live ray tests its two boxes
    uint32_t n_items = 0;                                // wave-uniform
    for (int b0 = 0; b0 < nb; b0 += g) {
        const int b = b0 + bl;
        uint32_t link = 0, mine = 0;
        bool inner = false;
        if (b < nb) {
            const uint4 q = D.reach[b];
            const bool in = ray_live & ((bh & q.x) == q.x);
            link = q.y;
            mine = in ? q.z : 0u;                        // a leaf's triangles (an inner child has none)
            inner = in & (q.z == 0u);
        }
        n_box += 2u * (uint32_t)pt_popc(pt_ballot(inner));     // the pair behind an inner child: two more boxes
        // the k-th triangle of every leaf entered in this round, one vote per k
        for (uint32_t k = 0;; k++) {
            const unsigned long long mk = pt_ballot(mine > k);
            if (mk == 0ull) break;
            const uint32_t n = (uint32_t)pt_popc(mk);
            if (mine > k) {
                const uint32_t pos = n_items + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                D.items[pos] = (uint16_t)((uint32_t)r | ((link + k) << 6));
            }
            n_items += n;
            n_tri += n;
        }
    }
    deal_rounds(D, lane, n_items, tmin, tmax0);          // (fences at both ends also when the list is empty)
    return D.keys[lane];
}

// Called by ALL 64 lanes of the wave (lanes without a ray pass live = false): the rounds that work the item list off
// use every lane, whether or not its own pixel still has samples to trace.
// kInPlace = false: every leaf is dealt (a list that fills up is worked off and reused), no in-place loop is compiled.
template <bool kInPlace>
__device__ __forceinline__ Hit traverse_uniform_dealt(const RenderParams &P, const DealView &D, int lane, bool live, v3 o, v3 d, float tmin,
                                                      float tmax0, unsigned long long &n_box, unsigned long long &n_tri, int part_lanes = 0)
{
    // The closest hit so far is ONE 64-bit key, t bits << 32 | visit rank << 16 | slot, in the in-place loops as in the
    // list: "closer, or as close and visited earlier" is a single unsigned compare, and the lane keeps neither the
    // barycentrics nor a separate t_max (they are recomputed for the winner where a texture needs them).
    unsigned long long best = ~0ull;
    const v3 inv = mk3(pt_rcp(d.x), pt_rcp(d.y), pt_rcp(d.z));
    const ConstF4 nodes = as_constant(P.nodes);
    const ConstF4 tris = as_constant(P.tris);
    D.ray_a[lane] = make_float4(o.x, o.y, o.z, d.x);
    D.ray_b[lane] = make_float2(d.y, d.z);
    D.keys[lane] = ~0ull;
    uint32_t n_items = 0;                                // wave-uniform
    bool any_items = false;                              // wave-uniform: some test of this segment went through the list
    unsigned long long act = live ? 1ull : 0ull;
    const int n_pairs = part_lanes ? 0 : P.n_pairs;      // a part of a split tile (wave-uniform) walks by lanes instead
    if (part_lanes) {
        best = part_walk_by_lanes(P, D, lane, live, inv, tmin, tmax0, n_box, n_tri, part_lanes);
        any_items = false;
    }
    for (int p = 0; p < n_pairs; p++) {
        const bool on = (act >> p) & 1ull;
        const unsigned long long on_mask = pt_ballot(on);
        if (on_mask == 0ull) continue;
        const float4 a = nodes[4 * p], b = nodes[4 * p + 1], c = nodes[4 * p + 2], lf = nodes[4 * p + 3];
        const uint32_t llink = __float_as_uint(lf.x), lcnt = __float_as_uint(lf.y);
        const uint32_t rlink = __float_as_uint(lf.z), rcnt = __float_as_uint(lf.w);
        const bool hl = on && box_hit(a.x, a.y, a.z, a.w, b.x, b.y, o, inv);
        const bool hr = on && box_hit(b.z, b.w, c.x, c.y, c.z, c.w, o, inv);
        n_box += 2u * (uint32_t)__popcll(on_mask);
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const bool hit = side ? hr : hl;
            const uint32_t link = side ? rlink : llink, cnt = side ? rcnt : lcnt;
            if (cnt == 0) {
                if (hit) act |= 1ull << link;
            } else if (const unsigned long long hit_mask = pt_ballot(hit)) {
                const uint32_t lanes = (uint32_t)__popcll(hit_mask);
                n_tri += (unsigned long long)cnt * lanes;
                if (!kInPlace) {
                    // every leaf is dealt; a list that cannot take another triangle's pairs is worked off first
                    const uint32_t rank_in = __builtin_amdgcn_mbcnt_hi((uint32_t)(hit_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit_mask, 0u));
                    any_items = true;
                    uint32_t k0 = 0;
                    while (k0 < cnt) {
                        const uint32_t room = ((uint32_t)D.cap - n_items) / lanes;
                        if (room == 0) { deal_rounds(D, lane, n_items, tmin, tmax0); n_items = 0; continue; }
                        const uint32_t take = cnt - k0 < room ? cnt - k0 : room;
                        if (hit) {
                            const uint32_t pos = n_items + rank_in;
                            for (uint32_t k = 0; k < take; k++) D.items[pos + k * lanes] = (uint16_t)((uint32_t)lane | ((link + k0 + k) << 6));
                        }
                        n_items += take * lanes;
                        k0 += take;
                    }
                } else if ((int)lanes < D.deal_min && n_items + cnt * lanes <= (uint32_t)D.cap) {
                    // sparse leaf: its (ray, triangle) pairs go on the list, triangle-major so that a round's lanes read
                    // one or two triangle records (LDS broadcast) and ascending ray records
                    any_items = true;
                    if (hit) {
                        const uint32_t pos = n_items + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit_mask, 0u));
                        for (uint32_t k = 0; k < cnt; k++) D.items[pos + k * lanes] = (uint16_t)((uint32_t)lane | ((link + k) << 6));
                    }
                    n_items += cnt * lanes;
                } else {
                    for (uint32_t k = link; k < link + cnt; k++) {
                        const float4 ta = tris[3 * k], tb = tris[3 * k + 1];
                        const float e2z = tris.scalar(3 * k + 2, 0);
                        const uint32_t low = (__float_as_uint(tris.scalar(3 * k + 2, 1)) << 16) | k;       // rank << 16 | slot, wave-uniform
                        if (hit) {
                            float t, u, v;
                            const bool ok = mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, e2z), o, d, t, u, v);
                            const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | low;
                            if (ok & (t < tmax0) & (t > tmin) & (key < best)) best = key;     // triangle.h:98,100
                        }
                    }
                }
            }
        }
    }
    if (any_items) {                                     // wave-uniform
        deal_rounds(D, lane, n_items, tmin, tmax0);
        const unsigned long long key = D.keys[lane];
        if (key < best) best = key;                      // a dealt triangle is closer (or as close and visited earlier)
    }
    Hit h; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.slot = -1;
    if (best != ~0ull) {
        h.t = __uint_as_float((uint32_t)(best >> 32));
        h.slot = (int)(best & 0xffffu);
        if (D.recompute_uv) {                            // textured scene: the winner's barycentrics (same arithmetic, same bits)
            const float4 ta = D.tris[3 * h.slot], tb = D.tris[3 * h.slot + 1];
            const float e2z = reinterpret_cast<const float *>(D.tris + 3 * h.slot + 2)[0];
            float t;
            (void)mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, e2z), o, d, t, h.u, h.v);
        }
    }
    return h;
}

// The same wave-uniform walk for bigger trees (cornell_duck: 332 inner nodes, leaves of up to
// 1603 triangles): instead of a 64-bit mask each lane keeps the NEXT pair on its path plus its
// pending right siblings on the per-lane LDS stack.  Pairs are numbered in visit order, so the
// pending set is a stack and `next` is always its minimum; the wave scans p upward and a lane
// takes part in pair p iff next == p.  Leaves are tested in lockstep by every lane that hit
// their box, with triangle records coming through the scalar cache -- a 1603-triangle leaf costs
// 1603 uniform trips instead of 1603 divergent gather trips per lane.  Cost per ray segment is
// the UNION of the lanes' paths, so the host picks this walk only for small / medium trees.
// kParts > 1 (cooperative kernel): the kParts waves of a workgroup walk the SAME 64 rays; every wave does
// the whole box walk (aabb::hit ignores ray_t, so the walk does not depend on triangle results) but tests
// only its contiguous share of each leaf's triangles.  The caller merges the kParts partial closest hits.
template <int kParts = 1>
__device__ __forceinline__ Hit traverse_uniform_stack(const RenderParams &P, v3 o, v3 d, float tmin, float tmax0,
                                                      uint32_t *stack, int stride, unsigned long long &n_box, unsigned long long &n_tri,
                                                      int part = 0, uint32_t *rank_out = nullptr)
{
    Hit h; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.slot = -1;
    uint32_t best_rank = 0xffffffffu;
    float tmax = tmax0;
    const v3 inv = mk3(pt_rcp(d.x), pt_rcp(d.y), pt_rcp(d.z));
    const ConstF4 nodes = as_constant(P.nodes);
    const ConstF4 tris = as_constant(P.tris);
    uint32_t next = 0;
    int sp = 0;
    bool walking = true;
    const int n_pairs = P.n_pairs;
    for (int p = 0; p < n_pairs; p++) {
        const bool on = walking && next == (uint32_t)p;
        const unsigned long long on_mask = pt_ballot(on);
        if (on_mask == 0ull) {
            if (pt_ballot(walking) == 0ull) break;        // every lane of the wave is done
            continue;
        }
        const float4 a = nodes[4 * p], b = nodes[4 * p + 1], c = nodes[4 * p + 2], lf = nodes[4 * p + 3];
        const uint32_t llink = __float_as_uint(lf.x), lcnt = __float_as_uint(lf.y);
        const uint32_t rlink = __float_as_uint(lf.z), rcnt = __float_as_uint(lf.w);
        const bool hl = on && box_hit(a.x, a.y, a.z, a.w, b.x, b.y, o, inv);
        const bool hr = on && box_hit(b.z, b.w, c.x, c.y, c.z, c.w, o, inv);
        n_box += 2u * (uint32_t)__popcll(on_mask);       // wave totals, as in traverse_uniform
        if (on) {
            const bool tl = hl && lcnt == 0, tr = hr && rcnt == 0;
            if (tl && tr) {
                stack[sp * stride] = rlink;
                sp++;
                next = llink;
            } else if (tl || tr) {
                next = tl ? llink : rlink;
            } else if (sp == 0) {
                walking = false;
            } else {
                sp--;
                next = stack[sp * stride];
            }
        }
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const bool hit = side ? hr : hl;
            const uint32_t link = side ? rlink : llink, cnt = side ? rcnt : lcnt;
            const unsigned long long hit_mask = cnt != 0 ? pt_ballot(hit) : 0ull;
            if (hit_mask != 0ull) {
                uint32_t k0 = link, k1 = link + cnt;
                if (kParts > 1) {
                    const uint32_t share = (cnt + kParts - 1) / kParts;
                    k0 = link + (uint32_t)part * share;
                    k1 = k0 + share < link + cnt ? k0 + share : link + cnt;
                }
                if (k1 > k0) n_tri += (unsigned long long)(k1 - k0) * (uint32_t)__popcll(hit_mask);
                for (uint32_t k = k0; k < k1; k++) {
                    const float4 ta = tris[3 * k], tb = tris[3 * k + 1];
                    const float e2z = tris.scalar(3 * k + 2, 0);
                    const uint32_t krank = kParts > 1 ? __float_as_uint(tris.scalar(3 * k + 2, 1)) : 0u;
                    if (hit) {
                        float t, u, v;
                        if (tri_hit(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, e2z), o, d, tmin, tmax, t, u, v)) {
                            tmax = t; h.t = t; h.u = u; h.v = v; h.slot = (int)k;
                            if (kParts > 1) best_rank = krank;
                        }
                    }
                }
            }
        }
    }
    if (kParts > 1 && rank_out) *rank_out = best_rank;
    return h;
}

// ---- the stack walk with DEALT sparse leaves (kWalk 5) -------------------------------------------------------------
// traverse_uniform_stack runs a leaf's triangle loop for the whole wave as soon as one lane's ray hit the leaf's box.
// On the duck that is fine for the one leaf of 1 603 triangles, which nearly every ray hits, and wasteful for the ~330
// small leaves, each hit by a few lanes: the wave runs ~4 900 triangle iterations per trip of which a lane needs 2 120
// (PMC: 56 % lane utilisation, 5.4 s per C3 frame).  As in kWalk 4, a leaf that fewer than `deal_min` lanes hit is not
// tested in place: its (ray, triangle) pairs go on the wave's LDS list -- 32-bit items here, lane | slot << 6 -- and the
// list is worked off 64 items at a time, a chunk whenever it fills up and the rest after the walk; triangle records come
// from memory (the table does not fit LDS), rays and keys from LDS.  Closest hit = min over the key
// (t bits, visit rank, slot), in place and in the list alike -- triangle.h:100's strict in-order update.
struct DealView32 {
    uint32_t *items;                 // this wave's item list: lane | slot << 6
    float4 *ray_a;                   // this wave's 64 x {o.x, o.y, o.z, d.x}
    float2 *ray_b;                   // this wave's 64 x {d.y, d.z}
    unsigned long long *keys;        // this wave's 64 closest-hit keys: t bits << 32 | rank << 16 | slot
    int cap;                         // item list capacity
    int deal_min;                    // a leaf hit by at least this many lanes is tested in place
    float4 *stage;                   // this wave's kStageTris x 3 float4: the chunk of a dense leaf being tested (below)
};
// Dense leaves (the duck's 1 603-triangle leaf, hit by nearly every ray) used to stream their records through the scalar
// cache: 77 KB per trip of every wave through 16 KB, 28 % of the scalar requests missed (SQC_DCACHE_MISSES), and the waves of
// a SIMD, all sweeping the same leaf, stalled on the same lines together.  Leaves of at least kStageMin triangles are now
// copied kStageTris records at a time into the wave's own LDS area with one coalesced vector load per lane (through L1 / L2,
// not the scalar cache) and tested from there (uniform-address ds_read_b128 = broadcast; the operands are VGPRs instead of
// SGPRs, the arithmetic and its order per ray are the same).  C3 with the reference's tree 4 208 -> 3 861 ms.  16 records per
// chunk keep the workgroup's LDS at five per CU (21 or 32 records: four per CU, 4 250 ms; 32 with a shorter item list:
// 3 874 ms); the next chunk on its way by global_load_lds while the current one is tested: 3 934 against 3 879 ms, not kept.
constexpr uint32_t kStageTris = 16, kStageMin = 32;

__device__ __forceinline__ void deal_rounds32(const RenderParams &P, const DealView32 &D, int lane, uint32_t n_items, float tmin, float tmax0)
{
    wave_lds_fence();
    const float4 *tg = reinterpret_cast<const float4 *>(P.tris);
    for (uint32_t j0 = 0; j0 < n_items; j0 += kWave) {
        const uint32_t j = j0 + (uint32_t)lane;
        if (j < n_items) {
            const uint32_t it = D.items[j];
            const uint32_t r = it & 63u, k = it >> 6;
            const float4 ra = D.ray_a[r];
            const float2 rb = D.ray_b[r];
            const float4 ta = tg[3 * (size_t)k], tb = tg[3 * (size_t)k + 1];
            const float2 tc = reinterpret_cast<const float2 *>(tg + 3 * (size_t)k + 2)[0];
            float t, u, v;
            const bool ok = mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x),
                                    mk3(ra.x, ra.y, ra.z), mk3(ra.w, rb.x, rb.y), t, u, v);
            if (ok & (t < tmax0) & (t > tmin))           // triangle.h:98 with the segment's initial interval
                atomicMin(&D.keys[r], ((unsigned long long)__float_as_uint(t) << 32) | ((unsigned long long)__float_as_uint(tc.y) << 16) | k);
        }
    }
    wave_lds_fence();
}

// Called by ALL 64 lanes of the wave (lanes without a ray pass live = false), like traverse_uniform_dealt.
__device__ __forceinline__ Hit traverse_uniform_stack_dealt(const RenderParams &P, const DealView32 &D, int lane, bool live, v3 o, v3 d, float tmin,
                                                            float tmax0, uint32_t *stack, int stride, unsigned long long &n_box,
                                                            unsigned long long &n_tri)
{
    unsigned long long best = ~0ull;
    const v3 inv = mk3(pt_rcp(d.x), pt_rcp(d.y), pt_rcp(d.z));
    const ConstF4 nodes = as_constant(P.nodes);
    const ConstF4 tris = as_constant(P.tris);
    D.ray_a[lane] = make_float4(o.x, o.y, o.z, d.x);
    D.ray_b[lane] = make_float2(d.y, d.z);
    D.keys[lane] = ~0ull;
    uint32_t n_items = 0;                                // wave-uniform
    bool any_items = false;                              // wave-uniform
    uint32_t next = 0;
    int sp = 0;
    bool walking = live;
    const int n_pairs = P.n_pairs;
    for (int p = 0; p < n_pairs; p++) {
        const bool on = walking && next == (uint32_t)p;
        const unsigned long long on_mask = pt_ballot(on);
        if (on_mask == 0ull) {
            if (pt_ballot(walking) == 0ull) break;        // every lane of the wave is done
            continue;
        }
        const float4 a = nodes[4 * p], b = nodes[4 * p + 1], c = nodes[4 * p + 2], lf = nodes[4 * p + 3];
        const uint32_t llink = __float_as_uint(lf.x), lcnt = __float_as_uint(lf.y);
        const uint32_t rlink = __float_as_uint(lf.z), rcnt = __float_as_uint(lf.w);
        const bool hl = on && box_hit(a.x, a.y, a.z, a.w, b.x, b.y, o, inv);
        const bool hr = on && box_hit(b.z, b.w, c.x, c.y, c.z, c.w, o, inv);
        n_box += 2u * (uint32_t)pt_popc(on_mask);
        if (on) {
            const bool tl = hl && lcnt == 0, tr = hr && rcnt == 0;
            if (tl && tr) {
                stack[sp * stride] = rlink;
                sp++;
                next = llink;
            } else if (tl || tr) {
                next = tl ? llink : rlink;
            } else if (sp == 0) {
                walking = false;
            } else {
                sp--;
                next = stack[sp * stride];
            }
        }
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const bool hit = side ? hr : hl;
            const uint32_t link = side ? rlink : llink, cnt = side ? rcnt : lcnt;
            const unsigned long long hit_mask = cnt != 0 ? pt_ballot(hit) : 0ull;
            if (hit_mask == 0ull) continue;
            const uint32_t lanes = (uint32_t)pt_popc(hit_mask);
            n_tri += (unsigned long long)cnt * lanes;
            if ((int)lanes < D.deal_min) {
                // sparse leaf: its pairs go on the list, triangle-major; a list that cannot take another triangle's pairs
                // is worked off first (a big leaf hit by a few lanes passes through in chunks)
                const uint32_t rank_in = __builtin_amdgcn_mbcnt_hi((uint32_t)(hit_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit_mask, 0u));
                any_items = true;
                uint32_t k0 = 0;
                while (k0 < cnt) {
                    const uint32_t room = ((uint32_t)D.cap - n_items) / lanes;
                    if (room == 0) { deal_rounds32(P, D, lane, n_items, tmin, tmax0); n_items = 0; continue; }
                    const uint32_t take = cnt - k0 < room ? cnt - k0 : room;
                    if (hit) {
                        const uint32_t pos = n_items + rank_in;
                        for (uint32_t k = 0; k < take; k++) D.items[pos + k * lanes] = (uint32_t)lane | ((link + k0 + k) << 6);
                    }
                    n_items += take * lanes;
                    k0 += take;
                }
            } else {
                // two triangles per iteration: both records are requested before either is used, so the pair costs one
                // scalar-cache round trip and one pass through the loop's control instead of two
                auto test = [&](float4 ta, float4 tb, float e2z, uint32_t low) {
                    float t, u, v;
                    const bool ok = mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, e2z), o, d, t, u, v);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | low;
                    if (ok & (t < tmax0) & (t > tmin) & (key < best)) best = key;     // triangle.h:98,100
                };
                uint32_t k = link;
                const uint32_t k_end = link + cnt;
                if (cnt >= kStageMin) {
                    const float4 *tg = reinterpret_cast<const float4 *>(P.tris);
                    for (; k < k_end; k += kStageTris) {
                        const uint32_t n_chunk = k_end - k < kStageTris ? k_end - k : kStageTris;
                        wave_lds_fence();                                   // the previous chunk's reads are done
                        for (uint32_t i = (uint32_t)lane; i < 3u * n_chunk; i += kWave) D.stage[i] = tg[3 * (size_t)k + i];
                        wave_lds_fence();
                        uint32_t j = 0;
                        for (; j + 1 < n_chunk; j += 2) {
                            const float4 ta = D.stage[3 * j], tb = D.stage[3 * j + 1], tc = D.stage[3 * j + 2];
                            const float4 ua = D.stage[3 * j + 3], ub = D.stage[3 * j + 4], uc = D.stage[3 * j + 5];
                            const uint32_t tl = (__float_as_uint(tc.y) << 16) | (k + j), ul = (__float_as_uint(uc.y) << 16) | (k + j + 1);
                            if (hit) { test(ta, tb, tc.x, tl); test(ua, ub, uc.x, ul); }
                        }
                        if (j < n_chunk) {
                            const float4 ta = D.stage[3 * j], tb = D.stage[3 * j + 1], tc = D.stage[3 * j + 2];
                            if (hit) test(ta, tb, tc.x, (__float_as_uint(tc.y) << 16) | (k + j));
                        }
                    }
                    k = k_end;
                }
                for (; k + 1 < k_end; k += 2) {
                    const float4 ta = tris[3 * k], tb = tris[3 * k + 1], ua = tris[3 * k + 3], ub = tris[3 * k + 4];
                    const float te = tris.scalar(3 * k + 2, 0), ue = tris.scalar(3 * k + 5, 0);
                    const uint32_t tl = (__float_as_uint(tris.scalar(3 * k + 2, 1)) << 16) | k;       // rank << 16 | slot, wave-uniform
                    const uint32_t ul = (__float_as_uint(tris.scalar(3 * k + 5, 1)) << 16) | (k + 1);
                    if (hit) { test(ta, tb, te, tl); test(ua, ub, ue, ul); }
                }
                if (k < k_end) {
                    const float4 ta = tris[3 * k], tb = tris[3 * k + 1];
                    const float te = tris.scalar(3 * k + 2, 0);
                    const uint32_t tl = (__float_as_uint(tris.scalar(3 * k + 2, 1)) << 16) | k;
                    if (hit) test(ta, tb, te, tl);
                }
            }
        }
    }
    if (any_items) {                                     // wave-uniform
        deal_rounds32(P, D, lane, n_items, tmin, tmax0);
        const unsigned long long key = D.keys[lane];
        if (key < best) best = key;
    }
    Hit h; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.slot = -1;
    if (best != ~0ull) {
        h.t = __uint_as_float((uint32_t)(best >> 32));
        h.slot = (int)(best & 0xffffu);
        if (P.any_tex) {                                 // textured scene: the winner's barycentrics (same arithmetic, same bits)
            const float4 *tg = reinterpret_cast<const float4 *>(P.tris);
            const float4 ta = tg[3 * (size_t)h.slot], tb = tg[3 * (size_t)h.slot + 1];
            const float e2z = reinterpret_cast<const float *>(tg + 3 * (size_t)h.slot + 2)[0];
            float t;
            (void)mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, e2z), o, d, t, h.u, h.v);
        }
    }
    return h;
}

// Opt-in walk over the SAH tree (pt_set_bvh_mode(dev, 1); not the reference's traversal): per-lane
// LDS stack, nearer child first, subtrees whose (epsilon-inflated) box starts beyond the closest
// hit so far are pruned.  Triangles are tested with the same Moller-Trumbore arithmetic and the
// closest hit is min over (t, rank) -- rank = the triangle's position in the reference's visit
// order -- which is what the reference's in-order `t < ray_t.max` update computes over the
// triangles it tests.  Differences from the reference are confined to rays for which the two
// traversals test different triangles AND such a triangle wins.
// Records of the SAH tree are addressed by 32-bit BYTE OFFSETS from the array bases (pt_upload_scene scales the
// builder's links: pair index x 64, leaf-order slot x 48; both arrays stay far below 4 GB): a record's address is then
// "uniform base + per-lane 32-bit offset", which a global load takes as it is, instead of a 64-bit multiply-add per step.
struct SahWalk {
    uint32_t pair;                // byte offset of the NodePair to visit next
    int sp;
    float tbest;                  // t of the closest hit so far (the segment's t_max while there is none)
    uint32_t best_rank;
    uint32_t best_off;            // byte offset of its TriIsect record, kNoPair = none.  The barycentrics are not carried
                                  // along the walk (two registers and two selects per triangle test): sah_hit()
                                  // recomputes them for the winner, same arithmetic, same operands -> the same bits
    v3 inv;
    uint32_t k0, n0, k1, n1;      // pending leaf triangles (byte offset, count): range 0 first, then range 1
    uint32_t spill;               // blocks of 8 entries this lane's stack has moved out of LDS (SahStack; almost always 0)
    uint32_t sel;                 // 64-byte form only (sahq_begin): v_perm selector of a child's three NEAR plane bytes
#ifdef PT_STALE
    uint32_t stale, visits;       // diagnostic build: node visits that found both children beyond the closest hit / all visits
#endif
    // `pair` == kNoPair: no node is left to visit.  (A register compare a vote can use directly: a bool carried across
    // trips is materialised as 0 / 1 and compared again for every ballot.)
};
constexpr uint32_t kNoPair = 0xFFFFFFFFu;

// The walk's per-lane stack.  LDS holds the top `cap` entries of every lane ([entry][lane]: conflict-free); what a deep
// ray pushes beyond that moves, eight entries at a time, to a per-lane area in device memory (`spill`, one block per
// (wave, block index, lane): 32 contiguous bytes) and comes back when the lane has popped down to it.  The worst-case
// bound of the 8 M-triangle terrain's tree is 37 entries per lane, 38 KB per workgroup, which held the SAH kernel to four
// waves per SIMD; rays rarely hold more than a dozen.  One compare + one untaken scalar branch per node visit and per pop.
// cap = 2^30 and spill = null where the whole stack is in LDS (the simple SAH kernels, the KAT kernel, PTAMD_SAH_STACK=0):
// the checks below then never fire.
struct SahStack {
    uint32_t *lds;                // this lane's column
    int stride;                   // lanes per entry row (workgroup size)
    int cap;                      // entries per lane in LDS
    uint32_t *spill;              // this lane's first block in device memory (block b at spill + b * 64 * 8 dwords), or null
    unsigned long long *n_spills; // counts blocks moved out (counters[5], read by pt_debug_counters_read: tests prove the path ran), or null
    // the tree's top levels in LDS (PTAMD_SAH_TOP, default 64 records): records at byte offsets below top_bytes are read from
    // `top` instead of from memory.  profiles/r03_topk_lds.json
    const float4 *top;
    uint32_t top_bytes;
};
// A node record's k-th float4: from the workgroup's LDS copy of the top levels, or from memory.
#define PT_NODE_F4(S, P, off, k) (((off) < (S).top_bytes) ? (S).top[(off) / 16u + (k)] \
                                                        : reinterpret_cast<const float4 *>(reinterpret_cast<const char *>((P).nodes) + (size_t)(off))[k])
constexpr int kSpillBlock = 8;

// Makes room for n more pushes (n <= 3 < kSpillBlock <= cap / 2).  kSpill: 0 = the kernel keeps whole stacks in LDS (no code
// at all: the chain of dependent steps of a lone wave is what bounds a rank's share of a frame, and every instruction in it
// counts), 1 = it has the overflow area, 2 = decided per launch.
template <int kSpill>
__device__ __forceinline__ void sah_stack_reserve(const SahStack &S, SahWalk &W, int n)
{
    if (kSpill == 0) return;
    if ((kSpill == 1 || S.spill != nullptr) && __builtin_expect(pt_ballot(W.sp + n > S.cap) != 0ull, 0)) {
        if (W.sp + n > S.cap) {
            uint32_t *g = S.spill + (size_t)W.spill * (64 * kSpillBlock);
            uint32_t e[kSpillBlock];
#pragma unroll
            for (int k = 0; k < kSpillBlock; k++) e[k] = S.lds[k * S.stride];
            reinterpret_cast<uint4 *>(g)[0] = make_uint4(e[0], e[1], e[2], e[3]);
            reinterpret_cast<uint4 *>(g)[1] = make_uint4(e[4], e[5], e[6], e[7]);
            for (int k = kSpillBlock; k < W.sp; k++) S.lds[(k - kSpillBlock) * S.stride] = S.lds[k * S.stride];
            W.sp -= kSpillBlock;
            W.spill++;
            if (S.n_spills) atomicAdd(S.n_spills, 1ull);
        }
    }
}
// The lane's LDS part is empty: fetch its latest block back, if there is one.  Returns false when the whole stack is empty.
template <int kSpill>
__device__ __forceinline__ bool sah_stack_refill(const SahStack &S, SahWalk &W)
{
    if (kSpill == 0 || W.spill == 0) return false;
    W.spill--;
    const uint32_t *g = S.spill + (size_t)W.spill * (64 * kSpillBlock);
    const uint4 a = reinterpret_cast<const uint4 *>(g)[0], b = reinterpret_cast<const uint4 *>(g)[1];
    S.lds[0 * S.stride] = a.x; S.lds[1 * S.stride] = a.y; S.lds[2 * S.stride] = a.z; S.lds[3 * S.stride] = a.w;
    S.lds[4 * S.stride] = b.x; S.lds[5 * S.stride] = b.y; S.lds[6 * S.stride] = b.z; S.lds[7 * S.stride] = b.w;
    W.sp = kSpillBlock;
    return true;
}
// What a lane of the SAH walk wants next: a triangle test, a node visit, or nothing (walk complete).
__device__ __forceinline__ bool sah_wants_tri(const SahWalk &W) { return W.n0 != 0; }
__device__ __forceinline__ bool sah_wants_node(const SahWalk &W) { return (W.n0 == 0) & (W.pair != kNoPair); }
__device__ __forceinline__ bool sah_walking(const SahWalk &W) { return (W.n0 != 0) | (W.pair != kNoPair); }

__device__ __forceinline__ void sah_begin(SahWalk &W, v3 d, float tmax0)
{
    W.pair = 0; W.sp = 0; W.tbest = tmax0; W.best_rank = 0xFFFFFFFFu;
    W.inv = mk3(pt_rcp(d.x), pt_rcp(d.y), pt_rcp(d.z));
    W.best_off = 0xFFFFFFFFu;
    W.k0 = W.n0 = W.k1 = W.n1 = 0;
    W.spill = 0;
    W.sel = 0;
}

// The 64-byte form's additions to sah_begin (sahq_node_eval): reciprocals clamped to +-2^100 -- an exactly axis-parallel ray
// (get_ray's cancellation produces ~2 400 of them per C4 frame) then has a finite slope and byte * (unit * inv) can never be
// 0 * inf -- and the byte selector that picks, per axis, the plane the ray meets first.
constexpr uint32_t kSahqFarXor = 0x00070503u;       // near selector ^ this = far selector (x: 0 <-> 3, y: 1 <-> 4, z: 2 <-> 5)
__device__ __forceinline__ void sahq_begin(SahWalk &W)
{
    const float big = 1.2676506002282294e30f;        // 2^100
    W.inv = mk3(fminf(fmaxf(W.inv.x, -big), big), fminf(fmaxf(W.inv.y, -big), big), fminf(fmaxf(W.inv.z, -big), big));
    // bytes of {hi, lo}: 0 xlo, 1 ylo, 2 zlo, 3 xhi, 4 yhi, 5 zhi; 0x0c = constant zero
    W.sel = 0x0c020100u ^ ((__float_as_uint(W.inv.x) >> 31) * 0x03u) ^ ((__float_as_uint(W.inv.y) >> 31) * 0x0500u) ^ ((__float_as_uint(W.inv.z) >> 31) * 0x070000u);
}

// One MICRO-step of the SAH walk: either one Moller-Trumbore test (if this lane has leaf triangles
// pending) or one node visit (both child boxes; leaf children become pending ranges, inner
// children are descended into / pushed, nearer first).  A wave trip therefore runs the triangle
// block once and the node block once whatever mix of lanes needs which, instead of a whole leaf
// loop that only the lanes standing on a leaf take part in (PMC before: 19 % VALU lane
// utilisation on the duck).  Pending triangles go first, so the next node visit prunes against
// the closest hit they may have produced.  Returns false when the walk is complete.
__device__ __forceinline__ bool sah_tri_step(const RenderParams &P, SahWalk &W, v3 o, v3 d, float tmin, uint32_t &n_tri)
{
    // up to TWO pending triangles per step (leaves hold up to four, stored back to back), in one trip of the wave's loop.
    // (The second record is loaded under `if (two)` into registers pre-set with a copy of the first: the compiler then waits
    // for the first record before it requests the second -- two memory round trips in a row.  Trees that live in the caches
    // keep this form: the second trip is short there and the form needs ten registers fewer, which the six-wave build of
    // the 128-byte-node kernel has not got (16 spilled registers instead of 7 with sah_tri_step_together, C3 102.5 -> 107.8 ms).)
    const float4 *tp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(P.tris) + (size_t)W.k0);
    const bool two = W.n0 >= 2;
    const float4 ta = tp[0], tb = tp[1], tc = tp[2];
    float4 ua = ta, ub = tb, uc = tc;
    if (two) { ua = tp[3]; ub = tp[4]; uc = tp[5]; }
    {
        float t, u, v;
        const bool ok = mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d, t, u, v);
        const uint32_t rank = __float_as_uint(tc.y);
        if (ok & (t > tmin) & ((t < W.tbest) | ((t == W.tbest) & (rank < W.best_rank)))) {
            W.tbest = t; W.best_rank = rank; W.best_off = W.k0;
        }
    }
    if (two) {
        float t, u, v;
        const bool ok = mt_eval(mk3(ua.x, ua.y, ua.z), mk3(ua.w, ub.x, ub.y), mk3(ub.z, ub.w, uc.x), o, d, t, u, v);
        const uint32_t rank = __float_as_uint(uc.y);
        if (ok & (t > tmin) & ((t < W.tbest) | ((t == W.tbest) & (rank < W.best_rank)))) {
            W.tbest = t; W.best_rank = rank; W.best_off = W.k0 + (uint32_t)sizeof(TriIsect);
        }
    }
    const uint32_t done = two ? 2u : 1u;
    n_tri += done;
    W.k0 += done * (uint32_t)sizeof(TriIsect); W.n0 -= done;
    if (W.n0 == 0) { W.k0 = W.k1; W.n0 = W.n1; W.n1 = 0; }
    return sah_walking(W);
}

// The same step with BOTH records requested before either is used: one memory round trip per pair.  For trees beyond the caches
// (the 64-byte-node kernel), where a round trip is an HBM miss: C4 52.1 -> 47.6 ms.
__device__ __forceinline__ bool sah_tri_step_together(const RenderParams &P, SahWalk &W, v3 o, v3 d, float tmin, uint32_t &n_tri)
{
    // up to TWO pending triangles per step (leaves hold up to four, stored back to back): both records are requested
    // before either is used, so the pair costs one memory round trip and one trip of the wave's loop instead of two.
    // (Round 4 found that it did not: the second record used to be loaded under `if (two)` into registers pre-set with a
    // copy of the first, and the copy made the compiler wait for the first record -- s_waitcnt vmcnt(0) -- BEFORE it
    // issued the second's loads: two serial round trips per pair, four per leaf of four.  Now a lane without a second
    // triangle simply requests its first one again (same cache line), no exec-mask region, no copies, and the updates
    // are selects.)
    const float4 *tp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(P.tris) + (size_t)W.k0);
    const bool two = W.n0 >= 2;
    const float4 *up = two ? tp + 3 : tp;
    const float4 ta = tp[0], tb = tp[1], tc = tp[2];
    const float4 ua = up[0], ub = up[1], uc = up[2];
    {
        float t, u, v;
        const bool ok = mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), o, d, t, u, v);
        const uint32_t rank = __float_as_uint(tc.y);
        const bool take = ok & (t > tmin) & ((t < W.tbest) | ((t == W.tbest) & (rank < W.best_rank)));
        W.tbest = take ? t : W.tbest; W.best_rank = take ? rank : W.best_rank; W.best_off = take ? W.k0 : W.best_off;
    }
    {
        float t, u, v;
        const bool ok = mt_eval(mk3(ua.x, ua.y, ua.z), mk3(ua.w, ub.x, ub.y), mk3(ub.z, ub.w, uc.x), o, d, t, u, v);
        const uint32_t rank = __float_as_uint(uc.y);
        const bool take = two & ok & (t > tmin) & ((t < W.tbest) | ((t == W.tbest) & (rank < W.best_rank)));
        W.tbest = take ? t : W.tbest; W.best_rank = take ? rank : W.best_rank; W.best_off = take ? W.k0 + (uint32_t)sizeof(TriIsect) : W.best_off;
    }
    const uint32_t done = two ? 2u : 1u;
    n_tri += done;
    W.k0 += done * (uint32_t)sizeof(TriIsect); W.n0 -= done;
    if (W.n0 == 0) { W.k0 = W.k1; W.n0 = W.n1; W.n1 = 0; }
    return sah_walking(W);
}

__device__ __forceinline__ bool sah_node_step(const RenderParams &P, SahWalk &W, v3 o, v3 d, const SahStack &S,
                                              uint32_t &n_box)
{
    uint32_t *const stack = S.lds;
    const int stride = S.stride;
    sah_stack_reserve<2>(S, W, 1);
    const float4 *np = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(P.nodes) + (size_t)W.pair);
    const float4 a = np[0], b = np[1], c = np[2], lf = np[3];
    const uint32_t llink = __float_as_uint(lf.x), lcnt = __float_as_uint(lf.y);
    const uint32_t rlink = __float_as_uint(lf.z), rcnt = __float_as_uint(lf.w);
    const v3 inv = W.inv;
    float tnl, tnr;
    bool hl, hr;
#ifdef PT_STALE
    bool g_l = false, g_r = false;
#endif
    {
        float t1 = (a.x - o.x) * inv.x, t2 = (a.y - o.x) * inv.x, t3 = (a.z - o.y) * inv.y, t4 = (a.w - o.y) * inv.y;
        float t5 = (b.x - o.z) * inv.z, t6 = (b.y - o.z) * inv.z;
        tnl = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        float tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
        // the per-ray margin of the 4-wide forms (sahq_node_eval), here on the interval itself: 2^-19 of its ends
        tnl = fmaf(-fabsf(tnl), 1.9073486328125e-06f, tnl); tf = fmaf(fabsf(tf), 1.9073486328125e-06f, tf);
        hl = (tf >= fmaxf(tnl, 0.0f)) & (tnl <= W.tbest);
#ifdef PT_STALE
        g_l = tf >= fmaxf(tnl, 0.0f);
#endif
    }
    {
        float t1 = (b.z - o.x) * inv.x, t2 = (b.w - o.x) * inv.x, t3 = (c.x - o.y) * inv.y, t4 = (c.y - o.y) * inv.y;
        float t5 = (c.z - o.z) * inv.z, t6 = (c.w - o.z) * inv.z;
        tnr = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        float tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
        tnr = fmaf(-fabsf(tnr), 1.9073486328125e-06f, tnr); tf = fmaf(fabsf(tf), 1.9073486328125e-06f, tf);
        hr = (tf >= fmaxf(tnr, 0.0f)) & (tnr <= W.tbest);
#ifdef PT_STALE
        g_r = tf >= fmaxf(tnr, 0.0f);
#endif
    }
#ifdef PT_STALE
    W.stale += (!hl && !hr && (g_l || g_r)) ? 1u : 0u;      // visited only to find both children beyond the closest hit
    W.visits++;
#endif
    n_box += 2;
    // leaf children -> pending triangle ranges
    W.k0 = llink; W.n0 = (hl && lcnt) ? lcnt : 0u;
    W.k1 = rlink; W.n1 = (hr && rcnt) ? rcnt : 0u;
    if (W.n0 == 0) { W.k0 = W.k1; W.n0 = W.n1; W.n1 = 0; }
    // inner children -> next pair (nearer first, farther pushed) or pop
    const bool gl = hl && lcnt == 0, gr = hr && rcnt == 0;
    if (gl && gr) {
        const bool left_first = tnl <= tnr;
        stack[W.sp * stride] = left_first ? rlink : llink;
        W.sp++;
        W.pair = left_first ? llink : rlink;
    } else if (gl || gr) {
        W.pair = gl ? llink : rlink;
    } else if (W.sp == 0 && !sah_stack_refill<2>(S, W)) {
        W.pair = kNoPair;
    } else {
        W.sp--;
        W.pair = stack[W.sp * stride];
    }
    return sah_walking(W);
}

// ---- the same walk over the 4-wide form of the SAH tree (pt_internal.h collapse_sah_quads; RenderParams::sah == 2) -------
// A ray's chain of dependent fetches is what bounds the dearest pixels, the tail of a launch and every rank's share of a
// frame; four child boxes per visit roughly halve the visits.  A visit tests the four boxes, orders the children it has to
// enter, pushes all but the nearest and continues with the nearest; stack entries are
// link words (inner node or leaf + triangle count), so a leaf that is not the nearest waits on the stack like a node.
// State: exactly one of "pending triangles" (n0 != 0), "pending node" (pair != kNoPair), or nothing left.
__device__ __forceinline__ void sahw_take(SahWalk &W, uint32_t x)
{
    if (x & 15u) { W.k0 = x & ~15u; W.n0 = x & 15u; }
    else W.pair = x;
}
template <int kSpill>
__device__ __forceinline__ void sahw_pop(SahWalk &W, const SahStack &S)
{
    if (W.sp > 0 || sah_stack_refill<kSpill>(S, W)) {
        W.sp--;
        sahw_take(W, S.lds[W.sp * S.stride]);
    }
}

// kTogether: both records of a pair requested before either is used (sah_tri_step_together; trees beyond the caches)
template <int kSpill, bool kTogether = false>
__device__ __forceinline__ bool sahw_tri_step(const RenderParams &P, SahWalk &W, v3 o, v3 d, float tmin, const SahStack &S,
                                              uint32_t &n_tri)
{
    if (kTogether) (void)sah_tri_step_together(P, W, o, d, tmin, n_tri);
    else (void)sah_tri_step(P, W, o, d, tmin, n_tri);      // (k1 / n1 are never set in this form: the range just ends)
    // A triangle trip finishes its leaf (<= 4 triangles) instead of leaving the second pair to another vote: the same tests
    // in the same order for the lane, one trip of the wave less (C3 110.3 -> 106.9 ms, C4 52.8 -> 51.5 ms).  Going on with a
    // leaf that comes off the stack right after it loses: C3 129 ms, C4 60 ms.
    if (pt_ballot(W.n0 != 0) != 0ull) {
        if (W.n0 != 0) { if (kTogether) (void)sah_tri_step_together(P, W, o, d, tmin, n_tri); else (void)sah_tri_step(P, W, o, d, tmin, n_tri); }
    }
    if ((W.n0 == 0) & (W.pair == kNoPair)) sahw_pop<kSpill>(W, S);
    return sah_walking(W);
}

__device__ __forceinline__ void sahw_order(float &ka, uint32_t &la, float &kb, uint32_t &lb)
{
    const bool swap = kb < ka;
    const float k = swap ? kb : ka; kb = swap ? ka : kb; ka = k;
    const uint32_t l = swap ? lb : la; lb = swap ? la : lb; la = l;
}

// One child of a plane-major record (QuadF): its planes' ray parameters are fma(plane, inv, offset), the near planes'
// with the lowered offset, the far planes' with the raised one (sahw_node_step).
__device__ __forceinline__ void sahw_slab(float xn, float xf, float yn, float yf, float zn, float zf, v3 inv, float nx, float ny, float nz,
                                          float fx, float fy, float fz, float tbest, float &key)
{
    // (Packed v_pk_fma_f32 would halve these and compile cleanly from two-element vectors, but on gfx950 a packed f32
    // instruction issues several times slower than the two scalar ones it replaces (MI355X_MICROARCH.md constants table;
    // round 1's SLP measurement: C2 98 -> 118 ms), so they stay scalar.)
    const float t1 = fmaf(xn, inv.x, nx), t3 = fmaf(yn, inv.y, ny), t5 = fmaf(zn, inv.z, nz);
    const float t2 = fmaf(xf, inv.x, fx), t4 = fmaf(yf, inv.y, fy), t6 = fmaf(zf, inv.z, fz);
    const float tn = fmaxf(fmaxf(t1, t3), t5);
    const float tf = fminf(fminf(t2, t4), t6);
    const bool hit = (tf >= fmaxf(tn, 0.0f)) & (tn <= tbest);          // the test of sah_node_step, per child
    key = hit ? tn : __builtin_inff();
}

// The float form's addition to sah_begin: reciprocals clamped to +-2^100, so that plane * inv is never 0 * inf for an exactly
// axis-parallel ray (sahq_begin has the same clamp).
__device__ __forceinline__ void sahw_begin(SahWalk &W)
{
    const float big = 1.2676506002282294e30f;        // 2^100
    W.inv = mk3(fminf(fmaxf(W.inv.x, -big), big), fminf(fmaxf(W.inv.y, -big), big), fminf(fmaxf(W.inv.z, -big), big));
}

// A visit of the plane-major 128-byte record (QuadF, pt_internal.h).  The ray parameter of a plane is
// plane * inv - o * inv: one fma per plane against an offset formed once per visit, and the row of the four children's planes
// the ray meets first on an axis sits at an offset the sign of inv picks -- near rows and far rows are LOADED, not selected:
// 12 instructions per child where the subtract-multiply-min-max form had 25 (150 -> 100 per visit).
// Rounding: the offset -(o * inv) carries one rounding relative to |o * inv| that (plane - o) * inv does not have; the near
// offsets are lowered and the far offsets raised by 2^-19 of their magnitude: 2^-23 bounds that rounding, the rest widens every
// box by 2e-6 |o| -- the share of the slab / Moller-Trumbore difference that grows with the origin's distance, which the
// builder's inflation (1e-5 of the scene's coordinate magnitude, pt_bvh.cpp) cannot know; the fma's own rounding is of the
// kind every form has and is inside that inflation.
// kTop: 0 = every record from memory, 1 = the tree's top levels from the workgroup's LDS copy, 2 = decided per launch
template <int kTop, int kSpill>
__device__ __forceinline__ bool sahw_node_step(const RenderParams &P, SahWalk &W, v3 o, v3 d, const SahStack &S, uint32_t &n_box)
{
    uint32_t *const stack = S.lds;
    const int stride = S.stride;
    sah_stack_reserve<kSpill>(S, W, 3);
    v3 inv = W.inv;
#ifndef PT_SAHW_HOIST
    // (Left alone the compiler hoists everything below that depends on the ray only -- three row offsets, six plane offsets --
    // out of the walk's loop into nine registers per lane, which the six-wave build has not got: 29 spilled registers
    // instead of 16, C3 101.7 -> 113.3 ms.  Fifteen instructions per visit are cheaper.)
    asm volatile("" : "+v"(inv.x), "+v"(inv.y), "+v"(inv.z));
#endif
    const uint32_t off = W.pair;
    // byte offsets of the near / far rows: row 2k (+0) for a rising ray, 2k + 1 (+16) for a falling one; records are 128-byte aligned
    const uint32_t xn_o = off + ((__float_as_uint(inv.x) >> 31) << 4), xf_o = xn_o ^ 16u;
    const uint32_t yn_o = off + ((__float_as_uint(inv.y) >> 31) << 4), yf_o = yn_o ^ 16u;
    const uint32_t zn_o = off + ((__float_as_uint(inv.z) >> 31) << 4), zf_o = zn_o ^ 16u;
    float4 xn, xf, yn, yf, zn, zf, lk;
    if (kTop == 0 || (kTop == 2 && S.top_bytes == 0u)) {
        const char *base = reinterpret_cast<const char *>(P.nodes);
        xn = *reinterpret_cast<const float4 *>(base + (size_t)xn_o); xf = *reinterpret_cast<const float4 *>(base + (size_t)xf_o);
        yn = *reinterpret_cast<const float4 *>(base + (size_t)yn_o + 32); yf = *reinterpret_cast<const float4 *>(base + (size_t)yf_o + 32);
        zn = *reinterpret_cast<const float4 *>(base + (size_t)zn_o + 64); zf = *reinterpret_cast<const float4 *>(base + (size_t)zf_o + 64);
        lk = *reinterpret_cast<const float4 *>(base + (size_t)off + 96);
    } else {
        // A lane's record comes from the workgroup's LDS copy of the top levels or from memory.  Written as a select per float4
        // (`in_top ? lds[k] : global[k]`) the compiler selects the POINTER and fetches through seven flat_load_dwordx4 -- every
        // request then goes down both the LDS and the vector-memory path and is waited for on both counters.  Two address
        // spaces, two exec-masked groups of loads, one wait:
        typedef const __attribute__((address_space(3))) pt_f4 *LdsF4Ptr;
        typedef const __attribute__((address_space(1))) pt_f4 *GlobalF4Ptr;
        typedef const __attribute__((address_space(3))) char *LdsBytePtr;
        typedef const __attribute__((address_space(1))) char *GlobalBytePtr;
        const bool in_top = off < S.top_bytes;
        pt_f4 r0, r1, r2, r3, r4, r5, r6;
        if (!in_top) {
            GlobalBytePtr g = (GlobalBytePtr)(reinterpret_cast<const char *>(P.nodes));
            r0 = *(GlobalF4Ptr)(g + (size_t)xn_o); r1 = *(GlobalF4Ptr)(g + (size_t)xf_o);
            r2 = *(GlobalF4Ptr)(g + (size_t)yn_o + 32); r3 = *(GlobalF4Ptr)(g + (size_t)yf_o + 32);
            r4 = *(GlobalF4Ptr)(g + (size_t)zn_o + 64); r5 = *(GlobalF4Ptr)(g + (size_t)zf_o + 64);
            r6 = *(GlobalF4Ptr)(g + (size_t)off + 96);
        } else {
            LdsBytePtr t = (LdsBytePtr)(S.top);
            r0 = *(LdsF4Ptr)(t + xn_o); r1 = *(LdsF4Ptr)(t + xf_o);
            r2 = *(LdsF4Ptr)(t + yn_o + 32); r3 = *(LdsF4Ptr)(t + yf_o + 32);
            r4 = *(LdsF4Ptr)(t + zn_o + 64); r5 = *(LdsF4Ptr)(t + zf_o + 64);
            r6 = *(LdsF4Ptr)(t + off + 96);
        }
        xn = make_float4(r0.x, r0.y, r0.z, r0.w); xf = make_float4(r1.x, r1.y, r1.z, r1.w); yn = make_float4(r2.x, r2.y, r2.z, r2.w);
        yf = make_float4(r3.x, r3.y, r3.z, r3.w); zn = make_float4(r4.x, r4.y, r4.z, r4.w); zf = make_float4(r5.x, r5.y, r5.z, r5.w);
        lk = make_float4(r6.x, r6.y, r6.z, r6.w);
    }
    const float cx = -(o.x * inv.x), cy = -(o.y * inv.y), cz = -(o.z * inv.z);
    const float m = 1.9073486328125e-06f;          // 2^-19
    const float nx = fmaf(-fabsf(cx), m, cx), ny = fmaf(-fabsf(cy), m, cy), nz = fmaf(-fabsf(cz), m, cz);
    const float fx = fmaf(fabsf(cx), m, cx), fy = fmaf(fabsf(cy), m, cy), fz = fmaf(fabsf(cz), m, cz);
    float k0, k1, k2, k3;
    sahw_slab(xn.x, xf.x, yn.x, yf.x, zn.x, zf.x, inv, nx, ny, nz, fx, fy, fz, W.tbest, k0);
    sahw_slab(xn.y, xf.y, yn.y, yf.y, zn.y, zf.y, inv, nx, ny, nz, fx, fy, fz, W.tbest, k1);
    sahw_slab(xn.z, xf.z, yn.z, yf.z, zn.z, zf.z, inv, nx, ny, nz, fx, fy, fz, W.tbest, k2);
    sahw_slab(xn.w, xf.w, yn.w, yf.w, zn.w, zf.w, inv, nx, ny, nz, fx, fy, fz, W.tbest, k3);
    uint32_t l0 = __float_as_uint(lk.x), l1 = __float_as_uint(lk.y), l2 = __float_as_uint(lk.z), l3 = __float_as_uint(lk.w);
    n_box += 4;
    // The nearest child to be entered goes first; the others wait in whatever order they end up in.  A full sort (five
    // compare-exchanges instead of three) saves node visits but costs more than it saves on the scene that lives in the
    // caches (C3 125.8 -> 122.0 ms without it; C4 53.2 -> 53.8 ms).
    sahw_order(k0, l0, k1, l1); sahw_order(k0, l0, k2, l2); sahw_order(k0, l0, k3, l3);
    const float inf = __builtin_inff();
    W.pair = kNoPair;
    if (k3 < inf) { stack[W.sp * stride] = l3; W.sp++; }
    if (k2 < inf) { stack[W.sp * stride] = l2; W.sp++; }
    if (k1 < inf) { stack[W.sp * stride] = l1; W.sp++; }
    if (k0 < inf) sahw_take(W, l0);
    else sahw_pop<kSpill>(W, S);
    return sah_walking(W);
}

// ---- the 4-wide form in 64-byte records (QuadQ, pt_internal.h; RenderParams::sah == 3) ------------------------------------
// Same walk, same stack entries; a visit costs four dwordx4 requests per lane instead of seven.  A plane is
// origin + byte * unit (unit a power of two), so its ray parameter is byte * (unit * inv) + (origin - o) * inv: per visit
// three slopes and three offsets, per plane one conversion and one fma.  Which of a child's two planes per axis the ray meets
// first is the sign of inv, the same for every node: v_perm_b32 gathers the three near bytes and the three far bytes (W.sel),
// which replaces the six min / max of the symmetric slab test: 144 VALU instructions per visit where the decode-then-slab
// form had 193, C4 47.5 -> 46.0 ms (profiles/r04_c4_visit_slope.json: a visit's instructions and requests both sit at a knee --
// 24 padded multiplies cost 2.3 ms, one padded request 2.0 ms, two 8.7 ms -- so what is removed on one side pays in part).
// Rounding.  Against the exact parameter of the decoded plane the offset carries two roundings relative to |origin - o|
// (the subtraction, the product), which the unfolded form did not have: the near offsets are lowered and the far offsets
// raised by 2^-19 of their magnitude: 2^-22 bounds both, the rest widens every box by 2e-6 |origin - o| -- the share of the
// difference between slab and Moller-Trumbore arithmetic that grows with the origin's distance (the test's products round
// relative to |o - v0|), which the builder's inflation of every box (pt_bvh.cpp: 1e-5 of the scene's coordinate magnitude)
// cannot know.  The last rounding (the fma's, relative to t itself) is of the kind every form has and is inside that inflation.
__device__ __forceinline__ void sahq_slab(uint32_t lo, uint32_t hi, uint32_t sel_n, uint32_t sel_f, float ax, float ay, float az,
                                          float nx, float ny, float nz, float fx, float fy, float fz, float tbest, bool present, float &key)
{
    const uint32_t bn = __builtin_amdgcn_perm(hi, lo, sel_n), bf = __builtin_amdgcn_perm(hi, lo, sel_f);
    const float t1 = fmaf((float)(bn & 0xffu), ax, nx), t3 = fmaf((float)((bn >> 8) & 0xffu), ay, ny), t5 = fmaf((float)((bn >> 16) & 0xffu), az, nz);
    const float t2 = fmaf((float)(bf & 0xffu), ax, fx), t4 = fmaf((float)((bf >> 8) & 0xffu), ay, fy), t6 = fmaf((float)((bf >> 16) & 0xffu), az, fz);
    const float tn = fmaxf(fmaxf(t1, t3), t5);
    const float tf = fminf(fminf(t2, t4), t6);
    const bool hit = (tf >= fmaxf(tn, 0.0f)) & (tn <= tbest);
    key = (present & hit) ? tn : __builtin_inff();
}

template <int kSpill>
__device__ __forceinline__ void sahq_node_eval(SahWalk &W, v3 o, const SahStack &S, float4 q0, float4 q1, float4 q2, float4 lk, uint32_t &n_box)
{
    uint32_t *const stack = S.lds;
    const int stride = S.stride;
    const uint32_t exps = __float_as_uint(q0.w);
    const float sx = __uint_as_float((exps & 0xffu) << 23), sy = __uint_as_float(((exps >> 8) & 0xffu) << 23), sz = __uint_as_float(((exps >> 16) & 0xffu) << 23);
    uint32_t l0 = __float_as_uint(lk.x), l1 = __float_as_uint(lk.y), l2 = __float_as_uint(lk.z), l3 = __float_as_uint(lk.w);
    const v3 inv = W.inv;
    const float ax = sx * inv.x, ay = sy * inv.y, az = sz * inv.z;
    const float cx = (q0.x - o.x) * inv.x, cy = (q0.y - o.y) * inv.y, cz = (q0.z - o.z) * inv.z;
    const float m = 1.9073486328125e-06f;          // 2^-19
    const float nx = fmaf(-fabsf(cx), m, cx), ny = fmaf(-fabsf(cy), m, cy), nz = fmaf(-fabsf(cz), m, cz);
    const float fx = fmaf(fabsf(cx), m, cx), fy = fmaf(fabsf(cy), m, cy), fz = fmaf(fabsf(cz), m, cz);
    const uint32_t sel_n = W.sel, sel_f = W.sel ^ kSahqFarXor;
    float k0, k1, k2, k3;
    sahq_slab(__float_as_uint(q1.x), __float_as_uint(q2.x), sel_n, sel_f, ax, ay, az, nx, ny, nz, fx, fy, fz, W.tbest, l0 != 0xFFFFFFFFu, k0);
    sahq_slab(__float_as_uint(q1.y), __float_as_uint(q2.y), sel_n, sel_f, ax, ay, az, nx, ny, nz, fx, fy, fz, W.tbest, l1 != 0xFFFFFFFFu, k1);
    sahq_slab(__float_as_uint(q1.z), __float_as_uint(q2.z), sel_n, sel_f, ax, ay, az, nx, ny, nz, fx, fy, fz, W.tbest, l2 != 0xFFFFFFFFu, k2);
    sahq_slab(__float_as_uint(q1.w), __float_as_uint(q2.w), sel_n, sel_f, ax, ay, az, nx, ny, nz, fx, fy, fz, W.tbest, l3 != 0xFFFFFFFFu, k3);
    n_box += 4;
#ifdef PT_PAD_NODE
    // diagnostic build only (EXPERIMENTS.md, round 4 "what bounds a trip"): PT_PAD_NODE extra multiplies per visit, to
    // read the slope of frame time against VALU instructions; never compiled into the shipped library
    {
        float pad = k0;
#pragma unroll
        for (int i = 0; i < PT_PAD_NODE; i++) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(pad));
        if (pad == 1.2345e-30f) k0 = pad;          // (never true for a slab key; keeps the chain alive)
    }
#endif
    sahw_order(k0, l0, k1, l1); sahw_order(k0, l0, k2, l2); sahw_order(k0, l0, k3, l3);
    const float inf = __builtin_inff();
    W.pair = kNoPair;
    if (k3 < inf) { stack[W.sp * stride] = l3; W.sp++; }
    if (k2 < inf) { stack[W.sp * stride] = l2; W.sp++; }
    if (k1 < inf) { stack[W.sp * stride] = l1; W.sp++; }
    if (k0 < inf) sahw_take(W, l0);
    else sahw_pop<kSpill>(W, S);
}

template <int kTop, int kSpill>
__device__ __forceinline__ bool sahq_node_step(const RenderParams &P, SahWalk &W, v3 o, v3 d, const SahStack &S, uint32_t &n_box)
{
    sah_stack_reserve<kSpill>(S, W, 3);
    float4 q0, q1, q2, lk;
    if (kTop == 0 || (kTop == 2 && S.top_bytes == 0u)) {
        const float4 *np = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(P.nodes) + (size_t)W.pair);
        q0 = np[0]; q1 = np[1]; q2 = np[2]; lk = np[3];
    } else {
        const uint32_t off = W.pair;
        q0 = PT_NODE_F4(S, P, off, 0); q1 = PT_NODE_F4(S, P, off, 1); q2 = PT_NODE_F4(S, P, off, 2); lk = PT_NODE_F4(S, P, off, 3);
    }
#ifdef PT_PAD_LOAD
    // diagnostic build only: PT_PAD_LOAD extra requests per visit for pieces of the same record through an address the
    // compiler cannot see through (no common-subexpression), to read the slope of frame time against vector-memory requests
    {
        uint32_t off2 = W.pair;
        asm volatile("" : "+v"(off2));
        const float4 *np2 = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(P.nodes) + (size_t)off2);
#pragma unroll
        for (int i = 0; i < PT_PAD_LOAD; i++) { const float4 e = np2[i & 3]; if (e.x == 1.2345e-30f) q0.x = e.y; }
    }
#endif
    sahq_node_eval<kSpill>(W, o, S, q0, q1, q2, lk, n_box);
    return sah_walking(W);
}

// The finished walk's hit record.  want_uv: the scene has textures (the only reader of the barycentrics, pt_shade.h).
__device__ __forceinline__ Hit sah_hit(const RenderParams &P, const SahWalk &W, v3 o, v3 d)
{
    Hit h; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.slot = -1;
    if (W.best_off != 0xFFFFFFFFu) {
        h.t = W.tbest;
        h.slot = (int)(W.best_off / (uint32_t)sizeof(TriIsect));        // once per segment (a multiply-high by a constant)
        if (P.any_tex) {
            const float4 *tp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(P.tris) + (size_t)W.best_off);
            const float4 ta = tp[0], tb = tp[1];
            const float e2z = reinterpret_cast<const float *>(tp + 2)[0];
            float t;
            (void)mt_eval(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, e2z), o, d, t, h.u, h.v);
        }
    }
    return h;
}

__device__ __forceinline__ bool sah_step(const RenderParams &P, SahWalk &W, v3 o, v3 d, float tmin,
                                         const SahStack &S, uint32_t &n_box, uint32_t &n_tri)
{
    if (W.n0 != 0) return sah_tri_step(P, W, o, d, tmin, n_tri);
    return sah_node_step(P, W, o, d, S, n_box);
}

__device__ __forceinline__ Hit traverse_sah(const RenderParams &P, v3 o, v3 d, float tmin, float tmax0,
                                            uint32_t *stack, int stride, uint32_t &n_box, uint32_t &n_tri)
{
    SahWalk W;
    sah_begin(W, d, tmax0);
    const SahStack S = {stack, stride, P.stack_cap, nullptr, nullptr, nullptr, 0u};      // the whole stack in LDS: stack_cap is out of reach
    if (P.sah == 3) {
        sahq_begin(W);
        while (W.n0 != 0 ? sahw_tri_step<0, true>(P, W, o, d, tmin, S, n_tri) : sahq_node_step<0, 0>(P, W, o, d, S, n_box)) {}
    } else if (P.sah == 2) {
        sahw_begin(W);
        while (W.n0 != 0 ? sahw_tri_step<0>(P, W, o, d, tmin, S, n_tri) : sahw_node_step<0, 0>(P, W, o, d, S, n_box)) {}
    } else {
        while (sah_step(P, W, o, d, tmin, S, n_box, n_tri)) {}
    }
    return sah_hit(P, W, o, d);
}

} // namespace dev
} // namespace pt
