// pt_bvh.cpp -- host build of the reference-topology BVH.
//
// The reference builds its BVH with ONE device thread inside create_world<<<1,1>>>
// (DevicePathTracer.h:134-146 -> bvh.h:20-176).  The topology, the node numbering and the
// triangle permutation are part of the observable result (they fix the traversal's visit
// order and therefore exact-tie resolution), so this builder reproduces the
// same arithmetic -- including the default box that always contains the origin
// (aabb.h:12-14,90-96) -- on the host, where it runs once per scene upload.
#include "pt_internal.h"

#include <algorithm>
#include <cmath>
#include <future>
#include <thread>

namespace pt {
namespace {

struct Builder {
    // Working copy of the triangles, permuted in lockstep with out.indices (rec[i] describes triangle
    // out.indices[i]): a node's triangles are then one contiguous stream instead of a gather through the
    // permutation, which is what an 8 M-triangle build spends its time on otherwise.
    struct Rec { float p[9]; float c[3]; };      // vertices; triangle.h:27 centroid = (v0+v1+v2)*0.3333f
    std::vector<Rec> rec;
    RefBvh &out;

    // Host threads: the reference builds with one (device) thread, and what it builds -- topology, node numbers,
    // triangle permutation -- is observable, so threads are used only where the result cannot depend on them:
    //  * the split candidates of a big node are costed, and its bounds computed, over chunks of its triangles
    //    in parallel: counts add up and fminf / fmaxf over a set do not depend on the order (NaN coordinates
    //    are ignored by both forms; a box never ends at a signed zero because it always contains the default
    //    box [-5e-6, 5e-6]^3, so the one order-dependent case of fminf / fmaxf cannot be the result);
    //  * the two subtrees of a big node are built concurrently in node pools of their own (they own disjoint
    //    ranges of `indices`); the node numbers of the serial build (children allocated as a pair, then the
    //    whole left subtree, then the right one: bvh.h:74-93) are assigned afterwards by renumber().
    // The partition (bvh.h:56-72) defines the triangle order and stays serial.
    static constexpr uint32_t kChunkParallelMin = 1u << 17;
    static constexpr uint32_t kSubtreeParallelMin = 1u << 16;
    typedef std::vector<RefNode> Pool;

    explicit Builder(const float *pos, size_t n, RefBvh &o) : rec(n), out(o)
    {
        for (size_t i = 0; i < n; i++) {
            const float *t = pos + 9 * i;
            for (int k = 0; k < 9; k++) rec[i].p[k] = t[k];
            rec[i].c[0] = ((t[0] + t[3]) + t[6]) * 0.3333f;
            rec[i].c[1] = ((t[1] + t[4]) + t[7]) * 0.3333f;
            rec[i].c[2] = ((t[2] + t[5]) + t[8]) * 0.3333f;
        }
    }

    // aabb() + pad_to_minimums(): [0,0] expanded by delta/2 on each side.
    static void box_default(RefNode &b)
    {
        float delta = 0.00001;
        float padding = delta / 2;
        b.xmin = 0.0f - padding; b.xmax = 0.0f + padding;
        b.ymin = 0.0f - padding; b.ymax = 0.0f + padding;
        b.zmin = 0.0f - padding; b.zmax = 0.0f + padding;
    }
    // fminf(t, b) / fmaxf(t, b) of aabb::extend (aabb.h:80-88) as one minss / maxss: a NaN coordinate t is
    // ignored by both forms, b is never NaN, and the two forms differ only in which zero they return for
    // (+0, -0) -- which no box bound can be, since every box contains [-5e-6, 5e-6]^3 (partial boxes that start
    // empty are merged into such a box before they are used).  A libm call per min/max was 3/4 of the build.
    static float lo(float t, float b) { return t < b ? t : b; }
    static float hi(float t, float b) { return t > b ? t : b; }
    // aabb::extend for the three vertices of the triangle at position `at` of the permutation.
    void box_extend_tri(RefNode &b, uint32_t at) const
    {
        const float *t = rec[at].p;
        for (int k = 0; k < 3; k++) {
            b.xmin = lo(t[3 * k + 0], b.xmin); b.ymin = lo(t[3 * k + 1], b.ymin);
            b.zmin = lo(t[3 * k + 2], b.zmin);
            b.xmax = hi(t[3 * k + 0], b.xmax); b.ymax = hi(t[3 * k + 1], b.ymax);
            b.zmax = hi(t[3 * k + 2], b.zmax);
        }
    }
    static float box_area(const RefNode &b)          // aabb.h:75-78 (half area)
    {
        float ex = b.xmax - b.xmin, ey = b.ymax - b.ymin, ez = b.zmax - b.zmin;
        return ex * ey + ey * ez + ez * ex;
    }
    void update_bounds(RefNode &n) const              // bvh.h:96-106
    {
        if (n.triCount >= kChunkParallelMin) { update_bounds_chunked(n); return; }
        box_default(n);
        for (uint32_t i = 0; i < n.triCount; i++) box_extend_tri(n, n.firstTri + i);
    }
    static void box_empty(RefNode &b)
    {
        b.xmin = b.ymin = b.zmin = INFINITY; b.xmax = b.ymax = b.zmax = -INFINITY;
    }
    static void box_merge(RefNode &b, const RefNode &part)          // same forms as box_extend_tri
    {
        b.xmin = lo(part.xmin, b.xmin); b.ymin = lo(part.ymin, b.ymin); b.zmin = lo(part.zmin, b.zmin);
        b.xmax = hi(part.xmax, b.xmax); b.ymax = hi(part.ymax, b.ymax); b.zmax = hi(part.zmax, b.zmax);
    }
    static unsigned chunk_threads(uint32_t count)
    {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        return std::min<unsigned>(std::min(hw, 16u), std::max<uint32_t>(1, count >> 15));
    }
    // evaluateSAH (bvh.h:144-176) for all candidates of a node at once.  The reference makes one pass over the
    // node's triangles per candidate (nine passes, each extending one of two boxes by every triangle); here a
    // triangle is classified against the three planes of an axis (the same `centroid < pos` comparisons) and
    // extends ONE box per axis, that of its comparison pattern; a candidate's left / right box is then the union
    // of the pattern boxes on its side and its counts their sums.  Same boxes (fminf / fmaxf over the same
    // sets, see the note on order above), same counts, a third of the box updates and one gather per triangle.
    struct Patterns { RefNode box[3][8]; uint32_t cnt[3][8]; };
    void classify(const RefNode &n, uint32_t lo, uint32_t hi, const float (*plane)[3], const bool *axis_on, Patterns &P) const
    {
        for (int a = 0; a < 3; a++) for (int k = 0; k < 8; k++) { box_empty(P.box[a][k]); P.cnt[a][k] = 0; }
        for (uint32_t i = lo; i < hi; i++) {
            const uint32_t at = n.firstTri + i;
            const float *c3 = rec[at].c;
            for (int a = 0; a < 3; a++) {
                if (!axis_on[a]) continue;
                const int k = (c3[a] < plane[a][0] ? 1 : 0) | (c3[a] < plane[a][1] ? 2 : 0) | (c3[a] < plane[a][2] ? 4 : 0);
                P.cnt[a][k]++;
                box_extend_tri(P.box[a][k], at);
            }
        }
    }
    // cost[a][i] of the plane i of axis a (unused entries untouched)
    void evaluate_candidates(const RefNode &n, const float (*plane)[3], const bool *axis_on, float (*cost)[3]) const
    {
        const unsigned T = n.triCount >= kChunkParallelMin ? chunk_threads(n.triCount) : 1;
        std::vector<Patterns> parts(T);
        auto work = [&](unsigned t) {
            classify(n, (uint32_t)((uint64_t)n.triCount * t / T), (uint32_t)((uint64_t)n.triCount * (t + 1) / T), plane, axis_on, parts[t]);
        };
        std::vector<std::future<void>> jobs;
        for (unsigned t = 1; t < T; t++) jobs.push_back(std::async(std::launch::async, work, t));
        work(0);
        for (auto &j : jobs) j.get();
        for (int a = 0; a < 3; a++) {
            if (!axis_on[a]) continue;
            for (int i = 0; i < 3; i++) {
                RefNode lb, rb;
                box_default(lb); box_default(rb);
                uint32_t lc = 0, rc = 0;
                for (unsigned t = 0; t < T; t++)
                    for (int k = 0; k < 8; k++) {
                        if ((k >> i) & 1) { box_merge(lb, parts[t].box[a][k]); lc += parts[t].cnt[a][k]; }
                        else              { box_merge(rb, parts[t].box[a][k]); rc += parts[t].cnt[a][k]; }
                    }
                const float v = (float)(int)lc * box_area(lb) + (float)(int)rc * box_area(rb);
                cost[a][i] = v > 0 ? v : 1e30f;
            }
        }
    }
    void update_bounds_chunked(RefNode &n) const
    {
        const unsigned T = chunk_threads(n.triCount);
        std::vector<RefNode> parts(T);
        auto work = [&](unsigned t) {
            box_empty(parts[t]);
            const uint32_t lo = (uint32_t)((uint64_t)n.triCount * t / T), hi = (uint32_t)((uint64_t)n.triCount * (t + 1) / T);
            for (uint32_t i = lo; i < hi; i++) box_extend_tri(parts[t], n.firstTri + i);
        };
        std::vector<std::future<void>> jobs;
        for (unsigned t = 1; t < T; t++) jobs.push_back(std::async(std::launch::async, work, t));
        work(0);
        for (auto &j : jobs) j.get();
        box_default(n);
        for (unsigned t = 0; t < T; t++) box_merge(n, parts[t]);
    }
    // Builds the subtree under pool[idx] (node numbers local to `pool`); returns its deepest level.
    uint32_t subdivide(Pool &pool, uint32_t idx, uint32_t level, int forks)      // bvh.h:35-94
    {
        if (pool[idx].triCount <= 5) return level;
        const RefNode n = pool[idx];
        int axis = 0;
        float split = 0.0f, best = 1e30f;
        float plane[3][3], cost[3][3];
        bool axis_on[3];
        for (int a = 0; a < 3; a++) {                 // bvh.h:108-137
            float bmin = a == 0 ? n.xmin : (a == 1 ? n.ymin : n.zmin);
            float bmax = a == 0 ? n.xmax : (a == 1 ? n.ymax : n.zmax);
            axis_on[a] = !(bmin == bmax);
            float scale = (bmax - bmin) / 4;
            for (uint32_t i = 1; i < 4; i++) plane[a][i - 1] = bmin + (float)i * scale;
        }
        evaluate_candidates(n, plane, axis_on, cost);
        for (int a = 0; a < 3; a++)                   // first strictly lower cost wins, axis 0..2, i 1..3
            for (int i = 0; i < 3; i++)
                if (axis_on[a] && cost[a][i] < best) { split = plane[a][i]; axis = a; best = cost[a][i]; }
        float nosplit = (float)n.triCount * box_area(n);              // bvh.h:139-142
        if (best >= nosplit) return level;
        int i = (int)n.firstTri;
        int j = i + (int)n.triCount - 1;
        while (i <= j) {                               // bvh.h:56-72
            if (rec[(size_t)i].c[axis] < split) i++;
            else { std::swap(rec[(size_t)i], rec[(size_t)j]); std::swap(out.indices[i], out.indices[j--]); }
        }
        int leftCount = i - (int)n.firstTri;
        if (leftCount == 0 || leftCount == (int)n.triCount) return level;
        RefNode ln, rn;
        ln.left = ln.right = 0; rn.left = rn.right = 0;
        ln.firstTri = n.firstTri;  ln.triCount = (uint32_t)leftCount;
        rn.firstTri = (uint32_t)i; rn.triCount = n.triCount - (uint32_t)leftCount;
        const bool fork = forks > 0 && std::min(ln.triCount, rn.triCount) >= kSubtreeParallelMin;
        uint32_t deepest;
        if (fork) {
            Pool other(1, ln);
            auto fut = std::async(std::launch::async, [this, &other, level, forks] {
                update_bounds(other[0]);
                return subdivide(other, 0, level + 1, forks - 1);
            });
            const uint32_t r = (uint32_t)pool.size();
            pool.push_back(rn);
            update_bounds(pool[r]);
            deepest = subdivide(pool, r, level + 1, forks - 1);
            deepest = std::max(deepest, fut.get());
            const uint32_t base = (uint32_t)pool.size();
            for (RefNode nd : other) {
                if (nd.triCount == 0) { nd.left += base; nd.right += base; }
                pool.push_back(nd);
            }
            pool[idx].left = base; pool[idx].right = r;
        } else {
            const uint32_t l = (uint32_t)pool.size();
            pool.push_back(ln); pool.push_back(rn);
            update_bounds(pool[l]);
            update_bounds(pool[l + 1]);
            pool[idx].left = l; pool[idx].right = l + 1;
            deepest = subdivide(pool, l, level + 1, forks);
            deepest = std::max(deepest, subdivide(pool, l + 1, level + 1, forks));
        }
        pool[idx].triCount = 0;
        return deepest;
    }
    // The serial build's numbering (bvh.h:74-75,92-93): `nodesUsed++` twice for the children, then everything
    // the left subtree allocates, then everything the right subtree allocates.
    void renumber(const Pool &pool, uint32_t old_idx, uint32_t new_idx, uint32_t &used)
    {
        RefNode n = pool[old_idx];
        if (n.triCount == 0) {
            const uint32_t l = used++, r = used++;
            const uint32_t ol = n.left, orr = n.right;
            n.left = l; n.right = r;
            out.nodes[new_idx] = n;
            renumber(pool, ol, l, used);
            renumber(pool, orr, r, used);
        } else {
            out.nodes[new_idx] = n;
        }
    }

    // Worst-case use of the 64-entry traversal stack of bvh.h:179 (sentinel included):
    // a right sibling is pushed only when both children are inner nodes (bvh.h:227-236).
    uint32_t stack_need(uint32_t idx, uint32_t held) const
    {
        const RefNode &n = out.nodes[idx];
        if (n.triCount != 0) return held;
        bool li = out.nodes[n.left].triCount == 0, ri = out.nodes[n.right].triCount == 0;
        uint32_t need = held;
        if (li && ri) {
            need = std::max(need, stack_need(n.left, held + 1));
            need = std::max(need, stack_need(n.right, held));
        } else if (li) {
            need = std::max(need, stack_need(n.left, held));
        } else if (ri) {
            need = std::max(need, stack_need(n.right, held));
        }
        return need;
    }
};

} // namespace

void build_reference_bvh(const float *positions9, size_t n_tris, RefBvh &out)
{
    out = RefBvh();
    out.indices.resize(n_tris);
    for (size_t i = 0; i < n_tris; i++) out.indices[i] = (int)i;
    Builder b(positions9, n_tris, out);
    Builder::Pool pool;
    pool.reserve(std::min<size_t>(2 * n_tris, (size_t)1 << 22));
    pool.emplace_back();
    pool[0].left = pool[0].right = 0;
    pool[0].firstTri = 0;
    pool[0].triCount = (uint32_t)n_tris;
    b.update_bounds(pool[0]);
    out.depth = b.subdivide(pool, 0, 0, 6);
    out.nodes.assign(pool.size(), RefNode());
    uint32_t used = 1;
    b.renumber(pool, 0, 0, used);
    for (const RefNode &n : out.nodes) {
        if (n.triCount != 0) {
            out.leaves++;
            out.max_leaf = std::max(out.max_leaf, n.triCount);
        }
    }
    out.max_stack = b.stack_need(0, 1);
}

} // namespace pt

// ===================================================================================
// Opt-in "SAH" acceleration structure: tight bounds, 16-bin surface-area
// heuristic, leaves of <= 4 triangles (the product: <= 2, sah_leaf_size_for), boxes inflated by a small epsilon so that pruning by
// distance stays conservative with respect to the Moller-Trumbore arithmetic.  It does NOT
// reproduce the reference's set of tested triangles (that set is defined by the reference's
// origin-containing boxes), so results can differ from the reference in ill-conditioned cases
// (rays grazing a triangle's plane); DESIGN.md quantifies the difference.  Ties in t are
// resolved by `rank` (the reference's visit order when available) exactly as the reference
// resolves them.
// ===================================================================================
#include <cfloat>
#include <cstring>
#include <future>

namespace pt {
namespace {

struct SahTri { float bmin[3], bmax[3], c[3]; int face; };

struct SahNode {
    float bmin[3], bmax[3];
    int left = -1, right = -1;      // children (node ids) or -1
    uint32_t first = 0, count = 0;  // leaf range in the ordered triangle array
};

// Bin of a scaled centroid offset; non-finite coordinates (inf - inf, NaN from a broken file) land in bin 0
// instead of indexing out of range.
static inline int sah_bin(float scaled, int bins)
{
    if (!(scaled >= 0.0f)) return 0;
    return scaled < (float)bins ? (int)scaled : bins - 1;
}

struct SahBuilder {
    std::vector<SahTri> &tris;
    std::vector<SahNode> nodes;     // per-subtree pools are merged by the caller
    uint32_t max_leaf;              // ranges of at most this many triangles become leaves
    explicit SahBuilder(std::vector<SahTri> &t, uint32_t leaf = 4) : tris(t), max_leaf(leaf) {}

    static void grow(float *mn, float *mx, const float *a, const float *b)
    {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], a[k]); mx[k] = std::max(mx[k], b[k]); }
    }
    static float half_area(const float *mn, const float *mx)
    {
        float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        return ex * ey + ey * ez + ez * ex;
    }

    int build(uint32_t lo, uint32_t hi, int depth)
    {
        int id = (int)nodes.size();
        nodes.emplace_back();
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t i = lo; i < hi; i++) { grow(mn, mx, tris[i].bmin, tris[i].bmax); grow(cmn, cmx, tris[i].c, tris[i].c); }
        for (int k = 0; k < 3; k++) { nodes[id].bmin[k] = mn[k]; nodes[id].bmax[k] = mx[k]; }
        uint32_t n = hi - lo;
        if (n <= max_leaf || depth >= 60) { nodes[id].first = lo; nodes[id].count = n; return id; }
        constexpr int kBins = 16;
        float best = FLT_MAX;
        int best_axis = -1, best_bin = 0;
        for (int axis = 0; axis < 3; axis++) {
            float ext = cmx[axis] - cmn[axis];
            if (!(ext > 0.0f)) continue;
            uint32_t cnt[kBins] = {0};
            float bmn[kBins][3], bmx[kBins][3];
            for (int b = 0; b < kBins; b++) for (int k = 0; k < 3; k++) { bmn[b][k] = FLT_MAX; bmx[b][k] = -FLT_MAX; }
            float scale = (float)kBins / ext;
            for (uint32_t i = lo; i < hi; i++) {
                int b = sah_bin((tris[i].c[axis] - cmn[axis]) * scale, kBins);
                cnt[b]++;
                grow(bmn[b], bmx[b], tris[i].bmin, tris[i].bmax);
            }
            float rarea[kBins];
            uint32_t rcount[kBins];
            float amn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, amx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            uint32_t acc = 0;
            for (int b = kBins - 1; b > 0; b--) {
                if (cnt[b]) grow(amn, amx, bmn[b], bmx[b]);
                acc += cnt[b];
                rarea[b] = acc ? half_area(amn, amx) : 0.0f;
                rcount[b] = acc;
            }
            float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            acc = 0;
            for (int b = 0; b < kBins - 1; b++) {
                if (cnt[b]) grow(lmn, lmx, bmn[b], bmx[b]);
                acc += cnt[b];
                if (acc == 0 || rcount[b + 1] == 0) continue;
                float cost = (float)acc * half_area(lmn, lmx) + (float)rcount[b + 1] * rarea[b + 1];
                if (cost < best) { best = cost; best_axis = axis; best_bin = b; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            mid = lo + n / 2;                                   // all centroids coincide: split by count
        } else {
            float ext = cmx[best_axis] - cmn[best_axis], scale = (float)kBins / ext, base = cmn[best_axis];
            int axis = best_axis, bin = best_bin;
            auto it = std::partition(tris.begin() + lo, tris.begin() + hi, [&](const SahTri &t) {
                return sah_bin((t.c[axis] - base) * scale, kBins) <= bin;
            });
            mid = (uint32_t)(it - tris.begin());
            if (mid == lo || mid == hi) mid = lo + n / 2;
        }
        int l, r;
        if (n >= kParallelMin && depth < 6) {
            // big subtrees: the left half on another host thread with its own node pool (the two
            // halves touch disjoint ranges of `tris`), merged behind this pool afterwards.  The
            // resulting tree does not depend on the thread schedule.
            SahBuilder other(tris, max_leaf);
            auto fut = std::async(std::launch::async, [&other, lo, mid, depth] { return other.build(lo, mid, depth + 1); });
            r = build(mid, hi, depth + 1);
            int lroot = fut.get();
            const int base = (int)nodes.size();
            for (SahNode nd : other.nodes) {
                if (nd.left >= 0) { nd.left += base; nd.right += base; }
                nodes.push_back(nd);
            }
            l = lroot + base;
        } else {
            l = build(lo, mid, depth + 1);
            r = build(mid, hi, depth + 1);
        }
        nodes[id].left = l; nodes[id].right = r;
        return id;
    }
    static constexpr uint32_t kParallelMin = 1u << 17;
};

} // namespace

void build_sah_bvh(const float *positions9, size_t n_tris, SahBvh &out, uint32_t max_leaf)
{
    out = SahBvh();
    std::vector<SahTri> tris(n_tris);
    float smn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, smx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n_tris; i++) {
        const float *p = positions9 + 9 * i;
        SahTri &t = tris[i];
        for (int k = 0; k < 3; k++) {
            t.bmin[k] = std::min(p[k], std::min(p[3 + k], p[6 + k]));
            t.bmax[k] = std::max(p[k], std::max(p[3 + k], p[6 + k]));
            t.c[k] = 0.5f * (t.bmin[k] + t.bmax[k]);
            smn[k] = std::min(smn[k], t.bmin[k]); smx[k] = std::max(smx[k], t.bmax[k]);
        }
        t.face = (int)i;
    }
    SahBuilder b(tris, max_leaf < 1 ? 1u : max_leaf);
    b.nodes.reserve(n_tris);
    int root = b.build(0, (uint32_t)n_tris, 0);
    // The inflation that keeps pruning conservative against the Moller-Trumbore arithmetic: a ray the float test accepts passes
    // the exact triangle within a few units in the last place of the coordinates involved (the test's products carry ~3 roundings
    // relative to |o - v0|; for a grazing ray the in-plane error grows by 1 / cos but the perpendicular miss does not), so the
    // boxes grow by 1e-5 of the largest coordinate magnitude (~170 ulps; mag >= extent / 2) + 1e-6.  Rounds 1-3 also added 1e-4
    // of the scene's EXTENT -- 0.11 on the 1000-wide terrain whose cells are 0.49 wide: every leaf box 45 % wider per axis, 11.0
    // triangle tests per segment where 6.1 do (C4 45.8 -> 34.6 ms).  The f-1 gates (1e7 rays x {box, duck}, 1e6 + brute force on the
    // 8.4 M-triangle terrain) still pass at 1e-7 of the magnitude and fail at 1e-8 (profiles/r04_c4_visit_slope.json).  Origins far
    // outside the scene are the walk's business: its offsets' margin grows with |origin - o| (pt_walk.h sahq_node_eval).
    float mag = 0.0f;
    for (int k = 0; k < 3; k++) mag = std::max(mag, std::max(fabsf(smn[k]), fabsf(smx[k])));
    const float eps = 1e-5f * mag + 1e-6f;
    // flatten inner nodes to NodePair records in DFS pre-order; a leaf root becomes a pair whose
    // right child is an empty (inverted) box
    std::vector<int> pair_of(b.nodes.size(), -1);
    std::vector<int> order;
    {
        std::vector<int> todo{root};
        while (!todo.empty()) {
            int id = todo.back(); todo.pop_back();
            if (b.nodes[id].left < 0) continue;
            pair_of[id] = (int)order.size(); order.push_back(id);
            todo.push_back(b.nodes[id].right); todo.push_back(b.nodes[id].left);
        }
    }
    auto put_child = [&](NodePair &p, int side, int child) {
        float *mn[3] = {side ? &p.rxmin : &p.lxmin, side ? &p.rymin : &p.lymin, side ? &p.rzmin : &p.lzmin};
        float *mx[3] = {side ? &p.rxmax : &p.lxmax, side ? &p.rymax : &p.lymax, side ? &p.rzmax : &p.lzmax};
        uint32_t &link = side ? p.rlink : p.llink, &cnt = side ? p.rcnt : p.lcnt;
        if (child < 0) {                                        // empty: never hit
            for (int k = 0; k < 3; k++) { *mn[k] = INFINITY; *mx[k] = -INFINITY; }
            link = 0; cnt = 1;
            return;
        }
        const SahNode &c = b.nodes[child];
        for (int k = 0; k < 3; k++) { *mn[k] = c.bmin[k] - eps; *mx[k] = c.bmax[k] + eps; }
        if (c.left < 0) { link = c.first; cnt = c.count; }
        else { link = (uint32_t)pair_of[child]; cnt = 0; }
    };
    if (order.empty()) {
        out.pairs.resize(1);
        put_child(out.pairs[0], 0, root);
        put_child(out.pairs[0], 1, -1);
    } else {
        out.pairs.resize(order.size());
        for (size_t i = 0; i < order.size(); i++) {
            put_child(out.pairs[i], 0, b.nodes[order[i]].left);
            put_child(out.pairs[i], 1, b.nodes[order[i]].right);
        }
    }
    out.order.resize(n_tris);
    for (size_t i = 0; i < n_tris; i++) out.order[i] = tris[i].face;
    // depth of the pair tree = worst-case traversal stack
    std::vector<std::pair<uint32_t, uint32_t>> st{{0u, 1u}};
    while (!st.empty()) {
        auto [pi, d] = st.back(); st.pop_back();
        out.max_stack = std::max(out.max_stack, d);
        const NodePair &p = out.pairs[pi];
        if (p.lcnt == 0) st.push_back({p.llink, d + 1});
        if (p.rcnt == 0) st.push_back({p.rlink, d + 1});
        if (p.lcnt) out.max_leaf = std::max(out.max_leaf, p.lcnt);
        if (p.rcnt) out.max_leaf = std::max(out.max_leaf, p.rcnt);
    }
}


namespace {
struct QuadChild { float b[6]; uint32_t link, cnt; };

struct QuadCollapse {
    const std::vector<NodePair> &pairs;
    std::vector<NodePair> &out;          // two per quad
    bool ok = true;
    QuadCollapse(const std::vector<NodePair> &p, std::vector<NodePair> &o) : pairs(p), out(o) {}

    QuadChild child(uint32_t p, int side) const
    {
        const NodePair &n = pairs[p];
        QuadChild c;
        const float *f = side ? &n.rxmin : &n.lxmin;          // six consecutive floats per side
        for (int k = 0; k < 6; k++) c.b[k] = f[k];
        c.link = side ? n.rlink : n.llink; c.cnt = side ? n.rcnt : n.lcnt;
        return c;
    }
    static float half_area(const QuadChild &c)
    {
        const float ex = c.b[1] - c.b[0], ey = c.b[3] - c.b[2], ez = c.b[5] - c.b[4];
        return ex * ey + ey * ez + ez * ex;
    }
    // Returns the quad's index; *need = the most stack entries a walk below (and including) this quad can hold.
    uint32_t make(uint32_t p, uint32_t *need)
    {
        const uint32_t qi = (uint32_t)(out.size() / 2);
        out.emplace_back(); out.emplace_back();
        QuadChild ch[4];
        int n = 2;
        ch[0] = child(p, 0); ch[1] = child(p, 1);
        while (n < 4) {
            int pick = -1;
            float best = -1.0f;
            for (int i = 0; i < n; i++) {
                if (ch[i].cnt != 0) continue;                  // leaves (and the empty child of a leaf root) stay
                const float a = half_area(ch[i]);
                if (pick < 0 || a > best) { pick = i; best = a; }
            }
            if (pick < 0) break;
            const uint32_t pp = ch[pick].link;
            ch[pick] = child(pp, 0);
            ch[n++] = child(pp, 1);
        }
        // An absent child (fewer than four, or the empty half of a leaf root, whose inverted box the slab test would
        // ACCEPT: every axis interval of (+inf, -inf) comes out as (-inf, +inf)) gets a box with min = max = +inf on
        // every axis: per axis both plane distances are the same infinity, so the entry distance is +inf or the exit
        // distance -inf -- never entered, whatever the ray.
        uint32_t links[4] = {1u, 1u, 1u, 1u};
        bool absent[4] = {true, true, true, true};
        uint32_t deepest = 0;
        int real = 0;
        for (int i = 0; i < n; i++) {
            if (!(ch[i].b[0] <= ch[i].b[1])) continue;         // the empty half of a leaf root
            absent[i] = false;
            real++;
            if (ch[i].cnt == 0) {
                uint32_t sub = 0;
                const uint32_t q = make(ch[i].link, &sub);
                links[i] = q * 128u;
                deepest = std::max(deepest, sub);
            } else {
                if (ch[i].cnt > 15u) ok = false;
                links[i] = ch[i].link * (uint32_t)sizeof(TriIsect) | (ch[i].cnt & 15u);
            }
        }
        float f[24];
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 6; k++) f[6 * i + k] = absent[i] ? INFINITY : ch[i].b[k];
        NodePair &h0 = out[2 * (size_t)qi], &h1 = out[2 * (size_t)qi + 1];
        memcpy(&h0.lxmin, f, 12 * sizeof(float));
        memcpy(&h1.lxmin, f + 12, 12 * sizeof(float));
        h0.llink = links[0]; h0.lcnt = links[1]; h0.rlink = links[2]; h0.rcnt = links[3];
        h1.llink = h1.lcnt = h1.rlink = h1.rcnt = 0;
        *need = (uint32_t)std::max(real - 1, 0) + deepest;
        return qi;
    }
};
} // namespace

bool collapse_sah_quads(const SahBvh &in, std::vector<NodePair> &quads2, uint32_t &max_stack)
{
    static_assert(sizeof(NodePair) == 64, "a QuadNode is two NodePair-sized halves");
    quads2.clear();
    quads2.reserve(in.pairs.size() + 2);
    QuadCollapse c(in.pairs, quads2);
    uint32_t need = 0;
    (void)c.make(0, &need);
    max_stack = need + 1;
    return c.ok && (quads2.size() / 2) * 128ull < (1ull << 32);
}

uint32_t sah_quads_top_levels_first(std::vector<NodePair> &quads2, uint32_t n_top)
{
    const uint32_t nq = (uint32_t)(quads2.size() / 2);
    if (nq == 0 || n_top == 0) return 0;
    std::vector<uint32_t> bfs;                              // old indices, breadth first, at most n_top
    std::vector<uint8_t> taken(nq, 0);
    bfs.push_back(0); taken[0] = 1;
    for (size_t head = 0; head < bfs.size() && bfs.size() < n_top; head++) {
        const NodePair &h0 = quads2[2 * (size_t)bfs[head]];
        const float *box[4] = {&h0.lxmin, &h0.lxmin + 6, &quads2[2 * (size_t)bfs[head] + 1].lxmin, &quads2[2 * (size_t)bfs[head] + 1].lxmin + 6};
        const uint32_t link[4] = {h0.llink, h0.lcnt, h0.rlink, h0.rcnt};
        for (int c = 0; c < 4 && bfs.size() < n_top; c++) {
            if (!(box[c][0] < INFINITY) || (link[c] & 15u)) continue;      // absent, or a leaf
            const uint32_t q = link[c] / 128u;
            if (q < nq && !taken[q]) { taken[q] = 1; bfs.push_back(q); }
        }
    }
    std::vector<uint32_t> perm(nq);                         // old -> new
    uint32_t next = 0;
    for (uint32_t q : bfs) perm[q] = next++;
    for (uint32_t q = 0; q < nq; q++) if (!taken[q]) perm[q] = next++;
    std::vector<NodePair> out(quads2.size());
    for (uint32_t q = 0; q < nq; q++) {
        NodePair h0 = quads2[2 * (size_t)q], h1 = quads2[2 * (size_t)q + 1];
        const float *box[4] = {&h0.lxmin, &h0.lxmin + 6, &h1.lxmin, &h1.lxmin + 6};
        uint32_t *link[4] = {&h0.llink, &h0.lcnt, &h0.rlink, &h0.rcnt};
        for (int c = 0; c < 4; c++)
            if (box[c][0] < INFINITY && !(*link[c] & 15u)) *link[c] = perm[*link[c] / 128u] * 128u;
        out[2 * (size_t)perm[q]] = h0; out[2 * (size_t)perm[q] + 1] = h1;
    }
    quads2.swap(out);
    return (uint32_t)bfs.size();
}

void sah_quads_plane_major(std::vector<NodePair> &quads2)
{
    for (size_t qi = 0; qi + 1 < quads2.size(); qi += 2) {
        const NodePair h0 = quads2[qi], h1 = quads2[qi + 1];
        const float *box[4] = {&h0.lxmin, &h0.lxmin + 6, &h1.lxmin, &h1.lxmin + 6};
        QuadF f;
        for (int c = 0; c < 4; c++) for (int k = 0; k < 6; k++) f.plane[k][c] = box[c][k];
        f.link[0] = h0.llink; f.link[1] = h0.lcnt; f.link[2] = h0.rlink; f.link[3] = h0.rcnt;
        f.pad[0] = f.pad[1] = f.pad[2] = f.pad[3] = 0u;
        memcpy(&quads2[qi], &f, sizeof f);
    }
}

bool quantise_sah_quads(const std::vector<NodePair> &quads2, std::vector<QuadQ> &out)
{
    const size_t nq = quads2.size() / 2;
    out.assign(nq, QuadQ{});
    bool ok = true;
    for (size_t qi = 0; qi < nq; qi++) {
        const NodePair &h0 = quads2[2 * qi], &h1 = quads2[2 * qi + 1];
        const float *box[4] = {&h0.lxmin, &h0.lxmin + 6, &h1.lxmin, &h1.lxmin + 6};
        const uint32_t link[4] = {h0.llink, h0.lcnt, h0.rlink, h0.rcnt};
        QuadQ &q = out[qi];
        bool present[4];
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int c = 0; c < 4; c++) {
            present[c] = box[c][0] < INFINITY;             // absent children carry +inf on all six planes
            if (!present[c]) continue;
            for (int a = 0; a < 3; a++) {
                if (!std::isfinite(box[c][2 * a]) || !std::isfinite(box[c][2 * a + 1])) ok = false;
                mn[a] = std::min(mn[a], box[c][2 * a]); mx[a] = std::max(mx[a], box[c][2 * a + 1]);
            }
        }
        if (!ok) return false;
        float org[3] = {mn[0] < INFINITY ? mn[0] : 0.0f, mn[1] < INFINITY ? mn[1] : 0.0f, mn[2] < INFINITY ? mn[2] : 0.0f};
        uint32_t ex[3];
        for (int a = 0; a < 3; a++) {
            // the smallest power of two whose 255-fold reaches the far side, found against the decode itself
            int e = 1;                                     // biased exponent 1 = 2^-126
            const float ext = mx[a] > org[a] ? mx[a] - org[a] : 0.0f;
            if (ext > 0.0f) {
                int eb;
                (void)std::frexp(ext / 255.0f, &eb);        // ext / 255 = m * 2^eb, m in [0.5, 1): 2^eb >= ext / 255
                e = std::min(std::max(eb + 127, 1), 254);
            }
            while (e < 254 && !(quadq_plane(255u, (uint32_t)e, org[a]) >= mx[a])) e++;
            if (!(quadq_plane(255u, (uint32_t)e, org[a]) >= mx[a])) return false;
            ex[a] = (uint32_t)e;
        }
        q.ox = org[0]; q.oy = org[1]; q.oz = org[2];
        q.exps = ex[0] | ex[1] << 8 | ex[2] << 16;
        for (int c = 0; c < 4; c++) {
            q.lo[c] = q.hi[c] = 0;
            if (!present[c]) { q.link[c] = 0xFFFFFFFFu; continue; }
            // inner links are byte offsets of 128-byte records in quads2: here the records are 64 bytes
            q.link[c] = (link[c] & 15u) ? link[c] : link[c] / 2u;
            uint32_t byte[6];
            for (int a = 0; a < 3; a++) {
                union { uint32_t u; float f; } sc; sc.u = ex[a] << 23;
                const float lo = box[c][2 * a], hi = box[c][2 * a + 1];
                // low plane: the largest byte whose decoded plane is still <= lo; high plane: the smallest with >= hi
                long bl = (long)std::floor((double)(lo - org[a]) / (double)sc.f);
                bl = std::min(std::max(bl, 0l), 255l);
                while (bl > 0 && !(quadq_plane((uint32_t)bl, ex[a], org[a]) <= lo)) bl--;
                while (bl < 255 && quadq_plane((uint32_t)bl + 1u, ex[a], org[a]) <= lo) bl++;
                long bh = (long)std::ceil((double)(hi - org[a]) / (double)sc.f);
                bh = std::min(std::max(bh, 0l), 255l);
                while (bh < 255 && !(quadq_plane((uint32_t)bh, ex[a], org[a]) >= hi)) bh++;
                while (bh > 0 && quadq_plane((uint32_t)bh - 1u, ex[a], org[a]) >= hi) bh--;
                if (!(quadq_plane((uint32_t)bl, ex[a], org[a]) <= lo) || !(quadq_plane((uint32_t)bh, ex[a], org[a]) >= hi)) return false;
                byte[2 * a] = (uint32_t)bl; byte[2 * a + 1] = (uint32_t)bh;
            }
            q.lo[c] = byte[0] | byte[2] << 8 | byte[4] << 16 | byte[1] << 24;
            q.hi[c] = byte[3] | byte[5] << 8;
        }
    }
    return nq * 64ull < (1ull << 32);
}

} // namespace pt
