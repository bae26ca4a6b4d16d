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

namespace pt {
namespace {

struct Builder {
    const float *pos;                 // 9 floats per triangle
    std::vector<float> cx, cy, cz;    // triangle.h:27 centroid = (v0+v1+v2)*0.3333f
    RefBvh &out;
    uint32_t nodesUsed = 1;

    explicit Builder(const float *p, size_t n, RefBvh &o) : pos(p), out(o)
    {
        cx.resize(n); cy.resize(n); cz.resize(n);
        for (size_t i = 0; i < n; i++) {
            const float *t = pos + 9 * i;
            cx[i] = ((t[0] + t[3]) + t[6]) * 0.3333f;
            cy[i] = ((t[1] + t[4]) + t[7]) * 0.3333f;
            cz[i] = ((t[2] + t[5]) + t[8]) * 0.3333f;
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
    // aabb::extend, aabb.h:80-88, for the three vertices of triangle `tri`.
    void box_extend_tri(RefNode &b, int tri) const
    {
        const float *t = pos + 9 * (size_t)tri;
        for (int k = 0; k < 3; k++) {
            b.xmin = fminf(t[3 * k + 0], b.xmin); b.ymin = fminf(t[3 * k + 1], b.ymin);
            b.zmin = fminf(t[3 * k + 2], b.zmin);
            b.xmax = fmaxf(t[3 * k + 0], b.xmax); b.ymax = fmaxf(t[3 * k + 1], b.ymax);
            b.zmax = fmaxf(t[3 * k + 2], b.zmax);
        }
    }
    static float box_area(const RefNode &b)          // aabb.h:75-78 (half area)
    {
        float ex = b.xmax - b.xmin, ey = b.ymax - b.ymin, ez = b.zmax - b.zmin;
        return ex * ey + ey * ez + ez * ex;
    }
    float centroid(int tri, int axis) const
    {
        return axis == 0 ? cx[tri] : (axis == 1 ? cy[tri] : cz[tri]);
    }
    void update_bounds(uint32_t idx)                  // bvh.h:96-106
    {
        RefNode &n = out.nodes[idx];
        box_default(n);
        for (uint32_t i = 0; i < n.triCount; i++) box_extend_tri(n, out.indices[n.firstTri + i]);
    }
    float evaluate_sah(const RefNode &n, int axis, float p) const     // bvh.h:144-176
    {
        RefNode lb, rb;
        box_default(lb); box_default(rb);
        int lc = 0, rc = 0;
        for (uint32_t i = 0; i < n.triCount; i++) {
            int tri = out.indices[n.firstTri + i];
            if (centroid(tri, axis) < p) { lc++; box_extend_tri(lb, tri); }
            else                         { rc++; box_extend_tri(rb, tri); }
        }
        float cost = (float)lc * box_area(lb) + (float)rc * box_area(rb);
        return cost > 0 ? cost : 1e30f;
    }
    void subdivide(uint32_t idx, uint32_t level)      // bvh.h:35-94
    {
        if (level > out.depth) out.depth = level;
        {
            RefNode &n = out.nodes[idx];
            if (n.triCount <= 5) return;
        }
        RefNode n = out.nodes[idx];
        int axis = 0;
        float split = 0.0f, best = 1e30f;
        for (int a = 0; a < 3; a++) {                 // bvh.h:108-137
            float bmin = a == 0 ? n.xmin : (a == 1 ? n.ymin : n.zmin);
            float bmax = a == 0 ? n.xmax : (a == 1 ? n.ymax : n.zmax);
            if (bmin == bmax) continue;
            float scale = (bmax - bmin) / 4;
            for (uint32_t i = 1; i < 4; i++) {
                float cand = bmin + (float)i * scale;
                float cost = evaluate_sah(n, a, cand);
                if (cost < best) { split = cand; axis = a; best = cost; }
            }
        }
        float nosplit = (float)n.triCount * box_area(n);              // bvh.h:139-142
        if (best >= nosplit) return;
        int i = (int)n.firstTri;
        int j = i + (int)n.triCount - 1;
        while (i <= j) {                               // bvh.h:56-72
            if (centroid(out.indices[i], axis) < split) i++;
            else std::swap(out.indices[i], out.indices[j--]);
        }
        int leftCount = i - (int)n.firstTri;
        if (leftCount == 0 || leftCount == (int)n.triCount) return;
        uint32_t l = nodesUsed++, r = nodesUsed++;
        RefNode &ln = out.nodes[l];
        RefNode &rn = out.nodes[r];
        ln.left = ln.right = 0; rn.left = rn.right = 0;
        ln.firstTri = n.firstTri;  ln.triCount = (uint32_t)leftCount;
        rn.firstTri = (uint32_t)i; rn.triCount = n.triCount - (uint32_t)leftCount;
        out.nodes[idx].left = l; out.nodes[idx].right = r; out.nodes[idx].triCount = 0;
        update_bounds(l);
        update_bounds(r);
        subdivide(l, level + 1);
        subdivide(r, level + 1);
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
    out.nodes.assign(2 * n_tris - 1, RefNode());
    out.indices.resize(n_tris);
    for (size_t i = 0; i < n_tris; i++) out.indices[i] = (int)i;
    Builder b(positions9, n_tris, out);
    RefNode &root = out.nodes[0];
    root.left = root.right = 0;
    root.firstTri = 0;
    root.triCount = (uint32_t)n_tris;
    b.update_bounds(0);
    b.subdivide(0, 0);
    out.nodes.resize(b.nodesUsed);
    for (const RefNode &n : out.nodes) {
        if (n.triCount != 0) {
            out.leaves++;
            out.max_leaf = std::max(out.max_leaf, n.triCount);
        }
    }
    out.max_stack = b.stack_need(0, 1);
}

} // namespace pt
