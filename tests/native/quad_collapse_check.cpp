// CPU check of the 4-wide device form of the opt-in SAH tree (pt_bvh.cpp collapse_sah_quads) -- test infrastructure,
// built and run by tests/test_host_logic.py with plain g++ against the product's host sources.
//
//   quad_collapse_check <n_triangles> <seed>
//
// 1. structure: every QuadNode is reachable exactly once from the root, every leaf-order slot is covered exactly once,
//    links are well-formed (inner: multiple of 128 inside the array; leaf: multiple of 48 | count 1..15), absent children
//    have the never-entered box, and no walk order can push more entries than the reported bound;
// 2. the walk of pt_walk.h (sahw_node_step / sahw_tri_step: nearest child first, the others pushed as link words),
//    restated here on the host, finds the brute-force closest hit for random rays without exceeding that bound.
// Exit code 0 = all good; prints one summary line.
#include "pt_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <vector>

using namespace pt;

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct Quad {
    const float *box[4];
    uint32_t link[4];
};

Quad quad_at(const std::vector<NodePair> &q, uint32_t qi)
{
    const NodePair &h0 = q[2 * (size_t)qi], &h1 = q[2 * (size_t)qi + 1];
    Quad r;
    r.box[0] = &h0.lxmin; r.box[1] = &h0.lxmin + 6; r.box[2] = &h1.lxmin; r.box[3] = &h1.lxmin + 6;
    r.link[0] = h0.llink; r.link[1] = h0.lcnt; r.link[2] = h0.rlink; r.link[3] = h0.rcnt;
    return r;
}

bool absent(const float *b) { return !(b[0] < INFINITY); }

// plain Moller-Trumbore on raw vertices (the device's exact arithmetic is not the subject here)
bool hit_triangle(const float *p, const float *o, const float *d, float &t)
{
    float e1[3], e2[3], h[3], s[3], q[3];
    for (int k = 0; k < 3; k++) { e1[k] = p[3 + k] - p[k]; e2[k] = p[6 + k] - p[k]; }
    h[0] = d[1] * e2[2] - d[2] * e2[1]; h[1] = d[2] * e2[0] - d[0] * e2[2]; h[2] = d[0] * e2[1] - d[1] * e2[0];
    const float a = e1[0] * h[0] + e1[1] * h[1] + e1[2] * h[2];
    if (std::fabs(a) < 1e-12f) return false;
    const float f = 1.0f / a;
    for (int k = 0; k < 3; k++) s[k] = o[k] - p[k];
    const float u = f * (s[0] * h[0] + s[1] * h[1] + s[2] * h[2]);
    if (u < 0 || u > 1) return false;
    q[0] = s[1] * e1[2] - s[2] * e1[1]; q[1] = s[2] * e1[0] - s[0] * e1[2]; q[2] = s[0] * e1[1] - s[1] * e1[0];
    const float v = f * (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]);
    if (v < 0 || u + v > 1) return false;
    t = f * (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]);
    return t > 0.001f;
}

} // namespace

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 3000;
    const unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 5u;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-10, 10), S(-0.5, 0.5), D(-1, 1);
    std::vector<float> pos(9 * n);
    for (size_t i = 0; i < n; i++) {
        const float c[3] = {U(rng), U(rng), U(rng)};
        for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) pos[9 * i + 3 * v + k] = c[k] + S(rng);
    }
    SahBvh b;
    build_sah_bvh(pos.data(), n, b);
    std::vector<NodePair> q;
    uint32_t bound = 0;
    if (!collapse_sah_quads(b, q, bound)) { printf("collapse refused the tree\n"); return 1; }
    const size_t nq = q.size() / 2;
    // optional third argument: move that many top-level records (breadth first) to the front (sah_quads_top_levels_first, the
    // layout of the PTAMD_SAH_TOP option) -- every check below must hold for the renumbered tree as well
    if (argc > 3) {
        const uint32_t want = (uint32_t)atol(argv[3]);
        const uint32_t moved = sah_quads_top_levels_first(q, want);
        if (moved > want || moved > nq || (nq >= want && moved != want)) { printf("top-levels-first moved %u of %u asked (%zu quads)\n", moved, want, nq); return 1; }
        // the root stays the root; every record of the prefix but the root is the child of an EARLIER record (breadth first)
        std::vector<int> reach(nq, 0);
        reach[0] = 1;
        for (uint32_t i = 0; i < moved; i++) {
            if (!reach[i]) { printf("top-levels-first: record %u of the prefix is not below an earlier one\n", i); return 1; }
            const Quad Q = quad_at(q, i);
            for (int c = 0; c < 4; c++) if (!absent(Q.box[c]) && !(Q.link[c] & 15u) && Q.link[c] / 128u < nq) reach[Q.link[c] / 128u] = 1;
        }
    }

    // 1. structure
    std::vector<int> slot_seen(n, 0), quad_seen(nq, 0);
    struct Item { uint32_t link, pushed; };
    std::vector<Item> todo{{0u, 0u}};
    uint32_t worst_possible = 0;
    while (!todo.empty()) {
        const Item it = todo.back(); todo.pop_back();
        if (it.link & 15u) {
            const uint32_t off = it.link & ~15u, cnt = it.link & 15u;
            if (off % 48u) { printf("leaf link %u is not a TriIsect offset\n", it.link); return 1; }
            for (uint32_t k = 0; k < cnt; k++) {
                if (off / 48u + k >= n) { printf("leaf link %u leaves the triangle array\n", it.link); return 1; }
                slot_seen[off / 48u + k]++;
            }
            continue;
        }
        if (it.link % 128u || it.link / 128u >= nq) { printf("inner link %u is not a QuadNode offset\n", it.link); return 1; }
        const uint32_t qi = it.link / 128u;
        quad_seen[qi]++;
        const Quad Q = quad_at(q, qi);
        int real = 0;
        for (int i = 0; i < 4; i++) {
            if (absent(Q.box[i])) {
                for (int k = 0; k < 6; k++) if (Q.box[i][k] != INFINITY) { printf("absent child with a finite plane\n"); return 1; }
                continue;
            }
            if (!(Q.box[i][0] <= Q.box[i][1])) { printf("inverted box that the slab test would accept\n"); return 1; }
            real++;
        }
        const uint32_t pushed = it.pushed + (uint32_t)std::max(real - 1, 0);
        worst_possible = std::max(worst_possible, pushed);
        for (int i = 0; i < 4; i++) if (!absent(Q.box[i])) todo.push_back({Q.link[i], pushed});
    }
    for (size_t i = 0; i < n; i++) if (slot_seen[i] != 1) { printf("slot %zu covered %d times\n", i, slot_seen[i]); return 1; }
    for (size_t i = 0; i < nq; i++) if (quad_seen[i] != 1) { printf("quad %zu reached %d times\n", i, quad_seen[i]); return 1; }
    if (worst_possible + 1 != bound) { printf("stack bound %u, but a walk can hold %u + 1\n", bound, worst_possible); return 1; }

    // 2. the walk against brute force
    const int rays = n <= 20000 ? 20000 : 2000;
    uint32_t deepest = 0;
    double visits = 0;
    std::vector<uint32_t> stack(bound + 1);
    for (int r = 0; r < rays; r++) {
        const float o[3] = {U(rng), U(rng), U(rng)}, d[3] = {D(rng), D(rng), D(rng)};
        const float inv[3] = {1 / d[0], 1 / d[1], 1 / d[2]};
        float t_brute = std::numeric_limits<float>::max();
        int best_brute = -1;
        for (size_t i = 0; i < n; i++) {
            float t;
            if (hit_triangle(&pos[9 * (size_t)b.order[i]], o, d, t) && t < t_brute) { t_brute = t; best_brute = (int)i; }
        }
        uint32_t pair = 0, k0 = 0, n0 = 0;
        int sp = 0, best = -1;
        float tbest = std::numeric_limits<float>::max();
        auto take = [&](uint32_t x) { if (x & 15u) { k0 = x & ~15u; n0 = x & 15u; } else pair = x; };
        auto pop = [&]() { if (sp > 0) take(stack[--sp]); };
        while (n0 != 0 || pair != kNone) {
            if (n0 != 0) {
                const uint32_t s = k0 / 48u;
                float t;
                if (hit_triangle(&pos[9 * (size_t)b.order[s]], o, d, t) && t < tbest) { tbest = t; best = (int)s; }
                k0 += 48u; n0--;
                if (n0 == 0 && pair == kNone) pop();
                continue;
            }
            visits++;
            const Quad Q = quad_at(q, pair / 128u);
            float key[4];
            uint32_t link[4];
            for (int i = 0; i < 4; i++) {
                const float *f = Q.box[i];
                const float t1 = (f[0] - o[0]) * inv[0], t2 = (f[1] - o[0]) * inv[0], t3 = (f[2] - o[1]) * inv[1], t4 = (f[3] - o[1]) * inv[1];
                const float t5 = (f[4] - o[2]) * inv[2], t6 = (f[5] - o[2]) * inv[2];
                const float tn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
                const float tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
                key[i] = ((tf >= fmaxf(tn, 0.0f)) && (tn <= tbest)) ? tn : INFINITY;
                link[i] = Q.link[i];
            }
            for (int j = 1; j < 4; j++) if (key[j] < key[0]) { std::swap(key[0], key[j]); std::swap(link[0], link[j]); }
            pair = kNone;
            for (int j = 3; j >= 1; j--) {
                if (!(key[j] < INFINITY)) continue;
                if ((uint32_t)sp >= bound) { printf("walk pushed past the bound (%u entries)\n", bound); return 1; }
                stack[sp++] = link[j];
            }
            deepest = std::max(deepest, (uint32_t)sp);
            if (key[0] < INFINITY) take(link[0]); else pop();
        }
        const bool same = best == best_brute || (best >= 0 && best_brute >= 0 && std::fabs(tbest - t_brute) <= 1e-6f * std::fabs(t_brute));
        if (!same) { printf("ray %d: walk found slot %d (t %g), brute force %d (t %g)\n", r, best, tbest, best_brute, t_brute); return 1; }
    }
    // 3. the 64-byte form (quantise_sah_quads): same topology, every decoded child box CONTAINS the box it stands for (decoded
    //    with the device's own expression, quadq_plane), and is not looser than two units per plane.
    std::vector<QuadQ> qq;
    if (!quantise_sah_quads(q, qq) || qq.size() != nq) { printf("quantise refused the tree\n"); return 1; }
    double slack = 0.0, extent = 0.0;
    for (size_t qi = 0; qi < nq; qi++) {
        const Quad Q = quad_at(q, (uint32_t)qi);
        const QuadQ &z = qq[qi];
        const float org[3] = {z.ox, z.oy, z.oz};
        const uint32_t ex[3] = {z.exps & 0xffu, (z.exps >> 8) & 0xffu, (z.exps >> 16) & 0xffu};
        for (int i = 0; i < 4; i++) {
            if (absent(Q.box[i])) { if (z.link[i] != kNone) { printf("absent child with a link in the 64-byte form\n"); return 1; } continue; }
            const uint32_t want = (Q.link[i] & 15u) ? Q.link[i] : Q.link[i] / 2u;
            if (z.link[i] != want) { printf("quad %zu child %d: link %u, expected %u\n", qi, i, z.link[i], want); return 1; }
            const uint32_t by[6] = {z.lo[i] & 0xffu, z.lo[i] >> 24, (z.lo[i] >> 8) & 0xffu, z.hi[i] & 0xffu, (z.lo[i] >> 16) & 0xffu, (z.hi[i] >> 8) & 0xffu};
            for (int a = 0; a < 3; a++) {
                const float lo = quadq_plane(by[2 * a], ex[a], org[a]), hi = quadq_plane(by[2 * a + 1], ex[a], org[a]);
                union { uint32_t u; float f; } unit; unit.u = ex[a] << 23;
                if (!(lo <= Q.box[i][2 * a]) || !(hi >= Q.box[i][2 * a + 1])) { printf("quad %zu child %d axis %d: decoded [%g, %g] does not contain [%g, %g]\n", qi, i, a, lo, hi, Q.box[i][2 * a], Q.box[i][2 * a + 1]); return 1; }
                if (Q.box[i][2 * a] - lo > 2.0f * unit.f + 1e-6f * std::fabs(lo) || hi - Q.box[i][2 * a + 1] > 2.0f * unit.f + 1e-6f * std::fabs(hi)) { printf("quad %zu child %d axis %d: looser than two units\n", qi, i, a); return 1; }
                slack += (Q.box[i][2 * a] - lo) + (hi - Q.box[i][2 * a + 1]);
                extent += Q.box[i][2 * a + 1] - Q.box[i][2 * a];
            }
        }
    }
    // 4. the ray parameters the device computes on the 64-byte form (quadq_ray / quadq_child_span = pt_walk.h sahq_node_eval:
    //    slope and offset per axis, near / far byte by the sign of inv, offsets moved outwards by 2^-19):
    //    a. against the EXACT parameters of the exact planes origin + byte * unit (128-bit arithmetic): the near one is never
    //       above, the far one never below, by more than the half unit in the last place of the final fma -- for origins up
    //       to a million scene extents away and slopes up to the clamp, axis-parallel rays included;
    //    b. the walk with these parameters finds the brute-force closest hit.
    {
        std::uniform_real_distribution<float> P(0.0f, 1.0f);
        const int span_rays = 4000;
        long checked = 0;
        for (int r = 0; r < span_rays; r++) {
            const float far = r % 4 == 3 ? std::pow(10.0f, 6.0f * P(rng)) : 1.0f;       // every fourth origin far outside
            float o[3] = {U(rng) * far, U(rng) * far, U(rng) * far}, d[3] = {D(rng), D(rng), D(rng)};
            if (r % 5 == 0) d[r % 3] = (r & 8) ? 0.0f : -0.0f;                               // exactly axis-parallel
            if (r % 7 == 0) d[(r + 1) % 3] = 1e-38f * D(rng);                               // a slope beyond the clamp
            const QuadQ &z = qq[(size_t)(P(rng) * (float)nq) % nq];
            if (r % 11 == 0) o[r % 3] = (&z.ox)[r % 3];                                     // the origin exactly on the grid's origin plane
            const float inv[3] = {quadq_clamp_inv(1.0f / d[0]), quadq_clamp_inv(1.0f / d[1]), quadq_clamp_inv(1.0f / d[2])};
            const QuadqRay R = quadq_ray(z, o, inv);
            const float org[3] = {z.ox, z.oy, z.oz};
            for (int c = 0; c < 4; c++) {
                if (z.link[c] == kNone) continue;
                const uint32_t lo[3] = {z.lo[c] & 0xffu, (z.lo[c] >> 8) & 0xffu, (z.lo[c] >> 16) & 0xffu};
                const uint32_t hi[3] = {z.lo[c] >> 24, z.hi[c] & 0xffu, (z.hi[c] >> 8) & 0xffu};
                for (int k = 0; k < 3; k++) {
                    union { uint32_t u; float f; } unit; unit.u = ((z.exps >> (8 * k)) & 0xffu) << 23;
                    const uint32_t bn = R.neg[k] ? hi[k] : lo[k], bf = R.neg[k] ? lo[k] : hi[k];
                    const float tn = __builtin_fmaf((float)bn, R.a[k], R.near_off[k]), tf = __builtin_fmaf((float)bf, R.a[k], R.far_off[k]);
                    const __float128 en = ((__float128)org[k] + (__float128)bn * (__float128)unit.f - (__float128)o[k]) * (__float128)inv[k];
                    const __float128 ef = ((__float128)org[k] + (__float128)bf * (__float128)unit.f - (__float128)o[k]) * (__float128)inv[k];
                    if (!std::isfinite(tn) || !std::isfinite(tf)) { printf("span: non-finite parameter (ray %d)\n", r); return 1; }
                    const __float128 half_ulp = (__float128)5.9604644775390625e-08;          // 2^-24
                    const __float128 slack_n = (en < 0 ? -en : en) * half_ulp + (__float128)1e-45, slack_f = (ef < 0 ? -ef : ef) * half_ulp + (__float128)1e-45;
                    if ((__float128)tn > en + slack_n) { printf("span: near parameter %g above the exact %g (ray %d axis %d)\n", tn, (double)en, r, k); return 1; }
                    if ((__float128)tf < ef - slack_f) { printf("span: far parameter %g below the exact %g (ray %d axis %d)\n", tf, (double)ef, r, k); return 1; }
                    checked++;
                }
            }
        }
        const int qrays = n <= 20000 ? 5000 : 500;
        std::vector<uint32_t> st(bound + 1);
        for (int r = 0; r < qrays; r++) {
            float o[3] = {U(rng), U(rng), U(rng)}, d[3] = {D(rng), D(rng), D(rng)};
            if (r % 5 == 0) d[r % 3] = 0.0f;
            const float inv[3] = {quadq_clamp_inv(1.0f / d[0]), quadq_clamp_inv(1.0f / d[1]), quadq_clamp_inv(1.0f / d[2])};
            float t_brute = std::numeric_limits<float>::max();
            int best_brute = -1;
            for (size_t i = 0; i < n; i++) {
                float t;
                if (hit_triangle(&pos[9 * (size_t)b.order[i]], o, d, t) && t < t_brute) { t_brute = t; best_brute = (int)i; }
            }
            uint32_t node = 0, k0 = 0, n0 = 0;
            int sp = 0, best = -1;
            float tbest = std::numeric_limits<float>::max();
            auto take = [&](uint32_t x) { if (x & 15u) { k0 = x & ~15u; n0 = x & 15u; } else node = x; };
            auto pop = [&]() { if (sp > 0) take(st[--sp]); };
            while (n0 != 0 || node != kNone) {
                if (n0 != 0) {
                    const uint32_t s = k0 / 48u;
                    float t;
                    if (hit_triangle(&pos[9 * (size_t)b.order[s]], o, d, t) && t < tbest) { tbest = t; best = (int)s; }
                    k0 += 48u; n0--;
                    if (n0 == 0 && node == kNone) pop();
                    continue;
                }
                const QuadQ &z = qq[node / 64u];
                const QuadqRay R = quadq_ray(z, o, inv);
                float key[4];
                uint32_t link[4];
                for (int i = 0; i < 4; i++) {
                    float tn, tf;
                    quadq_child_span(z, i, R, tn, tf);
                    key[i] = (z.link[i] != kNone && tf >= fmaxf(tn, 0.0f) && tn <= tbest) ? tn : INFINITY;
                    link[i] = z.link[i];
                }
                for (int j = 1; j < 4; j++) if (key[j] < key[0]) { std::swap(key[0], key[j]); std::swap(link[0], link[j]); }
                node = kNone;
                for (int j = 3; j >= 1; j--) {
                    if (!(key[j] < INFINITY)) continue;
                    if ((uint32_t)sp >= bound) { printf("64-byte walk pushed past the bound (%u entries)\n", bound); return 1; }
                    st[sp++] = link[j];
                }
                if (key[0] < INFINITY) take(link[0]); else pop();
            }
            const bool same = best == best_brute || (best >= 0 && best_brute >= 0 && std::fabs(tbest - t_brute) <= 1e-6f * std::fabs(t_brute));
            if (!same) { printf("64-byte walk, ray %d: found slot %d (t %g), brute force %d (t %g)\n", r, best, tbest, best_brute, t_brute); return 1; }
        }
        if (checked == 0 && nq > 0 && n > 4) { printf("span check saw no plane\n"); return 1; }
    }
    // 5. the plane-major float form (sah_quads_plane_major / quadf_child_span = pt_walk.h sahw_node_step): same links, the
    //    planes transposed; its ray parameters bracket the exact (plane - o) * inv of every plane like the 64-byte form's, and
    //    the walk with them finds the brute-force closest hit.
    {
        std::vector<NodePair> pm = q;
        sah_quads_plane_major(pm);
        const QuadF *F = reinterpret_cast<const QuadF *>(pm.data());
        for (size_t qi = 0; qi < nq; qi++) {
            const Quad Q = quad_at(q, (uint32_t)qi);
            for (int c = 0; c < 4; c++) {
                if (F[qi].link[c] != Q.link[c]) { printf("plane-major: quad %zu child %d link changed\n", qi, c); return 1; }
                for (int k = 0; k < 6; k++) if (!(F[qi].plane[k][c] == Q.box[c][k])) { printf("plane-major: quad %zu child %d plane %d changed\n", qi, c, k); return 1; }
            }
        }
        std::uniform_real_distribution<float> P(0.0f, 1.0f);
        for (int r = 0; r < 4000; r++) {
            const float far = r % 4 == 3 ? std::pow(10.0f, 6.0f * P(rng)) : 1.0f;
            float o[3] = {U(rng) * far, U(rng) * far, U(rng) * far}, d[3] = {D(rng), D(rng), D(rng)};
            if (r % 5 == 0) d[r % 3] = (r & 8) ? 0.0f : -0.0f;
            if (r % 7 == 0) d[(r + 1) % 3] = 1e-38f * D(rng);
            const QuadF &z = F[(size_t)(P(rng) * (float)nq) % nq];
            if (r % 11 == 0) o[r % 3] = z.plane[2 * (r % 3)][0] < INFINITY ? z.plane[2 * (r % 3)][0] : 0.0f;      // the origin exactly on a plane
            const float inv[3] = {quadq_clamp_inv(1.0f / d[0]), quadq_clamp_inv(1.0f / d[1]), quadq_clamp_inv(1.0f / d[2])};
            for (int c = 0; c < 4; c++) {
                if (!(z.plane[0][c] < INFINITY)) {
                    float tn, tf;
                    quadf_child_span(z, c, o, inv, tn, tf);
                    if (tf >= fmaxf(tn, 0.0f) && tn <= std::numeric_limits<float>::max()) { printf("plane-major: an absent child was accepted\n"); return 1; }
                    continue;
                }
                for (int k = 0; k < 3; k++) {
                    const float off = -(o[k] * inv[k]);
                    const float near_off = __builtin_fmaf(-std::fabs(off), 1.9073486328125e-06f, off), far_off = __builtin_fmaf(std::fabs(off), 1.9073486328125e-06f, off);
                    const bool neg = std::signbit(inv[k]);
                    const float pn = z.plane[2 * k + (neg ? 1 : 0)][c], pf = z.plane[2 * k + (neg ? 0 : 1)][c];
                    const float tn = __builtin_fmaf(pn, inv[k], near_off), tf = __builtin_fmaf(pf, inv[k], far_off);
                    const __float128 en = ((__float128)pn - (__float128)o[k]) * (__float128)inv[k], ef = ((__float128)pf - (__float128)o[k]) * (__float128)inv[k];
                    const __float128 half_ulp = (__float128)5.9604644775390625e-08;
                    if (!std::isfinite(tn) || !std::isfinite(tf)) { printf("plane-major span: non-finite parameter (ray %d)\n", r); return 1; }
                    if ((__float128)tn > en + (en < 0 ? -en : en) * half_ulp + (__float128)1e-45) { printf("plane-major span: near parameter %g above the exact %g (ray %d axis %d)\n", tn, (double)en, r, k); return 1; }
                    if ((__float128)tf < ef - (ef < 0 ? -ef : ef) * half_ulp - (__float128)1e-45) { printf("plane-major span: far parameter %g below the exact %g (ray %d axis %d)\n", tf, (double)ef, r, k); return 1; }
                }
            }
        }
        const int frays = n <= 20000 ? 5000 : 500;
        std::vector<uint32_t> st(bound + 1);
        for (int r = 0; r < frays; r++) {
            float o[3] = {U(rng), U(rng), U(rng)}, d[3] = {D(rng), D(rng), D(rng)};
            if (r % 5 == 0) d[r % 3] = 0.0f;
            const float inv[3] = {quadq_clamp_inv(1.0f / d[0]), quadq_clamp_inv(1.0f / d[1]), quadq_clamp_inv(1.0f / d[2])};
            float t_brute = std::numeric_limits<float>::max();
            int best_brute = -1;
            for (size_t i = 0; i < n; i++) {
                float t;
                if (hit_triangle(&pos[9 * (size_t)b.order[i]], o, d, t) && t < t_brute) { t_brute = t; best_brute = (int)i; }
            }
            uint32_t node = 0, k0 = 0, n0 = 0;
            int sp = 0, best = -1;
            float tbest = std::numeric_limits<float>::max();
            auto take = [&](uint32_t x) { if (x & 15u) { k0 = x & ~15u; n0 = x & 15u; } else node = x; };
            auto pop = [&]() { if (sp > 0) take(st[--sp]); };
            while (n0 != 0 || node != kNone) {
                if (n0 != 0) {
                    const uint32_t s = k0 / 48u;
                    float t;
                    if (hit_triangle(&pos[9 * (size_t)b.order[s]], o, d, t) && t < tbest) { tbest = t; best = (int)s; }
                    k0 += 48u; n0--;
                    if (n0 == 0 && node == kNone) pop();
                    continue;
                }
                const QuadF &z = F[node / 128u];
                float key[4];
                uint32_t link[4];
                for (int i = 0; i < 4; i++) {
                    float tn, tf;
                    quadf_child_span(z, i, o, inv, tn, tf);
                    key[i] = (tf >= fmaxf(tn, 0.0f) && tn <= tbest) ? tn : INFINITY;
                    link[i] = z.link[i];
                }
                for (int j = 1; j < 4; j++) if (key[j] < key[0]) { std::swap(key[0], key[j]); std::swap(link[0], link[j]); }
                node = kNone;
                for (int j = 3; j >= 1; j--) {
                    if (!(key[j] < INFINITY)) continue;
                    if ((uint32_t)sp >= bound) { printf("plane-major walk pushed past the bound (%u entries)\n", bound); return 1; }
                    st[sp++] = link[j];
                }
                if (key[0] < INFINITY) take(link[0]); else pop();
            }
            const bool same = best == best_brute || (best >= 0 && best_brute >= 0 && std::fabs(tbest - t_brute) <= 1e-6f * std::fabs(t_brute));
            if (!same) { printf("plane-major walk, ray %d: found slot %d (t %g), brute force %d (t %g)\n", r, best, tbest, best_brute, t_brute); return 1; }
        }
    }
    printf("ok: %zu triangles, %zu pairs -> %zu quads, stack bound %u (pair form %u), deepest in %d rays %u, %.1f visits per ray; 64-byte form adds %.2f %% to the boxes' extents\n",
           n, b.pairs.size(), nq, bound, b.max_stack, rays, deepest, visits / rays, extent > 0 ? 100.0 * slack / extent : 0.0);
    return 0;
}
