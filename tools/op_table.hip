// tools/op_table.hip -- instruction counts of the integrator's building blocks with all lanes live, for the algorithmic
// numerator of bench.py's roofline: each kernel below does ONE thing per lane between its loads and its
// stores; tools/op_table.py compiles this file to ISA (same flags as the product), counts the VALU instructions of every
// kernel and subtracts the `base_*` kernel that has the same loads and stores and nothing in between.  Not product code.
#include "pt_walk.h"
#include "pt_shade.h"
using namespace pt;
using namespace pt::dev;

#define LOAD_RAY const float *r = in + 16 * (size_t)threadIdx.x; const v3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]);

// (the baselines load what the op kernels load and store it untouched: address arithmetic only)
extern "C" __global__ void base_ray(const float *in, float *out)
{
    const float4 *r4 = reinterpret_cast<const float4 *>(in) + 4 * (size_t)threadIdx.x;
    float4 *o4 = reinterpret_cast<float4 *>(out) + 4 * (size_t)threadIdx.x;
    o4[0] = r4[0]; o4[1] = r4[1]; o4[2] = r4[2]; o4[3] = r4[3];
}
extern "C" __global__ void op_box(const float *in, float *out)                  // aabb::hit, aabb.h:38-66
{
    LOAD_RAY
    const v3 inv = mk3(r[6], r[7], r[8]);
    out[threadIdx.x] = box_hit(r[9], r[10], r[11], r[12], r[13], r[14], o, inv) ? d.x : 0.0f;
}
extern "C" __global__ void op_tri(const float *in, float *out)                  // triangle::hit, triangle.h:63-113
{
    LOAD_RAY
    float t, u, v;
    const bool ok = tri_hit(mk3(r[6], r[7], r[8]), mk3(r[9], r[10], r[11]), mk3(r[12], r[13], r[14]), o, d, 0.001f, r[15], t, u, v);
    out[threadIdx.x] = ok ? t + u + v : 0.0f;
}
extern "C" __global__ void op_light(const float *in, float *out)                // triangle::pdf_value, triangle.h:32-40 + hitable_list.h:16-22
{
    LOAD_RAY
    float t, u, v;
    float pdfv = 0.0f;
    const v3 ln = mk3(r[6], r[9], r[12]);
    if (tri_hit(mk3(r[6], r[7], r[8]), mk3(r[9], r[10], r[11]), mk3(r[12], r[13], r[14]), o, d, 0.001f, 3.402823466e+38f, t, u, v)) {
        const float distance_squared = t * t * dot(d, d);
        const float cosine = fabsf(dot(d, ln) / length(d));
        pdfv = distance_squared / (cosine * r[15]);
    }
    out[threadIdx.x] = r[15] * pdfv;
}
extern "C" __global__ void base_state(RenderParams P, PathState *st, Hit *hits)
{
    PathState L = st[threadIdx.x];
    L.bounce = (unsigned)hits[threadIdx.x].slot;
    st[threadIdx.x] = L;
}
extern "C" __global__ void op_shade(RenderParams P, PathState *st, Hit *hits)   // camera.h:55-80 after BVH::hit: every branch, one light-loop body
{
    extern __shared__ float4 lds[];
    ShadeView V;
    V.shade = reinterpret_cast<const float4 *>(P.shade); V.mats = reinterpret_cast<const float4 *>(P.mats); V.lights = reinterpret_cast<const float4 *>(P.lights); V.end = lds;
    PathState L = st[threadIdx.x];
    shade<false, ShadeView>(L, P, V, hits[threadIdx.x]);
    st[threadIdx.x] = L;
}
extern "C" __global__ void op_start(RenderParams P, PathState *st, Hit *hits)   // DevicePathTracer.h:84-87 + camera::get_ray
{
    PathState L = st[threadIdx.x];
    (void)start_path(L, P, (int)threadIdx.x & 7, (int)threadIdx.x >> 3);
    L.bounce += (unsigned)hits[threadIdx.x].slot;
    st[threadIdx.x] = L;
}
extern "C" __global__ void op_tex(RenderParams P, const float *in, float *out)  // BaseColorTexture::value, Texture.h:23-77
{
    const v3 c = tex_value(P, (int)in[2], in[0], in[1]);
    out[threadIdx.x] = c.x + c.y + c.z;
}
extern "C" __global__ void op_rng(PathState *st, float *out)                    // one curand_uniform
{
    PathState L = st[threadIdx.x];
    out[threadIdx.x] = xorwow_uniform(L.rng);
    st[threadIdx.x] = L;
}
extern "C" __global__ void base_rng(PathState *st, float *out)
{
    PathState L = st[threadIdx.x];
    out[threadIdx.x] = L.col.x;
    st[threadIdx.x] = L;
}
