// pt_math.h -- arithmetic contract of the hot path, shared by the product's host code
// (camera set-up, BVH build) and its gfx950 kernels.  Product code: never includes oracle/.
//
// Contract (DESIGN.md "Arithmetic contract"): every + - * / sqrt below is ONE correctly
// rounded IEEE-754 operation in the order the reference's source text prescribes
// (helper_math.h operators), compiled with -ffp-contract=off; rsqrtf(x) = 1.0f/sqrtf(x)
// (helper_math.h:80-83); sin/cos/tan = pt_sincosf (explicit fmaf polynomial).
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

#if defined(__FAST_MATH__)
#error "the path tracer's arithmetic contract forbids -ffast-math"
#endif

namespace pt {

// Device-only short forms of two divisions of the contract.  Each returns the SAME bits as the IEEE
// expression in its comment for EVERY float input -- proven by running all 2^32 bit patterns through
// both on the GPU (pt_kat_exhaustive, tests/test_gpu_parity.py) -- at under half the instructions of
// the compiler's division sequence (v_div_scale x2, v_rcp, 5 fma/mul, v_div_fmas, v_div_fixup).
#if defined(__HIPCC__)
// Wave votes on a bool.  HIP's __ballot / __any take an int: the bool is first materialised as 0 / 1 in a VGPR
// (v_cndmask) and compared with zero again (v_cmp_ne) -- two VALU instructions per vote that these forms do not need
// (the condition already is a lane mask in SGPRs).  The walks vote several times per trip.
__device__ __forceinline__ unsigned long long pt_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool pt_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// Population count of a lane mask as a 32-bit scalar.  Through __builtin_popcountll the compiler compares two such counts
// as 64-bit values, for which the scalar unit has no ordered compare: it moves one of them into a vector register and
// compares there (v_mov + v_cmp_lt_u64 + s_and with exec) on every vote.
__device__ __forceinline__ int pt_popc(unsigned long long mask)
{
    int r;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(r) : "s"(mask) : "scc");
    return r;
}
// v_rcp (1 ulp) + one Newton step in FMA + v_div_fixup (zeros, infinities, NaN).  Measured exhaustively:
// equals 1.0f/x for every x with 2^-126 <= |x| <= 2^126 and for +-0, +-inf, NaN; it is wrong only for
// (some) denormal x and for |x| > 2^126 (denormal quotient), which the callers below route to the
// compiler's full sequence.
__device__ __forceinline__ float pt_rcp_core(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
}
__device__ __forceinline__ float pt_rcp(float x)                                // == 1.0f / x
{
    float r = pt_rcp_core(x);
    const bool odd = __builtin_amdgcn_classf(x, 0x90) | (__builtin_fabsf(x) > 0x1p126f);   // 0x90: +-denormal
    if (__builtin_expect(pt_any(odd), 0)) r = 1.0f / x;      // wave-uniform branch, practically never taken
    return r;
}
// Same, for a caller that discards the result whenever |x| <= 1e-8 (Moller-Trumbore's det).
__device__ __forceinline__ float pt_rcp_not_tiny(float x)
{
    float r = pt_rcp_core(x);
    if (__builtin_expect(pt_any(__builtin_fabsf(x) > 0x1p126f), 0)) r = 1.0f / x;
    return r;
}
// v_rsq (1 ulp) -> s = x*y, h = y/2, one residual step s + (x - s*s)*h in FMA.  Measured exhaustively:
// equals sqrtf(x) for every finite x >= 2^-102; below that (residual underflows), for zero, negatives,
// infinity and NaN the wave takes the compiler's sequence.
__device__ __forceinline__ float pt_sqrt(float x)                               // == sqrtf(x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(r, h, s);
    const bool odd = !((x >= 0x1p-100f) & (x <= 3.402823466e+38f));
    if (__builtin_expect(pt_any(odd), 0)) s = sqrtf(x);
    return s;
}
// x * (1/pi) in double, one FMA residual correction, round to float: equals the double division's
// float for all 2^32 inputs except -0 and +-inf, which the class test passes through.
__device__ __forceinline__ float pt_div_pi(float x)                             // == (float)((double)x / M_PI)
{
    const double inv_pi = 0.31830988618379067154;
    double xd = (double)x;
    double q = xd * inv_pi;
    double r = __builtin_fma(-q, M_PI, xd);
    q = __builtin_fma(r, inv_pi, q);
    return __builtin_amdgcn_classf(x, 0x264) ? x : (float)q;                    // 0x264: +-0, +-inf
}
#endif

struct v3 { float x, y, z; };

PT_HD v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD v3 operator*(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD v3 operator*(float s, v3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
PT_HD v3 operator/(v3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }   // helper_math.h:1015
PT_HD float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // :1266
PT_HD v3 cross(v3 a, v3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_HD float length(v3 v)                                                        // :1309
{
#if defined(__HIP_DEVICE_COMPILE__)
    return pt_sqrt(dot(v, v));
#else
    return sqrtf(dot(v, v));
#endif
}
PT_HD v3 normalize(v3 v)                                                        // :1327 + :80
{
#if defined(__HIP_DEVICE_COMPILE__)
    float inv = pt_rcp(pt_sqrt(dot(v, v)));
#else
    float inv = 1.0f / sqrtf(dot(v, v));
#endif
    return v * inv;
}

// sin/cos for 0 <= x <= 2*pi(+): Cody-Waite reduction by pi/2 with fmaf, Cephes
// single-precision minimax polynomials on [-pi/4, pi/4]; stands in for CUDA's cosf/sinf
// (helper_math.h:1524-1525) and tan (camera.h:25,27).  Bit-identical on x86-64 and gfx950.
PT_HD void pt_sincosf(float x, float &s, float &c)
{
    float kf = (float)(int)(x * 0.636619772f + 0.5f);
    int k = (int)kf;
    float r = fmaf(-kf, 1.57079637e+0f, x);
    r = fmaf(-kf, -4.37113883e-8f, r);
    float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float sr = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    int q = k & 3;
    float s0 = (q & 1) ? cr : sr;
    float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q == 1) || (q == 2)) ? -c0 : c0;
}
PT_HD float pt_tanf(float x)
{
    float s, c;
    pt_sincosf(x, s, c);
    return s / c;
}

// cuRAND XORWOW as used by the reference (curand_init(seed,0,0) DevicePathTracer.h:54,
// curand_uniform at every call site listed in SURVEY.md 8c-1).  State is re-derived from
// the seed in-kernel instead of being read from a 48 B/pixel table.
struct xorwow { uint32_t d, v0, v1, v2, v3, v4; };

PT_HD void xorwow_init(xorwow &st, uint64_t seed)
{
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    st.d = 6615241u + t1 + t0;
    st.v0 = 123456789u + t0;
    st.v1 = 362436069u ^ t0;
    st.v2 = 521288629u + t1;
    st.v3 = 88675123u ^ t1;
    st.v4 = 5783321u + t0;
}
PT_HD uint32_t xorwow_next(xorwow &st)
{
    uint32_t t = st.v0 ^ (st.v0 >> 2);
    st.v0 = st.v1; st.v1 = st.v2; st.v2 = st.v3; st.v3 = st.v4;
    st.v4 = (st.v4 ^ (st.v4 << 4)) ^ (t ^ (t << 1));
    st.d += 362437u;
    return st.v4 + st.d;
}
// curand_uniform: x * 2^-32 + 2^-33 in (0, 1]
PT_HD float xorwow_uniform(xorwow &st)
{
    return (float)xorwow_next(st) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

// float -> int with CUDA cvt.rzi.s32.f32 semantics (truncate, saturate, NaN -> 0), which
// is also what gfx950's v_cvt_i32_f32 does; spelled out so the host agrees.
PT_HD int f2i_rz_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

} // namespace pt
