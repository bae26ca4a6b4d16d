// tools/valu_issue_bench.hip -- what one vector instruction costs a wave's instruction stream on gfx950, by encoding:
// VOP2 (v_mul_f32, v_fmac_f32), VOP3 with three distinct VGPR sources (v_fma_f32, v_max3_f32), conversions, compares + selects.
// Independent instructions (16 accumulators), W waves per SIMD; prints cycles per instruction per wave and instructions
// per cycle per SIMD.  Not product code.   hipcc -O3 --offload-arch=gfx950 -o build/valu_issue_bench tools/valu_issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int kOp>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b, unsigned long long *cycles)
{
    float r[16];
    for (int i = 0; i < 16; i++) r[i] = (float)threadIdx.x + i;
    float x = a + threadIdx.x, y = b + threadIdx.x;
    unsigned ub = (unsigned)threadIdx.x * 0x01010101u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define OP_MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(x));
#define OP_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(x), "v"(y));
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
#define OP_FMA3(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(x), "v"(y), "v"(r[(i + 1) & 15]));
#define OP_MAX3(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y));
#define OP_CVT(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r[i]) : "v"(ub));
#define OP_MIN(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(r[i]) : "v"(x));
#define OP_CMPSEL(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(x) : "vcc");
#define OP_SUBMUL(i) asm volatile("v_sub_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %2" : "+v"(r[i]) : "v"(x), "v"(y));
        if (kOp == 0) { REP16(OP_MUL) REP16(OP_MUL) REP16(OP_MUL) REP16(OP_MUL) }
        if (kOp == 1) { REP16(OP_FMAC) REP16(OP_FMAC) REP16(OP_FMAC) REP16(OP_FMAC) }
        if (kOp == 2) { REP16(OP_FMA) REP16(OP_FMA) REP16(OP_FMA) REP16(OP_FMA) }
        if (kOp == 3) { REP16(OP_FMA3) REP16(OP_FMA3) REP16(OP_FMA3) REP16(OP_FMA3) }
        if (kOp == 4) { REP16(OP_MAX3) REP16(OP_MAX3) REP16(OP_MAX3) REP16(OP_MAX3) }
        if (kOp == 5) { REP16(OP_CVT) REP16(OP_CVT) REP16(OP_CVT) REP16(OP_CVT) }
        if (kOp == 6) { REP16(OP_MIN) REP16(OP_MIN) REP16(OP_MIN) REP16(OP_MIN) }
        if (kOp == 7) { REP16(OP_CMPSEL) REP16(OP_CMPSEL) }
        if (kOp == 8) { REP16(OP_SUBMUL) REP16(OP_SUBMUL) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += r[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
typedef void (*K)(float *, int, float, float, unsigned long long *);
int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; unsigned long long *cyc;
    const int max_blocks = cus * 4 * 8;
    CHECK(hipMalloc((void **)&out, (size_t)max_blocks * 64 * 4)); CHECK(hipMalloc((void **)&cyc, (size_t)max_blocks * 8));
    struct M { K k; const char *name; int per_iter; };
    const M ms[] = {{k<0>, "v_mul_f32 (VOP2)", 64}, {k<1>, "v_fmac_f32 (VOP2)", 64}, {k<2>, "v_fma_f32 d = a*b+d (VOP3, 3 VGPR)", 64}, {k<3>, "v_fma_f32 d = a*b+c (VOP3, 3 VGPR + dst)", 64},
                    {k<4>, "v_max3_f32 (VOP3, 3 VGPR)", 64}, {k<5>, "v_cvt_f32_ubyte1 (VOP1)", 64}, {k<6>, "v_min_f32 (VOP2)", 64}, {k<7>, "v_cmp + v_cndmask pair", 64}, {k<8>, "v_sub + v_mul pair", 64}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-44s", "instruction");
    for (int w : {1, 2, 4, 5, 8}) printf(" | %d wave%s/SIMD: cyc/instr/wave  instr/clk/SIMD", w, w > 1 ? "s" : " ");
    printf("\n");
    for (const M &m : ms) {
        printf("%-44s", m.name);
        for (int w : {1, 2, 4, 5, 8}) {
            const int blocks = cus * 4 * w, iters = 2000;
            hipLaunchKernelGGL(m.k, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0f, 2.0f, cyc);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(m.k, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f, 2.0f, cyc);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms_; CHECK(hipEventElapsedTime(&ms_, e0, e1));
            const double clk = ms_ * 1e-3 * (prop.clockRate * 1e3);
            const double instr_per_wave = (double)iters * m.per_iter;
            printf(" | %22.2f %15.3f", clk / instr_per_wave, instr_per_wave * w / clk);
        }
        printf("\n");
    }
    return 0;
}
