// tools/ta_gather_bench.hip -- what a divergent per-lane gather costs on the vector-memory path of one gfx950 CU.
// Not part of the product: it answers the question the C4 kernel's layout hangs on:
// is a lane's request priced per 16 bytes, per instruction, or per cache line touched?
//
//   hipcc -O3 --offload-arch=gfx950 -o build/ta_gather_bench tools/ta_gather_bench.hip && build/ta_gather_bench
//
// Every lane of every wave reads `record` bytes at a pseudo-random record of a working set (independent addresses:
// throughput, not a dependent chain), in one of several instruction mixes; the table is lane-requests per clock per CU.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode: which loads a lane issues per record
//  0: 1 x dword          1: 1 x dwordx2        2: 1 x dwordx4         3: 2 x dwordx4 (32 B)   4: 3 x dwordx4 (48 B)
//  5: 4 x dwordx4 (64 B) 6: x4 + x4 + dword (36 B, 4-aligned)         7: x4 + x4 + x2 (40 B, 8-aligned)
//  8: 1 x dwordx4 at +4 (unaligned)            9: 3 x dwordx4 over a 48 B record at 48 B stride (straddles lines)
// 10: 5 x dwordx4 (80 B)  11: x4 x4 x4(unaligned) over 36 B stride = two 36-byte records (72 B) in 5 loads? -> 9 x x4 over 144 B
template <int kMode>
__global__ __launch_bounds__(256) void gather(const char *base, uint32_t n_records, uint32_t stride, int iters, float *sink, int active_lanes)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    if ((int)(threadIdx.x & 63) >= active_lanes) { sink[gid] = 0.0f; return; }
    uint32_t h = mix(gid * 2654435761u + 12345u);
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t r = (uint32_t)(((uint64_t)mix(h) * n_records) >> 32);
        const char *p = base + (size_t)r * stride;
        if (kMode == 0) acc += *(const float *)p;
        else if (kMode == 1) { f2 a = *(const f2 *)p; acc += a.x + a.y; }
        else if (kMode == 2) { f4 a = *(const f4 *)p; acc += a.x + a.w; }
        else if (kMode == 3) { f4 a = ((const f4 *)p)[0], b = ((const f4 *)p)[1]; acc += a.x + b.w; }
        else if (kMode == 4 || kMode == 9) { f4 a = ((const f4 *)p)[0], b = ((const f4 *)p)[1], c = ((const f4 *)p)[2]; acc += a.x + b.w + c.y; }
        else if (kMode == 5) { f4 a = ((const f4 *)p)[0], b = ((const f4 *)p)[1], c = ((const f4 *)p)[2], d = ((const f4 *)p)[3]; acc += a.x + b.w + c.y + d.z; }
        else if (kMode == 6) { f4 a = ((const f4 *)p)[0], b = ((const f4 *)p)[1]; float c = ((const float *)p)[8]; acc += a.x + b.w + c; }
        else if (kMode == 7) { f4 a = ((const f4 *)p)[0], b = ((const f4 *)p)[1]; f2 c = ((const f2 *)p)[4]; acc += a.x + b.w + c.y; }
        else if (kMode == 8) { f4 a = *(const f4 *)(p + 4); acc += a.x + a.w; }
        else if (kMode == 10) { f4 a = ((const f4 *)p)[0], b = ((const f4 *)p)[1], c = ((const f4 *)p)[2], d = ((const f4 *)p)[3], e = ((const f4 *)p)[4]; acc += a.x + b.w + c.y + d.z + e.x; }
        else if (kMode == 11) { float s = 0; for (int k = 0; k < 9; k++) { f4 a = ((const f4 *)p)[k]; s += a.x + a.w; } acc += s; }
    }
    sink[gid] = acc;
}


// ---- part 2: R rays per wave want one 64-byte (or 48-byte) record each ------------------------------------------------
// kCoop 0: each of the R ray lanes issues kPieces x dwordx4 for its own record (what the SAH kernel does today);
//       1: the record's 16-byte pieces are fetched by ADJACENT lanes (lane l: piece l % kPieces of ray l / kPieces), one
//          instruction per 64 / kPieces rays;   2: piece-major (lane l: piece l / 16 of ray l % 16; kPieces = 4 only);
//       3: as 1, by global_load_lds_dwordx4 (the data lands in LDS, lane-linear)
template <int kCoop, int kPieces>
__global__ __launch_bounds__(256) void rays_kernel(const char *base, uint32_t n_records, int iters, float *sink, int rays)
{
    __shared__ f4 lds[4 * 64];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = gid >> 6;
    float acc = 0.0f;
    const int per_instr = kCoop == 2 ? 16 : 64 / kPieces;
    for (int it = 0; it < iters; it++) {
        if (kCoop == 0) {
            if (lane < rays) {
                const uint32_t r = (uint32_t)(((uint64_t)mix((wave * 64u + (uint32_t)lane) * 2654435761u + (uint32_t)it * 97u) * n_records) >> 32);
                const f4 *p = (const f4 *)(base + (size_t)r * 64);
                for (int k = 0; k < kPieces; k++) { f4 a = p[k]; acc += a.x + a.w; }
            }
        } else {
            for (int r0 = 0; r0 < rays; r0 += per_instr) {
                const int ray = r0 + (kCoop == 2 ? (lane & 15) : lane / kPieces), piece = kCoop == 2 ? (lane >> 4) : lane % kPieces;
                if (ray < rays && lane < per_instr * kPieces) {
                    const uint32_t r = (uint32_t)(((uint64_t)mix((wave * 64u + (uint32_t)ray) * 2654435761u + (uint32_t)it * 97u) * n_records) >> 32);
                    const f4 *p = (const f4 *)(base + (size_t)r * 64) + piece;
                    if (kCoop == 3) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                         (__attribute__((address_space(3))) void *)(lds + (threadIdx.x & ~63)), 16, 0, 0);
                    } else { f4 a = *p; acc += a.x + a.w; }
                }
            }
            if (kCoop == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += lds[threadIdx.x].x; }
        }
    }
    sink[gid] = acc;
}
typedef void (*RayKern)(const char *, uint32_t, int, float *, int);
typedef void (*Kern)(const char *, uint32_t, uint32_t, int, float *, int);

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, mhz);
    const size_t max_bytes = (size_t)2 << 30;
    char *buf; float *sink;
    CHECK(hipMalloc((void **)&buf, max_bytes + 4096));
    CHECK(hipMemset(buf, 0, max_bytes + 4096));
    const int waves_per_cu = 16, blocks = cus * waves_per_cu / 4;
    CHECK(hipMalloc((void **)&sink, (size_t)blocks * 256 * 4));
    struct Mode { Kern k; const char *name; uint32_t stride; int loads; int bytes; };
    const Mode modes[] = {
        {gather<0>, "1 x dword", 64, 1, 4}, {gather<1>, "1 x dwordx2", 64, 1, 8}, {gather<2>, "1 x dwordx4", 64, 1, 16},
        {gather<3>, "2 x dwordx4 / 64 B rec", 64, 2, 32}, {gather<4>, "3 x dwordx4 / 64 B rec", 64, 3, 48}, {gather<5>, "4 x dwordx4 / 64 B rec", 64, 4, 64},
        {gather<6>, "x4 x4 x1 / 36 B stride", 36, 3, 36}, {gather<7>, "x4 x4 x2 / 40 B stride", 40, 3, 40}, {gather<8>, "1 x dwordx4 at +4", 64, 1, 16},
        {gather<9>, "3 x dwordx4 / 48 B stride", 48, 3, 48}, {gather<10>, "5 x dwordx4 / 80 B stride", 80, 5, 80}, {gather<11>, "9 x dwordx4 / 144 B stride", 144, 9, 144},
    };
    const size_t sets[] = {(size_t)16 << 10, (size_t)1 << 20, (size_t)64 << 20, (size_t)2 << 30};
    const char *set_names[] = {"16 KB (L1)", "1 MB (L2)", "64 MB (MALL)", "2 GB (HBM)"};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int lanes : {64, 24}) {
        printf("\n== %d of 64 lanes active, %d waves per CU ==\n", lanes, waves_per_cu);
        printf("%-28s", "mode");
        for (const char *s : set_names) printf(" | %-26s", s);
        printf("\n%-28s", "");
        for (int s = 0; s < 4; s++) printf(" | %8s %8s %8s", "lane/clk", "B/clk", "rec/clk");
        printf("   (per CU)\n");
        for (const Mode &m : modes) {
            printf("%-28s", m.name);
            for (int s = 0; s < 4; s++) {
                const uint32_t n_rec = (uint32_t)(sets[s] / m.stride);
                const int iters = s < 2 ? 2000 : 400;
                hipLaunchKernelGGL(m.k, dim3(blocks), dim3(256), 0, 0, buf, n_rec, m.stride, iters / 10, sink, lanes);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(m.k, dim3(blocks), dim3(256), 0, 0, buf, n_rec, m.stride, iters, sink, lanes);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double clocks = ms * 1e-3 * mhz * 1e6;
                const double recs = (double)blocks * 4 * lanes * iters / cus;          // records per CU
                printf(" | %8.3f %8.1f %8.3f", recs * m.loads / clocks, recs * m.bytes / clocks, recs / clocks);
            }
            printf("\n");
        }
    }

    {
        struct RM { RayKern k; const char *name; };
        const RM rms[] = {{rays_kernel<0, 4>, "own lane, 4 x dwordx4"}, {rays_kernel<1, 4>, "4 adjacent lanes x dwordx4"}, {rays_kernel<2, 4>, "piece-major lanes x dwordx4"},
                          {rays_kernel<3, 4>, "4 adjacent lanes, load_lds"}, {rays_kernel<0, 3>, "own lane, 3 x dwordx4"}, {rays_kernel<1, 3>, "3 adjacent lanes x dwordx4"},
                          {rays_kernel<0, 1>, "own lane, 1 x dwordx4"}};
        printf("\n== R rays per wave fetch one record each: records per clock per CU (64-byte stride) ==\n");
        for (int s = 0; s < 4; s += 1) {
            printf("-- working set %s\n%-30s", set_names[s], "rays per wave ->");
            const int rs[] = {4, 8, 16, 24, 32, 48, 64};
            for (int r : rs) printf(" %7d", r);
            printf("\n");
            for (const RM &m : rms) {
                printf("%-30s", m.name);
                for (int r : rs) {
                    const uint32_t n_rec = (uint32_t)(sets[s] / 64);
                    const int iters = s < 2 ? 1000 : 200;
                    hipLaunchKernelGGL(m.k, dim3(blocks), dim3(256), 0, 0, buf, n_rec, iters / 10, sink, r);
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(m.k, dim3(blocks), dim3(256), 0, 0, buf, n_rec, iters, sink, r);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    const double clocks = ms * 1e-3 * mhz * 1e6;
                    printf(" %7.3f", (double)blocks * 4 * r * iters / cus / clocks);
                }
                printf("\n");
            }
        }
    }
    return 0;
}
