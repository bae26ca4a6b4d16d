// pt_device.hip -- gfx950 kernels of the hot path and the per-GPU engine behind the C ABI.
//
// Replaces, for one GPU, DevicePathTracer (DevicePathTracer.h:167-392) and the kernels it
// launches: render_init (:46-55, folded into the render kernel), render (:73-120),
// create_world / create_lights / create_camera (:134-158, done on the host at upload).
//
// Kernel shape: one lane = one pixel, one wave64 = one 8x8 pixel tile, 4 waves
// per workgroup.  Each lane runs a path STATE MACHINE: every trip of the wave loop a live
// lane traces exactly one ray segment of whatever (sample, bounce) it is on, and rolls
// straight into its next sample when a path ends, so short paths do not idle behind the
// longest path of the wave.  The per-lane traversal stack lives in LDS.  No MFMA: this is
// a divergent gather workload.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>          // types and prototypes only: librccl is dlopen()ed by the native gather, never linked
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "pt_kernels.h"
#include "pt_walk.h"
#include "pt_shade.h"

namespace pt {

// ====================================================================== device code ==
namespace dev {

// Waves per SIMD the register allocator must make room for: 8 (64 VGPRs) costs a few spills in the
// per-segment shading code, none in the traversal loops, and buys latency hiding (C2 80.6 -> 77.5 ms).
#ifndef PT_RENDER_WAVES
#define PT_RENDER_WAVES 8
#endif

// kWalk: 0 = per-lane two-phase walk, 1 = wave-uniform walk with a 64-bit path mask (<= 64 inner
// nodes), 2 = wave-uniform walk with the per-lane LDS stack (medium trees), 3 = SAH tree walk.
// render, DevicePathTracer.h:73-120, with render_init (:46-55) folded in and
// camera::ray_color (camera.h:49-83) flattened into a per-lane state machine.
template <bool kLds, bool kPacked, int kWalk, bool kLegacy = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_RENDER_WAVES, PT_RENDER_WAVES))) void render_kernel(RenderParams P)
{
    extern __shared__ float4 lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    int x, y, pixel_index;
    const bool in_frame = lane_pixel(P, tid, x, y, pixel_index);
    const LdsLayout LL = lds_carve<kLds>(P, lds_raw, kBlock);
    const SceneView<kLds> S = stage_scene<kLds>(P, lds_raw, tid, kBlock);
    const ShadeView V = stage_shade(P, LL.tail, tid, kBlock);
    typedef typename QueueEntry<kPacked>::type qentry;
    uint32_t *stack = LL.stack + tid;
    qentry *queue = reinterpret_cast<qentry *>(LL.queue) + tid;

    PathState L;
    path_init(L, P, pixel_index);
    bool alive = in_frame && P.depth > 0;     // depth 0: ray_color returns (0,0,0) untraced ...
    if (in_frame && P.depth == 0) L.n_paths = (uint32_t)P.spp;      // ... for camera rays that are still generated (:84-87)
#ifdef PT_STAMPS
    // diagnostic build only (tools/stamps.sh): shader-clock cycles spent inside the traversal vs the
    // whole wave loop, summed per wave into counters[5..7]; never compiled into the shipped library
    unsigned long long stamp_trav = 0, stamp_trips = 0;
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime();
#endif

    unsigned long long wave_box = 0, wave_tri = 0;   // wave totals of the wave-uniform walks (kWalk 1, 2)
    while (alive) {
        if (L.need_new && !start_path(L, P, x, y)) { alive = false; continue; }
        L.n_seg++;
#ifdef PT_STAMPS
        const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
        Hit h;
        if (kWalk == 1) h = traverse_uniform(P, L.ro, L.rd, 0.001f, 3.402823466e+38f, wave_box, wave_tri);
        else if (kWalk == 2) h = traverse_uniform_stack(P, L.ro, L.rd, 0.001f, 3.402823466e+38f, stack, kBlock, wave_box, wave_tri);
        else if (kWalk == 3) h = traverse_sah(P, L.ro, L.rd, 0.001f, 3.402823466e+38f, stack, kBlock, L.n_box, L.n_tri);
        else h = traverse<kLds, kPacked>(S, L.ro, L.rd, 0.001f, 3.402823466e+38f, stack, queue, kBlock, P.queue_entries, L.n_box, L.n_tri);
#ifdef PT_STAMPS
        stamp_trav += __builtin_amdgcn_s_memtime() - st0;
        stamp_trips++;
#endif
        if (kLegacy) spheres_hit(P, L.ro, L.rd, 0.001f, h);
        if (kWalk == 0 || kWalk == 3) fold_counters(L, P.fold_bits);      // per-lane walks count in 32 bits per lane
        shade<kLegacy>(L, P, V, h);
    }

    if (in_frame) write_pixel(P, L.col, pixel_index);
#ifdef PT_STAMPS
    {
        const unsigned long long total = __builtin_amdgcn_s_memtime() - stamp_begin;
        // the wave's trip count / traversal cycles = those of its longest-living lane
        unsigned long long tr = stamp_trips, tv = stamp_trav;
        for (int off = 32; off > 0; off >>= 1) {
            unsigned long long o1 = __shfl_xor(tr, off, 64), o2 = __shfl_xor(tv, off, 64);
            if (o1 > tr) { tr = o1; tv = o2; }
        }
        if (lane == 0) {
            atomicMax(&P.counters[5], total);            // longest wave (cycles)
            atomicAdd(&P.counters[6], tv);               // sum over waves: traversal cycles of the longest lane
            atomicAdd(&P.counters[7], total);            // sum over waves: wave lifetime
        }
    }
#endif
    flush_counters(P, L, lane, wave_box, wave_tri);
    record_tile_cost(P, tid, wave_max(L.n_seg));      // trips of this wave's loop = segments of its busiest lane
}

// Mask walk with dealt sparse leaves (kWalk 4, traverse_uniform_dealt): same integrator, but the rounds that work the
// item list off use all 64 lanes, so the wave loop is convergent (every lane stays until the wave's last pixel is done)
// and lanes without a ray of their own pass live = false.
#ifndef PT_DEAL_WAVES
#define PT_DEAL_WAVES 7
#endif
#ifdef PT_TIMELINE
// diagnostic build only (tools/timeline.sh): per wave {start, end} of the 100 MHz real-time counter, hardware id and tile,
// for an occupancy-over-time picture of one launch; never compiled into the shipped library
__device__ uint32_t g_timeline[4 * 131072];
#endif

// The frame both dealt kernels share: persistent waves that draw their tiles from the launch's counter, convergent
// trips (every lane stays until the wave's last pixel is done, so the item rounds have all 64 lanes), issue priority by
// remaining samples, split tiles.  `Walk` supplies the traversal and the shading view of its kernel (MaskDealtWalk: the
// box's, kWalk 4; StackDealtWalk: the duck's, kWalk 5).
template <class Walk>
__device__ __forceinline__ void render_dealt_body(const RenderParams &P, const Walk &Wk, const int lane)
{
    // Split tiles (see render_sah_body): here a trip's cost is mostly the wave's (box walk, shading), so a part of 16
    // pixels still costs ~0.65 of the tile per trip -- worth it only when the tile's chain bounds the launch, i.e. for
    // a rank's share of a multi-GPU frame, where there are fewer tiles than wave slots (split_count_kernel decides).
    const uint32_t parts = (uint32_t)P.split_parts;
    uint32_t n_split = 0, n_slots_all = 0;
    {
        KernargParams &K = *fresh_kernarg();
        if (K.n_split) n_split = K.n_split[0];
        n_slots_all = K.n_slots + n_split * (parts - 1u);
    }
    // Persistent waves: the launch has as many workgroups as the device holds at once, and every WAVE takes its next tile
    // from a counter when its own tile is done.  With one workgroup per four tiles a finished wave's slot stayed empty until
    // its three siblings were done too (5 % of the slots of a C2 frame), and the shading tables were staged 8 100 times.
    while (true) {
        KernargParams &K = *fresh_kernarg();
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(K.tile_next, 1u);
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        if (slot >= n_slots_all) break;                    // every wave gets here: the counter only grows
#ifdef PT_TIMELINE
        const uint32_t tl_start = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
        const SlotWork sw = slot_work(K, slot, n_split, parts, lane);
        const int tile = sw.tile;
        // a part's rays sit in its first 64 / parts lanes; walks that can (MaskDealtWalk with the boxes staged) spread them
        const int part_lanes = sw.is_part ? (int)(64u / parts) : 0;
        int x, y, pixel_index;
        const bool in_frame = tile_pixel(K, tile, sw.tile_lane, x, y, pixel_index) && sw.lane_on;
        PathState L;
        path_init(L, P, pixel_index);
        bool alive = in_frame && P.depth > 0;
        // wave totals in scalar registers: box / triangle tests (as in the mask walk), segments (lanes that trace in a trip)
        // and the trip count itself, which is this tile's cost for the next launch's order
        unsigned long long wave_box = 0, wave_tri = 0;
        uint32_t wave_seg = 0, trips = 0;
        // Issue priority by remaining work (P.lrpt): a wave that still has most of its samples ahead outranks one that is
        // nearly done -- persistent waves all have the same age, and without this some wave of every SIMD starves.
        int prio_now = 0;
        while (true) {
            const unsigned long long alive_mask = pt_ballot(alive);
            if (alive_mask == 0ull) break;
            if (P.lrpt) {
                const int s_now = __builtin_amdgcn_readlane(L.s, (int)__builtin_ctzll(alive_mask));     // a tile's pixels progress within a few %
                const int left4 = (P.spp - s_now) * 4;
                const int want = left4 > 3 * P.spp ? 3 : left4 > 2 * P.spp ? 2 : left4 > P.spp ? 1 : 0;
                if (want != prio_now) {
                    prio_now = want;
                    if (want == 3) __builtin_amdgcn_s_setprio(3);
                    else if (want == 2) __builtin_amdgcn_s_setprio(2);
                    else if (want == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
            }
            bool trace = alive;
#ifdef PT_STAMPS
            // diagnostic build only (tools/stamps.sh): shader-clock cycles of the three parts of a trip, summed per wave
            // into counters[5..7] = {path start, walk, shading}; never compiled into the shipped library
            const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
            if (alive && L.need_new && !start_path(L, P, x, y)) { alive = false; trace = false; }
#ifdef PT_STAMPS
            const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
            const unsigned long long trace_mask = pt_ballot(trace);
            if (trace_mask != 0ull) {                          // (the last trip of a wave only discovers that every pixel is done)
                wave_seg += (uint32_t)__popcll(trace_mask);
                trips++;
            }
            const Hit h = Wk.hit(P, lane, trace, L.ro, L.rd, wave_box, wave_tri, part_lanes);
#ifdef PT_STAMPS
            const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
            if (trace) Wk.shade_hit(L, P, h);
#ifdef PT_STAMPS
            {
                const unsigned long long st3 = __builtin_amdgcn_s_memtime();
                if (lane == 0) { atomicAdd(&P.counters[5], st1 - st0); atomicAdd(&P.counters[6], st2 - st1); atomicAdd(&P.counters[7], st3 - st2); }
            }
#endif
        }
        KernargParams &E = *fresh_kernarg();
        if (in_frame) write_pixel(E, L.col, pixel_index);
        L.n_paths = in_frame ? (P.depth == 0 ? (uint32_t)P.spp : (uint32_t)L.s) : 0u;      // samples started (depth 0: rays generated, untraced)
        L.n_seg = 0;
        flush_counters(E, L, lane, wave_box, wave_tri, wave_seg);
        if (E.tile_cost && lane == 0) {
            if (sw.is_part) atomicMax(&E.tile_cost[tile], trips);      // a tile's trips = those of its busiest pixel, in whichever part
            else E.tile_cost[tile] = trips;
        }
#ifdef PT_TIMELINE
        if (lane == 0 && slot < 131072u) {
            g_timeline[4 * slot] = tl_start;
            g_timeline[4 * slot + 1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            g_timeline[4 * slot + 2] = (uint32_t)tile | (sw.is_part ? 0x80000000u : 0u);
            g_timeline[4 * slot + 3] = trips;
        }
#endif
    }
}

// kWalk 4: the mask walk with dealt sparse leaves (traverse_uniform_dealt), shading tables, triangle and onb tables in LDS.
template <bool kInPlace, bool kParts = true>
struct MaskDealtWalk {
    ShadeViewLds V;
    DealView D;
    __device__ __forceinline__ Hit hit(const RenderParams &P, int lane, bool trace, const v3 &ro, const v3 &rd,
                                       unsigned long long &wave_box, unsigned long long &wave_tri, int part_lanes) const
    {
        return traverse_uniform_dealt<kInPlace>(P, D, lane, trace, ro, rd, 0.001f, 3.402823466e+38f, wave_box, wave_tri,
                                                kParts && D.boxes ? part_lanes : 0);
    }
    __device__ __forceinline__ void shade_hit(PathState &L, const RenderParams &P, const Hit &h) const { shade<false, ShadeViewLds>(L, P, V, h); }
};

template <bool kInPlace, bool kParts = true>
__device__ __forceinline__ void render_kernel_dealt_impl(const RenderParams &P)
{
    extern __shared__ float4 lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const LdsLayout LL = lds_carve<false>(P, lds_raw, kBlock);
    MaskDealtWalk<kInPlace, kParts> Wk;
    Wk.V = stage_shade_lds(P, LL.tail, tid, kBlock);      // the host picks this kernel only when the tables fit
    Wk.D = deal_carve(P, Wk.V.end, tid, kBlock);
    render_dealt_body(P, Wk, lane);
}
template <bool kInPlace>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_DEAL_WAVES, PT_DEAL_WAVES))) void render_kernel_dealt(RenderParams P)
{
    render_kernel_dealt_impl<kInPlace>(P);
}
// The same kernel built for eight waves per SIMD (64 VGPRs, 21 of the 7-wave build's registers spilled): for launches that
// fill the device more than twice over the eighth wave buys more than the spills cost (C2 59.7 -> 58.9 ms, C5 878 -> 874 ms);
// a rank's share, whose time is one wave's chain, loses to them (an eighth of C2 14.6 -> 15.5 ms) and keeps the build above.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void render_kernel_dealt8(RenderParams P)
{
    render_kernel_dealt_impl<true, false>(P);       // (dense launches never split: no part code in this build)
}

// kWalk 5: the stack walk with dealt sparse leaves (traverse_uniform_stack_dealt) for trees of up to 4 096 inner nodes (the
// duck): per-lane LDS stack, triangle records from memory.
#ifndef PT_DEAL_STACK_WAVES
#define PT_DEAL_STACK_WAVES 5
#endif
struct StackDealtWalk {
    ShadeView V;
    DealView32 D;
    uint32_t *stack;
    __device__ __forceinline__ Hit hit(const RenderParams &P, int lane, bool trace, const v3 &ro, const v3 &rd,
                                       unsigned long long &wave_box, unsigned long long &wave_tri, int /*part_lanes: parts walk like whole tiles here*/) const
    {
        return traverse_uniform_stack_dealt(P, D, lane, trace, ro, rd, 0.001f, 3.402823466e+38f, stack, kBlock, wave_box, wave_tri);
    }
    __device__ __forceinline__ void shade_hit(PathState &L, const RenderParams &P, const Hit &h) const { shade<false, ShadeView>(L, P, V, h); }
};

__device__ __forceinline__ void render_kernel_dealt_stack_impl(const RenderParams &P)
{
    extern __shared__ float4 lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const LdsLayout LL = lds_carve<false>(P, lds_raw, kBlock);
    StackDealtWalk Wk;
    Wk.V = stage_shade(P, LL.tail, tid, kBlock);
    Wk.D = deal_carve32(P, Wk.V.end, tid);
    Wk.stack = LL.stack + tid;
    render_dealt_body(P, Wk, lane);
}
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_DEAL_STACK_WAVES, PT_DEAL_STACK_WAVES))) void render_kernel_dealt_stack(RenderParams P)
{
    render_kernel_dealt_stack_impl(P);
}
// Built for six waves per SIMD (80 VGPRs, 76 B of scratch per lane), run with a 384-entry item list so that six workgroups fit a
// CU's LDS: for launches of more than twice the device's wave slots (C3 with the reference's tree 3 862 -> 3 716 ms; the five-
// wave build with the same list: 3 916 ms).  Round 2 found no gain in a sixth wave: then the waves of a SIMD still stalled
// together on the dense leaf's scalar-cache misses.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void render_kernel_dealt_stack6(RenderParams P)
{
    render_kernel_dealt_stack_impl(P);
}

// Cooperative kernel for reference trees with big leaves (cornell_duck: one leaf of 1603 triangles, 2 120
// triangle tests per ray segment).  The four waves of a workgroup render the SAME 8x8 tile: each holds
// the full path state of its 64 pixels and does the box walk and the shading redundantly (bit-identical,
// so they never diverge from one another), but tests only its quarter of every leaf's triangles; the
// four partial closest hits are merged through LDS once per ray segment, min over (t, leaf-order slot) =
// what the in-order `t < ray_t.max` update of the reference computes.  A tile's samples are one
// sequential chain (a pixel's RNG stream), so this is the only way to put more than one wave on the
// dearest tiles: the chain gets ~3.5x shorter for ~10 % more work, which is what a small share of the
// frame (one rank of a multi-GPU run) and the tail of a whole frame are bound by.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_RENDER_WAVES, PT_RENDER_WAVES))) void render_kernel_coop(RenderParams P)
{
    extern __shared__ float4 lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int part = tid / kWave;                                     // which quarter of each leaf this wave tests
    const int tile = P.tile_order ? (int)P.tile_order[blockIdx.x] : (int)blockIdx.x;
    int x, y, pixel_index;
    const bool in_frame = tile_pixel(P, tile, lane, x, y, pixel_index);
    const LdsLayout LL = lds_carve<false>(P, lds_raw, kBlock);
    const ShadeView V = stage_shade(P, LL.tail, tid, kBlock);
    uint32_t *stack = LL.stack + tid;
    // two merge buffers of {float4 partial hit, visit rank} per thread, used alternately: one barrier per trip
    float4 *merge = reinterpret_cast<float4 *>(LL.queue);
    uint32_t *merge_rank = reinterpret_cast<uint32_t *>(merge + 2 * kBlock);

    PathState L;
    path_init(L, P, pixel_index);
    bool alive = in_frame && P.depth > 0;
    if (in_frame && P.depth == 0) L.n_paths = (uint32_t)P.spp;
    unsigned long long n_box = 0, n_tri = 0;                          // wave totals; box tests are counted by wave 0 only
    // The four waves hold bit-identical state, so they run the same number of trips and meet at one barrier
    // per trip; trip n+1 writes the other buffer, and by the time buffer n is written again (trip n+2) every
    // wave has passed the barrier of trip n+1, i.e. has finished reading it.
    int buf = 0;
    while (pt_any(alive)) {
        bool trace = alive;
        if (alive && L.need_new && !start_path(L, P, x, y)) { alive = false; trace = false; }
        Hit h; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.slot = -1;
        // equal t (a ray through a shared edge) goes to the triangle the reference visits first; that order is
        // not the slot order (a leaf child is tested before its inner sibling's subtree), so the partial
        // hits carry the visit rank stored with each triangle
        uint32_t rank = 0xffffffffu;
        if (trace) {
            L.n_seg++;
            h = traverse_uniform_stack<kWavesPerBlock>(P, L.ro, L.rd, 0.001f, 3.402823466e+38f, stack, kBlock, n_box, n_tri, part, &rank);
        }
        merge[buf * kBlock + tid] = make_float4(h.t, h.u, h.v, __int_as_float(h.slot));
        merge_rank[buf * kBlock + tid] = rank;
        __syncthreads();
        if (trace) {
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; w++) {
                const float4 m = merge[buf * kBlock + w * kWave + lane];
                const uint32_t r = merge_rank[buf * kBlock + w * kWave + lane];
                const int slot = __float_as_int(m.w);
                if (slot >= 0 && (h.slot < 0 || m.x < h.t || (m.x == h.t && r < rank))) { h.t = m.x; h.u = m.y; h.v = m.z; h.slot = slot; rank = r; }
            }
            shade(L, P, V, h);
        }
        buf ^= 1;
    }
    const uint32_t cost = wave_max(L.n_seg);
    if (part == 0) {
        if (in_frame) write_pixel(P, L.col, pixel_index);
    } else {
        L.n_paths = L.n_seg = L.n_light = 0;                          // counted once, by wave 0
    }
    flush_counters(P, L, lane, part == 0 ? n_box : 0ull, n_tri);
    if (part == 0 && P.tile_cost && lane == 0) P.tile_cost[tile] = cost;
}

// Render kernel for the SAH tree: same integrator, but the wave does not finish one ray segment
// per trip.  Path lengths through a deep tree differ by orders of magnitude between the lanes of
// a wave (a ray skimming the terrain visits hundreds of nodes, its neighbour ten), so the wave
// alternates between (1) giving every idle lane its next ray, (2) node visits for all lanes that
// are walking -- until at least half of those that were walking have finished -- and (3) shading
// the finished ones, while the stragglers keep their traversal state (pair, LDS stack, closest
// hit) across trips.  Per ray the sequence of tests is exactly traverse_sah's.
#ifdef PT_OCC
// diagnostic build only (tools/occupancy_census.sh): where the SAH kernel's lane slots go.  Per micro-step trip: [0] trips,
// [1] lanes that executed the step, [2] lanes that wanted the OTHER kind of step (the vote's losers), [3] lanes whose ray is
// finished and waits for the shading pass, [4] lanes without a pixel to work on (pixel done, off the frame, idle part lanes);
// per shading pass: [5] passes, [6] lanes shading or starting a path; [7] outer trips.  Never compiled into the shipped library.
__device__ unsigned long long g_occ[8];
#endif

// kForm: 0 = NodePair records, 1 = the 4-wide form (128-byte records), 2 = the 4-wide form in 64-byte records (QuadQ).
// kTop / kSpill (pt_walk.h): the tree's top levels in LDS and the short LDS stack with its overflow area, each compiled in
// (1), compiled out (0) or decided per launch (2).  The kernels the library picks by itself are specialised -- a cache-resident
// tree gets the top levels and whole stacks, a tree beyond the caches the short stack and no top levels -- because a lone
// wave's chain of dependent micro-steps is what bounds a rank's share of a frame, and generality costs instructions in it
// (an eighth of the duck's frame: 47.4 ms with everything decided at run time, 43.9 ms with round 2's kernel).
template <int kForm, int kTop = 2, int kSpill = 2>
__device__ __forceinline__ void render_sah_body(const RenderParams &P)
{
    extern __shared__ float4 lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const LdsLayout LL = lds_carve<false>(P, lds_raw, kBlock);
    const ShadeView V = stage_shade(P, LL.tail, tid, kBlock);
    // the top of every lane's stack in LDS, what a deep ray pushes beyond it in this wave's part of the launch's spill area
    const SahStack stack = {LL.stack + tid, kBlock, P.stack_cap,
                            P.stack_spill ? P.stack_spill + ((size_t)(blockIdx.x * kWavesPerBlock + (tid >> 6)) * (size_t)P.spill_blocks * 64 + (size_t)lane) * kSpillBlock
                                          : nullptr,
#ifdef PT_STALE
                            nullptr,                           // (that diagnostic build keeps its own figures in counters[5..6])
#else
                            P.counters + 5,
#endif
                            V.end, (uint32_t)(P.top_records * P.top_record_bytes)};
    if (kTop != 0 && P.top_records > 0) {                  // the tree's top levels, once per workgroup of the persistent launch
        const float4 *gn = reinterpret_cast<const float4 *>(P.nodes);
        const int n4 = P.top_records * (P.top_record_bytes / 16);
        for (int i = tid; i < n4; i += kBlock) V.end[i] = gn[i];
        __syncthreads();
    }
    // Persistent waves, as in render_kernel_dealt: every wave takes its next tile from the launch's counter.
    // Split tiles.  A pixel's samples are one chain, and on a scene like C4's terrain the few dearest tiles (rays skimming
    // the ground towards the horizon) cost five times the median: started first, they still end last, alone on the
    // device, and the frame is as long as their chain, not as the device's work.  Such a tile's pixels diverge (each
    // trip of the wave serves one kind of micro-step), so fewer pixels per wave shorten the chain: measured alone,
    // 39 ms as 8 x 8, 30-35 ms as 4 x 1, 22 ms for a single pixel (tools/lone_tile.sh).  The first n_split tiles of the
    // order (split_count_kernel) are therefore handed out as `parts` slots of 64 / parts pixels each; the other lanes of
    // those waves idle, which costs a per cent of the slots.  Which wave renders a pixel never changes its value.
    const uint32_t parts = (uint32_t)P.split_parts;
    uint32_t n_split = 0, n_slots_all = 0;
    {
        KernargParams &K = *fresh_kernarg();
        if (K.n_split) n_split = K.n_split[0];
        n_slots_all = K.n_slots + n_split * (parts - 1u);
    }
    while (true) {
        KernargParams &K = *fresh_kernarg();
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(K.tile_next, 1u);
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        if (slot >= n_slots_all) break;                    // every wave gets here: the counter only grows
#ifdef PT_TIMELINE
        const uint32_t tl_start = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
        const SlotWork sw = slot_work(K, slot, n_split, parts, lane);
        const int tile = sw.tile, tile_lane = sw.tile_lane;
        const bool lane_on = sw.lane_on, is_part = sw.is_part;
        int x, y;
        bool in_frame;
        PathState L;
        { int pixel_index; in_frame = tile_pixel(K, tile, tile_lane, x, y, pixel_index) && lane_on; path_init(L, P, pixel_index); }      // the pixel index is recomputed for the final store
        SahWalk W;
        sah_begin(W, mk3(1.0f, 1.0f, 1.0f), 0.0f);
        W.pair = kNoPair;                                      // nobody walks yet
#ifdef PT_STALE
        W.stale = W.visits = 0;
#endif
        bool alive = in_frame && P.depth > 0;
        uint32_t cost = 0;                                    // wave-uniform: micro-step trips + 8 per outer trip
        // Box / triangle tests are counted per lane (one masked add per micro-step; a wave-level sum of the vote's popcounts
        // was tried: the compiler keeps such a loop-carried uniform sum in vector registers and spends four instructions per
        // trip on its carry) and folded into wider per-lane words once per outer trip, long before 32 bits can wrap.
        uint32_t wave_seg = 0;
#ifdef PT_OCC
        unsigned long long occ[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        while (true) {
            cost += 8;
            fold_counters(L, P.fold_bits);
#ifdef PT_OCC
            occ[7]++; occ[5]++; occ[6] += (unsigned long long)pt_popc(pt_ballot(alive && !sah_walking(W)));
#endif
            bool begins = false;
            if (alive && !sah_walking(W)) {
                if (L.need_new && !start_path(L, P, x, y)) {
                    alive = false;
                } else {
                    begins = true;
                    sah_begin(W, L.rd, 3.402823466e+38f);
                    if (kForm == 2) sahq_begin(W);
                    else if (kForm == 1) sahw_begin(W);
                }
            }
            wave_seg += (uint32_t)__builtin_popcountll(pt_ballot(begins));
            const uint32_t total = (uint32_t)pt_popc(pt_ballot(W.n0 != 0) | pt_ballot(W.pair != kNoPair));
            if (total == 0) break;                            // wave-uniform: nobody has work left
            while (true) {
                // who wants what: two register compares serve the vote AND the loop's exit test (lanes still walking)
                const unsigned long long tri_mask = pt_ballot(W.n0 != 0), pair_mask = pt_ballot(W.pair != kNoPair);
                const uint32_t still = (uint32_t)pt_popc(tri_mask | pair_mask);
                if (still * 100u <= total * (uint32_t)P.spec_pct) break;      // (first pass: still == total, never true: spec_pct <= 99)
                // one kind of micro-step per trip, chosen by majority vote: the minority waits a trip
                // rather than making the whole wave run its block for a few lanes
                const int want_tri = pt_popc(tri_mask), want_node = pt_popc(pair_mask & ~tri_mask);
#ifdef PT_OCC
                {
                    const int run = want_tri >= want_node ? want_tri : want_node, lose = want_tri >= want_node ? want_node : want_tri;
                    const int waiting = pt_popc(pt_ballot(alive) & ~(tri_mask | pair_mask));
                    occ[0]++; occ[1] += (unsigned long long)run; occ[2] += (unsigned long long)lose; occ[3] += (unsigned long long)waiting;
                    occ[4] += (unsigned long long)(64 - run - lose - waiting);
                }
#endif
                if (want_tri >= want_node) {
                    if (sah_wants_tri(W)) {
                        if (kForm == 2) (void)sahw_tri_step<kSpill, true>(P, W, L.ro, L.rd, 0.001f, stack, L.n_tri);
                        else if (kForm != 0) (void)sahw_tri_step<kSpill>(P, W, L.ro, L.rd, 0.001f, stack, L.n_tri);
                        else (void)sah_tri_step(P, W, L.ro, L.rd, 0.001f, L.n_tri);
                    }
                } else {
                    if (sah_wants_node(W)) {
                        if (kForm == 2) (void)sahq_node_step<kTop, kSpill>(P, W, L.ro, L.rd, stack, L.n_box);
                        else if (kForm == 1) (void)sahw_node_step<kTop, kSpill>(P, W, L.ro, L.rd, stack, L.n_box);
                        else (void)sah_node_step(P, W, L.ro, L.rd, stack, L.n_box);
                    }
                }
                cost++;
            }
            if (alive && !sah_walking(W)) shade(L, P, V, sah_hit(P, W, L.ro, L.rd));
        }
        KernargParams &E = *fresh_kernarg();
        {
            int x0, y0, pixel_index;
            (void)tile_pixel(E, tile, tile_lane, x0, y0, pixel_index);
            if (in_frame) write_pixel(E, L.col, pixel_index);
        }
        // paths = samples started (all of them when depth == 0: camera rays are generated, then ray_color returns untraced)
        L.n_paths = in_frame ? (P.depth == 0 ? (uint32_t)P.spp : (uint32_t)L.s) : 0u;
        L.n_seg = 0;
#ifdef PT_OCC
        if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_occ[k], occ[k]);
#endif
        flush_counters(E, L, lane, 0ull, 0ull, wave_seg);
#ifdef PT_STALE
        {
            const uint32_t s5 = wave_sum(W.stale), s6 = wave_sum(W.visits);
            if (lane == 0) { atomicAdd(&E.counters[5], (unsigned long long)s5); atomicAdd(&E.counters[6], (unsigned long long)s6); }
        }
#endif
        if (E.tile_cost && lane == 0) {
            // a part stands for its tile with the longest part's cost, scaled to what the whole tile would have cost
            // (parts of 4 pixels measure ~0.8 of the tile's chain), so that the tile is split again next time
            if (is_part) atomicMax(&E.tile_cost[tile], cost + cost / 4u);
            else E.tile_cost[tile] = cost;
        }
#ifdef PT_TIMELINE
        if (lane == 0 && slot < 131072u) {
            g_timeline[4 * slot] = tl_start;
            g_timeline[4 * slot + 1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            g_timeline[4 * slot + 2] = (uint32_t)tile | (is_part ? 0x80000000u : 0u);
            g_timeline[4 * slot + 3] = cost;
        }
#endif
    }
}

// Two builds of the same body.  Its natural allocation is 93 VGPRs (5 waves/SIMD).  Scenes whose tree
// and triangles stay in cache are issue-bound and gain from a sixth wave even at the price of a few
// spills (C3 143 -> 136 ms); the 1 GB terrain is bound by memory latency per dependent step and loses
// to the spills (C4 84 -> 88 ms), so it keeps the natural allocation.
__global__ __launch_bounds__(kBlock) void render_kernel_sah(RenderParams P) { render_sah_body<0>(P); }
// the 4-wide form of the tree (RenderParams::sah == 2), cache-resident trees: top levels in LDS, short stack (24 KB per
// workgroup on the duck; whole stacks + the top levels would be 35 KB, four workgroups per CU), built for SIX waves per SIMD
// (80 VGPRs, 28 B of scratch per lane): such trees are bound by instruction issue, and with the LDS out of the way the sixth
// wave pays for its spills (C3 107.3 -> 102.5 ms; the 64-byte-node kernel, bound by the request path, gains nothing)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void render_kernel_sahw6(RenderParams P) { render_sah_body<1, 1, 1>(P); }
// ... and at five waves (the natural allocation is a register over: held to 96) for launches of less than twice the device's wave slots, whose
// chains the spills lengthen (a quarter of the duck's frame: 49.0 ms against 52.1 ms with the six-wave build)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void render_kernel_sahw(RenderParams P) { render_sah_body<1, 1, 1>(P); }
// ... for launches with fewer tiles than the device has wave slots (a rank's share of a multi-GPU frame): nothing but the walk
__global__ __launch_bounds__(kBlock) void render_kernel_sahw_lean(RenderParams P) { render_sah_body<1, 0, 0>(P); }
// the 4-wide form in 64-byte records (RenderParams::sah == 3), trees beyond the caches: short LDS stack + overflow area
// (five waves per SIMD asked for: with both records of a triangle pair in flight at once the allocation sits at the fifth
// wave's limit of 96 VGPRs; a sixth wave -- 80 VGPRs, 16 spilled -- loses: C4 47.6 -> 49.4 ms)
#ifndef PT_SAHQ_WAVES
#define PT_SAHQ_WAVES 5
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_SAHQ_WAVES, PT_SAHQ_WAVES))) void render_kernel_sahq(RenderParams P) { render_sah_body<2, 0, 1>(P); }
// either form with both options decided per launch (PTAMD_SAH_TOP / PTAMD_SAH_STACK / PTAMD_SAH_QUANT set by hand, and the tests)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void render_kernel_sahw_any(RenderParams P) { render_sah_body<1, 2, 2>(P); }
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void render_kernel_sahq_any(RenderParams P) { render_sah_body<2, 2, 2>(P); }

// ---- known-answer kernels (bit-exactness contract) -----------------------------------
__global__ void kat_arith_kernel(int op, const float *a, const float *b, float *o0, float *o1, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r0 = 0.0f, r1 = 0.0f;
    switch (op) {
    case 0: pt_sincosf(a[i], r0, r1); break;
    case 1: r0 = a[i] / b[i]; break;
    case 2: r0 = sqrtf(a[i]); break;
    case 3: r0 = (float)((double)a[i] / M_PI); break;
    case 4: {
        xorwow st;
        xorwow_init(st, (uint64_t)__float_as_uint(a[i]));
        unsigned skip = __float_as_uint(b[i]);
        for (unsigned k = 0; k < skip; k++) (void)xorwow_next(st);
        r0 = xorwow_uniform(st);
        break;
    }
    default: break;
    }
    o0[i] = r0;
    if (o1) o1[i] = r1;
}

// ---- exchange layout of a set of task rectangles (the native gather, pt_gather_frame) ------------------------------
// A device's rendered rectangles travel as ONE byte string, plane-major over its task list: the RGB rows of task 0,
// task 1, ..., then their Y rows, then the U samples they own (even framebuffer rows and columns), then the V samples.
// For full-width row bands this is byte for byte scheduling.pack_bands (the Python path's layout).
struct PackTask {
    int x0, x1, r0, r1;               // pixel columns [x0, x1), framebuffer rows [r0, r1) (row 0 = top)
    int k0, k1, c0, c1;               // chroma columns [k0, k1), chroma rows [c0, c1)
    size_t off_rgb, off_y, off_u, off_v;   // byte offsets of this task's four pieces in the packed string
};

template <bool kPack>
__global__ void pack_tasks_kernel(uint8_t *rgb, uint8_t *yuv, int W, int H, const PackTask *tasks, uint8_t *packed)
{
    const PackTask t = tasks[blockIdx.y];
    const size_t w = (size_t)(t.x1 - t.x0), h = (size_t)(t.r1 - t.r0), cw = (size_t)(t.k1 - t.k0), ch = (size_t)(t.c1 - t.c0);
    const size_t n_rgb = w * 3 * h, n_y = w * h, n_c = cw * ch, total = (size_t)W * H;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rgb + n_y + 2 * n_c; i += stride) {
        uint8_t *fb, *pk;
        if (i < n_rgb) {
            const size_t row = i / (w * 3), col = i % (w * 3);
            fb = rgb + ((size_t)(t.r0 + row) * W + t.x0) * 3 + col; pk = packed + t.off_rgb + i;
        } else if (i < n_rgb + n_y) {
            const size_t j = i - n_rgb, row = j / w, col = j % w;
            fb = yuv + (size_t)(t.r0 + row) * W + t.x0 + col; pk = packed + t.off_y + j;
        } else {
            size_t j = i - n_rgb - n_y;
            const int plane = j >= n_c ? 1 : 0;
            if (plane) j -= n_c;
            const size_t row = j / cw, col = j % cw;
            fb = yuv + total + (size_t)plane * (total / 4) + (size_t)(t.c0 + row) * (W / 2) + t.k0 + col;
            pk = packed + (plane ? t.off_v : t.off_u) + j;
        }
        if (kPack) *pk = *fb; else *fb = *pk;
    }
}

// Per tile row of a launch: the sum of its tiles' recorded costs (pt_band_costs_read).
__global__ void tile_row_cost_kernel(const uint32_t *cost, int tiles_x, unsigned long long *out)
{
    unsigned long long sum = 0;
    for (int i = threadIdx.x; i < tiles_x; i += blockDim.x) sum += cost[(size_t)blockIdx.x * tiles_x + i];
    sum = wave_sum64(sum);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// After a launch's costs are sorted (descending): how many of the dearest tiles the NEXT launch of this schedule hands
// to several waves each.  Nothing is split unless the dearest tile is critical: its chain alone is at least `pct` % of
// what the whole launch would take with perfectly even work (sum of costs / wave slots the device holds) -- such a tile,
// started first, still ends last.  Then every tile dearer than `ratio` % of the dearest is split (a part's chain is
// about `ratio` % of its tile's, so splitting cheaper tiles could not shorten the launch), at most `max_split` of them
// (the extra, partly idle waves must stay a small share of the slots: filling an under-occupied device with parts
// slows every chain on it).
__global__ void split_count_kernel(const uint32_t *sorted_cost, unsigned n, unsigned slots_resident,
                                   unsigned max_split, unsigned pct, unsigned ratio, uint32_t *n_split)
{
    __shared__ unsigned long long part[4];
    __shared__ uint32_t s_count;
    unsigned long long sum = 0;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) sum += sorted_cost[i];
    sum = wave_sum64(sum);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const unsigned long long total = part[0] + part[1] + part[2] + part[3];
    const unsigned long long even = total / (slots_resident ? slots_resident : 1u);      // cost units of an evenly spread launch
    const unsigned long long top = n ? sorted_cost[0] : 0u;
    uint32_t mine = 0;
    if (top != 0 && 100ull * top >= (unsigned long long)pct * even)
        for (unsigned i = threadIdx.x; i < n && i < max_split; i += blockDim.x) mine += 100ull * sorted_cost[i] > (unsigned long long)ratio * top ? 1u : 0u;
    if (mine) atomicAdd(&s_count, mine);
    __syncthreads();
    if (threadIdx.x == 0) *n_split = s_count;                         // sorted descending: those are the first entries of the order
}

// Before a launch that splits tiles: their cost entries start from zero (the parts record with atomicMax).  Not done by
// split_count_kernel, because pt_band_costs_read and the next sort still read the finished launch's costs after it.
__global__ void split_reset_kernel(const uint32_t *order, uint32_t *cost, const uint32_t *n_split)
{
    const uint32_t count = *n_split;
    for (unsigned i = threadIdx.x; i < count; i += blockDim.x) cost[order[i]] = 0;
}

__global__ void iota_kernel(uint32_t *a, uint32_t *b, unsigned n)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = i; if (b) b[i] = i; }
}

// All bit patterns first .. first+count-1 through a short form and the IEEE expression it replaces.
__global__ void kat_exhaustive_kernel(int op, uint32_t first, uint64_t count, unsigned long long *n_bad,
                                      uint32_t *examples, int max_examples)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t bits = first + (uint32_t)i;
        const float x = __uint_as_float(bits);
        float want, got;
        if (op == 0) { want = 1.0f / x; got = pt_rcp(x); }
        else if (op == 1) { want = (float)((double)x / M_PI); got = pt_div_pi(x); }
        else if (op == 2) { want = 1.0f / x; got = pt_rcp_not_tiny(x); if (__builtin_fabsf(x) <= 1e-8f) got = want; }
        else if (op == 3) { want = 1.0f / x; got = pt_rcp_core(x); }  // unguarded: shows the checker can fail
        else { want = sqrtf(x); got = pt_sqrt(x); }
        const bool same = (__float_as_uint(want) == __float_as_uint(got)) || (want != want && got != got);
        if (!same) {
            bad++;
            unsigned long long k = atomicAdd(&n_bad[1], 1ull);
            if (k < (unsigned long long)max_examples) examples[k] = bits;
        }
    }
    if (bad) atomicAdd(&n_bad[0], bad);
}

// uniform: the reference tree through the wave-uniform stack walk (traverse_uniform_stack: the same tests per ray in the same
// order as the per-lane walk, triangle and node records through the scalar cache): what makes a million rays through the
// unpruned walk of an 8 M-triangle reference tree a matter of seconds (the C4-scale gate, tests/test_gpu_c4.py).
__global__ void kat_hit_kernel(RenderParams P, const float *org, const float *dir, float *t_out,
                               int *face_out, size_t n, int uniform, uint32_t *box_out, uint32_t *tri_out)
{
    extern __shared__ float4 lds_raw[];
    const LdsLayout L = lds_carve<false>(P, lds_raw, (int)blockDim.x);
    const SceneView<false> S = stage_scene<false>(P, lds_raw, (int)threadIdx.x, (int)blockDim.x);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < n;
    size_t k = live ? i : 0;
    v3 o = mk3(org[3 * k], org[3 * k + 1], org[3 * k + 2]);
    v3 d = mk3(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
    uint32_t nb = 0, nt = 0;
    unsigned long long wb = 0, wt = 0;
    Hit h = P.sah ? traverse_sah(P, o, d, 0.001f, 3.402823466e+38f, L.stack + threadIdx.x, (int)blockDim.x, nb, nt)
          : uniform ? traverse_uniform_stack<1>(P, o, d, 0.001f, 3.402823466e+38f, L.stack + threadIdx.x, (int)blockDim.x, wb, wt)
                  : traverse<false, false>(S, o, d, 0.001f, 3.402823466e+38f, L.stack + threadIdx.x,
                                           reinterpret_cast<uint2 *>(L.queue) + threadIdx.x, (int)blockDim.x, P.queue_entries, nb, nt);
    if (live) {
        if (t_out) t_out[i] = h.slot >= 0 ? h.t : -1.0f;
        int face = -1;
        if (h.slot >= 0) face = P.shade[h.slot].face;
        if (face_out) face_out[i] = face;
        if (box_out) box_out[i] = uniform ? (uint32_t)wb : nb;
        if (tri_out) tri_out[i] = uniform ? (uint32_t)wt : nt;
    }
}

} // namespace dev

// ======================================================================== host code ==

// camera::recalculate_camera_params, camera.h:21-36: uniform per launch, so it runs on the
// host (the reference recomputes it in every thread for every sample, racing on one object).
CameraBasis camera_basis(const pt_camera &cfg)
{
    CameraBasis cam;
    v3 from = mk3(cfg.lookFrom[0], cfg.lookFrom[1], cfg.lookFrom[2]);
    v3 front = mk3(cfg.front[0], cfg.front[1], cfg.front[2]);
    v3 lookAt = from + front;
    float theta_v = (float)((double)cfg.vfov * M_PI / 180);
    float half_height = pt_tanf(theta_v / 2);
    float theta_h = (float)((double)cfg.hfov * M_PI / 180);
    float half_width = pt_tanf(theta_h / 2);
    v3 vup = mk3(0.0f, 1.0f, 0.0f);
    v3 w = normalize(from - lookAt);
    v3 u = normalize(cross(vup, w));
    v3 v = cross(w, u);
    cam.origin = from;
    cam.llc = ((from - half_width * u) - half_height * v) - w;
    cam.horizontal = (2 * half_width) * u;
    cam.vertical = (2 * half_height) * v;
    return cam;
}

} // namespace pt

using namespace pt;

#define PT_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return set_error(PT_ERR_HIP, std::string("HIP ERROR = ") + std::to_string((int)e_) + \
                                             " (" + hipGetErrorString(e_) + ") at " + __FILE__ +  \
                                             ":" + std::to_string(__LINE__) + " '" #call "'");   \
    } while (0)

// Temporary device allocations of one C-ABI call: freed on every return path, PT_HIP's early returns included.
struct DevBufs {
    std::vector<void *> ptrs;
    ~DevBufs() { for (void *p : ptrs) (void)hipFree(p); }
    hipError_t alloc(void **out, size_t bytes)
    {
        hipError_t e = hipMalloc(out, bytes ? bytes : 1);
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
    void release() { ptrs.clear(); }          // ownership passes to the caller
};

// Adaptive tile schedule of one task geometry (pt_set_tile_schedule): the render kernel records what
// every 8x8 tile cost, a device radix sort turns that into next launch's wave-slot -> tile order
// (dearest first), so the launch ends on cheap tiles instead of on a straggler that started last.
struct TileSchedule {
    int key[11];
    unsigned slots = 0;
    void *base = nullptr;         // one stream-ordered allocation, carved into the buffers below
    uint32_t *order = nullptr, *cost = nullptr, *sorted_cost = nullptr, *iota = nullptr;
    uint32_t *n_split = nullptr;  // how many of the dearest tiles the next launch splits (split_count_kernel)
    void *temp = nullptr;
    size_t temp_bytes = 0;
    bool has_order = false;
    hipEvent_t last_use = nullptr;
    hipStream_t last_stream = nullptr;
    bool used = false;
    int busy = 0;                 // launches that hold this entry outside dev->mu (between choosing it and recording last_use)
    uint64_t stamp = 0;
    uint64_t id = 0;              // unique per allocation: a recycled ADDRESS must never pass for the schedule it replaced
};

struct pt_device {
    int index = 0;
    // scene (device)
    NodePair *d_nodes = nullptr;
    TriIsect *d_tris = nullptr;
    TriShade *d_shade = nullptr;
    float4 *d_onb = nullptr;      // onb(normal) per slot, 3 x float4 (small scenes: the dealt kernel's LDS table)
    LightRec *d_lights = nullptr;
    MatRec *d_mats = nullptr;
    TexDesc *d_texs = nullptr;
    float *d_texels = nullptr;
    unsigned long long *d_counters = nullptr;
    bool has_scene = false;
    int n_lights = 0;
    // scene (host mirrors for introspection)
    RefBvh bvh;
    std::vector<int> light_faces;
    size_t n_tris = 0, n_pairs = 0, n_mats = 0, sah_stack = 0;
    bool lds_scene = false;       // node + triangle arrays staged in LDS by every workgroup
    bool packed_queue = false;    // leaf-queue entries fit one dword (< 65536 triangles)
    int walk = 0;                 // traversal schedule: 0 per-lane, 1 uniform/mask, 2 uniform/stack, 3 SAH tree
    bool sah = false;             // uploaded with the opt-in SAH tree instead of the reference topology
    bool sah_wide = true;         // upload the SAH tree in its 4-wide form (PTAMD_SAH_WIDE=0: the pair form)
    bool sah_wide_on = false;     // ... and the uploaded scene has it
    int sah_top = 64;             // 4-wide forms: this many top-level records (breadth first) go first and are staged in LDS (PTAMD_SAH_TOP; 0 = none)
    uint32_t sah_top_moved = 0;   // ... how many the uploaded scene has in front
    bool sah_by_hand = false;     // PTAMD_SAH_TOP / PTAMD_SAH_STACK given: both options as asked, in the kernels that decide them per launch
    int sah_quant = -1;           // the 4-wide form in 64-byte records (QuadQ): -1 = for trees beyond 64 MB, 0 = never, 1 = always (PTAMD_SAH_QUANT)
    bool sah_quant_on = false;    // ... and the uploaded scene has it
    bool sah_persistent = true;   // SAH walk: wave keeps stragglers' traversal state across trips
    int spec_pct = -1;            // ... and leaves the walk loop when <= this % of the walkers remain; -1 = by scene size (fill_params)
    bool deal_cap_by_hand = false; // PTAMD_DEAL_CAP given
    int deal_cap = 768, deal_min = 48;   // dealt walk (kWalk 4): item list capacity per wave; in-place threshold (lanes)
    int lrpt = 1;                 // dealt kernel: issue priority by remaining samples (PTAMD_LRPT=0: off)
    bool deal_auto = true;        // automatic choice prefers the dealt walk over the plain mask walk (PTAMD_DEAL=0: off)
    bool any_tex = false;
    int fold_bits = 31;           // per-lane test counters fold at 2^fold_bits (PTAMD_FOLD_BITS lowers it: tests of the folding)
    int bvh_mode = 0;             // pt_set_bvh_mode: applies to the next pt_upload_scene
    int force_walk = -1;          // pt_set_traversal_mode: -1 auto, else the requested schedule
    int queue_entries = 8;        // per-lane leaf queue capacity (phase A -> phase B)
    // framebuffer binding
    int W = 0, H = 0;
    uint8_t *rgb = nullptr, *yuv = nullptr;
    unsigned block_x = 8, block_y = 8;
    // extension: legacy material classes + spheres (pt_set_legacy_scene), applied by the next upload
    std::vector<pt_legacy_material> legacy_next;
    std::vector<pt_sphere> spheres_next;
    float4 *d_legacy = nullptr, *d_spheres = nullptr;
    int n_spheres = 0;
    bool legacy_on = false;       // the uploaded scene uses a legacy material or a sphere
    int coop_mode = -1;           // cooperative kernel: -1 auto (stack walk, big leaf, small launch), 0 off, 1 on
    int max_waves = 8192;         // wave slots of the device: CUs x 4 SIMDs x 8
    size_t lds_limit = 64 * 1024; // LDS one workgroup may ask for (hipDeviceProp_t::maxSharedMemoryPerMultiProcessor)
    std::vector<std::pair<const void *, size_t>> lds_raised;   // kernels whose dynamic-LDS cap was raised, and to what
    // adaptive tile schedule
    int tile_schedule = 1;        // 0 raster order, 1 cost-ordered from the previous launch of the same task
    std::vector<TileSchedule *> schedules;
    uint64_t schedule_clock = 0;
    // kernel timing
    std::mutex mu;
    // per launch: the two timing events and, for the persistent kernels, the counter its waves take their tiles from.
    // The triple is reused only after the launch's second event has completed, so launches in flight on different
    // streams never share a counter.
    struct LaunchEvents { hipEvent_t first = nullptr, second = nullptr; uint32_t *next = nullptr;
                          uint32_t *spill = nullptr; size_t spill_bytes = 0; };      // spill: the launch's SahStack overflow area, grown on demand
    std::vector<LaunchEvents> pending, free_events;
    std::vector<uint32_t *> next_chunks;      // tile counters, 256 per allocation, one 64-byte line each
    size_t next_used = 0;
    int cu_count = 256;
    int sched_min_blocks = 64;    // launches with fewer workgroups keep raster order (PTAMD_SCHED_MIN_BLOCKS: the fuzzer lowers it)
    int split_ratio = 0;          // ... and every tile dearer than this share of the dearest (0: 65 % dealt, 80 % SAH; PTAMD_SPLIT_RATIO)
    int split_pct = 0;           // ... when the dearest costs at least this % of an evenly spread launch's share per slot (0: 110 dealt, 90 SAH; PTAMD_SPLIT_PCT)
    int deal_split = 4;           // dealt kernel: likewise (PTAMD_DEAL_SPLIT: 0 off, 2, 4, 8)
    bool part_lanes = true;       // dealt mask walk: parts of split tiles test all child boxes at once, lanes = (ray, box) (PTAMD_PART_LANES)
    int sah_split = 16;           // SAH kernel: waves a critically dear tile is handed to (PTAMD_SAH_SPLIT: 0 off, 16, 32, 64)
    int lean_pct = 100;           // SAH tree, cache-resident: launches of up to this % of the device's wave slots take the lean kernel (PTAMD_LEAN_PCT)
    bool deal_dense = true;       // dealt mask walk: the 8-wave build for launches of more than 2 x the device's wave slots (PTAMD_DEAL_DENSE=0: never)
    int resident_cap = 0;         // persistent kernels: at most this many workgroups per CU (PTAMD_RESIDENT; 0 = what fits)
    // Reference trees of up to this many inner nodes take the wave-uniform stack walks (PTAMD_STACK_MAX_PAIRS).  Rounds 1-3 sent
    // trees beyond 4 096 inner nodes to the per-lane walk ("a wave pays for the UNION of its lanes' paths"); but the reference's
    // tree does not prune and its boxes contain the origin, so every ray visits a good part of it anyway, and the uniform walk's
    // scalar operands beat the per-lane gather at every size measured (profiles/r04_scene_sweep.json: the terrain at 73 k ..
    // 524 k triangles 4.4 / 10.7 / 29.1 / 58.0 s per-lane against 1.66 / 3.12 / 7.50 / 13.1 s): no limit any more.
    int stack_max_pairs = 1 << 28;
    int sah_stack_lds = 16;       // persistent SAH kernel: stack entries per lane kept in LDS, deeper ones spill to device memory (PTAMD_SAH_STACK: 0 = all in LDS, else 11..64)
    struct Resident { const void *kern; size_t lds; int blocks; };
    std::vector<Resident> resident_blocks;      // per kernel and LDS size: workgroups one CU holds (persistent grids)
    double ms_total = 0.0;
    uint64_t launches = 0;
    const char *last_kernel = "";  // what the latest render launch ran (pt_kernel_name_read)
    // latest row-band launch, for pt_band_costs_read: its schedule (null = no costs recorded) and geometry
    // The latest row-band launch of every stream (pt_band_costs_read / _on): its schedule (by id; 0 = no costs recorded) and
    // geometry.  A host that keeps several frames in flight, each on its own stream, reads the costs of the frame it has just
    // completed, not those of the newest launch.
    struct BandLaunch { hipStream_t stream; uint64_t sched_id; int tiles_x, tiles_y, band_rows, local_rows; uint64_t stamp; };
    std::vector<BandLaunch> band_launches;
    uint64_t band_clock = 0;
    uint64_t last_sched_id = 0;              // TileSchedule::id of the latest launch's schedule (pt_split_tiles_read), and its stream
    uint64_t schedule_ids = 0;
    hipStream_t last_sched_stream = nullptr;
};

// Traversal schedule for the uploaded tree.  The wave-uniform walks cost the UNION of the 64
// lanes' paths per ray segment, which is fine while the tree is small (incoherent rays touch
// most of it anyway under the reference's no-pruning traversal) and ruinous for big trees.
static int choose_walk(const pt_device *d)
{
    const bool mask_ok = d->n_pairs <= 64, stack_ok = d->n_pairs <= (size_t)d->stack_max_pairs;
    // item = lane | slot << 6 in 16 bits; the dealt kernel always stages the shading tables in LDS (fill_params' rule)
    const bool deal_ok = mask_ok && d->n_tris <= 1024 && (d->n_tris * 48 + d->n_mats * 48 + (size_t)d->n_lights * 80) <= 16 * 1024;
    if (d->legacy_on) return d->force_walk <= 0 ? 0 : -1;      // the extension is compiled into the per-lane walk only
    if (d->sah) return 3;                     // the SAH tree has its own walk
    if (d->force_walk == 0) return 0;
    if (d->force_walk == 1) return mask_ok ? 1 : -1;
    if (d->force_walk == 2) return stack_ok ? 2 : -1;
    // stack walk with dealt sparse leaves: key = t bits | rank << 16 | slot, both in 16 bits
    const bool deal_stack_ok = stack_ok && d->n_tris <= 65536;
    if (d->force_walk == 4) return deal_ok ? 4 : -1;
    if (d->force_walk == 5) return deal_stack_ok ? 5 : -1;
    return deal_ok && d->deal_auto ? 4 : (mask_ok ? 1 : (stack_ok ? (deal_stack_ok && d->deal_auto ? 5 : 2) : 0));
}

// `hs` != null: stream-ordered free (the schedule's last launch has finished); null: plain free after
// a device synchronisation.
static void free_schedule(TileSchedule *t, hipStream_t hs = nullptr, bool async = false)
{
    if (t->base) { if (async) (void)hipFreeAsync(t->base, hs); else (void)hipFree(t->base); }
    if (t->last_use) (void)hipEventDestroy(t->last_use);
    delete t;
}

// Caller holds dev->mu and has synchronised the device (or knows nothing is in flight).
static void drop_schedules(pt_device *d)
{
    for (TileSchedule *t : d->schedules) free_schedule(t);
    d->schedules.clear();
}

static void free_scene(pt_device *d)
{
    (void)hipFree(d->d_nodes); (void)hipFree(d->d_tris); (void)hipFree(d->d_shade); (void)hipFree(d->d_onb); d->d_onb = nullptr;
    (void)hipFree(d->d_lights); (void)hipFree(d->d_mats); (void)hipFree(d->d_texs);
    (void)hipFree(d->d_texels); (void)hipFree(d->d_legacy); (void)hipFree(d->d_spheres);
    d->d_legacy = nullptr; d->d_spheres = nullptr; d->n_spheres = 0; d->legacy_on = false;
    d->d_nodes = nullptr; d->d_tris = nullptr; d->d_shade = nullptr; d->d_lights = nullptr;
    d->d_mats = nullptr; d->d_texs = nullptr; d->d_texels = nullptr;
    d->has_scene = false;
}

template <typename T>
static pt_status upload_vec(const std::vector<T> &v, T **out)
{
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    PT_HIP(hipMalloc((void **)out, bytes));
    if (!v.empty()) PT_HIP(hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

static pt_status drain_events(pt_device *d)
{
    for (auto &pr : d->pending) {
        float ms = 0.0f;
        PT_HIP(hipEventSynchronize(pr.second));
        PT_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        d->ms_total += (double)ms;
        d->launches++;
        d->free_events.push_back(pr);
    }
    d->pending.clear();
    return PT_OK;
}

extern "C" {

pt_status pt_device_create(int device_index, pt_device **out)
try {
    clear_error();
    if (!out) return set_error(PT_ERR_INVALID, "null out pointer");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_error(PT_ERR_NO_DEVICE, "no HIP device visible: the gfx950 kernels are mandatory, there is no CPU fallback");
    if (device_index < 0 || device_index >= count) return set_error(PT_ERR_INVALID, "device index out of range");
    PT_HIP(hipSetDevice(device_index));
    std::unique_ptr<pt_device> d(new pt_device());
    d->index = device_index;
    // diagnostic A/B switches for the sweeps under tools/ (DESIGN.md section 4b); none changes a byte of output,
    // out-of-range values are ignored
    if (getenv("PTAMD_SAH_SIMPLE")) d->sah_persistent = false;
    if (const char *e = getenv("PTAMD_SAH_WIDE")) d->sah_wide = atoi(e) != 0;
    if (const char *e = getenv("PTAMD_SAH_TOP")) { const int v = atoi(e); if (v >= 0 && v <= 1024) { d->sah_top = v; d->sah_by_hand = true; } }
    if (const char *e = getenv("PTAMD_SAH_QUANT")) { const int v = atoi(e); if (v >= -1 && v <= 1) d->sah_quant = v; }
    if (const char *e = getenv("PTAMD_SPEC_PCT")) { const int v = atoi(e); if (v >= 0 && v <= 99) d->spec_pct = v; }   // 100 would leave the walk loop before its first step
    if (const char *e = getenv("PTAMD_DEAL")) d->deal_auto = atoi(e) != 0;
    if (const char *e = getenv("PTAMD_LRPT")) d->lrpt = atoi(e) != 0;
    if (const char *e = getenv("PTAMD_DEAL_MIN")) { const int v = atoi(e); if (v >= 0 && v <= 65) d->deal_min = v; }
    if (const char *e = getenv("PTAMD_DEAL_CAP")) { const int v = atoi(e); if (v >= 64 && v <= 8192) { d->deal_cap = v & ~7; d->deal_cap_by_hand = true; } }
    if (const char *e = getenv("PTAMD_SCHED_MIN_BLOCKS")) { const int v = atoi(e); if (v >= 1 && v <= 1 << 20) d->sched_min_blocks = v; }
    if (const char *e = getenv("PTAMD_SPLIT_RATIO")) { const int v = atoi(e); if (v >= 1 && v <= 100) d->split_ratio = v; }
    if (const char *e = getenv("PTAMD_SPLIT_PCT")) { const int v = atoi(e); if (v >= 10 && v <= 1000) d->split_pct = v; }
    if (const char *e = getenv("PTAMD_DEAL_SPLIT")) { const int v = atoi(e); if (v == 0 || v == 2 || v == 4 || v == 8) d->deal_split = v; }
    if (const char *e = getenv("PTAMD_PART_LANES")) d->part_lanes = atoi(e) != 0;
    if (const char *e = getenv("PTAMD_SAH_SPLIT")) { const int v = atoi(e); if (v == 0 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) d->sah_split = v; }
    if (const char *e = getenv("PTAMD_SAH_STACK")) { const int v = atoi(e); if (v == 0 || (v >= 11 && v <= 64)) { d->sah_stack_lds = v; d->sah_by_hand = true; } }
    if (const char *e = getenv("PTAMD_STACK_MAX_PAIRS")) { const int v = atoi(e); if (v >= 64 && v <= (1 << 28)) d->stack_max_pairs = v; }
    if (const char *e = getenv("PTAMD_LEAN_PCT")) { const int v = atoi(e); if (v >= 0 && v <= 1000) d->lean_pct = v; }
    if (const char *e = getenv("PTAMD_DEAL_DENSE")) d->deal_dense = atoi(e) != 0;
    if (const char *e = getenv("PTAMD_RESIDENT")) { const int v = atoi(e); if (v >= 1 && v <= 16) d->resident_cap = v; }
    if (const char *e = getenv("PTAMD_FOLD_BITS")) { const int v = atoi(e); if (v >= 8 && v <= 31) d->fold_bits = v; }
    if (const char *e = getenv("PTAMD_COOP")) { const int v = atoi(e); if (v >= -1 && v <= 1) d->coop_mode = v; }
    {
        hipDeviceProp_t prop;
        PT_HIP(hipGetDeviceProperties(&prop, device_index));
        d->max_waves = std::max(1, prop.multiProcessorCount) * 4 * 8;
        d->cu_count = std::max(1, prop.multiProcessorCount);
        d->lds_limit = std::max<size_t>((size_t)64 * 1024, std::max((size_t)prop.sharedMemPerBlock, (size_t)prop.maxSharedMemoryPerMultiProcessor));
    }
    PT_HIP(hipMalloc((void **)&d->d_counters, 8 * sizeof(unsigned long long)));
    PT_HIP(hipMemset(d->d_counters, 0, 8 * sizeof(unsigned long long)));
    *out = d.release();
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_device_destroy(pt_device *dev)
try {
    clear_error();
    if (!dev) return PT_OK;
    (void)hipSetDevice(dev->index);
    (void)hipDeviceSynchronize();
    free_scene(dev);
    drop_schedules(dev);
    (void)hipFree(dev->d_counters);
    for (auto &pr : dev->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); (void)hipFree(pr.spill); }
    for (auto &pr : dev->free_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); (void)hipFree(pr.spill); }
    for (uint32_t *c : dev->next_chunks) (void)hipFree(c);
    delete dev;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_upload_scene(pt_device *dev, const pt_triangle *tris, size_t n, const pt_material *mats,
                          size_t m, const pt_texture *texs, size_t k)
try {
    clear_error();
    if (!dev || !tris || !mats || n == 0 || m == 0 || (k && !texs))
        return set_error(PT_ERR_INVALID, "scene needs at least one triangle and one material");
    if (n > (size_t)1 << 30) return set_error(PT_ERR_INVALID, "too many triangles");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipDeviceSynchronize());

    // loadMaterials, DevicePathTracer.h:267-291, incl. the texture-pointer inheritance quirk
    // (:269-279: the two pointers are declared outside the loop and never reset).
    std::vector<MatRec> mrec(m);
    int baseTex = -1, emisTex = -1;
    for (size_t i = 0; i < m; i++) {
        if (mats[i].baseColorTextureIdx >= 0) baseTex = mats[i].baseColorTextureIdx;
        if (mats[i].emissiveTextureIdx >= 0) emisTex = mats[i].emissiveTextureIdx;
        if (baseTex >= (int)k || emisTex >= (int)k) return set_error(PT_ERR_INVALID, "material texture index out of range");
        MatRec &r = mrec[i];
        r.br = mats[i].baseColor[0]; r.bg = mats[i].baseColor[1]; r.bb = mats[i].baseColor[2];
        r.er = mats[i].emissiveFactor[0]; r.eg = mats[i].emissiveFactor[1]; r.eb = mats[i].emissiveFactor[2];
        r.baseTex = baseTex; r.emisTex = emisTex;
        r.e50r = r.er * 50.0f; r.e50g = r.eg * 50.0f; r.e50b = r.eb * 50.0f;      // material.h:85
        r.is_light = (r.e50r > 0.0001f || r.e50g > 0.0001f || r.e50b > 0.0001f) ? 1 : 0;
    }
    // loadTextures, DevicePathTracer.h:241-265: all texels in one pool
    std::vector<TexDesc> tdesc(k);
    std::vector<float> pool;
    for (size_t i = 0; i < k; i++) {
        if (texs[i].width < 0 || texs[i].height < 0) return set_error(PT_ERR_INVALID, "negative texture size");
        size_t cnt = (size_t)texs[i].width * (size_t)texs[i].height * 3;
        if (cnt && !texs[i].data) return set_error(PT_ERR_INVALID, "texture without data");
        tdesc[i].w = texs[i].width; tdesc[i].h = texs[i].height;
        tdesc[i].offset = (uint32_t)pool.size(); tdesc[i].pad = 0;
        pool.insert(pool.end(), texs[i].data, texs[i].data + cnt);
    }
    // loadTrianglesWithTextures, DevicePathTracer.h:293-309: faces + light list in face order
    std::vector<float> pos(9 * n);
    std::vector<int> lights;
    for (size_t i = 0; i < n; i++) {
        if (tris[i].materialIdx < 0 || (size_t)tris[i].materialIdx >= m)
            return set_error(PT_ERR_INVALID, "triangle material index out of range");
        memcpy(&pos[9 * i], tris[i].v0.position, 12);
        memcpy(&pos[9 * i + 3], tris[i].v1.position, 12);
        memcpy(&pos[9 * i + 6], tris[i].v2.position, 12);
        const float *e = mats[tris[i].materialIdx].emissiveFactor;
        if (e[0] > 0.0001f || e[1] > 0.0001f || e[2] > 0.0001f) lights.push_back((int)i);   // :305
    }
    // extension: legacy material kinds / spheres for this upload
    if (!dev->legacy_next.empty() && dev->legacy_next.size() != m)
        return set_error(PT_ERR_INVALID, "pt_set_legacy_scene: one legacy record per material is required");
    for (const pt_sphere &sp : dev->spheres_next)
        if (sp.materialIdx < 0 || (size_t)sp.materialIdx >= m) return set_error(PT_ERR_INVALID, "sphere material index out of range");
    bool legacy_on = !dev->spheres_next.empty();
    for (const pt_legacy_material &lm : dev->legacy_next) legacy_on = legacy_on || lm.kind != 0;
    auto is_universal = [&](int mat) { return dev->legacy_next.empty() || dev->legacy_next[(size_t)mat].kind == 0; };
    if (lights.empty()) {
        // the live material samples the light list on every bounce (hitable_list.h:23-26): an empty list
        // is only tolerable when nothing in the scene uses that material
        bool any_universal = false;
        for (size_t i = 0; i < n; i++) any_universal = any_universal || is_universal(tris[i].materialIdx);
        for (const pt_sphere &sp : dev->spheres_next) any_universal = any_universal || is_universal(sp.materialIdx);
        if (any_universal)
            return set_error(PT_ERR_NO_LIGHTS, "scene has no emissive face: the reference's light sampling dereferences an empty list (hitable_list.h:23-26)");
    }
    if (legacy_on && dev->bvh_mode == 1)
        return set_error(PT_ERR_UNSUPPORTED, "the legacy-material / sphere extension runs on the reference tree only (pt_set_bvh_mode 0)");

    // create_world, DevicePathTracer.h:134-146 -> BVH::BVH, bvh.h:20-33, on the host
    const bool sah = dev->bvh_mode == 1;
    // The reference tree is also built under the SAH option: its visit order ranks the triangles for
    // exact-tie resolution (8.4 M triangles: ~2 s on 8 host cores, pt_bvh.cpp).  Beyond 2^25 triangles it is
    // skipped and ties resolve by face index.
    const bool build_ref = !sah || n <= (size_t)1 << 25;
    RefBvh bvh;
    if (build_ref) {
        build_reference_bvh(pos.data(), n, bvh);
        if (!sah && bvh.max_stack > 64)
            return set_error(PT_ERR_STACK, "BVH needs more than the reference's 64 traversal-stack entries (bvh.h:179)");
    } else {
        bvh.nodes.assign(1, RefNode());
        bvh.nodes[0].triCount = (uint32_t)n;      // placeholder: "one leaf", never traversed
        bvh.indices.resize(n);
        for (size_t i = 0; i < n; i++) bvh.indices[i] = (int)i;
    }
    SahBvh sahbvh;
    bool sah_wide_on = false, sah_quant_on = false;
    uint32_t sah_top_moved = 0;
    if (sah) {
        uint32_t max_leaf = sah_leaf_size_for(n);
        if (const char *e = getenv("PTAMD_SAH_LEAF")) { const int v = atoi(e); if (v >= 1 && v <= 15) max_leaf = (uint32_t)v; }
        build_sah_bvh(pos.data(), n, sahbvh, max_leaf);
        if (sahbvh.max_stack > 64) return set_error(PT_ERR_STACK, "SAH tree deeper than 64 levels");
    }

    // flatten: inner nodes -> NodePair records, triangles -> leaf order
    std::vector<uint32_t> pair_of(bvh.nodes.size(), 0xFFFFFFFFu);
    uint32_t n_pairs = 0;
    {
        // number the inner nodes in DFS pre-order (node, left subtree, right subtree) = the order in
        // which the reference traversal can reach them; child links therefore always point forward
        std::vector<uint32_t> todo;
        if (bvh.nodes[0].triCount == 0) todo.push_back(0);
        while (!todo.empty()) {
            uint32_t id = todo.back();
            todo.pop_back();
            pair_of[id] = n_pairs++;
            const RefNode &nd = bvh.nodes[id];
            if (bvh.nodes[nd.right].triCount == 0) todo.push_back(nd.right);
            if (bvh.nodes[nd.left].triCount == 0) todo.push_back(nd.left);
        }
    }
    bool root_leaf = bvh.nodes[0].triCount != 0;
    std::vector<NodePair> pairs(root_leaf ? 1 : n_pairs);
    auto fill_child = [&](const RefNode &c, uint32_t ci, float *box6, uint32_t &link, uint32_t &cnt) {
        box6[0] = c.xmin; box6[1] = c.xmax; box6[2] = c.ymin; box6[3] = c.ymax; box6[4] = c.zmin; box6[5] = c.zmax;
        cnt = c.triCount;
        link = c.triCount ? c.firstTri : pair_of[ci];
    };
    auto fill_pair = [&](NodePair &p, uint32_t li, uint32_t ri) {
        float lb[6], rb[6];
        fill_child(bvh.nodes[li], li, lb, p.llink, p.lcnt);
        fill_child(bvh.nodes[ri], ri, rb, p.rlink, p.rcnt);
        p.lxmin = lb[0]; p.lxmax = lb[1]; p.lymin = lb[2]; p.lymax = lb[3]; p.lzmin = lb[4]; p.lzmax = lb[5];
        p.rxmin = rb[0]; p.rxmax = rb[1]; p.rymin = rb[2]; p.rymax = rb[3]; p.rzmin = rb[4]; p.rzmax = rb[5];
    };
    if (root_leaf) {
        // bvh.h:26,:189-190: an unsplit root has left = right = 0, i.e. it is its own two children.
        fill_pair(pairs[0], 0, 0);
    } else {
        for (size_t i = 0; i < bvh.nodes.size(); i++)
            if (bvh.nodes[i].triCount == 0) fill_pair(pairs[pair_of[i]], bvh.nodes[i].left, bvh.nodes[i].right);
    }
    // rank[face] = position of the face in the reference's visit order (pairs in DFS pre-order, left
    // leaf before right leaf, triangles in `indices` order): the order in which bvh.h:178-246 meets
    // the triangles and therefore the tie-breaker between equal t values.
    std::vector<uint32_t> rank(n);
    {
        uint32_t next = 0;
        auto rank_leaf = [&](const RefNode &lf) { for (uint32_t k = 0; k < lf.triCount; k++) rank[(size_t)bvh.indices[lf.firstTri + k]] = next++; };
        if (root_leaf || !build_ref) {
            for (size_t i = 0; i < n; i++) rank[(size_t)bvh.indices[i]] = (uint32_t)i;
        } else {
            std::vector<uint32_t> todo{0};
            while (!todo.empty()) {
                uint32_t id = todo.back(); todo.pop_back();
                const RefNode &nd = bvh.nodes[id];
                const RefNode &l = bvh.nodes[nd.left], &r = bvh.nodes[nd.right];
                if (l.triCount) rank_leaf(l);
                if (r.triCount) rank_leaf(r);
                if (r.triCount == 0) todo.push_back(nd.right);
                if (l.triCount == 0) todo.push_back(nd.left);
            }
        }
    }
    if (sah) {
        // the SAH walk addresses records by 32-bit byte offsets (pt_walk.h SahWalk): scale the builder's links
        if (n * sizeof(TriIsect) >= ((size_t)1 << 32) || sahbvh.pairs.size() * sizeof(NodePair) >= ((size_t)1 << 32))
            return set_error(PT_ERR_UNSUPPORTED, "SAH tree: node and triangle records are addressed by 32-bit byte offsets (< 89 M triangles)");
        uint32_t wide_stack = 0;
        sah_wide_on = dev->sah_wide && collapse_sah_quads(sahbvh, pairs, wide_stack) && wide_stack <= 64;
        if (sah_wide_on) {
            sahbvh.max_stack = wide_stack;       // entries a walk of the 4-wide form can have pushed at once
            sah_top_moved = dev->sah_top > 0 ? sah_quads_top_levels_first(pairs, (uint32_t)dev->sah_top) : 0u;
            // Trees that do not fit the caches take the 4-wide form at half the bytes (QuadQ: 8-bit child planes, rounded
            // outwards): a visit is four per-lane 16-byte requests instead of seven, and on the 8.4 M-triangle terrain the
            // vector-memory path was 70 % busy serving them (profiles/r03_sah_lane_census.json).  A visit is 144
            // VALU instructions (byte -> ray parameter by one conversion and one fma, pt_walk.h sahq_node_eval) against 100 for the
            // float planes.  Cache-resident trees keep the 128-byte form (plane-major, below): with their top levels in LDS it is
            // the faster one there (C3 100.8 ms against 108.1 ms for the 64-byte form, profiles/r04_c4_visit_slope.json).
            const size_t wide_bytes = pairs.size() * sizeof(NodePair) + n * sizeof(TriIsect);
            if (dev->sah_quant == 1 || (dev->sah_quant < 0 && wide_bytes > ((size_t)64 << 20))) {
                std::vector<QuadQ> qq;
                if (quantise_sah_quads(pairs, qq)) {
                    static_assert(sizeof(QuadQ) == sizeof(NodePair), "a QuadQ record travels as one NodePair-sized record");
                    pairs.resize(qq.size());
                    memcpy(pairs.data(), qq.data(), qq.size() * sizeof(QuadQ));
                    sah_quant_on = true;
                }
            }
            if (!sah_quant_on) sah_quads_plane_major(pairs);      // the float form travels plane-major (QuadF)
        } else {
            pairs = sahbvh.pairs;
            for (NodePair &q : pairs) {
                q.llink *= q.lcnt ? (uint32_t)sizeof(TriIsect) : (uint32_t)sizeof(NodePair);
                q.rlink *= q.rcnt ? (uint32_t)sizeof(TriIsect) : (uint32_t)sizeof(NodePair);
            }
        }
    }
    const std::vector<int> &slot_face = sah ? sahbvh.order : bvh.indices;
    std::vector<TriIsect> isect(n);
    std::vector<TriShade> shade(n);
    auto make_isect = [&](const pt_triangle &t, float *out12, float *area, float *v1v2) {
        v3 v0 = mk3(t.v0.position[0], t.v0.position[1], t.v0.position[2]);
        v3 v1 = mk3(t.v1.position[0], t.v1.position[1], t.v1.position[2]);
        v3 v2 = mk3(t.v2.position[0], t.v2.position[1], t.v2.position[2]);
        v3 e1 = v1 - v0, e2 = v2 - v0;                     // triangle.h:66-67
        v3 cr = cross(e1, e2);
        v3 nn = normalize(cr);                             // triangle.h:102
        out12[0] = v0.x; out12[1] = v0.y; out12[2] = v0.z; out12[3] = e1.x; out12[4] = e1.y; out12[5] = e1.z;
        out12[6] = e2.x; out12[7] = e2.y; out12[8] = e2.z; out12[9] = nn.x; out12[10] = nn.y; out12[11] = nn.z;
        if (area) *area = length(cr) * 0.5f;               // triangle.h:28
        if (v1v2) { v1v2[0] = v1.x; v1v2[1] = v1.y; v1v2[2] = v1.z; v1v2[3] = v2.x; v1v2[4] = v2.y; v1v2[5] = v2.z; }
    };
    for (size_t s = 0; s < n; s++) {
        int face = slot_face[s];
        const pt_triangle &t = tris[face];
        make_isect(t, &isect[s].v0x, nullptr, nullptr);
        TriShade &sh = shade[s];
        sh.u0 = t.v0.texCoords[0]; sh.v0 = t.v0.texCoords[1];
        sh.u1 = t.v1.texCoords[0]; sh.v1 = t.v1.texCoords[1];
        sh.u2 = t.v2.texCoords[0]; sh.v2 = t.v2.texCoords[1];
        sh.material = t.materialIdx; sh.face = face;
        sh.nx = isect[s].nx; sh.ny = isect[s].ny; sh.nz = isect[s].nz; sh.pad = 0.0f;
        memcpy(&isect[s].nx, &rank[(size_t)face], 4);       // traversal-side record carries the rank instead
    }
    std::vector<LightRec> lrec(lights.size());
    for (size_t i = 0; i < lights.size(); i++) {
        float v1v2[6];
        make_isect(tris[lights[i]], &lrec[i].v0x, &lrec[i].area, v1v2);
        lrec[i].v1x = v1v2[0]; lrec[i].v1y = v1v2[1]; lrec[i].v1z = v1v2[2];
        lrec[i].v2x = v1v2[3]; lrec[i].v2y = v1v2[4]; lrec[i].v2z = v1v2[5];
        lrec[i].pad = 0.0f;
    }

    free_scene(dev);
    pt_status st;
    if ((st = upload_vec(pairs, &dev->d_nodes)) != PT_OK) return st;
    if ((st = upload_vec(isect, &dev->d_tris)) != PT_OK) return st;
    if ((st = upload_vec(shade, &dev->d_shade)) != PT_OK) return st;
    {
        // onb::onb(normal), onb.h:8-13, per triangle: what shade() would build from the hit's normal on every scattering
        // bounce.  Only small scenes get the table (it is the dealt kernel's, staged in LDS).
        std::vector<float4> onb;
        if (n <= 1024) {
            onb.resize(3 * n);
            for (size_t k = 0; k < n; k++) {
                const v3 nn = mk3(shade[k].nx, shade[k].ny, shade[k].nz);
                const v3 a2 = normalize(nn);
                const v3 a = fabsf(a2.x) > 0.9f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
                const v3 a1 = normalize(cross(a2, a));
                const v3 a0 = cross(a2, a1);
                onb[3 * k] = make_float4(a0.x, a0.y, a0.z, a1.x);
                onb[3 * k + 1] = make_float4(a1.y, a1.z, a2.x, a2.y);
                onb[3 * k + 2] = make_float4(a2.z, 0.0f, 0.0f, 0.0f);
            }
        }
        if ((st = upload_vec(onb, &dev->d_onb)) != PT_OK) return st;
    }
    if ((st = upload_vec(lrec, &dev->d_lights)) != PT_OK) return st;
    if ((st = upload_vec(mrec, &dev->d_mats)) != PT_OK) return st;
    if ((st = upload_vec(tdesc, &dev->d_texs)) != PT_OK) return st;
    if ((st = upload_vec(pool, &dev->d_texels)) != PT_OK) return st;
    if (legacy_on) {
        std::vector<float4> lrec4(2 * m), srec4(2 * std::max<size_t>(dev->spheres_next.size(), 1));
        for (size_t i = 0; i < m; i++) {
            pt_legacy_material lm;
            memset(&lm, 0, sizeof lm);
            if (!dev->legacy_next.empty()) lm = dev->legacy_next[i];
            lrec4[2 * i] = make_float4(lm.albedo[0], lm.albedo[1], lm.albedo[2], lm.param);
            lrec4[2 * i + 1] = make_float4(__builtin_bit_cast(float, (int32_t)lm.kind), 0.0f, 0.0f, 0.0f);
        }
        for (size_t i = 0; i < dev->spheres_next.size(); i++) {
            const pt_sphere &sp = dev->spheres_next[i];
            srec4[2 * i] = make_float4(sp.center[0], sp.center[1], sp.center[2], sp.radius);
            srec4[2 * i + 1] = make_float4(__builtin_bit_cast(float, (int32_t)sp.materialIdx), 0.0f, 0.0f, 0.0f);
        }
        if ((st = upload_vec(lrec4, &dev->d_legacy)) != PT_OK) return st;
        if ((st = upload_vec(srec4, &dev->d_spheres)) != PT_OK) return st;
        dev->n_spheres = (int)dev->spheres_next.size();
    }
    dev->legacy_on = legacy_on;
    dev->any_tex = false;
    for (const MatRec &r : mrec) dev->any_tex = dev->any_tex || r.baseTex >= 0 || r.emisTex >= 0;
    dev->bvh = std::move(bvh);
    dev->light_faces = lights;
    dev->n_lights = (int)lights.size();
    dev->n_tris = n;
    dev->n_mats = m;
    dev->n_pairs = pairs.size();
    // Stage the whole scene in LDS when it is small (Cornell box: 10 pairs + 36 triangles =
    // 2.4 KB); the queue holds every leaf a ray can meet when the tree is small, else 8.
    const size_t scene_bytes = pairs.size() * sizeof(NodePair) + n * sizeof(TriIsect);
    dev->lds_scene = scene_bytes <= 24 * 1024;
    dev->packed_queue = n < 65536;
    dev->sah = sah;
    dev->sah_wide_on = sah && sah_wide_on;
    dev->sah_quant_on = sah && sah_wide_on && sah_quant_on;
    dev->sah_top_moved = sah && sah_wide_on ? sah_top_moved : 0u;
    dev->sah_stack = sah ? sahbvh.max_stack : 0;
    if (sah && getenv("PTAMD_VERBOSE"))
        fprintf(stderr, "ptamd: SAH tree %zu pairs, %s form %zu records, walk stack %u entries\n", sahbvh.pairs.size(),
                sah_quant_on ? "4-wide 64-byte" : sah_wide_on ? "4-wide" : "pair", sah_wide_on && !sah_quant_on ? pairs.size() / 2 : pairs.size(), sahbvh.max_stack);
    dev->walk = std::max(choose_walk(dev), 0);
    dev->queue_entries = (int)std::min<uint32_t>(std::max<uint32_t>(dev->bvh.leaves + 1, 4), 12);
    dev->has_scene = true;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_upload_host_scene(pt_device *dev, const pt_host_scene *scene)
try {
    if (!scene) { clear_error(); return set_error(PT_ERR_INVALID, "null scene"); }
    const HostSceneImpl &s = host_scene_impl(scene);
    return pt_upload_scene(dev, s.triangles.data(), s.triangles.size(), s.materials.data(),
                           s.materials.size(), s.textures.data(), s.textures.size());
} PT_CATCH_ALL

pt_status pt_upload_scene_strided(pt_device *dev, const void *triangles, size_t n, const pt_triangle_layout *layout,
                                  const pt_material *mats, size_t m, const pt_texture *texs, size_t k)
try {
    std::vector<pt_triangle> packed(n);
    pt_status st = pt_triangles_repack(triangles, n, layout, packed.data());
    if (st != PT_OK) return st;
    return pt_upload_scene(dev, packed.data(), n, mats, m, texs, k);
} PT_CATCH_ALL

pt_status pt_bind_framebuffer(pt_device *dev, int width, int height, uint8_t *rgb, uint8_t *yuv)
try {
    clear_error();
    if (!dev || width <= 0 || height <= 0 || !rgb || !yuv) return set_error(PT_ERR_INVALID, "bad framebuffer");
    if ((long long)width * (long long)height > (1ll << 30)) return set_error(PT_ERR_INVALID, "framebuffer too large");
    if ((width | height) & 1)
        return set_error(PT_ERR_UNSUPPORTED, "odd framebuffer sizes: the reference's YUV420 indexing (DevicePathTracer.h:107-119) writes past its planes; only even widths and heights are accepted");
    dev->W = width; dev->H = height; dev->rgb = rgb; dev->yuv = yuv;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_set_bvh_mode(pt_device *dev, int mode)
try {
    clear_error();
    if (!dev || mode < 0 || mode > 1) return set_error(PT_ERR_INVALID, "bvh mode must be 0 (reference topology) or 1 (SAH tree)");
    dev->bvh_mode = mode;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_set_legacy_scene(pt_device *dev, const pt_legacy_material *materials, size_t n_materials,
                              const pt_sphere *spheres, size_t n_spheres)
try {
    clear_error();
    if (!dev || (n_materials && !materials) || (n_spheres && !spheres)) return set_error(PT_ERR_INVALID, "null argument");
    for (size_t i = 0; i < n_materials; i++)
        if (materials[i].kind < 0 || materials[i].kind > 4) return set_error(PT_ERR_INVALID, "legacy material kind must be 0..4");
    for (size_t i = 0; i < n_spheres; i++)
        if (!(spheres[i].radius == spheres[i].radius)) return set_error(PT_ERR_INVALID, "sphere radius is NaN");
    dev->legacy_next.assign(materials, materials + n_materials);
    dev->spheres_next.assign(spheres, spheres + n_spheres);
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_set_cooperative_mode(pt_device *dev, int mode)
try {
    clear_error();
    if (!dev || mode < -1 || mode > 1) return set_error(PT_ERR_INVALID, "mode must be -1 (auto), 0 (off) or 1 (on)");
    dev->coop_mode = mode;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_set_tile_schedule(pt_device *dev, int mode)
try {
    clear_error();
    if (!dev || mode < 0 || mode > 1) return set_error(PT_ERR_INVALID, "mode must be 0 (raster order) or 1 (cost-ordered)");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(dev->mu);
    dev->tile_schedule = mode;
    drop_schedules(dev);
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_set_traversal_mode(pt_device *dev, int mode)
try {
    clear_error();
    if (!dev || mode < 0 || mode > 5) return set_error(PT_ERR_INVALID, "mode must be 0 (auto), 1 (per-lane), 2 (wave-uniform, mask), 3 (wave-uniform, stack), 4 (mask walk with dealt leaves) or 5 (stack walk with dealt leaves)");
    const int saved = dev->force_walk;
    dev->force_walk = mode >= 4 ? mode : mode - 1;  // -1 auto, 0 per-lane, 1 mask, 2 stack, 4 mask + dealt leaves, 5 stack + dealt leaves
    if (dev->has_scene) {
        int w = choose_walk(dev);
        if (w < 0) {
            dev->force_walk = saved;
            return set_error(PT_ERR_UNSUPPORTED, "this tree is too big for the requested wave-uniform walk (mask: <= 64 inner nodes; dealt mask: also <= 1024 triangles; dealt stack: <= 65536 triangles)");
        }
        dev->walk = w;
    }
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_set_thread_block_size(pt_device *dev, unsigned x, unsigned y)
try {
    clear_error();
    if (!dev || x == 0 || y == 0) return set_error(PT_ERR_INVALID, "bad thread block size");
    dev->block_x = x; dev->block_y = y;     // hint only: waves always cover 8x8 pixel tiles
    return PT_OK;
} PT_CATCH_ALL

// Bytes of dynamic LDS for a workgroup of `block` lanes (mirrors dev::lds_carve).
static size_t lds_bytes(const RenderParams &P, int block, bool stage)
{
    size_t f4 = 0;
    if (stage) f4 += (size_t)P.n_pairs * 4 + (size_t)P.n_tris * 3;
    f4 += ((size_t)P.stack_entries * block + 3) / 4;
    f4 += ((size_t)P.queue_entries * block * P.queue_dwords + 3) / 4;
    if (P.shade_lds) f4 += (P.shade_lds == 1 ? (size_t)P.n_tris * 3 : 0) + (size_t)P.n_mats * 3 + (size_t)P.n_lights * 5;
    f4 += (size_t)P.top_records * (size_t)(P.top_record_bytes / 16);
    if (P.deal_cap > 0 && P.stack_entries == 0)       // dev::deal_carve (kWalk 4: no stack)
        f4 += (size_t)P.n_tris * 3 /* onb table, stage_shade_lds */ + (size_t)P.n_tris * 3 +
              (size_t)(block / dev::kWave) * (2 * (size_t)dev::kWave + ((size_t)P.deal_cap * 2 + 15) / 16) +
              (P.part_boxes ? 6 * (size_t)P.n_pairs + (size_t)(block / dev::kWave) * 32 : 0);      // boxes, reach table, ray_c
    if (P.deal_cap > 0 && P.stack_entries != 0)       // dev::deal_carve32 (kWalk 5: stack walk, 32-bit items, no triangle table)
        f4 += (size_t)(block / dev::kWave) * (2 * (size_t)dev::kWave + ((size_t)P.deal_cap * 4 + 15) / 16 + 3 * (size_t)dev::kStageTris);
    return f4 * 16;
}

static pt_status fill_params(pt_device *dev, RenderParams &P)
{
    P.nodes = dev->d_nodes; P.tris = dev->d_tris; P.shade = dev->d_shade; P.onb = dev->d_onb; P.lights = dev->d_lights;
    P.mats = dev->d_mats; P.texs = dev->d_texs; P.texels = dev->d_texels;
    P.rgb = dev->rgb; P.yuv = dev->yuv; P.counters = dev->d_counters;
    P.n_lights = dev->n_lights;
    P.light_weight = 1.0f / (float)dev->n_lights;                        // hitable_list.h:17
    P.light_index_scale = (double)(dev->n_lights - 1) + 0.999999;        // hitable_list.h:24
    P.W = dev->W; P.H = dev->H;
    P.stack_entries = (int)std::max<uint32_t>(dev->bvh.max_stack, 2) - 1;
    P.n_pairs = (int)dev->n_pairs;
    P.n_tris = (int)dev->n_tris;
    P.queue_entries = dev->queue_entries;
    P.sah = dev->sah ? (dev->sah_quant_on ? 3 : dev->sah_wide_on ? 2 : 1) : 0;
    // How soon the SAH kernel's walk loop lets finished lanes shade and take their next segment.  Measured (tools/exp_pct.sh):
    // a tree that lives in the caches wants few, well-filled shading passes (C3: 40 % -> 128.8 ms; 50: 131.2, 65: 138.7);
    // a tree in HBM, whose every step waits on memory, wants its idle lanes refilled early (C4: 80 % -> 67.5 ms; 50: 72.2,
    // 35: 77.3, 95: 81.4).  The 4-wide form's visits are fatter and fewer, which moves the big-scene optimum down
    // (C4: 57.4 / 57.9 / 53.0 / 55.0 / 56.0 / 58.0 / 65.8 ms at 30 / 40 / 50 / 60 / 70 / 80 / 90 %; C3 stays at 40).
    // Round 4's tree (leaves of two, boxes 1e-5 of the magnitude wider instead of 1e-4 of the extent) moves it again: C4 35.0 /
    // 31.6 / 31.9 / 32.2 / 32.8 / 33.7 ms at 20 / 25 / 30 / 35 / 40 / 50 %, C3 95.8 / 95.9 / 96.4 / 98.0 / 99.3 at 30 / 35 / 40 / 45 / 50:
    // 30 for both 4-wide forms (the pair form keeps 80 / 40).
    {
        const size_t scene_bytes = dev->n_pairs * sizeof(NodePair) + dev->n_tris * sizeof(TriIsect);
        P.spec_pct = dev->spec_pct >= 0 ? dev->spec_pct : (dev->sah_wide_on ? 30 : scene_bytes > ((size_t)64 << 20) ? 80 : 40);
    }
    P.fold_bits = dev->fold_bits;
    P.deal_cap = 0; P.part_boxes = 0; P.deal_min = dev->deal_min; P.any_tex = dev->any_tex ? 1 : 0; P.lrpt = dev->lrpt;
    P.tile_order = nullptr; P.tile_cost = nullptr; P.band_list = nullptr;
    P.legacy_mats = dev->d_legacy; P.spheres = dev->d_spheres; P.n_spheres = dev->n_spheres;
    // SAH walks check for room before they push (SahStack): with the whole stack in LDS the check must never fire, hence + 3
    P.stack_spill = nullptr; P.spill_blocks = 0; P.stack_cap = 0x3fffffff;      // whole stack in LDS: the check never fires
    P.top_records = 0; P.top_record_bytes = 64;
    if (dev->sah) P.stack_entries = (int)dev->sah_stack;
    P.queue_dwords = (dev->packed_queue && !dev->legacy_on) ? 1 : 2;      // the extension kernel uses the unpacked queue
    P.n_mats = (int)dev->n_mats;
    // shading tables in LDS when they are small (Cornell box: 1.9 KB)
    // 1: TriShade, MatRec and LightRec tables all fit; 2: only the (small) material and light tables; 0: none
    P.shade_lds = (dev->n_tris * 48 + dev->n_mats * 48 + (size_t)dev->n_lights * 80) <= 16 * 1024 ? 1
                  : (dev->n_mats * 48 + (size_t)dev->n_lights * 80) <= 8 * 1024 ? 2 : 0;
    return PT_OK;
}

// Finds (or makes) the schedule of this task geometry; null = render in raster order (feature off,
// cache full of busy entries, or an allocation failed).  Caller holds dev->mu.
static TileSchedule *schedule_for(pt_device *dev, const RenderParams &P, unsigned slots, hipStream_t hs)
{
    // (the framebuffer is part of the key: a caller that keeps two frames in flight, each in its own framebuffer on its own
    // stream, gets a schedule per framebuffer -- launches that share one are ordered behind each other, see launch_render)
    const unsigned long long fb_bits = (unsigned long long)(uintptr_t)P.rgb;
    const int key[11] = {P.task_w, P.task_h, P.task_ox, P.task_oy, P.band_rows, P.band_ranks, P.band_rank, P.W, P.H,
                         (int)(uint32_t)fb_bits, (int)(uint32_t)(fb_bits >> 32)};
    dev->schedule_clock++;
    for (TileSchedule *t : dev->schedules)
        if (memcmp(t->key, key, sizeof key) == 0 && t->slots == slots) { t->stamp = dev->schedule_clock; return t; }
    if (dev->schedules.size() >= 16) {
        // evict the least recently used schedule whose last launch has finished
        size_t victim = dev->schedules.size();
        for (size_t i = 0; i < dev->schedules.size(); i++) {
            TileSchedule *t = dev->schedules[i];
            // busy: some launch is between picking this entry and recording last_use -- never evictable, whatever
            // the event says (an event that was never recorded, or whose previous record has completed, also
            // queries as hipSuccess)
            const bool idle = t->busy == 0 && (!t->used || hipEventQuery(t->last_use) == hipSuccess);
            if (idle && (victim == dev->schedules.size() || t->stamp < dev->schedules[victim]->stamp)) victim = i;
        }
        (void)hipGetLastError();
        if (victim == dev->schedules.size()) return nullptr;
        free_schedule(dev->schedules[victim], hs, true);
        dev->schedules.erase(dev->schedules.begin() + (long)victim);
    }
    std::unique_ptr<TileSchedule> t(new TileSchedule);
    memcpy(t->key, key, sizeof key);
    t->slots = slots;
    t->stamp = dev->schedule_clock;
    t->id = ++dev->schedule_ids;
    // One stream-ordered allocation (hipMallocAsync: no device-wide synchronisation, so scheduling
    // policies that move their rectangles every frame, DSFL / DSDL, can afford a fresh schedule per launch)
    const size_t bytes = (((size_t)slots * sizeof(uint32_t)) + 255) & ~(size_t)255;
    bool ok = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, t->temp_bytes, t->cost, t->sorted_cost, t->iota, t->order,
                                                           (int)slots, 0, 32, hs) == hipSuccess;
    if (ok) ok = hipMallocAsync(&t->base, 4 * bytes + 256 + std::max<size_t>(t->temp_bytes, 256), hs) == hipSuccess;
    if (ok) {
        uint8_t *b = (uint8_t *)t->base;
        t->order = (uint32_t *)b; t->cost = (uint32_t *)(b + bytes); t->sorted_cost = (uint32_t *)(b + 2 * bytes);
        t->iota = (uint32_t *)(b + 3 * bytes); t->n_split = (uint32_t *)(b + 4 * bytes); t->temp = b + 4 * bytes + 256;
        ok = hipEventCreateWithFlags(&t->last_use, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { (void)hipGetLastError(); free_schedule(t.release(), hs, true); return nullptr; }
    hipLaunchKernelGGL(dev::iota_kernel, dim3((slots + 255) / 256), dim3(256), 0, hs, t->iota, t->order, slots);
    (void)hipMemsetAsync(t->cost, 0, bytes, hs);
    (void)hipMemsetAsync(t->n_split, 0, 256, hs);
    dev->schedules.push_back(t.get());
    return t.release();
}

// The kernel a launch of the uploaded scene runs, and whether it is the cooperative (4 waves per tile) form.
typedef void (*RenderKernel)(RenderParams);
struct KernelChoice { RenderKernel fn; const char *name; };
#define PT_KERNEL(...) KernelChoice{__VA_ARGS__, "pt::" #__VA_ARGS__}
static KernelChoice pick_kernel(const pt_device *dev, int walk, bool coop, bool stage, bool packed, bool lean = false, bool dense = false)
{
    if (lean) return PT_KERNEL(dev::render_kernel_sahw_lean);
    if (dense && walk == 4 && dev->deal_min <= 64) return PT_KERNEL(dev::render_kernel_dealt8);
    if (dense && walk == 5 && !coop) return PT_KERNEL(dev::render_kernel_dealt_stack6);
    if (walk == 3 && dev->sah_persistent) {
        if (dev->sah_quant_on) return dev->sah_by_hand ? PT_KERNEL(dev::render_kernel_sahq_any) : PT_KERNEL(dev::render_kernel_sahq);
        if (dev->sah_wide_on) return dev->sah_by_hand ? PT_KERNEL(dev::render_kernel_sahw_any) : dense ? PT_KERNEL(dev::render_kernel_sahw6) : PT_KERNEL(dev::render_kernel_sahw);
        return PT_KERNEL(dev::render_kernel_sah);
    }
    if (walk == 3) return PT_KERNEL(dev::render_kernel<false, true, 3>);
    if (walk == 1) return PT_KERNEL(dev::render_kernel<false, true, 1>);
    if (walk == 4) return dev->deal_min > 64 ? PT_KERNEL(dev::render_kernel_dealt<false>) : PT_KERNEL(dev::render_kernel_dealt<true>);
    if (walk == 5 && !coop) return PT_KERNEL(dev::render_kernel_dealt_stack);
    if (coop) return PT_KERNEL(dev::render_kernel_coop);
    if (walk == 2 || walk == 5) return PT_KERNEL(dev::render_kernel<false, true, 2>);
    if (dev->legacy_on) return PT_KERNEL(dev::render_kernel<false, false, 0, true>);
    if (stage) return packed ? PT_KERNEL(dev::render_kernel<true, true, 0>) : PT_KERNEL(dev::render_kernel<true, false, 0>);
    return packed ? PT_KERNEL(dev::render_kernel<false, true, 0>) : PT_KERNEL(dev::render_kernel<false, false, 0>);
}
#undef PT_KERNEL

// Dynamic LDS of a launch against what the device gives one workgroup.  Above the 64 KB every HIP kernel may
// use by default the limit has to be raised per kernel; above the hardware's (160 KB on gfx950) the launch
// would fail with an opaque HIP error, so it is refused here with the reason (a per-lane stack of `max_stack`
// entries for 256 lanes plus the per-lane leaf queue: reference trees deeper than ~40 levels in the per-lane walk).
static pt_status ensure_lds(pt_device *dev, const void *kern, size_t lds)
{
    if (lds <= (size_t)64 * 1024) return PT_OK;
    if (lds > dev->lds_limit)
        return set_error(PT_ERR_UNSUPPORTED, "this tree needs " + std::to_string(lds) + " B of LDS per workgroup (per-lane traversal stacks of " +
                                                 std::to_string(dev->bvh.max_stack) + " entries), the device offers " + std::to_string(dev->lds_limit));
    std::lock_guard<std::mutex> lk(dev->mu);
    for (auto &kv : dev->lds_raised) if (kv.first == kern && kv.second >= lds) return PT_OK;
    PT_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dev->lds_raised.push_back({kern, lds});
    return PT_OK;
}

// Workgroups of `kern` one CU holds at once with `lds` bytes of dynamic LDS (the grid of a persistent launch).
static int resident_per_cu(pt_device *dev, const void *kern, size_t lds)
{
    std::lock_guard<std::mutex> lk(dev->mu);
    for (auto &r : dev->resident_blocks) if (r.kern == kern && r.lds == lds) return r.blocks;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, dev::kBlock, lds) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 8; }
    dev->resident_blocks.push_back({kern, lds, n});
    return n;
}

static pt_status launch_render(pt_device *dev, RenderParams &P, pt_stream stream)
{
    long long tiles = (long long)P.tiles_x * P.tiles_y;
    unsigned blocks = (unsigned)((tiles + dev::kWavesPerBlock - 1) / dev::kWavesPerBlock);
    const int walk = dev->walk;
    const bool stage = dev->lds_scene && walk == 0 && !dev->legacy_on, packed = dev->packed_queue && !dev->legacy_on;
    if (walk != 0) P.queue_entries = 0;
    if (walk == 1 || walk == 4) P.stack_entries = 0;
    P.deal_cap = (walk == 4 || walk == 5) ? dev->deal_cap : 0;
    // Launches with more than twice as many tiles as the device has wave slots are bound by throughput and take the builds
    // with one more wave per SIMD (dealt mask walk: eight; dealt stack walk and the SAH kernel of cache-resident trees: six);
    // smaller ones, whose time is some wave's chain, keep the spill-free builds.
    const bool dense = (walk == 4 || walk == 5 || (walk == 3 && dev->sah_persistent && dev->sah_wide_on && !dev->sah_quant_on)) && dev->deal_dense &&
                       tiles >= (walk == 4 ? 2ll * dev->max_waves : 2ll * dev->cu_count * 6 * dev::kWavesPerBlock);
    if (dense && walk == 5 && !dev->deal_cap_by_hand) P.deal_cap = 384;      // six workgroups' item lists and stacks in a CU's LDS
    // Parts of split tiles walk by lanes (part_walk_by_lanes, pt_walk.h) where the tree allows it: all 2 n_pairs hit bits in
    // one word, every triangle for every ray of a part within the item list.  Only launches that can have parts pay the 1.7 KB of LDS (dense
    // ones never split, and the eight-wave build has no room for it).
    // (an unsplit root is its own two children, bvh.h:26: every ray then lists the root's triangles twice)
    const unsigned long long lists_per_tri = (!dev->bvh.nodes.empty() && dev->bvh.nodes[0].triCount != 0) ? 2ull : 1ull;
    P.part_boxes = (walk == 4 && !dense && dev->part_lanes && dev->deal_split >= 2 && dev->deal_split <= 16 && dev->tile_schedule == 1 &&
                    dev->n_pairs >= 1 && dev->n_pairs <= 16 &&
                    (64ull / (unsigned)dev->deal_split) * lists_per_tri * dev->n_tris <= (unsigned long long)P.deal_cap) ? 1 : 0;
    // big leaves + stack walk: four waves per tile, each testing a quarter of every leaf (render_kernel_coop)
    // It pays when the launch is small for the machine (a rank's share of a multi-GPU frame): measured on
    // the duck, 1/6 of a 1080p frame and less gains (1/8: 1.85 -> 1.09 s), 1/4 and more loses (whole frame
    // 5.3 -> 7.1 s) -- the redundant box walks and the two barriers per trip cost throughput.
    // (With the dealt stack walk, kWalk 5, persistent waves and split tiles do the same job better -- an eighth of the duck
    // frame 1.09 s against 1.18 s -- so there the cooperative kernel runs only when it is asked for.)
    const bool coop = (walk == 2 || walk == 5) && !dev->legacy_on &&
                      (dev->coop_mode == 1 || (walk == 2 && dev->coop_mode < 0 && dev->bvh.max_leaf >= 64 &&
                                               tiles * dev::kWavesPerBlock <= 3ll * dev->max_waves));
    if (coop) { blocks = (unsigned)tiles; P.queue_entries = 10; P.queue_dwords = 1; P.deal_cap = 0; }   // LDS per thread: 2 x (float4 partial hit + visit rank)
    // The library's own choice.  Launches that fill the device are bound by throughput: short LDS stacks (a fifth workgroup
    // per CU), and for cache-resident trees (128-byte nodes) the top levels in LDS.  A launch with fewer tiles than the device
    // has wave slots (a rank's share of a multi-GPU frame) lasts as long as its dearest pixel's chain of dependent micro-steps,
    // and there every instruction in the step counts: an eighth of the duck's frame takes 43.9 ms with the bare walk against
    // 46.5 - 47.4 ms with the options in (while the whole frame takes 123 against 110 ms): such launches get the lean kernel.
    // By hand (PTAMD_SAH_TOP / _STACK): both options as asked, always.
    bool lean = false;
    if (walk == 3 && dev->sah_persistent && dev->sah_wide_on && !dev->sah_quant_on && !dev->sah_by_hand) {
        const size_t lds_lean = lds_bytes(P, dev::kBlock, stage);            // P still has whole stacks and no top levels here
        const long long slots = (long long)dev->cu_count * resident_per_cu(dev, (const void *)dev::render_kernel_sahw_lean, lds_lean) * dev::kWavesPerBlock;
        lean = tiles * 100ll <= slots * (long long)dev->lean_pct;
    }
    const int stack_lds = dev->sah_stack_lds;
    const bool short_stack = !lean;
    const bool top_levels = !lean && (dev->sah_by_hand || !dev->sah_quant_on);
    if (walk == 3 && dev->sah_persistent && short_stack && stack_lds > 0 && (int)dev->sah_stack > stack_lds) {
        // The persistent SAH kernel keeps only the top of every lane's stack in LDS (SahStack, pt_walk.h): the worst-case
        // bound of a deep tree (C4: 37 entries = 38 KB per workgroup) would otherwise cap its occupancy.
        P.stack_entries = P.stack_cap = stack_lds;
        P.spill_blocks = ((int)dev->sah_stack + 3 - stack_lds + dev::kSpillBlock - 1) / dev::kSpillBlock + 1;
    }
    if (walk == 3 && dev->sah_persistent && dev->sah_wide_on && top_levels && dev->sah_top_moved > 0) {
        // The tree's top levels in LDS (north_star's wording): every ray visits the root and its children, and a visit served
        // from LDS returns in ~64 cycles instead of an L1 / L2 round trip -- two of the duck's 5.2 visits per segment.
        // C3 123.7 -> 110.2 ms with 8 .. 96 records (profiles/r03_topk_lds.json); C4 (14.5 visits per segment, bound by the
        // request path rather than by latency) 52.8 -> 52.4 ms; 256 records cost a resident workgroup and lose.
        P.top_records = (int)dev->sah_top_moved;
        P.top_record_bytes = dev->sah_quant_on ? 64 : 128;
    }
    const size_t lds = lds_bytes(P, dev::kBlock, stage);
    const KernelChoice choice = pick_kernel(dev, walk, coop, stage, packed, lean, dense);
    const RenderKernel kern = choice.fn;
    {
        const pt_status st = ensure_lds(dev, (const void *)kern, lds);
        if (st != PT_OK) return st;
    }
    const bool persistent = walk == 4 || (walk == 5 && !coop) || (walk == 3 && dev->sah_persistent);
    hipStream_t hs = (hipStream_t)stream;
    pt_device::LaunchEvents ev;
    TileSchedule *sched = nullptr;
    P.tile_order = nullptr; P.tile_cost = nullptr; P.n_split = nullptr; P.split_parts = 16;
    const int split_parts = (walk == 4 || (walk == 5 && !coop)) ? dev->deal_split : (walk == 3 && dev->sah_persistent) ? dev->sah_split : 0;
    const bool split = split_parts > 0;
    // An error between here and the bookkeeping at the end gives the event pair back and releases the schedule.
    auto fail = [&](pt_status st) {
        std::lock_guard<std::mutex> lk(dev->mu);
        if (ev.first && ev.second) dev->free_events.push_back(ev);
        else { if (ev.first) (void)hipEventDestroy(ev.first); if (ev.second) (void)hipEventDestroy(ev.second); }
        if (sched) sched->busy--;
        return st;
    };
#define PT_HIP_LAUNCH(call)                                                                                \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(set_error(PT_ERR_HIP, std::string("HIP ERROR = ") + std::to_string((int)e_) + " (" + \
                                                  hipGetErrorString(e_) + ") at " + __FILE__ + ":" +       \
                                                  std::to_string(__LINE__) + " '" #call "'"));             \
    } while (0)
    // bytes of overflow area this launch will want (persistent SAH kernel with a short LDS stack): known before the launch's
    // first event is recorded, so that a (blocking) allocation never sits inside the timed span nor between queued launches
    size_t spill_need = 0;
    if (persistent && P.spill_blocks > 0) {
        int per_cu = resident_per_cu(dev, (const void *)kern, lds);
        if (dev->resident_cap > 0) per_cu = std::min(per_cu, dev->resident_cap);
        const unsigned resident = std::min(blocks, (unsigned)(dev->cu_count * per_cu));
        spill_need = (size_t)resident * dev::kWavesPerBlock * (size_t)P.spill_blocks * 64 * dev::kSpillBlock * sizeof(uint32_t);
    }
    {
        std::lock_guard<std::mutex> lk(dev->mu);
        // Finished launches give their events, tile counter and overflow area back at EVERY launch (oldest first; launches
        // finish roughly in the order they were enqueued): the pool stays as small as the number of launches in flight,
        // whether or not anybody reads the timings (a streaming host never does), and with it the number of overflow areas
        // (C4: 42 MB each).
        {
            size_t done = 0;
            while (done < dev->pending.size() && hipEventQuery(dev->pending[done].second) == hipSuccess) {
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, dev->pending[done].first, dev->pending[done].second) == hipSuccess) { dev->ms_total += (double)ms; dev->launches++; }
                dev->free_events.push_back(dev->pending[done]);
                done++;
            }
            if (done) dev->pending.erase(dev->pending.begin(), dev->pending.begin() + (long)done);
            (void)hipGetLastError();      // hipEventQuery reports "not ready" through the error state
        }
        if (!dev->free_events.empty()) {
            // prefer an entry whose overflow area is already big enough
            size_t pick = dev->free_events.size() - 1;
            for (size_t i = 0; i < dev->free_events.size(); i++) if (dev->free_events[i].spill_bytes >= spill_need && spill_need > 0) { pick = i; break; }
            ev = dev->free_events[pick];
            dev->free_events.erase(dev->free_events.begin() + (long)pick);
        }
        else {
            hipError_t e1 = hipEventCreate(&ev.first), e2 = e1 == hipSuccess ? hipEventCreate(&ev.second) : e1;
            if (e2 == hipSuccess && dev->next_used % 256 == 0) {
                uint32_t *chunk = nullptr;
                e2 = hipMalloc((void **)&chunk, 256 * 64);
                if (e2 == hipSuccess) dev->next_chunks.push_back(chunk);
            }
            if (e2 != hipSuccess) {
                if (ev.first) (void)hipEventDestroy(ev.first);
                if (ev.second) (void)hipEventDestroy(ev.second);
                return set_error(PT_ERR_HIP, std::string("HIP ERROR = ") + std::to_string((int)e2) + " (" + hipGetErrorString(e2) + ") creating a timing event");
            }
            ev.next = dev->next_chunks.back() + (dev->next_used % 256) * 16;
            dev->next_used++;
        }
        // one wave slot per tile; tiny launches are not worth a sort
        if (dev->tile_schedule == 1 && blocks >= (unsigned)dev->sched_min_blocks) sched = schedule_for(dev, P, coop ? blocks : blocks * dev::kWavesPerBlock, hs);
        if (sched) {
            // Pinned while this launch uses it outside the lock (its sort and its last_use record below): eviction
            // in schedule_for() skips busy entries, so another thread's launch can never free it under our feet.
            sched->busy++;
            // a schedule is read and rewritten by its launches: order them across streams
            if (sched->used && sched->last_stream != hs) {
                hipError_t e = hipStreamWaitEvent(hs, sched->last_use, 0);
                if (e != hipSuccess) {
                    sched->busy--; sched = nullptr;
                    dev->free_events.push_back(ev);
                    return set_error(PT_ERR_HIP, std::string("HIP ERROR = ") + std::to_string((int)e) + " (" + hipGetErrorString(e) + ") in hipStreamWaitEvent");
                }
            }
            sched->used = true; sched->last_stream = hs;
            P.tile_order = sched->order; P.tile_cost = sched->cost;
            if (split) { P.n_split = sched->n_split; P.split_parts = split_parts; }
        }
    }
    if (spill_need > 0 && ev.spill_bytes < spill_need) {
        // (first use of this entry, or a bigger launch than it has seen: the entry is ours alone, no launch is using its area)
        (void)hipFree(ev.spill); ev.spill = nullptr; ev.spill_bytes = 0;
        PT_HIP_LAUNCH(hipMalloc((void **)&ev.spill, spill_need));
        ev.spill_bytes = spill_need;
    }
    if (split && sched) {
        hipLaunchKernelGGL(dev::split_reset_kernel, dim3(1), dim3(256), 0, hs, sched->order, sched->cost, sched->n_split);
        PT_HIP_LAUNCH(hipGetLastError());
    }
    PT_HIP_LAUNCH(hipEventRecord(ev.first, hs));
    if (persistent) {
        // as many workgroups as the device holds at once; their waves take tiles from ev.next until it passes n_slots
        P.tile_next = ev.next; P.n_slots = blocks * dev::kWavesPerBlock;
        PT_HIP_LAUNCH(hipMemsetAsync(P.tile_next, 0, sizeof(uint32_t), hs));
        int per_cu = resident_per_cu(dev, (const void *)kern, lds);
        if (dev->resident_cap > 0) per_cu = std::min(per_cu, dev->resident_cap);
        blocks = std::min(blocks, (unsigned)(dev->cu_count * per_cu));
        if (P.spill_blocks > 0) P.stack_spill = ev.spill;      // the launch's own overflow area (allocated above): one run of blocks per resident wave
    }
    // Frames in flight: launches of this device that are still running when this one is enqueued share its wave slots with
    // it (the pending launches' end events, newest first; launches finish roughly in the order they were enqueued).
    unsigned sharing = 1;
    {
        std::lock_guard<std::mutex> lk(dev->mu);
        for (auto it = dev->pending.rbegin(); it != dev->pending.rend() && sharing < 8u; ++it) {
            if (hipEventQuery(it->second) == hipSuccess) break;
            sharing++;
        }
        (void)hipGetLastError();      // hipEventQuery reports "not ready" through the error state
    }
    // Issue priority by remaining samples keeps a lone launch's last waves from starving; when another launch's waves fill
    // the slots anyway it only costs (C2, two frames in flight: 55.7 ms per frame with it, 55.0 without).
    if (sharing > 1) P.lrpt = 0;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(dev::kBlock), lds, hs, P);
    PT_HIP_LAUNCH(hipGetLastError());
    PT_HIP_LAUNCH(hipEventRecord(ev.second, hs));
    if (sched) {
        // next launch's order: tiles by this launch's cost, dearest first (stable, so ties keep raster order)
        PT_HIP_LAUNCH(hipcub::DeviceRadixSort::SortPairsDescending(sched->temp, sched->temp_bytes, sched->cost, sched->sorted_cost,
                                                                   sched->iota, sched->order, (int)sched->slots, 0, 32, hs));
        if (split) {
            const unsigned resident = (unsigned)(dev->cu_count * resident_per_cu(dev, (const void *)kern, lds)) * dev::kWavesPerBlock;
            // extra waves the split may add: the wave slots this launch leaves empty, or an eighth of all of them
            const unsigned spare = std::max(resident / 8u, resident > sched->slots ? resident - sched->slots : 0u);
            // Launches that share the device (`sharing`, above): splitting buys a shorter chain with 2.6 x a tile's work, which
            // only pays while slots would idle otherwise: the "evenly spread launch" the dearest tile is compared with spreads
            // over this launch's SHARE of the slots (a quarter of C2, two frames in flight: 15.9 ms per frame split, 14.9
            // unsplit; profiles/r03_frames_in_flight.json).
            hipLaunchKernelGGL(dev::split_count_kernel, dim3(1), dim3(256), 0, hs, sched->sorted_cost, sched->slots, std::max(1u, resident / sharing), std::max(1u, spare / (unsigned)(split_parts - 1)), (unsigned)(dev->split_pct > 0 ? dev->split_pct : walk == 4 ? 110 : 90),
                               (unsigned)(dev->split_ratio > 0 ? dev->split_ratio : walk == 4 ? 65 : 80), sched->n_split);
            PT_HIP_LAUNCH(hipGetLastError());
        }
        PT_HIP_LAUNCH(hipEventRecord(sched->last_use, hs));
    }
#undef PT_HIP_LAUNCH
    {
        std::lock_guard<std::mutex> lk(dev->mu);
        if (sched) sched->busy--;                // last_use is recorded: from here on hipEventQuery(last_use) tells when it is idle
        dev->last_kernel = choice.name;
        dev->last_sched_id = sched ? sched->id : 0; dev->last_sched_stream = hs;
        if (P.band_rows > 0) {
            pt_device::BandLaunch bl{hs, sched ? sched->id : 0, P.tiles_x, P.tiles_y, P.band_rows, P.task_h, ++dev->band_clock};
            bool found = false;
            for (auto &e : dev->band_launches) if (e.stream == hs) { e = bl; found = true; }
            if (!found) {
                if (dev->band_launches.size() >= 32) {         // (streams come and go: forget the one used longest ago)
                    size_t old = 0;
                    for (size_t i = 1; i < dev->band_launches.size(); i++) if (dev->band_launches[i].stamp < dev->band_launches[old].stamp) old = i;
                    dev->band_launches.erase(dev->band_launches.begin() + (long)old);
                }
                dev->band_launches.push_back(bl);
            }
        }
        dev->pending.push_back(ev);
    }
    return PT_OK;
}

pt_status pt_render_task_async(pt_device *dev, const pt_task *task, const pt_camera *camera,
                               unsigned spp, unsigned depth, pt_stream stream)
try {
    clear_error();
    if (!dev || !task || !camera) return set_error(PT_ERR_INVALID, "null argument");
    if (task->width == 0) return PT_OK;                                  // DevicePathTracer.h:195-196
    if (!dev->has_scene || !dev->rgb) return set_error(PT_ERR_NO_SCENE, "render before pt_upload_scene / pt_bind_framebuffer");
    if (task->width < 0 || task->height <= 0 || spp > (1u << 24))
        return set_error(PT_ERR_INVALID, "bad task rectangle or sample count");
    // host-side shape check before launching a hand-written kernel
    if (task->offset_x < 0 || task->offset_y < 0 || task->offset_x + task->width > dev->W ||
        task->offset_y + task->height > dev->H)
        return set_error(PT_ERR_INVALID, "task rectangle leaves the framebuffer");
    PT_HIP(hipSetDevice(dev->index));                                    // :200
    RenderParams P;
    fill_params(dev, P);
    P.cam = camera_basis(*camera);
    P.task_w = task->width; P.task_h = task->height; P.task_ox = task->offset_x; P.task_oy = task->offset_y;
    P.tiles_x = (task->width + 7) / 8; P.tiles_y = (task->height + 7) / 8;
    P.band_rows = 0; P.band_ranks = 0; P.band_rank = 0;
    P.spp = (int)spp; P.depth = depth;
    return launch_render(dev, P, stream);
} PT_CATCH_ALL

pt_status pt_render_bands_async(pt_device *dev, int band_rows, int n_ranks, int rank, const pt_camera *camera,
                                unsigned spp, unsigned depth, pt_stream stream)
try {
    clear_error();
    if (!dev || !camera) return set_error(PT_ERR_INVALID, "null argument");
    if (!dev->has_scene || !dev->rgb) return set_error(PT_ERR_NO_SCENE, "render before pt_upload_scene / pt_bind_framebuffer");
    if (band_rows <= 0 || (band_rows & 1) || n_ranks <= 0 || rank < 0 || rank >= n_ranks || spp > (1u << 24))
        return set_error(PT_ERR_INVALID, "bad row-band request (band_rows must be a positive even number)");
    PT_HIP(hipSetDevice(dev->index));
    // local rows = rows of all bands b = rank, rank + n_ranks, ... that start inside the frame
    const int n_bands = (dev->H + band_rows - 1) / band_rows;
    int local_rows = 0;
    for (int b = rank; b < n_bands; b += n_ranks) local_rows += std::min(band_rows, dev->H - b * band_rows);
    if (local_rows == 0) return PT_OK;
    RenderParams P;
    fill_params(dev, P);
    P.cam = camera_basis(*camera);
    // a partial last band only ever is the last local band, so rows stay contiguous in j
    P.task_w = dev->W; P.task_h = local_rows; P.task_ox = 0; P.task_oy = 0;
    P.tiles_x = (dev->W + 7) / 8; P.tiles_y = (local_rows + 7) / 8;
    P.band_rows = band_rows; P.band_ranks = n_ranks; P.band_rank = rank;
    P.spp = (int)spp; P.depth = depth;
    return launch_render(dev, P, stream);
} PT_CATCH_ALL

pt_status pt_render_band_list_async(pt_device *dev, int band_rows, const int *bands, int n_bands, const pt_camera *camera,
                                    unsigned spp, unsigned depth, pt_stream stream)
try {
    clear_error();
    if (!dev || !camera || (n_bands > 0 && !bands)) return set_error(PT_ERR_INVALID, "null argument");
    if (!dev->has_scene || !dev->rgb) return set_error(PT_ERR_NO_SCENE, "render before pt_upload_scene / pt_bind_framebuffer");
    if (band_rows <= 0 || (band_rows & 1) || n_bands < 0 || spp > (1u << 24))
        return set_error(PT_ERR_INVALID, "bad row-band request (band_rows must be a positive even number)");
    if (n_bands == 0) return PT_OK;
    // host-side shape check before launching a hand-written kernel: bands inside the frame, no band twice, and a
    // partial last band of the frame only as the LAST entry (rows stay contiguous in the launch's local row index)
    const int frame_bands = (dev->H + band_rows - 1) / band_rows;
    std::vector<char> seen((size_t)frame_bands, 0);
    int local_rows = 0;
    uint32_t hash = 2166136261u;
    for (int i = 0; i < n_bands; i++) {
        const int b = bands[i];
        if (b < 0 || b >= frame_bands || seen[(size_t)b]) return set_error(PT_ERR_INVALID, "band list: index out of range or repeated");
        seen[(size_t)b] = 1;
        const int rows = std::min(band_rows, dev->H - b * band_rows);
        if (rows < band_rows && i != n_bands - 1) return set_error(PT_ERR_INVALID, "band list: the frame's partial last band must be the last entry");
        local_rows += rows;
        hash = (hash ^ (uint32_t)b) * 16777619u;
    }
    PT_HIP(hipSetDevice(dev->index));
    hipStream_t hs = (hipStream_t)stream;
    int *d_list = nullptr;
    PT_HIP(hipMallocAsync((void **)&d_list, sizeof(int) * (size_t)n_bands, hs));
    hipError_t e = hipMemcpyAsync(d_list, bands, sizeof(int) * (size_t)n_bands, hipMemcpyHostToDevice, hs);
    if (e == hipSuccess) e = hipStreamSynchronize(hs) == hipSuccess ? hipSuccess : hipErrorUnknown;    // `bands` is the caller's: done with it on return
    if (e != hipSuccess) { (void)hipFreeAsync(d_list, hs); return set_error(PT_ERR_HIP, "copying the band list to the device failed"); }
    RenderParams P;
    fill_params(dev, P);
    P.cam = camera_basis(*camera);
    P.task_w = dev->W; P.task_h = local_rows; P.task_ox = 0; P.task_oy = 0;
    P.tiles_x = (dev->W + 7) / 8; P.tiles_y = (local_rows + 7) / 8;
    // band_ranks / band_rank only identify the tile schedule of this band set here (the kernel reads band_list)
    P.band_rows = band_rows; P.band_ranks = -n_bands; P.band_rank = (int)hash;
    P.band_list = d_list;
    P.spp = (int)spp; P.depth = depth;
    const pt_status st = launch_render(dev, P, stream);
    (void)hipFreeAsync(d_list, hs);                       // stream-ordered: after the launch that reads it
    return st;
} PT_CATCH_ALL

// any_stream: the newest band launch of the device; otherwise the newest one enqueued on `stream`.
static pt_status band_costs_read(pt_device *dev, bool any_stream, pt_stream stream, uint64_t *costs, int max_bands, int *n_bands)
{
    clear_error();
    if (!dev || !costs || !n_bands || max_bands < 0) return set_error(PT_ERR_INVALID, "null argument");
    *n_bands = 0;
    PT_HIP(hipSetDevice(dev->index));
    TileSchedule *sched = nullptr;
    int tiles_x = 0, tiles_y = 0, band_rows = 0, local_rows = 0;
    hipStream_t hs = nullptr;
    {
        std::lock_guard<std::mutex> lk(dev->mu);
        const pt_device::BandLaunch *bl = nullptr;
        for (const auto &e : dev->band_launches)
            if (any_stream ? (!bl || e.stamp > bl->stamp) : e.stream == (hipStream_t)stream) bl = &e;
        if (!bl) return PT_OK;                                 // no band launch (on that stream) yet
        tiles_x = bl->tiles_x; tiles_y = bl->tiles_y; band_rows = bl->band_rows; local_rows = bl->local_rows;
        hs = bl->stream;
        for (TileSchedule *t : dev->schedules) if (bl->sched_id && t->id == bl->sched_id) sched = t;      // still cached? (by id: an evicted schedule's address can be reused)
        if (sched) sched->busy++;
    }
    if (!sched || tiles_y <= 0) return PT_OK;              // nothing recorded (tile schedule off, tiny launch, or evicted)
    if ((size_t)tiles_x * (size_t)tiles_y > (size_t)sched->slots) {      // never index a cost buffer beyond its slots
        std::lock_guard<std::mutex> lk(dev->mu);
        sched->busy--;
        return set_error(PT_ERR_INTERNAL, "pt_band_costs_read: the cached schedule does not match the band launch");
    }
    struct Unpin { pt_device *d; TileSchedule *t; ~Unpin() { std::lock_guard<std::mutex> lk(d->mu); t->busy--; } } unpin{dev, sched};
    const int nb = (local_rows + band_rows - 1) / band_rows;
    if (nb > max_bands) return set_error(PT_ERR_INVALID, "pt_band_costs_read: output array too small");
    DevBufs bufs;
    unsigned long long *d_rows = nullptr;
    PT_HIP(bufs.alloc((void **)&d_rows, sizeof(unsigned long long) * (size_t)tiles_y));
    hipLaunchKernelGGL(dev::tile_row_cost_kernel, dim3((unsigned)tiles_y), dim3(256), 0, hs, sched->cost, tiles_x, d_rows);
    PT_HIP(hipGetLastError());
    std::vector<unsigned long long> rows((size_t)tiles_y);
    PT_HIP(hipMemcpyAsync(rows.data(), d_rows, sizeof(unsigned long long) * (size_t)tiles_y, hipMemcpyDeviceToHost, hs));
    PT_HIP(hipStreamSynchronize(hs));
    // tile row i covers local rows [8i, 8i + 8): its cost goes to the bands those rows belong to, by row count
    for (int k = 0; k < nb; k++) costs[k] = 0;
    for (int i = 0; i < tiles_y; i++) {
        const int r0 = 8 * i, r1 = std::min(8 * i + 8, local_rows);
        for (int r = r0; r < r1;) {
            const int k = r / band_rows, stop = std::min(r1, (k + 1) * band_rows);
            costs[k] += rows[(size_t)i] * (unsigned long long)(stop - r) / (unsigned long long)(r1 - r0);
            r = stop;
        }
    }
    *n_bands = nb;
    return PT_OK;
}

pt_status pt_band_costs_read(pt_device *dev, uint64_t *costs, int max_bands, int *n_bands)
try {
    return band_costs_read(dev, true, nullptr, costs, max_bands, n_bands);
} PT_CATCH_ALL

pt_status pt_band_costs_read_on(pt_device *dev, pt_stream stream, uint64_t *costs, int max_bands, int *n_bands)
try {
    return band_costs_read(dev, false, stream, costs, max_bands, n_bands);
} PT_CATCH_ALL

pt_status pt_stream_sync(pt_device *dev, pt_stream stream)
try {
    clear_error();
    if (!dev) return set_error(PT_ERR_INVALID, "null device");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipGetLastError());
    PT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_device_sync(pt_device *dev)
try {
    clear_error();
    if (!dev) return set_error(PT_ERR_INVALID, "null device");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipGetLastError());
    PT_HIP(hipDeviceSynchronize());
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_stream_create(pt_device *dev, pt_stream *out)
try {
    clear_error();
    if (!dev || !out) return set_error(PT_ERR_INVALID, "null argument");
    PT_HIP(hipSetDevice(dev->index));
    hipStream_t s;
    PT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (pt_stream)s;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_stream_destroy(pt_device *dev, pt_stream stream)
try {
    clear_error();
    if (!dev) return set_error(PT_ERR_INVALID, "null device");
    PT_HIP(hipSetDevice(dev->index));
    if (stream) PT_HIP(hipStreamDestroy((hipStream_t)stream));
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_counters_read(pt_device *dev, pt_counters *out)
try {
    clear_error();
    if (!dev || !out) return set_error(PT_ERR_INVALID, "null argument");
    PT_HIP(hipSetDevice(dev->index));
    unsigned long long h[8];
    PT_HIP(hipMemcpy(h, dev->d_counters, sizeof h, hipMemcpyDeviceToHost));
    out->paths = h[0]; out->segments = h[1]; out->box_tests = h[2]; out->tri_tests = h[3]; out->light_tests = h[4];
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_split_tiles_read(pt_device *dev, uint32_t *n_split)
try {
    clear_error();
    if (!dev || !n_split) return set_error(PT_ERR_INVALID, "null argument");
    *n_split = 0;
    PT_HIP(hipSetDevice(dev->index));
    TileSchedule *sched = nullptr;
    hipStream_t hs;
    {
        std::lock_guard<std::mutex> lk(dev->mu);
        hs = dev->last_sched_stream;
        for (TileSchedule *t : dev->schedules) if (dev->last_sched_id && t->id == dev->last_sched_id) sched = t;      // still cached?
        if (sched) sched->busy++;
    }
    if (!sched) return PT_OK;
    struct Unpin { pt_device *d; TileSchedule *t; ~Unpin() { std::lock_guard<std::mutex> lk(d->mu); t->busy--; } } unpin{dev, sched};
    PT_HIP(hipMemcpyAsync(n_split, sched->n_split, sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
    PT_HIP(hipStreamSynchronize(hs));
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_debug_counters_read(pt_device *dev, uint64_t out8[8])
try {
    clear_error();
    if (!dev || !out8) return set_error(PT_ERR_INVALID, "null argument");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipMemcpy(out8, dev->d_counters, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_counters_reset(pt_device *dev)
try {
    clear_error();
    if (!dev) return set_error(PT_ERR_INVALID, "null device");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipDeviceSynchronize());
    PT_HIP(hipMemset(dev->d_counters, 0, 8 * sizeof(unsigned long long)));
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_kernel_time_read(pt_device *dev, double *ms_total, uint64_t *n_launches)
try {
    clear_error();
    if (!dev || !ms_total) return set_error(PT_ERR_INVALID, "null argument");
    PT_HIP(hipSetDevice(dev->index));
    std::lock_guard<std::mutex> lk(dev->mu);
    pt_status st = drain_events(dev);
    if (st != PT_OK) return st;
    *ms_total = dev->ms_total;
    if (n_launches) *n_launches = dev->launches;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_kernel_name_read(pt_device *dev, char *out, size_t cap)
try {
    clear_error();
    if (!dev || !out || cap == 0) return set_error(PT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(dev->mu);
    snprintf(out, cap, "%s", dev->last_kernel);
    return PT_OK;
} PT_CATCH_ALL

#ifdef PT_OCC
PT_API pt_status pt_debug_occ_read(pt_device *dev, unsigned long long *out8, int reset)
try {
    clear_error();
    if (!dev || !out8) return set_error(PT_ERR_INVALID, "bad argument");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipDeviceSynchronize());
    PT_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(dev::g_occ), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; PT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dev::g_occ), z, sizeof z)); }
    return PT_OK;
} PT_CATCH_ALL
#endif

#ifdef PT_TIMELINE
PT_API pt_status pt_debug_timeline_read(pt_device *dev, uint32_t *out, size_t n_words)
try {
    clear_error();
    if (!dev || !out || n_words > 4 * 131072) return set_error(PT_ERR_INVALID, "bad argument");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(dev::g_timeline), n_words * 4));
    return PT_OK;
} PT_CATCH_ALL
#endif

pt_status pt_kernel_time_reset(pt_device *dev)
try {
    clear_error();
    if (!dev) return set_error(PT_ERR_INVALID, "null device");
    PT_HIP(hipSetDevice(dev->index));
    std::lock_guard<std::mutex> lk(dev->mu);
    pt_status st = drain_events(dev);
    if (st != PT_OK) return st;
    dev->ms_total = 0.0; dev->launches = 0;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_bvh_info_read(pt_device *dev, pt_bvh_info *out)
try {
    clear_error();
    if (!dev || !out) return set_error(PT_ERR_INVALID, "null argument");
    if (!dev->has_scene) return set_error(PT_ERR_NO_SCENE, "no scene uploaded");
    out->nodes = (uint32_t)dev->bvh.nodes.size(); out->leaves = dev->bvh.leaves; out->max_leaf = dev->bvh.max_leaf;
    out->depth = dev->bvh.depth; out->max_stack = dev->bvh.max_stack; out->lights = (uint32_t)dev->n_lights;
    out->triangles = (uint32_t)dev->n_tris;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_bvh_copy(pt_device *dev, float *bounds6, uint32_t *links4, int *tri_indices, int *light_faces)
try {
    clear_error();
    if (!dev) return set_error(PT_ERR_INVALID, "null device");
    if (!dev->has_scene) return set_error(PT_ERR_NO_SCENE, "no scene uploaded");
    for (size_t i = 0; i < dev->bvh.nodes.size(); i++) {
        const RefNode &n = dev->bvh.nodes[i];
        if (bounds6) { float *b = bounds6 + 6 * i; b[0] = n.xmin; b[1] = n.xmax; b[2] = n.ymin; b[3] = n.ymax; b[4] = n.zmin; b[5] = n.zmax; }
        if (links4) { uint32_t *l = links4 + 4 * i; l[0] = n.left; l[1] = n.right; l[2] = n.firstTri; l[3] = n.triCount; }
    }
    if (tri_indices) memcpy(tri_indices, dev->bvh.indices.data(), sizeof(int) * dev->n_tris);
    if (light_faces) memcpy(light_faces, dev->light_faces.data(), sizeof(int) * dev->light_faces.size());
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_fb_alloc(pt_device *dev, int width, int height, uint8_t **rgb, uint8_t **yuv)
try {
    clear_error();
    if (!dev || width <= 0 || height <= 0 || !rgb || !yuv) return set_error(PT_ERR_INVALID, "bad argument");
    PT_HIP(hipSetDevice(dev->index));
    size_t total = (size_t)width * (size_t)height;
    DevBufs bufs;                                  // a failure after the first hipMalloc frees it again
    uint8_t *r = nullptr, *y = nullptr;
    PT_HIP(bufs.alloc((void **)&r, total * 3));
    PT_HIP(bufs.alloc((void **)&y, total + 2 * (total / 4)));
    PT_HIP(hipMemset(r, 0, total * 3));
    PT_HIP(hipMemset(y, 0, total + 2 * (total / 4)));
    bufs.release();
    *rgb = r; *yuv = y;
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_fb_free(pt_device *dev, uint8_t *rgb, uint8_t *yuv)
try {
    clear_error();
    if (!dev) return set_error(PT_ERR_INVALID, "null device");
    PT_HIP(hipSetDevice(dev->index));
    if (dev->rgb == rgb) { dev->rgb = nullptr; dev->yuv = nullptr; }
    if (rgb) PT_HIP(hipFree(rgb));
    if (yuv) PT_HIP(hipFree(yuv));
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_fb_download(pt_device *dev, const uint8_t *src, uint8_t *dst, size_t bytes, pt_stream stream)
try {
    clear_error();
    if (!dev || !src || !dst) return set_error(PT_ERR_INVALID, "null argument");
    PT_HIP(hipSetDevice(dev->index));
    PT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    PT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_fb_download_task(pt_device *dev, const pt_task *task, uint8_t *rgb_host, uint8_t *yuv_host, pt_stream stream)
try {
    clear_error();
    if (!dev || !task || !rgb_host || !yuv_host) return set_error(PT_ERR_INVALID, "null argument");
    if (!dev->rgb) return set_error(PT_ERR_NO_SCENE, "no framebuffer bound");
    if (task->width <= 0 || task->height <= 0) return PT_OK;
    const int W = dev->W, H = dev->H;
    if (task->offset_x < 0 || task->offset_y < 0 || task->offset_x + task->width > W || task->offset_y + task->height > H)
        return set_error(PT_ERR_INVALID, "task rectangle leaves the framebuffer");
    PT_HIP(hipSetDevice(dev->index));
    hipStream_t hs = (hipStream_t)stream;
    const int r0 = H - task->offset_y - task->height, r1 = H - task->offset_y;    // fb rows, 0 = top
    const int x0 = task->offset_x, x1 = task->offset_x + task->width;
    size_t off = (size_t)r0 * W * 3 + (size_t)x0 * 3;
    PT_HIP(hipMemcpy2DAsync(rgb_host + off, (size_t)W * 3, dev->rgb + off, (size_t)W * 3, (size_t)(x1 - x0) * 3,
                            (size_t)(r1 - r0), hipMemcpyDeviceToHost, hs));
    off = (size_t)r0 * W + (size_t)x0;
    PT_HIP(hipMemcpy2DAsync(yuv_host + off, (size_t)W, dev->yuv + off, (size_t)W, (size_t)(x1 - x0), (size_t)(r1 - r0),
                            hipMemcpyDeviceToHost, hs));
    const int half = W / 2, total = W * H, uv = total / 4;
    const int c0 = (r0 + 1) / 2, c1 = std::min((r1 + 1) / 2, H / 2);              // chroma rows of even fb rows
    const int k0 = (x0 + 1) / 2, k1 = std::min((x1 + 1) / 2, half);               // chroma cols of even fb cols
    if (c1 > c0 && k1 > k0) {
        for (int plane = 0; plane < 2; plane++) {
            size_t o = (size_t)total + (size_t)plane * uv + (size_t)c0 * half + (size_t)k0;
            PT_HIP(hipMemcpy2DAsync(yuv_host + o, (size_t)half, dev->yuv + o, (size_t)half, (size_t)(k1 - k0),
                                    (size_t)(c1 - c0), hipMemcpyDeviceToHost, hs));
        }
    }
    PT_HIP(hipStreamSynchronize(hs));
    return PT_OK;
} PT_CATCH_ALL

// Layout of a task list in the exchange string (host mirror of PackTask); returns the string's length.
static size_t layout_tasks(int W, int H, const pt_task *tasks, int n, std::vector<dev::PackTask> &out)
{
    out.assign((size_t)n, dev::PackTask());
    const int half = W / 2;
    size_t n_rgb = 0, n_y = 0, n_c = 0;
    for (int i = 0; i < n; i++) {
        dev::PackTask &t = out[(size_t)i];
        const pt_task &k = tasks[i];
        t.x0 = k.offset_x; t.x1 = k.offset_x + k.width;
        t.r0 = H - k.offset_y - k.height; t.r1 = H - k.offset_y;           // framebuffer rows, 0 = top (DevicePathTracer.h:79)
        t.c0 = (t.r0 + 1) / 2; t.c1 = std::min((t.r1 + 1) / 2, H / 2);     // chroma rows of its even framebuffer rows
        t.k0 = (t.x0 + 1) / 2; t.k1 = std::min((t.x1 + 1) / 2, half);      // chroma columns of its even columns
        if (t.c1 < t.c0) t.c1 = t.c0;
        if (t.k1 < t.k0) t.k1 = t.k0;
        n_rgb += (size_t)(t.x1 - t.x0) * 3 * (size_t)(t.r1 - t.r0);
        n_y += (size_t)(t.x1 - t.x0) * (size_t)(t.r1 - t.r0);
        n_c += (size_t)(t.k1 - t.k0) * (size_t)(t.c1 - t.c0);
    }
    size_t o_rgb = 0, o_y = n_rgb, o_u = n_rgb + n_y, o_v = n_rgb + n_y + n_c;
    for (dev::PackTask &t : out) {
        t.off_rgb = o_rgb; t.off_y = o_y; t.off_u = o_u; t.off_v = o_v;
        o_rgb += (size_t)(t.x1 - t.x0) * 3 * (size_t)(t.r1 - t.r0);
        o_y += (size_t)(t.x1 - t.x0) * (size_t)(t.r1 - t.r0);
        const size_t c = (size_t)(t.k1 - t.k0) * (size_t)(t.c1 - t.c0);
        o_u += c; o_v += c;
    }
    return n_rgb + n_y + 2 * n_c;
}

static pt_status check_tasks(const pt_device *dev, const pt_task *tasks, int n)
{
    if (!dev->rgb) return set_error(PT_ERR_NO_SCENE, "no framebuffer bound");
    for (int i = 0; i < n; i++) {
        const pt_task &t = tasks[i];
        if (t.width < 0 || t.height < 0 || t.offset_x < 0 || t.offset_y < 0 || t.offset_x + t.width > dev->W || t.offset_y + t.height > dev->H)
            return set_error(PT_ERR_INVALID, "task rectangle leaves the framebuffer");
    }
    return PT_OK;
}

// Pack (or unpack) on `hs`; `packed` is a device buffer of at least the returned size.  No synchronisation.
static pt_status run_pack(pt_device *dev, bool pack, const pt_task *tasks, int n, uint8_t *packed, hipStream_t hs, size_t *bytes_out)
{
    std::vector<dev::PackTask> lay;
    const size_t bytes = layout_tasks(dev->W, dev->H, tasks, n, lay);
    if (bytes_out) *bytes_out = bytes;
    if (bytes == 0 || n == 0) return PT_OK;
    dev::PackTask *d_lay = nullptr;
    PT_HIP(hipMallocAsync((void **)&d_lay, sizeof(dev::PackTask) * lay.size(), hs));
    hipError_t e = hipMemcpyAsync(d_lay, lay.data(), sizeof(dev::PackTask) * lay.size(), hipMemcpyHostToDevice, hs);
    if (e == hipSuccess) e = hipStreamSynchronize(hs);       // `lay` is a local: the copy must have read it before we return
    if (e != hipSuccess) { (void)hipFreeAsync(d_lay, hs); return set_error(PT_ERR_HIP, "copying the task layout to the device failed"); }
    const dim3 grid(64, (unsigned)n);
    if (pack) hipLaunchKernelGGL(dev::pack_tasks_kernel<true>, grid, dim3(256), 0, hs, dev->rgb, dev->yuv, dev->W, dev->H, d_lay, packed);
    else hipLaunchKernelGGL(dev::pack_tasks_kernel<false>, grid, dim3(256), 0, hs, dev->rgb, dev->yuv, dev->W, dev->H, d_lay, packed);
    e = hipGetLastError();
    (void)hipFreeAsync(d_lay, hs);
    if (e != hipSuccess) return set_error(PT_ERR_HIP, std::string("pack kernel launch failed: ") + hipGetErrorString(e));
    return PT_OK;
}

pt_status pt_fb_packed_size(int width, int height, const pt_task *tasks, int n_tasks, size_t *bytes)
try {
    clear_error();
    if (width <= 0 || height <= 0 || n_tasks < 0 || (n_tasks && !tasks) || !bytes) return set_error(PT_ERR_INVALID, "bad argument");
    std::vector<dev::PackTask> lay;
    *bytes = layout_tasks(width, height, tasks, n_tasks, lay);
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_fb_pack_tasks(pt_device *dev, const pt_task *tasks, int n_tasks, uint8_t *packed_device, pt_stream stream)
try {
    clear_error();
    if (!dev || n_tasks < 0 || (n_tasks && (!tasks || !packed_device))) return set_error(PT_ERR_INVALID, "bad argument");
    pt_status st = check_tasks(dev, tasks, n_tasks);
    if (st != PT_OK) return st;
    PT_HIP(hipSetDevice(dev->index));
    return run_pack(dev, true, tasks, n_tasks, packed_device, (hipStream_t)stream, nullptr);
} PT_CATCH_ALL

pt_status pt_fb_unpack_tasks(pt_device *dev, const pt_task *tasks, int n_tasks, const uint8_t *packed_device, pt_stream stream)
try {
    clear_error();
    if (!dev || n_tasks < 0 || (n_tasks && (!tasks || !packed_device))) return set_error(PT_ERR_INVALID, "bad argument");
    pt_status st = check_tasks(dev, tasks, n_tasks);
    if (st != PT_OK) return st;
    PT_HIP(hipSetDevice(dev->index));
    return run_pack(dev, false, tasks, n_tasks, const_cast<uint8_t *>(packed_device), (hipStream_t)stream, nullptr);
} PT_CATCH_ALL

// ---- native gather over RCCL (one process, one communicator per GPU) ------------------------------------------------
// librccl is loaded on first use of a group of more than one device: a single-GPU build or run never needs it.
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { error = std::string("cannot load librccl: ") + dlerror(); return false; }
#define PT_SYM(field, sym) field = (decltype(field))dlsym(lib, sym); if (!field) { error = std::string("librccl lacks ") + sym; return false; }
        PT_SYM(CommInitAll, "ncclCommInitAll") PT_SYM(CommDestroy, "ncclCommDestroy") PT_SYM(GroupStart, "ncclGroupStart")
        PT_SYM(GroupEnd, "ncclGroupEnd") PT_SYM(Send, "ncclSend") PT_SYM(Recv, "ncclRecv") PT_SYM(GetErrorString, "ncclGetErrorString")
#undef PT_SYM
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;
} // namespace

struct pt_gather {
    std::vector<pt_device *> devs;
    // "rccl": grouped ncclSend / ncclRecv; "peer-copy": hipMemcpyPeerAsync of the same packed strings (librccl missing,
    // ncclCommInitAll failed, PTAMD_GATHER=peer, or one GPU listed twice); "none": a single device
    std::string transport = "none";
    std::string note;                        // why the fallback was taken (empty otherwise)
    std::vector<ncclComm_t> comms;           // one per device (rank i = devs[i]); empty unless transport == "rccl"
    std::vector<hipStream_t> streams;        // one per device
    std::vector<uint8_t *> send;             // device i's packed string (i > 0), on device i
    std::vector<uint8_t *> recv;             // where device 0 receives device i's string
    std::vector<size_t> cap;                 // capacity of send[i] / recv[i]
};

#define PT_NCCL(call)                                                                                       \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess)                                                                              \
            return set_error(PT_ERR_HIP, std::string("RCCL ERROR = ") + std::to_string((int)r_) + " (" +    \
                                             g_rccl.GetErrorString(r_) + ") at " + __FILE__ + ":" +         \
                                             std::to_string(__LINE__) + " '" #call "'");                    \
    } while (0)

static void gather_free(pt_gather *g)
{
    if (!g) return;
    for (size_t i = 0; i < g->devs.size(); i++) {
        (void)hipSetDevice(g->devs[i]->index);
        if (i < g->streams.size() && g->streams[i]) { (void)hipStreamSynchronize(g->streams[i]); (void)hipStreamDestroy(g->streams[i]); }
        if (i < g->send.size()) (void)hipFree(g->send[i]);
    }
    if (!g->devs.empty()) { (void)hipSetDevice(g->devs[0]->index); for (uint8_t *p : g->recv) (void)hipFree(p); }
    for (ncclComm_t c : g->comms) if (c) (void)g_rccl.CommDestroy(c);
    delete g;
}

pt_status pt_gather_create(pt_device *const *devices, int n_devices, pt_gather **out)
try {
    clear_error();
    if (!devices || n_devices <= 0 || !out) return set_error(PT_ERR_INVALID, "bad argument");
    std::unique_ptr<pt_gather, void (*)(pt_gather *)> g(new pt_gather(), gather_free);   // streams, buffers and communicators go with it on every error path
    std::vector<int> ids;
    bool twice = false;
    for (int i = 0; i < n_devices; i++) {
        if (!devices[i]) return set_error(PT_ERR_INVALID, "null device");
        for (int id : ids) if (id == devices[i]->index) twice = true;
        ids.push_back(devices[i]->index);
        g->devs.push_back(devices[i]);
    }
    g->send.assign((size_t)n_devices, nullptr); g->recv.assign((size_t)n_devices, nullptr); g->cap.assign((size_t)n_devices, 0);
    for (int i = 0; i < n_devices; i++) {
        PT_HIP(hipSetDevice(ids[(size_t)i]));
        hipStream_t s;
        PT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        g->streams.push_back(s);
    }
    if (n_devices > 1) {
        const char *force = getenv("PTAMD_GATHER");
        g->transport = "peer-copy";
        if (force && std::string(force) == "peer") g->note = "PTAMD_GATHER=peer";
        else if (twice) g->note = "one GPU is listed more than once (RCCL needs one communicator rank per GPU)";
        else {
            std::lock_guard<std::mutex> lk(g_rccl_mu);
            if (!g_rccl.load()) g->note = g_rccl.error;
            else {
                g->comms.assign((size_t)n_devices, nullptr);
                ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_devices, ids.data());
                if (r == ncclSuccess) g->transport = "rccl";
                else {
                    g->note = std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(r);
                    for (ncclComm_t &c : g->comms) if (c) { (void)g_rccl.CommDestroy(c); c = nullptr; }
                    g->comms.clear();
                }
            }
        }
        if (g->transport == "peer-copy")                 // the copies go device to device: ask for direct access where there is any
            for (int i = 1; i < n_devices; i++) {
                int can = 0;
                if (ids[(size_t)i] != ids[0] && hipDeviceCanAccessPeer(&can, ids[(size_t)i], ids[0]) == hipSuccess && can) {
                    (void)hipSetDevice(ids[(size_t)i]);
                    hipError_t e = hipDeviceEnablePeerAccess(ids[0], 0);
                    if (e != hipSuccess) (void)hipGetLastError();       // already enabled is fine; without it the copy is staged by the runtime
                }
            }
    }
    *out = g.release();
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_gather_destroy(pt_gather *g)
try {
    clear_error();
    gather_free(g);
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_gather_transport(const pt_gather *g, char *out, size_t cap)
try {
    clear_error();
    if (!g || !out || cap == 0) return set_error(PT_ERR_INVALID, "bad argument");
    std::string t = g->transport + (g->note.empty() ? "" : " (" + g->note + ")");
    snprintf(out, cap, "%s", t.c_str());
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_gather_frame(pt_gather *g, const pt_task *tasks, const int *task_device, int n_tasks, uint8_t *rgb_host, uint8_t *yuv_host)
try {
    clear_error();
    if (!g || n_tasks < 0 || (n_tasks && (!tasks || !task_device)) || !rgb_host || !yuv_host) return set_error(PT_ERR_INVALID, "bad argument");
    const int n = (int)g->devs.size();
    pt_device *root = g->devs[0];
    if (!root->rgb) return set_error(PT_ERR_NO_SCENE, "no framebuffer bound");
    // every device's rectangles, in task order
    std::vector<std::vector<pt_task>> mine((size_t)n);
    for (int i = 0; i < n_tasks; i++) {
        if (task_device[i] < 0 || task_device[i] >= n) return set_error(PT_ERR_INVALID, "task_device entry out of range");
        if (tasks[i].width > 0 && tasks[i].height > 0) mine[(size_t)task_device[i]].push_back(tasks[i]);
    }
    std::vector<size_t> bytes((size_t)n, 0);
    for (int i = 1; i < n; i++) {
        pt_device *d = g->devs[(size_t)i];
        if (d->W != root->W || d->H != root->H) return set_error(PT_ERR_INVALID, "pt_gather_frame: the devices' framebuffers differ in size");
        pt_status st = check_tasks(d, mine[(size_t)i].data(), (int)mine[(size_t)i].size());
        if (st != PT_OK) return st;
        std::vector<dev::PackTask> lay;
        bytes[(size_t)i] = layout_tasks(d->W, d->H, mine[(size_t)i].data(), (int)mine[(size_t)i].size(), lay);
        if (bytes[(size_t)i] > g->cap[(size_t)i]) {                       // grow the two staging strings of this peer
            PT_HIP(hipSetDevice(d->index));
            (void)hipFree(g->send[(size_t)i]); g->send[(size_t)i] = nullptr;
            PT_HIP(hipMalloc((void **)&g->send[(size_t)i], bytes[(size_t)i]));
            PT_HIP(hipSetDevice(root->index));
            (void)hipFree(g->recv[(size_t)i]); g->recv[(size_t)i] = nullptr;
            PT_HIP(hipMalloc((void **)&g->recv[(size_t)i], bytes[(size_t)i]));
            g->cap[(size_t)i] = bytes[(size_t)i];
        }
    }
    // 1. every peer packs its rectangles on its own stream (its render streams have been synchronised by the caller)
    for (int i = 1; i < n; i++) {
        if (bytes[(size_t)i] == 0) continue;
        PT_HIP(hipSetDevice(g->devs[(size_t)i]->index));
        pt_status st = run_pack(g->devs[(size_t)i], true, mine[(size_t)i].data(), (int)mine[(size_t)i].size(), g->send[(size_t)i], g->streams[(size_t)i], nullptr);
        if (st != PT_OK) return st;
    }
    // 2. ONE grouped exchange: peer i sends its string to rank 0 over its own xGMI link, rank 0 posts the matching receives
    if (n > 1 && g->transport == "rccl") {
        PT_NCCL(g_rccl.GroupStart());
        // inside the group an error must not return: the group would stay open and swallow every later RCCL call of the
        // process.  Remember the first failure, close the group, report then.
        ncclResult_t bad = ncclSuccess;
        const char *what = "";
        for (int i = 1; i < n && bad == ncclSuccess; i++) {
            if (bytes[(size_t)i] == 0) continue;
            bad = g_rccl.Send(g->send[(size_t)i], bytes[(size_t)i], ncclUint8, 0, g->comms[(size_t)i], g->streams[(size_t)i]);
            if (bad != ncclSuccess) { what = "ncclSend"; break; }
            bad = g_rccl.Recv(g->recv[(size_t)i], bytes[(size_t)i], ncclUint8, i, g->comms[0], g->streams[0]);
            if (bad != ncclSuccess) what = "ncclRecv";
        }
        ncclResult_t end = g_rccl.GroupEnd();
        if (bad != ncclSuccess)
            return set_error(PT_ERR_HIP, std::string("RCCL ERROR = ") + std::to_string((int)bad) + " (" + g_rccl.GetErrorString(bad) + ") in " + what + " of pt_gather_frame (group closed)");
        PT_NCCL(end);
    } else if (n > 1) {
        // peer-copy transport: the same strings, one device-to-device copy per peer on the peer's stream (xGMI where the
        // GPUs have peer access); GPU 0's stream waits for each of them before it scatters
        for (int i = 1; i < n; i++) {
            if (bytes[(size_t)i] == 0) continue;
            pt_device *d = g->devs[(size_t)i];
            PT_HIP(hipSetDevice(d->index));
            PT_HIP(hipMemcpyPeerAsync(g->recv[(size_t)i], root->index, g->send[(size_t)i], d->index, bytes[(size_t)i], g->streams[(size_t)i]));
            hipEvent_t ev;
            PT_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipError_t e = hipEventRecord(ev, g->streams[(size_t)i]);
            if (e == hipSuccess) { (void)hipSetDevice(root->index); e = hipStreamWaitEvent(g->streams[0], ev, 0); }
            (void)hipEventDestroy(ev);                     // destruction is deferred until the recorded work is done
            PT_HIP(e);
        }
    }
    // 3. rank 0 scatters what it received into its framebuffer (its own rectangles are there already) and hands the
    //    whole frame to the host with ONE pair of copies
    PT_HIP(hipSetDevice(root->index));
    for (int i = 1; i < n; i++) {
        if (bytes[(size_t)i] == 0) continue;
        pt_status st = run_pack(root, false, mine[(size_t)i].data(), (int)mine[(size_t)i].size(), g->recv[(size_t)i], g->streams[0], nullptr);
        if (st != PT_OK) return st;
    }
    const size_t total = (size_t)root->W * (size_t)root->H;
    PT_HIP(hipMemcpyAsync(rgb_host, root->rgb, total * 3, hipMemcpyDeviceToHost, g->streams[0]));
    PT_HIP(hipMemcpyAsync(yuv_host, root->yuv, total + 2 * (total / 4), hipMemcpyDeviceToHost, g->streams[0]));
    PT_HIP(hipStreamSynchronize(g->streams[0]));
    for (int i = 1; i < n; i++) { PT_HIP(hipSetDevice(g->devs[(size_t)i]->index)); PT_HIP(hipStreamSynchronize(g->streams[(size_t)i])); }
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_kat_arith(pt_device *dev, int op, const float *a, const float *b, float *out0, float *out1, size_t n)
try {
    clear_error();
    if (!dev || !a || !out0 || op < 0 || op > 4) return set_error(PT_ERR_INVALID, "bad argument");
    if (n == 0) return PT_OK;
    PT_HIP(hipSetDevice(dev->index));
    float *da = nullptr, *db = nullptr, *d0 = nullptr, *d1 = nullptr;
    size_t bytes = n * sizeof(float);
    DevBufs bufs;
    PT_HIP(bufs.alloc((void **)&da, bytes)); PT_HIP(bufs.alloc((void **)&db, bytes));
    PT_HIP(bufs.alloc((void **)&d0, bytes)); PT_HIP(bufs.alloc((void **)&d1, bytes));
    PT_HIP(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    if (b) PT_HIP(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice)); else PT_HIP(hipMemset(db, 0, bytes));
    unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(dev::kat_arith_kernel, dim3(blocks), dim3(256), 0, 0, op, da, db, d0, d1, n);
    PT_HIP(hipGetLastError());
    PT_HIP(hipDeviceSynchronize());
    PT_HIP(hipMemcpy(out0, d0, bytes, hipMemcpyDeviceToHost));
    if (out1) PT_HIP(hipMemcpy(out1, d1, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_kat_exhaustive(pt_device *dev, int op, uint32_t first_bits, uint64_t count, uint64_t *n_bad,
                            uint32_t *examples, int max_examples)
try {
    clear_error();
    if (!dev || !n_bad || op < 0 || op > 4 || max_examples < 0 || (max_examples > 0 && !examples) ||
        count > (1ull << 32))
        return set_error(PT_ERR_INVALID, "bad argument");
    *n_bad = 0;
    if (count == 0) return PT_OK;
    PT_HIP(hipSetDevice(dev->index));
    unsigned long long *dbad = nullptr; uint32_t *dex = nullptr;
    DevBufs bufs;
    PT_HIP(bufs.alloc((void **)&dbad, 16));
    PT_HIP(bufs.alloc((void **)&dex, sizeof(uint32_t) * (size_t)(max_examples > 0 ? max_examples : 1)));
    PT_HIP(hipMemset(dbad, 0, 16));
    PT_HIP(hipMemset(dex, 0, sizeof(uint32_t) * (size_t)(max_examples > 0 ? max_examples : 1)));
    uint64_t want_blocks = (count + 255) / 256;
    unsigned blocks = (unsigned)(want_blocks < 65536 ? want_blocks : 65536);
    hipLaunchKernelGGL(dev::kat_exhaustive_kernel, dim3(blocks), dim3(256), 0, 0, op, first_bits, count, dbad, dex, max_examples);
    PT_HIP(hipGetLastError());
    PT_HIP(hipDeviceSynchronize());
    unsigned long long host_bad[2] = {0, 0};
    PT_HIP(hipMemcpy(host_bad, dbad, 16, hipMemcpyDeviceToHost));
    if (max_examples > 0) PT_HIP(hipMemcpy(examples, dex, sizeof(uint32_t) * (size_t)max_examples, hipMemcpyDeviceToHost));
    *n_bad = host_bad[0];
    return PT_OK;
} PT_CATCH_ALL

static pt_status kat_walk(pt_device *dev, const float *org, const float *dirs, float *t_out, int *face_out, uint32_t *box_out, uint32_t *tri_out, size_t n)
{
    clear_error();
    if (!dev || !org || !dirs) return set_error(PT_ERR_INVALID, "null argument");
    if (!dev->has_scene) return set_error(PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    PT_HIP(hipSetDevice(dev->index));
    RenderParams P;
    memset(&P, 0, sizeof P);
    fill_params(dev, P);
    float *dorg = nullptr, *ddir = nullptr, *dt = nullptr; int *df = nullptr; uint32_t *db = nullptr, *dn = nullptr;
    DevBufs bufs;
    PT_HIP(bufs.alloc((void **)&dorg, n * 12)); PT_HIP(bufs.alloc((void **)&ddir, n * 12));
    PT_HIP(bufs.alloc((void **)&dt, n * 4)); PT_HIP(bufs.alloc((void **)&df, n * 4));
    if (box_out) PT_HIP(bufs.alloc((void **)&db, n * 4));
    if (tri_out) PT_HIP(bufs.alloc((void **)&dn, n * 4));
    PT_HIP(hipMemcpy(dorg, org, n * 12, hipMemcpyHostToDevice));
    PT_HIP(hipMemcpy(ddir, dirs, n * 12, hipMemcpyHostToDevice));
    unsigned blocks = (unsigned)((n + 255) / 256);
    P.queue_dwords = 2; P.shade_lds = 0;
    // a big reference tree (beyond the stack walk's 4 096 inner nodes, where rendering takes the per-lane walk): the wave-uniform
    // walk is the only one that gets through ~2.5e6 triangle tests per ray in reasonable time
    const int uniform = (!dev->sah && dev->n_pairs > 4096) ? 1 : 0;
    if (uniform) P.queue_entries = 0;
    size_t lds = lds_bytes(P, 256, false);
    {
        const pt_status st = ensure_lds(dev, (const void *)dev::kat_hit_kernel, lds);
        if (st != PT_OK) return st;
    }
    hipLaunchKernelGGL(dev::kat_hit_kernel, dim3(blocks), dim3(256), lds, 0, P, dorg, ddir, dt, df, n, uniform, db, dn);
    PT_HIP(hipGetLastError());
    PT_HIP(hipDeviceSynchronize());
    if (t_out) PT_HIP(hipMemcpy(t_out, dt, n * 4, hipMemcpyDeviceToHost));
    if (face_out) PT_HIP(hipMemcpy(face_out, df, n * 4, hipMemcpyDeviceToHost));
    if (box_out) PT_HIP(hipMemcpy(box_out, db, n * 4, hipMemcpyDeviceToHost));
    if (tri_out) PT_HIP(hipMemcpy(tri_out, dn, n * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_kat_closest_hit(pt_device *dev, const float *org, const float *dirs, float *t_out, int *face_out, size_t n)
try {
    if (!t_out || !face_out) { clear_error(); return set_error(PT_ERR_INVALID, "null argument"); }
    return kat_walk(dev, org, dirs, t_out, face_out, nullptr, nullptr, n);
} PT_CATCH_ALL

pt_status pt_kat_walk_cost(pt_device *dev, const float *org, const float *dirs, uint32_t *box_out, uint32_t *tri_out, size_t n)
try {
    return kat_walk(dev, org, dirs, nullptr, nullptr, box_out, tri_out, n);
} PT_CATCH_ALL

} // extern "C"
