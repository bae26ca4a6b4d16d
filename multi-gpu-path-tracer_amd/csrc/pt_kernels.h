// pt_kernels.h -- launch parameter block shared by pt_device.hip's host and device code.
#pragma once

#include <stdint.h>

#include "pt_internal.h"

namespace pt {

struct CameraBasis {          // camera.h:21-36, evaluated once per launch on the host
    v3 origin, llc, horizontal, vertical;
};

struct RenderParams {
    const NodePair *nodes;
    const TriIsect *tris;
    const TriShade *shade;
    const float4 *onb;                // 3 x float4 per slot {a0, a1, a2, pad}: onb(normal) per triangle (dealt kernel only)
    const LightRec *lights;
    const MatRec *mats;
    const TexDesc *texs;
    const float *texels;
    const float4 *legacy_mats;        // extension (pt_set_legacy_scene): 2 x float4 per material {albedo, param}, {kind}
    const float4 *spheres;            // extension: 2 x float4 per sphere {centre, radius}, {material}
    int n_spheres;
    uint8_t *rgb;
    uint8_t *yuv;
    unsigned long long *counters;     // 5 x u64: paths, segments, box, tri, light
    const uint32_t *tile_order;       // adaptive schedule: wave slot -> tile (a permutation), or null = raster order
    uint32_t *tile_cost;              // adaptive schedule: shader-clock cost of each tile, written per wave, or null
    const uint32_t *n_split;          // SAH kernel: how many of the dearest tiles (the first of tile_order) go to `split_parts` waves each, or null
    int split_parts;                  // 16: parts of 4 x 1 pixels; 32: 2 x 1; 64: single pixels
    uint32_t *tile_next;              // persistent launches: the next slot to hand out (zeroed before the launch), or null
    uint32_t n_slots;                 // persistent launches: slots (tiles) of this launch
    CameraBasis cam;
    double light_index_scale;         // (n-1) + 0.999999, hitable_list.h:24
    float light_weight;               // 1.0f / n, hitable_list.h:17
    int n_lights;
    int W, H;                         // Resolution
    int task_w, task_h, task_ox, task_oy;
    int tiles_x, tiles_y;             // 8x8 pixel tiles covering the task
    int band_rows, band_ranks, band_rank;   // band_ranks > 0: task rows are the rows of the row bands owned by band_rank
    const int *band_list;             // non-null: task rows are the rows of these bands, in this order (pt_render_band_list_async)
    int spp;
    unsigned depth;
    int stack_entries;                // per-lane LDS stack slots (reference tree: max_stack - 1; SAH tree: its bound, or fewer with stack_spill)
    int stack_cap;                    // SAH walks: entries a lane may hold in LDS before it spills (SahStack::cap); huge when everything fits
    uint32_t *stack_spill;            // persistent SAH kernel: where stacks deeper than stack_entries continue (SahStack), or null
    int spill_blocks;                 // ... blocks of 8 entries per lane in it
    int top_records;                  // persistent SAH kernel: the first records of `nodes` (the tree's top levels) staged in LDS, or 0
    int top_record_bytes;             // ... 64 or 128 bytes each
    int queue_entries;                // per-lane LDS leaf-queue slots (>= 2; 0 for the wave-uniform walks)
    int queue_dwords;                 // dwords per queue entry (1 packed, 2 otherwise)
    int n_pairs, n_tris;              // array lengths (for LDS staging)
    int n_mats;                       // material records
    int spec_pct;                     // SAH kernel: leave the walk loop when <= this % of the walkers remain
    int sah;                          // 1: nodes/tris hold the opt-in SAH tree (traverse_sah)
    int shade_lds;                    // 1: shade / material / light tables are staged in LDS; 2: material and light tables only
    int deal_cap;                     // dealt walk (kWalk 4): per-wave item list capacity, 0 otherwise
    int deal_min;                     // ... a leaf hit by >= this many lanes of the wave is tested in place
    int part_boxes;                   // dealt walk: child boxes + reach table staged in LDS, PARTS of split tiles walk by lanes (part_walk_by_lanes)
    int lrpt;                         // dealt kernel: s_setprio by the share of samples still to trace
    int any_tex;                      // some material has a texture: hits need their barycentrics
    int fold_bits;                    // per-lane 32-bit test counters are folded into wider ones at 2^fold_bits (31; tests lower it)
};

CameraBasis camera_basis(const pt_camera &cfg);

} // namespace pt
