// pt_common.cpp -- error reporting and the host-side task generators of the C ABI.
#include <stdlib.h>

#include "pt_internal.h"

#include <new>
#include <stdexcept>

#include <cstring>
#include <vector>

namespace pt {

static thread_local std::string g_error;

pt_status set_error(pt_status code, const std::string &msg)
{
    g_error = msg;
    return code;
}
void clear_error() { g_error.clear(); }
pt_status exception_status() noexcept
{
    try {
        try { throw; }
        catch (const std::bad_alloc &) { return set_error(PT_ERR_NOMEM, "host allocation failed"); }
        catch (const std::length_error &e) { return set_error(PT_ERR_NOMEM, std::string("host allocation failed: ") + e.what()); }
        catch (const std::exception &e) { return set_error(PT_ERR_INTERNAL, std::string("internal error: ") + e.what()); }
        catch (...) { return set_error(PT_ERR_INTERNAL, "internal error: unknown exception"); }
    } catch (...) {
        return PT_ERR_NOMEM;       // even the message could not be stored
    }
}

} // namespace pt

using namespace pt;

extern "C" {

const char *pt_last_error_string(void) { return g_error.c_str(); }
const char *pt_version(void) { return "ptamd 0.1 (gfx950)"; }

// Hardware queues.  Launches on different streams only overlap on the device while each stream has a hardware queue of its
// own; the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES queues (4 unless the variable says otherwise) and reads
// the variable ONCE, when it initialises.  Frames in flight (RendererConfig::framesInFlight, scheduling.py) want one queue per
// frame: a fourth launch that shares a queue waits behind another stream's persistent kernel (an eighth of C2 with four frames
// in flight: 9.9 ms per frame on 4 queues, 7.8 on 8).  So the library asks for 8 when it is LOADED, unless the host has set the
// variable itself -- for a host that links libptamd.so that is before main(), before any HIP call -- and tells what it found:
// the hosts cap their frames in flight at pt_hw_queue_count().  (A process that initialised the HIP runtime BEFORE loading the
// library, with the variable unset, runs on the runtime's 4 whatever is set afterwards: export GPU_MAX_HW_QUEUES there.)
static int g_hw_queues = 4;
__attribute__((constructor)) static void ptamd_on_load()
{
    (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    const int v = e ? atoi(e) : 4;
    g_hw_queues = v >= 1 ? v : 4;
}
int pt_hw_queue_count(void) { return g_hw_queues; }

pt_status pt_host_bvh_build(const pt_triangle *tris, size_t n, pt_bvh_info *info, float *bounds6,
                            uint32_t *links4, int *tri_indices)
try {
    clear_error();
    if (!tris || n == 0) return set_error(PT_ERR_INVALID, "no triangles");
    std::vector<float> pos(9 * n);
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            pos[9 * i + k] = tris[i].v0.position[k];
            pos[9 * i + 3 + k] = tris[i].v1.position[k];
            pos[9 * i + 6 + k] = tris[i].v2.position[k];
        }
    }
    RefBvh bvh;
    build_reference_bvh(pos.data(), n, bvh);
    if (info) {
        info->nodes = (uint32_t)bvh.nodes.size(); info->leaves = bvh.leaves; info->max_leaf = bvh.max_leaf;
        info->depth = bvh.depth; info->max_stack = bvh.max_stack; info->lights = 0; info->triangles = (uint32_t)n;
    }
    for (size_t i = 0; i < bvh.nodes.size(); i++) {
        const RefNode &nd = bvh.nodes[i];
        if (bounds6) { float *b = bounds6 + 6 * i; b[0] = nd.xmin; b[1] = nd.xmax; b[2] = nd.ymin; b[3] = nd.ymax; b[4] = nd.zmin; b[5] = nd.zmax; }
        if (links4) { uint32_t *l = links4 + 4 * i; l[0] = nd.left; l[1] = nd.right; l[2] = nd.firstTri; l[3] = nd.triCount; }
    }
    if (tri_indices) for (size_t i = 0; i < n; i++) tri_indices[i] = bvh.indices[i];
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_host_sah_build(const pt_triangle *tris, size_t n, pt_bvh_info *info, float *pairs16, int *order)
try {
    clear_error();
    if (!tris || n == 0) return set_error(PT_ERR_INVALID, "no triangles");
    std::vector<float> pos(9 * n);
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            pos[9 * i + k] = tris[i].v0.position[k];
            pos[9 * i + 3 + k] = tris[i].v1.position[k];
            pos[9 * i + 6 + k] = tris[i].v2.position[k];
        }
    SahBvh bvh;
    build_sah_bvh(pos.data(), n, bvh, sah_leaf_size_for(n));
    if (info) {
        info->nodes = (uint32_t)bvh.pairs.size(); info->leaves = 0; info->max_leaf = bvh.max_leaf; info->depth = bvh.max_stack;
        info->max_stack = bvh.max_stack; info->lights = 0; info->triangles = (uint32_t)n;
        for (const NodePair &p : bvh.pairs) info->leaves += (p.lcnt != 0) + (p.rcnt != 0);
    }
    static_assert(sizeof(NodePair) == 64, "NodePair is 16 dwords");
    if (pairs16) memcpy(pairs16, bvh.pairs.data(), bvh.pairs.size() * sizeof(NodePair));
    if (order) memcpy(order, bvh.order.data(), n * sizeof(int));
    return PT_OK;
} PT_CATCH_ALL

// RenderManager::getTaskLayout (RenderManager.h:42-59) + TaskGenerator::generateEqualTasks
// (TaskGenerator.h:58-80): rows of at most max_tasks_in_row rectangles; the last column
// and the last row absorb the remainders.
pt_status pt_tasks_equal_layout(int task_count, int max_tasks_in_row, int width, int height,
                                pt_task *out)
try {
    clear_error();
    if (task_count <= 0 || max_tasks_in_row <= 0 || width <= 0 || height <= 0 || !out)
        return set_error(PT_ERR_INVALID, "bad task layout request");
    std::vector<std::vector<int>> layout;
    int task = 0;
    while (task < task_count) {
        layout.emplace_back();
        int in_row = 0;
        while (in_row < max_tasks_in_row && task < task_count) {
            layout.back().push_back(task);
            task++; in_row++;
        }
    }
    int task_h = height / (int)layout.size();
    for (size_t i = 0; i < layout.size(); i++) {
        int task_w = width / (int)layout[i].size();
        for (size_t j = 0; j < layout[i].size(); j++) {
            pt_task &t = out[layout[i][j]];
            t.width = task_w; t.offset_x = task_w * (int)j;
            t.height = task_h; t.offset_y = task_h * (int)i;
        }
    }
    for (size_t i = 0; i < layout.size(); i++) {
        pt_task &t = out[layout[i].back()];
        t.width = width - t.offset_x;
    }
    for (int idx : layout.back()) out[idx].height = height - out[idx].offset_y;
    return PT_OK;
} PT_CATCH_ALL

// TaskGenerator::generateEqualTasks(task_count, width, height), TaskGenerator.h:46-55.
pt_status pt_tasks_equal_row(int task_count, int width, int height, pt_task *out)
try {
    clear_error();
    if (task_count <= 0 || width <= 0 || height <= 0 || !out)
        return set_error(PT_ERR_INVALID, "bad task request");
    int task_w = width / task_count;
    for (int i = 0; i < task_count - 1; i++) out[i] = {task_w, height, i * task_w, 0};
    out[task_count - 1] = {width - (task_count - 1) * task_w, height, (task_count - 1) * task_w, 0};
    return PT_OK;
} PT_CATCH_ALL

// Multi-GPU partition: full-width bands of band_rows framebuffer rows, dealt
// round-robin to ranks.  Framebuffer row r (0 = top) is task row y = H-1-r, so band b covers
// fb rows [b*band_rows, min(H,(b+1)*band_rows)) = task offset_y = H - end, height = end-start.
pt_status pt_tasks_row_bands(int width, int height, int band_rows, int n_ranks, int rank,
                             pt_task *out, int max_tasks, int *n_out)
try {
    clear_error();
    if (width <= 0 || height <= 0 || band_rows <= 0 || (band_rows & 1) || n_ranks <= 0 || rank < 0 ||
        rank >= n_ranks || !n_out || (max_tasks > 0 && !out))
        return set_error(PT_ERR_INVALID, "bad row-band request (band_rows must be a positive even number)");
    int n_bands = (height + band_rows - 1) / band_rows;
    int n = 0;
    for (int b = rank; b < n_bands; b += n_ranks) {
        int start = b * band_rows;
        int end = start + band_rows < height ? start + band_rows : height;
        if (n < max_tasks) out[n] = {width, end - start, 0, height - end};
        n++;
    }
    *n_out = n;
    if (n > max_tasks) return set_error(PT_ERR_INVALID, "out_tasks too small");
    return PT_OK;
} PT_CATCH_ALL

} // extern "C"
