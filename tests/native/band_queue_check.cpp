// CPU check of the DQRB policy's band queue (csrc/host/BandQueue.h) -- test infrastructure, built and run by
// tests/test_host_logic.py with plain g++ against the product's host headers.  No GPU call is made.
//   band_queue_check <width> <height> <threads>
// 1. the bands tile the frame exactly (full width, multiples of 8 rows, 64..128 of them where the height allows);
// 2. `threads` workers pulling concurrently get every band exactly once, the owner map names who pulled it;
// 3. beginFrame() puts last frame's dearest bands first (ties keep the lower band first) and clears the frame's state.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "host/BandQueue.h"

using namespace ptamd;

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080, T = argc > 3 ? atoi(argv[3]) : 8;
    BandQueue q;
    q.build(W, H);
    int y = 0;
    for (const RenderTask &b : q.bands) {
        if (b.offset_x != 0 || b.width != W || b.offset_y != y || b.height <= 0) { printf("band at y %d is not a full-width band in order\n", y); return 1; }
        if (b.height % 8 && b.offset_y + b.height != H) { printf("band height %d is not a multiple of 8\n", b.height); return 1; }
        y += b.height;
    }
    if (y != H) { printf("bands cover %d of %d rows\n", y, H); return 1; }
    const size_t n = q.bands.size();
    if (H >= 8 * 64 && (n < 64 || n > 136)) { printf("%zu bands for %d rows\n", n, H); return 1; }
    for (int frame = 0; frame < 3; frame++) {
        q.beginFrame();
        if (q.everyBandExactlyOnce()) { printf("a fresh frame claims to be complete\n"); return 1; }
        std::vector<std::vector<int>> got((size_t)T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] { for (int b = q.pull(t); b >= 0; b = q.pull(t)) { got[(size_t)t].push_back(b); q.bands[(size_t)b].time = 1000 - b + 7 * (b % 5); } });
        for (auto &x : th) x.join();
        if (!q.everyBandExactlyOnce()) { printf("frame %d: a band was not pulled exactly once\n", frame); return 1; }
        size_t total = 0;
        for (int t = 0; t < T; t++) for (int b : got[(size_t)t]) { total++; if (q.owner[(size_t)b] != t) { printf("owner map disagrees for band %d\n", b); return 1; } }
        if (total != n) { printf("%zu pulls for %zu bands\n", total, n); return 1; }
        if (q.pull(0) != -1) { printf("the queue hands out a band after the last\n"); return 1; }
    }
    q.beginFrame();
    for (size_t k = 1; k < n; k++) {
        const int a = q.order[k - 1], b = q.order[k];
        if (q.bands[(size_t)a].time < q.bands[(size_t)b].time || (q.bands[(size_t)a].time == q.bands[(size_t)b].time && a > b)) { printf("order position %zu is not dearest-first\n", k); return 1; }
    }
    printf("ok: %dx%d -> %zu bands of %d rows, %d workers\n", W, H, n, q.bands[0].height, T);
    return 0;
}
