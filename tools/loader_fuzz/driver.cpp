// ASan/UBSan driver: loads every file named on the command line through the C ABI loaders.
#include <stdio.h>
#include <string.h>
#include <vector>
#include "pt_amd.h"
int main(int argc, char **argv)
{
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; i++) {
        pt_host_scene *s = nullptr;
        size_t n = strlen(argv[i]);
        pt_status st = (n > 4 && !strcmp(argv[i] + n - 4, ".obj")) ? pt_scene_load_obj(argv[i], &s) : pt_scene_load_glb(argv[i], &s);
        if (st == PT_OK) {
            ok++;
            // both tree builders + the task generators on whatever geometry came out
            const size_t n = pt_scene_triangle_count(s);
            if (n && n < 200000) {
                pt_bvh_info info;
                std::vector<float> bounds(6 * (2 * n)), pairs(16 * n);
                std::vector<uint32_t> links(4 * (2 * n));
                std::vector<int> idx(n), order(n);
                (void)pt_host_bvh_build(pt_scene_triangles(s), n, &info, bounds.data(), links.data(), idx.data());
                (void)pt_host_sah_build(pt_scene_triangles(s), n, &info, pairs.data(), order.data());
            }
            pt_scene_free(s);
        } else bad++;
    }
    printf("loaded %d refused %d\n", ok, bad);
    return 0;
}
