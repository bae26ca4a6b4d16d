// ASan/UBSan driver: loads every file named on the command line through the C ABI loaders.
#include <stdio.h>
#include <string.h>
#include "pt_amd.h"
int main(int argc, char **argv)
{
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; i++) {
        pt_host_scene *s = nullptr;
        size_t n = strlen(argv[i]);
        pt_status st = (n > 4 && !strcmp(argv[i] + n - 4, ".obj")) ? pt_scene_load_obj(argv[i], &s) : pt_scene_load_glb(argv[i], &s);
        if (st == PT_OK) { ok++; pt_scene_free(s); } else bad++;
    }
    printf("loaded %d refused %d\n", ok, bad);
    return 0;
}
