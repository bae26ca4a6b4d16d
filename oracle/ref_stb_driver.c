/* ref_stb_driver.c -- TEST INFRASTRUCTURE ONLY (like everything under oracle/).
 *
 * The one piece of the reference's texture path that is plain C and vendored in its tree is
 * third-party/stb_image.h (used by SceneLoader::loadTextureFromFile, HostScene.cpp:6-7,19-30).
 * This translation unit does nothing but instantiate that header FROM WHERE IT LIES under
 * /root/reference (-I$(REFERENCE); nothing is copied into this repo) so that the product's own PNG
 * and JPEG decoders (csrc/pt_scene_io.cpp, csrc/pt_jpeg.cpp) can be pinned against the real thing.  Built by oracle/Makefile
 * into oracle/_ref/libstb_ref.so only when /root/reference exists; never linked into the product.
 */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_STDIO
#include "third-party/stb_image.h"

#include <string.h>

/* stbi_load_from_memory exactly as HostScene.cpp:19-26 calls it for an embedded texture (req_comp = 0).
 * Returns the number of decoded bytes (w*h*comp), -1 when stb_image refuses the data, -2 when `cap`
 * is too small; the bytes go to `out`. */
long ptref_stbi_load_from_memory(const unsigned char *data, int len, int req_comp, int *w, int *h, int *comp,
                                 unsigned char *out, long cap)
{
    unsigned char *p = stbi_load_from_memory(data, len, w, h, comp, req_comp);
    if (!p) return -1;
    long n = (long)*w * (long)*h * (long)(req_comp ? req_comp : *comp);
    if (n > cap) { stbi_image_free(p); return -2; }
    if (out) memcpy(out, p, (size_t)n);
    stbi_image_free(p);
    return n;
}

const char *ptref_stbi_failure_reason(void) { return stbi_failure_reason(); }
