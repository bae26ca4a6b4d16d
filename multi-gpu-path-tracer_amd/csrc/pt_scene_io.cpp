// pt_scene_io.cpp -- HostScene producers and the PPM sink (host only, no GPU).
//
// Replaces SceneLoader (HostScene.cpp:98-278), which leans on assimp 5.4.3 + stb_image
// (neither available here): a direct glTF-2.0-binary reader that reproduces what the
// reference's assimp pipeline hands to HostScene for the inputs in scope:
//   aiProcess_Triangulate | FindDegenerates | PreTransformVertices | SortByPType
//   (HostScene.cpp:101-108) => world transforms baked into vertices in float, triangles
//   regrouped by material index (ascending) in node-traversal order, degenerate faces
//   dropped; glTF2 importer => V flipped (v <- 1-v), baseColorFactor / emissiveFactor
//   always present.
// Parity of the render is defined downstream of HostScene.
#include "pt_internal.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

namespace pt {

void HostSceneImpl::rebuild_texture_views()
{
    for (size_t i = 0; i < textures.size(); i++) textures[i].data = texel_storage[i].data();
}

namespace {

// ----------------------------------------------------------------------- mini JSON --
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json *get(const char *key) const
    {
        if (kind != Obj) return nullptr;
        for (const auto &kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
    const Json &at(size_t i) const { return arr[i]; }
    double number(double dflt) const { return kind == Num ? num : dflt; }
    // out-of-range and non-finite numbers are not indices: the default (callers pass an invalid one) stands
    int integer(int dflt) const { return kind == Num && num >= -2147483648.0 && num <= 2147483647.0 ? (int)num : dflt; }
};

// A byte offset, length, stride or element count: absent -> dflt; anything but a finite number in [0, 2^48) -> false.
static bool json_size(const Json *j, size_t dflt, size_t &out)
{
    out = dflt;
    if (!j) return true;
    if (j->kind != Json::Num || !(j->num >= 0.0 && j->num < 281474976710656.0)) return false;
    out = (size_t)j->num;
    return true;
}

// Do `count` elements of `elem` bytes, `stride` apart, starting at `off`, lie inside `len` bytes?  (No product that
// can wrap: every operand is below 2^48 but count x stride is not.)
static bool span_fits(size_t off, size_t count, size_t stride, size_t elem, size_t len)
{
    if (count == 0) return off <= len;
    if (elem > len || off > len - elem) return false;
    return stride == 0 || count - 1 <= (len - elem - off) / stride;
}

struct JsonParser {
    const char *p, *end;
    bool ok = true;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool lit(const char *s)
    {
        size_t n = strlen(s);
        if ((size_t)(end - p) >= n && memcmp(p, s, n) == 0) { p += n; return true; }
        return false;
    }
    Json value(int depth = 0)
    {
        Json j;
        ws();
        if (p >= end || depth > 64) { ok = false; return j; }
        if (*p == '{') {
            p++; j.kind = Json::Obj; ws();
            if (p < end && *p == '}') { p++; return j; }
            while (ok) {
                ws();
                if (p >= end || *p != '"') { ok = false; break; }
                Json k = string();
                ws();
                if (p >= end || *p != ':') { ok = false; break; }
                p++;
                Json v = value(depth + 1);
                j.obj.emplace_back(k.str, std::move(v));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; break; }
                ok = false;
            }
        } else if (*p == '[') {
            p++; j.kind = Json::Arr; ws();
            if (p < end && *p == ']') { p++; return j; }
            while (ok) {
                j.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; break; }
                ok = false;
            }
        } else if (*p == '"') {
            j = string();
        } else if (lit("true")) { j.kind = Json::Bool; j.b = true; }
        else if (lit("false")) { j.kind = Json::Bool; j.b = false; }
        else if (lit("null")) { j.kind = Json::Null; }
        else {
            char *e = nullptr;
            std::string tmp(p, (size_t)std::min<ptrdiff_t>(end - p, 64));
            double d = strtod(tmp.c_str(), &e);
            if (e == tmp.c_str()) { ok = false; return j; }
            p += (e - tmp.c_str());
            j.kind = Json::Num; j.num = d;
        }
        return j;
    }
    Json string()
    {
        Json j; j.kind = Json::Str;
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                p++;
                switch (*p) {
                case 'n': j.str += '\n'; break;
                case 't': j.str += '\t'; break;
                case 'r': j.str += '\r'; break;
                case 'b': j.str += '\b'; break;
                case 'f': j.str += '\f'; break;
                case 'u':
                    if (p + 4 < end) {
                        unsigned cp = (unsigned)strtoul(std::string(p + 1, 4).c_str(), nullptr, 16);
                        j.str += (cp < 0x80) ? (char)cp : '?';
                        p += 4;
                    }
                    break;
                default: j.str += *p; break;
                }
                p++;
            } else {
                j.str += *p++;
            }
        }
        if (p >= end) ok = false; else p++;
        return j;
    }
};

// -------------------------------------------------------------------- PNG decoding --
// Stand-in for stbi_load_from_memory(..., req_comp = 0) (HostScene.cpp:19-26): returns
// 8-bit samples with the file's native channel count.
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

} // namespace
pt_status decode_jpeg(const uint8_t *data, size_t len, int &w, int &h, int &channels, std::vector<uint8_t> &out);   // pt_jpeg.cpp
namespace {

// Reverses the PNG filters of one (sub-)image of pw x ph pixels in place; `p` holds ph rows of
// 1 filter byte + `stride` data bytes.
pt_status png_unfilter(uint8_t *p, size_t stride, int ph, size_t bpp)
{
    std::vector<uint8_t> prev(stride, 0);
    for (int y = 0; y < ph; y++) {
        uint8_t *line = p + (size_t)y * (stride + 1);
        const int ft = line[0];
        uint8_t *cur = line + 1;
        if (ft > 4) return set_error(PT_ERR_IO, "bad PNG filter type");
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int v = cur[i];
            switch (ft) {
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: v += paeth(a, b, c); break;
            default: break;
            }
            cur[i] = (uint8_t)v;
        }
        memcpy(prev.data(), cur, stride);
    }
    return PT_OK;
}

// Behaviour pinned against the reference's own third-party/stb_image.h (tests/test_oracle_pins.py,
// oracle/_ref): every colour type, bit depths 1..16, all filters, Adam7 interlacing, palette and
// colour-key transparency (tRNS), 16-bit samples reduced to their high byte.
pt_status decode_png(const uint8_t *data, size_t len, int &w, int &h, int &channels,
                     std::vector<uint8_t> &out)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (len < 8 || memcmp(data, sig, 8) != 0) return set_error(PT_ERR_UNSUPPORTED, "embedded image is not a PNG");
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, have_trns = false;
    while (pos + 12 <= len) {
        uint32_t clen = be32(data + pos);
        const uint8_t *type = data + pos + 4;
        const uint8_t *body = data + pos + 8;
        if (pos + 12 + (size_t)clen > len) return set_error(PT_ERR_IO, "truncated PNG chunk");
        if (!memcmp(type, "IHDR", 4) && clen >= 13) {
            w = (int)be32(body); h = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + clen);
        else if (!memcmp(type, "tRNS", 4)) { trns.assign(body, body + clen); have_trns = true; }
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + clen);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)clen;
    }
    if (!have_ihdr || w <= 0 || h <= 0) return set_error(PT_ERR_IO, "PNG without IHDR");
    // stb_image.h: STBI_MAX_DIMENSIONS (1 << 24) per side and "too large" beyond 2^30 bytes; a texture of
    // more than 2^28 pixels is a corrupt header, not a texture
    if (w > (1 << 24) || h > (1 << 24) || (uint64_t)w * (uint64_t)h > ((uint64_t)1 << 28))
        return set_error(PT_ERR_IO, "PNG too large");
    if (interlace > 1) return set_error(PT_ERR_IO, "bad PNG interlace method");
    int spp;   // samples per pixel in the file
    switch (ctype) {
    case 0: spp = 1; break;
    case 2: spp = 3; break;
    case 3: spp = 1; break;
    case 4: spp = 2; break;
    case 6: spp = 4; break;
    default: return set_error(PT_ERR_IO, "bad PNG colour type");
    }
    if (!(depth == 8 || (depth == 16 && ctype != 3) || (depth < 8 && (depth == 1 || depth == 2 || depth == 4) && (ctype == 0 || ctype == 3))))
        return set_error(PT_ERR_UNSUPPORTED, "unsupported PNG bit depth");
    if (ctype == 3 && plte.empty()) return set_error(PT_ERR_IO, "palette PNG without PLTE");
    if (have_trns) {                                            // stb_image.h: "tRNS with alpha" / "bad tRNS len"
        if (ctype == 4 || ctype == 6) return set_error(PT_ERR_IO, "PNG tRNS chunk with an alpha colour type");
        if (ctype == 3 && trns.size() > plte.size() / 3) return set_error(PT_ERR_IO, "bad PNG tRNS length");
        if (ctype != 3 && trns.size() != (size_t)spp * 2) return set_error(PT_ERR_IO, "bad PNG tRNS length");
    }
    const size_t bits_pp = (size_t)spp * (size_t)depth;
    const size_t bpp = std::max<size_t>(1, bits_pp / 8);
    // sub-images: the whole picture, or the seven Adam7 passes {x0, y0, dx, dy}
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const int whole[1][4] = {{0, 0, 1, 1}};
    const int (*pass)[4] = interlace ? adam7 : whole;
    const int n_pass = interlace ? 7 : 1;
    size_t raw_need = 0;
    for (int k = 0; k < n_pass; k++) {
        const int pw = (w - pass[k][0] + pass[k][2] - 1) / pass[k][2], ph = (h - pass[k][1] + pass[k][3] - 1) / pass[k][3];
        if (pw > 0 && ph > 0) raw_need += (((size_t)pw * bits_pp + 7) / 8 + 1) * (size_t)ph;
    }
    // deflate cannot expand by more than ~1032:1: do not allocate what the data cannot fill
    if (raw_need > idat.size() * 1040 + 1024) return set_error(PT_ERR_IO, "PNG inflate failed");
    std::vector<uint8_t> raw(raw_need);
    uLongf rawlen = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || rawlen != raw.size()) return set_error(PT_ERR_IO, "PNG inflate failed");
    // samples at file precision, whole picture
    std::vector<uint16_t> smp((size_t)w * (size_t)h * (size_t)spp);
    size_t off = 0;
    for (int k = 0; k < n_pass; k++) {
        const int x0 = pass[k][0], y0 = pass[k][1], dx = pass[k][2], dy = pass[k][3];
        const int pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = ((size_t)pw * bits_pp + 7) / 8;
        pt_status st = png_unfilter(raw.data() + off, stride, ph, bpp);
        if (st != PT_OK) return st;
        for (int y = 0; y < ph; y++) {
            const uint8_t *cur = raw.data() + off + (size_t)y * (stride + 1) + 1;
            for (int x = 0; x < pw; x++) {
                uint16_t *o = &smp[(((size_t)(y0 + y * dy)) * (size_t)w + (size_t)(x0 + x * dx)) * (size_t)spp];
                for (int c = 0; c < spp; c++) {
                    if (depth == 8) o[c] = cur[(size_t)x * spp + c];
                    else if (depth == 16) o[c] = (uint16_t)((cur[((size_t)x * spp + c) * 2] << 8) | cur[((size_t)x * spp + c) * 2 + 1]);
                    else {
                        const size_t bit = (size_t)x * depth;
                        o[c] = (uint16_t)((cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1));
                    }
                }
            }
        }
        off += (stride + 1) * (size_t)ph;
    }
    // expand to 8-bit samples: palette -> RGB (RGBA with tRNS); colour key -> extra alpha channel;
    // grey below 8 bits scaled to 0..255; 16-bit -> high byte
    const bool key = have_trns && ctype != 3;
    channels = (ctype == 3) ? (have_trns ? 4 : 3) : spp + (key ? 1 : 0);
    out.resize((size_t)w * (size_t)h * (size_t)channels);
    uint16_t keyv[3] = {0, 0, 0};
    if (key) for (int c = 0; c < spp; c++) {
        const uint16_t v = (uint16_t)((trns[2 * c] << 8) | trns[2 * c + 1]);
        keyv[c] = depth == 16 ? v : (uint16_t)(v & 255);      // stb_image.h keeps the low byte of a key below 16 bits
    }
    const int scale = depth < 8 ? 255 / ((1 << depth) - 1) : 1;
    for (size_t px = 0; px < (size_t)w * (size_t)h; px++) {
        const uint16_t *in = &smp[px * (size_t)spp];
        uint8_t *o = &out[px * (size_t)channels];
        if (ctype == 3) {
            const size_t idx = in[0];
            for (int c = 0; c < 3; c++) o[c] = (idx * 3 + c < plte.size()) ? plte[idx * 3 + c] : 0;   // past the palette: undefined in stb
            if (channels == 4) o[3] = idx < trns.size() ? trns[idx] : 255;
        } else {
            bool is_key = key;
            for (int c = 0; c < spp; c++) {
                o[c] = depth == 16 ? (uint8_t)(in[c] >> 8) : (uint8_t)(in[c] * scale);
                // the key is compared at 16 bits for 16-bit files, on the scaled 8-bit value otherwise
                if (key) is_key = is_key && (depth == 16 ? in[c] == keyv[c] : o[c] == (uint8_t)(keyv[c] * scale));
            }
            if (key) o[spp] = is_key ? 0 : 255;
        }
    }
    return PT_OK;
}

// ---------------------------------------------------------------------- glTF maths --
// Row-major 4x4 in float, mirroring assimp's aiMatrix4x4 (a1..d4) arithmetic.
struct M4 { float m[4][4]; };

M4 m4_identity()
{
    M4 r; memset(&r, 0, sizeof r);
    r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
    return r;
}
M4 m4_mul(const M4 &a, const M4 &b)
{
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
float m4_det3(const M4 &a)
{
    return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1])
         - a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0])
         + a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}

struct Accessor {
    const uint8_t *base = nullptr;
    size_t count = 0, stride = 0;
    int componentType = 0, ncomp = 0;
    bool normalized = false;
    std::shared_ptr<std::vector<uint8_t>> owned;     // sparse accessors are materialised here
};

size_t comp_size(int ct)
{
    switch (ct) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    default: return 0;
    }
}
int type_ncomp(const std::string &t)
{
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4") return 4;
    if (t == "MAT4") return 16;
    return 0;
}

} // namespace

// ------------------------------------------------------------------- GLB -> HostScene
static pt_status load_glb_impl(const char *path, HostSceneImpl &scene)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return set_error(PT_ERR_IO, std::string("cannot open ") + path);
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const uint8_t *json_ptr = nullptr, *bin_ptr = nullptr;
    size_t json_len = 0, bin_len = 0;
    size_t first = 0;
    if (file.size() >= 3 && file[0] == 0xEF && file[1] == 0xBB && file[2] == 0xBF) first = 3;      // UTF-8 BOM
    while (first < file.size() && (file[first] == ' ' || file[first] == '\n' || file[first] == '\r' || file[first] == '\t')) first++;
    if (file.size() >= 20 && memcmp(file.data(), "glTF", 4) == 0) {
        auto le32 = [&](size_t off) { uint32_t v; memcpy(&v, file.data() + off, 4); return v; };
        if (le32(4) != 2) return set_error(PT_ERR_UNSUPPORTED, "only glTF version 2 is supported");
        size_t total = std::min<size_t>(le32(8), file.size());
        size_t pos = 12;
        while (pos + 8 <= total) {
            uint32_t clen = le32(pos), ctype = le32(pos + 4);
            if (pos + 8 + (size_t)clen > total) return set_error(PT_ERR_IO, "truncated GLB chunk");
            if (ctype == 0x4E4F534Au && !json_ptr) { json_ptr = file.data() + pos + 8; json_len = clen; }
            else if (ctype == 0x004E4942u && !bin_ptr) { bin_ptr = file.data() + pos + 8; bin_len = clen; }
            pos += 8 + (size_t)clen;
        }
        if (!json_ptr) return set_error(PT_ERR_IO, "GLB has no JSON chunk");
    } else if (first < file.size() && file[first] == '{') {
        json_ptr = file.data() + first; json_len = file.size() - first;       // .gltf: JSON text, buffers and images by uri
    } else {
        return set_error(PT_ERR_IO, "not a glTF file (neither a .glb container nor .gltf JSON)");
    }
    JsonParser jp{(const char *)json_ptr, (const char *)json_ptr + json_len};
    Json root = jp.value();
    if (!jp.ok || root.kind != Json::Obj) return set_error(PT_ERR_IO, "glTF JSON does not parse");

    // ---- buffers: the GLB BIN chunk (buffer 0 without a uri), data: URIs, files next to the .gltf
    std::string dir(path);
    {
        size_t slash = dir.find_last_of("/\\");
        dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    }
    auto load_uri = [&](const std::string &uri, std::vector<uint8_t> &out_bytes, bool &was_data_uri) -> pt_status {
        was_data_uri = uri.compare(0, 5, "data:") == 0;
        if (was_data_uri) {
            size_t comma = uri.find(',');
            if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma)
                return set_error(PT_ERR_UNSUPPORTED, "only base64 data: URIs are supported");
            out_bytes.clear();
            uint32_t acc = 0; int nb = 0;
            for (size_t i = comma + 1; i < uri.size(); i++) {
                const char ch = uri[i];
                int v;
                if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
                else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
                else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
                else if (ch == '+' || ch == '-') v = 62;
                else if (ch == '/' || ch == '_') v = 63;
                else continue;                                   // padding, line breaks
                acc = (acc << 6) | (uint32_t)v; nb += 6;
                if (nb >= 8) { nb -= 8; out_bytes.push_back((uint8_t)((acc >> nb) & 0xff)); }
            }
            return PT_OK;
        }
        std::string rel;
        for (size_t i = 0; i < uri.size(); i++) {                // percent-decoding
            if (uri[i] == '%' && i + 2 < uri.size() && isxdigit((unsigned char)uri[i + 1]) && isxdigit((unsigned char)uri[i + 2])) {
                rel.push_back((char)strtol(uri.substr(i + 1, 2).c_str(), nullptr, 16));
                i += 2;
            } else rel.push_back(uri[i]);
        }
        const std::string full = (!rel.empty() && rel[0] == '/') ? rel : dir + "/" + rel;
        std::ifstream bf(full, std::ios::binary);
        if (!bf) return set_error(PT_ERR_IO, "cannot open " + full);
        out_bytes.assign((std::istreambuf_iterator<char>(bf)), std::istreambuf_iterator<char>());
        return PT_OK;
    };
    const Json *jbuf = root.get("buffers");
    std::vector<std::vector<uint8_t>> ext_buffers(jbuf ? jbuf->size() : 0);
    std::vector<std::pair<const uint8_t *, size_t>> buffers(jbuf ? jbuf->size() : 0, std::make_pair((const uint8_t *)nullptr, (size_t)0));
    for (size_t i = 0; jbuf && i < jbuf->size(); i++) {
        const Json *uri = jbuf->at(i).get("uri");
        if (!uri) {
            if (i == 0 && bin_ptr) buffers[i] = std::make_pair(bin_ptr, bin_len);
            continue;
        }
        bool is_data = false;
        pt_status st = load_uri(uri->str, ext_buffers[i], is_data);
        if (st != PT_OK) return st;
        buffers[i] = std::make_pair(ext_buffers[i].data(), ext_buffers[i].size());
    }
    if (buffers.empty() && bin_ptr) buffers.push_back(std::make_pair(bin_ptr, bin_len));

    const Json *jviews = root.get("bufferViews");
    const Json *jacc = root.get("accessors");
    auto view_span = [&](int view, const uint8_t *&ptr, size_t &len, size_t &stride) -> bool {
        if (!jviews || view < 0 || (size_t)view >= jviews->size()) return false;
        const Json &v = jviews->at((size_t)view);
        int buffer = v.get("buffer") ? v.get("buffer")->integer(0) : 0;
        if (buffer < 0 || (size_t)buffer >= buffers.size() || !buffers[(size_t)buffer].first) return false;
        size_t off;
        if (!json_size(v.get("byteOffset"), 0, off) || !json_size(v.get("byteLength"), 0, len) || !json_size(v.get("byteStride"), 0, stride)) return false;
        if (!span_fits(off, 1, 0, len, buffers[(size_t)buffer].second)) return false;
        ptr = buffers[(size_t)buffer].first + off;
        return true;
    };
    auto get_accessor = [&](int idx, Accessor &a) -> bool {
        if (!jacc || idx < 0 || (size_t)idx >= jacc->size()) return false;
        const Json &j = jacc->at((size_t)idx);
        a.componentType = j.get("componentType") ? j.get("componentType")->integer(0) : 0;
        a.ncomp = j.get("type") ? type_ncomp(j.get("type")->str) : 0;
        if (!json_size(j.get("count"), 0, a.count)) return false;
        a.normalized = j.get("normalized") && j.get("normalized")->b;
        const size_t cs = comp_size(a.componentType);
        if (!cs || !a.ncomp) return false;
        const size_t elem = cs * (size_t)a.ncomp;
        const Json *bv = j.get("bufferView");
        const Json *sparse = j.get("sparse");
        if (!bv && !sparse) return false;
        if (bv) {
            const uint8_t *ptr; size_t len, stride;
            if (!view_span(bv->integer(-1), ptr, len, stride)) return false;
            size_t off;
            if (!json_size(j.get("byteOffset"), 0, off)) return false;
            a.stride = stride ? stride : elem;
            if (!span_fits(off, a.count, a.stride, elem, len)) return false;
            a.base = ptr + off;
        }
        if (!sparse) return true;
        // sparse accessor (glTF 2.0 section 3.6.2.3): the base view, or zeros, with `count` elements replaced
        if (a.count > ((size_t)1 << 28)) return false;
        auto dense = std::make_shared<std::vector<uint8_t>>(a.count * elem, (uint8_t)0);
        if (bv) for (size_t i = 0; i < a.count; i++) memcpy(dense->data() + i * elem, a.base + i * a.stride, elem);
        const Json *si = sparse->get("indices"), *sv = sparse->get("values");
        size_t n, ioff, voff;
        if (!json_size(sparse->get("count"), 0, n)) return false;
        if (!si || !sv || !si->get("bufferView") || !sv->get("bufferView")) return false;
        const uint8_t *ip, *vp; size_t il, vl, st;
        if (!view_span(si->get("bufferView")->integer(-1), ip, il, st) || !view_span(sv->get("bufferView")->integer(-1), vp, vl, st)) return false;
        if (!json_size(si->get("byteOffset"), 0, ioff) || !json_size(sv->get("byteOffset"), 0, voff)) return false;
        const int ict = si->get("componentType") ? si->get("componentType")->integer(0) : 0;
        const size_t ics = comp_size(ict);
        if (!(ict == 5121 || ict == 5123 || ict == 5125) || !span_fits(ioff, n, ics, ics, il) || !span_fits(voff, n, elem, elem, vl)) return false;
        for (size_t k = 0; k < n; k++) {
            uint32_t at = 0;
            memcpy(&at, ip + ioff + k * ics, ics);               // little-endian host
            if (at >= a.count) return false;
            memcpy(dense->data() + (size_t)at * elem, vp + voff + k * elem, elem);
        }
        a.owned = dense;
        a.base = dense->data();
        a.stride = elem;
        return true;
    };
    auto read_float = [&](const Accessor &a, size_t i, int c) -> float {
        const uint8_t *p = a.base + i * a.stride + (size_t)c * comp_size(a.componentType);
        switch (a.componentType) {
        case 5126: { float v; memcpy(&v, p, 4); return v; }
        case 5121: return a.normalized ? (float)p[0] / 255.0f : (float)p[0];
        case 5123: { uint16_t v; memcpy(&v, p, 2); return a.normalized ? (float)v / 65535.0f : (float)v; }
        case 5120: { int8_t v; memcpy(&v, p, 1); return a.normalized ? fmaxf((float)v / 127.0f, -1.0f) : (float)v; }
        case 5122: { int16_t v; memcpy(&v, p, 2); return a.normalized ? fmaxf((float)v / 32767.0f, -1.0f) : (float)v; }
        default: return 0.0f;
        }
    };
    auto read_index = [&](const Accessor &a, size_t i) -> uint32_t {
        const uint8_t *p = a.base + i * a.stride;
        switch (a.componentType) {
        case 5121: return p[0];
        case 5123: { uint16_t v; memcpy(&v, p, 2); return v; }
        case 5125: { uint32_t v; memcpy(&v, p, 4); return v; }
        default: return 0;
        }
    };

    // ---- textures: one HostTexture per distinct image referenced as baseColor / emissive,
    // in material order (HostScene.cpp:51-72 walks materials x texture types; duplicates of
    // one image are deduplicated here, which only changes texture indices).
    const Json *jmats = root.get("materials");
    const Json *jtex = root.get("textures");
    const Json *jimg = root.get("images");
    std::map<int, int> image_to_tex;
    auto texture_for = [&](const Json *texinfo, int &out_idx) -> pt_status {
        out_idx = -1;
        if (!texinfo) return PT_OK;
        int ti = texinfo->get("index") ? texinfo->get("index")->integer(-1) : -1;
        if (!jtex || ti < 0 || (size_t)ti >= jtex->size()) return PT_OK;
        const Json *src = jtex->at((size_t)ti).get("source");
        int img = src ? src->integer(-1) : -1;
        if (!jimg || img < 0 || (size_t)img >= jimg->size()) return PT_OK;
        auto it = image_to_tex.find(img);
        if (it != image_to_tex.end()) { out_idx = it->second; return PT_OK; }
        const Json &ji = jimg->at((size_t)img);
        const Json *bv = ji.get("bufferView");
        const uint8_t *ptr = nullptr; size_t len = 0, stride = 0;
        std::vector<uint8_t> image_file;
        // Images in a bufferView or a data: URI reach the reference as assimp "embedded textures" ->
        // stbi_load_from_memory(req_comp = 0) + the 3-bytes-per-pixel copy loop (HostScene.cpp:18-26,36-46).
        // Images in a file next to the .gltf take the other branch, stbi_load(path, req_comp = 3) (:28-30):
        // stb_image converts them to RGB itself (grey replicated, alpha dropped).
        bool external_file = false;
        if (bv) {
            if (!view_span(bv->integer(-1), ptr, len, stride)) return set_error(PT_ERR_IO, "bad image bufferView");
        } else if (ji.get("uri")) {
            bool is_data = false;
            pt_status st = load_uri(ji.get("uri")->str, image_file, is_data);
            if (st != PT_OK) return st;
            ptr = image_file.data(); len = image_file.size();
            external_file = !is_data;
        } else {
            return set_error(PT_ERR_IO, "image without bufferView or uri");
        }
        int w = 0, h = 0, ch = 0;
        std::vector<uint8_t> px;
        // stb_image tries PNG first and JPEG later (stb_image.h stbi__load_main); GLB images are one of the two
        const bool is_jpeg = len >= 3 && ptr[0] == 0xff && ptr[1] == 0xd8 && ptr[2] == 0xff;
        pt_status st = is_jpeg ? decode_jpeg(ptr, len, w, h, ch, px) : decode_png(ptr, len, w, h, ch, px);
        if (st != PT_OK) return st;
        if (external_file && ch != 3) {                          // stbi__convert_format to 3 channels
            std::vector<uint8_t> rgb((size_t)w * (size_t)h * 3);
            for (size_t i = 0; i < (size_t)w * (size_t)h; i++)
                for (int c = 0; c < 3; c++) rgb[3 * i + c] = px[i * (size_t)ch + (ch >= 3 ? (size_t)c : 0)];
            px.swap(rgb);
            ch = 3;
        }
        // HostScene.cpp:36-46: the copy loop assumes 3 bytes per pixel whatever the file has.
        if (ch < 3) return set_error(PT_ERR_UNSUPPORTED, "textures with fewer than 3 channels are read out of bounds by the reference (HostScene.cpp:38-46)");
        std::vector<float> texels((size_t)w * (size_t)h * 3);
        size_t total_bytes = (size_t)w * (size_t)h * 3;
        for (size_t i = 0; i < total_bytes; i++) texels[i] = (float)px[i];
        out_idx = (int)scene.textures.size();
        image_to_tex[img] = out_idx;
        scene.texel_storage.push_back(std::move(texels));
        pt_texture t; t.width = w; t.height = h; t.data = nullptr;
        scene.textures.push_back(t);
        return PT_OK;
    };

    // ---- materials (HostScene.cpp:145-190)
    size_t nmat = jmats ? jmats->size() : 0;
    for (size_t i = 0; i < nmat; i++) {
        const Json &jm = jmats->at(i);
        pt_material m;
        m.baseColor[0] = m.baseColor[1] = m.baseColor[2] = 1.0f;
        m.emissiveFactor[0] = m.emissiveFactor[1] = m.emissiveFactor[2] = 0.0f;
        m.baseColorTextureIdx = -1; m.emissiveTextureIdx = -1;
        const Json *pbr = jm.get("pbrMetallicRoughness");
        if (pbr) {
            const Json *bc = pbr->get("baseColorFactor");
            if (bc && bc->size() >= 3) for (int k = 0; k < 3; k++) m.baseColor[k] = (float)bc->at((size_t)k).number(1.0);
        }
        const Json *ef = jm.get("emissiveFactor");
        if (ef && ef->size() >= 3) for (int k = 0; k < 3; k++) m.emissiveFactor[k] = (float)ef->at((size_t)k).number(0.0);
        // texture types are visited in aiTextureType order: EMISSIVE (4) before BASE_COLOR (12)
        pt_status st = texture_for(jm.get("emissiveTexture"), m.emissiveTextureIdx);
        if (st != PT_OK) return st;
        st = texture_for(pbr ? pbr->get("baseColorTexture") : nullptr, m.baseColorTextureIdx);
        if (st != PT_OK) return st;
        scene.materials.push_back(m);
        scene.material_names.push_back(jm.get("name") ? jm.get("name")->str : std::string());
    }

    // ---- node hierarchy -> (world matrix, mesh) instances in traversal order
    const Json *jnodes = root.get("nodes");
    const Json *jmeshes = root.get("meshes");
    const Json *jscenes = root.get("scenes");
    int scene_idx = root.get("scene") ? root.get("scene")->integer(0) : 0;
    struct Instance { M4 world; int mesh; };
    std::vector<Instance> instances;
    auto local_matrix = [&](const Json &n) -> M4 {
        M4 m = m4_identity();
        const Json *jm = n.get("matrix");
        if (jm && jm->size() == 16) {
            for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) m.m[r][c] = (float)jm->at((size_t)(c * 4 + r)).number(0.0);
            return m;
        }
        const Json *t = n.get("translation"), *q = n.get("rotation"), *s = n.get("scale");
        if (t && t->size() == 3) {
            M4 tm = m4_identity();
            for (int k = 0; k < 3; k++) tm.m[k][3] = (float)t->at((size_t)k).number(0.0);
            m = m4_mul(m, tm);
        }
        if (q && q->size() == 4) {
            float x = (float)q->at(0).number(0), y = (float)q->at(1).number(0), z = (float)q->at(2).number(0), w = (float)q->at(3).number(1);
            M4 rm = m4_identity();
            rm.m[0][0] = 1.0f - 2.0f * (y * y + z * z); rm.m[0][1] = 2.0f * (x * y - z * w); rm.m[0][2] = 2.0f * (x * z + y * w);
            rm.m[1][0] = 2.0f * (x * y + z * w); rm.m[1][1] = 1.0f - 2.0f * (x * x + z * z); rm.m[1][2] = 2.0f * (y * z - x * w);
            rm.m[2][0] = 2.0f * (x * z - y * w); rm.m[2][1] = 2.0f * (y * z + x * w); rm.m[2][2] = 1.0f - 2.0f * (x * x + y * y);
            m = m4_mul(m, rm);
        }
        if (s && s->size() == 3) {
            M4 sm = m4_identity();
            for (int k = 0; k < 3; k++) sm.m[k][k] = (float)s->at((size_t)k).number(1.0);
            m = m4_mul(m, sm);
        }
        return m;
    };
    // glTF 2.0 section 3.5.2: the node hierarchy is a set of strict trees (no node has two parents), so a walk never
    // visits more nodes than the file holds; this also stops cycles and DAGs that instantiate exponentially
    bool node_err = false;
    size_t visits = 0;
    auto walk = [&](auto &&self, int node, const M4 &parent, int depth) -> void {
        if (node_err) return;
        if (!jnodes || node < 0 || (size_t)node >= jnodes->size() || depth > 256 || ++visits > jnodes->size()) { node_err = true; return; }
        const Json &n = jnodes->at((size_t)node);
        M4 world = m4_mul(parent, local_matrix(n));
        if (n.get("mesh")) instances.push_back({world, n.get("mesh")->integer(-1)});
        const Json *ch = n.get("children");
        if (ch) for (size_t i = 0; i < ch->size(); i++) self(self, ch->at(i).integer(-1), world, depth + 1);
    };
    if (jscenes && (size_t)scene_idx < jscenes->size()) {
        const Json *sn = jscenes->at((size_t)scene_idx).get("nodes");
        if (sn) for (size_t i = 0; i < sn->size(); i++) walk(walk, sn->at(i).integer(-1), m4_identity(), 0);
    }
    if (node_err) return set_error(PT_ERR_IO, "bad node index, cycle or shared node in the glTF scene graph");

    // ---- triangles, grouped by material index ascending (PreTransformVertices), then by
    // instance order, then by primitive order.
    bool uses_default_material = false;
    std::vector<std::vector<pt_triangle>> by_material(nmat + 1);
    for (const Instance &inst : instances) {
        if (!jmeshes || inst.mesh < 0 || (size_t)inst.mesh >= jmeshes->size()) return set_error(PT_ERR_IO, "bad mesh index");
        const Json *prims = jmeshes->at((size_t)inst.mesh).get("primitives");
        if (!prims) continue;
        bool flip = m4_det3(inst.world) < 0.0f;
        for (size_t pi = 0; pi < prims->size(); pi++) {
            const Json &pr = prims->at(pi);
            int mode = pr.get("mode") ? pr.get("mode")->integer(4) : 4;
            if (mode <= 3) continue;          // points / lines: removed by SortByPType (HostScene.cpp:99)
            if (mode > 6) return set_error(PT_ERR_IO, "bad primitive mode");
            const Json *attrs = pr.get("attributes");
            if (!attrs || !attrs->get("POSITION")) continue;
            Accessor apos, auv, aidx;
            if (!get_accessor(attrs->get("POSITION")->integer(-1), apos) || apos.ncomp != 3)
                return set_error(PT_ERR_IO, "bad POSITION accessor");
            bool has_uv = attrs->get("TEXCOORD_0") && get_accessor(attrs->get("TEXCOORD_0")->integer(-1), auv) && auv.ncomp == 2;
            bool has_idx = pr.get("indices") != nullptr;
            if (has_idx && !get_accessor(pr.get("indices")->integer(-1), aidx)) return set_error(PT_ERR_IO, "bad index accessor");
            int mat = pr.get("material") ? pr.get("material")->integer(-1) : -1;
            if (mat < 0 || (size_t)mat >= nmat) { mat = (int)nmat; uses_default_material = true; }
            std::vector<pt_vertex> verts(apos.count);
            for (size_t v = 0; v < apos.count; v++) {
                float x = read_float(apos, v, 0), y = read_float(apos, v, 1), z = read_float(apos, v, 2);
                const M4 &w = inst.world;
                verts[v].position[0] = w.m[0][0] * x + w.m[0][1] * y + w.m[0][2] * z + w.m[0][3];
                verts[v].position[1] = w.m[1][0] * x + w.m[1][1] * y + w.m[1][2] * z + w.m[1][3];
                verts[v].position[2] = w.m[2][0] * x + w.m[2][1] * y + w.m[2][2] * z + w.m[2][3];
                if (has_uv && v < auv.count) {
                    verts[v].texCoords[0] = read_float(auv, v, 0);
                    verts[v].texCoords[1] = 1.0f - read_float(auv, v, 1);      // glTF2 importer V flip
                } else {
                    verts[v].texCoords[0] = verts[v].texCoords[1] = 0.0f;      // HostScene.cpp:214-216
                }
            }
            size_t nidx = has_idx ? aidx.count : apos.count;
            auto index_at = [&](size_t k) -> uint32_t { return has_idx ? read_index(aidx, k) : (uint32_t)k; };
            // TRIANGLES (4): consecutive triples.  TRIANGLE_STRIP (5) / TRIANGLE_FAN (6) are unrolled the way
            // assimp's glTF2 importer does (recalled from assimp 5.4, not verifiable here):
            // strip triangle n = (n, n+1, n+2), odd n as (n+1, n, n+2) to keep the orientation; fan triangle
            // n = (first, last vertex of triangle n-1, n+2).
            const size_t nfaces = mode == 4 ? nidx / 3 : (nidx >= 3 ? nidx - 2 : 0);
            for (size_t t = 0; t < nfaces; t++) {
                uint32_t i0, i1, i2;
                if (mode == 4) { i0 = index_at(3 * t); i1 = index_at(3 * t + 1); i2 = index_at(3 * t + 2); }
                else if (mode == 5) {
                    if (t & 1) { i0 = index_at(t + 1); i1 = index_at(t); i2 = index_at(t + 2); }
                    else { i0 = index_at(t); i1 = index_at(t + 1); i2 = index_at(t + 2); }
                } else { i0 = index_at(0); i1 = index_at(t + 1); i2 = index_at(t + 2); }
                if (i0 >= apos.count || i1 >= apos.count || i2 >= apos.count) return set_error(PT_ERR_IO, "vertex index out of range");
                if (flip) std::swap(i0, i2);
                pt_triangle tri;
                tri.v0 = verts[i0]; tri.v1 = verts[i1]; tri.v2 = verts[i2];
                tri.textureIdx = 0;
                tri.materialIdx = mat;
                auto same = [](const pt_vertex &a, const pt_vertex &b) {
                    return a.position[0] == b.position[0] && a.position[1] == b.position[1] && a.position[2] == b.position[2];
                };
                if (same(tri.v0, tri.v1) || same(tri.v1, tri.v2) || same(tri.v0, tri.v2)) continue;   // FindDegenerates + SortByPType
                by_material[(size_t)mat].push_back(tri);
            }
        }
    }
    if (uses_default_material) {
        pt_material m;
        m.baseColor[0] = m.baseColor[1] = m.baseColor[2] = 1.0f;
        m.emissiveFactor[0] = m.emissiveFactor[1] = m.emissiveFactor[2] = 0.0f;
        m.baseColorTextureIdx = m.emissiveTextureIdx = -1;
        scene.materials.push_back(m);
        scene.material_names.push_back("DefaultMaterial");
    }
    for (auto &group : by_material) scene.triangles.insert(scene.triangles.end(), group.begin(), group.end());
    if (scene.triangles.empty()) return set_error(PT_ERR_IO, "GLB contains no triangles");
    scene.rebuild_texture_views();
    return PT_OK;
}

// ------------------------------------------------------------------- OBJ -> HostScene
// Minimal Wavefront reader: v / vt / f (fan-triangulated) / usemtl / mtllib (newmtl, Ka, Kd,
// Ke).  Material mapping onto the live UniversalMaterial follows the name-prefix rule of
// obj_loader.h:65-96 / README.md:60-76: lambertian* -> baseColor = Ka; diffuse_light* ->
// emissiveFactor = Kd; anything else -> baseColor = Kd, emissiveFactor = Ke.
static pt_status load_obj_impl(const char *path, HostSceneImpl &scene)
{
    std::ifstream f(path);
    if (!f) return set_error(PT_ERR_IO, std::string("cannot open ") + path);
    struct Mtl { float ka[3] = {1, 1, 1}, kd[3] = {1, 1, 1}, ke[3] = {0, 0, 0}; float ns = 0.0f, ni = 1.0f; };
    std::map<std::string, Mtl> mtls;
    std::map<std::string, int> mat_index;
    std::vector<float> pos, uv;
    int cur_mat = -1;
    auto material_for = [&](const std::string &name) -> int {
        auto it = mat_index.find(name);
        if (it != mat_index.end()) return it->second;
        Mtl m = mtls.count(name) ? mtls[name] : Mtl();
        pt_material pm;
        pm.baseColorTextureIdx = pm.emissiveTextureIdx = -1;
        for (int k = 0; k < 3; k++) { pm.baseColor[k] = m.kd[k]; pm.emissiveFactor[k] = m.ke[k]; }
        if (name.rfind("lambertian", 0) == 0) for (int k = 0; k < 3; k++) pm.baseColor[k] = m.ka[k];
        if (name.rfind("diffuse_light", 0) == 0) for (int k = 0; k < 3; k++) { pm.emissiveFactor[k] = m.kd[k]; pm.baseColor[k] = 0.0f; }
        // the legacy class obj_loader.h:65-96 picks by name prefix: lambertian(Ka), metal(Ka, shininess Ns),
        // dielectric(Ni), diffuse_light(Kd) -- offered to pt_set_legacy_scene, ignored by the live path
        pt_legacy_material lm;
        memset(&lm, 0, sizeof lm);
        if (name.rfind("lambertian", 0) == 0) { lm.kind = 1; for (int k = 0; k < 3; k++) lm.albedo[k] = m.ka[k]; }
        else if (name.rfind("metal", 0) == 0) { lm.kind = 2; for (int k = 0; k < 3; k++) lm.albedo[k] = m.ka[k]; lm.param = m.ns; }
        else if (name.rfind("dielectric", 0) == 0) { lm.kind = 3; lm.param = m.ni; }
        else if (name.rfind("diffuse_light", 0) == 0) { lm.kind = 4; for (int k = 0; k < 3; k++) lm.albedo[k] = m.kd[k]; }
        int idx = (int)scene.materials.size();
        scene.materials.push_back(pm);
        scene.material_names.push_back(name);
        scene.legacy.resize(scene.materials.size());
        scene.legacy[(size_t)idx] = lm;
        mat_index[name] = idx;
        return idx;
    };
    std::string dir(path);
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    std::string line;
    while (std::getline(f, line)) {
        if (line.size() < 2) continue;
        if (line[0] == 'v' && line[1] == ' ') {
            float x, y, z;
            if (sscanf(line.c_str() + 2, "%f %f %f", &x, &y, &z) == 3) { pos.push_back(x); pos.push_back(y); pos.push_back(z); }
        } else if (line[0] == 'v' && line[1] == 't') {
            float u = 0, v = 0;
            sscanf(line.c_str() + 3, "%f %f", &u, &v);
            uv.push_back(u); uv.push_back(v);
        } else if (line[0] == 'f' && line[1] == ' ') {
            std::vector<pt_vertex> poly;
            const char *p = line.c_str() + 2;
            while (*p) {
                while (*p == ' ' || *p == '\t' || *p == '\r') p++;
                if (!*p) break;
                char *e;
                long vi = strtol(p, &e, 10), ti = 0;
                if (e == p) break;
                p = e;
                if (*p == '/') {
                    p++;
                    if (*p != '/') { ti = strtol(p, &e, 10); p = e; }
                    if (*p == '/') { p++; strtol(p, &e, 10); p = e; }
                }
                long nv = (long)pos.size() / 3, nt = (long)uv.size() / 2;
                if (vi < 0) vi = nv + 1 + vi;
                if (ti < 0) ti = nt + 1 + ti;
                if (vi < 1 || vi > nv) return set_error(PT_ERR_IO, "OBJ vertex index out of range");
                pt_vertex v;
                for (int k = 0; k < 3; k++) v.position[k] = pos[(size_t)(vi - 1) * 3 + (size_t)k];
                v.texCoords[0] = (ti >= 1 && ti <= nt) ? uv[(size_t)(ti - 1) * 2] : 0.0f;
                v.texCoords[1] = (ti >= 1 && ti <= nt) ? uv[(size_t)(ti - 1) * 2 + 1] : 0.0f;
                poly.push_back(v);
            }
            if (cur_mat < 0) cur_mat = material_for("default");
            for (size_t k = 1; k + 1 < poly.size(); k++) {
                pt_triangle t; t.v0 = poly[0]; t.v1 = poly[k]; t.v2 = poly[k + 1];
                t.textureIdx = 0; t.materialIdx = cur_mat;
                scene.triangles.push_back(t);
            }
        } else if (line.compare(0, 7, "usemtl ") == 0) {
            std::string name = line.substr(7);
            while (!name.empty() && (name.back() == '\r' || name.back() == ' ')) name.pop_back();
            cur_mat = material_for(name);
        } else if (line.compare(0, 7, "mtllib ") == 0) {
            std::string name = line.substr(7);
            while (!name.empty() && (name.back() == '\r' || name.back() == ' ')) name.pop_back();
            std::ifstream mf(dir + "/" + name);
            std::string ml, curname;
            while (std::getline(mf, ml)) {
                if (ml.compare(0, 7, "newmtl ") == 0) {
                    curname = ml.substr(7);
                    while (!curname.empty() && (curname.back() == '\r' || curname.back() == ' ')) curname.pop_back();
                    mtls[curname] = Mtl();
                } else if (!curname.empty() && ml.size() > 3 && ml[0] == 'N' && (ml[1] == 's' || ml[1] == 'i') && ml[2] == ' ') {
                    float v;
                    if (sscanf(ml.c_str() + 3, "%f", &v) == 1) (ml[1] == 's' ? mtls[curname].ns : mtls[curname].ni) = v;
                } else if (!curname.empty() && ml.size() > 3 && ml[0] == 'K') {
                    float a, b, c;
                    if (sscanf(ml.c_str() + 3, "%f %f %f", &a, &b, &c) == 3) {
                        float *dst = ml[1] == 'a' ? mtls[curname].ka : (ml[1] == 'd' ? mtls[curname].kd : (ml[1] == 'e' ? mtls[curname].ke : nullptr));
                        if (dst) { dst[0] = a; dst[1] = b; dst[2] = c; }
                    }
                }
            }
        }
    }
    if (scene.triangles.empty()) return set_error(PT_ERR_IO, "OBJ contains no faces");
    return PT_OK;
}

} // namespace pt

// =================================================================== C ABI (host) ====
using namespace pt;

struct pt_host_scene { HostSceneImpl impl; };

extern "C" {

pt_status pt_scene_load_glb(const char *path, pt_host_scene **out)
try {
    clear_error();
    if (!path || !out) return set_error(PT_ERR_INVALID, "null argument");
    std::unique_ptr<pt_host_scene> s(new pt_host_scene());
    pt_status st = load_glb_impl(path, s->impl);
    if (st != PT_OK) return st;
    *out = s.release();
    return PT_OK;
} PT_CATCH_ALL

/* Legacy-class records derived from the material names of a loaded scene (OBJ + MTL only; all kind 0 for
 * glTF input).  Returns the number of materials; copies up to `cap` records. */
size_t pt_scene_legacy_materials(const pt_host_scene *scene, pt_legacy_material *out, size_t cap)
{
    if (!scene) return 0;
    const HostSceneImpl &s = scene->impl;
    for (size_t i = 0; out && i < s.materials.size() && i < cap; i++) {
        if (i < s.legacy.size()) out[i] = s.legacy[i];
        else memset(&out[i], 0, sizeof out[i]);
    }
    return s.materials.size();
}

pt_status pt_scene_load_obj(const char *path, pt_host_scene **out)
try {
    clear_error();
    if (!path || !out) return set_error(PT_ERR_INVALID, "null argument");
    std::unique_ptr<pt_host_scene> s(new pt_host_scene());
    pt_status st = load_obj_impl(path, s->impl);
    if (st != PT_OK) return st;
    *out = s.release();
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_scene_create(const pt_triangle *triangles, size_t n_triangles,
                          const pt_material *materials, size_t n_materials,
                          const pt_texture *textures, size_t n_textures, pt_host_scene **out)
try {
    clear_error();
    if (!out || (!triangles && n_triangles) || (!materials && n_materials) || (!textures && n_textures))
        return set_error(PT_ERR_INVALID, "null argument");
    std::unique_ptr<pt_host_scene> s(new pt_host_scene());
    s->impl.triangles.assign(triangles, triangles + n_triangles);
    s->impl.materials.assign(materials, materials + n_materials);
    s->impl.material_names.assign(n_materials, std::string());
    for (size_t i = 0; i < n_textures; i++) {
        if (textures[i].width < 0 || textures[i].height < 0 || (!textures[i].data && textures[i].width * textures[i].height > 0))
            return set_error(PT_ERR_INVALID, "bad texture");
        size_t n = (size_t)textures[i].width * (size_t)textures[i].height * 3;
        s->impl.texel_storage.emplace_back(textures[i].data, textures[i].data + n);
        pt_texture t = textures[i]; t.data = nullptr;
        s->impl.textures.push_back(t);
    }
    s->impl.rebuild_texture_views();
    *out = s.release();
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_triangles_repack(const void *triangles, size_t n, const pt_triangle_layout *L, pt_triangle *out)
try {
    clear_error();
    if (!L || (n && (!triangles || !out))) return set_error(PT_ERR_INVALID, "null argument");
    auto fits = [&](size_t off, size_t bytes) { return off <= L->stride && bytes <= L->stride - off; };
    bool ok = L->stride >= 4 && fits(L->textureIdx, 4) && fits(L->materialIdx, 4);
    for (int v = 0; v < 3; v++) ok = ok && fits(L->position[v], 12) && fits(L->texCoords[v], 8);
    if (!ok) return set_error(PT_ERR_INVALID, "triangle layout: a field lies outside the record stride");
    const unsigned char *base = static_cast<const unsigned char *>(triangles);
    for (size_t i = 0; i < n; i++) {
        const unsigned char *rec = base + i * L->stride;
        pt_vertex *vs[3] = {&out[i].v0, &out[i].v1, &out[i].v2};
        for (int v = 0; v < 3; v++) {                    // memcpy: the caller's record need not be aligned for float
            std::memcpy(vs[v]->position, rec + L->position[v], 12);
            std::memcpy(vs[v]->texCoords, rec + L->texCoords[v], 8);
        }
        std::memcpy(&out[i].textureIdx, rec + L->textureIdx, 4);
        std::memcpy(&out[i].materialIdx, rec + L->materialIdx, 4);
    }
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_scene_create_strided(const void *triangles, size_t n_triangles, const pt_triangle_layout *layout,
                                  const pt_material *materials, size_t n_materials,
                                  const pt_texture *textures, size_t n_textures, pt_host_scene **out)
try {
    std::vector<pt_triangle> packed(n_triangles);
    pt_status st = pt_triangles_repack(triangles, n_triangles, layout, packed.data());
    if (st != PT_OK) return st;
    return pt_scene_create(packed.data(), n_triangles, materials, n_materials, textures, n_textures, out);
} PT_CATCH_ALL

void pt_scene_free(pt_host_scene *scene) { delete scene; }
size_t pt_scene_triangle_count(const pt_host_scene *s) { return s ? s->impl.triangles.size() : 0; }
size_t pt_scene_material_count(const pt_host_scene *s) { return s ? s->impl.materials.size() : 0; }
size_t pt_scene_texture_count(const pt_host_scene *s) { return s ? s->impl.textures.size() : 0; }
const pt_triangle *pt_scene_triangles(const pt_host_scene *s) { return s ? s->impl.triangles.data() : nullptr; }
const pt_material *pt_scene_materials(const pt_host_scene *s) { return s ? s->impl.materials.data() : nullptr; }
const pt_texture *pt_scene_textures(const pt_host_scene *s) { return s ? s->impl.textures.data() : nullptr; }
const char *pt_scene_material_name(const pt_host_scene *s, size_t i)
{
    return (s && i < s->impl.material_names.size()) ? s->impl.material_names[i].c_str() : "";
}

pt_status pt_scene_apply_cornell_convention(pt_host_scene *scene)
try {
    clear_error();
    if (!scene) return set_error(PT_ERR_INVALID, "null scene");
    HostSceneImpl &s = scene->impl;
    for (size_t i = 0; i < s.materials.size(); i++) {
        pt_material &m = s.materials[i];
        bool zero = m.emissiveFactor[0] == 0.0f && m.emissiveFactor[1] == 0.0f && m.emissiveFactor[2] == 0.0f;
        if (zero && s.material_names[i].rfind("diffuse_light", 0) == 0)
            m.emissiveFactor[0] = m.emissiveFactor[1] = m.emissiveFactor[2] = 1.0f;
    }
    auto swap_axes = [](pt_vertex &v) {
        float y = v.position[1], z = v.position[2];
        v.position[1] = z;
        v.position[2] = -y;
    };
    for (pt_triangle &t : s.triangles) { swap_axes(t.v0); swap_axes(t.v1); swap_axes(t.v2); }
    return PT_OK;
} PT_CATCH_ALL

// Deterministic terrain: grid x grid cells over [-500,500]^2 (y up), two triangles per cell,
// height = 40*sin*cos + integer-hash noise; one grey material; a 2-triangle emissive quad
// above the centre (BASELINE.json config 4).
pt_status pt_scene_make_terrain(unsigned grid, pt_host_scene **out)
try {
    clear_error();
    if (!out || grid == 0 || grid > 8192) return set_error(PT_ERR_INVALID, "grid must be in 1..8192");
    std::unique_ptr<pt_host_scene> s(new pt_host_scene());
    HostSceneImpl &sc = s->impl;
    pt_material grey; grey.baseColor[0] = grey.baseColor[1] = grey.baseColor[2] = 0.6f;
    grey.emissiveFactor[0] = grey.emissiveFactor[1] = grey.emissiveFactor[2] = 0.0f;
    grey.baseColorTextureIdx = grey.emissiveTextureIdx = -1;
    pt_material light = grey;
    light.baseColor[0] = light.baseColor[1] = light.baseColor[2] = 0.0f;
    // emitted radiance = emissiveFactor * 50 (material.h:85): 0.25 keeps most of the lit terrain inside 0..255
    light.emissiveFactor[0] = light.emissiveFactor[1] = light.emissiveFactor[2] = 0.25f;
    sc.materials = {grey, light};
    sc.material_names = {"lambertian_terrain", "diffuse_light"};
    auto hash = [](uint32_t x, uint32_t z) {
        uint32_t h = x * 0x9E3779B1u ^ (z * 0x85EBCA77u + 1u);
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        return h;
    };
    auto height = [&](unsigned ix, unsigned iz) -> float {
        double fx = (double)ix / (double)grid, fz = (double)iz / (double)grid;
        double base = 40.0 * sin(fx * 12.566370614359172) * cos(fz * 9.42477796076938);
        double noise = ((double)(hash(ix, iz) & 0xFFFFu) / 65535.0 - 0.5) * (600.0 / (double)grid);
        return (float)(base + noise);
    };
    sc.triangles.reserve((size_t)grid * grid * 2 + 2);
    float step = 1000.0f / (float)grid;
    std::vector<float> row0(grid + 1), row1(grid + 1);
    for (unsigned ix = 0; ix <= grid; ix++) row0[ix] = height(ix, 0);
    for (unsigned iz = 0; iz < grid; iz++) {
        for (unsigned ix = 0; ix <= grid; ix++) row1[ix] = height(ix, iz + 1);
        for (unsigned ix = 0; ix < grid; ix++) {
            float x0 = -500.0f + step * (float)ix, x1 = -500.0f + step * (float)(ix + 1);
            float z0 = -500.0f + step * (float)iz, z1 = -500.0f + step * (float)(iz + 1);
            pt_vertex a = {{x0, row0[ix], z0}, {0, 0}}, b = {{x1, row0[ix + 1], z0}, {0, 0}};
            pt_vertex c = {{x1, row1[ix + 1], z1}, {0, 0}}, d = {{x0, row1[ix], z1}, {0, 0}};
            // counter-clockwise seen from +y => geometric normal points up
            pt_triangle t0 = {a, d, c, 0, 0}, t1 = {a, c, b, 0, 0};
            sc.triangles.push_back(t0);
            sc.triangles.push_back(t1);
        }
        row0.swap(row1);
    }
    // emissive quad 200 x 200 at y = 300, facing down
    pt_vertex la = {{-100, 300, -100}, {0, 0}}, lb = {{100, 300, -100}, {0, 0}};
    pt_vertex lc = {{100, 300, 100}, {0, 0}}, ld = {{-100, 300, 100}, {0, 0}};
    pt_triangle l0 = {la, lb, lc, 0, 1}, l1 = {la, lc, ld, 0, 1};
    sc.triangles.push_back(l0);
    sc.triangles.push_back(l1);
    *out = s.release();
    return PT_OK;
} PT_CATCH_ALL

pt_status pt_write_ppm(const char *path, int width, int height, const uint8_t *rgb)
try {
    clear_error();
    if (!path || !rgb || width <= 0 || height <= 0) return set_error(PT_ERR_INVALID, "bad argument");
    FILE *fp = fopen(path, "wb");
    if (!fp) return set_error(PT_ERR_IO, std::string("cannot write ") + path);
    fprintf(fp, "P6\n%d %d\n255\n", width, height);
    size_t n = (size_t)width * (size_t)height * 3;
    size_t w = fwrite(rgb, 1, n, fp);
    fclose(fp);
    if (w != n) return set_error(PT_ERR_IO, "short write");
    return PT_OK;
} PT_CATCH_ALL

} // extern "C"

namespace pt {
const HostSceneImpl &host_scene_impl(const pt_host_scene *s) { return s->impl; }
}
