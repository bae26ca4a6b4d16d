// pt_jpeg.cpp -- JPEG texture decoding for the GLB loader.
//
// The reference decodes embedded textures with its vendored third-party/stb_image.h
// (SceneLoader::loadTextureFromFile, HostScene.cpp:19-26, req_comp = 0), so "the same texels" means
// stb_image's output, not any JPEG decoder's: the entropy decoding is fixed by ITU-T T.81, but the
// inverse DCT, the chroma upsampling and the colour conversion are implementation choices.  This file
// follows T.81 for the bit stream and restates stb_image's arithmetic for those three steps
// (stb_image.h:2415-2527 integer IDCT after IJG's jidctint, :3465-3528 "3/4 + 1/4" upsampling,
// :3660-3684 20-bit fixed-point YCbCr -> RGB, :3949-3975 CMYK / YCCK), and is pinned byte-for-byte
// against the real header through oracle/_ref (tests/test_oracle_pins.py, tests/golden/stb_jpeg_vectors.npz).
// Supported like there: 8-bit baseline / extended-sequential / progressive Huffman JPEG, 1, 3 or 4
// components, sampling factors 1..4 with integer ratios, restart intervals, 8- and 16-bit DQT.
#include <string.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "pt_internal.h"

namespace pt {

namespace {

const uint8_t kZigzag[64 + 15] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    // a run that overshoots the block lands here instead of past the array (corrupt data, T.81 F.2.2.2)
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {
    uint8_t size[257];
    uint16_t code[256];
    uint8_t value[256];
    uint32_t maxcode[18];      // left-aligned to 16 bits, exclusive
    int delta[17];             // value index = code + delta[len]
    bool defined = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int x = 0, y = 0;          // sample extent that covers the image
    int w2 = 0, h2 = 0;        // allocated extent (whole MCUs)
    std::vector<uint8_t> data; // decoded samples, w2 x h2
    std::vector<int16_t> coef; // progressive: coefficients, 64 per block, (w2/8) x (h2/8) blocks
};

struct Decoder {
    const uint8_t *p = nullptr, *end = nullptr;
    // entropy-coded segment reader (T.81 F.2.2.5): MSB-first bit buffer, FF00 -> FF, stops at a marker
    uint32_t bits = 0;
    int nbits = 0;
    int marker = -1;           // marker met inside the entropy-coded data, -1 = none
    bool nomore = false;

    int W = 0, H = 0, ncomp = 0;
    bool progressive = false, jfif = false;
    int adobe_transform = -1, rgb_ids = 0;
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    int restart_interval = 0, todo = 0;
    Component comp[4];
    Huffman dc[4], ac[4];
    uint16_t dequant[4][64];
    // current scan
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0, eob_run = 0;
};

bool fail(const char *msg)
{
    set_error(PT_ERR_IO, msg);
    return false;
}

int get8(Decoder &d) { return d.p < d.end ? *d.p++ : 0; }
int get16(Decoder &d) { int a = get8(d); return (a << 8) | get8(d); }

bool build_huffman(Huffman &h, const int count[16])
{
    int k = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < count[i]; j++) {
            if (k >= 256) return fail("JPEG: bad Huffman size list");
            h.size[k++] = (uint8_t)(i + 1);
        }
    h.size[k] = 0;
    unsigned code = 0;
    k = 0;
    for (int len = 1; len <= 16; len++) {
        h.delta[len] = k - (int)code;
        if (h.size[k] == len) {
            while (h.size[k] == len) h.code[k++] = (uint16_t)(code++);
            if (code - 1 >= (1u << len)) return fail("JPEG: bad Huffman code lengths");
        }
        h.maxcode[len] = code << (16 - len);
        code <<= 1;
    }
    h.maxcode[17] = 0xffffffffu;
    h.defined = true;
    return true;
}

void fill_bits(Decoder &d)
{
    while (d.nbits <= 24) {
        int b = d.nomore ? 0 : get8(d);
        if (b == 0xff) {
            int c = get8(d);
            while (c == 0xff) c = get8(d);          // fill bytes
            if (c != 0) {                           // a marker ends the segment: feed zeros from here on
                d.marker = c;
                d.nomore = true;
                return;
            }
        }
        d.bits |= (uint32_t)b << (24 - d.nbits);
        d.nbits += 8;
    }
}

int huff_decode(Decoder &d, const Huffman &h)
{
    if (d.nbits < 16) fill_bits(d);
    const uint32_t top = d.bits >> 16;
    int len = 1;
    while (top >= h.maxcode[len]) len++;
    if (len == 17) { d.nbits -= 16; return -1; }
    if (len > d.nbits) return -1;
    const int idx = (int)((d.bits >> (32 - len)) & ((1u << len) - 1)) + h.delta[len];
    if (idx < 0 || idx >= 256) return -1;
    d.nbits -= len;
    d.bits <<= len;
    return h.value[idx];
}

int get_bits(Decoder &d, int n)
{
    if (n == 0) return 0;
    if (d.nbits < n) fill_bits(d);
    if (d.nbits < n) return 0;
    const int v = (int)(d.bits >> (32 - n));
    d.bits <<= n;
    d.nbits -= n;
    return v;
}

int get_bit(Decoder &d) { return get_bits(d, 1); }

// T.81 F.2.2.1 EXTEND
int extend_receive(Decoder &d, int n)
{
    if (n == 0) return 0;
    const int v = get_bits(d, n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
}

inline uint8_t clamp255(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// Inverse DCT, stb_image.h:2428-2527: IJG's "islow" integer algorithm, constants at 12 fractional bits,
// 2 extra bits kept between the passes, +128 level shift folded into the final rounding.
inline int fx(double x) { return (int)(x * 4096 + 0.5); }

// Intermediates are 64-bit and are cut back to 32 bits only where the reference shifts: same bits as its
// 32-bit arithmetic on any real image, and no signed overflow (undefined behaviour) on corrupt ones,
// where dequantised coefficients can be anything.
#define PT_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                       \
    long long t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                         \
    p2 = s2; p3 = s6;                                                                                     \
    p1 = (p2 + p3) * fx(0.5411961f);                                                                      \
    t2 = p1 + p3 * fx(-1.847759065f);                                                                     \
    t3 = p1 + p2 * fx(0.765366865f);                                                                      \
    p2 = s0; p3 = s4;                                                                                     \
    t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                                         \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                               \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                                   \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                               \
    p5 = (p3 + p4) * fx(1.175875602f);                                                                    \
    t0 = t0 * fx(0.298631336f); t1 = t1 * fx(2.053119869f);                                               \
    t2 = t2 * fx(3.072711026f); t3 = t3 * fx(1.501321110f);                                               \
    p1 = p5 + p1 * fx(-0.899976223f); p2 = p5 + p2 * fx(-2.562915447f);                                   \
    p3 = p3 * fx(-1.961570560f); p4 = p4 * fx(-0.390180644f);                                             \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idct_block(uint8_t *out, int stride, const int16_t in[64])
{
    int val[64];
    for (int i = 0; i < 8; i++) {
        const int16_t *c = in + i;
        int *v = val + i;
        PT_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
        x0 += 512; x1 += 512; x2 += 512; x3 += 512;
        v[0] = (int32_t)(x0 + t3) >> 10;  v[56] = (int32_t)(x0 - t3) >> 10;
        v[8] = (int32_t)(x1 + t2) >> 10;  v[48] = (int32_t)(x1 - t2) >> 10;
        v[16] = (int32_t)(x2 + t1) >> 10; v[40] = (int32_t)(x2 - t1) >> 10;
        v[24] = (int32_t)(x3 + t0) >> 10; v[32] = (int32_t)(x3 - t0) >> 10;
    }
    for (int i = 0; i < 8; i++) {
        const int *v = val + 8 * i;
        uint8_t *o = out + (size_t)i * (size_t)stride;
        PT_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        const long long bias = 65536 + (128 << 17);
        x0 += bias; x1 += bias; x2 += bias; x3 += bias;
        o[0] = clamp255((int32_t)(x0 + t3) >> 17); o[7] = clamp255((int32_t)(x0 - t3) >> 17);
        o[1] = clamp255((int32_t)(x1 + t2) >> 17); o[6] = clamp255((int32_t)(x1 - t2) >> 17);
        o[2] = clamp255((int32_t)(x2 + t1) >> 17); o[5] = clamp255((int32_t)(x2 - t1) >> 17);
        o[3] = clamp255((int32_t)(x3 + t0) >> 17); o[4] = clamp255((int32_t)(x3 - t0) >> 17);
    }
}

// DC bookkeeping with the reference's validity checks (stb_image.h "bad delta", "can't merge dc and ac"):
// the running predictor must not overflow and the scaled DC must fit the 16-bit coefficient.
bool next_dc(Component &c, int diff, int scale, int16_t &out)
{
    const long long dc = (long long)c.dc_pred + diff;
    if (dc > 2147483647ll || dc < -2147483647ll - 1) return fail("JPEG: DC predictor overflow");
    c.dc_pred = (int)dc;
    const long long v = dc * scale;
    if (v > 32767 || v < -32768) return fail("JPEG: DC coefficient out of range");
    out = (int16_t)v;
    return true;
}

// One block of a sequential scan (T.81 F.2.2): DC difference, AC run/size pairs; dequantised on the fly.
bool decode_block_sequential(Decoder &d, int16_t blk[64], Component &c)
{
    const Huffman &hdc = d.dc[c.td], &hac = d.ac[c.ta];
    const uint16_t *dq = d.dequant[c.tq];
    const int t = huff_decode(d, hdc);
    if (t < 0 || t > 15) return fail("JPEG: bad Huffman code");
    memset(blk, 0, 64 * sizeof(int16_t));
    if (!next_dc(c, t ? extend_receive(d, t) : 0, dq[0], blk[0])) return false;
    int k = 1;
    do {
        const int rs = huff_decode(d, hac);
        if (rs < 0) return fail("JPEG: bad Huffman code");
        const int s = rs & 15, r = rs >> 4;
        if (s == 0) {
            if (rs != 0xf0) break;              // EOB
            k += 16;
        } else {
            k += r;
            const int zig = kZigzag[k++];
            blk[zig] = (int16_t)(extend_receive(d, s) * dq[zig]);
        }
    } while (k < 64);
    return true;
}

// Progressive DC scan (T.81 G.1.2.1): first pass codes the difference of the point-transformed DC,
// refinement passes append one bit.
bool decode_block_prog_dc(Decoder &d, int16_t blk[64], Component &c)
{
    if (d.se != 0) return fail("JPEG: DC and AC coefficients in one progressive scan");
    if (d.ah == 0) {
        memset(blk, 0, 64 * sizeof(int16_t));
        const int t = huff_decode(d, d.dc[c.td]);
        if (t < 0 || t > 15) return fail("JPEG: bad Huffman code");
        if (!next_dc(c, t ? extend_receive(d, t) : 0, 1 << d.al, blk[0])) return false;
    } else if (get_bit(d)) {
        blk[0] = (int16_t)(blk[0] + (1 << d.al));
    }
    return true;
}

// Progressive AC scan (T.81 G.1.2.2 / G.1.2.3).
bool decode_block_prog_ac(Decoder &d, int16_t blk[64], Component &c)
{
    if (d.ss == 0) return fail("JPEG: DC and AC coefficients in one progressive scan");
    const Huffman &hac = d.ac[c.ta];
    if (d.ah == 0) {
        const int shift = d.al;
        if (d.eob_run) { d.eob_run--; return true; }
        int k = d.ss;
        do {
            const int rs = huff_decode(d, hac);
            if (rs < 0) return fail("JPEG: bad Huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    d.eob_run = 1 << r;
                    if (r) d.eob_run += get_bits(d, r);
                    d.eob_run--;
                    break;
                }
                k += 16;
            } else {
                k += r;
                const int zig = kZigzag[k++];
                blk[zig] = (int16_t)(extend_receive(d, s) * (1 << shift));
            }
        } while (k <= d.se);
        return true;
    }
    // refinement: one more bit for the coefficients that are already nonzero, new +-1 coefficients in between
    const int16_t bit = (int16_t)(1 << d.al);
    auto refine = [&](int16_t *p) {
        if (get_bit(d) && (*p & bit) == 0) *p = (int16_t)(*p > 0 ? *p + bit : *p - bit);
    };
    if (d.eob_run) {
        d.eob_run--;
        for (int k = d.ss; k <= d.se; k++) {
            int16_t *p = &blk[kZigzag[k]];
            if (*p != 0) refine(p);
        }
        return true;
    }
    int k = d.ss;
    do {
        const int rs = huff_decode(d, hac);
        if (rs < 0) return fail("JPEG: bad Huffman code");
        int s = rs & 15, r = rs >> 4;
        if (s == 0) {
            if (r < 15) {
                d.eob_run = (1 << r) - 1;
                if (r) d.eob_run += get_bits(d, r);
                r = 64;                         // run to the end of the band, refining on the way
            }                                   // else r = 15: sixteen zero-history coefficients (ZRL)
        } else {
            if (s != 1) return fail("JPEG: bad Huffman code");
            s = get_bit(d) ? bit : -bit;
        }
        while (k <= d.se) {
            int16_t *p = &blk[kZigzag[k++]];
            if (*p != 0) {
                refine(p);
            } else {
                if (r == 0) { *p = (int16_t)s; break; }
                r--;
            }
        }
    } while (k <= d.se);
    return true;
}

void reset_entropy(Decoder &d)
{
    d.bits = 0; d.nbits = 0; d.nomore = false; d.marker = -1;
    for (int i = 0; i < 4; i++) d.comp[i].dc_pred = 0;
    d.todo = d.restart_interval ? d.restart_interval : 0x7fffffff;
    d.eob_run = 0;
}

// After `restart_interval` MCUs the segment ends with an RSTn marker (T.81 E.2.4).  False = stop the scan.
bool after_unit(Decoder &d)
{
    if (--d.todo <= 0) {
        if (d.nbits < 24) fill_bits(d);
        if (!(d.marker >= 0xd0 && d.marker <= 0xd7)) return false;
        reset_entropy(d);
    }
    return true;
}

bool decode_scan(Decoder &d)
{
    reset_entropy(d);
    int16_t blk[64];
    auto one_block = [&](Component &c, int bx, int by) -> bool {
        if (!d.progressive) {
            if (!decode_block_sequential(d, blk, c)) return false;
            idct_block(c.data.data() + (size_t)c.w2 * (size_t)by * 8 + (size_t)bx * 8, c.w2, blk);
            return true;
        }
        int16_t *b = c.coef.data() + 64 * ((size_t)bx + (size_t)by * (size_t)(c.w2 / 8));
        return d.ss == 0 ? decode_block_prog_dc(d, b, c) : decode_block_prog_ac(d, b, c);
    };
    if (d.scan_n == 1) {
        // non-interleaved: only the blocks that cover the image (T.81 A.2.2)
        Component &c = d.comp[d.order[0]];
        const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                if (!one_block(c, i, j)) return false;
                if (!after_unit(d)) return true;
            }
        return true;
    }
    if (d.progressive && d.ss != 0) return fail("JPEG: interleaved progressive AC scan");
    for (int j = 0; j < d.mcuy; j++)
        for (int i = 0; i < d.mcux; i++) {
            for (int k = 0; k < d.scan_n; k++) {
                Component &c = d.comp[d.order[k]];
                for (int y = 0; y < c.v; y++)
                    for (int x = 0; x < c.h; x++)
                        if (!one_block(c, i * c.h + x, j * c.v + y)) return false;
            }
            if (!after_unit(d)) return true;
        }
    return true;
}

// Next marker: the one the entropy decoder ran into, or the next FFxx in the stream.
int next_marker(Decoder &d)
{
    if (d.marker >= 0) { int m = d.marker; d.marker = -1; return m; }
    int x = get8(d);
    if (x != 0xff) return -1;
    while (x == 0xff && d.p < d.end) x = get8(d);
    return x;
}

bool process_marker(Decoder &d, int m)
{
    switch (m) {
    case -1: return fail("JPEG: expected a marker");
    case 0xdd:
        if (get16(d) != 4) return fail("JPEG: bad DRI length");
        d.restart_interval = get16(d);
        return true;
    case 0xdb: {
        int L = get16(d) - 2;
        while (L > 0) {
            const int q = get8(d), p = q >> 4, t = q & 15;
            if (p > 1) return fail("JPEG: bad DQT type");
            if (t > 3) return fail("JPEG: bad DQT table");
            for (int i = 0; i < 64; i++) d.dequant[t][kZigzag[i]] = (uint16_t)(p ? get16(d) : get8(d));
            L -= p ? 129 : 65;
        }
        return L == 0 ? true : fail("JPEG: bad DQT length");
    }
    case 0xc4: {
        int L = get16(d) - 2;
        while (L > 0) {
            const int q = get8(d), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) return fail("JPEG: bad DHT header");
            int sizes[16], n = 0;
            for (int i = 0; i < 16; i++) { sizes[i] = get8(d); n += sizes[i]; }
            if (n > 256) return fail("JPEG: bad DHT header");
            L -= 17;
            Huffman &h = tc ? d.ac[th] : d.dc[th];
            if (!build_huffman(h, sizes)) return false;
            for (int i = 0; i < n; i++) h.value[i] = (uint8_t)get8(d);
            L -= n;
        }
        return L == 0 ? true : fail("JPEG: bad DHT length");
    }
    default: break;
    }
    if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {
        int L = get16(d);
        if (L < 2) return fail("JPEG: bad APP / COM length");
        L -= 2;
        if (m == 0xe0 && L >= 5) {
            static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
            bool ok = true;
            for (int i = 0; i < 5; i++) ok = (get8(d) == tag[i]) && ok;
            L -= 5;
            if (ok) d.jfif = true;
        } else if (m == 0xee && L >= 12) {
            static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
            bool ok = true;
            for (int i = 0; i < 6; i++) ok = (get8(d) == tag[i]) && ok;
            L -= 6;
            if (ok) {
                get8(d); get16(d); get16(d);
                d.adobe_transform = get8(d);
                L -= 6;
            }
        }
        if (d.p + L > d.end) return fail("JPEG: truncated segment");
        d.p += L;
        return true;
    }
    return fail("JPEG: unknown marker");
}

bool frame_header(Decoder &d)
{
    const int Lf = get16(d);
    if (Lf < 11) return fail("JPEG: bad SOF length");
    if (get8(d) != 8) return fail("JPEG: only 8-bit samples are supported");
    d.H = get16(d);
    d.W = get16(d);
    if (d.H == 0) return fail("JPEG: no height in the frame header");
    if (d.W == 0) return fail("JPEG: zero width");
    if (d.W > (1 << 24) || d.H > (1 << 24)) return fail("JPEG: image too large");
    const int c = get8(d);
    if (c != 1 && c != 3 && c != 4) return fail("JPEG: bad component count");
    d.ncomp = c;
    if (Lf != 8 + 3 * c) return fail("JPEG: bad SOF length");
    d.rgb_ids = 0;
    for (int i = 0; i < c; i++) {
        static const uint8_t rgb[3] = {'R', 'G', 'B'};
        Component &k = d.comp[i];
        k.id = get8(d);
        if (c == 3 && k.id == rgb[i]) d.rgb_ids++;
        const int q = get8(d);
        k.h = q >> 4; k.v = q & 15;
        if (k.h < 1 || k.h > 4) return fail("JPEG: bad horizontal sampling factor");
        if (k.v < 1 || k.v > 4) return fail("JPEG: bad vertical sampling factor");
        k.tq = get8(d);
        if (k.tq > 3) return fail("JPEG: bad quantisation table index");
    }
    // (stb_image refuses beyond 2^31 bytes of output; more than 2^28 pixels is a corrupt header, not a texture)
    if ((uint64_t)d.W * (uint64_t)d.H > ((uint64_t)1 << 28)) return fail("JPEG: image too large");
    d.hmax = d.vmax = 1;
    for (int i = 0; i < c; i++) { d.hmax = std::max(d.hmax, d.comp[i].h); d.vmax = std::max(d.vmax, d.comp[i].v); }
    for (int i = 0; i < c; i++)
        if (d.hmax % d.comp[i].h || d.vmax % d.comp[i].v) return fail("JPEG: fractional sampling ratios");
    d.mcux = (d.W + d.hmax * 8 - 1) / (d.hmax * 8);
    d.mcuy = (d.H + d.vmax * 8 - 1) / (d.vmax * 8);
    for (int i = 0; i < c; i++) {
        Component &k = d.comp[i];
        k.x = (d.W * k.h + d.hmax - 1) / d.hmax;
        k.y = (d.H * k.v + d.vmax - 1) / d.vmax;
        k.w2 = d.mcux * k.h * 8;
        k.h2 = d.mcuy * k.v * 8;
        k.data.assign((size_t)k.w2 * (size_t)k.h2, 0);
        if (d.progressive) k.coef.assign((size_t)k.w2 * (size_t)k.h2, 0);
    }
    return true;
}

bool scan_header(Decoder &d)
{
    const int Ls = get16(d);
    d.scan_n = get8(d);
    if (d.scan_n < 1 || d.scan_n > 4 || d.scan_n > d.ncomp) return fail("JPEG: bad SOS component count");
    if (Ls != 6 + 2 * d.scan_n) return fail("JPEG: bad SOS length");
    for (int i = 0; i < d.scan_n; i++) {
        const int id = get8(d), q = get8(d);
        int which = 0;
        while (which < d.ncomp && d.comp[which].id != id) which++;
        if (which == d.ncomp) return fail("JPEG: scan names an unknown component");
        d.comp[which].td = q >> 4; d.comp[which].ta = q & 15;
        if (d.comp[which].td > 3 || d.comp[which].ta > 3) return fail("JPEG: bad Huffman table index");
        d.order[i] = which;
    }
    d.ss = get8(d); d.se = get8(d);
    const int aa = get8(d);
    d.ah = aa >> 4; d.al = aa & 15;
    if (d.progressive) {
        if (d.ss > 63 || d.se > 63 || d.ss > d.se || d.ah > 13 || d.al > 13) return fail("JPEG: bad SOS");
    } else {
        if (d.ss != 0 || d.ah != 0 || d.al != 0) return fail("JPEG: bad SOS");
        d.se = 63;
    }
    return true;
}

// Progressive: all scans are in, dequantise and transform every block that covers the image (:3080-3097).
void finish_progressive(Decoder &d)
{
    for (int n = 0; n < d.ncomp; n++) {
        Component &c = d.comp[n];
        const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                int16_t *b = c.coef.data() + 64 * ((size_t)i + (size_t)j * (size_t)(c.w2 / 8));
                for (int k = 0; k < 64; k++) b[k] = (int16_t)(b[k] * d.dequant[c.tq][k]);
                idct_block(c.data.data() + (size_t)c.w2 * (size_t)j * 8 + (size_t)i * 8, c.w2, b);
            }
    }
}

// ---- upsampling of one output row (stb_image.h:3455-3528, :3646-3656) ---------------------------
// near / far = the two source rows this output row lies between (3/4 near + 1/4 far).
const uint8_t *resample_row(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near[0]; return out; }
        out[0] = near[0];
        out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            const int n = 3 * near[i] + 2;
            out[i * 2] = (uint8_t)((n + near[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n + near[i + 1]) >> 2);
        }
        out[i * 2] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
        out[i * 2 + 1] = near[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return out; }
        int t1 = 3 * near[0] + far[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++)                       // any other ratio: nearest sample
        for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    return out;
}

inline uint8_t blinn_8x8(int x, int y)
{
    const unsigned t = (unsigned)(x * y + 128);
    return (uint8_t)((t + (t >> 8)) >> 8);
}

} // namespace

pt_status decode_jpeg(const uint8_t *data, size_t len, int &w, int &h, int &channels, std::vector<uint8_t> &out)
{
    std::unique_ptr<Decoder> dp(new Decoder);
    Decoder &d = *dp;
    d.p = data; d.end = data + len;
    memset(d.dequant, 0, sizeof d.dequant);
    if (len < 4 || data[0] != 0xff || data[1] != 0xd8) return set_error(PT_ERR_UNSUPPORTED, "embedded image is not a JPEG");
    d.p += 2;
    // tables and miscellaneous segments up to the frame header
    int m = next_marker(d);
    while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
        if (m == 0xd9 || (m == -1 && d.p >= d.end)) return set_error(PT_ERR_IO, "JPEG: no frame header");
        if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)
            return set_error(PT_ERR_UNSUPPORTED, "JPEG: only baseline, extended-sequential and progressive Huffman coding are supported");
        if (!process_marker(d, m)) return PT_ERR_IO;
        m = next_marker(d);
        while (m == -1) {                                   // junk between segments
            if (d.p >= d.end) return set_error(PT_ERR_IO, "JPEG: no frame header");
            m = next_marker(d);
        }
    }
    d.progressive = m == 0xc2;
    if (!frame_header(d)) return PT_ERR_IO;
    // scans
    m = next_marker(d);
    while (m != 0xd9) {
        if (m == 0xda) {
            if (!scan_header(d)) return PT_ERR_IO;
            for (int i = 0; i < d.scan_n; i++) {
                const Component &c = d.comp[d.order[i]];
                const bool need_dc = d.ss == 0 && !(d.progressive && d.ah != 0), need_ac = !d.progressive || d.ss != 0;
                if ((need_dc && !d.dc[c.td].defined) || (need_ac && !d.ac[c.ta].defined))
                    return set_error(PT_ERR_IO, "JPEG: scan uses an undefined Huffman table");
            }
            if (!decode_scan(d)) return PT_ERR_IO;
            if (d.marker < 0) {
                // entropy data ended without a marker in sight: skip to the next one
                while (d.p < d.end) {
                    if (get8(d) != 0xff) continue;
                    int x = 0xff;
                    while (x == 0xff && d.p < d.end) x = get8(d);
                    if (x != 0 && x != 0xff) { d.marker = x; break; }
                }
            }
            m = next_marker(d);
            if (m >= 0xd0 && m <= 0xd7) m = next_marker(d);
        } else if (m == 0xdc) {
            const int Ld = get16(d), NL = get16(d);
            if (Ld != 4 || NL != d.H) return set_error(PT_ERR_IO, "JPEG: bad DNL segment");
            m = next_marker(d);
        } else {
            // stb_image stops at anything it cannot parse here and keeps what it has (:3436); that is well
            // defined for a sequential file whose scans are in (every block was written), while a progressive
            // file would skip its final dequantise + IDCT pass and return uninitialised memory -> refused
            if (!process_marker(d, m)) {
                if (d.progressive) return set_error(PT_ERR_IO, "JPEG: progressive stream ends without an end-of-image marker");
                clear_error();
                break;
            }
            m = next_marker(d);
        }
    }
    if (d.progressive) finish_progressive(d);

    // upsample row by row and convert (load_jpeg_image, stb_image.h:3865-4025, req_comp = 0)
    w = d.W; h = d.H;
    channels = d.ncomp >= 3 ? 3 : 1;
    const bool is_rgb = d.ncomp == 3 && (d.rgb_ids == 3 || (d.adobe_transform == 0 && !d.jfif));
    out.assign((size_t)w * (size_t)h * (size_t)channels, 0);
    struct Row { int hs, vs, ystep, ypos, w_lores; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } rs[4];
    for (int k = 0; k < d.ncomp; k++) {
        const Component &c = d.comp[k];
        rs[k].hs = d.hmax / c.h; rs[k].vs = d.vmax / c.v;
        rs[k].ystep = rs[k].vs >> 1;
        rs[k].w_lores = (w + rs[k].hs - 1) / rs[k].hs;
        rs[k].ypos = 0;
        rs[k].line0 = rs[k].line1 = c.data.data();
        rs[k].buf.assign((size_t)w + 3 + 8, 0);
    }
    const uint8_t *row[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int j = 0; j < h; j++) {
        uint8_t *o = out.data() + (size_t)channels * (size_t)w * (size_t)j;
        for (int k = 0; k < d.ncomp; k++) {
            Row &r = rs[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            row[k] = resample_row(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < d.comp[k].y) r.line1 += d.comp[k].w2;
            }
        }
        if (d.ncomp == 1) {
            memcpy(o, row[0], (size_t)w);
            continue;
        }
        const bool ycc = (d.ncomp == 3 && !is_rgb) || (d.ncomp == 4 && d.adobe_transform != 0);
        for (int i = 0; i < w; i++) {
            int r, g, b;
            if (ycc) {
                // 20-bit fixed point; the blue-difference term of green loses its low 16 bits (:3660-3684)
                const int yf = (row[0][i] << 20) + (1 << 19);
                const int cr = row[2][i] - 128, cb = row[1][i] - 128;
                r = yf + cr * (fx(1.40200f) << 8);
                g = yf + (cr * -(fx(0.71414f) << 8)) + (int)(((unsigned)(cb * -(fx(0.34414f) << 8))) & 0xffff0000u);
                b = yf + cb * (fx(1.77200f) << 8);
                r >>= 20; g >>= 20; b >>= 20;
                r = clamp255(r); g = clamp255(g); b = clamp255(b);
            } else {
                r = row[0][i]; g = row[1][i]; b = row[2][i];
            }
            if (d.ncomp == 4) {
                const int k4 = row[3][i];
                if (d.adobe_transform == 0) { r = blinn_8x8(r, k4); g = blinn_8x8(g, k4); b = blinn_8x8(b, k4); }                 // CMYK
                else if (d.adobe_transform == 2) { r = blinn_8x8(255 - r, k4); g = blinn_8x8(255 - g, k4); b = blinn_8x8(255 - b, k4); }  // YCCK
            }
            o[3 * i] = (uint8_t)r; o[3 * i + 1] = (uint8_t)g; o[3 * i + 2] = (uint8_t)b;
        }
    }
    return PT_OK;
}

} // namespace pt
