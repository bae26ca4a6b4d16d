"""Small JPEG encoder for the loader tests (test data generator, no product code): baseline and
progressive (spectral selection + successive approximation) Huffman JPEGs with the Annex K tables,
any of the usual chroma subsamplings, optional restart intervals, JFIF or Adobe (RGB / CMYK) markers."""
import struct

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63])
Q_LUM = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51,
                  87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120,
                  101, 72, 92, 95, 98, 112, 100, 103, 99])
Q_CHR = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99,
                  99, 99, 99] + [99] * 32)
# Annex K Huffman tables: (bits[16], values)
DC_LUM = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHR = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUM = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
          [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
           0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
           0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
           0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
           0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
           0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
           0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
           0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
AC_CHR = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
          [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
           0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
           0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
           0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
           0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
           0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
           0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
           0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])


# progressive AC scans also use the EOBn symbols, which Annex K's tables lack: a flat table of all 256 symbols
AC_FLAT = ([0, 0, 0, 0, 0, 0, 0, 255, 1, 0, 0, 0, 0, 0, 0, 0], list(range(256)))


def _codes(table):
    bits, vals = table
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, length):
        if length == 0:
            return
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _mag(v):
    a = abs(int(v))
    s = a.bit_length()
    return s, (int(v) if v >= 0 else int(v) + (1 << s) - 1)


_D = np.array([[(np.sqrt(0.125) if u == 0 else 0.5) * np.cos((2 * x + 1) * u * np.pi / 16) for x in range(8)] for u in range(8)])


def _planes(img, sub, cmyk):
    """component planes (float, level-shifted later) + sampling factors"""
    img = np.asarray(img, np.float64)
    if img.ndim == 2:
        return [img], [(1, 1)]
    if cmyk or sub == "rgb":
        return [img[..., k] for k in range(img.shape[2])], [(1, 1)] * img.shape[2]
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128
    cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128
    hv = {"444": (1, 1), "422": (2, 1), "420": (2, 2), "440": (1, 2), "411": (4, 1)}[sub]
    return [y, cb, cr], [hv, (1, 1), (1, 1)]


def encode(img, sub="444", quality=85, restart=0, progressive=False, jfif=True, adobe=None, cmyk=False, q16=False):
    """img: (H, W) grey or (H, W, 3|4) uint8.  sub: 444/422/420/440/411, or "rgb" to store RGB planes as
    they are (use with jfif=False, adobe=0).  adobe: None or the APP14 colour-transform byte."""
    planes, samp = _planes(img, sub, cmyk)
    H, W = planes[0].shape
    ncomp = len(planes)
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    scale = 5000 / quality if quality < 50 else 200 - 2 * quality
    qt = [np.clip((q * scale + 50) // 100, 1, 255 if not q16 else 1000).astype(np.int64) for q in (Q_LUM, Q_CHR)]
    if q16:
        qt[0] = qt[0] + 300                              # force 16-bit quantisation tables
    tq = [0 if (k == 0 or cmyk or sub == "rgb") else 1 for k in range(ncomp)]
    mcux, mcuy = (W + 8 * hmax - 1) // (8 * hmax), (H + 8 * vmax - 1) // (8 * vmax)
    # quantised coefficients per component: [block rows][block cols][64] in zigzag order
    coef = []
    for k, (pl, (h, v)) in enumerate(zip(planes, samp)):
        fx, fy = hmax // h, vmax // v                    # this component is subsampled by (fx, fy)
        ph, pw = mcuy * v * 8, mcux * h * 8
        full = np.pad(pl, ((0, mcuy * vmax * 8 - H), (0, mcux * hmax * 8 - W)), mode="edge")
        small = full.reshape(ph, fy, pw, fx).mean(axis=(1, 3)) - 128.0
        blocks = small.reshape(ph // 8, 8, pw // 8, 8).transpose(0, 2, 1, 3)
        dct = np.einsum("ux,abxy,vy->abuv", _D, blocks, _D)
        q = np.rint(dct.reshape(ph // 8, pw // 8, 64) / qt[tq[k]]).astype(np.int64)
        coef.append(q[..., ZIGZAG])
    out = bytearray(b"\xff\xd8")

    def seg(marker, payload):
        out.extend(struct.pack(">BBH", 0xFF, marker, len(payload) + 2) + payload)
    if jfif:
        seg(0xE0, b"JFIF\0\1\1\0\0\1\0\1\0\0")
    if adobe is not None:
        seg(0xEE, b"Adobe\0\x64\0\0\0\0" + bytes([adobe]))
    for t in sorted(set(tq)):
        if q16:
            seg(0xDB, bytes([0x10 | t]) + b"".join(struct.pack(">H", int(v)) for v in qt[t][ZIGZAG]))
        else:
            seg(0xDB, bytes([t]) + bytes(int(v) for v in qt[t][ZIGZAG]))
    ids = [1, 2, 3, 4][:ncomp]
    seg(0xC2 if progressive else 0xC0, struct.pack(">BHHB", 8, H, W, ncomp) +
        b"".join(bytes([ids[k], (samp[k][0] << 4) | samp[k][1], tq[k]]) for k in range(ncomp)))
    tabs = {(0, 0): DC_LUM, (0, 1): DC_CHR, (1, 0): AC_FLAT if progressive else AC_LUM, (1, 1): AC_FLAT if progressive else AC_CHR}
    for (cls, tid), t in tabs.items():
        if tid in set(tq):
            seg(0xC4, bytes([(cls << 4) | tid]) + bytes(t[0]) + bytes(t[1]))
    if restart:
        seg(0xDD, struct.pack(">H", restart))
    dc_codes = {0: _codes(DC_LUM), 1: _codes(DC_CHR)}
    ac_codes = {0: _codes(tabs[(1, 0)]), 1: _codes(tabs[(1, 1)])}

    def scan(comps, ss, se, ah, al):
        """one scan over components `comps` (interleaved when more than one)"""
        seg(0xDA, bytes([len(comps)]) + b"".join(bytes([ids[k], (tq[k] << 4) | tq[k]]) for k in comps) + bytes([ss, se, (ah << 4) | al]))
        bw = _Bits()
        pred = {k: 0 for k in comps}
        state = {"eobrun": 0, "pending": []}

        def flush_eobrun(k):
            n = state["eobrun"]
            if n:
                s = n.bit_length() - 1
                c, ln = ac_codes[tq[k]][s << 4]
                bw.put(c, ln)
                bw.put(n - (1 << s), s)
                state["eobrun"] = 0
            for b in state["pending"]:
                bw.put(b, 1)
            state["pending"] = []

        def block(k, zz):
            t = tq[k]
            if ss == 0:                                  # DC (baseline: followed by the ACs)
                if ah == 0:
                    v = int(zz[0]) >> al if progressive else int(zz[0])
                    s, bits = _mag(v - pred[k])
                    pred[k] = v
                    c, ln = dc_codes[t][s]
                    bw.put(c, ln)
                    bw.put(bits, s)
                else:
                    bw.put((int(zz[0]) >> al) & 1, 1)
                if progressive:
                    return
            lo = max(ss, 1)
            if not progressive or ah == 0:               # first pass over ACs (or baseline)
                run = 0
                for i in range(lo, se + 1):
                    v = int(zz[i])
                    if progressive:
                        v = (abs(v) >> al) * (1 if v >= 0 else -1)
                    if v == 0:
                        run += 1
                        continue
                    flush_eobrun(k)
                    while run > 15:
                        c, ln = ac_codes[t][0xF0]
                        bw.put(c, ln)
                        run -= 16
                    s, bits = _mag(v)
                    c, ln = ac_codes[t][(run << 4) | s]
                    bw.put(c, ln)
                    bw.put(bits, s)
                    run = 0
                if run:
                    if progressive:
                        state["eobrun"] += 1
                        if state["eobrun"] == 0x7FFF:
                            flush_eobrun(k)
                    else:
                        c, ln = ac_codes[t][0x00]
                        bw.put(c, ln)
                return
            # AC refinement pass
            absv = [abs(int(zz[i])) >> al for i in range(64)]
            last_new = 0
            for i in range(lo, se + 1):
                if absv[i] == 1:
                    last_new = i
            run, corr = 0, []
            for i in range(lo, se + 1):
                a = absv[i]
                if a == 0:
                    run += 1
                    continue
                while run > 15 and i <= last_new:
                    flush_eobrun(k)
                    c, ln = ac_codes[t][0xF0]
                    bw.put(c, ln)
                    run -= 16
                    for b in corr:
                        bw.put(b, 1)
                    corr = []
                if a > 1:
                    corr.append(a & 1)
                    continue
                flush_eobrun(k)
                c, ln = ac_codes[t][(run << 4) | 1]
                bw.put(c, ln)
                bw.put(1 if zz[i] >= 0 else 0, 1)
                for b in corr:
                    bw.put(b, 1)
                corr, run = [], 0
            if run or corr:
                state["eobrun"] += 1
                state["pending"] += corr
                if state["eobrun"] == 0x7FFF or len(state["pending"]) > 900:
                    flush_eobrun(k)

        n_units = 0
        rst = 0

        def maybe_restart(k_last):
            nonlocal n_units, rst
            n_units += 1
            if restart and n_units % restart == 0 and n_units < total_units:
                flush_eobrun(k_last)
                bw.flush()
                bw.out.extend(bytes([0xFF, 0xD0 + (rst & 7)]))
                rst += 1
                for k in pred:
                    pred[k] = 0
        if len(comps) > 1:
            total_units = mcux * mcuy
            for my in range(mcuy):
                for mx in range(mcux):
                    for k in comps:
                        h, v = samp[k]
                        for by in range(v):
                            for bx in range(h):
                                block(k, coef[k][my * v + by, mx * h + bx])
                    maybe_restart(comps[-1])
        else:
            k = comps[0]
            h, v = samp[k]
            bwid = (((W * h + hmax - 1) // hmax) + 7) // 8     # non-interleaved: only the blocks that cover the image
            bhei = (((H * v + vmax - 1) // vmax) + 7) // 8
            total_units = bwid * bhei
            for by in range(bhei):
                for bx in range(bwid):
                    block(k, coef[k][by, bx])
                    maybe_restart(k)
        flush_eobrun(comps[-1])
        bw.flush()
        out.extend(bw.out)

    allc = list(range(ncomp))
    if not progressive:
        scan(allc, 0, 63, 0, 0)
    else:
        scan(allc, 0, 0, 0, 1)                           # DC, all but the lowest bit
        for k in allc:
            scan([k], 1, 5, 0, 2)                        # low ACs, top bits
        for k in allc:
            scan([k], 6, 63, 0, 2)
        for k in allc:
            scan([k], 1, 63, 2, 1)                       # AC refinement
        scan(allc, 0, 0, 1, 0)                           # DC refinement
        for k in allc:
            scan([k], 1, 63, 1, 0)
    out.extend(b"\xff\xd9")
    return bytes(out)


def _test_image(rng, H, W, noise=12):
    y, x = np.mgrid[0:H, 0:W]
    img = np.stack([(x * 255 // max(W - 1, 1)), (y * 255 // max(H - 1, 1)), ((x + y) * 7) % 256], -1).astype(np.float64)
    img += rng.normal(0, noise, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def jpeg_variants(seed=5):
    """name -> JPEG bytes: every chroma subsampling x {sequential, progressive} x {no restarts, DRI} on four
    frame sizes (incl. 1x1 and sizes that are not multiples of the MCU), plus RGB-as-is (Adobe transform 0 /
    no APP marker), grey, 16-bit quantisation tables, CMYK, YCCK, four components without an Adobe marker,
    and a saturated checker."""
    rng = np.random.default_rng(seed)
    out = {}
    for (H, W) in [(37, 53), (1, 1), (8, 9), (33, 7)]:
        for sub in ("444", "422", "420", "440", "411"):
            for prog in (False, True):
                for rst in (0, 3):
                    name = "%dx%d_%s%s%s" % (W, H, sub, "_prog" if prog else "", "_rst" if rst else "")
                    out[name] = encode(_test_image(rng, H, W), sub=sub, progressive=prog, restart=rst,
                                       quality=int(rng.integers(30, 96)))
    img = _test_image(rng, 40, 56)
    cmyk = np.concatenate([img, img[..., :1]], -1)
    out["rgb_adobe"] = encode(img, sub="rgb", jfif=False, adobe=0)
    out["rgb_noapp"] = encode(img, sub="rgb", jfif=False)
    out["grey"] = encode(img[..., 0])
    out["grey_prog"] = encode(img[..., 0], progressive=True)
    out["q16"] = encode(img, q16=True, sub="420")
    out["cmyk"] = encode(cmyk, cmyk=True, adobe=0, jfif=False)
    out["ycck_prog"] = encode(cmyk, cmyk=True, adobe=2, jfif=False, progressive=True)
    out["four_no_adobe"] = encode(cmyk, cmyk=True, jfif=False)
    out["saturated"] = encode((rng.integers(0, 2, (24, 24, 3)) * 255).astype(np.uint8), sub="420", quality=95)
    return out


def jpeg_malformed(seed=6):
    """name -> bytes whose treatment by stb_image is well defined: refused, or decoded from what is there."""
    rng = np.random.default_rng(seed)
    base = encode(_test_image(rng, 24, 40), sub="420")
    assert b"\xff\xc0\x00\x11\x08\x00\x18\x00\x28" in base
    return {
        "no_eoi": base[:-2],
        "cut_in_scan": base[:len(base) * 2 // 3],
        "cut_in_header": base[:40],
        "precision12": base.replace(b"\xff\xc0\x00\x11\x08", b"\xff\xc0\x00\x11\x0c"),
        "arithmetic": base.replace(b"\xff\xc0", b"\xff\xc9"),
        "lossless": base.replace(b"\xff\xc0", b"\xff\xc3"),
        "zero_width": base.replace(b"\xff\xc0\x00\x11\x08\x00\x18\x00\x28", b"\xff\xc0\x00\x11\x08\x00\x18\x00\x00"),
        "soi_only": b"\xff\xd8\xff\xd9",
        "junk_after_eoi": base + b"garbage",
        "junk_before_sof": base[:2] + b"\x00\x00" + base[2:],
        "comment": base[:2] + b"\xff\xfe\x00\x07hello" + base[2:],
    }
