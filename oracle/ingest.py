"""CPU oracle of the frame ingest.  TEST INFRASTRUCTURE ONLY: nothing under
visual_odometry_amd/ imports this file.

What it restates: `imread`, `rgb2gray` and `im2uint8` as the reference's frame loops call them
(main.m:15-30,52-59; utils/dataLoaderKitti.m:29; utils/dataLoaderParking.m:29;
utils/dataLoaderMalaga.m:29-30) and the loaders' bookkeeping (utils/dataLoader*.m, utils/loadGeneralData.m).

PARITY UNPINNED for rgb2gray: it is MATLAB toolbox code, absent from the reference, and the reference
holds no image fixture.  Normative definition N11:

    gray = floor(((cr*R + cg*G) + cb*B) + 0.5) in IEEE double, nothing fused, clamped to 255, with
    (cr, cg, cb) = (0.298936021293775390, 0.587043074451121360, 0.114020904255103250)
    -- the first row of inv([1 .956 .621; 1 -.272 -.647; 1 -1.106 1.703]) that the rgb2gray
    documentation names; im2uint8 of a uint16 sample = round(x / 257).

PNG decoding is pinned differently: the format is a public standard (ISO/IEC 15948), the decoder below
follows it (zlib stream -> row filters None/Sub/Up/Average/Paeth -> samples), and the CPU tests check it
byte for byte against an independent decoder (Pillow, present in the image) on files Pillow wrote.
"""
import struct
import zlib

import numpy as np

GRAY_R, GRAY_G, GRAY_B = 0.298936021293775390, 0.587043074451121360, 0.114020904255103250
PNG_MAGIC = b"\x89PNG\r\n\x1a\n"
_SPP = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def rgb2gray(rgb):
    """N11.  rgb: (..., 3) uint8 -> (...) uint8."""
    a = np.asarray(rgb, np.uint8).astype(np.float64)
    s = (GRAY_R * a[..., 0] + GRAY_G * a[..., 1]) + GRAY_B * a[..., 2]
    return np.minimum(np.floor(s + 0.5), 255.0).astype(np.uint8)


def im2uint8(x):
    x = np.asarray(x)
    if x.dtype == np.uint8:
        return x
    if x.dtype == np.uint16:
        return ((x.astype(np.uint32) + 128) // 257).astype(np.uint8)
    raise TypeError("im2uint8: uint8 or uint16 input")


def _unfilter(ft, cur, prev, bpp):
    """One row, in place on a uint8 array `cur`; prev is the reconstructed row above (zeros on row 0)."""
    n = len(cur)
    if ft == 0:
        return
    if ft == 2:
        cur += prev                                # uint8 wrap-around is the specified arithmetic
        return
    c = cur.astype(np.int32)
    p = prev.astype(np.int32)
    if ft == 1:
        for i in range(bpp, n):
            c[i] = (c[i] + c[i - bpp]) & 255
    elif ft == 3:
        for i in range(n):
            a = c[i - bpp] if i >= bpp else 0
            c[i] = (c[i] + ((a + p[i]) >> 1)) & 255
    elif ft == 4:
        for i in range(n):
            a = c[i - bpp] if i >= bpp else 0
            b = p[i]
            cc = p[i - bpp] if i >= bpp else 0
            pp = a + b - cc
            pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - cc)
            pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
            c[i] = (c[i] + pred) & 255
    else:
        raise ValueError("png: unknown row filter")
    cur[:] = c.astype(np.uint8)


def png_decode(data):
    """-> (H, W) or (H, W, 3) uint8: alpha dropped, 16-bit samples through im2uint8, gray below 8 bits
    scaled to 0..255, palette expanded.  Pure-Python row loops: small images only."""
    if data[:8] != PNG_MAGIC:
        raise ValueError("not a PNG")
    pos, idat, pal, hdr = 8, [], None, None
    while pos + 12 <= len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        crc, = struct.unpack(">I", data[pos + 8 + ln:pos + 12 + ln])
        if zlib.crc32(typ + body) != crc and typ in (b"IHDR", b"IDAT"):
            raise ValueError("png: checksum mismatch")
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            pal = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + ln
    W, H, bd, ct, _, _, il = hdr
    if il:
        raise NotImplementedError("png: interlaced")
    spp = _SPP[ct]
    rb = (W * spp * bd + 7) // 8
    bpp = max(1, spp * bd // 8)
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    if len(raw) != H * (rb + 1):
        raise ValueError("png: pixel data truncated")
    rows = raw.reshape(H, rb + 1)
    rec = np.zeros((H, rb), np.uint8)
    prev = np.zeros(rb, np.uint8)
    for y in range(H):
        cur = rows[y, 1:].copy()
        _unfilter(int(rows[y, 0]), cur, prev, bpp)
        rec[y] = cur
        prev = cur
    if bd < 8:
        bits = np.unpackbits(rec, axis=1)[:, :W * bd].reshape(H, W, bd)
        v = np.zeros((H, W), np.int32)
        for k in range(bd):
            v = (v << 1) | bits[:, :, k]
        if ct == 3:
            return pal[v]
        return (v * (255 // ((1 << bd) - 1))).astype(np.uint8)
    if bd == 8:
        s = rec.reshape(H, W, spp)
    else:
        s = im2uint8((rec[:, 0::2].astype(np.uint16) << 8 | rec[:, 1::2]).reshape(H, W, spp))
    if ct == 3:
        return pal[s[:, :, 0]]
    if ct in (0, 4):
        return np.ascontiguousarray(s[:, :, 0])
    return np.ascontiguousarray(s[:, :, :3])


def png_encode(img, filters=None, level=6):
    """Test-fixture writer (8-bit gray or RGB).  filters: per-row filter types (default: cycle 0..4) so that
    every reconstruction rule is exercised without depending on what another encoder chooses."""
    a = np.asarray(img, np.uint8)
    H, W = a.shape[:2]
    spp = 1 if a.ndim == 2 else a.shape[2]
    ct = {1: 0, 3: 2, 4: 6, 2: 4}[spp]
    rec = a.reshape(H, W * spp).astype(np.int32)
    out = bytearray()
    zero = np.zeros(W * spp, np.int32)
    for y in range(H):
        ft = (y % 5) if filters is None else int(filters[y % len(filters)])
        cur, up = rec[y], (rec[y - 1] if y else zero)
        left = np.concatenate([np.zeros(spp, np.int32), cur[:-spp]])
        ul = np.concatenate([np.zeros(spp, np.int32), up[:-spp]])
        if ft == 0:
            pred = zero
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = up
        elif ft == 3:
            pred = (left + up) >> 1
        else:
            p = left + up - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        out.append(ft)
        out += ((cur - pred) & 255).astype(np.uint8).tobytes()

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    z = zlib.compress(bytes(out), level)
    half = len(z) // 2                              # two IDAT chunks: the stream continues across them
    return (PNG_MAGIC + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, ct, 0, 0, 0)) + chunk(b"IDAT", z[:half])
            + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b""))


def pnm_decode(data):
    tok, pos = [], 2
    while len(tok) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        e = pos
        while data[e:e + 1].isdigit():
            e += 1
        tok.append(int(data[pos:e]))
        pos = e
    W, H, mv = tok
    ch = 1 if data[1:2] == b"5" else 3
    if mv < 256:
        v = np.frombuffer(data, np.uint8, W * H * ch, pos + 1).astype(np.int64)
    else:
        v = np.frombuffer(data, ">u2", W * H * ch, pos + 1).astype(np.int64)
    v = (np.minimum(v, mv) * 510 + mv) // (2 * mv)
    return v.astype(np.uint8).reshape((H, W) if ch == 1 else (H, W, 3))


def imread(data):
    if data[:8] == PNG_MAGIC:
        return png_decode(data)
    if data[:2] in (b"P5", b"P6"):
        return pnm_decode(data)
    if data[:2] == b"\xff\xd8":
        return jpeg_decode(data)
    raise NotImplementedError("oracle imread: PNG, baseline JPEG and binary PGM/PPM")


def to_gray(img):
    """What every frame loop of the reference feeds the pipeline: uint8 gray (main.m:20-30,54-59)."""
    return rgb2gray(img) if img.ndim == 3 else img


# ------------------------------------------------------------------------------------------------
# N12: baseline JPEG.  The stream is ITU-T T.81 (public standard); the arithmetic after the entropy decoder
# is not fixed by the standard, so it is DEFINED here, as the common integer decoders do it, and pinned by
# the CPU tests against Pillow's decoder (byte for byte on files of every sampling / quality tried):
#   dequantise: coefficient * table entry (int; the product and the IDCT's sums never leave 32 bits for a
#     valid stream -- the native code lets them wrap for a damaged one, this restatement does not model that);
#   inverse DCT: Loeffler-Ligtenberg-Moschytz 8-point, 13-bit constants, columns first with the result
#     rounded to 2 extra bits ((x + 1024) >> 11), then rows ((x + 131072) >> 18), + 128, range-limited
#     through a 1024-entry wrap (x & 1023 -> 0..255);
#   chroma at 2:1: triangle filter, (3 near + far + 1|2) >> 2 per axis, or (3(3a+b) + (3c+d) + 8|7) >> 4
#     for both, neighbours clamped to the component's true size;
#   YCbCr -> RGB: 16.16 fixed point (91881, 22554, 46802, 116130), + 32768, arithmetic shift, clamped.
# Pure-Python bit reader: small images only.
# ------------------------------------------------------------------------------------------------
def _zigzag():
    t, r, c = [], 0, 0
    for _ in range(64):
        t.append(r * 8 + c)
        if (r + c) % 2 == 0:
            if c == 7: r += 1
            elif r == 0: c += 1
            else: r, c = r - 1, c + 1
        else:
            if r == 7: c += 1
            elif c == 0: r += 1
            else: r, c = r + 1, c - 1
    return t


_ZZ = _zigzag()


def _idct8(v):
    """v: (8, ...) int64 -> (8, ...) un-descaled sums."""
    z2, z3 = v[2], v[6]
    z1 = (z2 + z3) * 4433
    t2, t3 = z1 - z3 * 15137, z1 + z2 * 6270
    t0, t1 = (v[0] + v[4]) * 8192, (v[0] - v[4]) * 8192
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    o0, o1, o2, o3 = v[7], v[5], v[3], v[1]
    z1, z2, z3, z4 = o0 + o3, o1 + o2, o0 + o2, o1 + o3
    z5 = (z3 + z4) * 9633
    o0, o1, o2, o3 = o0 * 2446, o1 * 16819, o2 * 25172, o3 * 12299
    z1, z2, z3, z4 = z1 * -7373, z2 * -20995, z3 * -16069 + z5, z4 * -3196 + z5
    o0, o1, o2, o3 = o0 + z1 + z3, o1 + z2 + z4, o2 + z2 + z3, o3 + z1 + z4
    return np.stack([t10 + o3, t11 + o2, t12 + o1, t13 + o0, t13 - o0, t12 - o1, t11 - o2, t10 - o3])


def _range_limit(x):
    i = x & 1023
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896)))


def jpeg_idct_block(coef, qt):
    """coef, qt: (8, 8) natural order -> (8, 8) uint8 samples."""
    d = coef.astype(np.int64) * qt.astype(np.int64)
    ws = (_idct8(d) + 1024) >> 11                      # columns: axis 0 is the frequency / row index
    out = (_idct8(ws.T) + 131072) >> 18                 # rows
    return _range_limit(out.T).astype(np.uint8)


def _chroma(p, dw, dh, hs, vs, W, H):
    """Upsample plane p (true size dh x dw) to H x W with the clamped triangle filter."""
    p = p.astype(np.int64)
    y, x = np.arange(H), np.arange(W)
    if not hs and not vs:
        return p[:H, :W]
    if hs and not vs:
        i = x >> 1
        nb = np.clip(np.where(x & 1, i + 1, i - 1), 0, dw - 1)
        return (3 * p[:H][:, i] + p[:H][:, nb] + np.where(x & 1, 2, 1)) >> 2
    j = y >> 1
    jn = np.clip(np.where(y & 1, j + 1, j - 1), 0, dh - 1)
    if not hs:
        return (3 * p[j][:, :W] + p[jn][:, :W] + np.where(y & 1, 2, 1)[:, None]) >> 2
    cs = 3 * p[j] + p[jn]                               # H x (plane width)
    i = x >> 1
    nb = np.clip(np.where(x & 1, i + 1, i - 1), 0, dw - 1)
    return (3 * cs[:, i] + cs[:, nb] + np.where(x & 1, 7, 8)) >> 4


def jpeg_decode(data):
    """Baseline JPEG -> (H, W) or (H, W, 3) uint8 (N12)."""
    if data[:2] != b"\xff\xd8":
        raise ValueError("not a JPEG")
    pos, qt, huff, ri = 2, {}, {}, 0
    frame = None
    while True:
        while data[pos] != 0xFF:
            pos += 1
        while data[pos] == 0xFF:
            pos += 1
        m = data[pos]; pos += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        ln = struct.unpack(">H", data[pos:pos + 2])[0]
        s = data[pos + 2:pos + ln]
        if m in (0xC0, 0xC1):
            prec, H, W, nc = struct.unpack(">BHHB", s[:6])
            comps = [(s[6 + 3 * c], s[7 + 3 * c] >> 4, s[7 + 3 * c] & 15, s[8 + 3 * c]) for c in range(nc)]
            if nc == 1:
                comps = [(comps[0][0], 1, 1, comps[0][3])]
            frame = (H, W, nc, comps)
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8):
            raise NotImplementedError("jpeg: not baseline")
        elif m == 0xDB:
            i = 0
            while i < len(s):
                pq, tq = s[i] >> 4, s[i] & 15
                i += 1
                t = np.zeros(64, np.int64)
                for k in range(64):
                    t[_ZZ[k]] = struct.unpack(">H", s[i + 2 * k:i + 2 * k + 2])[0] if pq else s[i + k]
                i += 128 if pq else 64
                qt[tq] = t.reshape(8, 8)
        elif m == 0xC4:
            i = 0
            while i < len(s):
                tc, th = s[i] >> 4, s[i] & 15
                bits = list(s[i + 1:i + 17]); i += 17
                vals = list(s[i:i + sum(bits)]); i += sum(bits)
                table, code, k = {}, 0, 0
                for l in range(1, 17):
                    for _ in range(bits[l - 1]):
                        table[(l, code)] = vals[k]; code += 1; k += 1
                    code <<= 1
                huff[(tc, th)] = table
        elif m == 0xDD:
            ri = struct.unpack(">H", s[:2])[0]
        elif m == 0xDA:
            ns = s[0]
            sel = [(s[1 + 2 * c], s[2 + 2 * c] >> 4, s[2 + 2 * c] & 15) for c in range(ns)]
            pos += ln
            break
        pos += ln
    H, W, nc, comps = frame
    hmax, vmax = comps[0][1], comps[0][2]
    mcux, mcuy = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    # unstuff the entropy-coded segment into a bit list per restart interval
    seg, i = [], pos
    cur = bytearray()
    while True:
        b = data[i]
        if b == 0xFF:
            b2 = data[i + 1]
            if b2 == 0:
                cur.append(0xFF); i += 2; continue
            seg.append(bytes(cur)); cur = bytearray()
            if 0xD0 <= b2 <= 0xD7:
                i += 2; continue
            break
        cur.append(b); i += 1

    class Reader:
        def __init__(self, by):
            self.bits = np.unpackbits(np.frombuffer(by, np.uint8)).tolist() + [0] * 64
            self.p = 0

        def take(self, n):
            v = 0
            for _ in range(n):
                v = (v << 1) | self.bits[self.p]; self.p += 1
            return v

        def sym(self, table):
            code = 0
            for l in range(1, 17):
                code = (code << 1) | self.bits[self.p]; self.p += 1
                if (l, code) in table:
                    return table[(l, code)]
            raise ValueError("jpeg: bad code")

    def extend(v, s):
        return v - (1 << s) + 1 if s and v < (1 << (s - 1)) else v

    planes = [np.zeros((mcuy * c[2] * 8, mcux * c[1] * 8), np.uint8) for c in comps]
    rd, si, pred, left = Reader(seg[0]), 0, [0] * nc, ri
    for my in range(mcuy):
        for mx in range(mcux):
            if ri and left == 0:
                si += 1; rd = Reader(seg[si]); pred = [0] * nc; left = ri
            for c, (cid, h, v, tq) in enumerate(comps):
                td, ta = sel[c][1], sel[c][2]
                for by in range(v):
                    for bx in range(h):
                        blk = np.zeros(64, np.int64)
                        s_ = rd.sym(huff[(0, td)])
                        pred[c] += extend(rd.take(s_), s_)
                        blk[0] = pred[c]
                        k = 1
                        while k < 64:
                            rs = rd.sym(huff[(1, ta)])
                            r, s_ = rs >> 4, rs & 15
                            if s_ == 0:
                                if r != 15:
                                    break
                                k += 16; continue
                            k += r
                            blk[_ZZ[k]] = extend(rd.take(s_), s_); k += 1
                        y0, x0 = (my * v + by) * 8, (mx * h + bx) * 8
                        planes[c][y0:y0 + 8, x0:x0 + 8] = jpeg_idct_block(blk.reshape(8, 8), qt[tq])
            left -= 1
    if nc == 1:
        return np.ascontiguousarray(planes[0][:H, :W])
    hs, vs = hmax == 2, vmax == 2
    dw, dh = -(-W // hmax), -(-H // vmax)
    Y = planes[0][:H, :W].astype(np.int64)
    u = _chroma(planes[1], dw, dh, hs, vs, W, H) - 128
    v = _chroma(planes[2], dw, dh, hs, vs, W, H) - 128
    R = Y + ((91881 * v + 32768) >> 16)
    G = Y + ((-22554 * u - 46802 * v + 32768) >> 16)
    B = Y + ((116130 * u + 32768) >> 16)
    return np.clip(np.stack([R, G, B], -1), 0, 255).astype(np.uint8)
