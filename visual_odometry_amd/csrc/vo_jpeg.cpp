// vo_jpeg.cpp -- baseline (sequential Huffman, 8-bit) JPEG decoding on the host; see vo_jpeg.h.
// Follows ITU-T T.81 for the stream and definition N12 (oracle/ingest.py) for the arithmetic after it.
// Decoded: one interleaved scan, gray or YCbCr with chroma at 1:1, 2:1 horizontally and/or vertically,
// restart intervals.  Not decoded (VO_ERR_UNSUPPORTED): progressive, lossless, arithmetic coding, 12-bit,
// multi-scan files, CMYK / Adobe RGB, other sampling factors.
#include "vo_jpeg.h"

#include <cstring>

#include "../../include/vo_hip.h"

namespace {

struct Huff {
    bool present = false;
    uint8_t bits[17];
    uint8_t vals[256];
    uint16_t look[512];  // 9-bit prefix -> (length << 8) | symbol, 0 when the code is longer
    int maxcode[18], mincode[17], valptr[17];
};

bool huff_build(Huff* h) {
    int code = 0, k = 0;
    memset(h->look, 0, sizeof(h->look));
    for (int l = 1; l <= 16; ++l) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        for (int i = 0; i < h->bits[l]; ++i, ++k, ++code) {
            if (l <= 9) {
                const int first = code << (9 - l), cnt = 1 << (9 - l);
                if (first + cnt > 512) return false;
                for (int e = 0; e < cnt; ++e) h->look[first + e] = (uint16_t)((l << 8) | h->vals[k]);
            }
        }
        if (code > (1 << l)) return false;
        h->maxcode[l] = h->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
    h->present = true;
    return true;
}

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int cnt = 0;
    int marker = 0;  // a marker met in the entropy data: zeros are fed from there on
    int fake = 0;    // how many of the cnt bits in acc are such zeros (behind a marker or the end of the buffer)
    bool overrun = false;  // a code or value was read out of those zeros: the data ended before the decoder did

    void fill() {
        if (cnt <= 32 && !marker && p + 4 <= end) {  // four plain bytes at once when none of them is 0xff
            const uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
            if (!((w & ~(w + 0x01010101u) & 0x80808080u))) {  // no byte equals 0xff
                acc |= (uint64_t)w << (32 - cnt);
                cnt += 32;
                p += 4;
                if (cnt > 56) return;
            }
        }
        while (cnt <= 56) {
            unsigned b = 0;
            bool real = false;
            if (!marker && p < end) {
                b = *p;
                real = true;
                if (b == 0xff) {
                    const unsigned b2 = p + 1 < end ? p[1] : 0xd9;
                    if (b2 == 0) p += 2;  // stuffed zero
                    else {
                        marker = (int)b2;
                        b = 0;
                        real = false;
                    }
                } else {
                    ++p;
                }
            }
            acc |= (uint64_t)b << (56 - cnt);
            cnt += 8;
            if (!real) fake += 8;
        }
    }
    unsigned peek(int n) const { return (unsigned)(acc >> (64 - n)); }
    void skip(int n) {
        acc <<= n;
        cnt -= n;
        if (cnt < fake) {  // bits taken from the zero fill
            overrun = true;
            fake = cnt;
        }
    }
    int receive_extend(int s) {
        if (s == 0) return 0;
        const int v = (int)peek(s);
        skip(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    int decode(const Huff& h) {
        fill();
        const unsigned e = h.look[peek(9)];
        if (e) {
            skip((int)(e >> 8));
            return (int)(e & 255);
        }
        for (int l = 10; l <= 16; ++l) {
            const int code = (int)peek(l);
            if (code <= h.maxcode[l]) {
                skip(l);
                return h.vals[h.valptr[l] + code - h.mincode[l]];
            }
        }
        return -1;
    }
    // byte-align at a restart boundary and step over the RSTn marker
    bool restart() {
        acc = 0;
        cnt = 0;
        fake = 0;
        if (!marker) {
            while (p + 1 < end && !(p[0] == 0xff && p[1] != 0 && p[1] != 0xff)) ++p;
            if (p + 1 >= end) return false;
            marker = p[1];
        }
        if (marker < 0xd0 || marker > 0xd7) return false;
        p += 2;
        marker = 0;
        return true;
    }
};

struct ZigZag {  // position in the 8x8 block (row-major) of the k-th coefficient of the zigzag scan
    uint8_t at[64 + 16];
    ZigZag() {
        int r = 0, c = 0;
        for (int i = 0; i < 64; ++i) {
            at[i] = (uint8_t)(r * 8 + c);
            if ((r + c) % 2 == 0) {  // moving up-right
                if (c == 7) ++r;
                else if (r == 0) ++c;
                else { --r; ++c; }
            } else {                 // moving down-left
                if (r == 7) ++c;
                else if (c == 0) ++r;
                else { ++r; --c; }
            }
        }
        for (int i = 64; i < 80; ++i) at[i] = 63;
    }
};
const ZigZag ZZ;
inline int zigzag_natural(int k) { return ZZ.at[k]; }

inline unsigned be16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }

struct Parsed {
    JpegFrame f;
    uint16_t qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int cid[3], td[3], ta[3];
    int ri = 0;
    bool have_sof = false;
    size_t scan = 0;  // offset of the entropy-coded data
};

int fail(std::string* err, const char* what, int code = VO_ERR_BAD_ARG) {
    *err = std::string("jpeg: ") + what;
    return code;
}

int parse(const uint8_t* d, size_t n, Parsed* P, bool header_only, std::string* err) {
    if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return fail(err, "no SOI marker");
    size_t pos = 2;
    int adobe_transform = -1;
    for (;;) {
        while (pos < n && d[pos] != 0xff) ++pos;  // tolerate garbage between segments
        while (pos < n && d[pos] == 0xff) ++pos;  // fill bytes
        if (pos >= n) return fail(err, "no scan in the file");
        const int m = d[pos++];
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
        if (m == 0xd9) return fail(err, "no scan in the file");
        if (pos + 2 > n) return fail(err, "truncated segment");
        const size_t len = be16(d + pos);
        if (len < 2 || pos + len > n) return fail(err, "truncated segment");
        const uint8_t* s = d + pos + 2;
        const size_t sl = len - 2;
        if (m == 0xc0 || m == 0xc1) {
            // one frame per file: a later SOF would re-size the frame behind the header the caller sized its
            // buffers from (vo_jpeg_header returns at the first one)
            if (P->have_sof) return fail(err, "more than one frame header");
            if (sl < 6) return fail(err, "bad SOF");
            if (s[0] != 8) return fail(err, "only 8-bit samples are decoded", VO_ERR_UNSUPPORTED);
            JpegFrame& f = P->f;
            f.H = (int)be16(s + 1);
            f.W = (int)be16(s + 3);
            f.nc = s[5];
            if (f.W == 0 || f.H == 0) return fail(err, "empty image");
            if (f.nc != 1 && f.nc != 3) return fail(err, "only gray and YCbCr files are decoded", VO_ERR_UNSUPPORTED);
            if (sl < 6 + 3 * (size_t)f.nc) return fail(err, "bad SOF");
            for (int c = 0; c < f.nc; ++c) {
                P->cid[c] = s[6 + 3 * c];
                f.comp[c].h = s[7 + 3 * c] >> 4;
                f.comp[c].v = s[7 + 3 * c] & 15;
                f.comp[c].tq = s[8 + 3 * c] & 3;
            }
            if (f.nc == 1) f.comp[0].h = f.comp[0].v = 1;  // sampling factors of a single component mean nothing
            f.hmax = f.comp[0].h;
            f.vmax = f.comp[0].v;
            bool ok = f.hmax >= 1 && f.hmax <= 2 && f.vmax >= 1 && f.vmax <= 2;
            for (int c = 1; c < f.nc; ++c) ok = ok && f.comp[c].h == 1 && f.comp[c].v == 1;
            if (!ok) return fail(err, "sampling factors other than 1x1 / 2x1 / 1x2 / 2x2 luma over 1x1 chroma", VO_ERR_UNSUPPORTED);
            P->have_sof = true;
            if (header_only) return VO_OK;
        } else if (m >= 0xc2 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return fail(err, "progressive / lossless / arithmetic-coded files are not decoded", VO_ERR_UNSUPPORTED);
        } else if (m == 0xcc) {
            return fail(err, "arithmetic-coded files are not decoded", VO_ERR_UNSUPPORTED);
        } else if (m == 0xdb) {
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                ++i;
                if (tq > 3 || pq > 1 || i + (pq ? 128u : 64u) > sl) return fail(err, "bad DQT");
                for (int k = 0; k < 64; ++k) {
                    P->qt[tq][zigzag_natural(k)] = pq ? (uint16_t)be16(s + i + 2 * k) : s[i + k];
                }
                i += pq ? 128 : 64;
                P->have_qt[tq] = true;
            }
        } else if (m == 0xc4) {
            size_t i = 0;
            while (i < sl) {
                if (i + 17 > sl) return fail(err, "bad DHT");
                const int tc = s[i] >> 4, th = s[i] & 15;
                if (tc > 1 || th > 3) return fail(err, "bad DHT");
                Huff* h = tc ? &P->ac[th] : &P->dc[th];
                int total = 0;
                h->bits[0] = 0;
                for (int l = 1; l <= 16; ++l) total += (h->bits[l] = s[i + l]);
                i += 17;
                if (total > 256 || i + total > sl) return fail(err, "bad DHT");
                memcpy(h->vals, s + i, total);
                i += total;
                if (!huff_build(h)) return fail(err, "bad Huffman table");
            }
        } else if (m == 0xdd) {
            if (sl < 2) return fail(err, "bad DRI");
            P->ri = (int)be16(s);
        } else if (m == 0xee) {
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
        } else if (m == 0xda) {
            if (!P->have_sof) return fail(err, "scan before the frame header");
            const JpegFrame& f = P->f;
            if (sl < 1 || s[0] != f.nc) return fail(err, "files with more than one scan are not decoded", VO_ERR_UNSUPPORTED);
            if (sl < 1 + 2 * (size_t)f.nc + 3) return fail(err, "bad SOS");
            for (int c = 0; c < f.nc; ++c) {
                if (s[1 + 2 * c] != P->cid[c]) return fail(err, "scan components out of frame order", VO_ERR_UNSUPPORTED);
                P->td[c] = s[2 + 2 * c] >> 4;
                P->ta[c] = s[2 + 2 * c] & 15;
                if (P->td[c] > 3 || P->ta[c] > 3 || !P->dc[P->td[c]].present || !P->ac[P->ta[c]].present)
                    return fail(err, "scan names a missing Huffman table");
                if (!P->have_qt[f.comp[c].tq]) return fail(err, "frame names a missing quantisation table");
            }
            const uint8_t* t = s + 1 + 2 * f.nc;
            if (t[0] != 0 || t[1] != 63 || t[2] != 0) return fail(err, "not a baseline scan", VO_ERR_UNSUPPORTED);
            if (f.nc == 3 && adobe_transform == 0) return fail(err, "Adobe RGB files are not decoded", VO_ERR_UNSUPPORTED);
            P->scan = pos + len;
            return VO_OK;
        }
        pos += len;
    }
}

}  // namespace

int vo_jpeg_header(const uint8_t* d, size_t n, int* W, int* H, int* nc, std::string* err) {
    Parsed P;
    const int rc = parse(d, n, &P, true, err);
    if (rc != VO_OK) return rc;
    *W = P.f.W;
    *H = P.f.H;
    *nc = P.f.nc;
    return VO_OK;
}

namespace {
void layout(JpegFrame* fp, const Parsed& P) {
    JpegFrame& f = *fp;
    const int mcux = (f.W + 8 * f.hmax - 1) / (8 * f.hmax), mcuy = (f.H + 8 * f.vmax - 1) / (8 * f.vmax);
    size_t nb = 0;
    for (int c = 0; c < f.nc; ++c) {
        JpegComp& k = f.comp[c];
        k.bw = mcux * k.h;
        k.bh = mcuy * k.v;
        k.dw = (f.W * k.h + f.hmax - 1) / f.hmax;
        k.dh = (f.H * k.v + f.vmax - 1) / f.vmax;
        k.block0 = nb;
        nb += (size_t)k.bw * k.bh;
        memcpy(f.qt[c], P.qt[k.tq], sizeof(f.qt[c]));
    }
    f.nblocks = nb;
}
}  // namespace

int vo_jpeg_layout(const uint8_t* d, size_t n, JpegFrame* out, std::string* err) {
    Parsed P;
    const int rc = parse(d, n, &P, false, err);
    if (rc != VO_OK) return rc;
    layout(&P.f, P);
    *out = P.f;
    return VO_OK;
}

int vo_jpeg_entropy(const uint8_t* d, size_t n, JpegFrame* out, int16_t* coef, std::string* err) {
    Parsed P;
    int rc = parse(d, n, &P, false, err);
    if (rc != VO_OK) return rc;
    layout(&P.f, P);
    JpegFrame& f = P.f;
    const int mcux = f.comp[0].bw / f.comp[0].h, mcuy = f.comp[0].bh / f.comp[0].v;
    memset(coef, 0, f.nblocks * 64 * sizeof(int16_t));

    Bits br;
    br.p = d + P.scan;
    br.end = d + n;
    int pred[3] = {0, 0, 0};
    int until_restart = P.ri;
    for (int my = 0; my < mcuy; ++my) {
        for (int mx = 0; mx < mcux; ++mx) {
            // a marker (an early EOI, a stray segment) or the end of the buffer in the middle of the scan: the
            // reader feeds zeros from there, and an MCU decoded out of them is not image data
            if (br.overrun) return fail(err, "entropy-coded data ends before the last MCU");
            if (P.ri && until_restart == 0) {
                if (!br.restart()) return fail(err, "restart marker missing");
                pred[0] = pred[1] = pred[2] = 0;
                until_restart = P.ri;
            }
            for (int c = 0; c < f.nc; ++c) {
                const JpegComp& k = f.comp[c];
                const Huff& hd = P.dc[P.td[c]];
                const Huff& ha = P.ac[P.ta[c]];
                for (int by = 0; by < k.v; ++by)
                    for (int bx = 0; bx < k.h; ++bx) {
                        int16_t* blk = coef + 64 * (k.block0 + (size_t)(my * k.v + by) * k.bw + (mx * k.h + bx));
                        int s = br.decode(hd);
                        if (s < 0 || s > 11) return fail(err, "corrupt entropy-coded data");
                        pred[c] += br.receive_extend(s);  // decode() left >= 41 bits in the buffer
                        blk[0] = (int16_t)pred[c];
                        for (int kk = 1; kk < 64;) {
                            const int rs = br.decode(ha);
                            if (rs < 0) return fail(err, "corrupt entropy-coded data");
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;  // end of block
                                kk += 16;
                                continue;
                            }
                            kk += r;
                            if (kk > 63) return fail(err, "corrupt entropy-coded data");
                            blk[zigzag_natural(kk)] = (int16_t)br.receive_extend(s);
                            ++kk;
                        }
                    }
            }
            --until_restart;
        }
    }
    if (br.overrun) return fail(err, "entropy-coded data ends before the last MCU");
    br.fill();
    if (!br.marker) return fail(err, "entropy-coded data truncated");  // a complete scan is followed by a marker (EOI)
    *out = f;
    return VO_OK;
}

void vo_jpeg_reconstruct(const JpegFrame& f, const int16_t* coef, uint8_t* out, std::vector<uint8_t>* planes) {
    size_t off[3], total = 0;
    for (int c = 0; c < f.nc; ++c) {
        off[c] = total;
        total += (size_t)f.comp[c].bw * 8 * f.comp[c].bh * 8;
    }
    if (planes->size() < total) planes->resize(total);
    for (int c = 0; c < f.nc; ++c) {
        const JpegComp& k = f.comp[c];
        const int pitch = k.bw * 8;
        uint8_t* pl = planes->data() + off[c];
        const uint16_t* q = f.qt[c];
        for (int by = 0; by < k.bh; ++by)
            for (int bx = 0; bx < k.bw; ++bx) {
                const int16_t* blk = coef + 64 * (k.block0 + (size_t)by * k.bw + bx);
                int ws[64];
                for (int col = 0; col < 8; ++col) {  // columns first
                    int in[8], o[8];
                    for (int r = 0; r < 8; ++r) in[r] = vo_jpeg_dequant(blk[r * 8 + col], q[r * 8 + col]);
                    vo_jpeg_idct8(in, o);
                    for (int r = 0; r < 8; ++r) ws[r * 8 + col] = vo_jpeg_descale(o[r], 11);
                }
                for (int r = 0; r < 8; ++r) {
                    int o[8];
                    vo_jpeg_idct8(ws + r * 8, o);
                    uint8_t* dst = pl + (size_t)(by * 8 + r) * pitch + bx * 8;
                    for (int col = 0; col < 8; ++col) dst[col] = (uint8_t)vo_jpeg_range(vo_jpeg_descale(o[col], 18));
                }
            }
    }
    const uint8_t* Y = planes->data() + off[0];
    const int py = f.comp[0].bw * 8;
    if (f.nc == 1) {
        for (int y = 0; y < f.H; ++y) memcpy(out + (size_t)y * f.W, Y + (size_t)y * py, f.W);
        return;
    }
    const int hs = f.hmax == 2, vs = f.vmax == 2;
    const uint8_t* Cb = planes->data() + off[1];
    const uint8_t* Cr = planes->data() + off[2];
    const int pc = f.comp[1].bw * 8, dw = f.comp[1].dw, dh = f.comp[1].dh;
    for (int y = 0; y < f.H; ++y) {
        uint8_t* o = out + (size_t)y * f.W * 3;
        for (int x = 0; x < f.W; ++x) {
            const int cb = vo_jpeg_chroma(Cb, pc, dw, dh, hs, vs, x, y), cr = vo_jpeg_chroma(Cr, pc, dw, dh, hs, vs, x, y);
            int r, g, b;
            vo_jpeg_ycc(Y[(size_t)y * py + x], cb, cr, &r, &g, &b);
            o[3 * x] = (uint8_t)r;
            o[3 * x + 1] = (uint8_t)g;
            o[3 * x + 2] = (uint8_t)b;
        }
    }
}
