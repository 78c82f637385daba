// vo_decode.cpp -- host-side image decoders of the frame ingest: PNG (zlib inflate +
// the five row filters, every colour type, 1..16 bit, non-interlaced) and binary PGM/PPM -> 8-bit gray or
// interleaved RGB; baseline JPEG through vo_jpeg.cpp.  What `imread` (+ im2uint8) does at main.m:15-30,52-59, utils/dataLoaderKitti.m:29,
// utils/dataLoaderParking.m:29.  Plain C++ built with g++ (its code for the serial filter chains runs about
// twice as fast as the HIP host compiler's); no HIP in here.
#include "vo_decode.h"
#include "vo_jpeg.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/vo_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// decoders (host)
// ------------------------------------------------------------------------------------------------
inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

const uint8_t PNG_MAGIC[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

int png_samples_per_pixel(int color_type) {
    switch (color_type) {
        case 0: return 1;  // gray
        case 2: return 3;  // RGB
        case 3: return 1;  // palette index
        case 4: return 2;  // gray + alpha
        case 6: return 4;  // RGB + alpha
    }
    return 0;
}

int png_header(const uint8_t* d, size_t n, ImgHdr* h, std::string* err) {
    if (n < 8 + 25 || be32(d + 8) != 13 || memcmp(d + 12, "IHDR", 4) != 0) {
        *err = "png: no IHDR chunk";
        return VO_ERR_BAD_ARG;
    }
    const uint8_t* p = d + 16;
    if (be32(d + 29) != (uint32_t)crc32(crc32(0, nullptr, 0), d + 12, 17)) {
        *err = "png: IHDR checksum mismatch";
        return VO_ERR_BAD_ARG;
    }
    const uint32_t W = be32(p), H = be32(p + 4);
    h->bit_depth = p[8];
    h->color_type = p[9];
    h->interlace = p[12];
    if (W == 0 || H == 0 || W > 65535u || H > 65535u || (uint64_t)W * H > VO_DEC_MAX_PIXELS) {
        *err = "png: image size out of range";
        return VO_ERR_BAD_ARG;
    }
    h->W = (int)W;
    h->H = (int)H;
    const int spp = png_samples_per_pixel(h->color_type);
    const int bd = h->bit_depth;
    bool ok = spp != 0 && p[10] == 0 && p[11] == 0;
    if (h->color_type == 0) ok = ok && (bd == 1 || bd == 2 || bd == 4 || bd == 8 || bd == 16);
    else if (h->color_type == 3) ok = ok && (bd == 1 || bd == 2 || bd == 4 || bd == 8);
    else ok = ok && (bd == 8 || bd == 16);
    if (!ok) {
        *err = "png: invalid colour type / bit depth";
        return VO_ERR_BAD_ARG;
    }
    if (h->interlace != 0) {
        *err = "png: Adam7-interlaced files are not supported";
        return VO_ERR_UNSUPPORTED;
    }
    h->channels = (h->color_type == 0 || h->color_type == 4) ? 1 : 3;
    h->kind = 1;
    return VO_OK;
}

// next whitespace-delimited unsigned integer of a PNM header ('#' starts a comment)
bool pnm_token(const uint8_t* d, size_t n, size_t* pos, int* out) {
    size_t i = *pos;
    for (;;) {
        while (i < n && (d[i] == ' ' || d[i] == '\t' || d[i] == '\n' || d[i] == '\r')) ++i;
        if (i < n && d[i] == '#') {
            while (i < n && d[i] != '\n') ++i;
            continue;
        }
        break;
    }
    if (i >= n || d[i] < '0' || d[i] > '9') return false;
    long v = 0;
    while (i < n && d[i] >= '0' && d[i] <= '9') {
        v = v * 10 + (d[i] - '0');
        if (v > 1000000) return false;
        ++i;
    }
    *pos = i;
    *out = (int)v;
    return true;
}

int pnm_header(const uint8_t* d, size_t n, ImgHdr* h, std::string* err) {
    size_t pos = 2;
    int W, H, mv;
    if (!pnm_token(d, n, &pos, &W) || !pnm_token(d, n, &pos, &H) || !pnm_token(d, n, &pos, &mv) || pos >= n) {
        *err = "pnm: malformed header";
        return VO_ERR_BAD_ARG;
    }
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535 || (uint64_t)W * H > VO_DEC_MAX_PIXELS || mv <= 0 || mv > 65535) {
        *err = "pnm: size or maxval out of range";
        return VO_ERR_BAD_ARG;
    }
    h->W = W;
    h->H = H;
    h->maxval = mv;
    h->bit_depth = mv < 256 ? 8 : 16;
    h->channels = d[1] == '5' ? 1 : 3;
    h->data_off = pos + 1;  // exactly one whitespace byte after maxval
    h->kind = 2;
    const size_t need = (size_t)W * H * h->channels * (mv < 256 ? 1 : 2);
    if (n - h->data_off < need) {
        *err = "pnm: pixel data truncated";
        return VO_ERR_BAD_ARG;
    }
    return VO_OK;
}

int image_header(const uint8_t* d, size_t n, ImgHdr* h, std::string* err) {
    if (n >= 8 && memcmp(d, PNG_MAGIC, 8) == 0) return png_header(d, n, h, err);
    if (n >= 3 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) return pnm_header(d, n, h, err);
    if (n >= 2 && d[0] == 0xff && d[1] == 0xd8) {
        const int rc = vo_jpeg_header(d, n, &h->W, &h->H, &h->channels, err);
        h->bit_depth = 8;
        h->kind = 3;
        if (rc == VO_OK && (uint64_t)h->W * h->H > VO_DEC_MAX_PIXELS) {
            *err = "jpeg: image size out of range";
            return VO_ERR_BAD_ARG;
        }
        return rc;
    }
    *err = "unknown image format (PNG, baseline JPEG and binary PGM/PPM are decoded)";
    return VO_ERR_UNSUPPORTED;
}

// im2uint8 of a uint16 sample: round(x / 257)
inline uint8_t u16_to_u8(unsigned v) { return (uint8_t)((v + 128u) / 257u); }

// Row filters (ISO/IEC 15948 section 9): x = the byte being rebuilt, a = the byte one pixel to its left,
// b = the byte above, c = the byte above-left.  The Paeth predictor picks whichever of a, b, c is nearest
// to a + b - c; with pa = |b - c|, pb = |a - c|, pc = |(b - c) + (a - c)| that is three differences per byte.
// The loops carry a and c in registers, one chain per byte of the pixel.
template <int BPP>
void unfilter_sub(uint8_t* cur, size_t rb) {
    for (size_t i = BPP; i < rb; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - BPP]);
}

template <int BPP>
void unfilter_avg(uint8_t* cur, const uint8_t* prev, size_t rb) {
    for (size_t i = 0; i < (size_t)BPP && i < rb; ++i) cur[i] = (uint8_t)(cur[i] + (prev[i] >> 1));
    for (size_t i = BPP; i < rb; ++i) cur[i] = (uint8_t)(cur[i] + ((cur[i - BPP] + prev[i]) >> 1));
}

inline int paeth_step(int a, int c, int b, int x) {
    const int db = b - c, pa = abs(db);  // off the chain: needs only the row above
    const int da = a - c;
    const int pb = abs(da), pc = abs(db + da);
    const bool use_a = (pa <= pb) & (pa <= pc);
    const int bc = (pb <= pc) ? b : c;
    return ((use_a ? a : bc) + x) & 255;
}

// one scalar chain per byte of the pixel, each run over the whole row (the row sits in L1)
template <int BPP>
void unfilter_paeth(uint8_t* cur, const uint8_t* prev, size_t rb) {
    for (int k = 0; k < BPP; ++k) {
        int a = 0, c = 0;
        for (size_t i = k; i < rb; i += BPP) {
            const int b = prev[i];
            a = paeth_step(a, c, b, cur[i]);
            cur[i] = (uint8_t)a;
            c = b;
        }
    }
}

template <int BPP>
int unfilter_row_t(int ft, uint8_t* cur, const uint8_t* prev, size_t rb) {
    switch (ft) {
        case 0: return 0;
        case 1: unfilter_sub<BPP>(cur, rb); return 0;
        case 2:
            for (size_t i = 0; i < rb; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
            return 0;
        case 3: unfilter_avg<BPP>(cur, prev, rb); return 0;
        case 4: unfilter_paeth<BPP>(cur, prev, rb); return 0;
    }
    return -1;
}

// undo the row filter of `cur` (rb bytes, a multiple of bpp) in place; prev = the rebuilt row above (zeros above row 0)
int png_unfilter_row(int ft, uint8_t* cur, const uint8_t* prev, size_t rb, size_t bpp) {
    switch (bpp) {
        case 1: return unfilter_row_t<1>(ft, cur, prev, rb);
        case 2: return unfilter_row_t<2>(ft, cur, prev, rb);
        case 3: return unfilter_row_t<3>(ft, cur, prev, rb);
        case 4: return unfilter_row_t<4>(ft, cur, prev, rb);
        case 6: return unfilter_row_t<6>(ft, cur, prev, rb);
        case 8: return unfilter_row_t<8>(ft, cur, prev, rb);
    }
    return -1;
}

int png_decode(const uint8_t* d, size_t n, const ImgHdr& h, uint8_t* out, DecodeScratch* sc, std::string* err) {
    const int spp = png_samples_per_pixel(h.color_type), bd = h.bit_depth;
    const size_t rb = ((size_t)h.W * spp * bd + 7) / 8;
    const size_t bpp = (size_t)(spp * bd + 7) / 8;
    const size_t raw_len = (size_t)h.H * (rb + 1);
    if (sc->raw.size() < raw_len) sc->raw.resize(raw_len);
    uint8_t* raw = sc->raw.data();

    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) {
        *err = "png: inflateInit failed";
        return VO_ERR_OOM;
    }
    zs.next_out = raw;
    zs.avail_out = (uInt)raw_len;
    uint8_t pal[256 * 3];
    int npal = 0;
    bool ended = false, done = false;
    size_t pos = 8;
    int zrc = Z_OK;
    while (pos + 12 <= n && !ended) {
        const uint32_t len = be32(d + pos);
        const uint8_t* type = d + pos + 4;
        if ((size_t)len > n - pos - 12) {
            inflateEnd(&zs);
            *err = "png: chunk runs past the end of the file";
            return VO_ERR_BAD_ARG;
        }
        const uint8_t* body = d + pos + 8;
        if (memcmp(type, "IDAT", 4) == 0) {
            if (be32(body + len) != (uint32_t)crc32(crc32(0, nullptr, 0), type, len + 4)) {
                inflateEnd(&zs);
                *err = "png: IDAT checksum mismatch";
                return VO_ERR_BAD_ARG;
            }
            if (!done && len) {
                zs.next_in = const_cast<Bytef*>(body);
                zs.avail_in = len;
                zrc = inflate(&zs, Z_NO_FLUSH);
                if (zrc == Z_STREAM_END) done = true;
                else if (zrc != Z_OK && !(zrc == Z_BUF_ERROR && zs.avail_out == 0)) {
                    inflateEnd(&zs);
                    *err = "png: corrupt compressed stream";
                    return VO_ERR_BAD_ARG;
                }
            }
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) {
                inflateEnd(&zs);
                *err = "png: bad palette";
                return VO_ERR_BAD_ARG;
            }
            memcpy(pal, body, len);
            npal = (int)(len / 3);
        } else if (memcmp(type, "IEND", 4) == 0) {
            ended = true;
        }
        pos += 12 + (size_t)len;
    }
    const size_t got = raw_len - zs.avail_out;
    inflateEnd(&zs);
    if (got != raw_len) {
        *err = "png: pixel data truncated";
        return VO_ERR_BAD_ARG;
    }
    if (h.color_type == 3 && npal == 0) {
        *err = "png: palette image without PLTE";
        return VO_ERR_BAD_ARG;
    }

    const int W = h.W, och = h.channels;
    if (sc->zero.size() < rb) sc->zero.assign(rb, 0);
    const uint8_t* prev = sc->zero.data();
    for (int y = 0; y < h.H; ++y) {
        uint8_t* row = raw + (size_t)y * (rb + 1);
        if (png_unfilter_row(row[0], row + 1, prev, rb, bpp) != 0) {
            *err = "png: unknown row filter";
            return VO_ERR_BAD_ARG;
        }
        const uint8_t* s = row + 1;
        prev = s;
        uint8_t* o = out + (size_t)y * W * och;
        if (bd == 8 && (h.color_type == 0 || h.color_type == 2)) {
            memcpy(o, s, (size_t)W * och);
        } else if (h.color_type == 3) {
            const int mask = (1 << bd) - 1;
            for (int x = 0; x < W; ++x) {
                const int bit = x * bd;
                int idx = (s[bit >> 3] >> (8 - bd - (bit & 7))) & mask;
                if (idx >= npal) {
                    *err = "png: palette index out of range";
                    return VO_ERR_BAD_ARG;
                }
                o[3 * x] = pal[3 * idx];
                o[3 * x + 1] = pal[3 * idx + 1];
                o[3 * x + 2] = pal[3 * idx + 2];
            }
        } else if (bd < 8) {  // gray 1 / 2 / 4 bit, scaled to the full range
            const int mask = (1 << bd) - 1, mul = 255 / mask;
            for (int x = 0; x < W; ++x) {
                const int bit = x * bd;
                o[x] = (uint8_t)(((s[bit >> 3] >> (8 - bd - (bit & 7))) & mask) * mul);
            }
        } else if (bd == 8) {  // alpha dropped
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < och; ++c) o[x * och + c] = s[x * spp + c];
        } else {  // 16 bit, big-endian samples
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < och; ++c) {
                    const uint8_t* q = s + 2 * ((size_t)x * spp + c);
                    o[x * och + c] = u16_to_u8(((unsigned)q[0] << 8) | q[1]);
                }
        }
    }
    return VO_OK;
}

int pnm_decode(const uint8_t* d, size_t n, const ImgHdr& h, uint8_t* out) {
    const size_t cnt = (size_t)h.W * h.H * h.channels;
    const uint8_t* s = d + h.data_off;
    if (h.maxval == 255) {
        memcpy(out, s, cnt);
    } else if (h.maxval < 256) {  // rescale to 0..255, round to nearest
        for (size_t i = 0; i < cnt; ++i) {
            const unsigned v = s[i] > h.maxval ? h.maxval : s[i];
            out[i] = (uint8_t)((v * 255u * 2u + h.maxval) / (2u * h.maxval));
        }
    } else {
        for (size_t i = 0; i < cnt; ++i) {
            unsigned v = ((unsigned)s[2 * i] << 8) | s[2 * i + 1];
            if (v > (unsigned)h.maxval) v = h.maxval;
            out[i] = (uint8_t)(((unsigned long long)v * 255u * 2u + h.maxval) / (2ull * h.maxval));
        }
    }
    return VO_OK;
}

int image_decode(const uint8_t* d, size_t n, const ImgHdr& h, uint8_t* out, DecodeScratch* sc, std::string* err) {
    if (h.kind == 1) return png_decode(d, n, h, out, sc, err);
    if (h.kind == 2) return pnm_decode(d, n, h, out);
    if (h.kind == 3) {
        JpegFrame f{};
        int rc = vo_jpeg_layout(d, n, &f, err);
        if (rc != VO_OK) return rc;
        // `out` was sized from the header the caller parsed: the frame decoded here must be that frame
        if (f.W != h.W || f.H != h.H || f.nc != h.channels || (uint64_t)f.W * f.H > VO_DEC_MAX_PIXELS) {
            *err = "jpeg: frame geometry differs from the header";
            return VO_ERR_BAD_ARG;
        }
        if (sc->coef.size() < f.nblocks * 64) sc->coef.resize(f.nblocks * 64);
        rc = vo_jpeg_entropy(d, n, &f, sc->coef.data(), err);
        if (rc != VO_OK) return rc;
        vo_jpeg_reconstruct(f, sc->coef.data(), out, &sc->planes);
        return VO_OK;
    }
    *err = "no decoder";
    return VO_ERR_UNSUPPORTED;
}

int read_file(const char* path, std::vector<uint8_t>* buf, size_t* n, std::string* err) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        *err = std::string("cannot open ") + path;
        return VO_ERR_BAD_ARG;
    }
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (len <= 0) {
        fclose(f);
        *err = std::string("empty file ") + path;
        return VO_ERR_BAD_ARG;
    }
    if (buf->size() < (size_t)len) buf->resize((size_t)len);
    const size_t got = fread(buf->data(), 1, (size_t)len, f);
    fclose(f);
    if (got != (size_t)len) {
        *err = std::string("short read on ") + path;
        return VO_ERR_BAD_ARG;
    }
    *n = got;
    return VO_OK;
}

}  // namespace

int vo_dec_header(const uint8_t* d, size_t n, ImgHdr* h, std::string* err) { return image_header(d, n, h, err); }
int vo_dec_decode(const uint8_t* d, size_t n, const ImgHdr& h, uint8_t* out, DecodeScratch* sc, std::string* err) {
    try {
        return image_decode(d, n, h, out, sc, err);
    } catch (const std::bad_alloc&) {  // scratch vectors: nothing crosses the C ABI as an exception
        *err = "out of host memory while decoding";
        return VO_ERR_OOM;
    }
}
int vo_dec_read_file(const char* path, std::vector<uint8_t>* buf, size_t* n, std::string* err) {
    try {
        return read_file(path, buf, n, err);
    } catch (const std::bad_alloc&) {
        *err = "out of host memory while reading";
        return VO_ERR_OOM;
    }
}
