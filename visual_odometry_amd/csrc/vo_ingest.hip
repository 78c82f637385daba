// vo_ingest.hip -- frame ingest: what imread + rgb2gray + im2uint8 do in the
// reference's frame loops (main.m:15-30,52-59, utils/dataLoaderKitti.m:29, utils/dataLoaderParking.m:29,
// utils/dataLoaderMalaga.m:29-30), as a decode pool feeding the device.
//
//   host  : PNG (zlib inflate + the five row filters, every colour type, 1..16 bit, non-interlaced) and
//           binary PGM/PPM decoders -> 8-bit gray or interleaved RGB, written straight into pinned memory
//   device: rgb2gray (fp64, the oracle's operation order, so the bytes are identical) fused behind the
//           upload on a copy stream of its own; the consumer's stream only waits on an event
//   pool  : worker threads decode ahead into a ring of slots; vo_ingest_next hands out device images in
//           file order and recycles a slot once the consumer's stream has passed the work that read it.
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "vo_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// decoders (host)
// ------------------------------------------------------------------------------------------------
struct ImgHdr {
    int H = 0, W = 0;
    int channels = 0;   // of the decoded output: 1 or 3
    int bit_depth = 0;  // of the file's samples
    int kind = 0;       // 1 PNG, 2 PNM
    // PNG
    int color_type = 0, interlace = 0;
    // PNM
    int maxval = 0;
    size_t data_off = 0;
};

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
    if (W == 0 || H == 0 || W > 65535u || H > 65535u) {
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
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535 || mv <= 0 || mv > 65535) {
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
        *err = "jpeg files are not decoded by this build";
        return VO_ERR_UNSUPPORTED;
    }
    *err = "unknown image format (PNG and binary PGM/PPM are decoded)";
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

template <int BPP>
void unfilter_paeth(uint8_t* cur, const uint8_t* prev, size_t rb) {
    int a[BPP], c[BPP];
    for (int k = 0; k < BPP; ++k) {  // first pixel: a = c = 0, the predictor is b
        if ((size_t)k >= rb) return;
        cur[k] = (uint8_t)(cur[k] + prev[k]);
        a[k] = cur[k];
        c[k] = prev[k];
    }
    size_t i = BPP;
    for (; i + BPP <= rb; i += BPP) {
#pragma unroll
        for (int k = 0; k < BPP; ++k) {
            const int b = prev[i + k];
            const int db = b - c[k], da = a[k] - c[k];
            const int pa = db < 0 ? -db : db, pb = da < 0 ? -da : da;
            const int dc = db + da, pc = dc < 0 ? -dc : dc;
            int pred = (pb < pa) ? b : a[k];
            const int pm = (pb < pa) ? pb : pa;
            pred = (pc < pm) ? c[k] : pred;
            a[k] = (pred + cur[i + k]) & 255;
            cur[i + k] = (uint8_t)a[k];
            c[k] = b;
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

struct DecodeScratch {  // per worker thread, grow-only
    std::vector<uint8_t> raw;
    std::vector<uint8_t> zero;
    std::vector<uint8_t> file;
};

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

// ------------------------------------------------------------------------------------------------
// rgb2gray (device).  Normative definition (oracle/ingest.py N11): fp64
//   gray = floor(((cr*R + cg*G) + cb*B) + 0.5), no fused multiply-add, clamped to 255.
// Pixel i's channel c is at in[i*sp + c*sc]: interleaved rows (sp 3, sc 1) or MATLAB's planes (sp 1, sc n).
// ------------------------------------------------------------------------------------------------
constexpr double GRAY_R = 0.298936021293775390, GRAY_G = 0.587043074451121360, GRAY_B = 0.114020904255103250;

__device__ inline unsigned gray_of(unsigned r, unsigned g, unsigned b) {
    const double s = __dadd_rn(__dadd_rn(__dmul_rn(GRAY_R, (double)r), __dmul_rn(GRAY_G, (double)g)), __dmul_rn(GRAY_B, (double)b));
    const unsigned v = (unsigned)floor(__dadd_rn(s, 0.5));
    return v > 255u ? 255u : v;
}

__global__ void __launch_bounds__(256) k_rgb2gray(const uint8_t* __restrict__ in, long n, long sp, long sc, int vec,
                                                  uint8_t* __restrict__ out) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;  // four pixels per thread
    const long i0 = q * 4;
    if (i0 >= n) return;
    if (vec && i0 + 4 <= n) {
        unsigned r[4], g[4], b[4];
        if (sp == 3) {  // 12 interleaved bytes = three aligned dwords
            const uint32_t* p = reinterpret_cast<const uint32_t*>(in + i0 * 3);
            const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
            r[0] = w0 & 255; g[0] = (w0 >> 8) & 255; b[0] = (w0 >> 16) & 255;
            r[1] = w0 >> 24; g[1] = w1 & 255; b[1] = (w1 >> 8) & 255;
            r[2] = (w1 >> 16) & 255; g[2] = w1 >> 24; b[2] = w2 & 255;
            r[3] = (w2 >> 8) & 255; g[3] = (w2 >> 16) & 255; b[3] = w2 >> 24;
        } else {  // planes, each dword-aligned
            const uint32_t wr = *reinterpret_cast<const uint32_t*>(in + i0);
            const uint32_t wg = *reinterpret_cast<const uint32_t*>(in + sc + i0);
            const uint32_t wb = *reinterpret_cast<const uint32_t*>(in + 2 * sc + i0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = (wr >> (8 * k)) & 255; g[k] = (wg >> (8 * k)) & 255; b[k] = (wb >> (8 * k)) & 255;
            }
        }
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o |= gray_of(r[k], g[k], b[k]) << (8 * k);
        *reinterpret_cast<uint32_t*>(out + i0) = o;
        return;
    }
    for (long i = i0; i < n && i < i0 + 4; ++i) out[i] = (uint8_t)gray_of(in[i * sp], in[i * sp + sc], in[i * sp + 2 * sc]);
}

int launch_rgb2gray(hipStream_t s, const uint8_t* d_in, long n, int planar, uint8_t* d_out) {
    if (n <= 0) return VO_OK;
    const long sp = planar ? 1 : 3, sc = planar ? n : 1;
    const bool al = ((uintptr_t)d_in % 4 == 0) && ((uintptr_t)d_out % 4 == 0) && (!planar || n % 4 == 0);
    const long quads = (n + 3) / 4;
    hipLaunchKernelGGL(k_rgb2gray, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, d_in, n, sp, sc, al ? 1 : 0, d_out);
    return hipGetLastError() == hipSuccess ? VO_OK : VO_ERR_HIP;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// the decode pool
// ------------------------------------------------------------------------------------------------
struct vo_ingest {
    struct Slot {
        uint8_t* h_pin = nullptr;
        size_t h_cap = 0;
        uint8_t* d_raw = nullptr;
        size_t d_raw_cap = 0;
        uint8_t* d_gray = nullptr;
        size_t d_gray_cap = 0;
        hipEvent_t ready = nullptr;     // copy stream: upload (+ rgb2gray) of this slot done
        hipEvent_t released = nullptr;  // consumer stream: everything that read this slot's image enqueued before it
        bool uploaded = false, has_released = false, gray_pass = false;
        int H = 0, W = 0, ch = 0;
        int status = VO_OK;
        std::string err;
        long frame = -1;
        int state = 0;  // 0 free, 1 being decoded, 2 ready, 3 handed to the consumer
    };
    vo_ctx* ctx = nullptr;
    std::vector<std::string> paths;
    int to_gray = 1;
    hipStream_t copy = nullptr;
    std::vector<Slot> slots;
    std::vector<std::thread> workers;
    std::mutex mu;       // slot states and the two cursors
    std::condition_variable cv;
    std::mutex hip_mu;   // one thread at a time talks to the copy stream / allocator
    long next_decode = 0, next_out = 0, current = -1;
    bool stop = false;
    std::atomic<long long> decode_ns{0};
};

namespace {

int slot_reserve(vo_ingest* g, vo_ingest::Slot* s, size_t raw_bytes, size_t gray_bytes) {
    if (s->h_cap < raw_bytes) {
        if (s->h_pin) (void)hipHostFree(s->h_pin);
        s->h_pin = nullptr;
        s->h_cap = 0;
        if (hipHostMalloc((void**)&s->h_pin, raw_bytes, hipHostMallocDefault) != hipSuccess) return VO_ERR_OOM;
        s->h_cap = raw_bytes;
    }
    if (s->d_raw_cap < raw_bytes) {
        if (s->d_raw) (void)hipFree(s->d_raw);
        s->d_raw = nullptr;
        s->d_raw_cap = 0;
        if (hipMalloc((void**)&s->d_raw, raw_bytes) != hipSuccess) return VO_ERR_OOM;
        s->d_raw_cap = raw_bytes;
    }
    if (s->d_gray_cap < gray_bytes) {
        if (s->d_gray) (void)hipFree(s->d_gray);
        s->d_gray = nullptr;
        s->d_gray_cap = 0;
        if (hipMalloc((void**)&s->d_gray, gray_bytes) != hipSuccess) return VO_ERR_OOM;
        s->d_gray_cap = gray_bytes;
    }
    return VO_OK;
}

void ingest_worker(vo_ingest* g) {
    (void)hipSetDevice(g->ctx->device);
    DecodeScratch sc;
    const long n = (long)g->paths.size();
    const long depth = (long)g->slots.size();
    for (;;) {
        long i;
        vo_ingest::Slot* s;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return g->stop || g->next_decode >= n || g->slots[g->next_decode % depth].state == 0; });
            if (g->stop || g->next_decode >= n) return;
            i = g->next_decode++;
            s = &g->slots[i % depth];
            s->state = 1;
            s->frame = i;
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::string err;
        size_t fn = 0;
        ImgHdr h;
        int rc = read_file(g->paths[i].c_str(), &sc.file, &fn, &err);
        if (rc == VO_OK) rc = image_header(sc.file.data(), fn, &h, &err);
        if (rc == VO_OK) {
            const size_t raw_bytes = (size_t)h.H * h.W * h.channels;
            const bool gray_pass = h.channels == 3 && g->to_gray;
            // the pinned buffer is rewritten below: its last upload must have left it
            if (s->uploaded && hipEventSynchronize(s->ready) != hipSuccess) rc = VO_ERR_HIP;
            if (rc == VO_OK) {
                std::lock_guard<std::mutex> hl(g->hip_mu);
                rc = slot_reserve(g, s, raw_bytes, gray_pass ? (size_t)h.H * h.W : 0);
            }
            if (rc == VO_OK) rc = image_decode(sc.file.data(), fn, h, s->h_pin, &sc, &err);
            if (rc == VO_OK) {
                std::lock_guard<std::mutex> hl(g->hip_mu);
                hipError_t e = hipSuccess;
                if (s->has_released) e = hipStreamWaitEvent(g->copy, s->released, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(s->d_raw, s->h_pin, raw_bytes, hipMemcpyHostToDevice, g->copy);
                if (e == hipSuccess && gray_pass && launch_rgb2gray(g->copy, s->d_raw, (long)h.H * h.W, 0, s->d_gray) != VO_OK)
                    e = hipErrorLaunchFailure;
                if (e == hipSuccess) e = hipEventRecord(s->ready, g->copy);
                if (e != hipSuccess) {
                    rc = VO_ERR_HIP;
                    err = std::string("upload: ") + hipGetErrorString(e);
                } else {
                    s->uploaded = true;
                }
            } else if (err.empty()) {
                err = "device or pinned allocation failed";
            }
            s->H = h.H;
            s->W = h.W;
            s->ch = gray_pass ? 1 : h.channels;
            s->gray_pass = gray_pass;
        }
        g->decode_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> lk(g->mu);
            s->status = rc;
            s->err = g->paths[i] + ": " + err;
            s->state = 2;
        }
        g->cv.notify_all();
    }
}

}  // namespace

extern "C" {

int vo_image_info(const uint8_t* bytes, size_t n, int* H, int* W, int* channels, int* bit_depth) {
    if (!bytes) return VO_ERR_BAD_ARG;
    ImgHdr h;
    std::string err;
    const int rc = image_header(bytes, n, &h, &err);
    if (rc != VO_OK) return rc;
    if (H) *H = h.H;
    if (W) *W = h.W;
    if (channels) *channels = h.channels;
    if (bit_depth) *bit_depth = h.bit_depth;
    return VO_OK;
}

int vo_image_decode(const uint8_t* bytes, size_t n, uint8_t* out, size_t out_cap) {
    if (!bytes || !out) return VO_ERR_BAD_ARG;
    ImgHdr h;
    std::string err;
    int rc = image_header(bytes, n, &h, &err);
    if (rc != VO_OK) return rc;
    if (out_cap < (size_t)h.H * h.W * h.channels) return VO_ERR_BAD_ARG;
    DecodeScratch sc;
    return image_decode(bytes, n, h, out, &sc, &err);
}

int vo_rgb2gray(vo_ctx* ctx, const uint8_t* rgb, int H, int W, int planar, int on_device, uint8_t* gray) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (H < 0 || W < 0 || (planar != 0 && planar != 1)) {
        ctx->last_error = "vo_rgb2gray: bad size or layout";
        return VO_ERR_BAD_ARG;
    }
    const long n = (long)H * W;
    if (n == 0) return VO_OK;
    if (!rgb || !gray) {
        ctx->last_error = "vo_rgb2gray: null image";
        return VO_ERR_BAD_ARG;
    }
    VO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (on_device) {
        if (launch_rgb2gray(ctx->stream, rgb, n, planar, gray) != VO_OK) {
            ctx->last_error = "k_rgb2gray launch failed";
            return VO_ERR_HIP;
        }
        return VO_OK;
    }
    vo_buf* b = &ctx->scratch[16];
    int rc = vo_buf_reserve(ctx, b, (size_t)n * 4 + 256);
    if (rc != VO_OK) return rc;
    uint8_t* d_in = (uint8_t*)b->p;
    uint8_t* d_out = d_in + (((size_t)n * 3 + 255) & ~(size_t)255);
    VO_HIP_CHECK(ctx, hipMemcpyAsync(d_in, rgb, (size_t)n * 3, hipMemcpyHostToDevice, ctx->stream));
    if (launch_rgb2gray(ctx->stream, d_in, n, planar, d_out) != VO_OK) {
        ctx->last_error = "k_rgb2gray launch failed";
        return VO_ERR_HIP;
    }
    VO_HIP_CHECK(ctx, hipMemcpyAsync(gray, d_out, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

int vo_ingest_open(vo_ctx* ctx, const char* const* paths, int n_paths, const vo_ingest_opts* opts, vo_ingest** out) {
    if (!ctx || !out) return VO_ERR_BAD_ARG;
    *out = nullptr;
    if (n_paths < 0 || (n_paths > 0 && !paths)) {
        ctx->last_error = "vo_ingest_open: bad path list";
        return VO_ERR_BAD_ARG;
    }
    int threads = opts ? opts->threads : 0, depth = opts ? opts->depth : 0;
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        if (threads > 8) threads = 8;
        if (threads < 1) threads = 1;
    }
    if (threads > 64) threads = 64;
    if (depth <= 0) depth = 2 * threads;
    if (depth < 2) depth = 2;
    if (depth > 256) depth = 256;
    VO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    vo_ingest* g = new (std::nothrow) vo_ingest();
    if (!g) return VO_ERR_OOM;
    g->ctx = ctx;
    g->to_gray = opts ? (opts->keep_rgb ? 0 : 1) : 1;
    for (int i = 0; i < n_paths; ++i) g->paths.emplace_back(paths[i] ? paths[i] : "");
    g->slots.resize(depth);
    hipError_t e = hipStreamCreateWithFlags(&g->copy, hipStreamNonBlocking);
    for (auto& s : g->slots) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.released, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        ctx->last_error = std::string("vo_ingest_open: ") + hipGetErrorString(e);
        *out = g;
        vo_ingest_close(g);
        *out = nullptr;
        return VO_ERR_HIP;
    }
    if ((long)threads > (long)n_paths) threads = n_paths > 0 ? n_paths : 1;
    for (int t = 0; t < threads; ++t) g->workers.emplace_back(ingest_worker, g);
    *out = g;
    return VO_OK;
}

int vo_ingest_count(vo_ingest* g) { return g ? (int)g->paths.size() : VO_ERR_BAD_ARG; }

int vo_ingest_next(vo_ingest* g, const uint8_t** d_img, int* H, int* W, int* channels, int* index) {
    if (!g) return VO_ERR_BAD_ARG;
    vo_ctx* ctx = g->ctx;
    const long n = (long)g->paths.size(), depth = (long)g->slots.size();
    if (hipSetDevice(ctx->device) != hipSuccess) return VO_ERR_HIP;
    if (g->current >= 0) {  // the consumer has moved on from the image handed out last time
        vo_ingest::Slot* p = &g->slots[g->current % depth];
        VO_HIP_CHECK(ctx, hipEventRecord(p->released, ctx->stream));
        {
            std::lock_guard<std::mutex> lk(g->mu);
            p->has_released = true;
            p->state = 0;
        }
        g->cv.notify_all();
        g->current = -1;
    }
    if (g->next_out >= n) return VO_END;
    vo_ingest::Slot* s = &g->slots[g->next_out % depth];
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv.wait(lk, [&] { return s->state == 2 && s->frame == g->next_out; });
        s->state = 3;
    }
    g->current = g->next_out++;
    if (s->status != VO_OK) {
        ctx->last_error = s->err;
        return s->status;
    }
    VO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, s->ready, 0));
    if (d_img) *d_img = s->gray_pass ? s->d_gray : s->d_raw;
    if (H) *H = s->H;
    if (W) *W = s->W;
    if (channels) *channels = s->ch;
    if (index) *index = (int)g->current;
    return VO_OK;
}

int vo_ingest_read(vo_ingest* g, uint8_t* host_out, size_t cap) {
    if (!g || !host_out) return VO_ERR_BAD_ARG;
    vo_ctx* ctx = g->ctx;
    if (g->current < 0) {
        ctx->last_error = "vo_ingest_read: no current image (call vo_ingest_next first)";
        return VO_ERR_BAD_ARG;
    }
    vo_ingest::Slot* s = &g->slots[g->current % (long)g->slots.size()];
    const size_t nb = (size_t)s->H * s->W * s->ch;
    if (s->status != VO_OK || cap < nb) {
        ctx->last_error = "vo_ingest_read: output too small or the current image failed to decode";
        return VO_ERR_BAD_ARG;
    }
    VO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    VO_HIP_CHECK(ctx, hipMemcpyAsync(host_out, s->gray_pass ? s->d_gray : s->d_raw, nb, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

int vo_ingest_stats(vo_ingest* g, double* decode_seconds_total, int* threads, int* depth) {
    if (!g) return VO_ERR_BAD_ARG;
    if (decode_seconds_total) *decode_seconds_total = 1e-9 * (double)g->decode_ns.load();
    if (threads) *threads = (int)g->workers.size();
    if (depth) *depth = (int)g->slots.size();
    return VO_OK;
}

int vo_ingest_close(vo_ingest* g) {
    if (!g) return VO_OK;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
    }
    g->cv.notify_all();
    for (auto& t : g->workers)
        if (t.joinable()) t.join();
    (void)hipSetDevice(g->ctx->device);
    if (g->copy) (void)hipStreamSynchronize(g->copy);
    (void)hipStreamSynchronize(g->ctx->stream);  // the consumer may still be reading the last image
    for (auto& s : g->slots) {
        if (s.h_pin) (void)hipHostFree(s.h_pin);
        if (s.d_raw) (void)hipFree(s.d_raw);
        if (s.d_gray) (void)hipFree(s.d_gray);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.released) (void)hipEventDestroy(s.released);
    }
    if (g->copy) (void)hipStreamDestroy(g->copy);
    delete g;
    return VO_OK;
}

}  // extern "C"
