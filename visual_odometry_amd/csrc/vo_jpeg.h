// vo_jpeg.h -- baseline JPEG decoding for the frame ingest (the Malaga extract's files,
// utils/dataLoaderMalaga.m:29-30, main.m:20-25,54-56), split where the work changes nature:
//   vo_jpeg_entropy      host: markers + Huffman decoding -> quantised coefficient blocks (serial bit stream)
//   vo_jpeg_reconstruct  host back half: dequantise + 8x8 inverse DCT + chroma upsampling + YCbCr -> RGB
// The decode pool runs the back half on the device instead (k_jpeg_idct / k_jpeg_pixels in vo_ingest.hip),
// from the same JpegFrame; both follow definition N12 (oracle/ingest.py) and agree bit for bit.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

struct JpegComp {
    int h = 1, v = 1;      // sampling factors
    int tq = 0;            // quantisation table
    int bw = 0, bh = 0;    // blocks per row / column of the component's plane (padded to whole MCUs)
    int dw = 0, dh = 0;    // the component's true size: ceil(W * h / hmax), ceil(H * v / vmax)
    size_t block0 = 0;     // index of the component's first block in the coefficient array (row-major blocks)
};

struct JpegFrame {
    int W = 0, H = 0, nc = 0;  // nc = 1 (gray) or 3 (YCbCr)
    int hmax = 1, vmax = 1;
    JpegComp comp[3];
    uint16_t qt[3][64];        // per component, natural (row-major) order
    size_t nblocks = 0;
};

// sizes only (first SOF marker); VO_ERR_UNSUPPORTED for what is not decoded (progressive, 12-bit, CMYK ...)
int vo_jpeg_header(const uint8_t* d, size_t n, int* W, int* H, int* nc, std::string* err);
// the frame's geometry and quantisation tables (everything up to the scan); f->nblocks sizes the coefficient array
int vo_jpeg_layout(const uint8_t* d, size_t n, JpegFrame* f, std::string* err);
// coef: room for nblocks x 64 int16 (vo_jpeg_layout); filled in natural order, NOT dequantised
int vo_jpeg_entropy(const uint8_t* d, size_t n, JpegFrame* f, int16_t* coef, std::string* err);
// out: H x W x nc uint8 interleaved rows; planes: scratch (grow-only)
void vo_jpeg_reconstruct(const JpegFrame& f, const int16_t* coef, uint8_t* out, std::vector<uint8_t>* planes);

// N12 pieces shared with the device code (plain integer arithmetic, usable from HIP through the macro)
#ifdef __HIPCC__
#define VO_JPEG_FN __host__ __device__ inline
#else
#define VO_JPEG_FN inline
#endif

// clamp as the decoders' range table does it: index taken modulo 1024 around a centre of 128
VO_JPEG_FN int vo_jpeg_range(int x) {
    const int i = x & 1023;
    return i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896));
}

// one 8-point pass of the 13-bit fixed-point Loeffler-Ligtenberg-Moschytz inverse DCT; o[k] are the
// un-descaled sums (the caller descales: 11 bits after the column pass, 18 after the row pass).  The sums are
// formed in 32-bit two's-complement arithmetic that wraps (unsigned), so a damaged file's out-of-range
// coefficients give some defined pixel values instead of a signed overflow; a valid file never wraps.
VO_JPEG_FN void vo_jpeg_idct8(const int in[8], int o[8]) {
    typedef uint32_t U;
    U z2 = (U)in[2], z3 = (U)in[6];
    U z1 = (z2 + z3) * (U)4433;
    const U t2 = z1 + z3 * (U)-15137, t3 = z1 + z2 * (U)6270;
    z2 = (U)in[0];
    z3 = (U)in[4];
    const U t0 = (z2 + z3) * (U)8192, t1 = (z2 - z3) * (U)8192;
    const U t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    U o0 = (U)in[7], o1 = (U)in[5], o2 = (U)in[3], o3 = (U)in[1];
    z1 = o0 + o3;
    z2 = o1 + o2;
    z3 = o0 + o2;
    U z4 = o1 + o3;
    const U z5 = (z3 + z4) * (U)9633;
    o0 *= (U)2446;
    o1 *= (U)16819;
    o2 *= (U)25172;
    o3 *= (U)12299;
    z1 *= (U)-7373;
    z2 *= (U)-20995;
    z3 *= (U)-16069;
    z4 *= (U)-3196;
    z3 += z5;
    z4 += z5;
    o0 += z1 + z3;
    o1 += z2 + z4;
    o2 += z2 + z3;
    o3 += z1 + z4;
    o[0] = (int)(t10 + o3);
    o[7] = (int)(t10 - o3);
    o[1] = (int)(t11 + o2);
    o[6] = (int)(t11 - o2);
    o[2] = (int)(t12 + o1);
    o[5] = (int)(t12 - o1);
    o[3] = (int)(t13 + o0);
    o[4] = (int)(t13 - o0);
}

// round-to-nearest right shift of such a sum (the rounding add wraps like the sums do)
VO_JPEG_FN int vo_jpeg_descale(int x, int bits) { return (int)((uint32_t)x + (1u << (bits - 1))) >> bits; }

// a coefficient times its quantiser step, wrapped the same way
VO_JPEG_FN int vo_jpeg_dequant(int c, int q) { return (int)((uint32_t)c * (uint32_t)q); }

// YCbCr -> RGB with 16-bit fixed-point coefficients (1.402, 0.34414, 0.71414, 1.772), rounded to nearest
VO_JPEG_FN void vo_jpeg_ycc(int y, int cb, int cr, int* r, int* g, int* b) {
    const int u = cb - 128, v = cr - 128;
    const int rr = y + ((91881 * v + 32768) >> 16);
    const int gg = y + ((-22554 * u - 46802 * v + 32768) >> 16);
    const int bb = y + ((116130 * u + 32768) >> 16);
    *r = rr < 0 ? 0 : (rr > 255 ? 255 : rr);
    *g = gg < 0 ? 0 : (gg > 255 ? 255 : gg);
    *b = bb < 0 ? 0 : (bb > 255 ? 255 : bb);
}

// chroma sample for output pixel (x, y) of a plane subsampled 2:1 horizontally (hs) and/or vertically (vs):
// triangle filter ("fancy" upsampling) with the neighbour indices clamped to the component's true size
VO_JPEG_FN int vo_jpeg_chroma(const uint8_t* p, int pitch, int dw, int dh, int hs, int vs, int x, int y) {
    if (!hs && !vs) return p[(size_t)y * pitch + x];
    if (hs && !vs) {
        const int i = x >> 1;
        const uint8_t* row = p + (size_t)y * pitch;
        if (x & 1) {
            const int nb = i + 1 < dw ? i + 1 : dw - 1;
            return (3 * row[i] + row[nb] + 2) >> 2;
        }
        const int nb = i > 0 ? i - 1 : 0;
        return (3 * row[i] + row[nb] + 1) >> 2;
    }
    const int j = y >> 1;
    int jn = (y & 1) ? j + 1 : j - 1;
    jn = jn < 0 ? 0 : (jn > dh - 1 ? dh - 1 : jn);
    const uint8_t* r0 = p + (size_t)j * pitch;
    const uint8_t* r1 = p + (size_t)jn * pitch;
    if (!hs) return (3 * r0[x] + r1[x] + ((y & 1) ? 2 : 1)) >> 2;
    const int i = x >> 1;
    int in_ = (x & 1) ? i + 1 : i - 1;
    in_ = in_ < 0 ? 0 : (in_ > dw - 1 ? dw - 1 : in_);
    const int cs = 3 * r0[i] + r1[i], ns = 3 * r0[in_] + r1[in_];
    return (3 * cs + ns + ((x & 1) ? 7 : 8)) >> 4;
}
