// vo_ingest.hip -- frame ingest: what imread + rgb2gray + im2uint8 do in the
// reference's frame loops (main.m:15-30,52-59, utils/dataLoaderKitti.m:29, utils/dataLoaderParking.m:29,
// utils/dataLoaderMalaga.m:29-30), as a decode pool feeding the device.
//
//   host  : the decoders of vo_decode.cpp (PNG, binary PGM/PPM) -> 8-bit gray or interleaved RGB, written
//           straight into pinned memory
//   device: rgb2gray (fp64, the oracle's operation order, so the bytes are identical) fused behind the
//           upload on a copy stream of its own; the consumer's stream only waits on an event
//   pool  : worker threads decode ahead into a ring of slots; vo_ingest_next hands out device images in
//           file order and recycles a slot once the consumer's stream has passed the work that read it.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "vo_common.h"
#include "vo_decode.h"
#include "vo_jpeg.h"

namespace {

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

// ------------------------------------------------------------------------------------------------
// JPEG back half (device): what follows the serial bit stream.  The host's Huffman decoder leaves quantised
// coefficient blocks in pinned memory; dequantisation, the 8x8 inverse DCT, chroma upsampling, YCbCr -> RGB
// and rgb2gray run here, with the arithmetic of definition N12 shared with the host decoder through
// vo_jpeg.h, so both produce the same bytes.
// ------------------------------------------------------------------------------------------------
struct JpegDev {
    int W, H, nc, hs, vs;
    int bw[3], bh[3], dw[3], dh[3];
    unsigned block0[3], plane0[3];  // first block / first plane byte of each component
    unsigned nblocks;
    uint16_t qt[3][64];
};

// 8 threads per block: thread j does column j, then row j; 32 blocks per workgroup.
// LDS image of a block: element (r, c) at r*9 + c, blocks 72 words apart: both passes are conflict-free.
__global__ void __launch_bounds__(256) k_jpeg_idct(const int16_t* __restrict__ coef, JpegDev a, uint8_t* __restrict__ planes) {
    __shared__ int ws[32 * 72];
    const int t = threadIdx.x, lb = t >> 3, j = t & 7;
    const unsigned b = blockIdx.x * 32u + (unsigned)lb;
    const bool live = b < a.nblocks;
    const int c = (a.nc == 3 && b >= a.block0[2]) ? 2 : ((a.nc == 3 && b >= a.block0[1]) ? 1 : 0);
    int* w = ws + lb * 72;
    if (live) {
        const int16_t* blk = coef + (size_t)b * 64;
        int in[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = vo_jpeg_dequant(blk[r * 8 + j], a.qt[c][r * 8 + j]);
        vo_jpeg_idct8(in, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r * 9 + j] = vo_jpeg_descale(o[r], 11);
    }
    __syncthreads();
    if (!live) return;
    int in[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = w[j * 9 + k];
    vo_jpeg_idct8(in, o);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lo |= (uint32_t)vo_jpeg_range(vo_jpeg_descale(o[k], 18)) << (8 * k);
        hi |= (uint32_t)vo_jpeg_range(vo_jpeg_descale(o[k + 4], 18)) << (8 * k);
    }
    const unsigned rel = b - a.block0[c];
    const unsigned by = rel / (unsigned)a.bw[c], bx = rel - by * (unsigned)a.bw[c];
    const size_t pitch = (size_t)a.bw[c] * 8;
    uint2* dst = reinterpret_cast<uint2*>(planes + a.plane0[c] + ((size_t)by * 8 + j) * pitch + (size_t)bx * 8);
    *dst = make_uint2(lo, hi);
}

// four pixels of a row per thread: Y + upsampled chroma -> RGB (-> gray when to_gray); gray files: the Y plane
__global__ void __launch_bounds__(256) k_jpeg_pixels(const uint8_t* __restrict__ planes, JpegDev a, int to_gray, int vec,
                                                     uint8_t* __restrict__ out) {
    const int qw = (a.W + 3) / 4;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long)qw * a.H) return;
    const int y = (int)(q / qw), x0 = (int)(q - (long)y * qw) * 4;
    const uint8_t* Y = planes + a.plane0[0] + (size_t)y * a.bw[0] * 8;
    const int och = (a.nc == 1 || to_gray) ? 1 : 3;
    uint8_t* o = out + ((size_t)y * a.W + x0) * och;
    unsigned px[12];
    const int cnt = a.W - x0 < 4 ? a.W - x0 : 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + (k < cnt ? k : cnt - 1);
        if (a.nc == 1) {
            px[k] = Y[x];
            continue;
        }
        const int pc = a.bw[1] * 8;
        const int cb = vo_jpeg_chroma(planes + a.plane0[1], pc, a.dw[1], a.dh[1], a.hs, a.vs, x, y);
        const int cr = vo_jpeg_chroma(planes + a.plane0[2], pc, a.dw[2], a.dh[2], a.hs, a.vs, x, y);
        int r, g, b;
        vo_jpeg_ycc(Y[x], cb, cr, &r, &g, &b);
        if (to_gray) px[k] = gray_of((unsigned)r, (unsigned)g, (unsigned)b);
        else {
            px[3 * k] = (unsigned)r;
            px[3 * k + 1] = (unsigned)g;
            px[3 * k + 2] = (unsigned)b;
        }
    }
    if (vec && cnt == 4) {
        uint32_t* ow = reinterpret_cast<uint32_t*>(o);
#pragma unroll
        for (int wd = 0; wd < 3; ++wd)
            if (wd < och) ow[wd] = px[4 * wd] | (px[4 * wd + 1] << 8) | (px[4 * wd + 2] << 16) | (px[4 * wd + 3] << 24);
        return;
    }
    for (int i = 0; i < cnt * och; ++i) o[i] = (uint8_t)px[i];
}

size_t jpeg_plane_bytes(const JpegFrame& f) {
    size_t t = 0;
    for (int c = 0; c < f.nc; ++c) t += (size_t)f.comp[c].bw * 8 * f.comp[c].bh * 8;
    return t;
}

int launch_jpeg_back_half(hipStream_t s, const JpegFrame& f, const int16_t* d_coef, uint8_t* d_planes, int to_gray,
                          uint8_t* d_out) {
    JpegDev a;
    memset(&a, 0, sizeof(a));
    a.W = f.W; a.H = f.H; a.nc = f.nc; a.hs = f.hmax == 2; a.vs = f.vmax == 2;
    unsigned p0 = 0;
    for (int c = 0; c < 3; ++c) {
        const JpegComp& k = f.comp[c < f.nc ? c : 0];
        a.bw[c] = k.bw; a.bh[c] = k.bh; a.dw[c] = k.dw; a.dh[c] = k.dh;
        a.block0[c] = c < f.nc ? (unsigned)k.block0 : (unsigned)f.nblocks;
        a.plane0[c] = c < f.nc ? p0 : 0;
        if (c < f.nc) p0 += (unsigned)k.bw * 8u * (unsigned)k.bh * 8u;
        memcpy(a.qt[c], f.qt[c < f.nc ? c : 0], sizeof(a.qt[c]));
    }
    a.nblocks = (unsigned)f.nblocks;
    hipLaunchKernelGGL(k_jpeg_idct, dim3((a.nblocks + 31) / 32), dim3(256), 0, s, d_coef, a, d_planes);
    const long quads = (long)((f.W + 3) / 4) * f.H;
    const int vec = f.W % 4 == 0 && (uintptr_t)d_out % 4 == 0;
    hipLaunchKernelGGL(k_jpeg_pixels, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, d_planes, a, to_gray, vec, d_out);
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
        uint8_t* d_gray = nullptr;  // what a device pass (rgb2gray, the JPEG back half) writes
        size_t d_gray_cap = 0;
        uint8_t* d_aux = nullptr;   // JPEG component planes
        size_t d_aux_cap = 0;
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
    long oldest_held = -1;  // frames oldest_held .. current are in the consumer's hands (vo_ingest_set_hold)
    int hold = 1;
    bool stop = false;
    std::atomic<long long> decode_ns{0};
};

namespace {

int slot_reserve(vo_ingest* g, vo_ingest::Slot* s, size_t raw_bytes, size_t gray_bytes, size_t aux_bytes = 0) {
    if (s->d_aux_cap < aux_bytes) {
        if (s->d_aux) (void)hipFree(s->d_aux);
        s->d_aux = nullptr;
        s->d_aux_cap = 0;
        if (hipMalloc((void**)&s->d_aux, aux_bytes) != hipSuccess) return VO_ERR_OOM;
        s->d_aux_cap = aux_bytes;
    }
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
        int rc = vo_dec_read_file(g->paths[i].c_str(), &sc.file, &fn, &err);
        if (rc == VO_OK) rc = vo_dec_header(sc.file.data(), fn, &h, &err);
        if (rc == VO_OK && h.kind == 3) {
            // JPEG: the bit stream is decoded here, everything after it on the device
            JpegFrame f{};
            rc = vo_jpeg_layout(sc.file.data(), fn, &f, &err);
            // the sizes below come from this second parse: hold it to the header's (pixel cap included)
            if (rc == VO_OK && (f.W != h.W || f.H != h.H || f.nc != h.channels || (uint64_t)f.W * f.H > VO_DEC_MAX_PIXELS)) {
                rc = VO_ERR_BAD_ARG;
                err = "jpeg: frame geometry differs from the header";
            }
            if (rc != VO_OK) f = JpegFrame{};
            const size_t raw_bytes = f.nblocks * 64 * sizeof(int16_t);
            const int och = (f.nc == 1 || g->to_gray) ? 1 : 3;
            if (rc == VO_OK && s->uploaded && hipEventSynchronize(s->ready) != hipSuccess) rc = VO_ERR_HIP;
            if (rc == VO_OK) {
                std::lock_guard<std::mutex> hl(g->hip_mu);
                rc = slot_reserve(g, s, raw_bytes, (size_t)f.H * f.W * och, jpeg_plane_bytes(f));
                if (rc != VO_OK) err = "device or pinned allocation failed";
            }
            if (rc == VO_OK) rc = vo_jpeg_entropy(sc.file.data(), fn, &f, reinterpret_cast<int16_t*>(s->h_pin), &err);
            if (rc == VO_OK) {
                std::lock_guard<std::mutex> hl(g->hip_mu);
                hipError_t e = hipSuccess;
                if (s->has_released) e = hipStreamWaitEvent(g->copy, s->released, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(s->d_raw, s->h_pin, raw_bytes, hipMemcpyHostToDevice, g->copy);
                if (e == hipSuccess && launch_jpeg_back_half(g->copy, f, reinterpret_cast<const int16_t*>(s->d_raw), s->d_aux,
                                                             g->to_gray, s->d_gray) != VO_OK)
                    e = hipErrorLaunchFailure;
                if (e == hipSuccess) e = hipEventRecord(s->ready, g->copy);
                if (e != hipSuccess) {
                    rc = VO_ERR_HIP;
                    err = std::string("upload: ") + hipGetErrorString(e);
                } else {
                    s->uploaded = true;
                }
            }
            s->H = f.H;
            s->W = f.W;
            s->ch = och;
            s->gray_pass = true;  // the image handed out is the device pass's output
        } else if (rc == VO_OK) {
            const size_t raw_bytes = (size_t)h.H * h.W * h.channels;
            const bool gray_pass = h.channels == 3 && g->to_gray;
            // the pinned buffer is rewritten below: its last upload must have left it
            if (s->uploaded && hipEventSynchronize(s->ready) != hipSuccess) rc = VO_ERR_HIP;
            if (rc == VO_OK) {
                std::lock_guard<std::mutex> hl(g->hip_mu);
                rc = slot_reserve(g, s, raw_bytes, gray_pass ? (size_t)h.H * h.W : 0);
            }
            if (rc == VO_OK) rc = vo_dec_decode(sc.file.data(), fn, h, s->h_pin, &sc, &err);
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
    const int rc = vo_dec_header(bytes, n, &h, &err);
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
    int rc = vo_dec_header(bytes, n, &h, &err);
    if (rc != VO_OK) return rc;
    if (out_cap < (size_t)h.H * h.W * h.channels) return VO_ERR_BAD_ARG;
    DecodeScratch sc;
    return vo_dec_decode(bytes, n, h, out, &sc, &err);
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
    if (depth < 3) depth = 3;   // two frames in the consumer's hands (vo_ingest_set_hold) and one being decoded
    if (depth > 256) depth = 256;
    VO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    vo_ingest* g = new (std::nothrow) vo_ingest();
    if (!g) return VO_ERR_OOM;
    g->ctx = ctx;
    g->to_gray = opts ? (opts->keep_rgb ? 0 : 1) : 1;
    try {
        for (int i = 0; i < n_paths; ++i) g->paths.emplace_back(paths[i] ? paths[i] : "");
        g->slots.resize(depth);
    } catch (const std::exception&) {
        delete g;
        ctx->last_error = "vo_ingest_open: out of host memory";
        return VO_ERR_OOM;
    }
    vo_int_register_streams(ctx->device, nullptr);   // a copy stream on this GPU: no high-priority lane beside it (vo_ctx_lane_priority)
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
    try {
        for (int t = 0; t < threads; ++t) g->workers.emplace_back(ingest_worker, g);
    } catch (const std::exception&) {  // no more threads to be had: run with the ones that started, if any
        if (g->workers.empty()) {
            ctx->last_error = "vo_ingest_open: cannot start a decode thread";
            vo_ingest_close(g);
            return VO_ERR_OOM;
        }
    }
    *out = g;
    return VO_OK;
}

int vo_ingest_count(vo_ingest* g) { return g ? (int)g->paths.size() : VO_ERR_BAD_ARG; }

int vo_ingest_next(vo_ingest* g, const uint8_t** d_img, int* H, int* W, int* channels, int* index) {
    if (!g) return VO_ERR_BAD_ARG;
    vo_ctx* ctx = g->ctx;
    const long n = (long)g->paths.size(), depth = (long)g->slots.size();
    if (hipSetDevice(ctx->device) != hipSuccess) return VO_ERR_HIP;
    // the consumer has moved on from the oldest image(s) in its hands: at most `hold` stay valid
    while (g->current >= 0 && g->current - g->oldest_held + 1 >= g->hold) {
        vo_ingest::Slot* p = &g->slots[g->oldest_held % depth];
        VO_HIP_CHECK(ctx, hipEventRecord(p->released, ctx->stream));
        {
            std::lock_guard<std::mutex> lk(g->mu);
            p->has_released = true;
            p->state = 0;
        }
        g->cv.notify_all();
        if (++g->oldest_held > g->current) { g->current = -1; g->oldest_held = -1; }
    }
    if (g->next_out >= n) return VO_END;
    vo_ingest::Slot* s = &g->slots[g->next_out % depth];
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv.wait(lk, [&] { return s->state == 2 && s->frame == g->next_out; });
        s->state = 3;
    }
    g->current = g->next_out++;
    if (g->oldest_held < 0) g->oldest_held = g->current;
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

int vo_ingest_set_hold(vo_ingest* g, int hold) {
    if (!g) return VO_ERR_BAD_ARG;
    if (g->next_out != 0 || hold < 1 || hold > (int)g->slots.size() - 1) {
        g->ctx->last_error = "vo_ingest_set_hold: before the first vo_ingest_next, 1 <= hold <= depth - 1";
        return VO_ERR_BAD_ARG;
    }
    g->hold = hold;
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
        if (s.d_aux) (void)hipFree(s.d_aux);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.released) (void)hipEventDestroy(s.released);
    }
    if (g->copy) (void)hipStreamDestroy(g->copy);
    delete g;
    return VO_OK;
}

}  // extern "C"
