// decode_fuzz.cpp -- the host decoders (vo_decode.cpp, vo_jpeg.cpp) fed damaged files under the address and
// undefined-behaviour sanitizers.  Test infrastructure: built and run by tests/test_decode_fuzz.py (host code
// only, no GPU, no HIP).  Every seed file given on the command line is decoded as it is (must succeed), then
// `iters` damaged copies of it are decoded: each must end in VO_OK or an error code, never in a sanitizer report.
//
//   decode_fuzz <iters> <seed> file...
//   decode_fuzz refuse file...      crafted bad files: each must be refused with an error code (no write, no report)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "vo_decode.h"
#include "vo_hip.h"

namespace {

struct Rng {
    uint64_t s;
    uint32_t next() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (uint32_t)(s >> 16);
    }
    uint32_t below(uint32_t n) { return n ? next() % n : 0; }
};

// one damaged copy: a few of byte flips, byte overwrites with marker-like values, a cut, a repeated stretch
void damage(const std::vector<uint8_t>& good, std::vector<uint8_t>* out, Rng* r) {
    *out = good;
    std::vector<uint8_t>& b = *out;
    const int edits = 1 + (int)r->below(4);
    for (int e = 0; e < edits && !b.empty(); ++e) {
        const uint32_t kind = r->below(8);
        // half of the edits land in the first bytes, where the headers and tables are
        const size_t at = r->below(r->below(2) && b.size() > 700 ? 700u : (uint32_t)b.size());
        if (kind < 3) {
            b[at] ^= (uint8_t)(1u << r->below(8));
        } else if (kind < 5) {
            static const uint8_t vals[] = {0x00, 0xff, 0x7f, 0x80, 0xd8, 0xd9, 0xc0, 0xc4, 0xda, 0xdb, 0xdd, 0x01, 0x10, 0x11};
            b[at] = vals[r->below(sizeof(vals))];
        } else if (kind == 5) {
            b.resize(at);  // cut short
        } else if (kind == 6) {
            const size_t len = 1 + r->below(64);
            for (size_t i = 0; i < len && at + i < b.size(); ++i) b[at + i] = (uint8_t)r->next();
        } else {
            const size_t len = 1 + r->below(32);  // a stretch said twice: lengths and offsets stop agreeing
            std::vector<uint8_t> piece(b.begin() + at, b.begin() + (at + len < b.size() ? at + len : b.size()));
            b.insert(b.begin() + at, piece.begin(), piece.end());
        }
    }
}

// decode one buffer the way vo_image_decode does: header, size the output from it, decode
int decode_once(const std::vector<uint8_t>& b, DecodeScratch* sc, std::vector<uint8_t>* out, std::string* err) {
    ImgHdr h;
    int rc = vo_dec_header(b.data(), b.size(), &h, err);
    if (rc != VO_OK) return rc;
    const uint64_t bytes = (uint64_t)h.H * (uint64_t)h.W * (uint64_t)h.channels;
    if (bytes == 0 || bytes > (64ull << 20)) return VO_ERR_BAD_ARG;  // a damaged size field: the ABI's caller sizes `out`, so stop here
    out->assign((size_t)bytes, 0);  // exact size: a write past the image is a sanitizer report
    return vo_dec_decode(b.data(), b.size(), h, out->data(), sc, err);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 3 && strcmp(argv[1], "refuse") == 0) {
        DecodeScratch sc;
        std::vector<uint8_t> buf, out;
        for (int f = 2; f < argc; ++f) {
            std::string err;
            size_t n = 0;
            if (vo_dec_read_file(argv[f], &buf, &n, &err) != VO_OK) {
                fprintf(stderr, "%s: %s\n", argv[f], err.c_str());
                return 1;
            }
            buf.resize(n);
            const int rc = decode_once(buf, &sc, &out, &err);
            if (rc >= 0) {
                fprintf(stderr, "%s: decoded with status %d, expected an error\n", argv[f], rc);
                return 1;
            }
            printf("%s: refused (%d) %s\n", argv[f], rc, err.c_str());
        }
        return 0;
    }
    if (argc < 4) {
        fprintf(stderr, "usage: decode_fuzz <iters> <seed> file... | decode_fuzz refuse file...\n");
        return 2;
    }
    const long iters = atol(argv[1]);
    Rng rng{(uint64_t)atoll(argv[2]) * 0x9e3779b97f4a7c15ull + 1};
    DecodeScratch sc;
    std::vector<uint8_t> good, bad, out;
    long ok = 0, refused = 0;
    for (int f = 3; f < argc; ++f) {
        std::string err;
        size_t n = 0;
        if (vo_dec_read_file(argv[f], &good, &n, &err) != VO_OK) {
            fprintf(stderr, "%s: %s\n", argv[f], err.c_str());
            return 1;
        }
        good.resize(n);
        if (decode_once(good, &sc, &out, &err) != VO_OK) {
            fprintf(stderr, "%s: the undamaged file does not decode: %s\n", argv[f], err.c_str());
            return 1;
        }
        for (long i = 0; i < iters; ++i) {
            damage(good, &bad, &rng);
            err.clear();
            const int rc = decode_once(bad, &sc, &out, &err);
            if (rc == VO_OK) {
                ++ok;
            } else if (rc < 0) {
                ++refused;
            } else {
                fprintf(stderr, "%s: iteration %ld returned %d\n", argv[f], i, rc);
                return 1;
            }
        }
    }
    printf("decode_fuzz: %ld decoded, %ld refused\n", ok, refused);
    return 0;
}
