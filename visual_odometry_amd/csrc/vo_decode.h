// vo_decode.h -- the host decoders (vo_decode.cpp) as the decode pool and the C ABI see them.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

// larger headers are refused before anything is allocated for them (2^27 pixels = 11585 x 11585)
constexpr uint64_t VO_DEC_MAX_PIXELS = 1ull << 27;

struct ImgHdr {
    int H = 0, W = 0;
    int channels = 0;   // of the decoded output: 1 or 3
    int bit_depth = 0;  // of the file's samples
    int kind = 0;       // 1 PNG, 2 PNM, 3 JPEG
    // PNG
    int color_type = 0, interlace = 0;
    // PNM
    int maxval = 0;
    size_t data_off = 0;
};

struct DecodeScratch {  // per worker thread, grow-only
    std::vector<uint8_t> raw;
    std::vector<uint8_t> zero;
    std::vector<uint8_t> file;
    std::vector<int16_t> coef;    // JPEG coefficient blocks
    std::vector<uint8_t> planes;  // JPEG component planes
};

// each returns VO_OK or a VO_ERR_* code and says why in *err
int vo_dec_header(const uint8_t* d, size_t n, ImgHdr* h, std::string* err);
int vo_dec_decode(const uint8_t* d, size_t n, const ImgHdr& h, uint8_t* out, DecodeScratch* sc, std::string* err);
int vo_dec_read_file(const char* path, std::vector<uint8_t>* buf, size_t* n, std::string* err);
