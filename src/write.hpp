// write.hpp -- drt::write_exr(fname, data, width, height): same signature and pixel semantics as
// the reference's src/write.hpp:9-26 (RGBA, 16-bit half, alpha 1, increasing-Y scan lines), but
// self-contained: OpenEXR (branch RB-2.5 in the reference's .gitmodules) is not available here, so
// this writes the OpenEXR 2 single-part scan-line container itself: ZIP-compressed blocks of 16 scan lines
// (the OpenEXR file-layout document's ZIP_COMPRESSION: byte de-interleave, delta predictor, zlib deflate) when
// built with -DDRT_EXR_ZLIB -lz, uncompressed scan lines otherwise.  Pixel parity with Imf::RgbaOutputFile,
// not byte parity (the reference's default is PIZ compression, a wavelet + Huffman coder of OpenEXR's own).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef DRT_EXR_ZLIB
#include <zlib.h>
#endif

#include "drt/vector.hpp"

namespace drt {

// IEEE binary32 -> binary16, round to nearest even, overflow to infinity, NaN stays NaN
inline uint16_t float_to_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u)                                   // inf / nan
        return (uint16_t)(sign | 0x7C00u | (mag > 0x7F800000u ? 0x0200u | ((mag >> 13) & 0x03FFu) : 0u));
    if (mag >= 0x477FF000u)                                   // rounds to >= 65520 -> inf
        return (uint16_t)(sign | 0x7C00u);
    if (mag < 0x38800000u) {                                  // subnormal half or zero
        if (mag < 0x33000000u)                                // < 2^-25 -> 0
            return (uint16_t)sign;
        const int shift = 126 - (int)(mag >> 23);             // 14..24
        const uint32_t mant = (mag & 0x007FFFFFu) | 0x00800000u;
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1u)))
            ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((mag - 0x38000000u) >> 13);                 // rebias exponent 127 -> 15
    const uint32_t rem = mag & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u)))
        ++h;                                                  // may carry into the exponent: correct
    return (uint16_t)(sign | h);
}

namespace exr_detail {
inline void put(std::vector<unsigned char>& b, const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; b.insert(b.end(), c, c + n); }
inline void put_str(std::vector<unsigned char>& b, const char* s) { put(b, s, std::strlen(s) + 1); }
inline void put_i32(std::vector<unsigned char>& b, int32_t v) { put(b, &v, 4); }
inline void put_f32(std::vector<unsigned char>& b, float v) { put(b, &v, 4); }
inline void attr(std::vector<unsigned char>& b, const char* name, const char* type, const std::vector<unsigned char>& value)
{
    put_str(b, name);
    put_str(b, type);
    put_i32(b, (int32_t)value.size());
    put(b, value.data(), value.size());
}
} // namespace exr_detail

template <typename T>
inline void write_exr(const char* fname, const Vector<T, 3>* data, std::size_t width, std::size_t height)
{
    using namespace exr_detail;
    std::vector<unsigned char> head;
    const uint32_t magic = 20000630u, version = 2u;
    put(head, &magic, 4);
    put(head, &version, 4);
    {   // channels, alphabetical: A B G R, HALF (pixel type 1), linear 0, sampling 1 1
        std::vector<unsigned char> v;
        for (const char* ch : {"A", "B", "G", "R"}) {
            put_str(v, ch);
            put_i32(v, 1);
            const unsigned char plinear[4] = {0, 0, 0, 0};
            put(v, plinear, 4);
            put_i32(v, 1);
            put_i32(v, 1);
        }
        v.push_back(0);
        attr(head, "channels", "chlist", v);
    }
#ifdef DRT_EXR_ZLIB
    const unsigned char compression = 3;       // ZIP_COMPRESSION: blocks of 16 scan lines
    const size_t block_lines = 16;
#else
    const unsigned char compression = 0;       // NO_COMPRESSION: one scan line per block
    const size_t block_lines = 1;
#endif
    { std::vector<unsigned char> v(1, compression); attr(head, "compression", "compression", v); }
    {
        std::vector<unsigned char> v;
        put_i32(v, 0); put_i32(v, 0); put_i32(v, (int32_t)width - 1); put_i32(v, (int32_t)height - 1);
        attr(head, "dataWindow", "box2i", v);
        attr(head, "displayWindow", "box2i", v);
    }
    { std::vector<unsigned char> v(1, 0); attr(head, "lineOrder", "lineOrder", v); }        // INCREASING_Y
    { std::vector<unsigned char> v; put_f32(v, 1.f); attr(head, "pixelAspectRatio", "float", v); }
    { std::vector<unsigned char> v; put_f32(v, 0.f); put_f32(v, 0.f); attr(head, "screenWindowCenter", "v2f", v); }
    { std::vector<unsigned char> v; put_f32(v, 1.f); attr(head, "screenWindowWidth", "float", v); }
    head.push_back(0);

    // the blocks: y of the first line, byte count, data (scan lines, each A B G R planes of halfs)
    const size_t line_bytes = width * 4 * 2;
    const size_t n_blocks = (height + block_lines - 1) / block_lines;
    std::vector<std::vector<unsigned char>> blocks(n_blocks);
    const uint16_t one = float_to_half(1.f);
    for (size_t b = 0; b < n_blocks; ++b) {
        const size_t y0 = b * block_lines, y1 = std::min(height, y0 + block_lines);
        std::vector<unsigned char> raw((y1 - y0) * line_bytes);
        for (size_t y = y0; y < y1; ++y) {
            uint16_t* line = reinterpret_cast<uint16_t*>(raw.data() + (y - y0) * line_bytes);
            for (size_t x = 0; x < width; ++x) {
                const Vector<T, 3>& rgb = data[y * width + x];
                line[0 * width + x] = one;                                   // A
                line[1 * width + x] = float_to_half((float)real(rgb[2]));  // B
                line[2 * width + x] = float_to_half((float)real(rgb[1]));  // G
                line[3 * width + x] = float_to_half((float)real(rgb[0]));  // R
            }
        }
#ifdef DRT_EXR_ZLIB
        // ZIP block: even bytes then odd bytes, each byte replaced by its difference to the previous one (+ 128), deflate;
        // a block that does not shrink is stored raw (the reader tells by the size)
        std::vector<unsigned char> tmp(raw.size());
        {
            unsigned char* t1 = tmp.data();
            unsigned char* t2 = tmp.data() + (raw.size() + 1) / 2;
            for (size_t i = 0; i < raw.size(); ++i)
                *((i & 1) ? t2++ : t1++) = raw[i];
            int prev = tmp[0];
            for (size_t i = 1; i < tmp.size(); ++i) {
                const int cur = tmp[i];
                tmp[i] = (unsigned char)(cur - prev + (128 + 256));
                prev = cur;
            }
        }
        uLongf out_size = compressBound((uLong)tmp.size());
        std::vector<unsigned char> packed(out_size);
        if (compress(packed.data(), &out_size, tmp.data(), (uLong)tmp.size()) != Z_OK)
            throw std::runtime_error("write_exr: zlib compress failed");
        if (out_size < raw.size()) {
            packed.resize(out_size);
            blocks[b].swap(packed);
        } else {
            blocks[b].swap(raw);
        }
#else
        blocks[b].swap(raw);
#endif
    }

    std::FILE* f = std::fopen(fname, "wb");
    if (!f)
        throw std::runtime_error(std::string("write_exr: cannot open ") + fname);
    std::fwrite(head.data(), 1, head.size(), f);
    uint64_t off = head.size() + 8ull * n_blocks;
    for (size_t b = 0; b < n_blocks; ++b) {
        std::fwrite(&off, 8, 1, f);
        off += 8 + blocks[b].size();
    }
    for (size_t b = 0; b < n_blocks; ++b) {
        const int32_t yy = (int32_t)(b * block_lines), size = (int32_t)blocks[b].size();
        std::fwrite(&yy, 4, 1, f);
        std::fwrite(&size, 4, 1, f);
        std::fwrite(blocks[b].data(), 1, blocks[b].size(), f);
    }
    std::fclose(f);
}

} // namespace drt
