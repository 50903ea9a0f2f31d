// wave_tracer_amd — PNG reader for bitmap textures (host only; zlib for the inflate).
//
// What the reference accepts (src/bitmap/texture2d_loader.cpp:183-227, src/bitmap/load2d.cpp:200-300): PNG files of bit depth 8 or 16
// — grey, grey + alpha, RGB, RGBA (palette images are expanded to RGB / RGBA) — decoded to normalised floats; 8-bit images are
// sRGB-encoded and 16-bit images linear unless the texture node says otherwise (load2d.cpp:290-291, `colour_encoding`, `gamma`:
// src/texture/bitmap.cpp:81-123); the colour channels are linearised, alpha is not (include/wt/bitmap/texture2d.hpp:262-268).
// Not interlaced images only (Adam7 is rejected with a message).
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene_builder.h"

namespace wth {

namespace {
uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return pa <= pb && pa <= pc ? a : pb <= pc ? b : c;
}
}   // namespace

// encoding: 0 = the file's default (8 bit: sRGB, 16 bit: linear), 1 = linear, 2 = sRGB, 3 = gamma (value ^ gamma)
std::vector<float> load_png(const std::string& path, uint32_t& width, uint32_t& height, uint32_t& channels, int encoding, double gamma) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("(bitmap loader) cannot open " + path);
    std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("(bitmap loader) " + path + ": not a PNG file");
    uint32_t depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    bool have_ihdr = false, done = false;
    for (size_t p = 8; p + 12 <= file.size() && !done;) {
        const uint32_t len = be32(&file[p]);
        if (p + 12 + (size_t)len > file.size()) throw std::runtime_error("(bitmap loader) " + path + ": truncated chunk");
        const std::string type(reinterpret_cast<const char*>(&file[p + 4]), 4);
        const unsigned char* d = &file[p + 8];
        if (type == "IHDR") {
            if (len != 13) throw std::runtime_error("(bitmap loader) " + path + ": bad IHDR");
            width = be32(d);
            height = be32(d + 4);
            depth = d[8];
            ctype = d[9];
            interlace = d[12];
            have_ihdr = true;
        } else if (type == "PLTE")
            plte.assign(d, d + len);
        else if (type == "tRNS")
            trns.assign(d, d + len);
        else if (type == "IDAT")
            idat.insert(idat.end(), d, d + len);
        else if (type == "IEND")
            done = true;
        p += 12 + (size_t)len;
    }
    if (!have_ihdr || !width || !height || idat.empty()) throw std::runtime_error("(bitmap loader) " + path + ": missing IHDR / IDAT");
    if (interlace) throw std::runtime_error("(bitmap loader) " + path + ": interlaced PNG files are not supported");
    uint32_t samples = 0;
    switch (ctype) {
    case 0: samples = 1; break;
    case 2: samples = 3; break;
    case 3: samples = 1; break;
    case 4: samples = 2; break;
    case 6: samples = 4; break;
    default: throw std::runtime_error("(bitmap loader) " + path + ": unknown colour type");
    }
    if (!(depth == 8 || depth == 16) || (ctype == 3 && depth != 8))
        throw std::runtime_error("(bitmap loader) " + path + ": bit depth " + std::to_string(depth) + " is not supported (8 or 16)");
    if (width == 0 || height == 0 || width > 65536u || height > 65536u)   // (keeps the size arithmetic below far from wrapping)
        throw std::runtime_error("(bitmap loader) " + path + ": image dimensions out of range (1..65536)");
    const size_t bps = depth / 8, bpp = samples * bps, row = (size_t)width * bpp;
    std::vector<unsigned char> raw((row + 1) * (size_t)height);
    uLongf got = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || got != raw.size()) throw std::runtime_error("(bitmap loader) " + path + ": corrupt image data");
    // ---- undo the scanline filters (PNG specification, section 9)
    std::vector<unsigned char> img(row * (size_t)height);
    for (uint32_t y = 0; y < height; ++y) {
        const unsigned char* in = &raw[(row + 1) * y];
        unsigned char* out = &img[row * y];
        const unsigned char* up = y ? &img[row * (y - 1)] : nullptr;
        const int ft = in[0];
        if (ft > 4) throw std::runtime_error("(bitmap loader) " + path + ": unknown scanline filter");
        for (size_t i = 0; i < row; ++i) {
            const int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = up && i >= bpp ? up[i - bpp] : 0;
            const int x = in[1 + i];
            const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : paeth(a, b, c);
            out[i] = (unsigned char)((x + pred) & 0xFF);
        }
    }
    // ---- to normalised, linearised floats
    const bool palette = ctype == 3;
    channels = palette ? (trns.empty() ? 3u : 4u) : samples;
    const uint32_t colour_channels = channels == 2 ? 1u : channels == 4 ? 3u : channels;
    int enc = encoding;
    if (enc == 0) enc = depth == 16 ? 1 : 2;
    auto linearise = [&](double v) {
        if (enc == 2) return v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4);   // IEC 61966-2-1
        if (enc == 3) return std::pow(v, gamma);
        return v;
    };
    std::vector<float> out((size_t)width * height * channels);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        double v[4] = {0, 0, 0, 1};
        if (palette) {
            const size_t idx = img[i];
            if (3 * idx + 2 >= plte.size()) throw std::runtime_error("(bitmap loader) " + path + ": palette index out of range");
            for (int c = 0; c < 3; ++c) v[c] = plte[3 * idx + c] / 255.0;
            v[3] = idx < trns.size() ? trns[idx] / 255.0 : 1.0;
        } else {
            for (uint32_t c = 0; c < samples; ++c) {
                const unsigned char* s = &img[i * bpp + c * bps];
                v[c] = depth == 8 ? s[0] / 255.0 : ((s[0] << 8) | s[1]) / 65535.0;
            }
        }
        for (uint32_t c = 0; c < channels; ++c) out[i * channels + c] = (float)(c < colour_channels ? linearise(v[c]) : v[c]);
    }
    return out;
}

}   // namespace wth
