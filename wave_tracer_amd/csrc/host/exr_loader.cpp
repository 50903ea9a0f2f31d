// wave_tracer_amd — OpenEXR reader for bitmap textures (host only; zlib for the inflate).
//
// What the reference gets from OpenEXR's RgbaInputFile (src/bitmap/load2d.cpp:38-75, dispatched on the ".exr" extension by
// src/bitmap/texture2d_loader.cpp:195-200), restated for the files a scene is likely to bring:
//   * the pixels of the DATA window, rows from the top, linear colour encoding;
//   * layout: RGBA as soon as the file has ANY of the channels R, G, B, A (load2d.cpp:47-49 tests `channels() & WRITE_RGBA`, which is non-zero
//     for an RGB file as well), a missing colour channel reading 0 and a missing A reading 1 (RgbaInputFile's fill values); luminance (1 channel)
//     for a file with Y and none of them; other channels are ignored;
//   * HALF precision whatever the file stores (load2d.cpp:39 has the TODO): FLOAT channels are rounded to the nearest half and clamped to
//     +-65504, UINT channels likewise, as OpenEXR converts them for an Rgba frame buffer.
// Read here: single-part scan-line files, compression NONE, RLE, ZIPS, ZIP, pixel types UINT / HALF / FLOAT, any data window, any line order,
// x / y sampling 1.  Refused with a message: tiled, multi-part and deep files, luminance-chroma (RY / BY) files, the PIZ / PXR24 / B44 / DWA
// codecs (re-save such a file with ZIP compression).
#include <limits>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene_builder.h"

namespace wth {

namespace {

float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t u;
    if (e == 0) {
        if (m == 0)
            u = s;
        else {   // subnormal half: normalise
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024u)) {
                mm <<= 1;
                ++sh;
            }
            u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31)
        u = s | 0x7f800000u | (m << 13);
    else
        u = s | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// the float OpenEXR hands to an Rgba (half) frame buffer: clamped to the finite half range, rounded to nearest even
float through_half(float f) {
    if (std::isnan(f)) return f;
    const float kMax = 65504.f;
    f = f > kMax ? kMax : (f < -kMax ? -kMax : f);
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t s = (u >> 16) & 0x8000u;
    const int32_t e = (int32_t)((u >> 23) & 255u) - 127 + 15;
    uint32_t m = u & 0x7fffffu;
    uint16_t h;
    if (e <= 0) {
        if (e < -10)
            h = (uint16_t)s;   // rounds to zero
        else {
            m |= 0x800000u;
            const int shift = 14 - e;   // 14 .. 24
            const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
            h = (uint16_t)(s | (q + ((rem > halfway || (rem == halfway && (q & 1u))) ? 1u : 0u)));
        }
    } else {
        const uint32_t q = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1fffu;
        h = (uint16_t)(s | (q + ((rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ? 1u : 0u)));   // (a carry into the exponent is the right result; 65504 was clamped above)
    }
    return half_to_float(h);
}

struct channel_t {
    std::string name;
    int type;   // 0 UINT, 1 HALF, 2 FLOAT
    int xs, ys;
};

// the byte predictor + de-interleaving OpenEXR applies behind its RLE and ZIP codecs
void unpredict(std::vector<unsigned char>& buf) {
    for (size_t i = 1; i < buf.size(); ++i) buf[i] = (unsigned char)(buf[i - 1] + buf[i] - 128);
    std::vector<unsigned char> out(buf.size());
    const size_t half = (buf.size() + 1) / 2;
    for (size_t i = 0, a = 0, b = half; i < buf.size(); ++i) out[i] = (i & 1) ? buf[b++] : buf[a++];
    buf.swap(out);
}

}   // namespace

std::vector<float> load_exr(const std::string& path, uint32_t& width, uint32_t& height, uint32_t& channels) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("(exr loader) cannot open " + path);
    std::vector<unsigned char> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto fail = [&](const std::string& w) -> void { throw std::runtime_error("(exr loader) " + path + ": " + w); };
    size_t p = 0;
    auto need = [&](size_t n) {
        if (p + n > b.size()) fail("file ends inside its header or pixel data");
    };
    auto i32 = [&]() {
        need(4);
        int32_t v;
        std::memcpy(&v, &b[p], 4);
        p += 4;
        return v;
    };
    auto str = [&]() {
        const size_t s = p;
        while (p < b.size() && b[p]) ++p;
        if (p >= b.size()) fail("unterminated string in the header");
        return std::string((const char*)&b[s], (p++) - s);
    };
    if (i32() != 20000630) fail("not an OpenEXR file");
    const int32_t version = i32();
    if ((version & 0xff) != 2) fail("OpenEXR version 2 expected");
    if (version & 0x200) fail("tiled files are not read (scan-line files only)");
    if (version & 0x800) fail("deep files are not read");
    if (version & 0x1000) fail("multi-part files are not read");
    std::vector<channel_t> chans;
    int compression = -1;
    int32_t dw[4] = {0, 0, -1, -1};
    bool have_dw = false;
    for (;;) {
        need(1);
        if (b[p] == 0) {
            ++p;
            break;
        }
        const std::string name = str(), type = str();
        const int32_t size = i32();
        if (size < 0) fail("attribute of negative size");
        need((size_t)size);
        const size_t end = p + (size_t)size;
        if (name == "channels" && type == "chlist") {
            while (p < end && b[p]) {
                channel_t c;
                c.name = str();
                c.type = i32();
                p += 4;   // pLinear + 3 reserved bytes
                c.xs = i32();
                c.ys = i32();
                chans.push_back(c);
            }
        } else if (name == "compression" && size == 1)
            compression = b[p];
        else if (name == "dataWindow" && size == 16) {
            std::memcpy(dw, &b[p], 16);
            have_dw = true;
        }
        p = end;
    }
    if (chans.empty() || compression < 0 || !have_dw) fail("header lacks channels, compression or dataWindow");
    if (dw[2] < dw[0] || dw[3] < dw[1]) fail("empty data window");
    static const char* kCodec[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
    if (compression > 3)
        fail(std::string("compression ") + (compression < 10 ? kCodec[compression] : "(unknown)") + " is not read (NONE, RLE, ZIPS and ZIP are): re-save the image with ZIP compression");
    const int64_t W64 = (int64_t)dw[2] - dw[0] + 1, H64 = (int64_t)dw[3] - dw[1] + 1;
    if (W64 > 65536 || H64 > 65536) fail("data window of " + std::to_string(W64) + " x " + std::to_string(H64) + " pixels: 65536 x 65536 at most");
    const uint32_t W = (uint32_t)W64, H = (uint32_t)H64;
    {   // a scan-line block costs the file at least its 8-byte table entry and its 8-byte header: a window the file cannot hold is a damaged header
        const uint64_t blocks = ((uint64_t)H + (compression == 3 ? 15u : 0u)) / (compression == 3 ? 16u : 1u);
        if (blocks * 16 > b.size()) fail("data window of " + std::to_string(H64) + " lines in a file of " + std::to_string(b.size()) + " bytes");
        // ... and its pixels at least 1/1000 of their raw size (a flat image under ZIP shrinks ~1000-fold at best): a window the file cannot plausibly hold
        // would otherwise be met with a multi-gigabyte allocation instead of a message
        const uint64_t raw_bytes = (uint64_t)W * H * 2u * (uint64_t)chans.size();
        if (raw_bytes / 1000u > b.size() + 65536u)
            fail("data window of " + std::to_string(W64) + " x " + std::to_string(H64) + " pixels in a file of " + std::to_string(b.size()) + " bytes");
    }
    int slot[5] = {-1, -1, -1, -1, -1};   // R G B A Y -> index into chans
    size_t row_bytes = 0;
    std::vector<size_t> chan_off(chans.size());
    for (size_t i = 0; i < chans.size(); ++i) {
        const channel_t& c = chans[i];
        if (c.type < 0 || c.type > 2) fail("channel " + c.name + ": unknown pixel type");
        if (c.name == "RY" || c.name == "BY") fail("luminance-chroma files are not read");
        if (c.xs != 1 || c.ys != 1) fail("channel " + c.name + ": subsampled channels are not read");
        chan_off[i] = row_bytes;
        row_bytes += (size_t)W * (c.type == 1 ? 2 : 4);
        const char* names[5] = {"R", "G", "B", "A", "Y"};
        for (int k = 0; k < 5; ++k)
            if (c.name == names[k]) slot[k] = (int)i;
    }
    // (a luminance file with alpha — Y + A, no colours — is RGBA with R = G = B = Y, as RgbaInputFile reads it)
    const bool y_as_rgb = slot[4] >= 0 && slot[0] < 0 && slot[1] < 0 && slot[2] < 0 && slot[3] >= 0;
    const bool rgba = slot[0] >= 0 || slot[1] >= 0 || slot[2] >= 0 || slot[3] >= 0;
    if (!rgba && slot[4] < 0) fail("none of the channels R, G, B, A, Y");
    const uint32_t C = rgba ? 4u : 1u;
    std::vector<float> px;
    try {
        px.assign((size_t)W * H * C, 0.f);
    } catch (const std::bad_alloc&) {
        fail("data window of " + std::to_string(W64) + " x " + std::to_string(H64) + " pixels: out of memory");
    }
    if (rgba)
        for (size_t i = 0; i < (size_t)W * H; ++i) px[4 * i + 3] = 1.f;   // RgbaInputFile's fill value for a missing A (missing colours: 0)
    const uint32_t lines_per_block = compression == 3 ? 16u : 1u;
    const uint32_t n_blocks = (H + lines_per_block - 1) / lines_per_block;
    need((size_t)n_blocks * 8);
    std::vector<uint64_t> offsets(n_blocks);
    std::memcpy(offsets.data(), &b[p], (size_t)n_blocks * 8);
    std::vector<unsigned char> raw;
    for (uint32_t blk = 0; blk < n_blocks; ++blk) {
        p = (size_t)offsets[blk];
        if (offsets[blk] > b.size()) fail("scan-line offset beyond the end of the file");
        const int32_t y0 = i32(), size = i32();
        if (size < 0) fail("scan-line block of negative size");
        need((size_t)size);
        if (y0 < dw[1] || y0 > dw[3]) fail("scan-line block outside the data window");
        const uint32_t rows = std::min<uint32_t>(lines_per_block, (uint32_t)(dw[3] - y0 + 1));
        const size_t want = row_bytes * rows;
        if ((size_t)size == want || compression == 0)   // stored as is (every codec falls back to that when it does not shrink the block)
            raw.assign(b.begin() + (long)p, b.begin() + (long)(p + (size_t)size));
        else if (compression == 1) {
            raw.clear();
            const unsigned char *s = &b[p], *e = s + size;
            while (s < e) {
                const int n = (signed char)*s++;
                if (n < 0) {
                    if (s + (-n) > e) fail("RLE data overruns its block");
                    raw.insert(raw.end(), s, s + (-n));
                    s += -n;
                } else {
                    if (s >= e) fail("RLE data overruns its block");
                    raw.insert(raw.end(), (size_t)n + 1, *s++);
                }
            }
            if (raw.size() != want) fail("RLE block of unexpected size");
            unpredict(raw);
        } else {
            raw.resize(want);
            uLongf got = (uLongf)want;
            if (uncompress(raw.data(), &got, &b[p], (uLong)size) != Z_OK || got != want) fail("ZIP block does not inflate to its scan lines");
            unpredict(raw);
        }
        if (raw.size() != want) fail("scan-line block of unexpected size");
        for (uint32_t r = 0; r < rows; ++r) {
            const unsigned char* line = raw.data() + (size_t)r * row_bytes;
            float* out = &px[(size_t)((uint32_t)(y0 - dw[1]) + r) * W * C];
            auto read = [&](int ci, uint32_t x) -> float {
                const unsigned char* q = line + chan_off[(size_t)ci];
                if (chans[(size_t)ci].type == 1) {
                    uint16_t h;
                    std::memcpy(&h, q + 2 * (size_t)x, 2);
                    return half_to_float(h);
                }
                if (chans[(size_t)ci].type == 2) {
                    float v;
                    std::memcpy(&v, q + 4 * (size_t)x, 4);
                    return through_half(v);
                }
                uint32_t u;
                std::memcpy(&u, q + 4 * (size_t)x, 4);
                return u > 65504u ? std::numeric_limits<float>::infinity() : through_half((float)u);   // (beyond half's range: +inf, like the half conversion)
            };
            for (uint32_t x = 0; x < W; ++x) {
                if (rgba) {
                    for (int k = 0; k < 4; ++k)
                        if (slot[k] >= 0) out[4 * (size_t)x + (size_t)k] = read(slot[k], x);
                    if (y_as_rgb) out[4 * (size_t)x] = out[4 * (size_t)x + 1] = out[4 * (size_t)x + 2] = read(slot[4], x);
                } else
                    out[x] = read(slot[4], x);
            }
        }
    }
    width = W;
    height = H;
    channels = C;
    return px;
}

}   // namespace wth
