// include/nrc_exr.hpp -- dependency-free reader of scan-line OpenEXR images with FLOAT channels (compression NONE, ZIPS or ZIP)
// into RGBA32F: what tinyexr's LoadEXR does for the reference when it loads its converged ground-truth images
// (src/Reference.cpp:617-631: reference/<scene>/0.exr, RGBA float, must match the render size).  The reference's own files are
// 1920x1080, channels A,B,G,R FLOAT, ZIP (16-line blocks); this build's exporter (nrc_renderer_export_exr) writes the same
// channels uncompressed.  Format recipe: SURVEY.md App. E.  Errors throw std::runtime_error("SkyRenderer ERROR: ...").
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace en {
namespace exr_detail {

[[noreturn]] inline void fail(const std::string& m) { throw std::runtime_error("SkyRenderer ERROR: " + m); }

// ---- inflate (RFC 1951) of a zlib stream (RFC 1950): table-free canonical Huffman decoding, enough for 16-line EXR blocks
struct Bits {
    const unsigned char* p;
    size_t n, pos = 0;
    uint32_t buf = 0;
    int cnt = 0;
    uint32_t get(int k)
    {
        uint32_t v = 0;
        for (int i = 0; i < k; i++) {
            if (cnt == 0) {
                if (pos >= n) fail("EXR: truncated zlib stream");
                buf = p[pos++];
                cnt = 8;
            }
            v |= (buf & 1u) << i;
            buf >>= 1;
            cnt--;
        }
        return v;
    }
};
struct Huff {
    uint16_t count[16] = {0}, symbol[288] = {0};
    void build(const uint8_t* len, int n)
    {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; i++)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(Bits& b) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; l++) {
            code |= (int)b.get(1);
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        fail("EXR: bad Huffman code");
    }
};
inline void inflate(const unsigned char* src, size_t n, std::vector<unsigned char>& out, size_t expect)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (n < 6) fail("EXR: truncated zlib stream");
    Bits b{src + 2, n - 2};      // 2-byte zlib header; the Adler-32 trailer is not checked
    out.clear();
    out.reserve(expect);
    for (;;) {
        const uint32_t last = b.get(1), type = b.get(2);
        if (type == 0) {
            b.cnt = 0;
            if (b.pos > b.n || 4 > b.n - b.pos) fail("EXR: truncated stored block");
            const uint32_t len = b.p[b.pos] | (b.p[b.pos + 1] << 8);
            b.pos += 4;
            if (len > b.n - b.pos) fail("EXR: truncated stored block");
            out.insert(out.end(), b.p + b.pos, b.p + b.pos + len);
            b.pos += len;
            if (out.size() > expect) fail("EXR: a compressed block inflates to the wrong size");
        } else if (type == 1 || type == 2) {
            Huff lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 144; i++) lens[i] = 8;
                for (int i = 144; i < 256; i++) lens[i] = 9;
                for (int i = 256; i < 280; i++) lens[i] = 7;
                for (int i = 280; i < 288; i++) lens[i] = 8;
                lit.build(lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)b.get(3);
                Huff lc;
                lc.build(cl, 19);
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = lc.decode(b);
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (i == 0) fail("EXR: bad code lengths"); val = lens[i - 1]; rep = 3 + (int)b.get(2); }
                        else if (sym == 17) rep = 3 + (int)b.get(3);
                        else rep = 11 + (int)b.get(7);
                        if (i + rep > nlen + ndist) fail("EXR: bad code lengths");
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                lit.build(lens, nlen);
                dist.build(lens + nlen, ndist);
            }
            for (;;) {
                const int sym = lit.decode(b);
                if (out.size() > expect) fail("EXR: a compressed block inflates to the wrong size");      // (never grow past what the chunk may hold)
                if (sym < 256) out.push_back((unsigned char)sym);
                else if (sym == 256) break;
                else {
                    if (sym > 285) fail("EXR: bad length symbol");
                    const int len = lbase[sym - 257] + (int)b.get(lext[sym - 257]);
                    const int ds = dist.decode(b);
                    if (ds > 29) fail("EXR: bad distance symbol");
                    const size_t d = dbase[ds] + b.get(dext[ds]);
                    if (d > out.size()) fail("EXR: distance beyond the window");
                    for (int k = 0; k < len; k++) out.push_back(out[out.size() - d]);
                }
            }
        } else {
            fail("EXR: bad deflate block type");
        }
        if (last) break;
    }
    if (out.size() != expect) fail("EXR: a compressed block inflates to the wrong size");
}

}  // namespace exr_detail

// LoadEXR(&rgba, &width, &height, path): RGBA32F, row-major, top scan line first; missing channels read 0 (A: 1)
inline std::vector<float> LoadExrRGBA(const std::string& path, uint32_t* width, uint32_t* height)
{
    using namespace exr_detail;
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("TinyEXR failed to load " + path);      // the reference's message (src/Reference.cpp:625)
    std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t p = 0;
    // (overflow-safe: p and n come from the file)
    auto need = [&](size_t n) { if (p > d.size() || n > d.size() - p) fail(path + ": truncated EXR file"); };
    auto u32 = [&]() { need(4); uint32_t v; std::memcpy(&v, &d[p], 4); p += 4; return v; };
    auto cstr = [&]() { std::string s; for (;;) { need(1); const char c = (char)d[p++]; if (!c) break; s.push_back(c); } return s; };
    if (u32() != 20000630u) fail(path + " is not an OpenEXR file");
    const uint32_t ver = u32();
    if (ver & 0x1a00u) fail(path + ": tiled / deep / multi-part EXR files are not supported");
    std::vector<std::string> channels;
    int compression = -1, xmin = 0, ymin = 0, xmax = -1, ymax = -1, line_order = 0;
    for (;;) {
        const std::string name = cstr();
        if (name.empty()) break;
        const std::string type = cstr();
        const uint32_t size = u32();
        need(size);
        const size_t a = p;
        const size_t attr_end = a + size;
        auto attr_need = [&](size_t n) { if (p > attr_end || n > attr_end - p) fail(path + ": EXR attribute " + name + " is too short"); };
        if (name == "channels") {
            for (;;) {
                attr_need(1);
                if (d[p] == 0) break;
                const std::string cn = cstr();
                attr_need(16);
                int32_t pixel_type;
                std::memcpy(&pixel_type, &d[p], 4);
                if (pixel_type != 2) fail(path + ": only FLOAT channels are supported");
                int32_t xs, ys;
                std::memcpy(&xs, &d[p + 8], 4); std::memcpy(&ys, &d[p + 12], 4);
                if (xs != 1 || ys != 1) fail(path + ": sub-sampled channels are not supported");
                p += 16;
                channels.push_back(cn);
            }
        } else if (name == "compression") {
            attr_need(1);
            compression = d[p];
        } else if (name == "dataWindow") {
            attr_need(16);
            int32_t w[4];
            std::memcpy(w, &d[p], 16);
            xmin = w[0]; ymin = w[1]; xmax = w[2]; ymax = w[3];
        } else if (name == "lineOrder") {
            attr_need(1);
            line_order = d[p];
        }
        p = a + size;
    }
    (void)line_order;      // chunks carry their own y coordinate
    if (channels.empty() || xmax < xmin || ymax < ymin) fail(path + ": EXR header without channels or data window");
    if (compression != 0 && compression != 2 && compression != 3) fail(path + ": EXR compression " + std::to_string(compression) + " is not supported (NONE, ZIPS, ZIP)");
    if ((int64_t)xmax - xmin >= 65536 || (int64_t)ymax - ymin >= 65536) fail(path + ": EXR data window is too large");
    const uint32_t W = (uint32_t)(xmax - xmin + 1), H = (uint32_t)(ymax - ymin + 1);
    const uint32_t lines_per_chunk = compression == 3 ? 16u : 1u;
    const uint32_t n_chunks = (H + lines_per_chunk - 1) / lines_per_chunk;
    need((size_t)n_chunks * 8);
    std::vector<uint64_t> offsets(n_chunks);
    std::memcpy(offsets.data(), &d[p], (size_t)n_chunks * 8);
    int slot[4] = {-1, -1, -1, -1};      // R, G, B, A -> index in the file's (alphabetical) channel list
    for (size_t c = 0; c < channels.size(); c++) {
        if (channels[c] == "R") slot[0] = (int)c;
        else if (channels[c] == "G") slot[1] = (int)c;
        else if (channels[c] == "B") slot[2] = (int)c;
        else if (channels[c] == "A") slot[3] = (int)c;
    }
    std::vector<float> rgba((size_t)W * H * 4, 0.0f);
    if (slot[3] < 0) for (size_t i = 0; i < (size_t)W * H; i++) rgba[i * 4 + 3] = 1.0f;
    const size_t line_bytes = (size_t)W * 4 * channels.size();
    std::vector<unsigned char> raw, tmp;
    for (uint32_t c = 0; c < n_chunks; c++) {
        if (offsets[c] > d.size()) fail(path + ": EXR chunk offset outside the file");
        p = (size_t)offsets[c];
        need(8);
        int32_t y0, packed;
        std::memcpy(&y0, &d[p], 4); std::memcpy(&packed, &d[p + 4], 4);
        p += 8;
        if (packed < 0) fail(path + ": EXR chunk of negative size");
        need((size_t)packed);
        if ((int64_t)y0 < ymin || (int64_t)y0 - ymin >= (int64_t)H) fail(path + ": EXR chunk outside the data window");
        const uint32_t row0 = (uint32_t)(y0 - ymin);
        const uint32_t rows = std::min(lines_per_chunk, H - row0);
        const size_t expect = line_bytes * rows;
        const unsigned char* src = &d[p];
        if (compression != 0 && (size_t)packed < expect) {
            inflate(src, (size_t)packed, tmp, expect);
            // undo the predictor, then de-interleave (first half -> even bytes, second half -> odd bytes)
            for (size_t i = 1; i < expect; i++) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);
            raw.resize(expect);
            const size_t half = (expect + 1) / 2;
            for (size_t i = 0; i < expect; i++) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
            src = raw.data();
        } else if ((size_t)packed != expect) {
            fail(path + ": EXR chunk of the wrong size");
        }
        for (uint32_t r = 0; r < rows; r++)
            for (int k = 0; k < 4; k++) {
                if (slot[k] < 0) continue;
                const unsigned char* line = src + line_bytes * r + (size_t)slot[k] * W * 4;
                float* dst = &rgba[((size_t)(row0 + r) * W) * 4 + k];
                for (uint32_t x = 0; x < W; x++) std::memcpy(dst + (size_t)x * 4, line + (size_t)x * 4, 4);
            }
    }
    if (width) *width = W;
    if (height) *height = H;
    return rgba;
}

}  // namespace en
