// include/nrc_vdb.hpp -- dependency-free reader of OpenVDB FloatGrid files (file version >= 222, tree type Tree_float_5_4_3)
// into the dense R8 volume the renderers upload: what vk::Texture3D::FromVDB does with OpenVDB v10 in the reference
// (src/Texture3D.cpp:12-82: dense-ify the first grid over its file_bbox, fill active tiles, require max == 0 or 1; :99-116: one
// byte per voxel, (uint8)(value * 255), memory order x + nx * (y + ny * z)).
//
// Format as recorded in SURVEY.md App. E (no OpenVDB source is available here); handles what the WDAS cloud files use:
// active-mask compression (flag 2), optional zlib-free payloads only -- a zip / blosc compressed file is rejected with a message.
// Errors throw std::runtime_error("SkyRenderer ERROR: ...") like Log::Error (src/Log.cpp:16-20).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace en {

struct VdbVolume {
    uint32_t nx = 0, ny = 0, nz = 0;          // extent of file_bbox
    int32_t bboxMin[3] = {0, 0, 0}, bboxMax[3] = {0, 0, 0};
    std::vector<float> values;                // dense, index x + nx * (y + ny * z)
    std::vector<uint8_t> density;             // R8 UNORM with the reference's truncating quantisation, same order
    uint64_t activeVoxels = 0;                // counted while reading
    int64_t fileVoxelCount = -1;              // the grid's file_voxel_count metadata (-1: absent)
    float maxValue = 0.0f;
    std::string gridName;
};

namespace vdb_detail {

[[noreturn]] inline void fail(const std::string& m) { throw std::runtime_error("SkyRenderer ERROR: " + m); }

struct Reader {
    const std::vector<unsigned char>& d;
    size_t p = 0;
    explicit Reader(const std::vector<unsigned char>& data) : d(data) {}
    const unsigned char* take(size_t n)
    {
        if (p > d.size() || n > d.size() - p) fail("VDB file is truncated");      // (overflow-safe: p and n come from the file)
        const unsigned char* r = d.data() + p;
        p += n;
        return r;
    }
    template <class T>
    T get()
    {
        T v;
        std::memcpy(&v, take(sizeof(T)), sizeof(T));      // little-endian host
        return v;
    }
    std::string str()
    {
        const uint32_t n = get<uint32_t>();
        const unsigned char* b = take(n);
        return std::string((const char*)b, n);
    }
    // metadata map: u32 count x {name, type, u32 size, payload}
    std::map<std::string, std::vector<unsigned char>> meta()
    {
        std::map<std::string, std::vector<unsigned char>> out;
        const uint32_t n = get<uint32_t>();
        for (uint32_t i = 0; i < n; i++) {
            const std::string name = str();
            (void)str();      // type
            const uint32_t size = get<uint32_t>();
            const unsigned char* b = take(size);
            out[name] = std::vector<unsigned char>(b, b + size);
        }
        return out;
    }
};

// node masks: N bits in 64-bit words, bit n = word n >> 6, bit n & 63
struct Mask {
    std::vector<uint64_t> w;
    void read(Reader& r, uint32_t nbits)
    {
        w.resize(nbits / 64);
        std::memcpy(w.data(), r.take(nbits / 8), nbits / 8);
    }
    bool test(uint32_t n) const { return (w[n >> 6] >> (n & 63)) & 1u; }
    uint32_t count() const
    {
        uint32_t c = 0;
        for (uint64_t x : w) c += (uint32_t)__builtin_popcountll(x);
        return c;
    }
};

// "compressed value array" of `count` slots with value mask `mask`
inline std::vector<float> values(Reader& r, uint32_t count, const Mask& mask, uint32_t flags, float background)
{
    const int8_t meta = r.get<int8_t>();
    float inactive0 = background, inactive1 = meta != 0 ? -background : background;
    if (meta == 2 || meta == 4 || meta == 5) {
        inactive0 = r.get<float>();
        if (meta == 5) inactive1 = r.get<float>();
    }
    if (meta == 1) inactive0 = -background;
    Mask sel;
    const bool has_sel = meta == 3 || meta == 4 || meta == 5;
    if (has_sel) sel.read(r, count);
    const uint32_t n = ((flags & 2u) && meta != 6) ? mask.count() : count;
    if (flags & 1u) fail("zip-compressed VDB payloads are not supported (the reference's cloud files are uncompressed)");
    if (flags & 4u) fail("blosc-compressed VDB payloads are not supported (the reference's cloud files are uncompressed)");
    const unsigned char* raw = r.take((size_t)n * 4);
    std::vector<float> out(count);
    if (n == count) {
        std::memcpy(out.data(), raw, (size_t)n * 4);
        return out;
    }
    uint32_t k = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (mask.test(i)) {
            std::memcpy(&out[i], raw + (size_t)k * 4, 4);
            k++;
        } else {
            out[i] = (has_sel && sel.test(i)) ? inactive1 : inactive0;
        }
    }
    return out;
}

}  // namespace vdb_detail

// vk::Texture3D::FromVDB(path), src/Texture3D.cpp:12-82 + the quantisation of :99-116
inline VdbVolume ReadVdb(const std::string& path)
{
    using namespace vdb_detail;
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path);
    std::vector<unsigned char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    Reader r(data);
    if (r.get<int64_t>() != 0x56444220) fail(path + " is not a VDB file");
    const uint32_t version = r.get<uint32_t>();
    if (version < 222) fail("VDB file version " + std::to_string(version) + " is not supported (need >= 222)");
    (void)r.get<uint32_t>(); (void)r.get<uint32_t>();      // library major / minor
    (void)r.get<uint8_t>();                                // has grid offsets
    (void)r.take(36);                                      // uuid
    (void)r.meta();
    if (r.get<uint32_t>() < 1) fail("VDB file holds no grid");
    VdbVolume v;
    v.gridName = r.str();
    const std::string gtype = r.str();
    (void)r.str();                                         // instance parent
    const int64_t grid_pos = r.get<int64_t>(), block_pos = r.get<int64_t>(), end_pos = r.get<int64_t>();
    if (gtype.find("Tree_float_5_4_3") == std::string::npos) fail("VDB grid type " + gtype + " is not supported (need Tree_float_5_4_3)");
    // the three stream positions come from the file: inside it, and in order
    if (grid_pos < 0 || block_pos < grid_pos || end_pos < block_pos || (uint64_t)end_pos > data.size()) fail("VDB grid descriptor points outside the file");
    r.p = (size_t)grid_pos;
    const uint32_t flags = r.get<uint32_t>();
    auto gmeta = r.meta();
    if (!gmeta.count("file_bbox_min") || !gmeta.count("file_bbox_max")) fail("VDB grid has no file_bbox metadata");
    if (gmeta["file_bbox_min"].size() < 12 || gmeta["file_bbox_max"].size() < 12) fail("VDB file_bbox metadata is too short");
    std::memcpy(v.bboxMin, gmeta["file_bbox_min"].data(), 12);
    std::memcpy(v.bboxMax, gmeta["file_bbox_max"].data(), 12);
    if (gmeta.count("file_voxel_count")) {
        if (gmeta["file_voxel_count"].size() < 8) fail("VDB file_voxel_count metadata is too short");
        std::memcpy(&v.fileVoxelCount, gmeta["file_voxel_count"].data(), 8);
    }
    const int64_t ext[3] = {(int64_t)v.bboxMax[0] - v.bboxMin[0] + 1, (int64_t)v.bboxMax[1] - v.bboxMin[1] + 1, (int64_t)v.bboxMax[2] - v.bboxMin[2] + 1};
    if (ext[0] <= 0 || ext[1] <= 0 || ext[2] <= 0 || ext[0] * ext[1] * ext[2] > ((int64_t)1 << 31)) fail("VDB file_bbox is empty or too large");
    v.nx = (uint32_t)ext[0]; v.ny = (uint32_t)ext[1]; v.nz = (uint32_t)ext[2];
    v.values.assign((size_t)v.nx * v.ny * v.nz, 0.0f);
    const std::string xform = r.str();
    // the payload length depends on the map type; every map of the WDAS cloud files is a UniformScaleMap = 5 x vec3d
    if (xform != "UniformScaleMap" && xform != "ScaleMap" && xform != "UniformScaleTranslateMap" && xform != "ScaleTranslateMap")
        fail("VDB transform " + xform + " is not supported");
    (void)r.take(xform.find("Translate") != std::string::npos ? 144 : 120);
    if (r.get<uint32_t>() != 1) fail("VDB tree with more than one buffer is not supported");
    const float background = r.get<float>();
    const uint32_t n_tiles = r.get<uint32_t>(), n_children = r.get<uint32_t>();

    auto fill = [&](const int32_t o[3], int32_t dim, float value) {      // an active tile: every voxel of the cube
        int64_t lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            lo[a] = std::max<int64_t>((int64_t)o[a] - v.bboxMin[a], 0);
            hi[a] = std::min<int64_t>((int64_t)o[a] - v.bboxMin[a] + dim, ext[a]);
        }
        v.activeVoxels += (uint64_t)dim * dim * dim;
        v.maxValue = std::max(v.maxValue, value);
        for (int64_t z = lo[2]; z < hi[2]; z++)
            for (int64_t y = lo[1]; y < hi[1]; y++)
                for (int64_t x = lo[0]; x < hi[0]; x++) v.values[(size_t)x + v.nx * ((size_t)y + (size_t)v.ny * z)] = value;
    };

    for (uint32_t t = 0; t < n_tiles; t++) {
        int32_t o[3] = {r.get<int32_t>(), r.get<int32_t>(), r.get<int32_t>()};
        const float val = r.get<float>();
        if (r.get<uint8_t>()) fill(o, 4096, val);
    }
    struct Leaf { int32_t o[3]; };
    std::vector<Leaf> leaves;                                        // in topology order
    for (uint32_t c = 0; c < n_children; c++) {
        const int32_t o5[3] = {r.get<int32_t>(), r.get<int32_t>(), r.get<int32_t>()};
        Mask cm5, vm5;
        cm5.read(r, 32768); vm5.read(r, 32768);
        const std::vector<float> vals5 = values(r, 32768, vm5, flags, background);
        for (uint32_t n = 0; n < 32768; n++) {
            const int32_t o4[3] = {o5[0] + 128 * (int32_t)(n >> 10), o5[1] + 128 * (int32_t)((n >> 5) & 31), o5[2] + 128 * (int32_t)(n & 31)};
            if (vm5.test(n) && !cm5.test(n)) fill(o4, 128, vals5[n]);
        }
        for (uint32_t n = 0; n < 32768; n++) {
            if (!cm5.test(n)) continue;
            const int32_t o4[3] = {o5[0] + 128 * (int32_t)(n >> 10), o5[1] + 128 * (int32_t)((n >> 5) & 31), o5[2] + 128 * (int32_t)(n & 31)};
            Mask cm4, vm4;
            cm4.read(r, 4096); vm4.read(r, 4096);
            const std::vector<float> vals4 = values(r, 4096, vm4, flags, background);
            for (uint32_t k = 0; k < 4096; k++) {
                const int32_t o3[3] = {o4[0] + 8 * (int32_t)(k >> 8), o4[1] + 8 * (int32_t)((k >> 4) & 15), o4[2] + 8 * (int32_t)(k & 15)};
                if (vm4.test(k) && !cm4.test(k)) fill(o3, 8, vals4[k]);
            }
            for (uint32_t k = 0; k < 4096; k++) {
                if (!cm4.test(k)) continue;
                Leaf l{{o4[0] + 8 * (int32_t)(k >> 8), o4[1] + 8 * (int32_t)((k >> 4) & 15), o4[2] + 8 * (int32_t)(k & 15)}};
                Mask topo;
                topo.read(r, 512);                                  // the leaf's value mask (again in front of its buffer)
                leaves.push_back(l);
            }
        }
    }
    if ((int64_t)r.p != block_pos) fail("VDB topology does not end where the grid descriptor says");
    for (const Leaf& l : leaves) {
        Mask vm;
        vm.read(r, 512);
        const std::vector<float> vals = values(r, 512, vm, flags, background);
        for (uint32_t n = 0; n < 512; n++) {                        // leaf voxel n: x = n >> 6, y = (n >> 3) & 7, z = n & 7
            if (!vm.test(n)) continue;
            v.activeVoxels++;
            v.maxValue = std::max(v.maxValue, vals[n]);
            const int64_t x = (int64_t)l.o[0] + (n >> 6) - v.bboxMin[0], y = (int64_t)l.o[1] + ((n >> 3) & 7) - v.bboxMin[1],
                          z = (int64_t)l.o[2] + (n & 7) - v.bboxMin[2];
            if (x < 0 || y < 0 || z < 0 || x >= ext[0] || y >= ext[1] || z >= ext[2]) continue;
            v.values[(size_t)x + v.nx * ((size_t)y + (size_t)v.ny * z)] = vals[n];
        }
    }
    if ((int64_t)r.p != end_pos) fail("VDB buffers do not end where the grid descriptor says");
    if (v.maxValue != 0.0f && v.maxValue != 1.0f) fail("VDB is not normalized");      // src/Texture3D.cpp:74
    v.density.resize(v.values.size());
    for (size_t i = 0; i < v.values.size(); i++) v.density[i] = (uint8_t)(v.values[i] * 255.0f);      // src/Texture3D.cpp:106
    return v;
}

}  // namespace en
