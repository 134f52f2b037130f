// exr_main <in.exr> <out.f32>: include/nrc_exr.hpp's reader on a file (the reference's ZIP-compressed ground-truth images or this
// build's uncompressed exports); writes RGBA32F row-major and prints {width, height}.  tests/test_io_vdb.py compares with io_exr.py.
#include <cstdio>

#include <nrc_exr.hpp>

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: exr_main in.exr out.f32\n"); return 2; }
    try {
        uint32_t w = 0, h = 0;
        const std::vector<float> img = en::LoadExrRGBA(argv[1], &w, &h);
        FILE* o = std::fopen(argv[2], "wb");
        if (!o) throw std::runtime_error("cannot open output");
        std::fwrite(img.data(), 4, img.size(), o);
        std::fclose(o);
        std::printf("{\"width\": %u, \"height\": %u}\n", w, h);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
