// vdb_main <file.vdb> <out.u8>: the C++ VDB reader of include/nrc_vdb.hpp (what en::HpmScene(appConfig, path) uses) on a file;
// prints the grid's facts as one JSON line and writes the dense R8 volume.  tests/test_io_vdb.py compares both with the Python
// reader's fixture.
#include <cstdio>

#include <nrc_vdb.hpp>

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: vdb_main file.vdb out.u8\n"); return 2; }
    try {
        const en::VdbVolume v = en::ReadVdb(argv[1]);
        FILE* o = std::fopen(argv[2], "wb");
        if (!o) throw std::runtime_error("cannot open output");
        std::fwrite(v.density.data(), 1, v.density.size(), o);
        std::fclose(o);
        std::printf("{\"nx\": %u, \"ny\": %u, \"nz\": %u, \"bbox_min\": [%d, %d, %d], \"bbox_max\": [%d, %d, %d], \"active_voxels\": %llu, \"file_voxel_count\": %lld, "
                    "\"max\": %.9g, \"grid\": \"%s\"}\n", v.nx, v.ny, v.nz, v.bboxMin[0], v.bboxMin[1], v.bboxMin[2], v.bboxMax[0], v.bboxMax[1], v.bboxMax[2],
                    (unsigned long long)v.activeVoxels, (long long)v.fileVoxelCount, v.maxValue, v.gridName.c_str());
        try {
            (void)en::ReadVdb(std::string(argv[1]) + ".does-not-exist");
            return 4;
        } catch (const std::runtime_error& e) {
            if (std::string(e.what()).rfind("SkyRenderer ERROR", 0) != 0) return 5;
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
