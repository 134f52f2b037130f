// The multi-GPU surface of include/nrc_hpm.hpp in ONE process: two host threads play two ranks of a frame sharded into interleaved
// 8-column strips -- each with its own NeuralRadianceCache, NrcHpmRenderer(tile) and en::Reference(..., &tile, &nrc) -- over a
// host-staged transport installed with SetCollectiveHooks (what a host with MPI would do).  Both run the benchmark-mode call of
// src/main.cu:140-150, Reference::CompareNrc, which now reduces the five sums over the ranks, and GatherFrame / the collective
// ExportOutputImageToFile.  tests/test_gpu_cpp_dropin.py compares the results with the single-GPU frame and nrc_compare_images.
//
//   sharded_main <scene.bin> <out.bin> <reference root (holds <scene>/0.exr of the GLOBAL size)> <export.exr> <17 AppConfig arguments>
// scene.bin as for dropin_main (frames = 0).  out.bin: f32 inv_proj_view[16], cam_pos[3], dir_light_dir[3] (what en::Camera / en::HpmScene computed); per rank 5 floats {mse, refMean, ownMean, ownVar, validPixelCount}, then rank 0's
// gathered frame f32 [H][W][4], then rank 1's.
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include <nrc_hpm.hpp>

namespace {

constexpr int kWorld = 2;

struct Transport {      // a two-party exchange through host memory
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    std::vector<unsigned char> slot[kWorld];
    void barrier()
    {
        std::unique_lock<std::mutex> lock(m);
        const unsigned long long g = generation;
        if (++arrived == kWorld) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lock, [&] { return generation != g; });
    }
};
struct Party {
    Transport* t;
    int rank;
};

int allreduce_f64(void* user, double* d_buf, uint32_t n, void* /*stream: the library has waited for it*/)
{
    Party* p = (Party*)user;
    std::vector<unsigned char>& mine = p->t->slot[p->rank];
    mine.resize((size_t)n * 8);
    if (hipMemcpy(mine.data(), d_buf, mine.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    p->t->barrier();
    std::vector<double> sum(n, 0.0);
    for (int r = 0; r < kWorld; r++)      // rank order on every rank: identical results
        for (uint32_t i = 0; i < n; i++) sum[i] += ((const double*)p->t->slot[r].data())[i];
    p->t->barrier();                      // (nobody overwrites a slot before everybody has read it)
    return hipMemcpy(d_buf, sum.data(), (size_t)n * 8, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}
int allgather(void* user, const void* d_send, void* d_recv, size_t bytes, void* /*stream*/)
{
    Party* p = (Party*)user;
    std::vector<unsigned char>& mine = p->t->slot[p->rank];
    mine.resize(bytes);
    if (hipMemcpy(mine.data(), d_send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    p->t->barrier();
    int rc = 0;
    for (int r = 0; r < kWorld; r++)
        if (hipMemcpy((char*)d_recv + (size_t)r * bytes, p->t->slot[r].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    p->t->barrier();
    return rc;
}

template <typename T>
void rd(FILE* f, T* p, size_t n)
{
    if (std::fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short scene file");
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        if (argc != 5 + 17) throw std::runtime_error("usage: sharded_main scene.bin out.bin <reference root> export.exr <17 AppConfig args>");
        std::vector<char*> cfg_argv{argv[0]};
        for (int i = 5; i < 5 + 17; i++) cfg_argv.push_back(argv[i]);
        FILE* f = std::fopen(argv[1], "rb");
        if (!f) throw std::runtime_error("cannot open scene file");
        uint32_t dims[5];
        rd(f, dims, 5);
        const uint32_t W = dims[0], H = dims[1], nx = dims[2], ny = dims[3], nz = dims[4];
        float env[4];
        rd(f, env, 4);
        std::vector<uint8_t> density((size_t)nx * ny * nz);
        rd(f, density.data(), density.size());
        std::fclose(f);

        Transport transport;
        float results[kWorld][5];
        std::vector<float> frames[kWorld];
        std::string errors[kWorld];
        auto run = [&](int rank) {
            try {
                en::AppConfig appConfig(cfg_argv);
                en::NeuralRadianceCache nrc(appConfig);
                Party party{&transport, rank};
                nrc.SetCollectiveHooks(rank, kWorld, allreduce_f64, allgather, &party);
                en::HpmScene hpmScene(appConfig, density.data(), nx, ny, nz, env, 1, 1);
                const nrc_tile tile{(uint32_t)rank, (uint32_t)kWorld, W, H, 8};
                const uint32_t block = 8, round = block * kWorld;
                const uint32_t rest = W % round;
                const uint32_t lw = W / round * block + (rest > (uint32_t)rank * block ? std::min(rest - (uint32_t)rank * block, block) : 0u);
                en::Camera camera(en::vec3(64.0f, 0.0f, 0.0f), en::vec3(-1.0f, 0.0f, 0.0f), en::vec3(0.0f, 1.0f, 0.0f),
                                  static_cast<float>(W) / static_cast<float>(H), en::radians(60.0f), 0.1f, 100.0f);
                en::NrcHpmRenderer renderer(lw, H, false, &camera, appConfig, hpmScene, nrc, nullptr, &tile);
                if (!renderer.IsSharded()) throw std::runtime_error("IsSharded() of a tile renderer");
                en::Reference reference(lw, H, appConfig, hpmScene, nullptr, argv[3], 1u << 30, &tile, &nrc);
                const en::Reference::Result r = reference.CompareNrc(renderer, &camera, nullptr);      // collective
                const float v[5] = {r.mse, r.refMean, r.ownMean, r.ownVar, (float)r.validPixelCount};
                std::memcpy(results[rank], v, sizeof(v));
                // the frame CompareNrc rendered was cleared by its SetCamera(oldCamera): render the reference view once more
                renderer.SetCamera(nullptr, reference.GetRefCamera());
                const float pin[4] = {0.6180339887f, 0.4142135623f, 0.7320508075f, 0.2360679775f};
                en::nrc_check(nrc_renderer_set_frame_random(renderer.Handle(), pin));
                renderer.Render(nullptr, false);
                float* d_full = nullptr;
                if (hipMalloc((void**)&d_full, (size_t)W * H * 16) != hipSuccess) throw std::runtime_error("hipMalloc failed");
                renderer.GatherFrame(d_full, nullptr);                                                 // collective
                frames[rank].resize((size_t)W * H * 4);
                if (hipDeviceSynchronize() != hipSuccess ||
                    hipMemcpy(frames[rank].data(), d_full, frames[rank].size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
                    throw std::runtime_error("hipMemcpy failed");
                (void)hipFree(d_full);
                renderer.ExportOutputImageToFile(nullptr, argv[4], /*root*/ 1);                        // collective; rank 1 writes
                reference.Destroy();
                renderer.Destroy();
                nrc.Destroy();
            } catch (const std::exception& e) {
                errors[rank] = e.what();
                std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
                std::_Exit(1);      // the other party would wait for ever
            }
        };
        std::thread peer(run, 1);
        run(0);
        peer.join();
        FILE* o = std::fopen(argv[2], "wb");
        if (!o) throw std::runtime_error("cannot open output file");
        {      // what en::Camera computed (the test renders its single-GPU counterpart with exactly these matrices)
            en::Camera camera(en::vec3(64.0f, 0.0f, 0.0f), en::vec3(-1.0f, 0.0f, 0.0f), en::vec3(0.0f, 1.0f, 0.0f),
                              static_cast<float>(W) / static_cast<float>(H), en::radians(60.0f), 0.1f, 100.0f);
            std::fwrite(camera.Matrices()->inv_proj_view, sizeof(float), 16, o);
            std::fwrite(camera.Matrices()->pos, sizeof(float), 3, o);
            en::AppConfig appConfig(cfg_argv);
            en::HpmScene hpmScene(appConfig, density.data(), nx, ny, nz, env, 1, 1);
            std::fwrite(hpmScene.Scene().dir_light_dir, sizeof(float), 3, o);
        }
        for (int r = 0; r < kWorld; r++) std::fwrite(results[r], sizeof(float), 5, o);
        for (int r = 0; r < kWorld; r++) std::fwrite(frames[r].data(), sizeof(float), frames[r].size(), o);
        std::fclose(o);
        std::printf("sharded ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
