// A host program written against the reference's C++ surface (include/nrc_hpm.hpp, namespace en) the way src/main.cu uses
// it: AppConfig from the 17 positional arguments, NeuralRadianceCache, NrcHpmRenderer, Render(queue, true) per frame,
// GetLoss(), GetImage().  tests/test_gpu_cpp_dropin.py feeds it a scene file, and compares what it writes with the same
// frames rendered through the Python mirror.
//
//   dropin_main <scene.bin> <out.bin> <frames> <17 positional AppConfig arguments>
//
// scene.bin: u32 width,height,nx,ny,nz; f32 size[3], density_factor, g, dir_light_dir[3], dir_light_strength,
//            point_light_pos[3], point_light_strength, point_light_color[3], env_strength, env rgba (1x1), inv_proj_view[16],
//            cam_pos[3]; f32 frame_random[frames][4]; u8 density[nx*ny*nz]
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include <nrc_hpm.hpp>

template <typename T>
static void rd(FILE* f, T* p, size_t n)
{
    if (std::fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short scene file");
}

int main(int argc, char** argv)
{
    try {
        if (argc != 4 + 17) throw std::runtime_error("usage: dropin_main scene.bin out.bin frames <17 AppConfig args>");
        const int frames = std::atoi(argv[3]);
        std::vector<char*> cfg_argv{argv[0]};
        for (int i = 4; i < argc; i++) cfg_argv.push_back(argv[i]);
        en::AppConfig appConfig(cfg_argv);

        FILE* f = std::fopen(argv[1], "rb");
        if (!f) throw std::runtime_error("cannot open scene file");
        uint32_t dims[5];
        rd(f, dims, 5);
        const uint32_t W = dims[0], H = dims[1];
        nrc_scene scene{};
        nrc_camera camera{};
        scene.nx = dims[2]; scene.ny = dims[3]; scene.nz = dims[4];
        rd(f, scene.size, 3); rd(f, &scene.density_factor, 1); rd(f, &scene.g, 1);
        rd(f, scene.dir_light_dir, 3); rd(f, &scene.dir_light_strength, 1);
        rd(f, scene.point_light_pos, 3); rd(f, &scene.point_light_strength, 1);
        rd(f, scene.point_light_color, 3); rd(f, &scene.env_strength, 1);
        float env[4];
        rd(f, env, 4);
        rd(f, camera.inv_proj_view, 16); rd(f, camera.pos, 3);
        std::vector<float> randoms((size_t)frames * 4);
        rd(f, randoms.data(), randoms.size());
        std::vector<uint8_t> density((size_t)scene.nx * scene.ny * scene.nz);
        rd(f, density.data(), density.size());
        std::fclose(f);
        scene.density = density.data();
        scene.env = env; scene.env_w = 1; scene.env_h = 1;

        // src/main.cu:175,203-210: the cache first, then the renderer that holds a reference to it
        en::NeuralRadianceCache nrc(appConfig);
        en::NrcHpmRenderer nrcHpmRenderer(W, H, false, &camera, appConfig, scene, nrc);
        nrcHpmRenderer.SetBlend(true);
        float loss = 0.0f;
        for (int i = 0; i < frames; i++) {
            en::nrc_check(nrc_renderer_set_frame_random(nrcHpmRenderer.Handle(), &randoms[(size_t)i * 4]));
            nrcHpmRenderer.Render(nullptr, true);                 // src/main.cu:287
            loss = nrc.GetLoss();                                   // src/main.cu:376
        }
        nrcHpmRenderer.EvaluateTimestampQueries();
        std::vector<float> image((size_t)W * H * 4);
        const float* d_image = nrcHpmRenderer.GetImage();
        if (hipMemcpy(image.data(), d_image, image.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            throw std::runtime_error("hipMemcpy failed");
        FILE* o = std::fopen(argv[2], "wb");
        if (!o) throw std::runtime_error("cannot open output file");
        std::fwrite(&loss, sizeof(float), 1, o);
        std::fwrite(image.data(), sizeof(float), image.size(), o);
        std::fclose(o);
        std::printf("frames %d loss %.9g frame %.3f ms name %s\n", frames, loss, nrcHpmRenderer.GetFrameTimeMS(), appConfig.GetName().c_str());
        // explicit Destroy() in the reference's order (src/main.cu:401-412); destructors are idempotent
        nrcHpmRenderer.Destroy();
        nrc.Destroy();
        // error behaviour: Log::Error throws std::runtime_error("SkyRenderer ERROR: ...")
        try {
            en::AppConfig bad(std::vector<char*>{argv[0]});
            std::printf("missing exception\n");
            return 2;
        } catch (const std::runtime_error& e) {
            if (std::string(e.what()).rfind("SkyRenderer ERROR", 0) != 0) return 3;
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
