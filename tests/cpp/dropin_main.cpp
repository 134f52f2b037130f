// A host program written against the reference's C++ surface (include/nrc_hpm.hpp, namespace en) the way src/main.cu uses
// it (:152-419): AppConfig from the 17 positional arguments, NeuralRadianceCache, HpmScene(appConfig), Camera(pos, viewDir, up,
// aspect, fov, near, far), NrcHpmRenderer(width, height, blend, &camera, appConfig, hpmScene, nrc), Render(queue, true) and a
// GetLoss() poll per frame, IsBlending(), GetImage().  tests/test_gpu_cpp_dropin.py feeds it a volume, and compares what it
// writes with the same frames rendered through the Python mirror.
//
//   dropin_main <scene.bin> <out.bin> <frames> <17 positional AppConfig arguments> [reference root directory]
// With the optional last argument the program also does what src/main.cu:140-150 does in benchmark mode: en::Reference (ground truth
// generated on first use, 16 blended frames here instead of 8192, exported as <root>/<scene>/0.exr and loaded back), CompareNrc and
// CompareMc with their Result{mse, refMean, ownMean, ownVar, validPixelCount} / GetRelBias() / GetCV(); appended to out.bin.
//
// scene.bin: u32 width,height,nx,ny,nz; f32 env rgba (1x1); f32 frame_random[frames][4]; u8 density[nx*ny*nz]
// out.bin:   f32 loss; f32 inv_proj_view[16], cam_pos[3] (what en::Camera computed); f32 dir_light_dir[3]; f32 image[h][w][4]
#include <hip/hip_runtime_api.h>

#include <cmath>
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
        if (argc != 4 + 17 && argc != 5 + 17) throw std::runtime_error("usage: dropin_main scene.bin out.bin frames <17 AppConfig args> [reference root]");
        const int frames = std::atoi(argv[3]);
        const char* refRoot = argc == 5 + 17 ? argv[4 + 17] : nullptr;
        std::vector<char*> cfg_argv{argv[0]};
        for (int i = 4; i < 4 + 17; i++) cfg_argv.push_back(argv[i]);
        en::AppConfig appConfig(cfg_argv);

        FILE* f = std::fopen(argv[1], "rb");
        if (!f) throw std::runtime_error("cannot open scene file");
        uint32_t dims[5];
        rd(f, dims, 5);
        const uint32_t W = dims[0], H = dims[1];
        const uint32_t nx = dims[2], ny = dims[3], nz = dims[4];
        float env[4];
        rd(f, env, 4);
        std::vector<float> randoms((size_t)frames * 4);
        rd(f, randoms.data(), randoms.size());
        std::vector<uint8_t> density((size_t)nx * ny * nz);
        rd(f, density.data(), density.size());
        std::fclose(f);

        // src/main.cu:175,203-210: the cache first, then the renderer that holds a reference to it
        en::NeuralRadianceCache nrc(appConfig);
        // the multi-GPU entry points of the C++ mirror, one rank: the library's own RCCL communicator sums the gradients of every
        // training step -- over one rank the identity, so the frames below still equal the run without a communicator bit for bit
        unsigned char commId[128];
        en::NeuralRadianceCache::CommUniqueId(commId);
        nrc.CommInit(commId, 0, 1);
        int commRank = -1, commWorld = -1;
        nrc.CommInfo(&commRank, &commWorld);
        if (commRank != 0 || commWorld != 1 || nrc.CommSparse()) throw std::runtime_error("CommInfo after CommInit(id, 0, 1)");
        // round 6: what the exchange carries (fp32 is the default and what this program keeps: its frames are compared bit for bit)
        nrc.SetExchangeDtype(NRC_EXCHANGE_F16);
        if (nrc.GetExchangeDtype() != NRC_EXCHANGE_F16) throw std::runtime_error("SetExchangeDtype(NRC_EXCHANGE_F16)");
        nrc.SetExchangeDtype(NRC_EXCHANGE_F32);
        if (nrc.GetExchangeDtype() != NRC_EXCHANGE_F32) throw std::runtime_error("SetExchangeDtype(NRC_EXCHANGE_F32)");
        en::HpmScene hpmScene(appConfig, density.data(), nx, ny, nz, env, 1, 1);          // src/main.cu:177
        const float aspectRatio = static_cast<float>(W) / static_cast<float>(H);
        en::Camera camera(en::vec3(64.0f, 0.0f, 0.0f), en::vec3(-1.0f, 0.0f, 0.0f), en::vec3(0.0f, 1.0f, 0.0f), aspectRatio,
                          en::radians(60.0f), 0.1f, 100.0f);                                // src/main.cu:180-187
        en::NrcHpmRenderer nrcHpmRenderer(W, H, false, &camera, appConfig, hpmScene, nrc);  // src/main.cu:203-210
        if (nrcHpmRenderer.IsBlending()) throw std::runtime_error("IsBlending() after blend = false");
        // round 6: the schedule cache through the C++ surface -- a table saved, cleared and loaded again holds this renderer's key, and a
        // second renderer of the same kind then starts on the cached schedule
        {
            en::ClearScheduleCache();
            const std::string key = nrcHpmRenderer.GetScheduleKey(), path = std::string(argv[2]) + ".sched";
            if (key.empty() || nrcHpmRenderer.GetScheduleSource() != "default") throw std::runtime_error("GetScheduleKey / GetScheduleSource");
            FILE* sf = std::fopen(path.c_str(), "w");
            if (!sf) throw std::runtime_error("cannot write the schedule file");
            std::fprintf(sf, "%s 1 3 16\n", key.c_str());
            std::fclose(sf);
            if (en::LoadScheduleCache(path) != 1 || en::SaveScheduleCache(path) != 1) throw std::runtime_error("LoadScheduleCache / SaveScheduleCache");
            en::NrcHpmRenderer second(W, H, false, &camera, appConfig, hpmScene, nrc);
            if (second.GetScheduleSource() != "cache") throw std::runtime_error("a renderer whose key is in the table starts on the cached schedule");
            second.Destroy();
            en::ClearScheduleCache();
            std::remove(path.c_str());
        }
        nrcHpmRenderer.SetBlend(true);
        if (!nrcHpmRenderer.IsBlending()) throw std::runtime_error("IsBlending() after SetBlend(true)");
        float polled = 0.0f;
        for (int i = 0; i < frames; i++) {
            en::nrc_check(nrc_renderer_set_frame_random(nrcHpmRenderer.Handle(), &randoms[(size_t)i * 4]));
            if (hpmScene.Update(false, 0.016f)) nrcHpmRenderer.SetSceneParams(hpmScene);  // src/main.cu:264 (static presets: no-op)
            nrcHpmRenderer.Render(nullptr, true);                 // src/main.cu:287
            polled = nrc.GetLossAsync();                            // src/main.cu:376 as a non-blocking poll: last completed step
            if (std::isnan(polled) || std::isinf(polled)) throw std::runtime_error("NaN loss");   // src/main.cu:380-384
        }
        const float loss = nrc.GetLoss();
        nrcHpmRenderer.EvaluateTimestampQueries();
        std::vector<float> image((size_t)W * H * 4);
        const float* d_image = nrcHpmRenderer.GetImageView();      // (the UI's view of the image, src/main.cu:375: the same device pointer)
        if (d_image != nrcHpmRenderer.GetImage()) throw std::runtime_error("GetImageView != GetImage");
        // round 5: the parameters in tiny-cuda-nn's own layout and the checkpoint file, through the C++ surface -- a cache of another seed
        // loaded from this one's checkpoint holds the same 26 624-style vectors, bit for bit
        {
            const std::string ckpt = std::string(argv[2]) + ".ckpt";
            nrc.SaveCheckpoint(ckpt);
            en::AppConfig other = appConfig;
            other.c.seed = 4242;
            en::NeuralRadianceCache nrc2(other);
            nrc2.LoadCheckpoint(ckpt);
            if (nrc2.ParamCountTcnn() != nrc.ParamCountTcnn() || nrc.ParamCountTcnn() == 0) throw std::runtime_error("ParamCountTcnn");
            for (int which = 0; which < 4; which++)
                if (nrc2.GetParamsTcnn(which) != nrc.GetParamsTcnn(which)) throw std::runtime_error("checkpoint round trip differs");
            nrc2.SetParamsTcnn(0, nrc.GetParamsTcnn(1));
            if (nrc2.GetParamsTcnn(0) != nrc.GetParamsTcnn(1)) throw std::runtime_error("SetParamsTcnn / GetParamsTcnn");
            nrc2.Destroy();
            std::remove(ckpt.c_str());
        }
        if (hipMemcpy(image.data(), d_image, image.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            throw std::runtime_error("hipMemcpy failed");
        FILE* o = std::fopen(argv[2], "wb");
        if (!o) throw std::runtime_error("cannot open output file");
        std::fwrite(&loss, sizeof(float), 1, o);
        std::fwrite(camera.Matrices()->inv_proj_view, sizeof(float), 16, o);
        std::fwrite(camera.Matrices()->pos, sizeof(float), 3, o);
        std::fwrite(hpmScene.Scene().dir_light_dir, sizeof(float), 3, o);
        std::fwrite(image.data(), sizeof(float), image.size(), o);
        if (refRoot) {
            // src/main.cu:140-150 (Benchmark): one NRC frame and one MC frame from the reference camera against the ground truth
            en::Reference reference(W, H, appConfig, hpmScene, nullptr, refRoot, 16);                // generates <root>/<id>/0.exr
            const en::Reference::Result nrcResult = reference.CompareNrc(nrcHpmRenderer, &camera, nullptr);
            en::McHpmRenderer mcHpmRenderer(W, H, 32, true, &camera, hpmScene);                      // src/main.cu:212
            const en::Reference::Result mcResult = reference.CompareMc(mcHpmRenderer, &camera, nullptr);
            mcHpmRenderer.EvaluateTimestampQueries();                                                // src/main.cu:284
            if (!(mcHpmRenderer.GetFrameTimeMS() > 0.0f) || mcHpmRenderer.GetImageView() != mcHpmRenderer.GetImage())
                throw std::runtime_error("McHpmRenderer::EvaluateTimestampQueries / GetFrameTimeMS / GetImageView");
            en::Reference again(W, H, appConfig, hpmScene, nullptr, refRoot, 1u << 30);               // the folder exists: loaded, not rendered
            // (pinned random numbers: CompareMc ends with SetCamera(oldCamera), which clears the accumulation image, so the frame
            // it compared is rendered once more below for the dump)
            const float pin[4] = {0.6180339887f, 0.4142135623f, 0.7320508075f, 0.2360679775f};
            en::nrc_check(nrc_mc_renderer_set_frame_random(mcHpmRenderer.Handle(), pin));
            const en::Reference::Result mcAgain = again.CompareMc(mcHpmRenderer, &camera, nullptr);
            mcHpmRenderer.SetCamera(nullptr, again.GetRefCamera());
            en::nrc_check(nrc_mc_renderer_set_frame_random(mcHpmRenderer.Handle(), pin));
            mcHpmRenderer.Render(nullptr);
            if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("hipDeviceSynchronize failed");
            for (const en::Reference::Result* r : {&nrcResult, &mcResult, &mcAgain}) {
                const float v[8] = {r->mse, r->refMean, r->ownMean, r->ownVar, (float)r->validPixelCount, r->GetRelBias(), r->GetCV(), r->GetRelVar()};
                std::fwrite(v, sizeof(float), 8, o);
            }
            std::vector<float> own((size_t)W * H * 4), ref((size_t)W * H * 4);
            if (hipMemcpy(own.data(), mcHpmRenderer.GetImage(), own.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(ref.data(), again.GetRefImage(), ref.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
                throw std::runtime_error("hipMemcpy failed");
            std::fwrite(ref.data(), sizeof(float), ref.size(), o);
            std::fwrite(own.data(), sizeof(float), own.size(), o);
            mcHpmRenderer.Destroy();
            again.Destroy();
            reference.Destroy();
        }
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
