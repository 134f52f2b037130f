// include/nrc_hpm.hpp -- the reference's C++ surface for the hot path, header-only over the C ABI (nrc_hpm.h).
//
// Same class and method names, argument order and meaning as
//   en::AppConfig             include/engine/AppConfig.hpp:9-66, src/AppConfig.cpp:154-182
//   en::NeuralRadianceCache   include/engine/graphics/NeuralRadianceCache.hpp:13-32
//   en::NrcHpmRenderer        include/engine/graphics/renderer/NrcHpmRenderer.hpp:16-41
//   en::McHpmRenderer         include/engine/graphics/renderer/McHpmRenderer.hpp:16-31
// with the Vulkan/CUDA-interop types replaced: VkQueue -> hipStream_t (as void*), the two cudaExternalSemaphore_t of
// Init() dropped (stream order), VkImage/VkImageView -> device pointer to the RGBA32F framebuffer.
// Errors throw std::runtime_error("SkyRenderer ERROR: ...") exactly like Log::Error(msg, true) (src/Log.cpp:16-20).
// Ownership as in the reference: the NRC does not own the four I/O buffers; the renderer holds a reference to the NRC,
// which must outlive it; Destroy() is explicit and idempotent, destructors call it.
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "nrc_hpm.h"

namespace en {

inline void nrc_check(int status)
{
    if (status != NRC_OK) throw std::runtime_error(nrc_last_error());
}

struct AppConfig {
    nrc_config c;

    AppConfig() { nrc_config_default(&c); }

    // the reference's 18-entry argv: program name + 17 positional arguments (src/AppConfig.cpp:154-182)
    explicit AppConfig(const std::vector<char*>& argv)
    {
        nrc_config_default(&c);
        if (argv.size() != 18) throw std::runtime_error("SkyRenderer ERROR: Argument count does not match requirements for AppConfig");
        size_t i = 1;
        std::strncpy(c.loss_fn, argv[i++], sizeof(c.loss_fn) - 1);
        std::strncpy(c.optimizer, argv[i++], sizeof(c.optimizer) - 1);
        c.learning_rate = std::stof(argv[i++]);
        c.ema_decay = std::stof(argv[i++]);
        c.pos_id = (uint32_t)std::stoi(argv[i++]);
        c.dir_id = (uint32_t)std::stoi(argv[i++]);
        c.nn_width = (uint32_t)std::stoi(argv[i++]);
        c.nn_depth = (uint32_t)std::stoi(argv[i++]);
        c.log2_infer_batch_size = (uint32_t)std::stoi(argv[i++]);
        c.log2_train_batch_size = (uint32_t)std::stoi(argv[i++]);
        c.train_batch_count = (uint32_t)std::stoi(argv[i++]);
        c.scene_id = (uint32_t)std::stoi(argv[i++]);
        c.train_ring_buf_size = std::stof(argv[i++]);
        c.train_spp = (uint32_t)std::stoi(argv[i++]);
        c.primary_ray_length = (uint32_t)std::stoi(argv[i++]);
        c.primary_ray_prob = std::stof(argv[i++]);
        c.train_ray_length = (uint32_t)std::stoi(argv[i++]);
    }

    std::string GetName() const      // src/AppConfig.cpp:184-205
    {
        std::string s;
        s += std::string(c.loss_fn) + "_" + c.optimizer + "_" + std::to_string(c.learning_rate) + "_" + std::to_string(c.ema_decay) + "_";
        s += std::to_string(c.pos_id) + "_" + std::to_string(c.dir_id) + "_" + std::to_string(c.nn_width) + "_" + std::to_string(c.nn_depth) + "_";
        s += std::to_string(c.log2_infer_batch_size) + "_" + std::to_string(c.log2_train_batch_size) + "_" + std::to_string(c.train_batch_count) + "_";
        s += std::to_string(c.scene_id) + "_" + std::to_string(c.train_ring_buf_size) + "_" + std::to_string(c.train_spp) + "_";
        s += std::to_string(c.primary_ray_length) + "_" + std::to_string(c.primary_ray_prob) + "_" + std::to_string(c.train_ray_length);
        return s;
    }
};

class NeuralRadianceCache {
public:
    explicit NeuralRadianceCache(const AppConfig& appConfig) { nrc_check(nrc_cache_create(&appConfig.c, &h_)); }
    ~NeuralRadianceCache() { Destroy(); }
    NeuralRadianceCache(const NeuralRadianceCache&) = delete;
    NeuralRadianceCache& operator=(const NeuralRadianceCache&) = delete;

    // device pointers, caller-owned; the reference's two external semaphores are replaced by `stream` order
    void Init(uint32_t inferCount, float* dInferInput, float* dInferOutput, float* dTrainInput, float* dTrainTarget,
              void* stream = nullptr)
    {
        nrc_check(nrc_cache_init(h_, inferCount, dInferInput, dInferOutput, dTrainInput, dTrainTarget, stream));
    }
    void InferAndTrain(const uint32_t* inferFilter, bool train) { nrc_check(nrc_cache_infer_and_train(h_, inferFilter, train ? 1 : 0)); }
    void Destroy()
    {
        if (h_) { nrc_cache_destroy(h_); h_ = nullptr; }
    }
    float GetLoss() const { return nrc_cache_get_loss(h_); }
    size_t GetInferBatchCount() const { return nrc_cache_get_infer_batch_count(h_); }
    size_t GetTrainBatchCount() const { return nrc_cache_get_train_batch_count(h_); }
    uint32_t GetInferBatchSize() const { return nrc_cache_get_infer_batch_size(h_); }
    uint32_t GetTrainBatchSize() const { return nrc_cache_get_train_batch_size(h_); }
    nrc_cache_t* Handle() const { return h_; }

private:
    nrc_cache_t* h_ = nullptr;
};

class NrcHpmRenderer {
public:
    NrcHpmRenderer(uint32_t width, uint32_t height, bool blend, const nrc_camera* camera, const AppConfig& appConfig,
                   const nrc_scene& hpmScene, NeuralRadianceCache& nrc, void* stream = nullptr, const nrc_tile* tile = nullptr)
    {
        nrc_check(nrc_renderer_create(width, height, blend ? 1 : 0, camera, &appConfig.c, &hpmScene, nrc.Handle(), tile, stream, &h_));
    }
    ~NrcHpmRenderer() { Destroy(); }
    NrcHpmRenderer(const NrcHpmRenderer&) = delete;
    NrcHpmRenderer& operator=(const NrcHpmRenderer&) = delete;

    void Render(void* /*queue: the stream given at construction*/, bool train) { nrc_check(nrc_renderer_render(h_, train ? 1 : 0)); }
    void Destroy()
    {
        if (h_) { nrc_renderer_destroy(h_); h_ = nullptr; }
    }
    void ExportOutputImageToFile(void* /*queue*/, const std::string& filePath) const { nrc_check(nrc_renderer_export_exr(h_, filePath.c_str())); }
    void EvaluateTimestampQueries() { (void)nrc_renderer_frame_time_ms(h_, stage_ms_); }
    const float* GetImage() const { return nrc_renderer_framebuffer(h_); }     // RGBA32F [height][width]
    const float* GetImage(void* consumerStream) const { return nrc_renderer_framebuffer_on(h_, consumerStream); }
    float GetFrameTimeMS() const { return nrc_renderer_frame_time_ms(h_, nullptr); }
    const float* GetStageTimesMS() const { return stage_ms_; }
    void SetCamera(void* /*queue*/, const nrc_camera* camera) { nrc_check(nrc_renderer_set_camera(h_, camera)); }
    void SetBlend(bool blend) { nrc_check(nrc_renderer_set_blend(h_, blend ? 1 : 0)); }
    void SetSceneParams(const nrc_scene& scene) { nrc_check(nrc_renderer_set_scene_params(h_, &scene)); }   // HpmScene::Update
    nrc_renderer_t* Handle() const { return h_; }

private:
    nrc_renderer_t* h_ = nullptr;
    float stage_ms_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

class McHpmRenderer {
public:
    McHpmRenderer(uint32_t width, uint32_t height, uint32_t pathLength, bool blend, const nrc_camera* camera,
                  const nrc_scene& scene, void* stream = nullptr, const nrc_tile* tile = nullptr)
    {
        nrc_check(nrc_mc_renderer_create(width, height, pathLength, blend ? 1 : 0, camera, &scene, tile, stream, &h_));
    }
    ~McHpmRenderer() { Destroy(); }
    McHpmRenderer(const McHpmRenderer&) = delete;
    McHpmRenderer& operator=(const McHpmRenderer&) = delete;

    void Render(void* /*queue*/) { nrc_check(nrc_mc_renderer_render(h_)); }
    void Destroy()
    {
        if (h_) { nrc_mc_renderer_destroy(h_); h_ = nullptr; }
    }
    void ExportOutputImageToFile(void* /*queue*/, const std::string& filePath) const { nrc_check(nrc_mc_renderer_export_exr(h_, filePath.c_str())); }
    const float* GetImage() const { return nrc_mc_renderer_framebuffer(h_); }
    void SetCamera(void* /*queue*/, const nrc_camera* camera) { nrc_check(nrc_mc_renderer_set_camera(h_, camera)); }
    void SetBlend(bool blend) { nrc_check(nrc_mc_renderer_set_blend(h_, blend ? 1 : 0)); }
    void SetSceneParams(const nrc_scene& scene) { nrc_check(nrc_mc_renderer_set_scene_params(h_, &scene)); }
    nrc_mc_renderer_t* Handle() const { return h_; }

private:
    nrc_mc_renderer_t* h_ = nullptr;
};

}  // namespace en
