// include/nrc_hpm.hpp -- the reference's C++ surface for the hot path, header-only over the C ABI (nrc_hpm.h).
//
// Same class and method names, argument order and meaning as
//   en::AppConfig             include/engine/AppConfig.hpp:9-66, src/AppConfig.cpp:154-182
//   en::NeuralRadianceCache   include/engine/graphics/NeuralRadianceCache.hpp:13-32
//   en::NrcHpmRenderer        include/engine/graphics/renderer/NrcHpmRenderer.hpp:16-41
//   en::McHpmRenderer         include/engine/graphics/renderer/McHpmRenderer.hpp:16-31
//   en::Camera                include/engine/graphics/Camera.hpp:18-60, src/Camera.cpp:164-174   (values only, no Vulkan)
//   en::HpmScene              include/engine/HpmScene.hpp:12-43, src/HpmScene.cpp:24-76          (values only, no Vulkan; loads the
//                             cloud itself with the dependency-free VDB reader of nrc_vdb.hpp, src/Texture3D.cpp:12-82)
//   en::Reference             include/engine/graphics/Reference.hpp:14-39, src/Reference.cpp:9-145,443-455,566-660
// with the Vulkan/CUDA-interop types replaced: VkQueue -> hipStream_t (as void*), the two cudaExternalSemaphore_t of
// Init() -> two hipEvent_t (as void*; the overload without them relies on stream order), VkImage/VkImageView -> device
// pointer to the RGBA32F framebuffer, glm::vec3 -> en::vec3.
// Errors throw std::runtime_error("SkyRenderer ERROR: ...") exactly like Log::Error(msg, true) (src/Log.cpp:16-20).
// Ownership as in the reference: the NRC does not own the four I/O buffers; the renderer holds a reference to the NRC,
// which must outlive it; Destroy() is explicit and idempotent, destructors call it.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <stdexcept>
#include <string>
#include <vector>

#include "nrc_exr.hpp"
#include "nrc_hpm.h"
#include "nrc_vdb.hpp"

namespace en {

inline void nrc_check(int status)
{
    if (status != NRC_OK) throw std::runtime_error(nrc_last_error());
}
// nrc_set_wave_priority_raise: false puts every kernel of the library at the hardware's default issue priority (a host whose own kernels --
// RCCL's, the runtime's fills -- run beside the renderer); NeuralRadianceCache::CommInit does it for more than one rank
inline void SetWavePriorityRaise(bool raise) { nrc_check(nrc_set_wave_priority_raise(raise ? 1 : 0)); }

// glm::vec3 stand-in for the few host-side values the path takes (camera pose, light angles)
struct vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
};
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }      // glm::radians

struct AppConfig {
    // AppConfig::HpmSceneConfig (include/engine/AppConfig.hpp:23-36, presets src/AppConfig.cpp:93-150)
    struct HpmSceneConfig {
        uint32_t id = UINT32_MAX;
        float dirLightStrength = 0.0f, pointLightStrength = 0.0f;
        std::string hdrEnvMapPath;
        float hdrEnvMapStrength = 0.0f, density = 0.0f;
        bool dynamic = false;
        HpmSceneConfig() = default;
        explicit HpmSceneConfig(uint32_t id_) : id(id_)
        {
            static const float presets[6][4] = {{16.0f, 0.0f, 0.0f, 0.6f}, {0.0f, 64.0f, 0.0f, 0.6f}, {0.0f, 128.0f, 0.0f, 1.0f},
                                                {16.0f, 0.0f, 0.0f, 0.25f}, {8.0f, 0.0f, 0.1f, 0.6f}, {0.0f, 0.0f, 1.0f, 1.6f}};
            if (id > 5) throw std::runtime_error("SkyRenderer ERROR: HpmSceneConfig ID is invalid");
            dirLightStrength = presets[id][0]; pointLightStrength = presets[id][1];
            hdrEnvMapStrength = presets[id][2]; density = presets[id][3];
        }
    };

    nrc_config c;
    HpmSceneConfig scene;

    AppConfig() { nrc_config_default(&c); scene = HpmSceneConfig(c.scene_id); }

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
        scene = HpmSceneConfig(c.scene_id);
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

// en::Camera (src/Camera.cpp): pose + projection -> CameraMatrices.invProjView and camera.pos, the two things the shaders read
// (nrc-descriptors.glsl:1-11).  UpdateUniformBuffer restates src/Camera.cpp:164-174 with glm's own fp32 formulas:
// glm::perspective (right-handed, depth -1..1), glm::lookAt (right-handed), glm::inverse (cofactor expansion).
class Camera {
public:
    Camera(const vec3& pos, const vec3& viewDir, const vec3& up, float aspectRatio, float fov, float nearPlane, float farPlane)
        : m_Pos(pos), m_ViewDir(viewDir), m_Up(up), m_AspectRatio(aspectRatio), m_Fov(fov), m_NearPlane(nearPlane), m_FarPlane(farPlane)
    {
        UpdateUniformBuffer();
    }
    void Destroy() {}
    void UpdateUniformBuffer()
    {
        float proj[16] = {0}, view[16] = {0}, pv[16];      // column-major, m[4*col + row]
        const float t = std::tan(m_Fov / 2.0f);
        proj[0] = 1.0f / (m_AspectRatio * t);
        proj[5] = 1.0f / t;
        proj[10] = -(m_FarPlane + m_NearPlane) / (m_FarPlane - m_NearPlane);
        proj[11] = -1.0f;
        proj[14] = -(2.0f * m_FarPlane * m_NearPlane) / (m_FarPlane - m_NearPlane);
        const vec3 f = normalize(m_ViewDir);               // normalize(center - eye), center = pos + viewDir
        const vec3 s = normalize(cross(f, m_Up));
        const vec3 u = cross(s, f);
        view[0] = s.x; view[4] = s.y; view[8] = s.z;
        view[1] = u.x; view[5] = u.y; view[9] = u.z;
        view[2] = -f.x; view[6] = -f.y; view[10] = -f.z;
        view[12] = -dot(s, m_Pos); view[13] = -dot(u, m_Pos); view[14] = dot(f, m_Pos);
        view[15] = 1.0f;
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) {
                float acc = 0.0f;
                for (int k = 0; k < 4; k++) acc += proj[4 * k + r] * view[4 * c + k];
                pv[4 * c + r] = acc;
            }
        invert(pv, m_Matrices.inv_proj_view);
        m_Matrices.pos[0] = m_Pos.x; m_Matrices.pos[1] = m_Pos.y; m_Matrices.pos[2] = m_Pos.z;
        m_Changed = true;
    }
    const vec3& GetPos() const { return m_Pos; }
    void SetPos(const vec3& pos) { m_Pos = pos; m_Changed = true; }
    const vec3& GetViewDir() const { return m_ViewDir; }
    void SetViewDir(const vec3& viewDir) { m_ViewDir = viewDir; m_Changed = true; }
    const vec3& GetUp() const { return m_Up; }
    void SetUp(const vec3& up) { m_Up = up; m_Changed = true; }
    bool HasChanged() const { return m_Changed; }
    void SetChanged(bool changed) { m_Changed = changed; }
    float GetAspectRatio() const { return m_AspectRatio; }
    void SetAspectRatio(float aspectRatio) { m_AspectRatio = aspectRatio; }
    void SetAspectRatio(uint32_t width, uint32_t height) { m_AspectRatio = (float)width / (float)height; }
    float GetFov() const { return m_Fov; }
    void SetFov(float fov) { m_Fov = fov; }
    float GetNearPlane() const { return m_NearPlane; }
    void SetNearPlane(float nearPlane) { m_NearPlane = nearPlane; }
    float GetFarPlane() const { return m_FarPlane; }
    void SetFarPlane(float farPlane) { m_FarPlane = farPlane; }
    void Move(const vec3& move) { m_Pos = vec3(m_Pos.x + move.x, m_Pos.y + move.y, m_Pos.z + move.z); m_Changed = true; }
    // what the renderers take in place of the camera's descriptor set
    const nrc_camera* Matrices() const { return &m_Matrices; }

private:
    static float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
    static vec3 cross(const vec3& a, const vec3& b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
    static vec3 normalize(const vec3& a)
    {
        const float inv = 1.0f / std::sqrt(dot(a, a));
        return vec3(a.x * inv, a.y * inv, a.z * inv);
    }
    static void invert(const float* m, float* out)      // glm::inverse(mat4): adjugate / determinant
    {
        float inv[16];
        inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
        inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
        inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
        inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
        inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
        inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
        inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
        inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
        inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
        inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
        inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
        inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
        inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
        inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
        inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
        inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
        const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
        if (det == 0.0f) throw std::runtime_error("SkyRenderer ERROR: camera matrix is singular");
        const float id = 1.0f / det;
        for (int i = 0; i < 16; i++) out[i] = inv[i] * id;
    }

    vec3 m_Pos, m_ViewDir, m_Up;
    bool m_Changed = true;
    float m_AspectRatio, m_Fov, m_NearPlane, m_FarPlane;
    nrc_camera m_Matrices{};
};

// en::HpmScene (src/HpmScene.cpp:24-76): the scene preset of the AppConfig (light strengths, medium density), the directional
// light at zenith -1.57 / azimuth 0 (src/HpmScene.cpp:28, VecFromAngles src/DirLight.cpp:5-14), a white point light at the origin
// (:30), g = 0.8 (:45), and the density volume.  The reference loads "data/volume/wdas_cloud_quarter.vdb" through OpenVDB inside
// this constructor; here the caller hands over the dense R8 volume (what Texture3D::FromVDB produces, src/Texture3D.cpp:12-82 --
// io_vdb.py parses the file) and, optionally, an RGBA32F environment map (default: the 1x1 map the reference ends up with for an
// empty hdrEnvMapPath: black, src/read_file.cpp:85-90; pass a 1x1 white texel for quirk Q9).  The scene does not own the arrays.
class HpmScene {
public:
    HpmScene(const AppConfig& appConfig, const uint8_t* density, uint32_t nx, uint32_t ny, uint32_t nz, const float* envRgba = nullptr,
             uint32_t envWidth = 0, uint32_t envHeight = 0)
        : m_ID(appConfig.scene.id), m_Dynamic(appConfig.scene.dynamic)
    {
        SetUp(appConfig, density, nx, ny, nz, envRgba, envWidth, envHeight);
    }
    // HpmScene(appConfig), src/HpmScene.cpp:24-55: the scene loads its density volume itself -- vk::Texture3D::FromVDB on
    // "data/volume/wdas_cloud_quarter.vdb" (:44) -- here with the reader of nrc_vdb.hpp; the environment map is the reference's:
    // whatever file it names, every texel ends up 1.0 (quirk Q9, src/read_file.cpp:129-130), an empty path gives one black texel
    explicit HpmScene(const AppConfig& appConfig, const std::string& vdbPath = "data/volume/wdas_cloud_quarter.vdb")
        : m_ID(appConfig.scene.id), m_Dynamic(appConfig.scene.dynamic), m_Volume(ReadVdb(vdbPath))
    {
        SetUp(appConfig, m_Volume.density.data(), m_Volume.nx, m_Volume.ny, m_Volume.nz, appConfig.scene.hdrEnvMapPath.empty() ? nullptr : m_WhiteEnv, 1, 1);
    }
    const VdbVolume& Volume() const { return m_Volume; }
    HpmScene(const HpmScene&) = delete;
    HpmScene& operator=(const HpmScene&) = delete;
    // HpmScene::Update(renderImgui, deltaTime), src/HpmScene.cpp:56-76: only a dynamic scene 3 moves (azimuth += dt / 2,
    // wrapped at 2 * 3.141); returns true when a value changed -- hand Scene() to the renderers' SetSceneParams then
    bool Update(bool /*renderImgui*/, float deltaTime)
    {
        if (!m_Dynamic || m_ID != 3) return false;
        SetDirLightAngles(m_Zenith, (float)std::fmod((double)(m_Azimuth + deltaTime * 0.5f), 2.0 * 3.141));
        return true;
    }
    void SetDirLightAngles(float zenith, float azimuth)     // VecFromAngles: Ry(azimuth) * Rx(zenith) * (0, 1, 0)
    {
        m_Zenith = zenith; m_Azimuth = azimuth;
        const double cz = std::cos((double)zenith), sz = std::sin((double)zenith), ca = std::cos((double)azimuth), sa = std::sin((double)azimuth);
        m_Scene.dir_light_dir[0] = (float)(sa * sz);
        m_Scene.dir_light_dir[1] = (float)cz;
        m_Scene.dir_light_dir[2] = (float)(ca * sz);
    }
    void Destroy() {}
    bool IsDynamic() const { return m_Dynamic; }
    void SetDynamic(bool dynamic) { m_Dynamic = dynamic; }
    const nrc_scene& Scene() const { return m_Scene; }
    nrc_scene& Scene() { return m_Scene; }

private:
    void SetUp(const AppConfig& appConfig, const uint8_t* density, uint32_t nx, uint32_t ny, uint32_t nz, const float* envRgba, uint32_t envWidth,
               uint32_t envHeight)
    {
        m_Scene.density = density; m_Scene.nx = nx; m_Scene.ny = ny; m_Scene.nz = nz;
        m_Scene.size[0] = m_Scene.size[1] = m_Scene.size[2] = 0.0f;      // normalize(extent) * 107.5, src/NrcHpmRenderer.cu:910-912
        m_Scene.density_factor = appConfig.scene.density;
        m_Scene.g = 0.8f;
        m_Scene.dir_light_strength = appConfig.scene.dirLightStrength;
        m_Scene.point_light_pos[0] = m_Scene.point_light_pos[1] = m_Scene.point_light_pos[2] = 0.0f;
        m_Scene.point_light_color[0] = m_Scene.point_light_color[1] = m_Scene.point_light_color[2] = 1.0f;
        m_Scene.point_light_strength = appConfig.scene.pointLightStrength;
        m_Scene.env_strength = appConfig.scene.hdrEnvMapStrength;
        if (envRgba) { m_Scene.env = envRgba; m_Scene.env_w = envWidth; m_Scene.env_h = envHeight; }
        else { m_Scene.env = m_BlackEnv; m_Scene.env_w = 1; m_Scene.env_h = 1; }
        SetDirLightAngles(-1.57f, 0.0f);
    }
    uint32_t m_ID;
    bool m_Dynamic;
    float m_Zenith = -1.57f, m_Azimuth = 0.0f;
    float m_BlackEnv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float m_WhiteEnv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    VdbVolume m_Volume;
    nrc_scene m_Scene{};
};

// schedules the renderers' tuners settled on in earlier processes (include/nrc_hpm.h: nrc_schedule_cache_load / _save / _clear); returns entries
inline int LoadScheduleCache(const std::string& path) { int n = 0; nrc_check(nrc_schedule_cache_load(path.c_str(), &n)); return n; }
inline int SaveScheduleCache(const std::string& path) { int n = 0; nrc_check(nrc_schedule_cache_save(path.c_str(), &n)); return n; }
inline void ClearScheduleCache() { nrc_check(nrc_schedule_cache_clear()); }

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
    // the reference's signature (include/engine/graphics/NeuralRadianceCache.hpp:15-22) with its two external semaphores as
    // hipEvent_t: InferAndTrain waits for cudaStartEvent and records cudaFinishedEvent (either may be null)
    void Init(uint32_t inferCount, float* dInferInput, float* dInferOutput, float* dTrainInput, float* dTrainTarget,
              void* cudaStartEvent, void* cudaFinishedEvent, void* stream)
    {
        nrc_check(nrc_cache_init_events(h_, inferCount, dInferInput, dInferOutput, dTrainInput, dTrainTarget, stream, cudaStartEvent,
                                        cudaFinishedEvent));
    }
    void InferAndTrain(const uint32_t* inferFilter, bool train) { nrc_check(nrc_cache_infer_and_train(h_, inferFilter, train ? 1 : 0)); }
    void Destroy()
    {
        if (h_) { nrc_cache_destroy(h_); h_ = nullptr; }
    }
    // GetLoss(): the reference's value -- the loss of the training step the last InferAndTrain / Render enqueued (m_Loss =
    // trainer->loss, src/NeuralRadianceCache.cu:154); waits for that step only.  GetLossAsync(): never blocks -- the most recent
    // COMPLETED step's loss and number, for a per-frame poll (src/main.cu:303,376) that must not drain the frame pipeline.
    float GetLoss() const { return nrc_cache_get_loss(h_); }
    float GetLossBlocking() const { return nrc_cache_get_loss(h_); }
    float GetLossAsync(uint32_t* step = nullptr, uint32_t* stepsEnqueued = nullptr) const
    {
        float loss = 0.0f;
        nrc_check(nrc_cache_get_loss_async(h_, &loss, step, stepsEnqueued));
        return loss;
    }
    size_t GetInferBatchCount() const { return nrc_cache_get_infer_batch_count(h_); }
    size_t GetTrainBatchCount() const { return nrc_cache_get_train_batch_count(h_); }
    uint32_t GetInferBatchSize() const { return nrc_cache_get_infer_batch_size(h_); }
    uint32_t GetTrainBatchSize() const { return nrc_cache_get_train_batch_size(h_); }
    nrc_cache_t* Handle() const { return h_; }

    // Multi-GPU (not in the reference): one process per GPU, each with its own cache and a renderer created with an nrc_tile.  Rank 0
    // draws a 128-byte id, the host distributes it (MPI, sockets, ...), every rank calls CommInit; from then on every training
    // step exchanges the gradients over RCCL on the training stream (include/nrc_hpm.h, nrc_cache_comm_init).
    static void CommUniqueId(void* out128) { nrc_check(nrc_comm_unique_id(out128)); }
    void CommInit(const void* uniqueId128, int rank, int world) { nrc_check(nrc_cache_comm_init(h_, uniqueId128, rank, world)); }
    void CommInfo(int* rank, int* world) const { nrc_check(nrc_cache_comm_info(h_, rank, world)); }
    bool CommSparse() const { return nrc_cache_comm_sparse(h_) != 0; }      // HashGrid table gradient exchanged as lists
    // what the exchange carries: NRC_EXCHANGE_F32 (default) or NRC_EXCHANGE_F16 -- the gradients as fp16 numbers pre-scaled by loss_scale
    void SetExchangeDtype(int dtype) { nrc_check(nrc_cache_set_exchange_dtype(h_, dtype)); }
    int GetExchangeDtype() const { return nrc_cache_get_exchange_dtype(h_); }
    // frame gather / metric reduction over a transport of the host's own (a cache without CommInit): include/nrc_hpm.h
    void SetCollectiveHooks(int rank, int world, nrc_allreduce_f64_fn allreduce, nrc_allgather_fn allgather, void* user)
    {
        nrc_check(nrc_cache_set_collective_hooks(h_, rank, world, allreduce, allgather, user));
    }

    // Parameters in tiny-cuda-nn's own layout (what the reference's trainer->params_full_precision() holds for the model of
    // src/NeuralRadianceCache.cu:39; output matrix 16 x nnWidth): which = 0 master weights, 1 EMA weights, 2 Adam m, 3 Adam v
    uint32_t ParamCountTcnn() const { return nrc_cache_param_count_tcnn(h_); }
    std::vector<float> GetParamsTcnn(int which = 0) const
    {
        std::vector<float> v(ParamCountTcnn());
        nrc_check(nrc_cache_get_params_tcnn(h_, which, v.data()));
        return v;
    }
    void SetParamsTcnn(int which, const std::vector<float>& v)
    {
        if (v.size() != ParamCountTcnn()) throw std::runtime_error("SkyRenderer ERROR: SetParamsTcnn: wrong parameter count");
        nrc_check(nrc_cache_set_params_tcnn(h_, which, v.data()));
    }
    // checkpoint file (the reference has none): model shape, step, weights / EMA weights / Adam moments
    void SaveCheckpoint(const std::string& filePath) const { nrc_check(nrc_cache_save_checkpoint(h_, filePath.c_str())); }
    void LoadCheckpoint(const std::string& filePath) { nrc_check(nrc_cache_load_checkpoint(h_, filePath.c_str())); }

private:
    nrc_cache_t* h_ = nullptr;
};

class NrcHpmRenderer {
public:
    NrcHpmRenderer(uint32_t width, uint32_t height, bool blend, const nrc_camera* camera, const AppConfig& appConfig,
                   const nrc_scene& hpmScene, NeuralRadianceCache& nrc, void* stream = nullptr, const nrc_tile* tile = nullptr)
    {
        nrc_check(nrc_renderer_create(width, height, blend ? 1 : 0, camera, &appConfig.c, &hpmScene, nrc.Handle(), tile, stream, &h_));
        if (tile) tile_ = *tile;
        else { tile_.x_offset = 0; tile_.x_stride = 1; tile_.global_w = width; tile_.global_h = height; tile_.x_block = 1; }
    }
    // the reference's own argument list (include/engine/graphics/renderer/NrcHpmRenderer.hpp:19-26)
    NrcHpmRenderer(uint32_t width, uint32_t height, bool blend, const Camera* camera, const AppConfig& appConfig,
                   const HpmScene& hpmScene, NeuralRadianceCache& nrc, void* stream = nullptr, const nrc_tile* tile = nullptr)
        : NrcHpmRenderer(width, height, blend, camera->Matrices(), appConfig, hpmScene.Scene(), nrc, stream, tile)
    {
    }
    ~NrcHpmRenderer() { Destroy(); }
    NrcHpmRenderer(const NrcHpmRenderer&) = delete;
    NrcHpmRenderer& operator=(const NrcHpmRenderer&) = delete;

    void Render(void* /*queue: the stream given at construction*/, bool train) { nrc_check(nrc_renderer_render(h_, train ? 1 : 0)); }
    // nFrames consecutive Render(queue, train) calls, enqueued by one call; frameRandoms: nFrames x 4 floats or nullptr
    void RenderFrames(void* /*queue*/, uint32_t nFrames, const float* frameRandoms, bool train)
    {
        nrc_check(nrc_renderer_render_frames(h_, nFrames, frameRandoms, train ? 1 : 0));
    }
    void Destroy()
    {
        if (h_) { nrc_renderer_destroy(h_); h_ = nullptr; }
    }
    // src/NrcHpmRenderer.cu:437-493.  A renderer that draws one tile of a multi-GPU frame exports the WHOLE frame: the call is then
    // collective over the frame's ranks (all-gather through the cache's communicator) and rank `root` writes the file.
    void ExportOutputImageToFile(void* /*queue*/, const std::string& filePath, int root = 0) const
    {
        if (IsSharded()) nrc_check(nrc_renderer_export_exr_gathered(h_, filePath.c_str(), root));
        else nrc_check(nrc_renderer_export_exr(h_, filePath.c_str()));
    }
    // the whole [global_h][global_w] RGBA32F frame into dGlobalImage (device) on every rank of the frame; collective when sharded
    void GatherFrame(float* dGlobalImage, void* stream) const { nrc_check(nrc_renderer_gather_frame(h_, dGlobalImage, stream)); }
    bool IsSharded() const { return tile_.x_stride > 1; }
    const nrc_tile& Tile() const { return tile_; }
    void EvaluateTimestampQueries() { (void)nrc_renderer_frame_time_ms(h_, stage_ms_); }
    const float* GetImage() const { return nrc_renderer_framebuffer(h_); }     // RGBA32F [height][width]
    // VkImageView GetImageView() of the reference (include/engine/graphics/renderer/NrcHpmRenderer.hpp:37, src/main.cu:375: what the
    // UI samples): there is one view of the output image here, the device pointer itself
    const float* GetImageView() const { return GetImage(); }
    const float* GetImage(void* consumerStream) const { return nrc_renderer_framebuffer_on(h_, consumerStream); }
    // the reads of the image enqueued on consumerStream end here: the next frame's compositing waits for them
    void ReleaseImage(void* consumerStream) { nrc_check(nrc_renderer_release_frame(h_, consumerStream)); }
    bool IsBlending() const { return nrc_renderer_is_blending(h_) != 0; }
    // where the schedule in use came from ("default" | "cache" | "tuner" | "pinned" | "pinned in part") and the key it is remembered under
    // (include/nrc_hpm.h, nrc_schedule_cache_load / _save)
    std::string GetScheduleSource() const { return nrc_renderer_schedule_source(h_); }
    std::string GetScheduleKey() const { return nrc_renderer_schedule_key(h_); }
    float GetFrameTimeMS() const { return nrc_renderer_frame_time_ms(h_, nullptr); }
    const float* GetStageTimesMS() const { return stage_ms_; }
    // the timestamp queries of every frame since the last statistics reset as a timeline: [frames][6] ms from the first frame's start
    // (gen_rays start / done, train rays done, inference done, compositing done, training done); nrc_renderer_frame_timeline
    std::vector<float> FrameTimeline(uint32_t maxFrames = 4096) const
    {
        std::vector<float> t((size_t)maxFrames * 6);
        uint32_t n = 0;
        nrc_check(nrc_renderer_frame_timeline(h_, t.data(), maxFrames, &n));
        t.resize((size_t)n * 6);
        return t;
    }
    void SetCamera(void* /*queue*/, const nrc_camera* camera) { nrc_check(nrc_renderer_set_camera(h_, camera)); }
    void SetCamera(void* queue, const Camera* camera) { SetCamera(queue, camera->Matrices()); }
    void SetBlend(bool blend) { nrc_check(nrc_renderer_set_blend(h_, blend ? 1 : 0)); }
    void SetSceneParams(const nrc_scene& scene) { nrc_check(nrc_renderer_set_scene_params(h_, &scene)); }   // HpmScene::Update
    void SetSceneParams(const HpmScene& scene) { SetSceneParams(scene.Scene()); }
    nrc_renderer_t* Handle() const { return h_; }

private:
    nrc_renderer_t* h_ = nullptr;
    nrc_tile tile_{};
    float stage_ms_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

class McHpmRenderer {
public:
    McHpmRenderer(uint32_t width, uint32_t height, uint32_t pathLength, bool blend, const nrc_camera* camera,
                  const nrc_scene& scene, void* stream = nullptr, const nrc_tile* tile = nullptr)
    {
        nrc_check(nrc_mc_renderer_create(width, height, pathLength, blend ? 1 : 0, camera, &scene, tile, stream, &h_));
    }
    McHpmRenderer(uint32_t width, uint32_t height, uint32_t pathLength, bool blend, const Camera* camera, const HpmScene& scene,
                  void* stream = nullptr, const nrc_tile* tile = nullptr)
        : McHpmRenderer(width, height, pathLength, blend, camera->Matrices(), scene.Scene(), stream, tile)
    {
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
    const float* GetImageView() const { return GetImage(); }      // include/engine/graphics/renderer/McHpmRenderer.hpp:25
    // include/engine/graphics/renderer/McHpmRenderer.hpp:22-23, called once per frame at src/main.cu:284: reads the frame's
    // timestamps; GetFrameTimeMS() is what it found (ms of the last completed frame, 0 before the first)
    void EvaluateTimestampQueries() { frame_ms_ = nrc_mc_renderer_frame_time_ms(h_); }
    float GetFrameTimeMS() const { return frame_ms_; }
    void SetCamera(void* /*queue*/, const nrc_camera* camera) { nrc_check(nrc_mc_renderer_set_camera(h_, camera)); }
    void SetCamera(void* queue, const Camera* camera) { SetCamera(queue, camera->Matrices()); }
    void SetBlend(bool blend) { nrc_check(nrc_mc_renderer_set_blend(h_, blend ? 1 : 0)); }
    bool IsBlending() const { return nrc_mc_renderer_is_blending(h_) != 0; }
    void SetSceneParams(const nrc_scene& scene) { nrc_check(nrc_mc_renderer_set_scene_params(h_, &scene)); }
    void SetSceneParams(const HpmScene& scene) { SetSceneParams(scene.Scene()); }
    nrc_mc_renderer_t* Handle() const { return h_; }

private:
    nrc_mc_renderer_t* h_ = nullptr;
    float frame_ms_ = 0.0f;
};

// en::Reference (src/Reference.cpp): the converged ground-truth image of a scene, seen from a fixed reference camera, and the
// comparison of a renderer's frame with it -- Result{mse, refMean, ownMean, ownVar, validPixelCount} by the three passes of
// data/shader/ref/{cmp1,norm,cmp2}.comp (here nrc_compare_images: deterministic fp64 tree sums).  The constructor does what
// CreateRefCameras / GenRefImages do (:443-455, :566-660): the camera of :447-454; if <referenceRoot><scene id>/ does not exist,
// an McHpmRenderer with PATH_LENGTH 64 blends `generateFrames` (8192) frames from that camera and exports 0.exr there; then
// 0.exr is loaded (it must have the render size) and uploaded.  (The reference's "#if __cplusplus >= 201703L" around the
// generation is inverted -- a C++17 build only warns and then fails to load the image; the intended behaviour is implemented.)
class Reference {
public:
    struct Result {
        float mse = 0.0f;                  // MSE of "not reference" to reference
        float refMean = 0.0f;              // mean of the reference image
        float ownMean = 0.0f;              // mean of the "not reference" image
        float ownVar = 0.0f;               // variance of the "not reference" image
        uint32_t validPixelCount = 0;      // pixels whose reference alpha is not 0
        float GetBias() const { return ownMean - refMean; }                 // src/Reference.cpp:9-27
        float GetRelBias() const { return GetBias() / refMean; }
        float GetRelVar() const { return ownVar / refMean; }
        float GetCV() const { return std::sqrt(ownVar) / ownMean; }
    };

    // Multi-GPU (new): with `tile` (this rank's shard of a tile->global_w x tile->global_h frame; width = the LOCAL column count) and
    // `comm` (the cache whose communicator -- nrc_cache_comm_init or collective hooks -- spans the frame's ranks) the reference image is
    // cut to this rank's columns and Compare* reduce the five sums over the ranks (nrc_compare_images_sharded): every rank gets the
    // whole frame's Result.  The reference image itself must exist then (generate it with a single-GPU run).
    Reference(uint32_t width, uint32_t height, const AppConfig& appConfig, const HpmScene& scene, void* queue,
              const std::string& referenceRoot = "reference/", uint32_t generateFrames = 8192, const nrc_tile* tile = nullptr,
              NeuralRadianceCache* comm = nullptr)
        : m_Width(width), m_Height(height), m_Queue(queue),
          m_RefCamera(vec3(64.0f, 0.0f, 0.0f), vec3(-1.0f, 0.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f),
                      static_cast<float>(tile ? tile->global_w : width) / static_cast<float>(tile ? tile->global_h : height), radians(60.0f), 0.1f, 100.0f),
          m_Comm(comm)
    {
        const std::string dir = referenceRoot + std::to_string(appConfig.scene.id) + "/";
        const std::string path = dir + "0.exr";
        if (tile && tile->x_stride > 1) {
            if (!comm) throw std::runtime_error("SkyRenderer ERROR: a sharded Reference needs the cache whose communicator spans the frame's ranks");
            uint32_t w = 0, h = 0;
            const std::vector<float> rgba = LoadExrRGBA(path, &w, &h);      // throws when the file is missing
            if (w != tile->global_w || h != tile->global_h) throw std::runtime_error("SkyRenderer ERROR: " + path + " has wrong resolution");
            const uint32_t block = tile->x_block ? tile->x_block : 1u;
            std::vector<float> local((size_t)width * height * 4);
            for (uint32_t y = 0; y < height; y++)
                for (uint32_t i = 0; i < width; i++) {
                    const uint32_t gx = (tile->x_offset + (i / block) * tile->x_stride) * block + i % block;      // include/nrc_hpm.h: nrc_tile
                    std::memcpy(&local[((size_t)y * width + i) * 4], &rgba[((size_t)y * w + gx) * 4], 16);
                }
            nrc_check(nrc_image_create(width, height, local.data(), &m_RefImage));
            m_Sharded = true;
            return;
        }
        if (!std::filesystem::is_directory(dir)) {
            std::printf("Reference folder for scene %u was not found. Creating reference images\n", appConfig.scene.id);
            McHpmRenderer refRenderer(width, height, 64, true, &m_RefCamera, scene, queue);
            std::filesystem::create_directories(dir);
            for (uint32_t frame = 0; frame < generateFrames; frame++) refRenderer.Render(queue);
            refRenderer.ExportOutputImageToFile(queue, path);
            refRenderer.Destroy();
        }
        uint32_t w = 0, h = 0;
        const std::vector<float> rgba = LoadExrRGBA(path, &w, &h);
        if (w != width || h != height) throw std::runtime_error("SkyRenderer ERROR: " + path + " has wrong resolution");
        nrc_check(nrc_image_create(width, height, rgba.data(), &m_RefImage));
    }
    ~Reference() { Destroy(); }
    Reference(const Reference&) = delete;
    Reference& operator=(const Reference&) = delete;

    // src/Reference.cpp:72-107: the renderer looks from the reference camera, renders one frame without training, is compared,
    // and gets its camera back
    Result CompareNrc(NrcHpmRenderer& renderer, const Camera* oldCamera, void* queue)
    {
        renderer.SetCamera(queue, &m_RefCamera);
        renderer.Render(queue, false);
        const Result result = Compare(renderer.GetImage());
        renderer.SetCamera(queue, oldCamera);
        return result;
    }
    Result CompareMc(McHpmRenderer& renderer, const Camera* oldCamera, void* queue)      // :109-145
    {
        renderer.SetCamera(queue, &m_RefCamera);
        renderer.Render(queue);
        const Result result = Compare(renderer.GetImage());
        renderer.SetCamera(queue, oldCamera);
        return result;
    }
    void Destroy()
    {
        if (m_RefImage) { nrc_image_destroy(m_RefImage); m_RefImage = nullptr; }
    }
    const float* GetRefImage() const { return m_RefImage; }      // device, RGBA32F [height][width]
    const Camera* GetRefCamera() const { return &m_RefCamera; }

private:
    Result Compare(const float* dOwnImage)
    {
        float r[5];
        if (m_Sharded) nrc_check(nrc_compare_images_sharded(m_Comm->Handle(), m_RefImage, dOwnImage, m_Width * m_Height, m_Queue, r));
        else nrc_check(nrc_compare_images(m_RefImage, dOwnImage, m_Width, m_Height, m_Queue, r));
        Result result;
        result.mse = r[0]; result.refMean = r[1]; result.ownMean = r[2]; result.ownVar = r[3]; result.validPixelCount = (uint32_t)r[4];
        std::printf("MSE: %f | rBias: %f | rVar: %f\n", result.mse, result.GetRelBias(), result.GetRelVar());      // Log::Info, :99-103
        return result;
    }
    uint32_t m_Width, m_Height;
    void* m_Queue;
    Camera m_RefCamera;
    float* m_RefImage = nullptr;
    NeuralRadianceCache* m_Comm = nullptr;
    bool m_Sharded = false;
};

}  // namespace en
