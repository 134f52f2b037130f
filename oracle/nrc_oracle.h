/*
 * oracle/nrc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C ABI of the CPU oracle: a plain restatement of the reference's hot path
 * (GLSL integrator shaders + the tiny-cuda-nn arithmetic it calls).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY PINNING:
 *   - integrator: pinned by the RNG known-answer vectors and, statistically,
 *     by reference/0/0.exr and reference/4/0.exr rendered on data/volume/wdas_cloud_sixteenth.vdb
 *     (tests/golden/, tests/test_oracle_golden.py).  The reference itself cannot be compiled
 *     here (needs Vulkan SDK, glslc, CUDA, tiny-cuda-nn, OpenVDB: none present, no network).
 *   - encoding / MLP / loss / optimizer: tiny-cuda-nn v1.6 is an un-vendored submodule
 *     (.gitmodules:1-4) and the reference holds no test or golden vector at that boundary
 *     => **parity unpinned** for the NN arithmetic; the formulas follow the published
 *     tiny-cuda-nn algorithms as recorded in SURVEY.md App. B and are this build's spec.
 */
#ifndef NRC_ORACLE_H
#define NRC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scene inputs: the values the reference uploads in its UBOs / textures
 * (include/nrc-descriptors.glsl:1-38, nrc-constants.glsl:18-26) */
typedef struct orc_scene {
    const uint8_t* density;      /* R8 UNORM voxels, index i + nx*(j + ny*k)  (src/Texture3D.cpp:99-111, one channel) */
    uint32_t nx, ny, nz;
    float size[3];               /* skySize = normalize(extent)*107.5 (src/NrcHpmRenderer.cu:910-912) */
    float density_factor;        /* VOLUME_DENSITY_FACTOR */
    float g;                     /* VOLUME_G */
    float dir_light_dir[3];
    float dir_light_strength;
    float point_light_pos[3];
    float point_light_strength;
    float point_light_color[3];
    float env_strength;          /* HDR_ENV_MAP_STRENGTH */
    const float* env;            /* RGBA32F, row-major env_h x env_w, LINEAR, clamp-to-edge */
    uint32_t env_w, env_h;
} orc_scene;

typedef struct orc_camera {
    float inv_proj_view[16];     /* column-major (glm) */
    float pos[3];
} orc_camera;

/* compat bits; 0 = faithful to the reference */
#define ORC_FIX_Q1_TRAIN_Y_DIST 1u   /* use trainYDist for TRAIN_Y_DIST */
#define ORC_FIX_Q2_TRAIN_RAY_LEN 2u  /* honour trainRayLength instead of the shader default 1 */

/* ---- RNG (include/random.glsl) ---- */
uint32_t orc_hash(uint32_t x);
float orc_random1(float x);
/* out[0] = state after InitRandom(uv), out[1..n] = n RandFloat(1.0) draws */
void orc_rng_kat(float u, float v, const float frame_random[4], int n, float* out);

/* ---- math spec (oracle/orc_math.h), exported for the GPU bit-parity test ---- */
void orc_math_eval(int fn, const float* a, const float* b, int n, float* out, float* out2);

/* ---- integrator ---- */
/* mc/render.comp.  out_rgba (W*H*4, row-major y*W+x) is read-modify-written (blend); rows [y0,y1). */
void orc_mc_render(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H,
                   uint32_t y0, uint32_t y1, uint32_t path_length, const float frame_random[4],
                   float blend_factor, float* out_rgba, float* info, int n_threads, uint64_t* n_fetch);

/* nrc/gen_rays.comp + nrc/prep_infer_rays.comp.  Images are W*H*4 row-major; infer_input is
 * [W*H][5] indexed x*H+y (zero where the pixel did not scatter, as after vkCmdFillBuffer). */
void orc_nrc_gen_rays(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H,
                      uint32_t y0, uint32_t y1, uint32_t primary_ray_length, float primary_ray_prob,
                      const float frame_random[4], float* primary_rgba, float* info,
                      float* nrc_origin, float* nrc_dir, float* infer_input,
                      int n_threads, uint64_t* n_fetch);
/* the same frame, plus per pixel the number of free flights each of its tracking walks drew, in program order (analysis:
 * tests/walk_model.py); walk_lengths is [H][W][walks_per_pixel], zero-filled by the caller */
void orc_nrc_walk_lengths(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
                          uint32_t primary_ray_length, float primary_ray_prob, const float frame_random[4], float* primary_rgba,
                          float* info, float* nrc_origin, float* nrc_dir, int n_threads, uint16_t* walk_lengths, uint32_t walks_per_pixel);

/* nrc/clear.comp + nrc/prep_train_rays.comp, deterministic two-phase ring semantics. */
void orc_nrc_prep_train(const orc_scene* sc, uint32_t W, uint32_t H, uint32_t TW, uint32_t TH,
                        uint32_t x_dist, uint32_t y_dist, uint32_t train_spp, uint32_t train_ray_length,
                        uint32_t ring_size, const float frame_random[4],
                        const float* info, const float* nrc_origin, const float* nrc_dir,
                        uint32_t* ring_head_tail, float* ring,
                        float* train_input, float* train_target, int n_threads);

/* nrc/render.comp */
void orc_nrc_composite(uint32_t W, uint32_t H, uint32_t show_nrc, float blend_factor,
                       const float* primary_rgba, const float* info, const float* infer_output,
                       float* out_rgba);

/* ref/cmp1,norm,cmp2.comp: result = {mse, refMean, ownMean, ownVar, validPixelCount} */
void orc_compare(const float* ref_rgba, const float* own_rgba, uint32_t W, uint32_t H, float* result5);

/* ---- neural radiance cache arithmetic (tiny-cuda-nn v1.6 semantics) ---- */
typedef struct orc_nn_config {
    uint32_t pos_id, dir_id;     /* AppConfig::NNEncodingConfig (src/AppConfig.cpp:11-87) */
    uint32_t width, depth;       /* n_neurons, n_hidden_layers */
    uint32_t loss_id;            /* 0 RelativeL2Luminance, 1 L2, 2 RelativeL2 */
    float learning_rate, ema_decay;
    uint32_t seed;
    uint32_t hashgrid_log2_size; /* posID 0 only: log2_hashmap_size (0 = 19, src/AppConfig.cpp:24) */
    uint32_t optimizer_id;       /* nested optimizer of the EMA wrapper (src/NeuralRadianceCache.cu:20-28): 0 Adam, 1 SGD */
} orc_nn_config;

void* orc_nn_create(const orc_nn_config* cfg);
void orc_nn_destroy(void* nn);
uint32_t orc_nn_param_count(void* nn);
uint32_t orc_nn_encoded_dims(void* nn);
/* which: 0 master weights, 1 EMA weights, 2 adam m, 3 adam v, 4 last gradient (already / loss_scale);
 * layout: MLP matrices, then (posID 0) the hash-grid table [entry][2] */
uint32_t orc_nn_mlp_param_count(void* nn);
float* orc_nn_buffer(void* nn, int which);
void orc_nn_set_step(void* nn, uint32_t step);
/* 13 * width initial values of rows 3..15 of tiny-cuda-nn's padded 16 x width output matrix (drawn by the init, read by nothing) */
const float* orc_nn_tcnn_dead_rows(void* nn);
/* pcg32 (seed, stream): n raw outputs, and the n floats next_float() makes of the same stream */
void orc_pcg32(uint64_t seed, uint64_t seq, uint32_t n, uint32_t* out_u32, float* out_float);
/* features after fp16 rounding, [n][encoded_dims] */
void orc_nn_encode(void* nn, const float* in, uint32_t n, float* out);
/* mode 0: fp32 everywhere; mode 1: fp16 weights/activations, wide accumulate (the product's mode) */
void orc_nn_forward(void* nn, const float* in, uint32_t n, int use_ema, int mode, float* out);
/* forward(non-EMA weights) + loss + backward into buffer 4.  loss normaliser N = 3*n_norm.
 * returns the summed loss (trainer->loss).  accumulate != 0 adds into buffer 4. */
float orc_nn_backward(void* nn, const float* in, const float* target, uint32_t n, uint32_t n_norm,
                      int accumulate);
/* Adam or SGD (+L2 1e-8) then EMA, from buffer 4 */
void orc_nn_optimizer_step(void* nn);

#ifdef __cplusplus
}
#endif
#endif
