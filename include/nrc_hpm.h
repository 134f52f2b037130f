/*
 * include/nrc_hpm.h -- C ABI of the MI355X-native NRC-HPM hot path (libnrc_hpm.so).
 *
 * Drop-in boundary for the reference's hot path.  Each entry point cites the
 * reference interface it replaces (paths relative to the reference checkout).  Plain pointers and sizes only;
 * `stream` arguments are hipStream_t passed as void* (0 = the null stream).  All functions return NRC_OK (0) or
 * a negative error code and never abort; nrc_last_error() gives the thread-local message (the reference throws
 * std::runtime_error("SkyRenderer ERROR: ..."), src/Log.cpp:16-20 -- the C++ layer in nrc_hpm.hpp does the same).
 *
 * Vulkan/CUDA-interop types of the reference are replaced as follows:
 *   VkQueue                         -> hipStream_t (stream order replaces both external semaphores)
 *   cudaExternalSemaphore_t x2      -> dropped
 *   VkImage / VkImageView           -> device pointer to an RGBA32F row-major framebuffer
 */
#ifndef NRC_HPM_H
#define NRC_HPM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRC_OK 0
#define NRC_ERR_INVALID (-1)    /* bad argument / unsupported configuration */
#define NRC_ERR_HIP (-2)        /* a HIP runtime call failed (message holds hipGetErrorString) */
#define NRC_ERR_STATE (-3)      /* call order violated (e.g. InferAndTrain before Init) */
#define NRC_ERR_COMM (-4)       /* multi-GPU: a collective failed or a peer did not answer in time; the communicator is aborted (see
                                 * nrc_cache_comm_status) -- every later call that needs it fails at once, destroy the cache */

const char* nrc_last_error(void);
const char* nrc_version(void);

/* ---------------------------------------------------------------------------------------------------------
 * en::AppConfig (include/engine/AppConfig.hpp:9-66; 17 positional CLI args src/AppConfig.cpp:154-182,
 * defaults src/main.cu:429-440).  Scene preset values (src/AppConfig.cpp:93-150) are carried explicitly. */
/* The reference passes lossFn / optimizer through to tiny-cuda-nn unchecked (src/NeuralRadianceCache.cu:17-26); this build implements a
 * CLOSED set -- the seven losses and two optimizers named below -- and nrc_cache_create fails with NRC_ERR_INVALID and a message that lists
 * them for any other tiny-cuda-nn name (Novograd, Shampoo, ...: update rules that could not be restated without the absent submodule;
 * CrossEntropy, Variance: they need a sample pdf the NRC path does not have). */
typedef struct nrc_config {
    char loss_fn[32];              /* "RelativeL2Luminance" | "L2" | "RelativeL2" | "L1" | "Mape" | "Smape" | "LogL1" (tiny-cuda-nn names) */
    char optimizer[32];            /* "Adam" (default) | "SGD": nested in the EMA wrapper, src/NeuralRadianceCache.cu:20-28 */
    float learning_rate;
    float ema_decay;
    uint32_t pos_id;               /* 0 HashGrid(16x2, 2^19) | 1 Identity | 2 TriangleWave-12 | 3 Frequency-12 */
    uint32_t dir_id;               /* 0 OneBlob-4 | 1 Identity | 2 TriangleWave-4 */
    uint32_t nn_width;             /* 64 (16, 32, 64, 128: tiny-cuda-nn's FullyFusedMLP widths) */
    uint32_t nn_depth;             /* n_hidden_layers, 6 */
    uint32_t log2_infer_batch_size;
    uint32_t log2_train_batch_size;
    uint32_t train_batch_count;
    uint32_t scene_id;
    float train_ring_buf_size;
    uint32_t train_spp;
    uint32_t primary_ray_length;
    float primary_ray_prob;
    uint32_t train_ray_length;
    /* additions of this build */
    uint32_t seed;                 /* weight-init seed (tiny-cuda-nn default 1337) */
    uint32_t compat_fix;           /* bit 0: fix quirk Q1 (TRAIN_Y_DIST), bit 1: fix quirk Q2 (trainRayLength); 0 = faithful */
    uint32_t hashgrid_log2_size;   /* posID 0: log2_hashmap_size; 0 = the reference's 19 (src/AppConfig.cpp:24) */
} nrc_config;

#define NRC_FIX_Q1_TRAIN_Y_DIST 1u
#define NRC_FIX_Q2_TRAIN_RAY_LEN 2u

/* fills the defaults of src/main.cu:432-439, except for the position encoding: posID 3 (Frequency, the north-star model with
 * the fully fused kernels) instead of the reference's 0 (HashGrid) -- set pos_id = 0 for the reference's exact default */
void nrc_config_default(nrc_config* cfg);

/* ---------------------------------------------------------------------------------------------------------
 * en::NeuralRadianceCache (include/engine/graphics/NeuralRadianceCache.hpp:10-64, src/NeuralRadianceCache.cu) */
typedef struct nrc_cache nrc_cache_t;

/* NeuralRadianceCache::NeuralRadianceCache(const AppConfig&)  (src/NeuralRadianceCache.cu:11-40) */
int nrc_cache_create(const nrc_config* cfg, nrc_cache_t** out);
/* NeuralRadianceCache::Init(inferCount, dCuInferInput, dCuInferOutput, dCuTrainInput, dCuTrainTarget, sem, sem)
 * (src/NeuralRadianceCache.cu:42-95).  Buffers are caller-owned DEVICE memory: [n][5] / [n][3] fp32 AoS. */
int nrc_cache_init(nrc_cache_t* c, uint32_t infer_count, float* d_infer_input, float* d_infer_output,
                   float* d_train_input, float* d_train_target, void* stream);
/* The same Init with the reference's semaphore pair kept in HIP terms (include/engine/graphics/NeuralRadianceCache.hpp:15-22:
 * cudaExternalSemaphore_t cudaStartSemaphore, cudaFinishedSemaphore): start_event / finished_event are hipEvent_t of the caller
 * (either may be NULL).  Every InferAndTrain first makes its stream wait for start_event (AwaitCudaStartSemaphore,
 * src/NeuralRadianceCache.cu:158-167) and records finished_event behind its last kernel (SignalCudaFinishedSemaphore, :169-177). */
int nrc_cache_init_events(nrc_cache_t* c, uint32_t infer_count, float* d_infer_input, float* d_infer_output,
                          float* d_train_input, float* d_train_target, void* stream, void* start_event, void* finished_event);
/* NeuralRadianceCache::InferAndTrain(const uint32_t* inferFilter, bool train) (src/NeuralRadianceCache.cu:97-103).
 * infer_filter: HOST array, one entry per inference batch, >0 => run; NULL => run every batch. */
int nrc_cache_infer_and_train(nrc_cache_t* c, const uint32_t* infer_filter, int train);
/* NeuralRadianceCache::Destroy (src/NeuralRadianceCache.cu:105-107); frees the object */
int nrc_cache_destroy(nrc_cache_t* c);
/* GetLoss / Get{Infer,Train}Batch{Count,Size} (src/NeuralRadianceCache.cu:109-132).
 * The reference reads the loss back synchronously inside every training step (trainer->loss, :154) and GetLoss() returns that
 * value; its main loop polls it every frame (src/main.cu:303,376).  Here a one-thread kernel behind every training step stores
 * {loss, step number} into host-mapped pinned memory on the training stream:
 *   nrc_cache_get_loss        reference semantics: the loss of the last training step that was ENQUEUED (InferAndTrain / Render
 *                             with train, nrc_cache_backward); waits for that step only, not for the device.  0 before the first.
 *                             (nrc_cache_get_loss_blocking is the same call under its round-2 name.)
 *   nrc_cache_get_loss_async  never blocks and never drains the renderer's frame pipeline: *loss = the loss of the most recent
 *                             step that has COMPLETED, *step = that step's number (1-based; 0: none yet), *steps_enqueued = the
 *                             number of the newest step enqueued -- so a caller sees how stale the value is (at most the
 *                             pipeline depth, four frames).  step / steps_enqueued may be NULL. */
float nrc_cache_get_loss(nrc_cache_t* c);
float nrc_cache_get_loss_blocking(nrc_cache_t* c);
int nrc_cache_get_loss_async(nrc_cache_t* c, float* loss, uint32_t* step, uint32_t* steps_enqueued);
size_t nrc_cache_get_infer_batch_count(nrc_cache_t* c);
size_t nrc_cache_get_train_batch_count(nrc_cache_t* c);
uint32_t nrc_cache_get_infer_batch_size(nrc_cache_t* c);
uint32_t nrc_cache_get_train_batch_size(nrc_cache_t* c);

/* --- finer-grained steps of the same path (what InferAndTrain is made of; used by the multi-GPU driver) --- */
/* network->inference on an arbitrary device buffer (src/NeuralRadianceCache.cu:142); use_ema=1 is what Inference does */
int nrc_cache_infer(nrc_cache_t* c, const float* d_input, float* d_output, uint32_t n, int use_ema);
/* trainer->training_step minus the optimizer (src/NeuralRadianceCache.cu:153): forward (non-EMA weights) + loss + backward.
 * n_norm: batch size the loss normaliser uses (= n on one GPU, the global batch when the batch is sharded). */
int nrc_cache_backward(nrc_cache_t* c, const float* d_input, const float* d_target, uint32_t n, uint32_t n_norm);
/* optimizer step (EMA{Adam}) from the gradient vector */
int nrc_cache_optimizer_step(nrc_cache_t* c);
/* device pointer / length of the fp32 gradient vector (sum over the local batch, times loss_scale 128) and of the
 * 2-float {loss, unused} cell, which directly follows the gradient vector in memory (loss_ptr == grad_ptr +
 * param_count): the multi-GPU driver all-reduces param_count + 2 floats between backward and optimizer_step.  From the first
 * call on, nrc_cache_optimizer_step reads this vector for EVERY parameter (a HashGrid table's gradient otherwise comes from the
 * packed fp16 table the backward pass accumulated into; unmodified, the vector holds the same values) */
float* nrc_cache_grad_ptr(nrc_cache_t* c);
uint32_t nrc_cache_param_count(nrc_cache_t* c);
float* nrc_cache_loss_ptr(nrc_cache_t* c);
/* multi-GPU, native exchange: rank 0 obtains a 128-byte RCCL unique id (nrc_comm_unique_id), distributes it, and every
 * rank calls nrc_cache_comm_init(id, rank, world) (collective).  From then on every train batch all-reduces (sum) the
 * gradient vector + loss cell with ncclAllReduce on the training stream and normalises the loss by the global batch. */
int nrc_comm_unique_id(void* out128);
int nrc_cache_comm_init(nrc_cache_t* c, const void* unique_id128, int rank, int world);
/* What travels in that exchange (BASELINE.json configs[3]: "RCCL fp16 grad all-reduce"; SURVEY.md 8e recommends both):
 *   NRC_EXCHANGE_F32 (default)  the fp32 gradient vector + loss cell, one ncclAllReduce(ncclFloat) of param_count + 2 words (103 KB for 6 x 64)
 *   NRC_EXCHANGE_F16            the gradients as tiny-cuda-nn holds them -- fp16, pre-scaled by loss_scale 128 (the local sum is accumulated in
 *                               fp32 and rounded ONCE) -- summed by ncclAllReduce(ncclHalf) (52 KB), widened back into the fp32 vector the
 *                               optimizer reads; the loss cell stays fp32, grouped with it.  A HashGrid model's matrices travel the same way,
 *                               its table lists carry fp16 values in either mode.  Replicas stay bit-identical (every rank receives the same
 *                               sum).  A gradient hook (nrc_cache_set_grad_hook) is handed fp16-rounded values in the fp32 vector and its
 *                               result is rounded again: for two ranks exactly the fp16 sum.
 * Set on every rank alike, before the first training step; nrc_cache_get_exchange_dtype returns the mode in use. */
#define NRC_EXCHANGE_F32 0
#define NRC_EXCHANGE_F16 1
int nrc_cache_set_exchange_dtype(nrc_cache_t* c, int dtype);
int nrc_cache_get_exchange_dtype(nrc_cache_t* c);
/* rank / size as the library's own RCCL communicator reports them (ncclCommUserRank / ncclCommCount); world = 0: none */
int nrc_cache_comm_info(nrc_cache_t* c, int* rank, int* world);
/* Failure detection on the exchange ("RCCL error -> status code").  An enqueued collective reports nothing by itself:
 * every call into the library that uses the communicator first polls ncclCommGetAsyncError (non-blocking), and every wait of the library
 * for work that sits behind a collective -- the frame gather, the sharded metrics, nrc_cache_get_loss_blocking, the collective export --
 * has a deadline (default 30 000 ms once the frame has more than one rank; 0 = wait for ever).  On an asynchronous error, a collective
 * hook that returns non-zero, or a missed deadline the communicator is aborted (ncclCommAbort: its kernels are killed, nothing is left
 * hanging on the streams), the call returns NRC_ERR_COMM with the reason, and so does every later call that would need the exchange;
 * the ranks that still work see the same through their own deadline.  nrc_cache_comm_status: NRC_OK, or NRC_ERR_COMM (polls, never blocks). */
int nrc_cache_comm_status(nrc_cache_t* c);
int nrc_cache_set_comm_timeout_ms(nrc_cache_t* c, uint32_t ms);
/* measurement: *avg_us = average duration of the training step's ncclAllReduce (gradient vector + loss cell, zeroed first), issued
 * `reps` times back to back on the cache's stream; collective -- every rank calls it with the same reps.  0 without a communicator. */
int nrc_cache_comm_time_exchange(nrc_cache_t* c, uint32_t reps, float* avg_us);
/* HashGrid models (posID 0): the trainable table's gradient -- 57 MB dense, a few per cent of it touched by a batch -- is
 * exchanged as all-gathered (entry, fp16x2 value) lists that every rank adds in rank order (replicas stay bit-identical), the
 * matrix gradients and the loss cell by ncclAllReduce as before.  Chosen at nrc_cache_comm_init when world x list capacity
 * (trainBatchSize x 16 levels x 8 corners, at most the table) < 2 x table entries, i.e. when the padded all-gather moves less
 * than the ring all-reduce; environment NRC_DEBUG=dense_grid_exchange forces the dense exchange.  nrc_cache_comm_sparse: 1 when the list exchange is the one in use.  The two debug entry points expose its
 * halves on one device (tests): the list of the last nrc_cache_backward -- words {count, 0, (entry, value) x capacity},
 * padding entries 0xffffffff, list_words >= 2 + 2 * nrc_cache_grid_list_capacity() -- and the gradient vector's table part
 * := sum of n_lists such lists (2 + 2 * capacity words apart), added in list order. */
int nrc_cache_comm_sparse(nrc_cache_t* c);
size_t nrc_cache_grid_list_capacity(nrc_cache_t* c);
int nrc_cache_grid_grad_pack(nrc_cache_t* c, uint32_t* host_list, size_t list_words);
int nrc_cache_grid_grad_apply(nrc_cache_t* c, const uint32_t* host_lists, uint32_t n_lists);
/* multi-GPU: the loss normaliser of InferAndTrain's train batches becomes 3 * trainBatchSize * factor (factor = world size) */
int nrc_cache_set_loss_norm_factor(nrc_cache_t* c, uint32_t factor);
/* move the cache's work to another hipStream_t (Init binds the first one) */
int nrc_cache_set_stream(nrc_cache_t* c, void* stream);
/* hook called between backward and the optimizer of every train batch (NULL = none); `stream` is the hipStream_t the
 * training kernels are ordered on -- the exchange (all-reduce) must be issued on it */
typedef void (*nrc_grad_hook)(void* user, float* d_grad, uint32_t n_params, float* d_loss, void* stream);
int nrc_cache_set_grad_hook(nrc_cache_t* c, nrc_grad_hook hook, void* user);
/* Collectives of a multi-GPU frame that are NOT the gradient exchange -- the frame gather and the metric reduction below -- go through
 * the communicator nrc_cache_comm_init made; a cache without one can be given the caller's transport instead (the tests' gloo
 * rehearsal, a host that brings MPI): allreduce sums n doubles in place over all ranks, allgather fills d_recv = [world][bytes_per_rank]
 * with every rank's d_send; both work on device memory and return 0 on success.  The library has waited for `stream` (a hipStream_t)
 * when it calls a hook, so a transport that stages through host memory may read the buffers at once; what the hook writes must be
 * complete, or ordered on `stream`, when it returns. */
typedef int (*nrc_allreduce_f64_fn)(void* user, double* d_buf, uint32_t n, void* stream);
typedef int (*nrc_allgather_fn)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);
int nrc_cache_set_collective_hooks(nrc_cache_t* c, int rank, int world, nrc_allreduce_f64_fn allreduce, nrc_allgather_fn allgather, void* user);
/* checkpointing: which = 0 master weights, 1 EMA weights, 2 Adam m, 3 Adam v, 4 gradient (host fp32 arrays) */
int nrc_cache_get_params(nrc_cache_t* c, int which, float* host_out);
int nrc_cache_set_params(nrc_cache_t* c, int which, const float* host_in);
/* the same vectors in tiny-cuda-nn v1.6's OWN parameter layout -- what a dump of its Trainer's params_full_precision holds for the model
 * tcnn::create_from_config(5, 3, cfg) builds at src/NeuralRadianceCache.cu:39, so that weights saved from the reference can be loaded
 * (and the other way round): the matrices in the same order and row-major orientation, the output matrix with its rows padded to 16
 * (16 x nnWidth; rows 3..15 feed the padded outputs nobody reads: kept on the host as initialised / last set, never trained), then the
 * encoding's table.  count = nrc_cache_param_count + 13 * nnWidth (26 624 for the 6 x 64 model with the 80-wide encoding). */
uint32_t nrc_cache_param_count_tcnn(nrc_cache_t* c);
int nrc_cache_get_params_tcnn(nrc_cache_t* c, int which, float* host_out);
int nrc_cache_set_params_tcnn(nrc_cache_t* c, int which, const float* host_in);
/* checkpoint FILE (the reference has none; SURVEY.md section 5 "checkpoint / resume"): little-endian, 64-byte header {"NRCCKPT1", posID,
 * dirID, nnWidth, nnDepth, hashgrid log2 size, tcnn parameter count, optimizer step, 0...} + the four vectors which = 0..3 in the
 * tiny-cuda-nn layout above.  load checks the header against the cache's model (NRC_ERR_INVALID on a mismatch, a short or unreadable
 * file) and leaves the cache untouched on failure. */
int nrc_cache_save_checkpoint(nrc_cache_t* c, const char* path);
int nrc_cache_load_checkpoint(nrc_cache_t* c, const char* path);
int nrc_cache_get_step(nrc_cache_t* c, uint32_t* step);
int nrc_cache_set_step(nrc_cache_t* c, uint32_t step);

/* ---------------------------------------------------------------------------------------------------------
 * Scene inputs: the values the reference uploads as UBOs / textures.  HOST pointers; copied
 * to the device at renderer creation. */
typedef struct nrc_scene {
    const uint8_t* density;        /* R8 UNORM voxels, index i + nx*(j + ny*k)  (src/Texture3D.cpp:99-111, one channel) */
    uint32_t nx, ny, nz;
    float size[3];                 /* skySize; all zero => normalize(extent)*107.5 (src/NrcHpmRenderer.cu:910-912) */
    float density_factor;          /* VolumeData density (scene preset) */
    float g;                       /* 0.8 (src/HpmScene.cpp:45) */
    float dir_light_dir[3];        /* src/DirLight.cpp:5-14 */
    float dir_light_strength;
    float point_light_pos[3];
    float point_light_strength;
    float point_light_color[3];
    float env_strength;
    const float* env;              /* RGBA32F row-major env_h x env_w (src/HdrEnvMap.cpp), LINEAR clamp-to-edge */
    uint32_t env_w, env_h;
} nrc_scene;

/* CameraMatrices.invProjView + camera.pos (include/engine/graphics/Camera.hpp:11-16, nrc-descriptors.glsl:1-11) */
typedef struct nrc_camera {
    float inv_proj_view[16];       /* column-major (glm) */
    float pos[3];
} nrc_camera;

/* Pixel-tile shard of a frame (new: SURVEY.md section 8e).  This instance renders strips of x_block adjacent columns of a
 * global_w x global_h frame, every x_stride-th strip starting with strip x_offset: local column i (i = 0..width-1) is the global
 * column (x_offset + (i / x_block) * x_stride) * x_block + i % x_block; width passed to create is the LOCAL column count.
 * x_block must be a power of two; 0 means 1 (single interleaved columns).  {0,1,W,H,0} = whole frame.  Strips of 8 columns keep
 * the 8x8-pixel tile a wavefront renders contiguous on the screen (coherent walks, as on one GPU) at the same load balance.
 * ABI note: the struct has FIVE fields (20 bytes) since nrc_version() "0.2"; "0.1" had four (16 bytes, no x_block) -- a caller built
 * against the older header must be recompiled (the library would read x_block past the end of its struct). */
typedef struct nrc_tile {
    uint32_t x_offset, x_stride, global_w, global_h;
    uint32_t x_block;
} nrc_tile;

/* ---------------------------------------------------------------------------------------------------------
 * en::NrcHpmRenderer (include/engine/graphics/renderer/NrcHpmRenderer.hpp:13-41, src/NrcHpmRenderer.cu) */
typedef struct nrc_renderer nrc_renderer_t;

/* NrcHpmRenderer(width, height, blend, camera, appConfig, hpmScene, nrc) (src/NrcHpmRenderer.cu:212-297).
 * The renderer allocates the four NRC I/O buffers and calls nrc_cache_init (as the reference ctor does, :259-266).
 * tile may be NULL. */
int nrc_renderer_create(uint32_t width, uint32_t height, int blend, const nrc_camera* camera, const nrc_config* cfg,
                        const nrc_scene* scene, nrc_cache_t* cache, const nrc_tile* tile, void* stream,
                        nrc_renderer_t** out);
/* NrcHpmRenderer::Render(VkQueue, bool train) (src/NrcHpmRenderer.cu:299-353) */
int nrc_renderer_render(nrc_renderer_t* r, int train);
/* n_frames consecutive Render(queue, train) calls enqueued by ONE call (the host loop of src/main.cu:287 without a trip through the
 * binding per frame); frame_randoms: n_frames x 4 floats, frame f's UniformData.random (NULL: the renderer draws them, as Render does) */
int nrc_renderer_render_frames(nrc_renderer_t* r, uint32_t n_frames, const float* frame_randoms, int train);
/* NrcHpmRenderer::SetCamera / SetBlend (src/NrcHpmRenderer.cu:561-610) */
int nrc_renderer_set_camera(nrc_renderer_t* r, const nrc_camera* camera);
int nrc_renderer_set_blend(nrc_renderer_t* r, int blend);
/* The uniform-buffer half of the scene -- DirLight / PointLight / VolumeData / HdrEnvMap strength (src/DirLight.cpp:31-49,
 * src/HpmScene.cpp:56-76 `HpmScene::Update`, the ImGui light editors): takes effect with the next Render and does not reset
 * blending (only a camera change does, src/NrcHpmRenderer.cu:561-604).  The density volume and the environment map of `scene`
 * are ignored (textures are fixed at creation). */
int nrc_renderer_set_scene_params(nrc_renderer_t* r, const nrc_scene* scene);
/* UniformData.showNrc (include/engine/graphics/renderer/NrcHpmRenderer.hpp:70-75) */
int nrc_renderer_set_show_nrc(nrc_renderer_t* r, int show);
/* UniformData.random: by default drawn per frame from std::mt19937(seed) (the reference uses glm::linearRand,
 * src/NrcHpmRenderer.cu:308); this pins the next frame's value (parity tests) */
int nrc_renderer_set_frame_random(nrc_renderer_t* r, const float random4[4]);
/* GetImage()/GetImageView() -> device pointer, RGBA32F, row-major height x width (local columns).  The renderer composites
 * on an internal stream; this call makes the stream passed to nrc_renderer_create wait (on the device) for the latest frame's
 * compositing, so work enqueued on that stream afterwards sees the finished image.  Call it again after every Render. */
const float* nrc_renderer_framebuffer(nrc_renderer_t* r);
/* same image, but orders `consumer_stream` (a display or read-back stream of the caller) behind the latest compositing
 * instead of the render stream: reading every frame then does not hold back the next frame's ray generation.  Pair it with
 * nrc_renderer_release_frame once the reads are enqueued. */
const float* nrc_renderer_framebuffer_on(nrc_renderer_t* r, void* consumer_stream);
/* The framebuffer is ONE image that every frame's compositing blends in place (as the reference's m_OutputImage).  A consumer
 * that reads it on a stream of its own (framebuffer_on) must say when it is done: release_frame records the end of the reads
 * enqueued so far on consumer_stream, and the next frame's compositing waits for it on the device (nothing else of the next
 * frame does).  Without the call, a read that is still running when the next frame composites sees a torn image.
 * Not needed for nrc_renderer_framebuffer(): reads enqueued on the render stream are ordered before the next frame by
 * stream order. */
int nrc_renderer_release_frame(nrc_renderer_t* r, void* consumer_stream);
/* NrcHpmRenderer::IsBlending (include/engine/graphics/renderer/NrcHpmRenderer.hpp:37) */
int nrc_renderer_is_blending(nrc_renderer_t* r);
/* ExportOutputImageToFile (src/NrcHpmRenderer.cu:437-493): scan-line EXR, FLOAT RGBA */
int nrc_renderer_export_exr(nrc_renderer_t* r, const char* path);
/* Multi-GPU (new: SURVEY.md section 8e, "gather tiles to rank 0 or reduce only the metrics").  A frame sharded by nrc_tile is put
 * together again: every rank of the frame calls, every rank receives the whole [global_h][global_w] RGBA32F image in d_global_rgba
 * (device).  One all-gather of the ranks' column strips through the cache's communicator (nrc_cache_comm_init, or the hooks of
 * nrc_cache_set_collective_hooks) and a de-interleave kernel; bit-identical to the frame one GPU renders.  An unsharded renderer
 * copies its framebuffer.  export_exr_gathered is ExportOutputImageToFile (src/NrcHpmRenderer.cu:437-493) of such a run: collective,
 * rank `root` writes the file. */
int nrc_renderer_gather_frame(nrc_renderer_t* r, float* d_global_rgba, void* stream);
int nrc_renderer_export_exr_gathered(nrc_renderer_t* r, const char* path, int root);
/* EvaluateTimestampQueries + GetFrameTimeMS (src/NrcHpmRenderer.cu:495-530,556-559): synchronises; stage_ms may be
 * NULL or float[8] = {clear(0), gen_rays, prep_infer(0: fused into gen_rays), train, prep_train, inference, composite,
 * total}.  The renderer pipelines frames over four streams (train-ray generation, training, inference + compositing of
 * frame N run beside gen_rays of frame N+1; DESIGN.md section 4 "Frame graph"), so the stages overlap, include the time they
 * wait for each other, and do not add up; total = latency of the frame, which exceeds the frame interval. */
float nrc_renderer_frame_time_ms(nrc_renderer_t* r, float* stage_ms);
/* the same stage times averaged over every frame rendered since the last reset (HIP events on the renderer's streams);
 * *frames = number of frames covered.  avg_ms == NULL with reset != 0 only forgets the frames so far, without reading an event */
int nrc_renderer_stage_stats(nrc_renderer_t* r, float avg_ms[8], uint32_t* frames, int reset);
/* Wave issue priority of the library's kernels (process-wide, current device): 1 (default) raises every kernel to s_setprio 3 -- no kernel of
 * the host can then outrank a path-integrator wave (the second guard behind the compiler flag) --, 0 leaves them all at
 * the hardware default.  Either way the whole library runs at ONE priority and the frame rate is the same; 0 is for a process whose
 * foreign kernels run beside the renderer (RCCL's all-reduce, the runtime's fills and copies): at priority 0 under waves at 3 a 28 MB
 * hipMemsetAsync took 184 us inside a frame, 65 us among equals.  nrc_cache_comm_init selects 0 for world > 1. */
int nrc_set_wave_priority_raise(int on);
/* the same events as a timeline (the reference's per-frame timestamp queries, src/NrcHpmRenderer.cu:495-515, kept for every frame since
 * the last reset): times_ms[f * 6 + k] = milliseconds from the first frame's start to event k of frame f -- 0 gen_rays starts, 1 gen_rays
 * done, 2 train rays done, 3 inference done, 4 compositing done, 5 training done -- for the first min(*frames, max_frames) frames;
 * synchronises, resets nothing.  Shows which stream a pipelined frame waits for (tools/frame_timeline.py). */
int nrc_renderer_frame_timeline(nrc_renderer_t* r, float* times_ms, uint32_t max_frames, uint32_t* frames);
/* NrcHpmRenderer::Destroy */
int nrc_renderer_destroy(nrc_renderer_t* r);
/* intermediate device buffers of the most recent frame, after synchronising all of the renderer's streams (tests /
 * multi-GPU; the sets rotate, so ask again after every Render): 0 primary colour+throughput [h][w][4], 1 primary info [h][w],
 * 2 nrc ray origin [h][w][4], 3 nrc ray dir [h][w][4] (train-grid pixels only unless set_full_vertex_images), 4 infer input [w*h][5], 5 infer output [w*h][3],
 * 6 train input [T][5], 7 train target [T][3], 8 train ring {head, tail, RayInfo[ring]}.  Buffers 4 and 5 are handed out in the
 * reference's order, query x*H+y (nrc/prep_infer_rays.comp:31), as a COPY made by this call: inside the renderer they are
 * tile-major (the 64 queries of an 8x8 pixel tile contiguous), and the cache's own API (nrc_cache_init / infer) speaks x*H+y
 * whatever its caller's order is.  Entries of pixels that did not scatter (info != 1) are zeros in both copies: the reference's zero-filled
 * query slots (src/NrcHpmRenderer.cu:1996), and a radiance nrc/render.comp:19-27 never reads -- the renderer's gen_rays and inference walk
 * the frame's list of scattered pixels and write nothing for the others. */
void* nrc_renderer_buffer(nrc_renderer_t* r, int which, size_t* bytes);
/* Empty-space early-out (on by default): camera rays that provably cannot come within a voxel of non-empty density skip their
 * delta-tracking walk -- the walk could only reject every tentative collision, leave the volume unscattered and produce env(rd),
 * and the RNG state behind it is never read, so the frame is bit-identical with and without (tests compare both against the
 * oracle, which always walks).  The 8x8-pixel tile mask behind it is rebuilt on the render stream whenever the camera changes.
 * on = 0 traces every ray (what count_fetches needs to report the ALGORITHM's look-ups rather than the executed ones). */
int nrc_renderer_set_empty_skip(nrc_renderer_t* r, int on);
/* Costliest-first launch order of gen_rays' 8x8-pixel tiles (on by default): every 16th frame records what each tile's wave cost,
 * a counting sort turns that into the order the following frames start their tiles in (the long walks through the cloud first,
 * the cheap rim in the tail of the launch).  The order is a permutation of the tiles and nothing else: frames are bit-identical
 * with and without.  tile_order copies the permutation in use (n = nrc_renderer_tile_order(r, NULL, 0) entries) to the host. */
/* The SCHEDULE of a renderer's frame graph: where and in which order work is placed -- no value changes a pixel or a weight (the tests render
 * under every combination and compare bit for bit).  The library chooses these itself: it starts from neutral values and, once the pipeline
 * has run for 128 frames, tries the alternatives on the caller's own frames against its frame timeline (one knob at a time, 24 frames per value,
 * the current value timed before and after the alternatives; ~250 frames in all) and keeps what is more than 1.5 % faster.  A caller that
 * knows better pins a knob with a value >= 0; -1 hands it (back) to the library.
 *   camera_priority_low  1: the camera kernels run at the default wave priority under the library's other kernels (pays where the
 *                        inference -> training chain bounds the frame: wide dense models), 0: at the common priority
 *   cost_order_lag       frames between a tile-cost sample and the first launch ordered by it (1..64; 2 or 3 are what the tuner tries)
 *   xcd_window           XCD-aware finish of the costliest-first launch order: tiles of a window of 32 x M ranks are dealt to the eight XCDs by
 *                        screen row (0 = off; the tuner tries 0, 2, 16); ignored on a device that does not have eight XCDs
 *   composite_defer      1: the compositing of the frames inside one nrc_renderer_render_frames call runs on the train-ray stream (not tuned) */
typedef struct nrc_schedule {
    int32_t camera_priority_low, cost_order_lag, xcd_window, composite_defer;
} nrc_schedule;
int nrc_renderer_set_schedule(nrc_renderer_t* r, const nrc_schedule* schedule);
/* the values in use now; *tuning_done (may be NULL) = 1 once nothing is left to choose */
int nrc_renderer_get_schedule(nrc_renderer_t* r, nrc_schedule* current, int* tuning_done);
/* Schedules survive the process.  What the tuner settles on is remembered in a process-wide table under a key that names everything the
 * choice depends on -- "<arch>:<CUs>cu:<XCDs>xcd|<model>|vol2^<log2 voxels>|<w>x<h>.of<global w>x<global h>|train<batches>x<rays>.len<train ray length>"
 * (nrc_renderer_schedule_key) -- and a renderer created while its key is in the table starts on that schedule and skips the trials, so a run
 * shorter than the tuner's ~400 frames is a tuned run all the same.  nrc_schedule_cache_save writes the table (text: one "<key> <pri> <lag>
 * <window>" line per entry), nrc_schedule_cache_load merges a file into it (damaged lines are skipped; entries of other devices or models
 * simply never match); *n_entries (may be NULL) = entries read / written.  nrc_renderer_schedule_source says where the schedule in use came
 * from: "default" | "cache" | "tuner" (this renderer's own trials have ended) | "pinned" | "pinned in part".  The Python mirror loads
 * nrc-hpm-renderer_amd/schedules.txt (the tuner's results on this pool's MI355X for the BASELINE presets, tools/tune_schedules.py) with
 * the library (NRC_SCHEDULE_CACHE=<file> adds the host's own, NRC_SCHEDULE_CACHE="" loads none); a C++ host calls en::LoadScheduleCache(path). */
const char* nrc_renderer_schedule_source(nrc_renderer_t* r);
const char* nrc_renderer_schedule_key(nrc_renderer_t* r);
int nrc_schedule_cache_load(const char* path, int* n_entries);
int nrc_schedule_cache_save(const char* path, int* n_entries);
int nrc_schedule_cache_clear(void);      /* forget every entry (renderers created afterwards start on the defaults and tune) */
int nrc_renderer_set_cost_order(nrc_renderer_t* r, int on);
size_t nrc_renderer_tile_order(nrc_renderer_t* r, uint32_t* host_out, size_t capacity);
/* Hot tiles (on by default): a pixel whose RNG state can run into DeltaTrack's cap of 128 collisions inside a tile the empty-space
 * mask rejects (see set_empty_skip) is ONE lane that walks for ~0.12 ms; started where the launch order has its (empty) tile --
 * at the very end -- it ends the launch that much later (one frame in four on the bench view: mean 0.213 -> 0.232 ms).  Each
 * gen_rays launch therefore also tests its pixels against the NEXT frame's random numbers (drawn one frame early, same sequence;
 * or the ones render_frames was given) -- one more hash per pixel, no launch of its own -- and the next launch starts up to 8 of
 * the tiles found first.  Scheduling only: every tile is traced exactly once either way.  hot_tiles copies the last frame's list
 * -- 8 entries (ty << 16 | tx) and their count -- and returns 1 when the list had been built by the previous frame's launch, 0
 * when by a kernel in front of gen_rays (first frame, pinned random numbers, another camera), -1 when the frame used none, -2 on
 * error. */
int nrc_renderer_set_hot_tiles(nrc_renderer_t* r, int on);
int nrc_renderer_hot_tiles(nrc_renderer_t* r, uint32_t* host_out9);
/* The NRC vertex images (buffers 2 and 3: nrcRayOrigin / nrcRayDir of gen_rays.comp:97-100) are read back only at the pixels of
 * the train grid (prep_train_rays.comp:113-118), so by default gen_rays stores them only there; on != 0 makes it store every
 * pixel that entered the volume, as the reference's images hold (tests, debugging).  vertex_image_bytes: what one frame stores. */
int nrc_renderer_set_full_vertex_images(nrc_renderer_t* r, int on);
size_t nrc_renderer_vertex_image_bytes(nrc_renderer_t* r);
/* per-stage timing events (the reference's eight timestamp queries, src/NrcHpmRenderer.cu:495-530) of train-ray generation,
 * training, inference and compositing: on by default; off = four timed event records fewer per frame, frame_time_ms / stage_stats
 * then report gen_rays only */
int nrc_renderer_set_stage_events(nrc_renderer_t* r, int on);
/* density look-ups executed by gen_rays (measurement: algorithmic bytes of the integrator).
 * Returns the count accumulated so far in *out (may be NULL), then enables/disables + zeroes the device counter. */
int nrc_renderer_count_fetches(nrc_renderer_t* r, int enable, unsigned long long* out);
/* train grid chosen by CalcTrainSubset (src/NrcHpmRenderer.cu:612-642): {TW, TH, xDist, yDist, ringSize} */
int nrc_renderer_train_grid(nrc_renderer_t* r, uint32_t out5[5]);

/* ---------------------------------------------------------------------------------------------------------
 * en::McHpmRenderer (include/engine/graphics/renderer/McHpmRenderer.hpp:10-31, src/McHpmRenderer.cpp) */
typedef struct nrc_mc_renderer nrc_mc_renderer_t;

/* McHpmRenderer(width, height, pathLength, blend, camera, scene) (src/McHpmRenderer.cpp:81-119) */
int nrc_mc_renderer_create(uint32_t width, uint32_t height, uint32_t path_length, int blend, const nrc_camera* camera,
                           const nrc_scene* scene, const nrc_tile* tile, void* stream, nrc_mc_renderer_t** out);
/* McHpmRenderer::Render(VkQueue) (src/McHpmRenderer.cpp:121-151) */
int nrc_mc_renderer_render(nrc_mc_renderer_t* r);
int nrc_mc_renderer_set_camera(nrc_mc_renderer_t* r, const nrc_camera* camera);
int nrc_mc_renderer_set_blend(nrc_mc_renderer_t* r, int blend);
int nrc_mc_renderer_set_empty_skip(nrc_mc_renderer_t* r, int on);   /* see nrc_renderer_set_empty_skip */
int nrc_mc_renderer_set_cost_order(nrc_mc_renderer_t* r, int on);   /* see nrc_renderer_set_cost_order */
int nrc_mc_renderer_is_blending(nrc_mc_renderer_t* r);          /* include/engine/graphics/renderer/McHpmRenderer.hpp */
int nrc_mc_renderer_set_scene_params(nrc_mc_renderer_t* r, const nrc_scene* scene);   /* see nrc_renderer_set_scene_params */
int nrc_mc_renderer_set_frame_random(nrc_mc_renderer_t* r, const float random4[4]);
const float* nrc_mc_renderer_framebuffer(nrc_mc_renderer_t* r);   /* RGBA32F, alpha = blended didScatter */
int nrc_mc_renderer_export_exr(nrc_mc_renderer_t* r, const char* path);
float nrc_mc_renderer_frame_time_ms(nrc_mc_renderer_t* r);
int nrc_mc_renderer_count_fetches(nrc_mc_renderer_t* r, int enable, unsigned long long* out);
int nrc_mc_renderer_destroy(nrc_mc_renderer_t* r);

/* ---------------------------------------------------------------------------------------------------------
 * Reference::Result metrics (include/engine/graphics/Reference.hpp:17-29, data/shader/ref/cmp1,norm,cmp2.comp):
 * device RGBA32F images of w*h pixels; result5 (host) = {mse, refMean, ownMean, ownVar, validPixelCount} */
int nrc_compare_images(const float* d_ref_rgba, const float* d_own_rgba, uint32_t w, uint32_t h, void* stream,
                       float result5[5]);

/* the same Result for a frame sharded over ranks (Reference::CompareNrc / CompareMc of a multi-GPU run, src/Reference.cpp:72-107):
 * every rank passes ITS pixels of the reference and of its image (n_local_pixels, any partition of the frame), the five sums are
 * formed locally in fp64 and summed over the ranks with two small all-reduces (4 + 1 doubles) through `comm`'s communicator; every
 * rank receives the whole frame's Result.  Equal to nrc_compare_images of the gathered frame up to the rounding of the fp64 sums. */
int nrc_compare_images_sharded(nrc_cache_t* comm, const float* d_ref_local_rgba, const float* d_own_local_rgba, uint32_t n_local_pixels,
                               void* stream, float result5[5]);

/* device-resident RGBA32F image [h][w] for nrc_compare_images: a copy of host_rgba (NULL: zeros) -- the reference image that
 * Reference::GenRefImages loads from reference/<scene>/0.exr into a VkImage (src/Reference.cpp:608-660) */
int nrc_image_create(uint32_t w, uint32_t h, const float* host_rgba, float** d_out);
int nrc_image_destroy(float* d_image);

/* THE environment switch of the library: NRC_DEBUG="name[=value],..." -- diagnostic and test switches only, none changes a result
 * (the list and what each does: csrc/nrc_common.hpp, debug_switch).
 * diagnostics (read when the library first allocates): NRC_DEBUG=poison_alloc fills every device allocation with 0xFF
 * bytes at creation; NRC_DEBUG=guard_alloc puts 4 KiB canaries around every allocation.  nrc_debug_check_guards returns -1 when the
 * guard mode is off, otherwise the number of allocations whose canaries were overwritten (0 = clean) and, in message, the first. */
int nrc_debug_check_guards(char* message, size_t message_bytes);

/* ---------------------------------------------------------------------------------------------------------
 * Test hooks (bit-parity of the math spec and RNG against the oracle; device pointers) */
int nrc_test_math(int fn, const float* d_a, const float* d_b, uint32_t n, float* d_out, float* d_out2, void* stream);
int nrc_test_rng(float u, float v, const float frame_random[4], uint32_t n, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
