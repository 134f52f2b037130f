// nrc_mlp.hpp -- the neural radiance cache arithmetic on gfx950: fused input encoding + fully fused MLP
// (inference), forward+loss+backward, split-K weight gradients, EMA{Adam}.  Replaces what the reference gets
// from tiny-cuda-nn v1.6 through tcnn::create_from_config / network->inference / trainer->training_step
// (src/NeuralRadianceCache.cu:16-39,142,153-154).
#pragma once
#include <vector>

#include "nrc_common.hpp"

namespace nrc {

struct MlpLayer {
    uint32_t out, in;
    uint32_t off;   // offset into the canonical fp32 parameter vector ([out][in] row-major)
};

// nrc/render.comp as the epilogue of the renderer's inference launch: the queries are in the renderer's tile-major order (query
// q_base + i = pixel (x, y) of 8x8 tile (tx, ty), see query_index in nrc_integrator.hip), so the 32 queries of an inference tile are
// four rows of eight pixels.  primary / info / framebuffer are [h][w] images.
struct CompositeArgs {
    const float* primary;      // float4: rgb, throughput
    const float* info;         // didScatter
    float* framebuffer;        // float4, blended in place
    uint32_t w, h, show_nrc, q_base;
    float blend_factor;
};

void mlp_set_wave_priority_raise(int on);      // nrc_common.hpp: NRC_RAISE_WAVE_PRIORITY's run-time switch, nrc_mlp.hip's kernels

class Mlp {
public:
    explicit Mlp(const nrc_config& cfg);
    ~Mlp();
    Mlp(const Mlp&) = delete;
    Mlp& operator=(const Mlp&) = delete;

    // network->inference: y = MLP_ema(encode(x)); in [n][5], out [n][3] (device, fp32)
    // skip_zero_queries (renderer only): 32-sample tiles whose queries are all exactly zero store 0 without running the network
    // live_list / live_count (renderer only, with skip_zero_queries): the indices of the queries that are not all zero, in any order
    // (k_gen_rays writes them); a model whose encoder gathers from a table walks the list instead of testing every query
    void infer(const float* d_in, float* d_out, uint32_t n, bool use_ema, hipStream_t s, bool skip_zero_queries = false,
               const CompositeArgs* composite = nullptr, const uint32_t* live_list = nullptr, const uint32_t* live_count = nullptr);
    // the fused 6x64 inference kernel can composite in its epilogue (nrc/render.comp); the generic kernels cannot
    bool can_composite() const { return fused_; }
    // EMA inference encodes the raw queries inside the MLP kernel (no feature buffer sized by the launch)
    bool encodes_in_kernel() const { return fused_ || enc80_generic_; }
    // forward (training weights) + loss + backward -> gradient vector (x loss_scale) and loss cell
    // widen_grid_grad (models with a trainable table): also write the table gradient into the fp32 gradient vector -- needed only
    // by readers of the vector (dense exchange, gradient hook, debug read-back); the optimizer reads the packed fp16 table itself
    // features_ready (generic models): the batch's fp16 features, encoded earlier by pre_encode -- backward() then launches no encoder
    void backward(const float* d_in, const float* d_target, uint32_t n, uint32_t n_norm, hipStream_t s, bool widen_grid_grad = true,
                  const void* features_ready = nullptr);
    // Encodings without trainable state (everything but HashGrid) do not depend on the weights: a caller that has the training inputs
    // before the training stream is free may encode them elsewhere (the renderer: on the train-ray stream, behind the rays).  Two
    // buffers of their own, by the caller's parity (not the one backward() encodes into: a step that encodes for itself may still be
    // reading that one); returns the features of the n samples at d_in.
    bool can_pre_encode() const { return !fused_ && !hash_; }
    const void* pre_encode(const float* d_in, uint32_t n, int parity, hipStream_t s);
    // the gradient vector was written from outside (exchange result, hook, nrc_cache_set_params): it is what the optimizer reads
    void grad_vector_is_source() { grid16_valid_ = false; }
    // EMA{Adam} step + re-pack of the fp16 MFMA fragment images.  loss_cell (host-mapped, may be null): where the step's
    // {loss, loss_seq} pair is published; returns true when the step's own launch did that (k_opt_pack), false when the caller
    // still has to (models with a trainable encoding, NRC_DEBUG=no_fused_opt)
    bool optimizer_step(hipStream_t s, uint32_t loss_seq = 0, unsigned long long* loss_cell = nullptr);
    void repack(hipStream_t s);

    uint32_t n_params() const { return n_params_; }
    uint32_t enc_dims() const { return enc_dims_; }
    float* grad_ptr() { return d_grad_; }
    float* loss_ptr() { return d_loss_; }
    float* buffer(int which);    // 0 w, 1 ema, 2 m, 3 v, 4 grad
    // tiny-cuda-nn's own parameter layout (params_full_precision of its Trainer): the same matrices in the same order with the output
    // matrix stored 16 x width (rows 3..15 feed the padded outputs nobody reads), then the table.  tcnn_param_count() = n_params() +
    // 13 * width.  to / from convert a host vector of buffer `which` (0..3; the dead rows of 4, the gradient, are zero); the dead
    // rows are kept on the host as they were initialised / last set, so that a dump read back is the dump
    uint32_t tcnn_param_count() const { return n_params_ + 13u * width_; }
    void to_tcnn_layout(int which, const float* own, float* tcnn) const;
    void from_tcnn_layout(int which, const float* tcnn, float* own);
    // sparse exchange of the HashGrid table gradient (nrc_mlp.hip, k_grid_pack): the gradient vector is
    // [n_mlp_params() matrix gradients][2 per table entry][2-word loss cell]
    bool has_grid() const { return hash_; }
    uint32_t n_mlp_params() const { return n_mlp_; }
    uint32_t grid_entries() const { return n_grid_entries_; }
    uint32_t grid_list_capacity(uint32_t n_batch) const;
    static size_t grid_list_words(uint32_t cap) { return 2 + 2 * (size_t)cap; }
    void grid_grad_pack(uint32_t* d_list, uint32_t cap, hipStream_t s);
    void grid_grad_apply(const uint32_t* d_lists, uint32_t n_lists, uint32_t cap, hipStream_t s);
    uint32_t step = 0;
    static constexpr float kLossScale = 128.0f;

private:
    int num_cus();
    int num_cus_ = 0;
    bool attr_infer_set_ = false, attr_infer4_set_ = false, attr_train_set_ = false, attr_train2_set_ = false;     // hipFuncSetAttribute done on this instance's device
    void ensure_train_workspace(uint32_t n);
    void ensure_features(uint32_t n, int slot);
    void launch_features(const float* d_in, uint32_t n, bool use_ema, int slot, hipStream_t s, bool skip_zero, const uint32_t* live_list = nullptr,
                         const uint32_t* live_count = nullptr);
#ifdef NRC_DIAG
    void infer_diagnostic(int abl, uint32_t blocks, size_t lds, hipStream_t s, const float* d_in, float* d_out, uint32_t n,
                          const void* image);
#endif

    nrc_config cfg_;
    uint32_t width_, depth_, enc_dims_, n_params_;
    uint32_t kw_ = 0;            // width the kernels run at: max(width_, 32)
    uint32_t loss_id_;
    bool sgd_ = false;           // nested optimizer: Adam (default) or SGD
    bool fused_ = false;         // north-star model (Frequency+OneBlob, 6x64): fully fused kernels; otherwise the generic path
    std::vector<MlpLayer> layers_;
    void* d_feat_[4] = {nullptr, nullptr, nullptr, nullptr};     // generic path: fp16 features [n][enc_dims]; [0] inference, [1] training (encoded by backward), [2] [3] training (pre_encode, by parity)
    uint32_t feat_n_[4] = {0, 0, 0, 0};
    int infer_set_ = 0;          // which of the double-buffered inference (EMA) image / table sets is current
    uint32_t n_mlp_ = 0;         // matrix parameters; (posID 0) the hash-grid table [entry][2] follows them in every vector
    bool hash_ = false;
    bool xcd8_ = true;           // the device has eight XCDs: the level-per-XCD launch mappings of the HashGrid kernels apply
    uint32_t hg_off_[17] = {0};  // per-level entry offsets
    uint32_t n_grid_entries_ = 0;
    void *d_t16_train_ = nullptr, *d_t16_ema_[2] = {nullptr, nullptr};   // half2-per-entry gather copies of the table
    void* d_denc_ = nullptr;     // fp16 [n][32] dL/d(grid features)
    void* d_grad16_ = nullptr;   // half2 per table entry: target of the packed gradient atomics
    // bin lists of the table gradient (k_grid_scatter / k_grid_gather): (entry, value) pairs per bin of 4 096 entries, pair counters, per-bin
    // {first entry, list offset, capacity}; d_grid_fix_: the table's fixed-point shadow (two int64 per entry) for what bypasses the lists
    void *d_grid_lists_ = nullptr, *d_grid_counters_ = nullptr, *d_grid_bin_entry0_ = nullptr, *d_grid_fix_ = nullptr;
    uint32_t grid_bin_first_[16] = {}, grid_bin_count_[16] = {}, grid_bin_cap_[16] = {}, grid_list0_[16] = {}, grid_bins_total_ = 0;
    bool attr_gather_set_ = false;

    float *d_w_ = nullptr, *d_ema_ = nullptr, *d_m_ = nullptr, *d_v_ = nullptr, *d_grad_ = nullptr, *d_loss_ = nullptr;
    std::vector<float> tcnn_dead_rows_[4];      // rows 3..15 of tiny-cuda-nn's 16 x width output matrix, per buffer 0..3
    // fp16 MFMA A-operand fragment images ([frag][lane][8 halfs]): inference (EMA), training forward, training dgrad (W^T)
    void *d_pk_infer_[2] = {nullptr, nullptr}, *d_pk_fwd_ = nullptr, *d_pk_bwd_ = nullptr;
    int32_t *d_src_fwd_ = nullptr, *d_src_bwd_ = nullptr;   // packed slot -> canonical index (-1 = zero)
    int32_t* d_dst_ = nullptr;           // [3][n_mlp_] parameter -> slot in the forward / EMA inference / backward image (k_opt_pack)
    bool fused_opt_ = false;
    bool grad16_clean_ = false;          // d_grad16_ is all zero (the optimizer cleared what the last backward() touched)
    bool grid16_valid_ = false;          // the packed fp16 table gradient of the last backward() is what the optimizer should read
    int32_t* d_src_inf_ = nullptr;       // the same for the EMA inference image (= d_src_fwd_ unless enc80_generic_)
    bool enc80_generic_ = false;         // generic model whose EMA inference encodes Frequency(12)+OneBlob(4) inside k_infer_gen
    uint32_t n_frag_fwd_ = 0, n_frag_bwd_ = 0;

    // training workspace
    uint32_t ws_n_ = 0;
    void* d_acts_ = nullptr;     // fp16 [enc + depth*width][n]   (transposed: neuron-major)
    void* d_deltas_ = nullptr;   // fp16 [depth*width + 32][n]
    float* d_slabs_ = nullptr;   // fp32 [n_chunks][n_params]
    float* d_loss_part_ = nullptr;
    void* d_tiles_ = nullptr;    // WgradTile[] (output tiles of the weight-gradient GEMMs)
    int n_wgrad_tiles_ = 0;
    void* d_tasks_ = nullptr;    // WgradTask[] (row-block tasks of k_wgrad2)
    int n_wgrad_tasks_ = 0;
    bool wgrad_old_ = false;     // round 3's k_wgrad (up to 64 neurons; NRC_DEBUG=wgrad_old=0|1)
    void build_wgrad_tasks();
    uint32_t wgrad_chunk(uint32_t n);
};

}  // namespace nrc
