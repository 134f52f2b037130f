// nrc_api.hip -- host side of libnrc_hpm: the reference's NeuralRadianceCache / NrcHpmRenderer / McHpmRenderer
// behaviour over one HIP stream, exported through the C ABI of include/nrc_hpm.h.
//
//   nrc::Cache       <- en::NeuralRadianceCache   (src/NeuralRadianceCache.cu)
//   nrc::Renderer    <- en::NrcHpmRenderer        (src/NrcHpmRenderer.cu:212-353,561-642,823-881,908-1061)
//   nrc::McRenderer  <- en::McHpmRenderer         (src/McHpmRenderer.cpp:81-151,432-449)
// One instance per GPU and stream; no hidden globals (SURVEY.md 8b "Threading").
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <random>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "nrc_integrator.hpp"
#include "nrc_mlp.hpp"

namespace nrc {

static thread_local std::string g_last_error;

// ---------------------------------------------------------------------------------------------------- device allocations
namespace {
constexpr size_t kGuardBytes = 4096;
struct AllocInfo {
    void* base;
    size_t bytes;
    std::string what;
};
struct AllocRegistry {
    std::mutex mu;
    std::unordered_map<void*, AllocInfo> live;      // user pointer -> allocation (guard mode only)
    bool poison = getenv("NRC_POISON_ALLOC") != nullptr && getenv("NRC_POISON_ALLOC")[0] != '0';
    bool guard = getenv("NRC_GUARD_ALLOC") != nullptr && getenv("NRC_GUARD_ALLOC")[0] != '0';
    unsigned long long violations = 0;
    std::string first_violation;
};
AllocRegistry& alloc_registry()
{
    static AllocRegistry r;
    return r;
}
// canary bytes of one allocation that no longer hold 0xA5 (host copy of both guards)
size_t guard_damage(const AllocInfo& a, std::string* where)
{
    std::vector<unsigned char> g(2 * kGuardBytes);
    if (hipMemcpy(g.data(), a.base, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (hipMemcpy(g.data() + kGuardBytes, (char*)a.base + kGuardBytes + a.bytes, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    size_t bad = 0;
    long first = -1;
    for (size_t i = 0; i < g.size(); i++)
        if (g[i] != 0xA5) { bad++; if (first < 0) first = (long)i; }
    if (bad && where)
        *where = a.what + " (" + std::to_string(a.bytes) + " B): " + std::to_string(bad) + " canary bytes overwritten, first at offset " +
                 (first < (long)kGuardBytes ? std::to_string(first - (long)kGuardBytes) : "+" + std::to_string(a.bytes + (size_t)first - kGuardBytes)) ;
    return bad;
}
}  // namespace

void dev_alloc(void** p, size_t bytes, const char* what)
{
    AllocRegistry& r = alloc_registry();
    if (!r.guard) {
        NRC_HIP(hipMalloc(p, bytes));
        if (r.poison && bytes) NRC_HIP(hipMemset(*p, 0xFF, bytes));
        return;
    }
    const size_t padded = (bytes + 255) & ~(size_t)255;      // keep the rear guard (and the user pointer) 256-byte aligned
    void* base = nullptr;
    NRC_HIP(hipMalloc(&base, padded + 2 * kGuardBytes));
    NRC_HIP(hipMemset(base, 0xA5, padded + 2 * kGuardBytes));
    *p = (char*)base + kGuardBytes;
    if (bytes) NRC_HIP(hipMemset(*p, r.poison ? 0xFF : 0x00, bytes));      // bytes..padded stay canary
    std::lock_guard<std::mutex> lock(r.mu);
    r.live[*p] = AllocInfo{base, bytes, what ? what : ""};
}

void dev_free(void* p)
{
    if (!p) return;
    AllocRegistry& r = alloc_registry();
    if (!r.guard) {
        (void)hipFree(p);
        return;
    }
    AllocInfo a{nullptr, 0, ""};
    {
        std::lock_guard<std::mutex> lock(r.mu);
        auto it = r.live.find(p);
        if (it == r.live.end()) { (void)hipFree(p); return; }
        a = it->second;
        r.live.erase(it);
    }
    (void)hipDeviceSynchronize();
    AllocInfo chk = a;
    chk.bytes = (a.bytes + 255) & ~(size_t)255;
    // the slack between bytes and its 256-byte round-up is canary too
    std::vector<unsigned char> slack(chk.bytes - a.bytes);
    size_t bad = 0;
    if (!slack.empty() && hipMemcpy(slack.data(), (char*)a.base + kGuardBytes + a.bytes, slack.size(), hipMemcpyDeviceToHost) == hipSuccess)
        for (unsigned char c : slack) bad += c != 0xA5;
    std::string where;
    bad += guard_damage(chk, &where);
    if (bad) {
        std::lock_guard<std::mutex> lock(r.mu);
        r.violations++;
        if (r.first_violation.empty()) r.first_violation = where.empty() ? a.what + ": slack bytes behind the allocation overwritten" : where;
        std::fprintf(stderr, "NRC_GUARD_ALLOC: %s\n", r.first_violation.c_str());
    }
    (void)hipFree(a.base);
}

// RCCL is bound at run time (dlopen by soname: inside a PyTorch process this resolves to the librccl.so.1 torch already
// loaded, so both share one library instance); the product has no link-time dependency on it.
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    static Rccl& get()
    {
        static Rccl r = [] {
            Rccl x;
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (x.handle) break;
            }
            if (!x.handle) fail("cannot load librccl.so.1 (multi-GPU gradient exchange needs RCCL)");
            x.get_unique_id = (decltype(x.get_unique_id))dlsym(x.handle, "ncclGetUniqueId");
            x.comm_init_rank = (decltype(x.comm_init_rank))dlsym(x.handle, "ncclCommInitRank");
            x.comm_destroy = (decltype(x.comm_destroy))dlsym(x.handle, "ncclCommDestroy");
            x.all_reduce = (decltype(x.all_reduce))dlsym(x.handle, "ncclAllReduce");
            x.all_gather = (decltype(x.all_gather))dlsym(x.handle, "ncclAllGather");
            x.group_start = (decltype(x.group_start))dlsym(x.handle, "ncclGroupStart");
            x.group_end = (decltype(x.group_end))dlsym(x.handle, "ncclGroupEnd");
            x.error_string = (decltype(x.error_string))dlsym(x.handle, "ncclGetErrorString");
            x.comm_count = (decltype(x.comm_count))dlsym(x.handle, "ncclCommCount");
            x.comm_user_rank = (decltype(x.comm_user_rank))dlsym(x.handle, "ncclCommUserRank");
            if (!x.get_unique_id || !x.comm_init_rank || !x.comm_destroy || !x.all_reduce) fail("librccl lacks the expected entry points");
            return x;
        }();
        return r;
    }
    void check(ncclResult_t r, const char* what)
    {
        if (r != ncclSuccess) fail(std::string(what) + " failed: " + (error_string ? error_string(r) : "rccl error"));
    }
};

// one thread: the training step's loss and its sequence number as ONE 8-byte store into host-mapped pinned memory
struct LossCell {
    float loss;
    uint32_t seq;
};
__global__ void k_publish_loss(const float* __restrict__ d_loss, uint32_t seq, unsigned long long* __restrict__ host_cell)
{
    LossCell c{d_loss[0], seq};
    unsigned long long bits;
    __builtin_memcpy(&bits, &c, 8);
    __hip_atomic_store(host_cell, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------------- Cache

class Cache {
    // validated before any size is derived from it: 2 << (log2 - 1) with log2 = 0 would shift by 0xFFFFFFFF
    static const nrc_config& checked(const nrc_config& cfg)
    {
        if (cfg.log2_infer_batch_size == 0 || cfg.log2_infer_batch_size > 30 || cfg.log2_train_batch_size < 5 ||
            cfg.log2_train_batch_size > 30)
            fail("log2 batch sizes out of range");
        if (cfg.train_batch_count == 0 || ((uint64_t)cfg.train_batch_count << cfg.log2_train_batch_size) > (1ull << 30))
            fail("trainBatchCount x trainBatchSize must be in 1..2^30");
        if (!(cfg.train_ring_buf_size >= 0.0f) || cfg.train_ring_buf_size > 1024.0f) fail("trainRingBufSize must be in 0..1024");
        return cfg;
    }

public:
    explicit Cache(const nrc_config& cfg)
        : cfg_(checked(cfg)),
          infer_batch_size_(2u << (cfg.log2_infer_batch_size - 1)),      // src/NeuralRadianceCache.cu:12-14
          train_batch_size_(2u << (cfg.log2_train_batch_size - 1)),
          train_batch_count_(cfg.train_batch_count),
          mlp_(new Mlp(cfg))
    {
        NRC_HIP(hipEventCreateWithFlags(&ev_loss_, hipEventDisableTiming));
        NRC_HIP(hipHostMalloc((void**)&h_loss_, sizeof(unsigned long long), hipHostMallocMapped));
        *h_loss_ = 0ull;
        NRC_HIP(hipHostGetDevicePointer((void**)&d_loss_cell_, h_loss_, 0));
        NRC_HIP(hipEventCreateWithFlags(&ev_owner_infer_, hipEventDisableTiming));
        NRC_HIP(hipEventCreateWithFlags(&ev_owner_train_, hipEventDisableTiming));
    }

    void init(uint32_t infer_count, float* d_in, float* d_out, float* d_tin, float* d_ttarget, hipStream_t s,
              hipEvent_t ev_start = nullptr, hipEvent_t ev_finished = nullptr)
    {
        bind(infer_count, d_in, d_out, d_tin, d_ttarget);
        stream_ = s;
        // the reference's cudaStartSemaphore / cudaFinishedSemaphore pair (src/NeuralRadianceCache.cu:158-177) as HIP events
        ev_start_ = ev_start;
        ev_finished_ = ev_finished;
    }

    // (re)bind the caller-owned I/O buffers without touching the stream of the stand-alone entry points: the renderer does
    // this every frame (its buffer sets rotate, and several renderers may share one cache) and names its streams explicitly
    void bind(uint32_t infer_count, float* d_in, float* d_out, float* d_tin, float* d_ttarget)
    {
        if (infer_count % 16 != 0) fail("NRC requires inferCount to be a multiple of 16");   // :52
        infer_count_ = infer_count;
        d_infer_in_ = d_in; d_infer_out_ = d_out; d_train_in_ = d_tin; d_train_target_ = d_ttarget;
        // batch slicing, :67-92
        infer_batches_.clear();
        const uint32_t full = infer_count / infer_batch_size_;
        for (uint32_t i = 0; i < full; i++) infer_batches_.push_back({i * infer_batch_size_, infer_batch_size_});
        const uint32_t last = infer_count - full * infer_batch_size_;
        if (last > 0) infer_batches_.push_back({full * infer_batch_size_, last});
        initialised_ = true;
    }

    void infer_and_train(const uint32_t* filter, bool train)       // :97-103
    {
        acquire(this, stream_, stream_);
        if (ev_start_) NRC_HIP(hipStreamWaitEvent(stream_, ev_start_, 0));           // AwaitCudaStartSemaphore, :158-167
        infer_all(filter, stream_);
        if (train) train_all(stream_, nullptr);
        if (ev_finished_) NRC_HIP(hipEventRecord(ev_finished_, stream_));            // SignalCudaFinishedSemaphore, :169-177
    }

    // Several users may drive one cache -- a training renderer and an evaluation renderer (Reference::CompareNrc,
    // src/Reference.cpp:71-107), or a renderer and the stand-alone entry points -- each with streams of its own.  A user's own
    // frames are ordered by its own events; when the user CHANGES, the new user's inference and training streams wait (on the
    // device) for everything the previous user had enqueued on its two cache streams, so weight images, feature buffers and
    // the gradient vector are never shared between two users in flight.  Costs nothing while the user stays the same.
    void acquire(const void* who, hipStream_t infer_stream, hipStream_t train_stream)
    {
        if (owner_ != nullptr && owner_ != who) {
            NRC_HIP(hipEventRecord(ev_owner_infer_, owner_infer_stream_));
            NRC_HIP(hipEventRecord(ev_owner_train_, owner_train_stream_));
            for (hipStream_t s : {infer_stream, train_stream}) {
                NRC_HIP(hipStreamWaitEvent(s, ev_owner_infer_, 0));
                NRC_HIP(hipStreamWaitEvent(s, ev_owner_train_, 0));
                if (train_stream == infer_stream) break;
            }
        }
        owner_ = who;
        owner_infer_stream_ = infer_stream;
        owner_train_stream_ = train_stream;
    }
    // a user whose streams are about to disappear (renderer destruction; it has synchronised them)
    void forget(const void* who)
    {
        if (owner_ == who) owner_ = nullptr;
    }

    // skip_zero: device-side replacement of the reference's per-batch host filter (src/NrcHpmRenderer.cu:332-337,
    // prep_infer_rays.comp:43-45): the renderer zero-fills the query of every pixel that did not scatter
    // composite: the renderer's compositing pass as the epilogue of the inference launches (fused 6x64 model only, can_composite)
    void infer_all(const uint32_t* filter, hipStream_t s, bool skip_zero = false, const CompositeArgs* composite = nullptr)          // Inference, :134-145
    {
        if (!initialised_) throw std::logic_error("SkyRenderer ERROR: InferAndTrain before Init");
        for (size_t i = 0; i < infer_batches_.size(); i++) {
            if (filter != nullptr && filter[i] == 0) continue;
            const auto& b = infer_batches_[i];
            CompositeArgs ca;
            if (composite) { ca = *composite; ca.q_base = b.first; }
            mlp_->infer(d_infer_in_ + (size_t)b.first * 5, d_infer_out_ + (size_t)b.first * 3, b.second, true, s, skip_zero, composite ? &ca : nullptr);
        }
    }
    bool can_composite() const { return mlp_->can_composite(); }

    // Train, :147-156, on stream `st` (backward, gradient hook, optimizer).  When `st` is not the stream inference runs on,
    // the optimizer additionally waits for ev_infer_done -- the inference pass BEFORE the latest one: the fp16 inference image
    // is double-buffered (Mlp::repack) and the optimizer overwrites the set that pass read.
    void train_all(hipStream_t st, hipEvent_t ev_infer_prev, hipEvent_t ev_infer_cur = nullptr)
    {
        if (!initialised_) throw std::logic_error("SkyRenderer ERROR: InferAndTrain before Init");
        for (uint32_t b = 0; b < train_batch_count_; b++) {
            const size_t o = (size_t)b * train_batch_size_;
            const bool vector_read = hook_ != nullptr || (comm_ != nullptr && !sparse_grid_);      // who reads the fp32 table gradient
            mlp_->backward(d_train_in_ + o * 5, d_train_target_ + o * 3, train_batch_size_,
                           train_batch_size_ * loss_norm_factor_, st, vector_read);
            // the one exchange step of the sharded path: sum the fp32 gradient vector + loss cell over the ranks (103 KB,
            // latency-bound) on the training stream; every rank then applies the identical optimizer step
            // A HashGrid model's vector is 57 MB of which a batch touches a few per cent: its table part travels as all-gathered
            // (entry, value) lists, added in rank order (Mlp::grid_grad_pack / grid_grad_apply); NRC_DENSE_GRID_EXCHANGE=1 keeps
            // the dense all-reduce.
            if (comm_ && sparse_grid_) exchange_sparse(st);
            else if (comm_) Rccl::get().check(Rccl::get().all_reduce(mlp_->grad_ptr(), mlp_->grad_ptr(), (size_t)mlp_->n_params() + 2, ncclFloat,
                                                               ncclSum, comm_, st), "ncclAllReduce");
            if (hook_) hook_(hook_user_, mlp_->grad_ptr(), mlp_->n_params(), mlp_->loss_ptr(), (void*)st);
            if (vector_read) mlp_->grad_vector_is_source();
            // the two inference weight sets alternate with every optimizer step: step 0 overwrites the set the PREVIOUS
            // inference pass read, step 1 the set the CURRENT pass is reading, later steps only sets no pass reads any more
            if (b == 0 && ev_infer_prev) NRC_HIP(hipStreamWaitEvent(st, ev_infer_prev, 0));
            if (b == 1 && ev_infer_cur) NRC_HIP(hipStreamWaitEvent(st, ev_infer_cur, 0));
            // the step's loss is published by the optimizer's own launch where it can be (Mlp::optimizer_step)
            loss_pushed_++;
            if (!mlp_->optimizer_step(st, loss_pushed_, d_loss_cell_)) {
                hipLaunchKernelGGL(k_publish_loss, dim3(1), dim3(1), 0, st, (const float*)mlp_->loss_ptr(), loss_pushed_, d_loss_cell_);
                NRC_HIP(hipGetLastError());
            }
            NRC_HIP(hipEventRecord(ev_loss_, st));
        }
    }

    void ensure_grid_lists(uint32_t n_lists)
    {
        const uint32_t cap = mlp_->grid_list_capacity(train_batch_size_);
        if (cap == grid_cap_ && n_lists <= grid_lists_) return;
        NRC_HIP(hipDeviceSynchronize());
        if (d_grid_send_) dev_free(d_grid_send_);
        if (d_grid_recv_) dev_free(d_grid_recv_);
        grid_cap_ = cap;
        grid_lists_ = n_lists;
        dev_alloc(&d_grid_send_, Mlp::grid_list_words(cap) * 4, "d_grid_send_");
        dev_alloc(&d_grid_recv_, Mlp::grid_list_words(cap) * 4 * n_lists, "d_grid_recv_");
    }
    void exchange_sparse(hipStream_t st)
    {
        Rccl& r = Rccl::get();
        if (!r.all_gather || !r.group_start || !r.group_end) fail("librccl lacks ncclAllGather / ncclGroupStart / ncclGroupEnd");
        ensure_grid_lists((uint32_t)comm_world_);
        mlp_->grid_grad_pack(d_grid_send_, grid_cap_, st);
        r.check(r.group_start(), "ncclGroupStart");
        r.check(r.all_reduce(mlp_->grad_ptr(), mlp_->grad_ptr(), mlp_->n_mlp_params(), ncclFloat, ncclSum, comm_, st), "ncclAllReduce");
        r.check(r.all_reduce(mlp_->loss_ptr(), mlp_->loss_ptr(), 2, ncclFloat, ncclSum, comm_, st), "ncclAllReduce");
        r.check(r.all_gather(d_grid_send_, d_grid_recv_, Mlp::grid_list_words(grid_cap_), ncclUint32, comm_, st), "ncclAllGather");
        r.check(r.group_end(), "ncclGroupEnd");
        mlp_->grid_grad_apply(d_grid_recv_, (uint32_t)comm_world_, grid_cap_, st);
    }
    // test surface of the sparse exchange (tests/test_gpu_mlp.py): the list of the last backward() on the host; lists applied in order
    size_t grid_pack_host(uint32_t* host_list, size_t cap_words)
    {
        ensure_grid_lists(1);
        if (cap_words < Mlp::grid_list_words(grid_cap_)) fail("grid_pack_host: list buffer too small");
        mlp_->grid_grad_pack(d_grid_send_, grid_cap_, stream_);
        NRC_HIP(hipMemcpyAsync(host_list, d_grid_send_, Mlp::grid_list_words(grid_cap_) * 4, hipMemcpyDeviceToHost, stream_));
        NRC_HIP(hipStreamSynchronize(stream_));
        if (host_list[0] > grid_cap_) fail("grid_pack_host: the batch touched more entries than a train batch can (backward on more than trainBatchSize samples?)");
        return grid_cap_;
    }
    void grid_apply_host(const uint32_t* host_lists, uint32_t n_lists)
    {
        ensure_grid_lists(n_lists);
        NRC_HIP(hipMemcpyAsync(d_grid_recv_, host_lists, Mlp::grid_list_words(grid_cap_) * 4 * n_lists, hipMemcpyHostToDevice, stream_));
        mlp_->grid_grad_apply(d_grid_recv_, n_lists, grid_cap_, stream_);
        NRC_HIP(hipStreamSynchronize(stream_));
    }
    uint32_t grid_list_capacity() const { return mlp_->has_grid() ? mlp_->grid_list_capacity(train_batch_size_) : 0u; }
    bool sparse_grid_exchange() const { return comm_ != nullptr && sparse_grid_; }

    // m_Loss = trainer->loss(*ctx) after every training step (src/NeuralRadianceCache.cu:154) is a device->host sync in the
    // reference.  Here a one-thread kernel behind every step stores {loss, step number} with one 8-byte store into host-mapped
    // pinned memory (no event polling: the runtime reports a recorded event complete only with the batch it was submitted in):
    //   get_loss(true)   GetLoss(): waits for the last step that was enqueued (only for that step, not for the device) -- the
    //                    reference's value, m_Loss of the step InferAndTrain just ran (src/NeuralRadianceCache.cu:154);
    //   get_loss(false)  GetLossAsync(): the loss of the most recent step that has COMPLETED and that step's number -- a plain
    //                    read of the cell; never blocks, never drains the frame pipeline (a per-frame poll like src/main.cu:303,376
    //                    that must not stall the renderer; lags the enqueued work by the pipeline depth, at most four frames).
    void push_loss(hipStream_t st)
    {
        loss_pushed_++;
        hipLaunchKernelGGL(k_publish_loss, dim3(1), dim3(1), 0, st, (const float*)mlp_->loss_ptr(), loss_pushed_, d_loss_cell_);
        NRC_HIP(hipGetLastError());
        NRC_HIP(hipEventRecord(ev_loss_, st));
    }
    float get_loss(bool wait, uint32_t* seq = nullptr)
    {
        if (seq) *seq = 0;
        if (loss_pushed_ == 0) return 0.0f;
        if (wait) NRC_HIP(hipEventSynchronize(ev_loss_));
        LossCell c;
        const unsigned long long bits = __atomic_load_n(h_loss_, __ATOMIC_ACQUIRE);
        std::memcpy(&c, &bits, 8);
        if (seq) *seq = c.seq;      // the training step the value belongs to (0: none has completed yet)
        return c.loss;
    }
    uint32_t loss_steps_enqueued() const { return loss_pushed_; }
    bool grad_ptr_exposed() const { return grad_ptr_exposed_; }
    void expose_grad_ptr() { grad_ptr_exposed_ = true; }

    Mlp& mlp() { return *mlp_; }
    hipStream_t stream() const { return stream_; }
    void set_stream(hipStream_t s) { stream_ = s; }
    size_t infer_batch_count() const { return infer_batches_.size(); }
    size_t train_batch_count() const { return train_batch_count_; }
    uint32_t infer_batch_size() const { return infer_batch_size_; }
    uint32_t train_batch_size() const { return train_batch_size_; }
    void set_hook(nrc_grad_hook h, void* u) { hook_ = h; hook_user_ = u; }
    void comm_init(const void* unique_id, int rank, int world)
    {
        if (world < 1 || rank < 0 || rank >= world) fail("bad rank / world size");
        if (comm_) fail("communicator already initialised");
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof(id));
        Rccl::get().check(Rccl::get().comm_init_rank(&comm_, world, id, rank), "ncclCommInitRank");
        loss_norm_factor_ = (uint32_t)world;
        comm_world_ = world;
        // lists pay when the padded all-gather (world x capacity entries into every rank) moves less than the ring all-reduce of
        // the dense table gradient (2 x entries through every rank): a rank's share of a sharded frame's train batch, not eight
        // full 16 384-ray batches.  NRC_DENSE_GRID_EXCHANGE=1 / NRC_SPARSE_GRID_EXCHANGE=1 force either.
        sparse_grid_ = mlp_->has_grid() &&
                       (unsigned long long)world * mlp_->grid_list_capacity(train_batch_size_) < 2ull * mlp_->grid_entries();
        if (getenv("NRC_SPARSE_GRID_EXCHANGE")) sparse_grid_ = mlp_->has_grid();
        if (getenv("NRC_DENSE_GRID_EXCHANGE")) sparse_grid_ = false;
    }
    // what the communicator itself reports (bench.py prints it: proof that the native exchange path is the one in use)
    void comm_info(int* rank, int* world)
    {
        *rank = 0; *world = 0;
        if (!comm_) return;
        Rccl& r = Rccl::get();
        if (!r.comm_count || !r.comm_user_rank) fail("librccl lacks ncclCommCount / ncclCommUserRank");
        r.check(r.comm_count(comm_, world), "ncclCommCount");
        r.check(r.comm_user_rank(comm_, rank), "ncclCommUserRank");
    }
    // measurement (bench.py, N > 1): average duration of the training step's all-reduce -- the gradient vector + loss cell, zeroed
    // first -- issued `reps` times back to back on the stand-alone stream; a collective call (every rank, same reps)
    float time_exchange(uint32_t reps)
    {
        if (!comm_ || reps == 0) return 0.0f;
        NRC_HIP(hipDeviceSynchronize());
        Rccl& r = Rccl::get();
        const size_t n = (sparse_grid_ ? (size_t)mlp_->n_mlp_params() : (size_t)mlp_->n_params()) + (sparse_grid_ ? 0 : 2);
        NRC_HIP(hipMemsetAsync(mlp_->grad_ptr(), 0, ((size_t)mlp_->n_params() + 2) * sizeof(float), stream_));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        NRC_HIP(hipEventCreate(&e0));
        NRC_HIP(hipEventCreate(&e1));
        for (uint32_t i = 0; i < 3; i++) r.check(r.all_reduce(mlp_->grad_ptr(), mlp_->grad_ptr(), n, ncclFloat, ncclSum, comm_, stream_), "ncclAllReduce");
        NRC_HIP(hipEventRecord(e0, stream_));
        for (uint32_t i = 0; i < reps; i++) r.check(r.all_reduce(mlp_->grad_ptr(), mlp_->grad_ptr(), n, ncclFloat, ncclSum, comm_, stream_), "ncclAllReduce");
        NRC_HIP(hipEventRecord(e1, stream_));
        NRC_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        NRC_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        mlp_->grad_vector_is_source();
        return ms * 1000.0f / (float)reps;
    }
    ~Cache()
    {
        if (comm_) (void)Rccl::get().comm_destroy(comm_);
        if (ev_loss_) (void)hipEventDestroy(ev_loss_);
        if (ev_owner_infer_) (void)hipEventDestroy(ev_owner_infer_);
        if (ev_owner_train_) (void)hipEventDestroy(ev_owner_train_);
        if (h_loss_) (void)hipHostFree(h_loss_);
    }
    void set_loss_norm_factor(uint32_t f) { loss_norm_factor_ = f ? f : 1; }
    const nrc_config& config() const { return cfg_; }

private:
    nrc_config cfg_;
    const uint32_t infer_batch_size_, train_batch_size_, train_batch_count_;
    std::unique_ptr<Mlp> mlp_;
    uint32_t infer_count_ = 0;
    float *d_infer_in_ = nullptr, *d_infer_out_ = nullptr, *d_train_in_ = nullptr, *d_train_target_ = nullptr;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_start_ = nullptr, ev_finished_ = nullptr;      // caller-owned (Init overload), may be null
    std::vector<std::pair<uint32_t, uint32_t>> infer_batches_;
    bool initialised_ = false;
    nrc_grad_hook hook_ = nullptr;
    void* hook_user_ = nullptr;
    ncclComm_t comm_ = nullptr;
    int comm_world_ = 0;
    bool sparse_grid_ = false;              // HashGrid table gradient exchanged as (entry, value) lists
    uint32_t *d_grid_send_ = nullptr, *d_grid_recv_ = nullptr;
    uint32_t grid_cap_ = 0, grid_lists_ = 0;
    uint32_t loss_norm_factor_ = 1;
    unsigned long long* h_loss_ = nullptr;    // pinned, host-mapped: LossCell written by k_publish_loss
    unsigned long long* d_loss_cell_ = nullptr;
    hipEvent_t ev_loss_ = nullptr;            // behind the newest publish (blocking GetLoss)
    uint32_t loss_pushed_ = 0;
    bool grad_ptr_exposed_ = false;           // nrc_cache_grad_ptr has handed the gradient vector to the caller
    const void* owner_ = nullptr;
    hipStream_t owner_infer_stream_ = nullptr, owner_train_stream_ = nullptr;
    hipEvent_t ev_owner_infer_ = nullptr, ev_owner_train_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------- scene upload
struct SceneDev {
    DevScene d{};
    void* d_density = nullptr;
    void* d_env = nullptr;
    void* d_boxes = nullptr;      // occupancy boxes for the empty-space tile mask (launch_tile_mask)
    uint32_t n_boxes = 0;
    void* d_occ_bits = nullptr;   // exact occupancy bits (DevScene::occ_bits)

    // exact occupancy for the kernels' LDS copy: the smallest cubic cell (>= 8 voxels) whose bit grid fits kOccMaxWords words
    void build_occupancy_bits(const nrc_scene& s)
    {
        d.occ_bits = nullptr;
        d.occ_shift = d.occ_gx = d.occ_gy = d.occ_words = 0;
        if (getenv("NRC_NO_OCCUPANCY")) return;
        uint32_t sh = 3;
        auto cells = [&](uint32_t n) { return (n + (1u << sh) - 1u) >> sh; };
        while ((uint64_t)cells(s.nx) * cells(s.ny) * cells(s.nz) > (uint64_t)kOccMaxWords * 32u) sh++;
        const uint32_t gx = cells(s.nx), gy = cells(s.ny), gz = cells(s.nz);
        // (a multiple of four words: k_gen_rays copies the table in 16-byte pieces)
        std::vector<uint32_t> bits((((size_t)gx * gy * gz + 31) / 32 + 3) & ~(size_t)3, 0u);
        for (uint32_t z = 0; z < s.nz; z++)
            for (uint32_t y = 0; y < s.ny; y++) {
                const uint8_t* row = s.density + ((size_t)z * s.ny + y) * s.nx;
                const size_t base = ((size_t)(z >> sh) * gy + (y >> sh)) * gx;
                for (uint32_t x = 0; x < s.nx; x++)
                    if (row[x]) {
                        const size_t cidx = base + (x >> sh);
                        bits[cidx >> 5] |= 1u << (cidx & 31);
                    }
            }
        dev_alloc(&d_occ_bits, bits.size() * 4, "d_occ_bits");
        NRC_HIP(hipMemcpy(d_occ_bits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
        d.occ_bits = (const uint32_t*)d_occ_bits;
        d.occ_shift = sh; d.occ_gx = gx; d.occ_gy = gy; d.occ_words = (uint32_t)bits.size();
    }

    // Occupancy of the volume in cells of 8^3 voxels: a cell counts as occupied when a non-zero voxel lies in it or within one
    // voxel of it (the margin that makes the tile mask conservative against every rounding in the ray / sample arithmetic: a
    // sample position is computed to ~1e-5 of a voxel).  Runs of occupied cells along x become world-space boxes.
    void build_occupancy(const nrc_scene& s)
    {
        const uint32_t nx = s.nx, ny = s.ny, nz = s.nz;
        const uint32_t gx = (nx + 7) / 8, gy = (ny + 7) / 8, gz = (nz + 7) / 8;
        std::vector<uint8_t> occ((size_t)gx * gy * gz, 0);
        for (uint32_t z = 0; z < nz; z++)
            for (uint32_t y = 0; y < ny; y++) {
                const uint8_t* row = s.density + ((size_t)z * ny + y) * nx;
                const uint32_t cz0 = (z ? z - 1 : 0) >> 3, cz1 = std::min(z + 1, nz - 1) >> 3;
                const uint32_t cy0 = (y ? y - 1 : 0) >> 3, cy1 = std::min(y + 1, ny - 1) >> 3;
                for (uint32_t x = 0; x < nx; x++) {
                    if (row[x] == 0) continue;
                    const uint32_t cx0 = (x ? x - 1 : 0) >> 3, cx1 = std::min(x + 1, nx - 1) >> 3;
                    for (uint32_t cz = cz0; cz <= cz1; cz++)
                        for (uint32_t cy = cy0; cy <= cy1; cy++)
                            for (uint32_t cx = cx0; cx <= cx1; cx++) occ[((size_t)cz * gy + cy) * gx + cx] = 1;
                }
            }
        std::vector<float> boxes;
        const double vs[3] = {(double)d.size[0] / nx, (double)d.size[1] / ny, (double)d.size[2] / nz};
        auto world = [&](int axis, uint32_t voxel) { return (float)(-0.5 * (double)d.size[axis] + vs[axis] * (double)voxel); };
        for (uint32_t cz = 0; cz < gz; cz++)
            for (uint32_t cy = 0; cy < gy; cy++) {
                const uint8_t* row = &occ[((size_t)cz * gy + cy) * gx];
                for (uint32_t cx = 0; cx < gx;) {
                    if (!row[cx]) { cx++; continue; }
                    uint32_t e = cx;
                    while (e + 1 < gx && row[e + 1]) e++;
                    const float lo[3] = {world(0, 8 * cx), world(1, 8 * cy), world(2, 8 * cz)};
                    const float hi[3] = {world(0, std::min(8 * (e + 1), nx)), world(1, std::min(8 * (cy + 1), ny)), world(2, std::min(8 * (cz + 1), nz))};
                    boxes.insert(boxes.end(), {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]});
                    cx = e + 1;
                }
            }
        n_boxes = (uint32_t)(boxes.size() / 6);
        if (n_boxes) {
            dev_alloc(&d_boxes, boxes.size() * 4, "d_boxes");
            NRC_HIP(hipMemcpy(d_boxes, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice));
        }
    }

    void upload(const nrc_scene& s)
    {
        if (!s.density || s.nx == 0 || s.ny == 0 || s.nz == 0) fail("scene has no density volume");
        if (!(s.density_factor > 0.0f)) fail("scene density factor must be positive");
        const size_t nvox = (size_t)s.nx * s.ny * s.nz;
        // the kernels index voxels with 24-bit multiply-adds and read them through a raw buffer whose out-of-range offset is 2^31
        if (s.nx >= (1u << 24) || (size_t)s.ny * s.nz >= ((size_t)1 << 24) || nvox >= ((size_t)1 << 31))
            fail("density volume too large (needs nx < 2^24, ny*nz < 2^24 and fewer than 2^31 voxels)");
        dev_alloc(&d_density, nvox, "d_density");
        NRC_HIP(hipMemcpy(d_density, s.density, nvox, hipMemcpyHostToDevice));
        d.density = (const uint8_t*)d_density;
        d.nx = s.nx; d.ny = s.ny; d.nz = s.nz;
        d.fnx = (float)s.nx; d.fny = (float)s.ny; d.fnz = (float)s.nz;
        float size[3] = {s.size[0], s.size[1], s.size[2]};
        if (size[0] == 0.0f && size[1] == 0.0f && size[2] == 0.0f) {
            // volumeSizeF = normalize(extent) * 107.5 (src/NrcHpmRenderer.cu:910-912)
            const float l = sqrtf((d.fnx * d.fnx + d.fny * d.fny) + d.fnz * d.fnz);
            size[0] = d.fnx / l * 107.5f; size[1] = d.fny / l * 107.5f; size[2] = d.fnz / l * 107.5f;
        }
        for (int k = 0; k < 3; k++) {
            d.size[k] = size[k];
            d.half_size[k] = size[k] * 0.5f;
            d.inv_size[k] = 1.0f / size[k];
        }
        const float tx = 2.0f * size[0], ty = 2.0f * size[1], tz = 2.0f * size[2];
        // length(2 * skySize) with the math spec's dot product (a chain of single-rounding FMAs, nrc_math.h) -- what the oracle and
        // the device code compute; a plain (x*x + y*y) + z*z can differ in the last bit and with it every exit point
        d.len2size = sqrtf(fmaf(tz, tz, fmaf(ty, ty, tx * tx)));
        build_occupancy(s);
        build_occupancy_bits(s);
        set_params(s);
        d.env = nullptr; d.env_w = d.env_h = 0;
        if (s.env && s.env_w && s.env_h) {
            const size_t eb = (size_t)s.env_w * s.env_h * 16;
            dev_alloc(&d_env, eb, "d_env");
            NRC_HIP(hipMemcpy(d_env, s.env, eb, hipMemcpyHostToDevice));
            d.env = (const float*)d_env; d.env_w = s.env_w; d.env_h = s.env_h;
        }
    }
    // the uniform-buffer part of the scene (DirLight / PointLight / VolumeData / HdrEnvMap UBOs: src/DirLight.cpp:31-49,
    // src/HpmScene.cpp:56-76): kernels take DevScene by value, so the next launch sees the new values
    void set_params(const nrc_scene& s)
    {
        if (!(s.density_factor > 0.0f)) fail("scene density factor must be positive");
        d.density_factor = s.density_factor;
        d.inv_max_density = 1.0f / s.density_factor;
        d.g = s.g;
        for (int k = 0; k < 3; k++) {
            d.dir_light_dir[k] = s.dir_light_dir[k];
            d.point_light_pos[k] = s.point_light_pos[k];
            d.point_light_color[k] = s.point_light_color[k];
        }
        d.dir_light_strength = s.dir_light_strength;
        d.point_light_strength = s.point_light_strength;
        d.env_strength = s.env_strength;
    }
    ~SceneDev()
    {
        if (d_density) dev_free(d_density);
        if (d_env) dev_free(d_env);
        if (d_boxes) dev_free(d_boxes);
        if (d_occ_bits) dev_free(d_occ_bits);
    }
};

static DevCamera to_dev(const nrc_camera& c)
{
    DevCamera d;
    std::memcpy(d.m, c.inv_proj_view, sizeof(d.m));
    std::memcpy(d.pos, c.pos, sizeof(d.pos));
    return d;
}

// The empty-space tile mask projects occupancy boxes with the forward transform of the camera whose inverse the caller hands over.
// Returns false when the pair (invProjView, pos) is not a perspective camera looking from `pos` (then rays do not consist of the
// points that project onto their pixel and the mask is not used).
static bool forward_transform(const nrc_camera& c, DevProjView* out)
{
    double a[4][8];
    for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++) {
            a[r][k] = (double)c.inv_proj_view[4 * k + r];      // column-major
            a[r][4 + k] = r == k ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; col++) {                          // Gauss-Jordan with partial pivoting
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (!(std::fabs(a[piv][col]) > 1e-300)) return false;
        for (int k = 0; k < 8; k++) std::swap(a[col][k], a[piv][k]);
        const double inv = 1.0 / a[col][col];
        for (int k = 0; k < 8; k++) a[col][k] *= inv;
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            const double f = a[r][col];
            for (int k = 0; k < 8; k++) a[r][k] -= f * a[col][k];
        }
    }
    double m[16];
    for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++) m[4 * k + r] = a[r][4 + k];
    // the eye of a perspective transform maps to clip (0, 0, z, 0)
    double e[4];
    for (int r = 0; r < 4; r++) e[r] = m[r] * c.pos[0] + m[4 + r] * c.pos[1] + m[8 + r] * c.pos[2] + m[12 + r];
    const double scale = std::fabs(e[2]) + 1e-30;
    if (!(std::fabs(e[0]) <= 1e-4 * scale && std::fabs(e[1]) <= 1e-4 * scale && std::fabs(e[3]) <= 1e-4 * scale)) return false;
    for (int k = 0; k < 16; k++) {
        if (!std::isfinite(m[k])) return false;
        out->m[k] = (float)m[k];
    }
    return true;
}

static DevFrame make_frame(uint32_t w, uint32_t h, const nrc_tile* tile)
{
    DevFrame f{};
    f.w = w; f.h = h;
    nrc_tile t = tile ? *tile : nrc_tile{0, 1, w, h, 1};
    if (t.x_block == 0) t.x_block = 1;
    if (t.x_stride == 0 || t.global_w == 0 || t.global_h == 0) fail("bad tile description");
    if (t.x_block & (t.x_block - 1)) fail("tile: x_block must be a power of two");
    if (t.global_h != h) fail("tile: global_h must equal the local height (column sharding)");
    uint32_t b = 0;
    while ((1u << b) < t.x_block) b++;
    // the last local column's global column (see global_x in nrc_integrator.hip)
    const uint64_t last = (((uint64_t)t.x_offset + (uint64_t)((w - 1) >> b) * t.x_stride) << b) + ((w - 1) & (t.x_block - 1));
    if (last >= t.global_w) fail("tile columns exceed the global frame");
    f.x_offset = t.x_offset; f.x_stride = t.x_stride; f.x_block_log2 = b;
    f.inv_gw = 1.0f / (float)t.global_w;      // ONE_OVER_RENDER_WIDTH (nrc-constants.glsl:28)
    f.inv_gh = 1.0f / (float)t.global_h;
    return f;
}

// uncompressed scan-line EXR, FLOAT channels A,B,G,R (what tinyexr's SaveEXR writes for the reference, minus ZIP)
static void write_exr(const std::string& path, const std::vector<float>& rgba, uint32_t w, uint32_t h)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path);
    auto put = [&](const void* p, size_t n) { f.write((const char*)p, (std::streamsize)n); };
    auto u32 = [&](uint32_t v) { put(&v, 4); };
    auto i32 = [&](int32_t v) { put(&v, 4); };
    auto str = [&](const char* s) { put(s, std::strlen(s) + 1); };
    u32(20000630); u32(2);
    str("channels"); str("chlist"); i32(4 * 18 + 1);
    for (const char* cn : {"A", "B", "G", "R"}) { str(cn); i32(2); u32(0); i32(1); i32(1); }
    { char z = 0; put(&z, 1); }
    str("compression"); str("compression"); i32(1); { char z = 0; put(&z, 1); }
    auto box = [&](const char* name) { str(name); str("box2i"); i32(16); i32(0); i32(0); i32((int32_t)w - 1); i32((int32_t)h - 1); };
    box("dataWindow"); box("displayWindow");
    str("lineOrder"); str("lineOrder"); i32(1); { char z = 0; put(&z, 1); }
    { float one = 1.0f; str("pixelAspectRatio"); str("float"); i32(4); put(&one, 4); }
    { float z2[2] = {0, 0}; str("screenWindowCenter"); str("v2f"); i32(8); put(z2, 8); }
    { float one = 1.0f; str("screenWindowWidth"); str("float"); i32(4); put(&one, 4); }
    { char z = 0; put(&z, 1); }
    const uint64_t table_pos = (uint64_t)f.tellp();
    const uint64_t line_bytes = 8 + (uint64_t)w * 16;
    for (uint32_t y = 0; y < h; y++) { uint64_t off = table_pos + 8ull * h + y * line_bytes; put(&off, 8); }
    std::vector<float> line((size_t)w * 4);
    const int order[4] = {3, 2, 1, 0};
    for (uint32_t y = 0; y < h; y++) {
        i32((int32_t)y); i32((int32_t)(w * 16));
        for (int c = 0; c < 4; c++)
            for (uint32_t x = 0; x < w; x++) line[(size_t)c * w + x] = rgba[((size_t)y * w + x) * 4 + order[c]];
        put(line.data(), line.size() * 4);
    }
}

// ---------------------------------------------------------------------------------------------------- empty-space skip
// the free-flight table: one per device for the life of the process (32 MB, a function of the hash RNG alone), built by the first
// renderer that applies a tile mask; renderers read it only when their mask is (re)built (launch_flight_select)
static const float* flight_table()
{
    static std::mutex mu;
    static std::vector<float*> tables;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    NRC_HIP(hipGetDevice(&dev));
    if ((size_t)dev >= tables.size()) tables.resize((size_t)dev + 1, nullptr);
    if (tables[dev] == nullptr) {
        float* t = nullptr;
        dev_alloc(&t, (size_t)kFlightStates * sizeof(float), "t");
        launch_flight_table(t, nullptr);
        NRC_HIP(hipStreamSynchronize(nullptr));
        tables[dev] = t;
    }
    return tables[dev];
}
// The tile mask's guarantee -- a camera ray through provably empty space leaves the volume unscattered -- needs the walk to end
// before DeltaTrack's 128-collision cap (path_trace.glsl:161-173).  skip_lambda bounds the optical depth of ANY ray of the scene:
// a segment between two points the sphere tracing of find_entry_exit stops at (within 0.125 of the box) is at most the box
// diagonal + 0.25 long; 0.2 % on top covers the rounding of the walk's and the table's fp32 sums.  Past ~100 most RNG states can
// reach the cap (the 128 flights cover 128 +- 11 on average) and the mask would be rejected tile after tile: it is not built.
// flight_bits / d_scratch9: per-renderer buffers (1 MB, 36 bytes).  Ends with a 36-byte read-back on `s` (a host wait: the
// mask is rebuilt only when the camera or the medium changes).
static bool skip_setup(const DevScene& d, DevFrame* fr, uint32_t* d_flight_bits, uint32_t* d_scratch9, hipStream_t s)
{
    const double diag = std::sqrt((double)d.size[0] * d.size[0] + (double)d.size[1] * d.size[1] + (double)d.size[2] * d.size[2]);
    const double lambda = (double)d.density_factor * (diag + 0.5) * 1.002;
    fr->flight_mode = 0; fr->flight_n = 0; fr->flight_bits = nullptr;
    if (!(lambda <= 100.0)) return false;
    launch_flight_select(flight_table(), (float)lambda, d_scratch9, d_flight_bits, s);
    uint32_t h[1 + kFlightListMax];
    NRC_HIP(hipMemcpyAsync(h, d_scratch9, sizeof(h), hipMemcpyDeviceToHost, s));
    NRC_HIP(hipStreamSynchronize(s));
    if (h[0] <= kFlightListMax) {
        fr->flight_mode = 1; fr->flight_n = h[0];
        for (uint32_t k = 0; k < kFlightListMax; k++) fr->flight_list[k] = k < h[0] ? h[1 + k] : 0xffffffffu;
    } else {
        fr->flight_mode = 2; fr->flight_n = h[0];
        fr->flight_bits = d_flight_bits;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------- Renderer
class Renderer {
public:
    Renderer(uint32_t w, uint32_t h, bool blend, const nrc_camera& cam, const nrc_config& cfg, const nrc_scene& scene,
             Cache& cache, const nrc_tile* tile, hipStream_t s)
        : w_(w), h_(h), blend_(blend), cam_(to_dev(cam)), cfg_(cfg), cache_(cache), stream_(s), rng_(cfg.seed)
    {
        if (w == 0 || h == 0) fail("render size must be non-zero");
        if (((size_t)w * h) % 16 != 0) fail("NRC requires inferCount to be a multiple of 16");   // before anything is allocated
        frame_ = make_frame(w, h, tile);
        nq_ = query_count(w, h);      // the inference buffers hold whole 8x8 pixel tiles (tile-major, query_index)
        scene_.upload(scene);
        calc_train_subset(cfg.train_batch_count * cache.train_batch_size());
        tg_.spp = cfg.train_spp;
        // quirk Q2: trainRayLength never reaches the shader, TRAIN_RAY_LENGTH stays 1 (NrcHpmRenderer.cu:991-994 vs :1036-1055)
        tg_.ray_length = (cfg.compat_fix & NRC_FIX_Q2_TRAIN_RAY_LEN) ? cfg.train_ray_length : 1u;
        tg_.ring_size = (uint32_t)(cfg.train_ring_buf_size * (float)(tg_.tw * tg_.th));   // :253
        const size_t px = (size_t)w * h, T = (size_t)tg_.tw * tg_.th;
        // everything gen_rays writes exists kGenSets times: frame N's train-ray generation (stream D), inference and
        // compositing (stream C) read set N % kGenSets while gen_rays (stream A) is already one or two frames ahead
        for (int k = 0; k < kGenSets; k++) {
            alloc(&d_primary2_[k], px * 16); alloc(&d_info2_[k], px * 4); alloc(&d_origin2_[k], px * 16);
            alloc(&d_dir2_[k], px * 16); alloc(&d_infer_in2_[k], (size_t)nq_ * 20);
        }
        d_primary_ = d_primary2_[0]; d_info_ = d_info2_[0]; d_origin_ = d_origin2_[0]; d_dir_ = d_dir2_[0];
        d_infer_in_ = d_infer_in2_[0];
        for (auto& e : ev_train_done_) NRC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : ev_comp_done_) NRC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        alloc(&d_out_, px * 16); alloc(&d_infer_out_, (size_t)nq_ * 12);
        // train rays are double-buffered as well: frame N+1's train-ray generation (stream D) overlaps frame N's backward pass
        for (int k = 0; k < 2; k++) { alloc(&d_train_in2_[k], T * 20); alloc(&d_train_target2_[k], T * 12); }
        d_train_in_ = d_train_in2_[0]; d_train_target_ = d_train_target2_[0];
        for (auto& e : ev_prep_done_) NRC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : ev_infer_done_) NRC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ring_entries_ = std::max<size_t>(T, tg_.ring_size);
        alloc(&d_ring_, 8 + ring_entries_ * 24);
        alloc(&d_scratch_, (2 * T + 4) * 4);
        alloc(&d_fetch_, 8);
        alloc(&d_tile_mask_, (size_t)tile_mask_words(w, h) * 4);
        alloc(&d_flight_bits_, (size_t)kFlightStates / 8 + 8);
        alloc(&d_flight_sel_, (1 + kFlightListMax) * 4);
        for (auto& h : d_hot_) { alloc(&h, 64); NRC_HIP(hipMemset(h, 0, 64)); }
        hot_promote_ = getenv("NRC_NO_HOT_TILES") == nullptr;
        hot_ahead_ = getenv("NRC_HOT_TILES_INLINE") == nullptr;
        // Compositing as the epilogue of the inference launch: built, bit-identical to k_composite
        // (test_fused_composite_epilogue_equals_the_separate_pass), and OFF -- frame 0.2702-0.2725 -> 0.291-0.294 ms.  The epilogue's
        // loads (primary colour, scatter flag, framebuffer) are a dependent round trip to memory per 32-sample tile in a persistent
        // kernel that runs two waves per SIMD, dead tiles included, and 124 instead of 117 VGPRs make the workgroups wait longer
        // for room beside gen_rays: the inference stage went from 0.31 to 0.73 ms and pulled the training chain with it.
        // NRC_FUSED_COMPOSITE=1 switches it on.
        fuse_composite_ = getenv("NRC_FUSED_COMPOSITE") != nullptr;      // diagnostic: always compute the list in front of gen_rays
        // costliest-first launch order of gen_rays' tiles: costs of frame N order frame N + 2 (sorted on stream D beside frame N + 1)
        n_slots_ = camera_slots(w, h);
        alloc(&d_tile_cost_, (size_t)n_slots_ * 4);
        for (auto& o : d_tile_order_) alloc(&o, (size_t)n_slots_ * 4);
        {
            std::vector<uint32_t> ident(n_slots_);
            for (uint32_t i = 0; i < n_slots_; i++) ident[i] = i;
            for (auto& o : d_tile_order_) NRC_HIP(hipMemcpy(o, ident.data(), (size_t)n_slots_ * 4, hipMemcpyHostToDevice));
        }
        NRC_HIP(hipEventCreateWithFlags(&ev_order_done_, hipEventDisableTiming));
        cost_order_ = getenv("NRC_NO_COST_ORDER") == nullptr;
        if (const char* e = getenv("NRC_COST_ORDER_EVERY")) order_every_ = (uint64_t)std::max(2, atoi(e));
        order_neighbours_ = getenv("NRC_COST_ORDER_NEIGHBOURS") != nullptr;
        if (const char* e = getenv("NRC_COST_ORDER_KEEP")) order_keep_ = (uint32_t)std::min(31, std::max(0, atoi(e)));
        nrc_cam_ = cam;
        empty_skip_ = getenv("NRC_NO_EMPTY_SKIP") == nullptr;
        // train-ray generation + backward overlap inference + compositing on a second stream (NRC_SINGLE_STREAM=1 disables)
        // HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4): a same-priority second stream
        // can land on the queue of the first one (observed under torch.distributed, where RCCL owns several streams) and then
        // nothing overlaps.  A high-priority stream comes from a separate queue pool.
        dense_infer_ = getenv("NRC_DENSE_INFER") != nullptr;      // diagnostic: run the network on unscattered pixels too
        if (!getenv("NRC_SINGLE_STREAM")) {
            int lo = 0, hi = 0;
            NRC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            NRC_HIP(hipStreamCreateWithPriority(&stream_b_, hipStreamNonBlocking, hi));
            // third stream: inference + compositing of frame N run beside gen_rays of frame N+1 (MFMA beside VALU work)
            // fourth stream: train-ray generation, so that the training stream carries only backward + optimizer
            if (!getenv("NRC_TWO_STREAMS")) {
                NRC_HIP(hipStreamCreateWithPriority(&stream_c_, hipStreamNonBlocking, hi));
                NRC_HIP(hipStreamCreateWithPriority(&stream_d_, hipStreamNonBlocking, hi));
            }
        }
        // CreateNrcTrainRingBuffer: head = tail = 0, pos = 0, dir = (0,0,1)  (:866-875)
        std::vector<uint32_t> ring(2 + ring_entries_ * 6, 0);
        for (size_t r = 0; r < ring_entries_; r++) { float one = 1.0f; std::memcpy(&ring[2 + 6 * r + 5], &one, 4); }
        NRC_HIP(hipMemcpy(d_ring_, ring.data(), ring.size() * 4, hipMemcpyHostToDevice));
        cache_.init(nq_, (float*)d_infer_in_, (float*)d_infer_out_, (float*)d_train_in_, (float*)d_train_target_, s);
    }

    ~Renderer()
    {
        (void)hipDeviceSynchronize();
        for (void* p : allocs_) dev_free(p);
        for (auto& set : ev_pool_)
            for (auto& e : set) if (e) (void)hipEventDestroy(e);
        if (stream_b_) (void)hipStreamDestroy(stream_b_);
        if (stream_c_) (void)hipStreamDestroy(stream_c_);
        if (stream_d_) (void)hipStreamDestroy(stream_d_);
        for (auto& e : ev_prep_done_) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_infer_done_) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_train_done_) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_comp_done_) if (e) (void)hipEventDestroy(e);
        if (ev_consumer_) (void)hipEventDestroy(ev_consumer_);
        if (ev_order_done_) (void)hipEventDestroy(ev_order_done_);
        cache_.forget(this);
    }

    void render(bool train)      // NrcHpmRenderer::Render, :299-353
    {
        // one event quintuple per frame since the last stats reset (the reference has 8 Vulkan timestamp queries, :495-515)
        if (ev_used_ == ev_pool_.size()) {
            if (ev_pool_.size() >= 4096) ev_used_ = 0;      // wrap: statistics then cover the most recent frames only
            else {
                std::array<hipEvent_t, 10> set{};
                for (auto& e : set) NRC_HIP(hipEventCreate(&e));
                ev_pool_.push_back(set);
            }
        }
        hipEvent_t* ev_ = ev_pool_[ev_used_++].data();
        const float blend_factor = 1.0f / (float)blend_index_;
        // (the random numbers of an unpinned frame may have been drawn one frame early, for the hot-tile list: same sequence)
        if (have_pinned_random_) { std::memcpy(frame_.random, pinned_random_, 16); have_pinned_random_ = false; }
        else if (have_next_random_) { std::memcpy(frame_.random, next_random_, 16); have_next_random_ = false; }
        else draw_random(frame_.random);
        if (blend_) blend_index_++;
        // Frame graph on four streams, pipelined across frames (no host sync anywhere):
        //   A: [wait composite(N-6), train rays(N-6)] gen_rays(N)
        //   D: [wait gen_rays(N), train(N-2)] train-ray generation(N)
        //   B: [wait train rays(N)] backward(N) -> (all-reduce) -> [wait inference(N)] optimizer(N)
        //   C: [wait gen_rays(N), train(N-1)] inference(N) -> composite(N)
        // so frame N's train rays, training and inference all overlap frame N+1's gen_rays (the MFMA kernels and the short
        // latency-bound kernels run beside the VALU-bound integrator); inference(N+1) still sees the weights after frame N's
        // training (quirk Q13 ordering).  gen_rays' outputs exist in six sets (the chain gen_rays -> train rays -> training
        // -> next frame's inference -> compositing spans almost three frame times on one GPU, and the gradient all-reduce of a
        // multi-GPU run sits on it too), the train rays double-buffered.
        // events: 0 frame start, 1 gen_rays done, 2 train rays done (D), 3 inference done, 4 composite done, 5 training done (B)
        hipStream_t A = stream_, B = stream_b_ ? stream_b_ : stream_, Cs = stream_c_ ? stream_c_ : stream_;
        hipStream_t D = stream_d_ ? stream_d_ : B;
        update_tile_mask(A);
        const int pp = (int)(frame_index_ & 1u);                  // train-ray set, training / inference events
        const int gp = (int)(frame_index_ % (uint64_t)kGenSets);  // gen_rays output set
        d_primary_ = d_primary2_[gp]; d_info_ = d_info2_[gp]; d_origin_ = d_origin2_[gp]; d_dir_ = d_dir2_[gp];
        d_infer_in_ = d_infer_in2_[gp]; d_train_in_ = d_train_in2_[pp]; d_train_target_ = d_train_target2_[pp];
        if (frame_index_ >= (uint64_t)kGenSets) {      // set gp was last read by frame N - kGenSets
            if (Cs != A) NRC_HIP(hipStreamWaitEvent(A, ev_comp_done_[gp], 0));
            if (D != A) NRC_HIP(hipStreamWaitEvent(A, ev_prep_done_[gp], 0));
        }
        if (frame_index_ >= 2 && D != B) NRC_HIP(hipStreamWaitEvent(D, ev_train_done_[pp], 0));   // train-ray set pp: frame N-2
        // launch order: costliest tiles first, from the tile costs of an earlier frame.  Every order_every_ frames gen_rays
        // records its per-tile times and a sort at the end of stream D's work for the frame (behind the train rays, so training
        // does not wait for it) turns them into the other order buffer; the frames from two later on launch in that order (A
        // waits for the sort's event, long complete by then).  A fifth stream for the sort is not an option: HIP multiplexes
        // streams onto four hardware queues, and with five in use inference stopped overlapping gen_rays (frame 0.31 -> 0.40 ms).
        if (cost_order_ && order_pending_ && frame_index_ >= order_pending_frame_ + 2) {
            if (D != A) NRC_HIP(hipStreamWaitEvent(A, ev_order_done_, 0));
            order_cur_ ^= 1;
            order_pending_ = false;
        }
        const bool sample_cost = cost_order_ && !order_pending_ && (frame_index_ % order_every_ == 0 || order_resample_);
        if (sample_cost) order_resample_ = false;
        frame_.tile_order = cost_order_ ? (const uint32_t*)d_tile_order_[order_cur_] : nullptr;
        frame_.tile_cost = sample_cost ? (uint32_t*)d_tile_cost_ : nullptr;
        // the first samples of a view replace the costs (a cold first launch, another camera); later ones keep a decaying maximum
        frame_.tile_cost_keep = order_fresh_ > 0 ? 0u : order_keep_;
        if (sample_cost && order_fresh_ > 0) order_fresh_--;
        // Hot tiles (DevFrame::hot_tiles): the list for this frame's random numbers was built by the previous frame's gen_rays
        // (DevFrame::hot_next: no launch and no event of its own -- as a kernel on a side stream it cost a cross-stream hand-over
        // of ~0.01 ms per frame, whichever stream and position); when it was not -- first frame, pinned random numbers, another
        // mask -- k_hot_tiles computes it here, in front of gen_rays.
        frame_.hot_tiles = nullptr;
        frame_.hot_next = frame_.hot_reset = nullptr;
        const int hb = (int)(frame_index_ % 3u);
        const bool promote = hot_promote_ && frame_.tile_mask != nullptr && frame_.flight_mode == 1u && frame_.flight_n > 0;
        if (promote) {
            if (!hot_chain_) {      // the rotation starts (again): all three lists empty
                for (auto& h : d_hot_) NRC_HIP(hipMemsetAsync(h, 0, 64, A));
                for (bool& r : hot_ready_) r = false;
            }
            last_hot_predicted_ = hot_ready_[hb] && hot_epoch_[hb] == mask_epoch_ && std::memcmp(hot_random_[hb], frame_.random, 16) == 0;
            if (!last_hot_predicted_) {
                NRC_HIP(hipMemsetAsync((uint32_t*)d_hot_[hb] + kHotTilesMax, 0, 4, A));
                launch_hot_tiles(frame_, (uint32_t*)d_hot_[hb], A);
            }
            frame_.hot_tiles = (const uint32_t*)d_hot_[hb];
            hot_ready_[hb] = false;
            if (hot_ahead_) {
                // the next frame's random numbers: announced by the caller (render_frames), or drawn now instead of then
                const float* nr = next_random_;
                if (have_hint_) nr = hint_random_;
                else if (!have_next_random_) { draw_random(next_random_); have_next_random_ = true; }
                const int hn = (hb + 1) % 3;
                std::memcpy(frame_.random_next, nr, 16);
                frame_.hot_next = (uint32_t*)d_hot_[hn];
                frame_.hot_reset = (uint32_t*)d_hot_[(hb + 2) % 3] + kHotTilesMax;
                hot_ready_[hn] = true;
                hot_epoch_[hn] = mask_epoch_;
                std::memcpy(hot_random_[hn], nr, 16);
            }
        }
        have_hint_ = false;
        hot_chain_ = promote && hot_ahead_;
        last_hot_ = promote ? hb : -1;
        NRC_HIP(hipEventRecord(ev_[0], A));
        launch_gen_rays(scene_.d, cam_, frame_, cfg_.primary_ray_length, cfg_.primary_ray_prob, (float*)d_primary_,
                        (float*)d_info_, (float*)d_origin_, (float*)d_dir_, (float*)d_infer_in_,
                        count_fetches_ ? (unsigned long long*)d_fetch_ : nullptr, tg_, full_vertex_images_, A);
        NRC_HIP(hipEventRecord(ev_[1], A));
        if (D != A) NRC_HIP(hipStreamWaitEvent(D, ev_[1], 0));
        if (Cs != A) NRC_HIP(hipStreamWaitEvent(Cs, ev_[1], 0));
        // the reference records prep_train_rays into every frame's pre-CUDA command buffer (:2039-2040), trained or not
        launch_prep_train(scene_.d, frame_, tg_, (const float*)d_info_, (const float*)d_origin_, (const float*)d_dir_,
                          (uint32_t*)d_ring_, (uint32_t*)d_scratch_, (float*)d_train_in_, (float*)d_train_target_, D);
        if (stage_events_) NRC_HIP(hipEventRecord(ev_[2], D));
        NRC_HIP(hipEventRecord(ev_prep_done_[gp], D));
        if (B != D) NRC_HIP(hipStreamWaitEvent(B, ev_prep_done_[gp], 0));
        if (sample_cost) {      // order buffer cur^1: its last reader is a gen_rays before this one on A
            launch_tile_order((const uint32_t*)d_tile_cost_, n_slots_, (uint32_t*)d_tile_order_[order_cur_ ^ 1], w_, order_neighbours_, D);
            NRC_HIP(hipEventRecord(ev_order_done_, D));
            order_pending_ = true;
            order_pending_frame_ = frame_index_;
        }
        if (B != Cs && frame_index_ > 0) NRC_HIP(hipStreamWaitEvent(Cs, ev_train_done_[pp ^ 1], 0));   // weights of frame N-1
        // (re)bind this renderer's I/O buffers: several renderers may share one cache (Reference::CompareNrc evaluates the
        // same NRC from another camera, src/Reference.cpp:71-107)
        cache_.acquire(this, Cs, B);
        cache_.bind(nq_, (float*)d_infer_in_, (float*)d_infer_out_, (float*)d_train_in_,
                    (float*)d_train_target_);
        // Compositing (nrc/render.comp) CAN be the epilogue of the inference launch for the fused 6x64 model (fuse_composite_, off by
        // default: slower, see the constructor): the queries are in tile-major order, so an inference tile's 32 pixels are four
        // 128-byte row segments of the images.  The generic models always use k_composite.  (The framebuffer is ONE image that compositing
        // blends in place: a consumer stream that was handed the previous frame and announced the end of its read holds this
        // frame's compositing back until then -- here in front of the fused launch, below in front of k_composite.)
        const bool fused_composite = fuse_composite_ && cache_.can_composite();
        if (fused_composite && consumer_pending_) {
            NRC_HIP(hipStreamWaitEvent(Cs, ev_consumer_, 0));
            consumer_pending_ = false;
        }
        CompositeArgs comp{(const float*)d_primary_, (const float*)d_info_, (float*)d_out_, w_, h_, show_nrc_, 0u, blend_factor};
        // no host read-back of the batch filter: every batch is launched, all-zero (unscattered) query tiles skip the network
        cache_.infer_all(nullptr, Cs, !dense_infer_, fused_composite ? &comp : nullptr);
        if (stage_events_) NRC_HIP(hipEventRecord(ev_[3], Cs));
        NRC_HIP(hipEventRecord(ev_infer_done_[pp], Cs));
        if (train) cache_.train_all(B, (B != Cs && frame_index_ > 0) ? ev_infer_done_[pp ^ 1] : nullptr,
                                    B != Cs ? ev_infer_done_[pp] : nullptr);
        if (stage_events_) NRC_HIP(hipEventRecord(ev_[5], B));
        NRC_HIP(hipEventRecord(ev_train_done_[pp], B));
        // the framebuffer is ONE image that compositing blends in place: a consumer stream that was handed the previous frame
        // (framebuffer_on) and announced the end of its read (release_frame) holds this frame's compositing back until then
        if (consumer_pending_) {
            NRC_HIP(hipStreamWaitEvent(Cs, ev_consumer_, 0));
            consumer_pending_ = false;
        }
        if (!fused_composite)
            launch_composite(frame_, show_nrc_, blend_factor, (const float*)d_primary_, (const float*)d_info_,
                             (const float*)d_infer_out_, (float*)d_out_, Cs);
        if (stage_events_) NRC_HIP(hipEventRecord(ev_[4], Cs));
        NRC_HIP(hipEventRecord(ev_comp_done_[gp], Cs));
        frame_index_++;
        timed_ = true;
    }

    // everything this renderer has enqueued (both streams) is complete
    void sync()
    {
        NRC_HIP(hipStreamSynchronize(stream_));
        if (stream_b_) NRC_HIP(hipStreamSynchronize(stream_b_));
        if (stream_c_) NRC_HIP(hipStreamSynchronize(stream_c_));
        if (stream_d_) NRC_HIP(hipStreamSynchronize(stream_d_));
    }

    // (re)builds the empty-space tile mask for the current camera on stream A, in front of the next gen_rays (stream order)
    void update_tile_mask(hipStream_t A)
    {
        if (!mask_dirty_) return;
        mask_dirty_ = false;
        mask_epoch_++;
        DevProjView pv;
        frame_.tile_mask = nullptr;
        if (!empty_skip_ || !forward_transform(nrc_cam_, &pv) || !skip_setup(scene_.d, &frame_, (uint32_t*)d_flight_bits_, (uint32_t*)d_flight_sel_, A)) return;
        launch_tile_mask((const float*)scene_.d_boxes, scene_.n_boxes, pv, frame_, (uint32_t*)d_tile_mask_, A);
        frame_.tile_mask = (const uint32_t*)d_tile_mask_;
    }
    void set_empty_skip(bool on)
    {
        sync();       // a frame in flight may be reading the mask
        empty_skip_ = on;
        mask_dirty_ = true;
    }

    void set_cost_order(bool on)
    {
        sync();
        cost_order_ = on;
    }
    void set_hot_tiles(bool on)
    {
        sync();
        hot_promote_ = on;
    }
    // the random numbers of the frame after the next one to be rendered (render_frames knows them): the hot-tile list is computed
    // for these instead of numbers drawn ahead
    void hint_next_random(const float* r) { std::memcpy(hint_random_, r, 16); have_hint_ = true; }
    // the hot-tile list of the last frame: kHotTilesMax entries + count; returns 1 when it was computed one frame ahead, 0 when in
    // front of gen_rays, -1 when the frame had none
    int hot_tiles(uint32_t* out9)
    {
        sync();
        if (last_hot_ < 0) return -1;
        NRC_HIP(hipMemcpy(out9, d_hot_[last_hot_], (kHotTilesMax + 1) * 4, hipMemcpyDeviceToHost));
        return last_hot_predicted_ ? 1 : 0;
    }
    // the permutation the next frame launches its tiles in (after the pending sort, if one is due)
    size_t tile_order(uint32_t* host_out, size_t capacity)
    {
        sync();
        if (host_out == nullptr) return n_slots_;
        if (capacity < n_slots_) throw std::invalid_argument("tile_order: capacity below the number of tile slots");
        const int cur = (order_pending_ && frame_index_ >= order_pending_frame_ + 2) ? order_cur_ ^ 1 : order_cur_;
        NRC_HIP(hipMemcpy(host_out, d_tile_order_[cur], (size_t)n_slots_ * 4, hipMemcpyDeviceToHost));
        return n_slots_;
    }

    void set_camera(const nrc_camera& c)       // SetCamera, :561-604: reset blending, clear the accumulation images
    {
        sync();
        cam_ = to_dev(c);
        nrc_cam_ = c;
        mask_dirty_ = true;
        order_resample_ = true;      // the tile costs belong to the old view: measure again with the next frame
        order_fresh_ = 2;
        blend_index_ = 1;
        const size_t px = (size_t)w_ * h_;
        NRC_HIP(hipMemsetAsync(d_out_, 0, px * 16, stream_));
        for (int k = 0; k < kGenSets; k++) {
            NRC_HIP(hipMemsetAsync(d_primary2_[k], 0, px * 16, stream_));
            NRC_HIP(hipMemsetAsync(d_info2_[k], 0, px * 4, stream_));
        }
    }
    void set_blend(bool b) { blend_ = b; blend_index_ = 1; }      // :606-610
    // the timing events of train-ray generation, training, inference and compositing (four timed hipEventRecord per frame, the
    // reference's timestamp queries :495-515) can be switched off; gen_rays stays bracketed
    void set_stage_events(bool on) { sync(); stage_events_ = on; ev_used_ = 0; timed_ = false; }
    void set_scene_params(const nrc_scene& s) { scene_.set_params(s); mask_dirty_ = true; }      // density_factor enters skip_lambda
    void set_show_nrc(bool s) { show_nrc_ = s ? 1u : 0u; }
    void set_frame_random(const float* r) { std::memcpy(pinned_random_, r, 16); have_pinned_random_ = true; }
    void set_count_fetches(bool on)
    {
        count_fetches_ = on;
        NRC_HIP(hipMemsetAsync(d_fetch_, 0, 8, stream_));
    }
    unsigned long long fetches()
    {
        unsigned long long v = 0;
        NRC_HIP(hipMemcpyAsync(&v, d_fetch_, 8, hipMemcpyDeviceToHost, stream_));
        NRC_HIP(hipStreamSynchronize(stream_));
        return v;
    }

    // average stage times over the frames rendered since the last reset; returns the number of frames
    uint32_t stage_stats(float* avg8, bool reset)
    {
        const size_t n = ev_used_;
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (avg8 == nullptr) {      // reset only: no event is read (a caller in front of a timed region must not keep the GPU waiting)
            if (reset) { ev_used_ = 0; timed_ = false; }
            return (uint32_t)n;
        }
        if (n > 0) sync();
        for (size_t f = 0; f < n; f++) {
            float st[8];
            stage_times(ev_pool_[f].data(), st, stage_events_);
            for (int k = 0; k < 8; k++) acc[k] += st[k];
        }
        if (avg8) for (int k = 0; k < 8; k++) avg8[k] = n ? (float)(acc[k] / (double)n) : 0.0f;
        if (reset) { ev_used_ = 0; timed_ = false; }
        return (uint32_t)n;
    }

    float frame_time_ms(float* stage)       // EvaluateTimestampQueries / GetFrameTimeMS, :495-530
    {
        if (!timed_ || ev_used_ == 0) return 0.0f;
        hipEvent_t* ev_ = ev_pool_[ev_used_ - 1].data();
        sync();
        float st[8];
        stage_times(ev_, st, stage_events_);
        if (stage) for (int k = 0; k < 8; k++) stage[k] = st[k];
        return st[7];
    }

    // {clear(0), gen_rays, prep_infer(0: fused into gen_rays), train (second stream), prep_train (second stream),
    //  inference, composite, total}
    static void stage_times(hipEvent_t* e, float* st, bool all_stages)
    {
        float gen = 0, prep = 0, inf = 0, trn = 0, comp = 0, total = 0;
        NRC_HIP(hipEventElapsedTime(&gen, e[0], e[1]));
        if (!all_stages) {      // only gen_rays is bracketed (set_stage_events(false)): the other stages read 0, the total is gen_rays'
            for (int k = 0; k < 8; k++) st[k] = 0.0f;
            st[1] = gen; st[7] = gen;
            return;
        }
        NRC_HIP(hipEventElapsedTime(&prep, e[1], e[2]));
        NRC_HIP(hipEventElapsedTime(&inf, e[1], e[3]));
        NRC_HIP(hipEventElapsedTime(&trn, e[2], e[5]));
        NRC_HIP(hipEventElapsedTime(&comp, e[3], e[4]));
        float ta = 0, tb = 0;
        NRC_HIP(hipEventElapsedTime(&ta, e[0], e[4]));
        NRC_HIP(hipEventElapsedTime(&tb, e[0], e[5]));
        total = ta > tb ? ta : tb;       // latency of this frame; throughput is higher (frames are pipelined)
        st[0] = 0; st[1] = gen; st[2] = 0; st[3] = trn; st[4] = prep; st[5] = inf; st[6] = comp; st[7] = total;
    }

    void export_exr(const char* path)
    {
        std::vector<float> host((size_t)w_ * h_ * 4);
        sync();
        NRC_HIP(hipMemcpyAsync(host.data(), d_out_, host.size() * 4, hipMemcpyDeviceToHost, stream_));
        NRC_HIP(hipStreamSynchronize(stream_));
        write_exr(path, host, w_, h_);
    }

    void* buffer(int which, size_t* bytes)
    {
        sync();      // intermediate buffers are produced on both streams
        const size_t px = (size_t)w_ * h_, T = (size_t)tg_.tw * tg_.th;
        void* p = nullptr; size_t b = 0;
        // which + 16 * (k + 1): buffer `which` (0..4) of gen_rays output set k instead of the last frame's set (diagnostics)
        if (which >= 16) {
            const int k = which / 16 - 1;
            if (k >= kGenSets || which % 16 > 4) fail("bad buffer id");
            void* const* sets[5] = {d_primary2_, d_info2_, d_origin2_, d_dir2_, d_infer_in2_};
            const size_t sizes[5] = {px * 16, px * 4, px * 16, px * 16, px * 20};
            if (bytes) *bytes = sizes[which % 16];
            if (which % 16 == 4) return public_queries(sets[4][k], 5);
            return sets[which % 16][k];
        }
        switch (which) {
        case 0: p = d_primary_; b = px * 16; break;
        case 1: p = d_info_; b = px * 4; break;
        case 2: p = d_origin_; b = px * 16; break;
        case 3: p = d_dir_; b = px * 16; break;
        case 4: p = public_queries(d_infer_in_, 5); b = px * 20; break;
        case 5: p = public_queries(d_infer_out_, 3); b = px * 12; break;
        case 6: p = d_train_in_; b = T * 20; break;
        case 7: p = d_train_target_; b = T * 12; break;
        case 8: p = d_ring_; b = 8 + ring_entries_ * 24; break;
        default: fail("bad buffer id");
        }
        if (bytes) *bytes = b;
        return p;
    }
    // the reference's x * H + y order of a query / radiance buffer (a copy: the renderer itself keeps them tile-major)
    void* public_queries(const void* tiled, uint32_t floats_per_query)
    {
        void*& dst = floats_per_query == 5 ? d_pub_in_ : d_pub_out_;
        if (!dst) alloc(&dst, (size_t)w_ * h_ * floats_per_query * 4);
        launch_query_layout(frame_, floats_per_query, (const float*)tiled, (float*)dst, stream_);
        NRC_HIP(hipStreamSynchronize(stream_));
        return dst;
    }
    // The compositing kernel runs on the renderer's own stream C.  Handing out the framebuffer orders the caller's stream
    // behind the latest compositing pass (device-side wait, no host block), so whatever the caller enqueues next on the stream
    // it passed in -- a copy, a display blit, CompareImages -- sees the finished frame, as with the reference's single queue.
    const float* framebuffer()
    {
        wait_frame(stream_);
        return (const float*)d_out_;
    }
    // orders any stream of the caller behind the latest compositing pass: a display / read-back stream can follow the frames
    // without stalling the render stream (which framebuffer() does)
    void wait_frame(hipStream_t consumer)
    {
        if (frame_index_ > 0)      // recorded on whichever stream composited (the render stream itself in the reduced modes)
            NRC_HIP(hipStreamWaitEvent(consumer, ev_comp_done_[(frame_index_ - 1) % (uint64_t)kGenSets], 0));
    }
    // the consumer's reads of the framebuffer enqueued so far on `consumer` finish before the next frame is composited
    void release_frame(hipStream_t consumer)
    {
        if (!ev_consumer_) NRC_HIP(hipEventCreateWithFlags(&ev_consumer_, hipEventDisableTiming));
        NRC_HIP(hipEventRecord(ev_consumer_, consumer));
        consumer_pending_ = true;
    }
    bool is_blending() const { return blend_; }
    void set_full_vertex_images(bool on) { full_vertex_images_ = on; }
    // bytes of NRC vertex data (origin + direction) gen_rays stores per frame
    size_t vertex_image_bytes() const { return (full_vertex_images_ ? (size_t)w_ * h_ : (size_t)tg_.tw * tg_.th) * 32; }
    const float* framebuffer_unordered() const { return (const float*)d_out_; }
    const TrainGrid& train_grid() const { return tg_; }
    hipStream_t stream() const { return stream_; }

private:
    void alloc(void** p, size_t bytes)
    {
        dev_alloc(p, bytes, "renderer buffer");
        NRC_HIP(hipMemset(*p, 0, bytes));     // the reference never clears its images at creation (quirk Q14); this build does
        allocs_.push_back(*p);
    }

    void calc_train_subset(uint32_t train_pixel_count)       // CalcTrainSubset, :612-642
    {
        const uint32_t sq = (uint32_t)std::sqrt((double)train_pixel_count);
        for (uint32_t factor = sq; factor >= 2; factor--) {
            if (train_pixel_count % factor == 0) {
                const uint32_t other = train_pixel_count / factor;
                const uint32_t bigger = std::max(factor, other), smaller = std::min(factor, other);
                if (w_ > h_) { tg_.tw = bigger; tg_.th = smaller; }
                else { tg_.tw = smaller; tg_.th = bigger; }
                tg_.x_dist = w_ / tg_.tw;
                const uint32_t y_dist = h_ / tg_.th;
                // quirk Q1: TRAIN_Y_DIST is specialised from trainXDist (:966-969)
                tg_.y_dist = (cfg_.compat_fix & NRC_FIX_Q1_TRAIN_Y_DIST) ? y_dist : tg_.x_dist;
                return;
            }
        }
        fail("Could not find suitable division of trainPixelCount");
    }

    uint32_t w_, h_;
    bool blend_;
    uint32_t blend_index_ = 1;
    uint32_t show_nrc_ = 1;
    DevCamera cam_;
    DevFrame frame_{};
    nrc_config cfg_;
    Cache& cache_;
    hipStream_t stream_;
    std::mt19937 rng_;
    SceneDev scene_;
    TrainGrid tg_{};
    size_t ring_entries_ = 0;
    void *d_primary_ = nullptr, *d_info_ = nullptr, *d_origin_ = nullptr, *d_dir_ = nullptr, *d_out_ = nullptr;
#ifndef NRC_GEN_SETS
#define NRC_GEN_SETS 6      // 4 -> 6: the 6x64 frame 0.2713 -> 0.2682 ms, the HashGrid frame 0.802 -> 0.758 ms (its chain is 3.4 ms long); 8 adds nothing
#endif
    static constexpr int kGenSets = NRC_GEN_SETS;
    void *d_info2_[kGenSets] = {}, *d_origin2_[kGenSets] = {}, *d_dir2_[kGenSets] = {};
    void *d_primary2_[kGenSets] = {}, *d_infer_in2_[kGenSets] = {};
    bool fuse_composite_ = false;
    uint32_t nq_ = 0;                                   // queries per frame in the renderer's tile-major order (query_count)
    void *d_pub_in_ = nullptr, *d_pub_out_ = nullptr;   // x * H + y copies of the query / radiance buffers, made when asked for
    hipEvent_t ev_train_done_[2] = {nullptr, nullptr}, ev_comp_done_[kGenSets] = {};
    uint64_t frame_index_ = 0;
    void *d_infer_in_ = nullptr, *d_infer_out_ = nullptr, *d_train_in_ = nullptr, *d_train_target_ = nullptr;
    void *d_ring_ = nullptr, *d_scratch_ = nullptr, *d_fetch_ = nullptr;
    std::vector<void*> allocs_;
    std::vector<std::array<hipEvent_t, 10>> ev_pool_;
    hipStream_t stream_b_ = nullptr, stream_c_ = nullptr, stream_d_ = nullptr;
    void *d_train_in2_[2] = {nullptr, nullptr}, *d_train_target2_[2] = {nullptr, nullptr};
    hipEvent_t ev_prep_done_[kGenSets] = {}, ev_infer_done_[2] = {nullptr, nullptr};
    size_t ev_used_ = 0;
    bool timed_ = false;
    float pinned_random_[4] = {0, 0, 0, 0};
    bool have_pinned_random_ = false;
    bool count_fetches_ = false;
    bool dense_infer_ = false;
    hipEvent_t ev_consumer_ = nullptr;
    bool consumer_pending_ = false;
    bool full_vertex_images_ = false;
    void* d_tile_mask_ = nullptr;
    void *d_flight_bits_ = nullptr, *d_flight_sel_ = nullptr;
    // hot-tile lists (DevFrame::hot_tiles), three in rotation (read / appended by the same gen_rays / zeroed for the one after);
    // [k] was built for the random numbers hot_random_[k]
    void* d_hot_[3] = {nullptr, nullptr, nullptr};
    bool hot_promote_ = true, hot_ahead_ = true, hot_chain_ = false, hot_ready_[3] = {false, false, false};
    uint64_t hot_epoch_[3] = {0, 0, 0}, mask_epoch_ = 0;
    float hot_random_[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float next_random_[4] = {0, 0, 0, 0}, hint_random_[4] = {0, 0, 0, 0};
    bool have_next_random_ = false, have_hint_ = false, last_hot_predicted_ = false;
    int last_hot_ = -1;
    void draw_random(float* r4)
    {
        std::uniform_real_distribution<float> u(0.0f, 1.0f);
        for (int k = 0; k < 4; k++) r4[k] = u(rng_);
    }
    void* d_tile_cost_ = nullptr;
    void* d_tile_order_[2] = {nullptr, nullptr};
    uint32_t n_slots_ = 0;
    hipEvent_t ev_order_done_ = nullptr;
    bool cost_order_ = true, order_pending_ = false;
    int order_cur_ = 0;
    uint64_t order_every_ = 4, order_pending_frame_ = 0;
    bool order_neighbours_ = false;
    bool order_resample_ = false;
    uint32_t order_keep_ = 4;        // DevFrame::tile_cost_keep once the view's first two samples are in
    int order_fresh_ = 2;
    nrc_camera nrc_cam_{};
    bool mask_dirty_ = true, empty_skip_ = true;
    bool stage_events_ = getenv("NRC_NO_STAGE_EVENTS") == nullptr;
};

// ---------------------------------------------------------------------------------------------------- McRenderer
class McRenderer {
public:
    McRenderer(uint32_t w, uint32_t h, uint32_t path_length, bool blend, const nrc_camera& cam, const nrc_scene& scene,
               const nrc_tile* tile, hipStream_t s)
        : w_(w), h_(h), path_length_(path_length), blend_(blend), cam_(to_dev(cam)), stream_(s), rng_(1337)
    {
        if (w == 0 || h == 0) fail("render size must be non-zero");
        frame_ = make_frame(w, h, tile);
        scene_.upload(scene);
        const size_t px = (size_t)w * h;
        dev_alloc(&d_out_, px * 16, "d_out_"); NRC_HIP(hipMemset(d_out_, 0, px * 16));
        dev_alloc(&d_info_, px * 4, "d_info_"); NRC_HIP(hipMemset(d_info_, 0, px * 4));
        dev_alloc(&d_fetch_, 8, "d_fetch_"); NRC_HIP(hipMemset(d_fetch_, 0, 8));
        dev_alloc(&d_tile_mask_, (size_t)tile_mask_words(w, h) * 4, "d_tile_mask_");
        dev_alloc(&d_flight_bits_, (size_t)kFlightStates / 8 + 8, "d_flight_bits_");
        dev_alloc(&d_flight_sel_, (1 + kFlightListMax) * 4, "d_flight_sel_");
        dev_alloc(&d_hot_, 64, "d_hot_");
        NRC_HIP(hipMemset(d_hot_, 0, 64));
        hot_promote_ = getenv("NRC_NO_HOT_TILES") == nullptr;
        NRC_HIP(hipEventCreate(&ev_[0])); NRC_HIP(hipEventCreate(&ev_[1]));
        nrc_cam_ = cam;
        empty_skip_ = getenv("NRC_NO_EMPTY_SKIP") == nullptr;
        // costliest-first tile launch order, as in the NRC renderer (one stream here: the sort simply follows the sampled frame)
        cost_order_ = getenv("NRC_NO_COST_ORDER") == nullptr;
        n_slots_ = camera_slots(w, h);
        dev_alloc(&d_tile_cost_, (size_t)n_slots_ * 4, "d_tile_cost_");
        dev_alloc(&d_tile_order_, (size_t)n_slots_ * 4, "d_tile_order_");
        std::vector<uint32_t> ident(n_slots_);
        for (uint32_t i = 0; i < n_slots_; i++) ident[i] = i;
        NRC_HIP(hipMemcpy(d_tile_order_, ident.data(), (size_t)n_slots_ * 4, hipMemcpyHostToDevice));
    }
    ~McRenderer()
    {
        if (d_out_) dev_free(d_out_);
        if (d_info_) dev_free(d_info_);
        if (d_fetch_) dev_free(d_fetch_);
        if (d_tile_mask_) dev_free(d_tile_mask_);
        if (d_flight_bits_) dev_free(d_flight_bits_);
        if (d_flight_sel_) dev_free(d_flight_sel_);
        if (d_hot_) dev_free(d_hot_);
        if (d_tile_cost_) dev_free(d_tile_cost_);
        if (d_tile_order_) dev_free(d_tile_order_);
        for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
    }
    void set_empty_skip(bool on) { empty_skip_ = on; mask_dirty_ = true; }      // same stream: ordered behind frames in flight
    void render()        // McHpmRenderer::Render, src/McHpmRenderer.cpp:121-151
    {
        if (mask_dirty_) {      // empty-space tile mask for the current camera (see Renderer::update_tile_mask)
            mask_dirty_ = false;
            DevProjView pv;
            frame_.tile_mask = nullptr;
            if (empty_skip_ && forward_transform(nrc_cam_, &pv) && skip_setup(scene_.d, &frame_, (uint32_t*)d_flight_bits_, (uint32_t*)d_flight_sel_, stream_)) {
                launch_tile_mask((const float*)scene_.d_boxes, scene_.n_boxes, pv, frame_, (uint32_t*)d_tile_mask_, stream_);
                frame_.tile_mask = (const uint32_t*)d_tile_mask_;
            }
        }
        const float blend_factor = 1.0f / (float)blend_index_;
        if (have_pinned_random_) { std::memcpy(frame_.random, pinned_random_, 16); have_pinned_random_ = false; }
        else { std::uniform_real_distribution<float> u(0.0f, 1.0f); for (float& r : frame_.random) r = u(rng_); }
        if (blend_) blend_index_++;
        const bool sample_cost = cost_order_ && (frame_index_ % 4 == 0 || order_fresh_ == 2);
        frame_.tile_order = cost_order_ ? (const uint32_t*)d_tile_order_ : nullptr;
        frame_.tile_cost = sample_cost ? (uint32_t*)d_tile_cost_ : nullptr;
        frame_.tile_cost_keep = order_fresh_ > 0 ? 0u : 4u;      // see Renderer::render
        if (sample_cost && order_fresh_ > 0) order_fresh_--;
        // hot tiles as in Renderer::render, computed in front of the launch (two tiny launches beside a 2.5 ms frame): a pixel in a
        // capped RNG state inside an empty tile otherwise starts its 32-vertex walk at the end of the launch
        frame_.hot_tiles = nullptr;
        frame_.hot_next = frame_.hot_reset = nullptr;
        if (hot_promote_ && frame_.tile_mask != nullptr && frame_.flight_mode == 1u && frame_.flight_n > 0) {
            NRC_HIP(hipMemsetAsync((uint32_t*)d_hot_ + kHotTilesMax, 0, 4, stream_));
            launch_hot_tiles(frame_, (uint32_t*)d_hot_, stream_);
            frame_.hot_tiles = (const uint32_t*)d_hot_;
        }
        NRC_HIP(hipEventRecord(ev_[0], stream_));
        launch_mc_render(scene_.d, cam_, frame_, path_length_, blend_factor, (float*)d_out_, (float*)d_info_,
                         count_fetches_ ? (unsigned long long*)d_fetch_ : nullptr, stream_);
        NRC_HIP(hipEventRecord(ev_[1], stream_));
        if (sample_cost) launch_tile_order((const uint32_t*)d_tile_cost_, n_slots_, (uint32_t*)d_tile_order_, w_, false, stream_);
        frame_index_++;
        timed_ = true;
    }
    void set_cost_order(bool on) { cost_order_ = on; }
    void set_camera(const nrc_camera& c)
    {
        cam_ = to_dev(c);
        nrc_cam_ = c;
        mask_dirty_ = true;
        order_fresh_ = 2;
        blend_index_ = 1;
        NRC_HIP(hipMemsetAsync(d_out_, 0, (size_t)w_ * h_ * 16, stream_));
    }
    void set_scene_params(const nrc_scene& s) { scene_.set_params(s); mask_dirty_ = true; }
    void set_blend(bool b) { blend_ = b; blend_index_ = 1; }
    bool is_blending() const { return blend_; }
    void set_frame_random(const float* r) { std::memcpy(pinned_random_, r, 16); have_pinned_random_ = true; }
    void set_count_fetches(bool on) { count_fetches_ = on; NRC_HIP(hipMemsetAsync(d_fetch_, 0, 8, stream_)); }
    unsigned long long fetches()
    {
        unsigned long long v = 0;
        NRC_HIP(hipMemcpyAsync(&v, d_fetch_, 8, hipMemcpyDeviceToHost, stream_));
        NRC_HIP(hipStreamSynchronize(stream_));
        return v;
    }
    float frame_time_ms()
    {
        if (!timed_) return 0.0f;
        float ms = 0;
        NRC_HIP(hipEventSynchronize(ev_[1]));
        NRC_HIP(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
        return ms;
    }
    void export_exr(const char* path)
    {
        std::vector<float> host((size_t)w_ * h_ * 4);
        NRC_HIP(hipMemcpyAsync(host.data(), d_out_, host.size() * 4, hipMemcpyDeviceToHost, stream_));
        NRC_HIP(hipStreamSynchronize(stream_));
        write_exr(path, host, w_, h_);
    }
    const float* framebuffer() const { return (const float*)d_out_; }

private:
    uint32_t w_, h_, path_length_;
    bool blend_;
    uint32_t blend_index_ = 1;
    DevCamera cam_;
    DevFrame frame_{};
    hipStream_t stream_;
    std::mt19937 rng_;
    SceneDev scene_;
    void *d_out_ = nullptr, *d_info_ = nullptr, *d_fetch_ = nullptr, *d_tile_mask_ = nullptr;
    void *d_flight_bits_ = nullptr, *d_flight_sel_ = nullptr;
    void* d_hot_ = nullptr;      // hot-tile list (DevFrame::hot_tiles), rebuilt in front of every launch
    bool hot_promote_ = true;
    nrc_camera nrc_cam_{};
    bool mask_dirty_ = true, empty_skip_ = true;
    hipEvent_t ev_[2] = {nullptr, nullptr};
    bool timed_ = false;
    float pinned_random_[4] = {0, 0, 0, 0};
    bool have_pinned_random_ = false;
    bool count_fetches_ = false;
    bool cost_order_ = true;
    int order_fresh_ = 2;            // samples that replace the tile costs instead of keeping their decaying maximum
    void *d_tile_cost_ = nullptr, *d_tile_order_ = nullptr;
    uint32_t n_slots_ = 0;
    uint64_t frame_index_ = 0;
};

}  // namespace nrc

// ==================================================================================================== C ABI
struct nrc_cache { nrc::Cache impl; explicit nrc_cache(const nrc_config& c) : impl(c) {} };
struct nrc_renderer { nrc::Renderer impl; template <class... A> explicit nrc_renderer(A&&... a) : impl(std::forward<A>(a)...) {} };
struct nrc_mc_renderer { nrc::McRenderer impl; template <class... A> explicit nrc_mc_renderer(A&&... a) : impl(std::forward<A>(a)...) {} };

template <class F>
static int guarded(F f)
{
    try {
        f();
        return NRC_OK;
    } catch (const nrc::HipError& e) {
        nrc::g_last_error = e.what();
        return NRC_ERR_HIP;
    } catch (const std::logic_error& e) {
        nrc::g_last_error = e.what();
        return NRC_ERR_STATE;
    } catch (const std::exception& e) {
        nrc::g_last_error = e.what();
        return NRC_ERR_INVALID;
    } catch (...) {
        nrc::g_last_error = "unknown error";
        return NRC_ERR_INVALID;
    }
}
#define NRC_REQUIRE(p) do { if (!(p)) { nrc::g_last_error = "SkyRenderer ERROR: null argument: " #p; return NRC_ERR_INVALID; } } while (0)

extern "C" {

const char* nrc_last_error(void) { return nrc::g_last_error.c_str(); }
const char* nrc_version(void) { return "nrc-hpm-renderer_amd 0.1 (gfx950)"; }

void nrc_config_default(nrc_config* c)     // src/main.cu:432-439, with posID 3 / dirID 0 (BASELINE.json north-star)
{
    std::memset(c, 0, sizeof(*c));
    std::strcpy(c->loss_fn, "RelativeL2Luminance");
    std::strcpy(c->optimizer, "Adam");
    c->learning_rate = 0.01f; c->ema_decay = 0.99f;
    c->pos_id = 3; c->dir_id = 0;
    c->nn_width = 64; c->nn_depth = 6;
    c->log2_infer_batch_size = 21; c->log2_train_batch_size = 14; c->train_batch_count = 4;
    c->scene_id = 4;
    c->train_ring_buf_size = 1.0f; c->train_spp = 1; c->primary_ray_length = 1; c->primary_ray_prob = 0.0f;
    c->train_ray_length = 32;
    c->seed = 1337; c->compat_fix = 0;
}

int nrc_cache_create(const nrc_config* cfg, nrc_cache_t** out)
{
    NRC_REQUIRE(cfg); NRC_REQUIRE(out);
    return guarded([&] { *out = new nrc_cache(*cfg); });
}
int nrc_cache_init(nrc_cache_t* c, uint32_t infer_count, float* d_in, float* d_out, float* d_tin, float* d_tt, void* stream)
{
    NRC_REQUIRE(c);
    return guarded([&] { c->impl.init(infer_count, d_in, d_out, d_tin, d_tt, (hipStream_t)stream); });
}
int nrc_cache_init_events(nrc_cache_t* c, uint32_t infer_count, float* d_in, float* d_out, float* d_tin, float* d_tt, void* stream,
                          void* start_event, void* finished_event)
{
    NRC_REQUIRE(c);
    return guarded([&] { c->impl.init(infer_count, d_in, d_out, d_tin, d_tt, (hipStream_t)stream, (hipEvent_t)start_event,
                                      (hipEvent_t)finished_event); });
}
int nrc_cache_infer_and_train(nrc_cache_t* c, const uint32_t* filter, int train)
{
    NRC_REQUIRE(c);
    return guarded([&] { c->impl.infer_and_train(filter, train != 0); });
}
int nrc_cache_destroy(nrc_cache_t* c)
{
    if (!c) return NRC_OK;
    return guarded([&] { delete c; });
}
float nrc_cache_get_loss(nrc_cache_t* c)
{
    float v = NAN;
    if (c) guarded([&] { v = c->impl.get_loss(true); });
    return v;
}
int nrc_cache_get_loss_async(nrc_cache_t* c, float* loss, uint32_t* step, uint32_t* steps_enqueued)
{
    NRC_REQUIRE(c); NRC_REQUIRE(loss);
    return guarded([&] {
        *loss = c->impl.get_loss(false, step);
        if (steps_enqueued) *steps_enqueued = c->impl.loss_steps_enqueued();
    });
}
float nrc_cache_get_loss_blocking(nrc_cache_t* c)
{
    float v = NAN;
    if (c) guarded([&] { v = c->impl.get_loss(true); });
    return v;
}
size_t nrc_cache_get_infer_batch_count(nrc_cache_t* c) { return c ? c->impl.infer_batch_count() : 0; }
size_t nrc_cache_get_train_batch_count(nrc_cache_t* c) { return c ? c->impl.train_batch_count() : 0; }
uint32_t nrc_cache_get_infer_batch_size(nrc_cache_t* c) { return c ? c->impl.infer_batch_size() : 0; }
uint32_t nrc_cache_get_train_batch_size(nrc_cache_t* c) { return c ? c->impl.train_batch_size() : 0; }

int nrc_cache_set_stream(nrc_cache_t* c, void* stream)
{
    NRC_REQUIRE(c);
    return guarded([&] { c->impl.set_stream((hipStream_t)stream); });
}
int nrc_cache_infer(nrc_cache_t* c, const float* d_in, float* d_out, uint32_t n, int use_ema)
{
    NRC_REQUIRE(c); NRC_REQUIRE(d_in); NRC_REQUIRE(d_out);
    return guarded([&] {
        c->impl.acquire(&c->impl, c->impl.stream(), c->impl.stream());
        c->impl.mlp().infer(d_in, d_out, n, use_ema != 0, c->impl.stream());
    });
}
int nrc_cache_backward(nrc_cache_t* c, const float* d_in, const float* d_target, uint32_t n, uint32_t n_norm)
{
    NRC_REQUIRE(c); NRC_REQUIRE(d_in); NRC_REQUIRE(d_target);
    return guarded([&] {
        c->impl.acquire(&c->impl, c->impl.stream(), c->impl.stream());
        c->impl.mlp().backward(d_in, d_target, n, n_norm ? n_norm : n, c->impl.stream());
        // the documented protocol (nrc_hpm.h): backward, all-reduce of nrc_cache_grad_ptr, optimizer step -- once a caller holds
        // the pointer, the fp32 vector (which it may have reduced) is what the optimizer must read, also for a trainable table
        // (whose packed fp16 gradient is read otherwise); unmodified, the widened fp32 copy holds the same values
        if (c->impl.grad_ptr_exposed()) c->impl.mlp().grad_vector_is_source();
        c->impl.push_loss(c->impl.stream());
    });
}
int nrc_cache_optimizer_step(nrc_cache_t* c)
{
    NRC_REQUIRE(c);
    return guarded([&] {
        c->impl.acquire(&c->impl, c->impl.stream(), c->impl.stream());
        c->impl.mlp().optimizer_step(c->impl.stream());
    });
}
float* nrc_cache_grad_ptr(nrc_cache_t* c)
{
    if (!c) return nullptr;
    c->impl.expose_grad_ptr();      // from now on the stand-alone optimizer step reads this vector, not the packed table gradient
    c->impl.mlp().grad_vector_is_source();
    return c->impl.mlp().grad_ptr();
}
uint32_t nrc_cache_param_count(nrc_cache_t* c) { return c ? c->impl.mlp().n_params() : 0; }
float* nrc_cache_loss_ptr(nrc_cache_t* c) { return c ? c->impl.mlp().loss_ptr() : nullptr; }
int nrc_cache_set_grad_hook(nrc_cache_t* c, nrc_grad_hook hook, void* user)
{
    NRC_REQUIRE(c);
    c->impl.set_hook(hook, user);
    return NRC_OK;
}
int nrc_comm_unique_id(void* out128)
{
    NRC_REQUIRE(out128);
    return guarded([&] {
        ncclUniqueId id;
        nrc::Rccl::get().check(nrc::Rccl::get().get_unique_id(&id), "ncclGetUniqueId");
        std::memcpy(out128, &id, sizeof(id));
    });
}
int nrc_cache_comm_init(nrc_cache_t* c, const void* unique_id128, int rank, int world)
{
    NRC_REQUIRE(c); NRC_REQUIRE(unique_id128);
    return guarded([&] { c->impl.comm_init(unique_id128, rank, world); });
}
int nrc_cache_comm_info(nrc_cache_t* c, int* rank, int* world)
{
    NRC_REQUIRE(c); NRC_REQUIRE(rank); NRC_REQUIRE(world);
    return guarded([&] { c->impl.comm_info(rank, world); });
}
int nrc_cache_comm_time_exchange(nrc_cache_t* c, uint32_t reps, float* avg_us)
{
    NRC_REQUIRE(c); NRC_REQUIRE(avg_us);
    return guarded([&] { *avg_us = c->impl.time_exchange(reps); });
}
int nrc_cache_comm_sparse(nrc_cache_t* c) { return c && c->impl.sparse_grid_exchange() ? 1 : 0; }
size_t nrc_cache_grid_list_capacity(nrc_cache_t* c) { return c ? c->impl.grid_list_capacity() : 0; }
int nrc_cache_grid_grad_pack(nrc_cache_t* c, uint32_t* host_list, size_t list_words)
{
    NRC_REQUIRE(c); NRC_REQUIRE(host_list);
    return guarded([&] { c->impl.grid_pack_host(host_list, list_words); });
}
int nrc_cache_grid_grad_apply(nrc_cache_t* c, const uint32_t* host_lists, uint32_t n_lists)
{
    NRC_REQUIRE(c); NRC_REQUIRE(host_lists); NRC_REQUIRE(n_lists > 0);
    return guarded([&] { c->impl.grid_apply_host(host_lists, n_lists); });
}
int nrc_cache_set_loss_norm_factor(nrc_cache_t* c, uint32_t factor)
{
    NRC_REQUIRE(c);
    c->impl.set_loss_norm_factor(factor);
    return NRC_OK;
}
int nrc_cache_get_params(nrc_cache_t* c, int which, float* host_out)
{
    NRC_REQUIRE(c); NRC_REQUIRE(host_out);
    return guarded([&] {
        NRC_HIP(hipDeviceSynchronize());      // training may be in flight on a renderer's internal stream
        NRC_HIP(hipMemcpy(host_out, c->impl.mlp().buffer(which), (size_t)c->impl.mlp().n_params() * 4, hipMemcpyDeviceToHost));
    });
}
int nrc_cache_set_params(nrc_cache_t* c, int which, const float* host_in)
{
    NRC_REQUIRE(c); NRC_REQUIRE(host_in);
    return guarded([&] {
        NRC_HIP(hipDeviceSynchronize());      // training / inference may be in flight on a renderer's internal streams
        NRC_HIP(hipMemcpy(c->impl.mlp().buffer(which), host_in, (size_t)c->impl.mlp().n_params() * 4, hipMemcpyHostToDevice));
        if (which == 4) c->impl.mlp().grad_vector_is_source();
        if (which == 0 || which == 1) { c->impl.mlp().repack(c->impl.stream()); NRC_HIP(hipStreamSynchronize(c->impl.stream())); }
    });
}
int nrc_cache_get_step(nrc_cache_t* c, uint32_t* step)
{
    NRC_REQUIRE(c); NRC_REQUIRE(step);
    *step = c->impl.mlp().step;
    return NRC_OK;
}
int nrc_cache_set_step(nrc_cache_t* c, uint32_t step)
{
    NRC_REQUIRE(c);
    c->impl.mlp().step = step;
    return NRC_OK;
}

int nrc_renderer_create(uint32_t w, uint32_t h, int blend, const nrc_camera* cam, const nrc_config* cfg, const nrc_scene* scene,
                        nrc_cache_t* cache, const nrc_tile* tile, void* stream, nrc_renderer_t** out)
{
    NRC_REQUIRE(cam); NRC_REQUIRE(cfg); NRC_REQUIRE(scene); NRC_REQUIRE(cache); NRC_REQUIRE(out);
    return guarded([&] { *out = new nrc_renderer(w, h, blend != 0, *cam, *cfg, *scene, cache->impl, tile, (hipStream_t)stream); });
}
int nrc_renderer_render(nrc_renderer_t* r, int train) { NRC_REQUIRE(r); return guarded([&] { r->impl.render(train != 0); }); }
int nrc_renderer_render_frames(nrc_renderer_t* r, uint32_t n_frames, const float* frame_randoms, int train)
{
    NRC_REQUIRE(r);
    return guarded([&] {
        for (uint32_t f = 0; f < n_frames; f++) {
            if (frame_randoms) {
                r->impl.set_frame_random(frame_randoms + 4 * (size_t)f);
                if (f + 1 < n_frames) r->impl.hint_next_random(frame_randoms + 4 * (size_t)(f + 1));
            }
            r->impl.render(train != 0);
        }
    });
}
int nrc_renderer_set_camera(nrc_renderer_t* r, const nrc_camera* c) { NRC_REQUIRE(r); NRC_REQUIRE(c); return guarded([&] { r->impl.set_camera(*c); }); }
int nrc_renderer_set_scene_params(nrc_renderer_t* r, const nrc_scene* scene)
{
    NRC_REQUIRE(r); NRC_REQUIRE(scene);
    return guarded([&] { r->impl.set_scene_params(*scene); });
}
int nrc_mc_renderer_set_scene_params(nrc_mc_renderer_t* r, const nrc_scene* scene)
{
    NRC_REQUIRE(r); NRC_REQUIRE(scene);
    return guarded([&] { r->impl.set_scene_params(*scene); });
}
int nrc_renderer_set_blend(nrc_renderer_t* r, int b) { NRC_REQUIRE(r); r->impl.set_blend(b != 0); return NRC_OK; }
int nrc_renderer_set_show_nrc(nrc_renderer_t* r, int s) { NRC_REQUIRE(r); r->impl.set_show_nrc(s != 0); return NRC_OK; }
int nrc_renderer_set_frame_random(nrc_renderer_t* r, const float* v) { NRC_REQUIRE(r); NRC_REQUIRE(v); r->impl.set_frame_random(v); return NRC_OK; }
const float* nrc_renderer_framebuffer(nrc_renderer_t* r) { return r ? r->impl.framebuffer() : nullptr; }
const float* nrc_renderer_framebuffer_on(nrc_renderer_t* r, void* consumer_stream)
{
    if (!r) return nullptr;
    const float* p = nullptr;
    if (guarded([&] { r->impl.wait_frame((hipStream_t)consumer_stream); p = r->impl.framebuffer_unordered(); }) != NRC_OK) return nullptr;
    return p;
}
int nrc_renderer_release_frame(nrc_renderer_t* r, void* consumer_stream)
{
    NRC_REQUIRE(r);
    return guarded([&] { r->impl.release_frame((hipStream_t)consumer_stream); });
}
int nrc_renderer_set_stage_events(nrc_renderer_t* r, int on)
{
    NRC_REQUIRE(r);
    return guarded([&] { r->impl.set_stage_events(on != 0); });
}
int nrc_renderer_set_empty_skip(nrc_renderer_t* r, int on)
{
    NRC_REQUIRE(r);
    return guarded([&] { r->impl.set_empty_skip(on != 0); });
}
int nrc_renderer_set_cost_order(nrc_renderer_t* r, int on)
{
    NRC_REQUIRE(r);
    return guarded([&] { r->impl.set_cost_order(on != 0); });
}
size_t nrc_renderer_tile_order(nrc_renderer_t* r, uint32_t* host_out, size_t capacity)
{
    if (!r) return 0;
    size_t n = 0;
    if (guarded([&] { n = r->impl.tile_order(host_out, capacity); }) != NRC_OK) return 0;
    return n;
}
int nrc_renderer_set_hot_tiles(nrc_renderer_t* r, int on)
{
    NRC_REQUIRE(r);
    return guarded([&] { r->impl.set_hot_tiles(on != 0); });
}
int nrc_renderer_hot_tiles(nrc_renderer_t* r, uint32_t* host_out9)
{
    NRC_REQUIRE(r); NRC_REQUIRE(host_out9);
    int rc = -2;
    if (guarded([&] { rc = r->impl.hot_tiles(host_out9); }) != NRC_OK) return -2;
    return rc;
}
int nrc_mc_renderer_set_cost_order(nrc_mc_renderer_t* r, int on)
{
    NRC_REQUIRE(r);
    return guarded([&] { r->impl.set_cost_order(on != 0); });
}
int nrc_mc_renderer_set_empty_skip(nrc_mc_renderer_t* r, int on)
{
    NRC_REQUIRE(r);
    return guarded([&] { r->impl.set_empty_skip(on != 0); });
}
int nrc_renderer_set_full_vertex_images(nrc_renderer_t* r, int on)
{
    NRC_REQUIRE(r);
    r->impl.set_full_vertex_images(on != 0);
    return NRC_OK;
}
size_t nrc_renderer_vertex_image_bytes(nrc_renderer_t* r) { return r ? r->impl.vertex_image_bytes() : 0; }
int nrc_renderer_is_blending(nrc_renderer_t* r) { return r && r->impl.is_blending() ? 1 : 0; }
int nrc_mc_renderer_is_blending(nrc_mc_renderer_t* r) { return r && r->impl.is_blending() ? 1 : 0; }
int nrc_renderer_export_exr(nrc_renderer_t* r, const char* path) { NRC_REQUIRE(r); NRC_REQUIRE(path); return guarded([&] { r->impl.export_exr(path); }); }
float nrc_renderer_frame_time_ms(nrc_renderer_t* r, float* stage_ms)
{
    float v = -1.0f;
    if (r) guarded([&] { v = r->impl.frame_time_ms(stage_ms); });
    return v;
}
int nrc_renderer_stage_stats(nrc_renderer_t* r, float avg_ms[8], uint32_t* frames, int reset)
{
    NRC_REQUIRE(r);
    return guarded([&] { uint32_t n = r->impl.stage_stats(avg_ms, reset != 0); if (frames) *frames = n; });
}
int nrc_renderer_destroy(nrc_renderer_t* r)
{
    if (!r) return NRC_OK;
    return guarded([&] { r->impl.sync(); delete r; });
}
void* nrc_renderer_buffer(nrc_renderer_t* r, int which, size_t* bytes)
{
    void* p = nullptr;
    if (r) guarded([&] { p = r->impl.buffer(which, bytes); });
    return p;
}
int nrc_renderer_train_grid(nrc_renderer_t* r, uint32_t out5[5])
{
    NRC_REQUIRE(r); NRC_REQUIRE(out5);
    const nrc::TrainGrid& t = r->impl.train_grid();
    out5[0] = t.tw; out5[1] = t.th; out5[2] = t.x_dist; out5[3] = t.y_dist; out5[4] = t.ring_size;
    return NRC_OK;
}
int nrc_renderer_count_fetches(nrc_renderer_t* r, int enable, unsigned long long* out)
{
    NRC_REQUIRE(r);
    return guarded([&] { if (out) *out = r->impl.fetches(); r->impl.set_count_fetches(enable != 0); });
}

int nrc_mc_renderer_create(uint32_t w, uint32_t h, uint32_t path_length, int blend, const nrc_camera* cam, const nrc_scene* scene,
                           const nrc_tile* tile, void* stream, nrc_mc_renderer_t** out)
{
    NRC_REQUIRE(cam); NRC_REQUIRE(scene); NRC_REQUIRE(out);
    return guarded([&] { *out = new nrc_mc_renderer(w, h, path_length, blend != 0, *cam, *scene, tile, (hipStream_t)stream); });
}
int nrc_mc_renderer_render(nrc_mc_renderer_t* r) { NRC_REQUIRE(r); return guarded([&] { r->impl.render(); }); }
int nrc_mc_renderer_set_camera(nrc_mc_renderer_t* r, const nrc_camera* c) { NRC_REQUIRE(r); NRC_REQUIRE(c); return guarded([&] { r->impl.set_camera(*c); }); }
int nrc_mc_renderer_set_blend(nrc_mc_renderer_t* r, int b) { NRC_REQUIRE(r); r->impl.set_blend(b != 0); return NRC_OK; }
int nrc_mc_renderer_set_frame_random(nrc_mc_renderer_t* r, const float* v) { NRC_REQUIRE(r); NRC_REQUIRE(v); r->impl.set_frame_random(v); return NRC_OK; }
const float* nrc_mc_renderer_framebuffer(nrc_mc_renderer_t* r) { return r ? r->impl.framebuffer() : nullptr; }
int nrc_mc_renderer_export_exr(nrc_mc_renderer_t* r, const char* path) { NRC_REQUIRE(r); NRC_REQUIRE(path); return guarded([&] { r->impl.export_exr(path); }); }
float nrc_mc_renderer_frame_time_ms(nrc_mc_renderer_t* r)
{
    float v = -1.0f;
    if (r) guarded([&] { v = r->impl.frame_time_ms(); });
    return v;
}
int nrc_mc_renderer_count_fetches(nrc_mc_renderer_t* r, int enable, unsigned long long* out)
{
    NRC_REQUIRE(r);
    return guarded([&] { if (out) *out = r->impl.fetches(); r->impl.set_count_fetches(enable != 0); });
}
int nrc_mc_renderer_destroy(nrc_mc_renderer_t* r)
{
    if (!r) return NRC_OK;
    return guarded([&] { delete r; });
}

int nrc_compare_images(const float* d_ref, const float* d_own, uint32_t w, uint32_t h, void* stream, float result5[5])
{
    NRC_REQUIRE(d_ref); NRC_REQUIRE(d_own); NRC_REQUIRE(result5);
    return guarded([&] {
        double* scratch = nullptr;
        float* d_res = nullptr;
        nrc::dev_alloc(&scratch, (8 + 5 * 256) * sizeof(double), "compare scratch");
        nrc::dev_alloc(&d_res, 5 * sizeof(float), "compare result");
        nrc::launch_compare(d_ref, d_own, w * h, scratch, d_res, (hipStream_t)stream);
        NRC_HIP(hipMemcpyAsync(result5, d_res, 5 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
        NRC_HIP(hipStreamSynchronize((hipStream_t)stream));
        nrc::dev_free(scratch);
        nrc::dev_free(d_res);
    });
}

int nrc_image_create(uint32_t w, uint32_t h, const float* host_rgba, float** d_out)
{
    NRC_REQUIRE(d_out); NRC_REQUIRE(w > 0 && h > 0);
    return guarded([&] {
        float* d = nullptr;
        nrc::dev_alloc(&d, (size_t)w * h * 16, "image");
        if (host_rgba) NRC_HIP(hipMemcpy(d, host_rgba, (size_t)w * h * 16, hipMemcpyHostToDevice));
        else NRC_HIP(hipMemset(d, 0, (size_t)w * h * 16));
        *d_out = d;
    });
}
int nrc_image_destroy(float* d_image)
{
    if (!d_image) return NRC_OK;
    return guarded([&] { nrc::dev_free(d_image); });
}

int nrc_debug_check_guards(char* message, size_t message_bytes)
{
    // NRC_GUARD_ALLOC=1: number of allocations (live ones checked now + freed ones found damaged) whose canaries were overwritten
    nrc::AllocRegistry& r = nrc::alloc_registry();
    if (message && message_bytes) message[0] = 0;
    if (!r.guard) return -1;
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(r.mu);
    unsigned long long n = r.violations;
    std::string first = r.first_violation;
    for (auto& kv : r.live) {
        nrc::AllocInfo chk = kv.second;
        chk.bytes = (chk.bytes + 255) & ~(size_t)255;
        std::string where;
        if (nrc::guard_damage(chk, &where)) { n++; if (first.empty()) first = where; }
    }
    if (message && message_bytes) { std::strncpy(message, first.c_str(), message_bytes - 1); message[message_bytes - 1] = 0; }
    return (int)n;
}

int nrc_test_math(int fn, const float* d_a, const float* d_b, uint32_t n, float* d_out, float* d_out2, void* stream)
{
    NRC_REQUIRE(d_a); NRC_REQUIRE(d_out);
    return guarded([&] { nrc::launch_test_math(fn, d_a, d_b ? d_b : d_a, n, d_out, d_out2, (hipStream_t)stream); });
}
int nrc_test_rng(float u, float v, const float fr[4], uint32_t n, float* d_out, void* stream)
{
    NRC_REQUIRE(fr); NRC_REQUIRE(d_out);
    return guarded([&] { nrc::launch_test_rng(u, v, fr, n, d_out, (hipStream_t)stream); });
}

}  // extern "C"
