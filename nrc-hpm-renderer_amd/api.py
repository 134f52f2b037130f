"""Host-side mirror of the reference's interface for the hot path, over the C ABI of include/nrc_hpm.h.

Same class / method names and argument meaning as the reference (so parity tests read like reference call sites):
  AppConfig             include/engine/AppConfig.hpp:9-66, src/AppConfig.cpp:154-182 (17 positional CLI args)
  NeuralRadianceCache   include/engine/graphics/NeuralRadianceCache.hpp:13-32
  NrcHpmRenderer        include/engine/graphics/renderer/NrcHpmRenderer.hpp:16-41
  McHpmRenderer         include/engine/graphics/renderer/McHpmRenderer.hpp:16-31
Errors raise RuntimeError("SkyRenderer ERROR: ...") like Log::Error(msg, true) (src/Log.cpp:16-20).

PyTorch is plumbing only (device memory, current stream, torch.distributed); all arithmetic runs in libnrc_hpm.so.
There is no CPU fallback: loading fails loudly if the HIP library is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NRC_HPM_LIB: tooling override (tools/loop_profile.py loads an instrumented build)
LIB_PATH = os.environ.get("NRC_HPM_LIB") or os.path.join(_HERE, "lib", "libnrc_hpm.so")

NRC_FIX_Q1_TRAIN_Y_DIST = 1
NRC_FIX_Q2_TRAIN_RAY_LEN = 2


class NrcConfig(C.Structure):
    _fields_ = [
        ("loss_fn", C.c_char * 32), ("optimizer", C.c_char * 32),
        ("learning_rate", C.c_float), ("ema_decay", C.c_float),
        ("pos_id", C.c_uint32), ("dir_id", C.c_uint32), ("nn_width", C.c_uint32), ("nn_depth", C.c_uint32),
        ("log2_infer_batch_size", C.c_uint32), ("log2_train_batch_size", C.c_uint32), ("train_batch_count", C.c_uint32),
        ("scene_id", C.c_uint32), ("train_ring_buf_size", C.c_float), ("train_spp", C.c_uint32),
        ("primary_ray_length", C.c_uint32), ("primary_ray_prob", C.c_float), ("train_ray_length", C.c_uint32),
        ("seed", C.c_uint32), ("compat_fix", C.c_uint32), ("hashgrid_log2_size", C.c_uint32),
    ]


class NrcScene(C.Structure):
    _fields_ = [
        ("density", C.c_void_p), ("nx", C.c_uint32), ("ny", C.c_uint32), ("nz", C.c_uint32),
        ("size", C.c_float * 3), ("density_factor", C.c_float), ("g", C.c_float),
        ("dir_light_dir", C.c_float * 3), ("dir_light_strength", C.c_float),
        ("point_light_pos", C.c_float * 3), ("point_light_strength", C.c_float),
        ("point_light_color", C.c_float * 3), ("env_strength", C.c_float),
        ("env", C.c_void_p), ("env_w", C.c_uint32), ("env_h", C.c_uint32),
    ]


class NrcCamera(C.Structure):
    _fields_ = [("inv_proj_view", C.c_float * 16), ("pos", C.c_float * 3)]


class NrcTile(C.Structure):
    _fields_ = [("x_offset", C.c_uint32), ("x_stride", C.c_uint32), ("global_w", C.c_uint32), ("global_h", C.c_uint32),
                ("x_block", C.c_uint32)]


GRAD_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p)
ALLREDUCE_F64_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p)
ALLGATHER_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

# every symbol include/nrc_hpm.h declares (tests/test_abi.py checks the built library exports all of them)
ABI_SYMBOLS = [
    "nrc_last_error", "nrc_version", "nrc_config_default",
    "nrc_cache_create", "nrc_cache_init", "nrc_cache_init_events", "nrc_cache_infer_and_train", "nrc_cache_destroy", "nrc_cache_get_loss",
    "nrc_cache_get_loss_blocking", "nrc_cache_get_loss_async", "nrc_cache_comm_info", "nrc_cache_comm_time_exchange", "nrc_renderer_release_frame", "nrc_renderer_is_blending", "nrc_mc_renderer_is_blending",
    "nrc_renderer_set_full_vertex_images", "nrc_renderer_vertex_image_bytes", "nrc_renderer_set_empty_skip", "nrc_mc_renderer_set_empty_skip",
    "nrc_cache_set_collective_hooks", "nrc_renderer_gather_frame", "nrc_renderer_export_exr_gathered", "nrc_compare_images_sharded",
    "nrc_renderer_set_cost_order", "nrc_renderer_set_schedule", "nrc_renderer_get_schedule", "nrc_renderer_schedule_source", "nrc_renderer_schedule_key",
    "nrc_schedule_cache_load", "nrc_schedule_cache_save", "nrc_schedule_cache_clear", "nrc_renderer_tile_order", "nrc_mc_renderer_set_cost_order", "nrc_renderer_set_hot_tiles", "nrc_renderer_hot_tiles",
    "nrc_cache_get_infer_batch_count", "nrc_cache_get_train_batch_count", "nrc_cache_get_infer_batch_size",
    "nrc_cache_get_train_batch_size", "nrc_cache_infer", "nrc_cache_backward", "nrc_cache_optimizer_step",
    "nrc_cache_grad_ptr", "nrc_cache_param_count", "nrc_cache_loss_ptr", "nrc_cache_set_loss_norm_factor",
    "nrc_comm_unique_id", "nrc_cache_comm_init", "nrc_cache_comm_sparse", "nrc_cache_grid_list_capacity",
    "nrc_cache_set_exchange_dtype", "nrc_cache_get_exchange_dtype",
    "nrc_cache_grid_grad_pack", "nrc_cache_grid_grad_apply",
    "nrc_cache_set_stream", "nrc_cache_set_grad_hook", "nrc_cache_get_params", "nrc_cache_set_params",
    "nrc_cache_get_step", "nrc_cache_set_step", "nrc_cache_param_count_tcnn", "nrc_cache_get_params_tcnn", "nrc_cache_set_params_tcnn",
    "nrc_cache_save_checkpoint", "nrc_cache_load_checkpoint", "nrc_cache_comm_status", "nrc_cache_set_comm_timeout_ms",
    "nrc_renderer_create", "nrc_renderer_render", "nrc_renderer_render_frames", "nrc_renderer_set_stage_events", "nrc_renderer_set_camera", "nrc_renderer_set_blend",
    "nrc_renderer_set_scene_params", "nrc_mc_renderer_set_scene_params",
    "nrc_renderer_set_show_nrc", "nrc_renderer_set_frame_random", "nrc_renderer_framebuffer", "nrc_renderer_framebuffer_on",
    "nrc_renderer_export_exr",
    "nrc_renderer_frame_time_ms", "nrc_renderer_stage_stats", "nrc_renderer_frame_timeline", "nrc_set_wave_priority_raise", "nrc_renderer_destroy", "nrc_renderer_buffer", "nrc_renderer_count_fetches",
    "nrc_renderer_train_grid",
    "nrc_mc_renderer_create", "nrc_mc_renderer_render", "nrc_mc_renderer_set_camera", "nrc_mc_renderer_set_blend",
    "nrc_mc_renderer_set_frame_random", "nrc_mc_renderer_framebuffer", "nrc_mc_renderer_export_exr",
    "nrc_mc_renderer_frame_time_ms", "nrc_mc_renderer_count_fetches", "nrc_mc_renderer_destroy",
    "nrc_compare_images", "nrc_test_math", "nrc_test_rng", "nrc_debug_check_guards", "nrc_image_create", "nrc_image_destroy",
]

_lib = None


def set_wave_priority_raise(on=True):
    """nrc_set_wave_priority_raise: 1 (default) every kernel of the library at s_setprio 3, 0 at the hardware default (a process whose foreign
    kernels -- torch's NCCL all-reduce, fills, copies -- run beside the renderer)"""
    _check(load_library().nrc_set_wave_priority_raise(C.c_int(int(bool(on)))))


def build_id():
    """the hash of sources + compiler + flags the loaded library was built from (csrc/Makefile: NRC_BUILD_ID, the tail of nrc_version())"""
    v = load_library().nrc_version().decode()
    return v.rsplit("build ", 1)[1] if "build " in v else "unknown (a library older than the build id, loaded through NRC_HPM_LIB)"


def load_library():
    """dlopen libnrc_hpm.so (built by __graft_entry__.build() / csrc/Makefile).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("SkyRenderer ERROR: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the NRC path)" % LIB_PATH)
    # ONE HIP runtime per process: torch bundles its own libamdhip64.so (soname libamdhip64.so.7).  Import torch first so
    # that libnrc_hpm.so's NEEDED libamdhip64.so.7 binds to the copy torch already loaded -- streams and device pointers
    # handed over from torch are only meaningful inside the same runtime instance.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.nrc_last_error.restype = C.c_char_p
    L.nrc_version.restype = C.c_char_p
    L.nrc_cache_get_loss.restype = C.c_float
    L.nrc_cache_get_loss_blocking.restype = C.c_float
    L.nrc_cache_get_infer_batch_count.restype = C.c_size_t
    L.nrc_cache_get_train_batch_count.restype = C.c_size_t
    L.nrc_cache_get_infer_batch_size.restype = C.c_uint32
    L.nrc_cache_grid_list_capacity.restype = C.c_size_t
    L.nrc_cache_get_train_batch_size.restype = C.c_uint32
    L.nrc_cache_grad_ptr.restype = C.c_void_p
    L.nrc_cache_loss_ptr.restype = C.c_void_p
    L.nrc_cache_param_count.restype = C.c_uint32
    L.nrc_cache_param_count_tcnn.restype = C.c_uint32
    L.nrc_cache_param_count_tcnn.argtypes = [C.c_void_p]
    L.nrc_renderer_framebuffer.restype = C.c_void_p
    L.nrc_renderer_framebuffer_on.restype = C.c_void_p
    L.nrc_renderer_buffer.restype = C.c_void_p
    L.nrc_renderer_frame_time_ms.restype = C.c_float
    L.nrc_renderer_vertex_image_bytes.restype = C.c_size_t
    L.nrc_renderer_vertex_image_bytes.argtypes = [C.c_void_p]
    L.nrc_renderer_tile_order.restype = C.c_size_t
    L.nrc_renderer_tile_order.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.nrc_mc_renderer_framebuffer.restype = C.c_void_p
    L.nrc_mc_renderer_frame_time_ms.restype = C.c_float
    for name in ("nrc_cache_get_loss", "nrc_cache_get_loss_blocking", "nrc_renderer_is_blending", "nrc_mc_renderer_is_blending",
                 "nrc_cache_get_infer_batch_count", "nrc_cache_get_train_batch_count",
                 "nrc_cache_get_infer_batch_size", "nrc_cache_get_train_batch_size", "nrc_cache_grad_ptr",
                 "nrc_cache_loss_ptr", "nrc_cache_param_count", "nrc_renderer_framebuffer", "nrc_mc_renderer_framebuffer",
                 "nrc_mc_renderer_frame_time_ms"):
        getattr(L, name).argtypes = [C.c_void_p]
    _lib = L
    # schedules the tuner settled on in earlier processes (include/nrc_hpm.h, nrc_schedule_cache_load): the package's own table -- this pool's
    # MI355X, the BASELINE presets -- and the host's (NRC_SCHEDULE_CACHE=<file>; "" = none of either)
    if hasattr(L, "nrc_schedule_cache_load"):
        L.nrc_renderer_schedule_source.restype = C.c_char_p
        L.nrc_renderer_schedule_key.restype = C.c_char_p
        user = os.environ.get("NRC_SCHEDULE_CACHE")
        for path in ([] if user == "" else [os.path.join(_HERE, "schedules.txt")] + ([user] if user else [])):
            if os.path.exists(path):
                L.nrc_schedule_cache_load(path.encode(), None)
    return L


def load_schedule_cache(path):
    """nrc_schedule_cache_load: merge a file of tuned schedules into the process-wide table; returns the number of entries read"""
    n = C.c_int(0)
    _check(load_library().nrc_schedule_cache_load(path.encode(), C.byref(n)))
    return n.value


def clear_schedule_cache():
    _check(load_library().nrc_schedule_cache_clear())


def save_schedule_cache(path):
    """nrc_schedule_cache_save: write the process-wide table (what the tuners of this process settled on + what was loaded)"""
    n = C.c_int(0)
    _check(load_library().nrc_schedule_cache_save(path.encode(), C.byref(n)))
    return n.value


class CommError(RuntimeError):
    """NRC_ERR_COMM: a collective failed or a peer did not answer within the communicator's deadline; the communicator is aborted"""


def _check(status):
    if status != 0:
        msg = load_library().nrc_last_error().decode() or "SkyRenderer ERROR: status %d" % status
        raise (CommError if status == -4 else RuntimeError)(msg)


def _vp(x):
    return C.c_void_p(int(x) if x is not None else 0)


def _dev_ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream_ptr(stream):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream().cuda_stream
    return C.c_void_p(int(getattr(stream, "cuda_stream", stream)))


# ------------------------------------------------------------------------------------------------------------------
class AppConfig:
    """en::AppConfig.  AppConfig(argv) takes the reference's 18-entry argv (program name + 17 args,
    src/AppConfig.cpp:154-182); AppConfig() gives the defaults of src/main.cu:432-439 with the north-star
    encoding (posID 3 Frequency, dirID 0 OneBlob)."""

    def __init__(self, argv=None, **overrides):
        c = NrcConfig()
        load_library().nrc_config_default(C.byref(c))
        if argv is not None:
            if len(argv) != 18:
                raise RuntimeError("SkyRenderer ERROR: Argument count does not match requirements for AppConfig")
            a = [str(x) for x in argv[1:]]
            c.loss_fn, c.optimizer = a[0].encode(), a[1].encode()
            c.learning_rate, c.ema_decay = float(a[2]), float(a[3])
            c.pos_id, c.dir_id = int(a[4]), int(a[5])
            c.nn_width, c.nn_depth = int(a[6]), int(a[7])
            c.log2_infer_batch_size, c.log2_train_batch_size, c.train_batch_count = int(a[8]), int(a[9]), int(a[10])
            c.scene_id = int(a[11])
            c.train_ring_buf_size, c.train_spp = float(a[12]), int(a[13])
            c.primary_ray_length, c.primary_ray_prob, c.train_ray_length = int(a[14]), float(a[15]), int(a[16])
        for k, v in overrides.items():
            if not hasattr(c, k):
                raise AttributeError(k)
            setattr(c, k, v.encode() if isinstance(v, str) else v)
        self.c = c

    def __getattr__(self, k):
        v = getattr(self.__dict__["c"], k)
        return v.decode() if isinstance(v, bytes) else v

    def GetName(self):      # src/AppConfig.cpp:184-205 (std::to_string formatting)
        c = self.c
        f = "%.6f"
        parts = [c.loss_fn.decode(), c.optimizer.decode(), f % c.learning_rate, f % c.ema_decay, c.pos_id, c.dir_id,
                 c.nn_width, c.nn_depth, c.log2_infer_batch_size, c.log2_train_batch_size, c.train_batch_count,
                 c.scene_id, f % c.train_ring_buf_size, c.train_spp, c.primary_ray_length, f % c.primary_ray_prob,
                 c.train_ray_length]
        return "_".join(str(p) for p in parts)


def make_c_scene(scene):
    """dict from nrc_hpm_renderer_amd.scene.make_scene -> nrc_scene (host pointers; keeps the arrays alive)."""
    s = NrcScene()
    dens = np.ascontiguousarray(scene["density"], np.uint8)
    env = np.ascontiguousarray(scene["env"], np.float32)
    s.density = dens.ctypes.data
    s.nx, s.ny, s.nz = scene["dims"]
    s.size[:] = [float(x) for x in scene["size"]]
    s.density_factor = scene["density_factor"]
    s.g = scene["g"]
    s.dir_light_dir[:] = [float(x) for x in scene["dir_light_dir"]]
    s.dir_light_strength = scene["dir_light_strength"]
    s.point_light_pos[:] = [float(x) for x in scene["point_light_pos"]]
    s.point_light_strength = scene["point_light_strength"]
    s.point_light_color[:] = [float(x) for x in scene["point_light_color"]]
    s.env_strength = scene["env_strength"]
    s.env = env.ctypes.data
    s.env_h, s.env_w = env.shape[0], env.shape[1]
    s._keep = (dens, env)
    return s


def make_c_camera(cam):
    c = NrcCamera()
    c.inv_proj_view[:] = [float(x) for x in np.asarray(cam["inv_proj_view"], np.float32).reshape(16)]
    c.pos[:] = [float(x) for x in cam["pos"]]
    return c


def _wrap_device(ptr, nbytes, dtype, shape):
    """torch view over a device pointer owned by the library (no copy)."""
    import torch

    class _Arr:
        pass

    a = _Arr()
    a.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    t = torch.as_tensor(a, device="cuda")
    return t.view(dtype).view(*shape)


# ------------------------------------------------------------------------------------------------------------------
class NeuralRadianceCache:
    """en::NeuralRadianceCache (src/NeuralRadianceCache.cu)."""
    MASTER, EMA, ADAM_M, ADAM_V, GRAD = range(5)

    def __init__(self, appConfig):
        self.L = load_library()
        self.cfg = appConfig
        h = C.c_void_p()
        _check(self.L.nrc_cache_create(C.byref(appConfig.c), C.byref(h)))
        self.h = h
        self._hook_keep = None
        self._bufs = None

    def Init(self, inferCount, dInferInput, dInferOutput, dTrainInput, dTrainTarget, stream=None, cudaStartEvent=None,
             cudaFinishedEvent=None):
        """Buffers: torch CUDA float32 tensors [n,5] / [n,3] (caller-owned, as in the reference).  cudaStartEvent /
        cudaFinishedEvent: torch.cuda.Event objects standing in for the reference's two external semaphores
        (include/engine/graphics/NeuralRadianceCache.hpp:15-22): InferAndTrain waits for the first and records the second."""
        self._bufs = (dInferInput, dInferOutput, dTrainInput, dTrainTarget, cudaStartEvent, cudaFinishedEvent)
        if cudaStartEvent is None and cudaFinishedEvent is None:
            _check(self.L.nrc_cache_init(self.h, C.c_uint32(inferCount), _dev_ptr(dInferInput), _dev_ptr(dInferOutput),
                                         _dev_ptr(dTrainInput), _dev_ptr(dTrainTarget), _stream_ptr(stream)))
            return
        ev = [C.c_void_p(int(e.cuda_event)) if e is not None else None for e in (cudaStartEvent, cudaFinishedEvent)]
        _check(self.L.nrc_cache_init_events(self.h, C.c_uint32(inferCount), _dev_ptr(dInferInput), _dev_ptr(dInferOutput),
                                            _dev_ptr(dTrainInput), _dev_ptr(dTrainTarget), _stream_ptr(stream), ev[0], ev[1]))

    def InferAndTrain(self, inferFilter, train):
        f = None
        if inferFilter is not None:
            f = np.ascontiguousarray(inferFilter, np.uint32)
        _check(self.L.nrc_cache_infer_and_train(self.h, f.ctypes.data_as(C.c_void_p) if f is not None else None,
                                                C.c_int(int(bool(train)))))

    def Destroy(self):
        if self.h:
            _check(self.L.nrc_cache_destroy(self.h))
            self.h = None

    def GetLoss(self, wait=True):
        """en::NeuralRadianceCache::GetLoss(): the loss of the last training step enqueued (src/NeuralRadianceCache.cu:154; waits for
        that step only).  wait=False = GetLossAsync()[0]."""
        if wait:
            return float(self.L.nrc_cache_get_loss(self.h))
        return self.GetLossAsync()[0]

    def GetLossAsync(self):
        """never blocks: (loss of the most recent COMPLETED step, that step's number, number of the newest step enqueued) -- the
        per-frame poll of src/main.cu:303,376 without draining the frame pipeline"""
        loss, step, enq = C.c_float(0), C.c_uint32(0), C.c_uint32(0)
        _check(self.L.nrc_cache_get_loss_async(self.h, C.byref(loss), C.byref(step), C.byref(enq)))
        return float(loss.value), int(step.value), int(enq.value)

    def GetInferBatchCount(self):
        return int(self.L.nrc_cache_get_infer_batch_count(self.h))

    def GetTrainBatchCount(self):
        return int(self.L.nrc_cache_get_train_batch_count(self.h))

    def GetInferBatchSize(self):
        return int(self.L.nrc_cache_get_infer_batch_size(self.h))

    def GetTrainBatchSize(self):
        return int(self.L.nrc_cache_get_train_batch_size(self.h))

    # ---- finer-grained steps (multi-GPU driver, tests) ----
    def SetStream(self, stream=None):
        _check(self.L.nrc_cache_set_stream(self.h, _stream_ptr(stream)))

    def Infer(self, dInput, dOutput, useEma=True):
        _check(self.L.nrc_cache_infer(self.h, _dev_ptr(dInput), _dev_ptr(dOutput), C.c_uint32(dInput.shape[0]),
                                      C.c_int(int(useEma))))

    def Backward(self, dInput, dTarget, nNorm=0):
        _check(self.L.nrc_cache_backward(self.h, _dev_ptr(dInput), _dev_ptr(dTarget), C.c_uint32(dInput.shape[0]),
                                         C.c_uint32(nNorm)))

    def OptimizerStep(self):
        _check(self.L.nrc_cache_optimizer_step(self.h))

    def ParamCount(self):
        return int(self.L.nrc_cache_param_count(self.h))

    def GradTensor(self):
        import torch
        n = self.ParamCount()
        return _wrap_device(self.L.nrc_cache_grad_ptr(self.h), n * 4, torch.float32, (n,))

    def LossTensor(self):
        import torch
        return _wrap_device(self.L.nrc_cache_loss_ptr(self.h), 8, torch.float32, (2,))

    def CommInit(self, unique_id, rank, world):
        """native RCCL gradient exchange: unique_id = 128 bytes from comm_unique_id() of rank 0 (collective call)"""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        _check(self.L.nrc_cache_comm_init(self.h, buf, C.c_int(rank), C.c_int(world)))

    def CommInfo(self):
        """(rank, world) as the library's RCCL communicator reports them; world 0 = no native communicator"""
        r, w = C.c_int(0), C.c_int(0)
        _check(self.L.nrc_cache_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def CommTimeExchange(self, reps=100):
        """average microseconds of one gradient all-reduce of the native exchange (collective call; 0.0 without a communicator)"""
        us = C.c_float(0)
        _check(self.L.nrc_cache_comm_time_exchange(self.h, C.c_uint32(reps), C.byref(us)))
        return float(us.value)

    def CommSparse(self):
        """True when the HashGrid table gradient travels as (entry, value) lists (native exchange of a posID 0 model)"""
        return bool(self.L.nrc_cache_comm_sparse(self.h))

    def GridGradPack(self):
        """(debug) the table gradient of the last Backward as the exchange list: uint32 words {count, 0, (entry, half2 bits) x
        capacity}, padding entries 0xffffffff"""
        cap = int(self.L.nrc_cache_grid_list_capacity(self.h))
        out = np.zeros(2 + 2 * cap, np.uint32)
        _check(self.L.nrc_cache_grid_grad_pack(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)))
        return out

    def GridGradApply(self, lists):
        """(debug) gradient vector's table part := sum of the lists (GridGradPack layout), added in the order given"""
        v = np.ascontiguousarray(np.stack(lists), np.uint32)
        _check(self.L.nrc_cache_grid_grad_apply(self.h, v.ctypes.data_as(C.c_void_p), C.c_uint32(v.shape[0])))

    def SetExchangeDtype(self, dtype):
        """nrc_cache_set_exchange_dtype: "f32" (default) or "f16" -- the gradients summed over the ranks as fp16 numbers pre-scaled by
        loss_scale 128 (BASELINE.json configs[3]); every rank alike"""
        _check(self.L.nrc_cache_set_exchange_dtype(self.h, C.c_int({"f32": 0, "f16": 1}[dtype])))

    def GetExchangeDtype(self):
        return "f16" if self.L.nrc_cache_get_exchange_dtype(self.h) == 1 else "f32"

    def SetLossNormFactor(self, factor):
        _check(self.L.nrc_cache_set_loss_norm_factor(self.h, C.c_uint32(factor)))

    def SetGradHook(self, fn):
        """fn(grad_tensor, loss_tensor) is called between backward and the optimizer of every train batch, with torch's
        current stream switched to the stream the training kernels are ordered on."""
        if fn is None:
            self._hook_keep = None
            _check(self.L.nrc_cache_set_grad_hook(self.h, None, None))
            return
        g, lo = self.GradTensor(), self.LossTensor()

        import torch

        def tramp(_user, _g, _n, _l, stream):
            s = torch.cuda.ExternalStream(int(stream or 0))
            with torch.cuda.stream(s):
                fn(g, lo)
                # What fn enqueued through torch must be COMPLETE when the library launches its next kernel on the stream.  Measured (round 6,
                # tools/_build/bisect_flaky.sh): an in-place torch kernel launched here on this very stream overlapped the library's next
                # kernel on it -- 300..4 700 of 25 792 gradient words kept their old value -- whenever tests/test_gpu_frame_graph.py had run
                # in the same process before (3 of 3), never in a fresh process whatever the number of streams or hardware queues
                # (tools/_build/hook_overlap_probe.py: 0 of 180); round 4 met the same between torch and the NULL stream on a warm GPU.
                # A stream synchronisation here ends it; the collective hooks below synchronise for the same reason.  (The library's
                # own RCCL path has no hook, and two of its own kernels on one stream have never been seen out of order: every bitwise test.)
                s.synchronize()

        self._hook_keep = GRAD_HOOK(tramp)
        _check(self.L.nrc_cache_set_grad_hook(self.h, self._hook_keep, None))

    def CommStatus(self):
        """nrc_cache_comm_status: raises CommError once the exchange has failed (polls RCCL's asynchronous error state; never blocks)"""
        _check(self.L.nrc_cache_comm_status(self.h))

    def SetCommTimeoutMs(self, ms):
        """deadline of the library's waits behind a collective (default 30 000 ms once the frame has more than one rank; 0: none)"""
        _check(self.L.nrc_cache_set_comm_timeout_ms(self.h, C.c_uint32(int(ms))))

    def SetRawCollectiveHooks(self, rank, world, allreduce, allgather):
        """nrc_cache_set_collective_hooks with the caller's own ALLREDUCE_F64_HOOK / ALLGATHER_HOOK callables (tests: failing transports)"""
        self._coll_keep = (ALLREDUCE_F64_HOOK(allreduce), ALLGATHER_HOOK(allgather))
        _check(self.L.nrc_cache_set_collective_hooks(self.h, C.c_int(rank), C.c_int(world), self._coll_keep[0], self._coll_keep[1], None))

    def SetCollectiveHooks(self, rank, world, group=None):
        """the frame gather / metric reduction of a cache WITHOUT a native RCCL communicator go through torch.distributed (any backend:
        the gloo rehearsals of tests/ stage through host memory).  nrc_cache_set_collective_hooks."""
        import torch
        import torch.distributed as dist

        def on(stream):
            return torch.cuda.stream(torch.cuda.ExternalStream(int(stream or 0)))

        def staged(t):
            if dist.get_backend(group) == "nccl":
                return t
            return t.cpu()      # host-staged transport (the library has waited for its own work on the stream: include/nrc_hpm.h)

        def allreduce(_user, buf, n, stream):
            try:
                with on(stream):
                    t = _wrap_device(buf, int(n) * 8, torch.float64, (int(n),))
                    h = staged(t)
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                    if h is not t:
                        t.copy_(h)
                        torch.cuda.current_stream().synchronize()
                return 0
            except Exception as e:      # noqa: BLE001 -- a Python exception must not unwind through the C frames
                print("collective hook (all-reduce) failed: %r" % (e,))
                return 1

        def allgather(_user, send, recv, nbytes, stream):
            try:
                with on(stream):
                    s_ = _wrap_device(send, int(nbytes), torch.uint8, (int(nbytes),))
                    r_ = _wrap_device(recv, int(nbytes) * world, torch.uint8, (world, int(nbytes)))
                    if dist.get_backend(group) == "nccl":
                        dist.all_gather_into_tensor(r_.view(-1), s_, group=group)
                    else:
                        parts = [torch.empty(int(nbytes), dtype=torch.uint8) for _ in range(world)]
                        hs = staged(s_)
                        dist.all_gather(parts, hs, group=group)
                        r_.copy_(torch.stack(parts))
                        torch.cuda.current_stream().synchronize()
                return 0
            except Exception as e:      # noqa: BLE001
                print("collective hook (all-gather) failed: %r" % (e,))
                return 1

        self._coll_keep = (ALLREDUCE_F64_HOOK(allreduce), ALLGATHER_HOOK(allgather))
        _check(self.L.nrc_cache_set_collective_hooks(self.h, C.c_int(rank), C.c_int(world), self._coll_keep[0], self._coll_keep[1], None))

    def GetParams(self, which=0):
        out = np.zeros(self.ParamCount(), np.float32)
        _check(self.L.nrc_cache_get_params(self.h, C.c_int(which), out.ctypes.data_as(C.c_void_p)))
        return out

    def SetParams(self, which, values):
        v = np.ascontiguousarray(values, np.float32)
        assert v.size == self.ParamCount()
        _check(self.L.nrc_cache_set_params(self.h, C.c_int(which), v.ctypes.data_as(C.c_void_p)))

    def ParamCountTcnn(self):
        return int(self.L.nrc_cache_param_count_tcnn(self.h))

    def GetParamsTcnn(self, which=0):
        """buffer `which` in tiny-cuda-nn's own layout (output matrix padded to 16 rows): what a weight dump of the reference holds"""
        out = np.zeros(self.ParamCountTcnn(), np.float32)
        _check(self.L.nrc_cache_get_params_tcnn(self.h, C.c_int(which), out.ctypes.data_as(C.c_void_p)))
        return out

    def SetParamsTcnn(self, which, values):
        v = np.ascontiguousarray(values, np.float32)
        assert v.size == self.ParamCountTcnn()
        _check(self.L.nrc_cache_set_params_tcnn(self.h, C.c_int(which), v.ctypes.data_as(C.c_void_p)))

    def SaveCheckpoint(self, path):
        """nrc_cache_save_checkpoint: model shape + step + weights / EMA weights / Adam moments (tiny-cuda-nn layout) in one file"""
        _check(self.L.nrc_cache_save_checkpoint(self.h, os.fsencode(path)))

    def LoadCheckpoint(self, path):
        _check(self.L.nrc_cache_load_checkpoint(self.h, os.fsencode(path)))

    def GetStep(self):
        s = C.c_uint32(0)
        _check(self.L.nrc_cache_get_step(self.h, C.byref(s)))
        return s.value

    def SetStep(self, step):
        _check(self.L.nrc_cache_set_step(self.h, C.c_uint32(step)))

    def state_dict(self):
        """checkpoint (the reference has none): fp32 master/EMA weights + Adam state + step"""
        return dict(step=self.GetStep(), **{k: self.GetParams(i) for i, k in enumerate(("w", "ema", "m", "v"))})

    def load_state_dict(self, sd):
        for i, k in enumerate(("w", "ema", "m", "v")):
            self.SetParams(i, sd[k])
        self.SetStep(int(sd["step"]))

    def __del__(self):
        try:
            self.Destroy()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------------
class NrcHpmRenderer:
    """en::NrcHpmRenderer (src/NrcHpmRenderer.cu).  `queue` arguments of the reference become the HIP stream."""
    BUF = dict(primary=0, info=1, origin=2, dir=3, infer_input=4, infer_output=5, train_input=6, train_target=7, ring=8)

    def __init__(self, width, height, blend, camera, appConfig, hpmScene, nrc, tile=None, stream=None):
        self.L = load_library()
        self.width, self.height = width, height
        self.nrc = nrc
        self._scene = make_c_scene(hpmScene)
        cam = make_c_camera(camera)
        t = None
        self.tile = tuple(int(x) for x in tile) if tile is not None else None
        if tile is not None:
            t = NrcTile(*[int(x) for x in tile])
        h = C.c_void_p()
        _check(self.L.nrc_renderer_create(C.c_uint32(width), C.c_uint32(height), C.c_int(int(blend)), C.byref(cam),
                                          C.byref(appConfig.c), C.byref(self._scene), nrc.h,
                                          C.byref(t) if t is not None else None, _stream_ptr(stream), C.byref(h)))
        self.h = h

    def Render(self, queue=None, train=False):
        _check(self.L.nrc_renderer_render(self.h, C.c_int(int(bool(train)))))

    def RenderFrames(self, frameRandoms, train=False):
        """len(frameRandoms) consecutive Render(queue, train) calls enqueued by one call into the library; frameRandoms: [n][4]"""
        r = np.ascontiguousarray(frameRandoms, np.float32).reshape(-1, 4)
        if not hasattr(self.L, "nrc_renderer_render_frames"):      # an older build loaded through NRC_HPM_LIB (A/B tooling)
            for row in r:
                self.SetFrameRandom(row)
                self.Render(None, train)
            return
        _check(self.L.nrc_renderer_render_frames(self.h, C.c_uint32(r.shape[0]), r.ctypes.data_as(C.c_void_p), C.c_int(int(bool(train)))))

    def SetCamera(self, queue, camera):
        cam = make_c_camera(camera)
        _check(self.L.nrc_renderer_set_camera(self.h, C.byref(cam)))

    def SetBlend(self, blend):
        _check(self.L.nrc_renderer_set_blend(self.h, C.c_int(int(blend))))

    def SetSceneParams(self, scene):
        """lights / medium constants of `scene` (dict of scene.make_scene or scene.HpmScene); textures are not re-uploaded"""
        sc_ = make_c_scene(scene.scene if hasattr(scene, "scene") else scene)
        _check(self.L.nrc_renderer_set_scene_params(self.h, C.byref(sc_)))

    def SetShowNrc(self, show):
        _check(self.L.nrc_renderer_set_show_nrc(self.h, C.c_int(int(show))))

    def SetFrameRandom(self, r4):
        r = (C.c_float * 4)(*[float(x) for x in r4])
        _check(self.L.nrc_renderer_set_frame_random(self.h, r))

    def ExportOutputImageToFile(self, queue, filePath, root=0):
        """src/NrcHpmRenderer.cu:437-493.  A sharded renderer (tile of a multi-GPU frame) exports the WHOLE frame: collective over the
        frame's ranks, rank `root` writes the file"""
        if self.tile is not None and self.tile[1] > 1:
            _check(self.L.nrc_renderer_export_exr_gathered(self.h, filePath.encode(), C.c_int(root)))
        else:
            _check(self.L.nrc_renderer_export_exr(self.h, filePath.encode()))

    def SetSchedule(self, camera_priority_low=-1, cost_order_lag=-1, xcd_window=-1, composite_defer=-1):
        """nrc_renderer_set_schedule: pin scheduling knobs (values >= 0) or hand them back to the library's tuner (-1); no knob changes a pixel"""
        v = (C.c_int32 * 4)(int(camera_priority_low), int(cost_order_lag), int(xcd_window), int(composite_defer))
        _check(self.L.nrc_renderer_set_schedule(self.h, v))

    def GetSchedule(self):
        """the schedule in use now: dict(camera_priority_low, cost_order_lag, xcd_window, composite_defer, tuning_done)"""
        v = (C.c_int32 * 4)()
        done = C.c_int(0)
        _check(self.L.nrc_renderer_get_schedule(self.h, v, C.byref(done)))
        d = dict(camera_priority_low=v[0], cost_order_lag=v[1], xcd_window=v[2], composite_defer=v[3], tuning_done=bool(done.value))
        if hasattr(self.L, "nrc_renderer_schedule_source"):
            d["source"] = (self.L.nrc_renderer_schedule_source(self.h) or b"").decode()
            d["key"] = (self.L.nrc_renderer_schedule_key(self.h) or b"").decode()
        return d

    def GatherFrame(self, stream=None):
        """the whole [global_h, global_w, 4] frame of a sharded renderer as a new torch CUDA tensor, on every rank (collective: one
        all-gather through the cache's communicator + a de-interleave kernel); an unsharded renderer returns a copy of its image"""
        import torch
        gw, gh = (self.tile[2], self.tile[3]) if self.tile is not None else (self.width, self.height)
        out = torch.empty((gh, gw, 4), device="cuda", dtype=torch.float32)
        _check(self.L.nrc_renderer_gather_frame(self.h, _dev_ptr(out), _stream_ptr(stream)))
        return out

    def GetFrameTimeMS(self):
        return float(self.L.nrc_renderer_frame_time_ms(self.h, None))

    def EvaluateTimestampQueries(self):
        st = (C.c_float * 8)()
        self.L.nrc_renderer_frame_time_ms(self.h, st)
        names = ("clear", "gen_rays", "prep_infer", "train", "prep_train", "infer", "render", "total")
        return dict(zip(names, [float(x) for x in st]))

    def StageStats(self, reset=True):
        """average per-stage ms over all frames since the last reset (HIP events on the render stream)"""
        st = (C.c_float * 8)()
        n = C.c_uint32(0)
        _check(self.L.nrc_renderer_stage_stats(self.h, st, C.byref(n), C.c_int(int(reset))))
        names = ("clear", "gen_rays", "prep_infer", "train", "prep_train", "infer", "render", "total")
        d = dict(zip(names, [float(x) for x in st]))
        d["frames"] = n.value
        return d

    def FrameTimeline(self, max_frames=4096):
        """[frames, 6] ms from the first frame's start (since the last reset) to each frame's events: gen_rays start / done, train rays
        done, inference done, compositing done, training done (nrc_renderer_frame_timeline)"""
        import numpy as np
        buf = (C.c_float * (6 * max_frames))()
        n = C.c_uint32(0)
        _check(self.L.nrc_renderer_frame_timeline(self.h, buf, C.c_uint32(max_frames), C.byref(n)))
        return np.ctypeslib.as_array(buf)[:6 * n.value].reshape(n.value, 6).copy()

    def ResetStageStats(self):
        """forget the frames rendered so far without reading their events (StageStats(reset=True) reads every one of them first)"""
        _check(self.L.nrc_renderer_stage_stats(self.h, None, None, C.c_int(1)))

    def GetImage(self, stream=None):
        """RGBA32F framebuffer as a torch CUDA tensor view [height, width, 4].  The stream given at construction (default) or
        `stream` (a torch stream / raw handle of the consumer) is ordered behind the frame's compositing."""
        import torch
        if stream is None:
            p = self.L.nrc_renderer_framebuffer(self.h)
        else:
            p = self.L.nrc_renderer_framebuffer_on(self.h, _stream_ptr(stream))
        return _wrap_device(p, self.width * self.height * 16, torch.float32, (self.height, self.width, 4))

    def ReleaseImage(self, stream):
        """the consumer's reads of GetImage(stream) enqueued so far end here; the next frame's compositing waits for them"""
        _check(self.L.nrc_renderer_release_frame(self.h, _stream_ptr(stream)))

    def IsBlending(self):
        return bool(self.L.nrc_renderer_is_blending(self.h))

    def SetStageEvents(self, on=True):
        """per-stage timing events (EvaluateTimestampQueries / StageStats) of every stage (default) or of gen_rays only"""
        _check(self.L.nrc_renderer_set_stage_events(self.h, C.c_int(int(on))))

    def SetEmptySkip(self, on=True):
        """exact empty-space early-out of camera rays (default on); off = every ray is traced"""
        _check(self.L.nrc_renderer_set_empty_skip(self.h, C.c_int(int(on))))

    def SetCostOrder(self, on=True):
        """costliest-first launch order of gen_rays' tiles from earlier frames' per-tile times (default on); frames do not change"""
        _check(self.L.nrc_renderer_set_cost_order(self.h, C.c_int(int(on))))

    def SetHotTiles(self, on=True):
        """start the tiles with a pixel in a capped RNG state first (default on); frames do not change"""
        _check(self.L.nrc_renderer_set_hot_tiles(self.h, C.c_int(int(on))))

    def HotTiles(self):
        """(tiles [(tx, ty), ...] gen_rays started first in the last frame, total count found, computed one frame ahead?) or None"""
        import numpy as np
        out = np.zeros(9, np.uint32)
        rc = self.L.nrc_renderer_hot_tiles(self.h, out.ctypes.data_as(C.c_void_p))
        if rc == -2:
            raise RuntimeError(self.L.nrc_last_error().decode() or "nrc_renderer_hot_tiles failed")
        if rc < 0:
            return None
        n = int(out[8])
        return [(int(t & 0xffff), int(t >> 16)) for t in out[:min(n, 8)]], n, rc == 1

    def TileOrder(self):
        """the tile permutation the next frame launches in (numpy uint32)"""
        import numpy as np
        n = self.L.nrc_renderer_tile_order(self.h, None, C.c_size_t(0))
        out = np.zeros(n, np.uint32)
        got = self.L.nrc_renderer_tile_order(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        if got != n:
            raise RuntimeError(self.L.nrc_last_error().decode() or "nrc_renderer_tile_order failed")
        return out

    def SetFullVertexImages(self, on=True):
        """store nrcRayOrigin / nrcRayDir for every pixel (the reference's images) instead of the train grid's pixels only"""
        _check(self.L.nrc_renderer_set_full_vertex_images(self.h, C.c_int(int(on))))

    def VertexImageBytes(self):
        return int(self.L.nrc_renderer_vertex_image_bytes(self.h))

    def Buffer(self, name):
        import torch
        nbytes = C.c_size_t(0)
        p = self.L.nrc_renderer_buffer(self.h, C.c_int(self.BUF[name]), C.byref(nbytes))
        if not p:
            _check(-1)
        if name == "ring":
            return _wrap_device(p, nbytes.value, torch.int32, (nbytes.value // 4,))
        inner = {"primary": 4, "info": 1, "origin": 4, "dir": 4, "infer_input": 5, "infer_output": 3,
                 "train_input": 5, "train_target": 3}[name]
        return _wrap_device(p, nbytes.value, torch.float32, (nbytes.value // (4 * inner), inner))

    def TrainGrid(self):
        o = (C.c_uint32 * 5)()
        _check(self.L.nrc_renderer_train_grid(self.h, o))
        return dict(tw=o[0], th=o[1], x_dist=o[2], y_dist=o[3], ring_size=o[4])

    def CountFetches(self, enable=True):
        v = C.c_ulonglong(0)
        _check(self.L.nrc_renderer_count_fetches(self.h, C.c_int(int(enable)), C.byref(v)))
        return v.value

    def Destroy(self):
        if self.h:
            _check(self.L.nrc_renderer_destroy(self.h))
            self.h = None

    def __del__(self):
        try:
            self.Destroy()
        except Exception:
            pass


class McHpmRenderer:
    """en::McHpmRenderer (src/McHpmRenderer.cpp)."""

    def __init__(self, width, height, pathLength, blend, camera, scene, tile=None, stream=None):
        self.L = load_library()
        self.width, self.height = width, height
        self._scene = make_c_scene(scene)
        cam = make_c_camera(camera)
        t = NrcTile(*[int(x) for x in tile]) if tile is not None else None
        h = C.c_void_p()
        _check(self.L.nrc_mc_renderer_create(C.c_uint32(width), C.c_uint32(height), C.c_uint32(pathLength),
                                             C.c_int(int(blend)), C.byref(cam), C.byref(self._scene),
                                             C.byref(t) if t is not None else None, _stream_ptr(stream), C.byref(h)))
        self.h = h

    def Render(self, queue=None):
        _check(self.L.nrc_mc_renderer_render(self.h))

    def SetCamera(self, queue, camera):
        cam = make_c_camera(camera)
        _check(self.L.nrc_mc_renderer_set_camera(self.h, C.byref(cam)))

    def SetBlend(self, blend):
        _check(self.L.nrc_mc_renderer_set_blend(self.h, C.c_int(int(blend))))

    def IsBlending(self):
        return bool(self.L.nrc_mc_renderer_is_blending(self.h))

    def SetEmptySkip(self, on=True):
        _check(self.L.nrc_mc_renderer_set_empty_skip(self.h, C.c_int(int(on))))

    def SetCostOrder(self, on=True):
        _check(self.L.nrc_mc_renderer_set_cost_order(self.h, C.c_int(int(on))))

    def SetSceneParams(self, scene):
        sc_ = make_c_scene(scene.scene if hasattr(scene, "scene") else scene)
        _check(self.L.nrc_mc_renderer_set_scene_params(self.h, C.byref(sc_)))

    def SetFrameRandom(self, r4):
        r = (C.c_float * 4)(*[float(x) for x in r4])
        _check(self.L.nrc_mc_renderer_set_frame_random(self.h, r))

    def ExportOutputImageToFile(self, queue, filePath):
        _check(self.L.nrc_mc_renderer_export_exr(self.h, filePath.encode()))

    def GetFrameTimeMS(self):
        return float(self.L.nrc_mc_renderer_frame_time_ms(self.h))

    def GetImage(self):
        import torch
        p = self.L.nrc_mc_renderer_framebuffer(self.h)
        return _wrap_device(p, self.width * self.height * 16, torch.float32, (self.height, self.width, 4))

    def CountFetches(self, enable=True):
        v = C.c_ulonglong(0)
        _check(self.L.nrc_mc_renderer_count_fetches(self.h, C.c_int(int(enable)), C.byref(v)))
        return v.value

    def Destroy(self):
        if self.h:
            _check(self.L.nrc_mc_renderer_destroy(self.h))
            self.h = None

    def __del__(self):
        try:
            self.Destroy()
        except Exception:
            pass


def comm_unique_id():
    buf = (C.c_char * 128)()
    _check(load_library().nrc_comm_unique_id(buf))
    return bytes(buf)


def CompareImages(ref, own, stream=None):
    """Reference::Result (include/engine/graphics/Reference.hpp:17-29): torch CUDA RGBA32F [h,w,4] images."""
    r = (C.c_float * 5)()
    h, w = ref.shape[0], ref.shape[1]
    _check(load_library().nrc_compare_images(_dev_ptr(ref), _dev_ptr(own), C.c_uint32(w), C.c_uint32(h),
                                             _stream_ptr(stream), r))
    return dict(mse=r[0], ref_mean=r[1], own_mean=r[2], own_var=r[3], valid=r[4])


def CompareImagesSharded(nrc, ref_local, own_local, stream=None):
    """Reference::Result of a frame sharded over ranks: every rank passes ITS pixels (torch CUDA RGBA32F [h, local_w, 4]) and receives
    the whole frame's metrics -- five local fp64 sums, two small all-reduces through `nrc`'s communicator (collective call)"""
    r = (C.c_float * 5)()
    n = int(ref_local.shape[0]) * int(ref_local.shape[1])
    assert tuple(ref_local.shape) == tuple(own_local.shape) and ref_local.is_contiguous() and own_local.is_contiguous()
    _check(load_library().nrc_compare_images_sharded(nrc.h, _dev_ptr(ref_local), _dev_ptr(own_local), C.c_uint32(n), _stream_ptr(stream), r))
    return dict(mse=r[0], ref_mean=r[1], own_mean=r[2], own_var=r[3], valid=r[4])


def check_guards():
    """NRC_DEBUG=guard_alloc: (number of allocations whose canaries were overwritten, description of the first); (-1, "") when off"""
    buf = C.create_string_buffer(512)
    n = load_library().nrc_debug_check_guards(buf, C.c_size_t(512))
    return int(n), buf.value.decode()


def test_math(fn, a, b=None):
    """bit-parity hook: evaluates math-spec function `fn` on the device (torch CUDA float tensors)."""
    import torch
    out, out2 = torch.empty_like(a), torch.empty_like(a)
    _check(load_library().nrc_test_math(C.c_int(fn), _dev_ptr(a), _dev_ptr(b) if b is not None else None,
                                        C.c_uint32(a.numel()), _dev_ptr(out), _dev_ptr(out2), _stream_ptr(None)))
    return out, out2


def test_rng(u, v, frame_random, n):
    import torch
    out = torch.zeros(n + 1, device="cuda")
    fr = (C.c_float * 4)(*[float(x) for x in frame_random])
    _check(load_library().nrc_test_rng(C.c_float(u), C.c_float(v), fr, C.c_uint32(n), _dev_ptr(out), _stream_ptr(None)))
    return out
