"""CPU oracle bindings -- TEST INFRASTRUCTURE ONLY.

ctypes view of oracle/nrc_oracle.h.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (nrc-hpm-renderer_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


class OrcScene(C.Structure):
    _fields_ = [
        ("density", C.c_void_p), ("nx", C.c_uint32), ("ny", C.c_uint32), ("nz", C.c_uint32),
        ("size", C.c_float * 3), ("density_factor", C.c_float), ("g", C.c_float),
        ("dir_light_dir", C.c_float * 3), ("dir_light_strength", C.c_float),
        ("point_light_pos", C.c_float * 3), ("point_light_strength", C.c_float),
        ("point_light_color", C.c_float * 3), ("env_strength", C.c_float),
        ("env", C.c_void_p), ("env_w", C.c_uint32), ("env_h", C.c_uint32),
    ]


class OrcCamera(C.Structure):
    _fields_ = [("inv_proj_view", C.c_float * 16), ("pos", C.c_float * 3)]


class OrcNNConfig(C.Structure):
    _fields_ = [
        ("pos_id", C.c_uint32), ("dir_id", C.c_uint32), ("width", C.c_uint32), ("depth", C.c_uint32),
        ("loss_id", C.c_uint32), ("learning_rate", C.c_float), ("ema_decay", C.c_float), ("seed", C.c_uint32),
        ("hashgrid_log2_size", C.c_uint32),
        ("optimizer_id", C.c_uint32),
    ]


def build(native=False, out_dir=None):
    """Compile the oracle with g++ (recipe = oracle/Makefile). Returns the .so path."""
    out_dir = out_dir or os.path.join(_DIR, "_build")
    os.makedirs(out_dir, exist_ok=True)
    if os.environ.get("NRC_ORACLE_ASAN") == "1":      # tests/test_oracle_asan.py: the -fsanitize=address,undefined build (make asan)
        subprocess.check_call(["make", "-C", _DIR, "asan"], stdout=subprocess.DEVNULL)
        return os.path.join(_DIR, "_build", "libnrc_oracle_asan.so")
    name = "libnrc_oracle_native.so" if native else "libnrc_oracle.so"
    so = os.path.join(out_dir, name)
    src = os.path.join(_DIR, "nrc_oracle.cpp")
    deps = [src, os.path.join(_DIR, "nrc_oracle.h"), os.path.join(_DIR, "orc_math.h")]
    if os.path.exists(so) and not native and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
        return so
    opt = ["-O3", "-march=native"] if native else ["-O2"]
    cmd = ["g++"] + opt + ["-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-pthread",
                           "-shared", "-o", so, src]
    subprocess.check_call(cmd)
    return so


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, native=False, out_dir=None):
        self.lib = C.CDLL(build(native, out_dir))
        L = self.lib
        L.orc_hash.restype = C.c_uint32
        L.orc_hash.argtypes = [C.c_uint32]
        L.orc_random1.restype = C.c_float
        L.orc_random1.argtypes = [C.c_float]
        L.orc_nn_create.restype = C.c_void_p
        L.orc_nn_create.argtypes = [C.POINTER(OrcNNConfig)]
        L.orc_nn_destroy.argtypes = [C.c_void_p]
        L.orc_nn_param_count.restype = C.c_uint32
        L.orc_nn_param_count.argtypes = [C.c_void_p]
        L.orc_nn_encoded_dims.restype = C.c_uint32
        L.orc_nn_encoded_dims.argtypes = [C.c_void_p]
        L.orc_nn_mlp_param_count.restype = C.c_uint32
        L.orc_nn_mlp_param_count.argtypes = [C.c_void_p]
        L.orc_nn_buffer.restype = C.POINTER(C.c_float)
        L.orc_nn_buffer.argtypes = [C.c_void_p, C.c_int]
        L.orc_nn_set_step.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_nn_backward.restype = C.c_float

    # ---- RNG / math ----
    def hash(self, x):
        return int(self.lib.orc_hash(C.c_uint32(x)))

    def random1(self, x):
        return float(self.lib.orc_random1(C.c_float(x)))

    def rng_kat(self, u, v, frame_random, n):
        out = np.zeros(n + 1, np.float32)
        fr = np.asarray(frame_random, np.float32)
        self.lib.orc_rng_kat(C.c_float(u), C.c_float(v), _ptr(fr), C.c_int(n), _ptr(out))
        return out

    def math_eval(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
        out = np.zeros_like(a)
        out2 = np.zeros_like(a)
        self.lib.orc_math_eval(C.c_int(fn), _ptr(a), _ptr(b), C.c_int(a.size), _ptr(out), _ptr(out2))
        return out, out2

    # ---- scene marshalling ----
    @staticmethod
    def make_scene(scene):
        """scene: dict produced by nrc_hpm_renderer_amd.scene.make_scene (numpy arrays + floats)."""
        s = OrcScene()
        dens = np.ascontiguousarray(scene["density"], np.uint8)
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
        env = np.ascontiguousarray(scene["env"], np.float32)
        s.env = env.ctypes.data
        s.env_h, s.env_w = env.shape[0], env.shape[1]
        s._keep = (dens, env)
        return s

    @staticmethod
    def make_camera(cam):
        c = OrcCamera()
        c.inv_proj_view[:] = [float(x) for x in np.asarray(cam["inv_proj_view"], np.float32).reshape(16)]
        c.pos[:] = [float(x) for x in cam["pos"]]
        return c

    # ---- integrator ----
    def mc_render(self, scene, cam, W, H, path_length, frame_random, blend=1.0, out=None, rows=None, threads=1):
        s, c = self.make_scene(scene), self.make_camera(cam)
        out = np.zeros((H, W, 4), np.float32) if out is None else out
        info = np.zeros((H, W), np.float32)
        y0, y1 = rows if rows else (0, H)
        fr = np.asarray(frame_random, np.float32)
        nf = C.c_uint64(0)
        self.lib.orc_mc_render(C.byref(s), C.byref(c), C.c_uint32(W), C.c_uint32(H), C.c_uint32(y0), C.c_uint32(y1),
                               C.c_uint32(path_length), _ptr(fr), C.c_float(blend), _ptr(out), _ptr(info),
                               C.c_int(threads), C.byref(nf))
        return out, info, nf.value

    def nrc_gen_rays(self, scene, cam, W, H, primary_ray_length, primary_ray_prob, frame_random, rows=None, threads=1):
        s, c = self.make_scene(scene), self.make_camera(cam)
        primary = np.zeros((H, W, 4), np.float32)
        info = np.zeros((H, W), np.float32)
        origin = np.zeros((H, W, 4), np.float32)
        direc = np.zeros((H, W, 4), np.float32)
        infer_in = np.zeros((W * H, 5), np.float32)
        y0, y1 = rows if rows else (0, H)
        fr = np.asarray(frame_random, np.float32)
        nf = C.c_uint64(0)
        self.lib.orc_nrc_gen_rays(C.byref(s), C.byref(c), C.c_uint32(W), C.c_uint32(H), C.c_uint32(y0), C.c_uint32(y1),
                                  C.c_uint32(primary_ray_length), C.c_float(primary_ray_prob), _ptr(fr),
                                  _ptr(primary), _ptr(info), _ptr(origin), _ptr(direc), _ptr(infer_in),
                                  C.c_int(threads), C.byref(nf))
        return dict(primary=primary, info=info, origin=origin, dir=direc, infer_input=infer_in, n_fetch=nf.value)

    def nrc_walk_lengths(self, scene, cam, W, H, primary_ray_length, primary_ray_prob, frame_random, walks_per_pixel=8, threads=1):
        """[H][W][walks_per_pixel] uint16: free flights drawn by each tracking walk of a pixel of the NRC frame, in program order"""
        s, c = self.make_scene(scene), self.make_camera(cam)
        primary = np.zeros((H, W, 4), np.float32)
        info = np.zeros((H, W), np.float32)
        origin = np.zeros((H, W, 4), np.float32)
        direc = np.zeros((H, W, 4), np.float32)
        out = np.zeros((H, W, walks_per_pixel), np.uint16)
        fr = np.asarray(frame_random, np.float32)
        self.lib.orc_nrc_walk_lengths(C.byref(s), C.byref(c), C.c_uint32(W), C.c_uint32(H), C.c_uint32(0), C.c_uint32(H),
                                      C.c_uint32(primary_ray_length), C.c_float(primary_ray_prob), _ptr(fr), _ptr(primary), _ptr(info),
                                      _ptr(origin), _ptr(direc), C.c_int(threads), out.ctypes.data_as(C.c_void_p), C.c_uint32(walks_per_pixel))
        return out, info

    def nrc_prep_train(self, scene, W, H, TW, TH, x_dist, y_dist, train_spp, train_ray_length, ring_size,
                       frame_random, info, origin, direc, head_tail, ring, threads=1):
        s = self.make_scene(scene)
        tin = np.zeros((TW * TH, 5), np.float32)
        tgt = np.zeros((TW * TH, 3), np.float32)
        fr = np.asarray(frame_random, np.float32)
        self.lib.orc_nrc_prep_train(C.byref(s), C.c_uint32(W), C.c_uint32(H), C.c_uint32(TW), C.c_uint32(TH),
                                    C.c_uint32(x_dist), C.c_uint32(y_dist), C.c_uint32(train_spp),
                                    C.c_uint32(train_ray_length), C.c_uint32(ring_size), _ptr(fr),
                                    _ptr(info), _ptr(origin), _ptr(direc), _ptr(head_tail), _ptr(ring),
                                    _ptr(tin), _ptr(tgt), C.c_int(threads))
        return tin, tgt

    def nrc_composite(self, W, H, show_nrc, blend, primary, info, infer_out, out):
        self.lib.orc_nrc_composite(C.c_uint32(W), C.c_uint32(H), C.c_uint32(show_nrc), C.c_float(blend),
                                   _ptr(primary), _ptr(info), _ptr(infer_out), _ptr(out))
        return out

    def compare(self, ref, own):
        H, W = ref.shape[:2]
        r = np.zeros(5, np.float32)
        self.lib.orc_compare(_ptr(np.ascontiguousarray(ref, np.float32)), _ptr(np.ascontiguousarray(own, np.float32)),
                             C.c_uint32(W), C.c_uint32(H), _ptr(r))
        return dict(mse=float(r[0]), ref_mean=float(r[1]), own_mean=float(r[2]), own_var=float(r[3]), valid=float(r[4]))

    def pcg32(self, seed, seq, n):
        u = np.zeros(n, np.uint32)
        f = np.zeros(n, np.float32)
        self.lib.orc_pcg32(C.c_uint64(seed), C.c_uint64(seq), C.c_uint32(n), u.ctypes.data_as(C.c_void_p), _ptr(f))
        return u, f

    # ---- NN ----
    def nn_create(self, pos_id=3, dir_id=0, width=64, depth=6, loss_id=0, lr=0.01, ema_decay=0.99, seed=1337,
                  hashgrid_log2_size=0, optimizer="Adam"):
        cfg = OrcNNConfig(pos_id, dir_id, width, depth, loss_id, lr, ema_decay, seed, hashgrid_log2_size,
                          {"Adam": 0, "SGD": 1}[optimizer])
        h = self.lib.orc_nn_create(C.byref(cfg))
        if not h:
            raise ValueError("unsupported encoding for the oracle")
        return OracleNN(self, h)


class OracleNN:
    MASTER, EMA, ADAM_M, ADAM_V, GRAD = range(5)

    def __init__(self, orc, handle):
        self.orc, self.h = orc, C.c_void_p(handle)
        self.n_params = int(orc.lib.orc_nn_param_count(self.h))
        self.enc_dims = int(orc.lib.orc_nn_encoded_dims(self.h))
        self.n_mlp = int(orc.lib.orc_nn_mlp_param_count(self.h))

    def buffer(self, which):
        p = self.orc.lib.orc_nn_buffer(self.h, C.c_int(which))
        return np.ctypeslib.as_array(p, shape=(self.n_params,))

    def set_step(self, step):
        self.orc.lib.orc_nn_set_step(self.h, C.c_uint32(step))

    def tcnn_params(self, which=0, width=None):
        """buffer `which` in tiny-cuda-nn's own layout: the output matrix padded to 16 rows (rows 3..15: the values the
        initialisation drew for them for which 0 / 1, zeros otherwise), then the table"""
        v = self.buffer(which)
        n_out = 3 * width
        head, tail = v[:self.n_mlp - n_out], v[self.n_mlp:]
        out3 = v[self.n_mlp - n_out:self.n_mlp]
        f = self.orc.lib.orc_nn_tcnn_dead_rows
        f.restype = C.POINTER(C.c_float)
        dead = np.ctypeslib.as_array(f(self.h), shape=(13 * width,)).copy()
        if which not in (0, 1):
            dead[:] = 0
        return np.concatenate([head, out3, dead, tail]).astype(np.float32)

    def encode(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((x.shape[0], self.enc_dims), np.float32)
        self.orc.lib.orc_nn_encode(self.h, _ptr(x), C.c_uint32(x.shape[0]), _ptr(out))
        return out

    def forward(self, x, use_ema=True, mode=1):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((x.shape[0], 3), np.float32)
        self.orc.lib.orc_nn_forward(self.h, _ptr(x), C.c_uint32(x.shape[0]), C.c_int(int(use_ema)), C.c_int(mode), _ptr(out))
        return out

    def backward(self, x, target, n_norm=None, accumulate=False):
        x = np.ascontiguousarray(x, np.float32)
        t = np.ascontiguousarray(target, np.float32)
        n = x.shape[0]
        return float(self.orc.lib.orc_nn_backward(self.h, _ptr(x), _ptr(t), C.c_uint32(n),
                                                  C.c_uint32(n_norm or n), C.c_int(int(accumulate))))

    def optimizer_step(self):
        self.orc.lib.orc_nn_optimizer_step(self.h)

    def __del__(self):
        try:
            self.orc.lib.orc_nn_destroy(self.h)
        except Exception:
            pass
