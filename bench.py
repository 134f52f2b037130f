#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NRC-HPM hot path on MI355X.

Metric (BASELINE.json): Msamples/s + ms/frame at 1080p, 256^3 cloud.  One *step* = one 4-spp frame of configs[1]
(1x MI355X NRC path: 256^3 cloud, 1920x1080, 4 spp, 6x64 fp16 MLP, HDR env map) = 4 blended sub-frames; every
sub-frame is what one reference frame is, `NrcHpmRenderer::Render(queue, true)` (src/main.cu:287): path integrator,
NRC inference for every pixel, train-ray generation, 16 384 train rays + one Adam step (configs[2]), compositing.
`--train 0` drops the training step.  One sample = one pixel path.

  python bench.py --gpus N --steps K --warmup W [--config c2|c4|c5]
N > 1: one rank per GPU under torch.distributed.run.  Started WITHOUT that launcher (`python bench.py --gpus 8`), this script
launches it itself -- before anything touches the GPU -- relays rank 0's single JSON line and exits with the job's status; it never
runs fewer ranks than it was asked for.  The frame is sharded by interleaved strips of 8 pixel columns and the MLP gradients are
all-reduced over RCCL each training step (by the library itself: nrc_cache_comm_init).  Rank 0 prints ONE JSON line.

  --config c2 (default)  configs[1]+[2]: 1920x1080 per GPU, 256^3 cloud, 4 spp, 6x64; N > 1 is WEAK scaling (every rank keeps a
                         ~1920x1080-pixel share of the same view at sqrt(N) x the resolution, 16 384 train rays per rank).  With N > 1 the line also carries
                         `strong_scaling_c4` (the configs[3] figure below, measured after the timed region) and the per-step
                         all-reduce time.
  --config c4            configs[3]: ONE 3840x2160 frame, 8 spp, sharded over the N ranks; STRONG scaling (the global frame and the
                         global train batch of 16 384 rays are fixed: each rank gets 1/N of both)
  --config c5            configs[4]: 512^3 seeded smoke, 8x128 MLP + one-blob, 1920x1080 per GPU, 4 spp (weak scaling like c2)
  --strong               any preset as STRONG scaling: its frame is the global frame.  `--gpus N --strong` splits the 1920x1080 frame of the
                         metric itself over the N ranks; the default N > 1 line carries that figure too (`strong_scaling_1080p`).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the renderer overlaps four HIP streams; give the runtime enough hardware queues that they never share one (must be set
# before the HIP runtime initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

MFMA_F16_PEAK_TFLOPS = 2500.0      # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0              # HBM3E spec (6.3 TB/s measured copy)
MLP_FLOP_PER_SAMPLE = 51584.0      # 2*(80*64 + 5*64*64 + 64*3), SURVEY.md 8(d)
MLP_BYTES_PER_SAMPLE = 32.0        # 20 B query + 12 B radiance


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--train", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--volume", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # model overrides (default = the north-star model of BASELINE.json: Frequency+OneBlob, 6x64)
    ap.add_argument("--pos-id", type=int, default=3)
    ap.add_argument("--dir-id", type=int, default=0)
    ap.add_argument("--nn-width", type=int, default=64)
    ap.add_argument("--nn-depth", type=int, default=6)
    ap.add_argument("--smoke-volume", action="store_true", help="configs[4]: seeded smoke plume instead of the fBm cloud")
    ap.add_argument("--compat-fix", type=int, default=0,
                    help="nrc_config.compat_fix: 0 = the reference's behaviour as shipped (quirk Q2: training targets are single-vertex estimates, "
                         "the CLI's trainRayLength 32 is ignored, src/NrcHpmRenderer.cu:991-994 vs :1036-1055), 2 = Q2 fixed (train ray length 32: "
                         "the algorithm the CLI asks for), 1 / 3 = Q1 (TRAIN_Y_DIST) fixed as well")
    ap.add_argument("--exchange", choices=["f32", "f16"], default="f32",
                    help="N > 1: what the per-step gradient all-reduce carries (nrc_cache_set_exchange_dtype): the fp32 vector (103 KB, default) or "
                         "the gradients as fp16 numbers pre-scaled by loss_scale 128 (52 KB; BASELINE.json configs[3] words it so)")
    ap.add_argument("--no-quality", action="store_true", help="skip the untimed quality leg (the trained frame against this build's own MC ground truth)")
    ap.add_argument("--config", choices=["c2", "c4", "c5"], default="c2", help="BASELINE.json preset (see the module docstring)")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: the frame named by --config / --width / --height IS the global frame and the global train batch stays 16 384 "
                         "rays (strong scaling) instead of every rank keeping a 1080p share (weak)")
    return ap.parse_args(argv)


def apply_preset(args):
    """returns `strong`: the preset fixes the GLOBAL frame and train batch (configs[3])"""
    strong = bool(getattr(args, "strong", False))
    if args.config == "c4":        # configs[3]: one 4K frame, 8 spp, tile shard, global train batch fixed
        args.width, args.height, args.spp, strong = 3840, 2160, 8, True
    elif args.config == "c5":      # configs[4]: 512^3 smoke + 8x128
        args.volume, args.smoke_volume, args.nn_width, args.nn_depth = 512, True, 128, 8
    return strong


# ---------------------------------------------------------------------------------------------------------------- self-launch
def visible_gpus():
    """GPUs this process could use, counted WITHOUT initialising the HIP runtime (a parent that has touched the GPU must not start
    the ranks): the KFD topology's nodes with SIMDs, narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES"""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
    except OSError:
        n = 0
    if n == 0:
        # no KFD view (some containers): nothing here may touch the HIP runtime -- this process starts the ranks -- so the request is
        # trusted (None) and a rank without a device fails by itself ("rank r has no device")
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as a fresh child job, relay rank 0's line"""
    shared = os.environ.get("NRC_BENCH_SHARED_GPU") == "1"      # rehearsal on ONE device: all ranks on cuda:0, gloo + hook exchange
    have = visible_gpus()
    if have is not None and have < args.gpus and not shared:
        print("bench.py: --gpus %d but only %d GPU(s) visible; refusing to run under an %d-GPU label"
              % (args.gpus, have, args.gpus), file=sys.stderr)
        return 2
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for ln in child.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    rc = child.wait()
    if line is not None:
        print(line)
    elif rc == 0:
        print("bench.py: the %d-rank job ended without a result line" % args.gpus, file=sys.stderr)
        rc = 3
    return rc


# ---------------------------------------------------------------------------------------------------------------- workloads
def global_frame(n_gpus, w, h, strong=False):
    """weak scaling: every rank renders ~w*h pixels (interleaved column strips) of the SAME VIEW at sqrt(N) times the resolution -- the
    aspect ratio, hence the share of the frame the medium covers and the work per pixel, stay those of the one-GPU frame (a 2w x h or
    4w x 2h frame at the same vertical field of view would halve the medium's share: the per-rank work would not be fixed); both sides
    are multiples of the 8-pixel tile, the pixel count is within 0.1 % of N*w*h (N = 2: 2712x1528, 4: 3840x2160, 8: 5432x3056 for
    1920x1080).  strong: w x h IS the global frame"""
    if strong or n_gpus == 1:
        return (w, h)
    r = float(n_gpus) ** 0.5
    return (max(8, int(round(w * r / 8.0)) * 8), max(8, int(round(h * r / 8.0)) * 8))


def gpu_mc_baseline(api, sc, scene, W, H, frames=20, keep_warm_ms=0.0):
    """the like-for-like GPU figure beside cpu_baseline: McHpmRenderer (mc/render.comp, PATH_LENGTH 32), same scene and camera.
    keep_warm_ms > 0: that much more of the same work is enqueued behind the measurement and NOT awaited -- the caller enqueues its
    warm-up steps at once, so the GPU does not fall idle in between (see main) -- and the renderer is returned for a later Destroy"""
    import torch
    cam = sc.make_camera(aspect=W / H)
    mc = api.McHpmRenderer(W, H, 32, True, cam, scene)
    for _ in range(3):
        mc.Render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        mc.Render()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = dict(value=W * H * frames / dt / 1e6, unit="Msamples/s", ms_per_frame=dt / frames * 1e3,
               kernel="k_mc_render (mc/render.comp, PATH_LENGTH 32)", sample="%d frames of %dx%d" % (frames, W, H))
    if keep_warm_ms > 0.0:
        for _ in range(max(1, int(keep_warm_ms / (dt / frames * 1e3) + 0.5))):
            mc.Render()
        return res, mc
    mc.Destroy()
    return res, None


def cpu_baseline(scene, W, H, budget_s=float(os.environ.get("NRC_BENCH_CPU_BUDGET_S", "20"))):
    """The reference has no CPU renderer (McHpmRenderer dispatches mc/render.comp, src/McHpmRenderer.cpp:93,875):
    the baseline is the oracle's restatement of mc/render.comp (PATH_LENGTH 32) on this box's host cores."""
    import numpy as np
    from nrc_hpm_renderer_amd import scene as sc
    from oracle import Oracle
    import tempfile
    threads = os.cpu_count() or 1
    orc = Oracle(native=True, out_dir=tempfile.mkdtemp(prefix="nrc_oracle_"))
    cam = sc.make_camera(aspect=W / H)
    fr = [0.25, 0.5, 0.75, 1.0]
    # bounded sample (~10-30 s of CPU work): whole frames when the box is fast enough, otherwise a row band through
    # the middle of the frame scaled to the budget (rows are the oracle's unit of thread parallelism)
    out = np.zeros((H, W, 4), np.float32)
    probe_rows = max(threads // 4, 8)
    y0 = (H - probe_rows) // 2
    t0 = time.time()
    orc.mc_render(scene, cam, W, H, 32, fr, out=out, rows=(y0, y0 + probe_rows), threads=threads)
    per_row = (time.time() - t0) / probe_rows
    rows = int(min(H, max(probe_rows, budget_s / max(per_row, 1e-6))))
    frames = 1
    if rows >= H:
        rows = H
        frames = int(max(1, min(16, budget_s / max(per_row * H, 1e-6))))
    y0 = (H - rows) // 2
    t0 = time.time()
    for _ in range(frames):
        orc.mc_render(scene, cam, W, H, 32, fr, out=out, rows=(y0, y0 + rows), threads=threads)
    t_all = time.time() - t0
    px = frames * rows * W
    sample = "%d frame(s) x rows [%d,%d) of the %dx%d frame" % (frames, y0, y0 + rows, W, H)
    return dict(value=px / t_all / 1e6, unit="Msamples/s", cores=threads, kind="port",
                sample="oracle mc/render.comp restatement (PATH_LENGTH 32, g++ -O3 -march=native, %d threads), %s, "
                       "same 256^3 cloud/scene/camera, %.1f s" % (threads, sample, t_all))


def issue_roofline(gen_ms, kernel="k_gen_rays<false>"):
    """the ruler the integrator is actually bound by: vector-instruction ISSUE.  From the newest committed counter
    passes of this command (profiles/rNN_pmc_sq_counters.txt: SQ_INSTS_VALU per launch), the measured issue rate of a SIMD at the kernel's
    occupancy (profiles/rNN_micro_issue_mix.txt: clocks per vector instruction at five waves per SIMD, tools/issue_mix.hip) and the loop
    profile of the counter build (profiles/rNN_loop_profile.txt: lanes doing useful work per issued tracking-loop trip).  Constants read from
    profiles/, labelled with their files; only the kernel's duration is this run's."""
    import glob
    import re

    def newest(pattern):
        hits = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        return hits[-1] if hits else None

    f_sq, f_mix, f_loop = newest("r[0-9][0-9]_pmc_sq_counters.txt"), newest("r[0-9][0-9]_micro_issue_mix.txt"), newest("r[0-9][0-9]_loop_profile.txt")
    if not f_sq or not f_mix:
        return None
    valu = None
    block = False
    for ln in open(f_sq):
        if not ln.startswith(" "):
            block = ln.startswith(kernel)
        elif block and ln.split()[0] == "SQ_INSTS_VALU":
            valu = float(ln.split()[1])
    m = re.search(r"v_fma_f32, 8 chains.*?5w\s+([0-9.]+)", open(f_mix).read())
    if valu is None or not m:
        return None
    clk_per_inst = float(m.group(1))
    simds, clock_ghz = 1024, 2.1      # 256 CUs x 4 SIMDs; the clock this kernel holds under load (tools/loop_profile.py stamps, +-5 % by box)
    floor_ms = valu * clk_per_inst / simds / (clock_ghz * 1e9) * 1e3
    out = dict(bound="valu-issue", valu_wave_instructions_per_launch=valu, clocks_per_instruction_at_5_waves_per_simd=clk_per_inst,
               simds=simds, clock_ghz_assumed=clock_ghz, issue_floor_ms=floor_ms, frac=floor_ms / gen_ms if gen_ms > 0 else None,
               source="%s (SQ_INSTS_VALU per launch), %s (issue rate) -- committed, not measured in this run" % (os.path.basename(f_sq), os.path.basename(f_mix)))
    if f_loop:
        util = {}
        for ln in open(f_loop):
            mm = re.match(r"(delta_track step|ratio_track step)\s+(\d+)\s+(\d+)\s+([0-9.]+)", ln)
            if mm:
                util[mm.group(1).split("_")[0]] = float(mm.group(4))
        if util:
            out["tracking_loop_lane_utilisation"] = util
            out["lane_utilisation_source"] = os.path.basename(f_loop) + " (tools/loop_profile.py, counter build)"
    return out


def quality_leg(job, mc_ms_per_frame, trained_frames, train_to=512, eval_frames=32, gt_frames=256):
    """What the timed path buys (VERDICT r05 "missing" 2), UNTIMED and after the timed region: the reference's own self-check --
    Reference::CompareNrc (src/Reference.cpp:72-107: the NRC image rendered without training, compared through cmp1/cmp2.comp =
    nrc_compare_images) -- of the cache this run trained, against this build's own ground truth of the bench scene generated like
    Reference::GenRefImages (src/Reference.cpp:566-606: McHpmRenderer, PATH_LENGTH 64, blended).  (The reference's EXRs are of ITS cloud and
    scenes: tools/convergence.py and tests/test_gpu_quality.py compare against those; profiles/r06_convergence_*.txt.)  One evaluation frame
    is one path per pixel -- its MSE is the primary path's noise -- so `eval_frames` evaluation frames are blended; the same wall time of
    plain Monte-Carlo frames is evaluated beside it.  In the faithful mode (quirk Q2) the cache can only learn the third vertex's direct
    light: `faithful_limit` is the image McHpmRenderer renders with PATH_LENGTH 3, the frame the faithful path converges to."""
    import math
    torch, api, sc, args = job.torch, job.api, job.sc, job.args
    W, H = job.local_w, job.gh

    def result(ref, img):
        r = api.CompareImages(ref, img)
        return dict(mse=r["mse"], rel_bias=(r["own_mean"] - r["ref_mean"]) / r["ref_mean"] if r["ref_mean"] else None,
                    cv=math.sqrt(max(r["own_var"], 0.0)) / r["own_mean"] if r["own_mean"] else None)

    def mc_image(path_length, frames, seed):
        mc = api.McHpmRenderer(W, H, path_length, True, job.cam, job.scene)
        frs = sc.frame_randoms(frames, seed=seed)
        for f in range(frames):
            mc.SetFrameRandom(frs[f])
            mc.Render()
        img = mc.GetImage().clone()
        torch.cuda.synchronize()
        mc.Destroy()
        return img

    more = max(0, train_to - trained_frames)
    if more and args.train:
        job.ren.RenderFrames(sc.frame_randoms(more, seed=2024), True)
        trained_frames += more
    gt = mc_image(64, gt_frames, 31337)
    ev = api.NrcHpmRenderer(W, H, True, job.cam, job.cfg, job.scene, job.nrc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev.RenderFrames(sc.frame_randoms(eval_frames, seed=4242), False)
    torch.cuda.synchronize()
    eval_ms = (time.perf_counter() - t0) * 1e3
    q = result(gt, ev.GetImage())
    ev.Destroy()
    n_mc = max(1, int(eval_ms / mc_ms_per_frame + 0.5))
    mc_same = result(gt, mc_image(32, n_mc, 777))
    limit = result(gt, mc_image(3, 64, 999))
    return dict(what="Reference::CompareNrc on the trained cache: %d blended evaluation frames (train = false) vs own ground truth"
                     % eval_frames,
                reference="McHpmRenderer PATH_LENGTH 64, %d blended frames of the bench scene (its own noise, ~single-frame MC MSE / %d, is inside `mse`)"
                          % (gt_frames, gt_frames),
                frames=trained_frames, eval_frames=eval_frames, eval_ms=eval_ms, rel_bias=q["rel_bias"], mse=q["mse"], cv=q["cv"],
                mc_equal_time=dict(frames=n_mc, mse=mc_same["mse"], rel_bias=mc_same["rel_bias"],
                                   note="McHpmRenderer PATH_LENGTH 32 blended for the evaluation frames' wall time"),
                faithful_limit=dict(rel_bias=limit["rel_bias"],
                                    note="McHpmRenderer PATH_LENGTH 3 (64 frames): what the frame converges to while quirk Q2 keeps the training "
                                         "targets single-vertex estimates; with --compat-fix 2 the limit is the ground truth itself"))


class Job:
    """one preset on this rank: scene, cache, renderer, exchange; `timed(steps, warmup)` is the contract's timed region"""

    def __init__(self, args, strong, rank, world, use_dist, shared_gpu):
        import torch
        import torch.distributed as dist
        from nrc_hpm_renderer_amd import api, scene as sc, parallel
        self.torch, self.dist, self.api, self.sc, self.parallel = torch, dist, api, sc, parallel
        self.args, self.strong, self.rank, self.world, self.use_dist = args, strong, rank, world, use_dist
        W, H = args.width, args.height
        # ---- synthetic inputs: seeded 256^3 fBm cloud, procedural HDR sky, scene preset 4 values
        vol = sc.cached_volume("smoke" if args.smoke_volume else "cloud", args.volume, seed=1337)
        self.scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
        self.gw, self.gh = global_frame(world, W, H, strong)
        tile = parallel.column_tile(rank, world, self.gw, self.gh)       # (x_offset, x_stride, global_w, global_h, x_block)
        self.local_w = parallel.local_width(rank, world, self.gw)
        cam = sc.make_camera(aspect=self.gw / self.gh)
        # strong scaling keeps the GLOBAL train batch at 16 384 rays (2^14 / world per rank; world must be a power of two <= 512)
        self.log2_train = 14
        if strong:
            if world & (world - 1) or world > 512:
                raise SystemExit("--config c4 needs a power-of-two world size")
            self.log2_train = 14 - (world.bit_length() - 1)
        cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=self.log2_train, log2_infer_batch_size=21, scene_id=4,
                            primary_ray_length=1, primary_ray_prob=0.0, train_spp=1, train_ring_buf_size=1.0, seed=1337,
                            pos_id=args.pos_id, dir_id=args.dir_id, nn_width=args.nn_width, nn_depth=args.nn_depth,
                            train_ray_length=32, compat_fix=int(getattr(args, "compat_fix", 0)))      # (trainRayLength: src/main.cu:438)
        self.cfg, self.cam = cfg, cam
        self.nrc = api.NeuralRadianceCache(cfg)
        self.ren = api.NrcHpmRenderer(self.local_w, self.gh, True, cam, cfg, self.scene, self.nrc, tile=tile)
        self.exchange = dict(path="none", rccl_rank=None, rccl_ranks=0)
        if use_dist and args.train:
            # RCCL all-reduce of the MLP gradients every training step: issued by the library itself on its training stream; if the
            # library cannot bring up its own communicator the same exchange goes through torch.distributed's communicator
            try:
                if shared_gpu:
                    raise RuntimeError("NRC_BENCH_SHARED_GPU=1: all ranks share one device, RCCL needs one device per rank")
                parallel.attach_gradient_allreduce(self.nrc, world, dtype=getattr(args, "exchange", "f32"))
                r_, w_ = self.nrc.CommInfo()              # what ncclCommUserRank / ncclCommCount say
                self.exchange = dict(path="native (nrc_cache_comm_init -> ncclAllReduce on the training stream)", rccl_rank=r_, rccl_ranks=w_,
                                     grid_gradient_lists=self.nrc.CommSparse())
                if w_ != world:
                    raise RuntimeError("RCCL communicator reports %d ranks, expected %d" % (w_, world))
            except RuntimeError as e:
                print("warning: native RCCL exchange unavailable (%s); using the torch.distributed hook" % e, file=sys.stderr)
                parallel.attach_gradient_allreduce(self.nrc, world, native=False, dtype=getattr(args, "exchange", "f32"))
                self.exchange = dict(path="torch.distributed hook (%s)" % dist.get_backend(), rccl_rank=rank, rccl_ranks=world)
                if world > 1 and dist.get_backend() == "nccl":
                    api.set_wave_priority_raise(False)      # torch's RCCL kernels run beside the library's: one priority for all (nrc_hpm.h)
                if world > 1:
                    self.nrc.SetCollectiveHooks(rank, world)      # frame gather / metric reduction over the same transport
        self.randoms = None
        self.ri = 0

    def step(self):
        """one spp-sample frame: spp x NrcHpmRenderer::Render(queue, train), enqueued by ONE call into the library
        (nrc_renderer_render_frames: the loop of src/main.cu:287 on the library's side of the binding)"""
        self.ren.SetBlend(True)            # progressive blend restarts: sub-frame i has blendFactor 1/(i+1)
        n = len(self.randoms)
        rows = [(self.ri + k) % n for k in range(self.args.spp)]
        self.ri += self.args.spp
        self.ren.RenderFrames(self.randoms[rows], bool(self.args.train))

    def barrier(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def prepare(self, steps, warmup):
        """host-side preparation of timed(): nothing of it may fall between the pre-warming GPU work and the warm-up steps"""
        self.randoms = self.sc.frame_randoms((steps + warmup) * self.args.spp + 8, seed=1337)

    def timed(self, steps, warmup):
        """W untimed warm-up steps, barrier + synchronize, EXACTLY K steps, barrier + synchronize; MAX over ranks"""
        if getattr(self, "randoms", None) is None:
            self.prepare(steps, warmup)
        for _ in range(warmup):
            self.step()
        self.barrier()
        self.ren.ResetStageStats()      # O(1): the GPU must not wait for the host here (see main: it leaves its load clocks within a millisecond or two)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.barrier()
        dt = time.perf_counter() - t0
        self.stats = self.ren.StageStats(reset=True)
        if self.use_dist:
            tt = self.torch.tensor([dt], device="cuda", dtype=self.torch.float64)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        # every rank's own pixel count (interleaved strips: local widths differ by at most one strip)
        samples = float(sum(self.parallel.local_width(r, self.world, self.gw) for r in range(self.world))) * self.gh * self.args.spp * steps
        return dict(value=samples / dt / 1e6, ms_per_step=dt / steps * 1e3, seconds=dt)

    def allreduce_us(self):
        """per-step all-reduce time of the native exchange (collective); None on the hook path"""
        if self.exchange["path"].startswith("native"):
            return self.nrc.CommTimeExchange(100)
        return None

    def frame_assembly_ms(self, reps=5):
        """what the product's own multi-GPU frame assembly costs (collective calls, every rank): nrc_renderer_gather_frame -- one
        all-gather of the ranks' column strips + a de-interleave -- and nrc_compare_images_sharded -- five local fp64 sums, two tiny
        all-reduces -- each averaged over `reps` calls behind one warm-up call; None where the frame is not sharded"""
        if self.world <= 1 or not (self.use_dist and self.args.train):
            return None
        torch = self.torch
        ref = torch.rand((self.gh, self.local_w, 4), device="cuda")
        own = self.ren.GetImage().contiguous()
        out = {}
        for name, fn in (("gather_frame_ms", lambda: self.ren.GatherFrame()),
                         ("compare_sharded_ms", lambda: self.api.CompareImagesSharded(self.nrc, ref, own))):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            out[name] = (time.perf_counter() - t0) / reps * 1e3
        return out

    def close(self):
        self.ren.Destroy()
        self.nrc.Destroy()


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))          # nothing has touched the GPU (or imported torch.cuda state) in this process
    strong = apply_preset(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a launcher-provided world size that contradicts --gpus is a mis-launch: the line would carry the wrong label
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    shared_gpu = os.environ.get("NRC_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no device (LOCAL_RANK %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run (also with one rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from nrc_hpm_renderer_amd import api, scene as sc

    W, H, spp = args.width, args.height, args.spp
    job = Job(args, strong, rank, world, use_dist, shared_gpu)
    ren, nrc, scene, exchange = job.ren, job.nrc, job.scene, job.exchange
    gw, gh, local_w, log2_train = job.gw, job.gh, job.local_w, job.log2_train
    north_star = (args.pos_id, args.dir_id, args.nn_width, args.nn_depth) == (3, 0, 64, 6)

    # The like-for-like GPU figure beside cpu_baseline (the same Monte-Carlo algorithm, k_mc_render) is measured FIRST, on every
    # rank, and the GPU is then kept busy until the warm-up steps are enqueued.  Reason (tools/step_profile.py): this GPU leaves its
    # load clocks within milliseconds of idleness -- after a 10 ms pause the first 20 frames run at 0.284 ms of gen_rays instead of
    # 0.244, after 2 ms at 0.252 -- and needs ~25 ms (100 frames) of load to get back; `--warmup 5` is 6 ms of work and 20 timed steps
    # are 22 ms, so a timed region that starts behind ANY host-side pause (the read-back of the Monte-Carlo leg, freeing its
    # renderer, drawing the random numbers) measures the ramp, not the renderer: 7 140-7 510 Msamples/s for `--steps 20 --warmup 5`
    # against 7 670-7 700 for `--steps 100 --warmup 10` or `--steps 20 --warmup 25`.  So everything the host needs is prepared
    # first, ~40 ms of untimed Monte-Carlo frames are enqueued behind the measured ones and not awaited, and the warm-up steps follow
    # at once; the barrier in front of the timed region still drains all of it.
    job.prepare(args.steps, args.warmup)
    mc_baseline, mc_keep = gpu_mc_baseline(api, sc, scene, W, H, keep_warm_ms=float(os.environ.get("NRC_BENCH_KEEP_WARM_MS", "40")))
    res = job.timed(args.steps, args.warmup)
    if mc_keep is not None:
        mc_keep.Destroy()
    stats = job.stats
    schedule = ren.GetSchedule()
    value, ms_per_step = res["value"], res["ms_per_step"]
    loss = nrc.GetLoss() if args.train else None
    # what the host spends enqueueing one frame (through the Python mirror, queues empty, nothing awaited): the launch path is on
    # the critical path only if this approaches the GPU's frame time
    ren.SetBlend(True)
    t_h = time.perf_counter()
    for _ in range(2):
        ren.RenderFrames(job.randoms[:4], bool(args.train))
    host_enqueue_ms = (time.perf_counter() - t_h) / 8 * 1e3
    torch.cuda.synchronize()
    ren.StageStats(reset=True)
    allreduce_us = job.allreduce_us() if use_dist and args.train else None
    nrc_dtype = nrc.GetExchangeDtype()
    assembly = job.frame_assembly_ms()

    # ---- integrator traffic model: density look-ups counted on the device for extra (untimed) sub-frames of the same seed --
    # n_fetch: the ALGORITHM's look-ups (every camera ray walked, as the reference and the oracle do; empty-space early-out off),
    # n_fetch_executed: what the timed kernel really issues (early-out on: rays through provably empty space skip their walk)
    def count(skip):
        ren.SetEmptySkip(skip)
        ren.CountFetches(True)
        ren.SetFrameRandom(job.randoms[0])
        ren.Render(None, False)
        torch.cuda.synchronize()
        return ren.CountFetches(False)

    n_fetch = count(False)
    n_fetch_executed = count(True)
    n_px = local_w * gh
    quality = None
    if world == 1 and args.train and not args.no_quality:
        quality = quality_leg(job, mc_baseline["ms_per_frame"], (args.steps + args.warmup) * spp + 8)

    out = None
    if rank == 0:
        # ---- per-launch figures.  k_infer: event-timed loop on the launch stream (same buffers the frame uses); k_gen_rays:
        # the renderer's own HIP events around the launch, averaged over the timed region.
        n_inf = n_px
        d_out = ren.Buffer("infer_output")

        def time_infer(d_in):
            # at the GPU's load clocks: it has been idle while the host counted look-ups above, and the first ~25 ms of work after an
            # idle millisecond run up to 15 % slower (see main; tools/bench_mlp.py shows the ramp launch by launch) -- 30 ms of
            # untimed launches go first, the timed ones follow without a host wait in between
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            warm = max(3, int(30.0 / (0.1 * max(n_inf, 1) / 2073600.0 * flop_scale) + 0.5))
            for _ in range(min(warm, 400)):
                nrc.Infer(d_in, d_out, True)
            reps = 20
            e0.record()
            for _ in range(reps):
                nrc.Infer(d_in, d_out, True)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        flop_scale = (args.nn_width / 64.0) ** 2 * args.nn_depth / 6.0      # rough cost of a launch relative to the 6x64 model's 0.1 ms
        # (a) dense pass over uniformly random queries (positions in the quirk-Q3 range): the conservative figure -- zero or
        #     repetitive operands let the chip hold a higher clock (cdna_hip_programming.md rule 25);
        # (b) dense pass over this frame's own query buffer (78 % all-zero queries of unscattered pixels)
        g = torch.Generator(device="cuda").manual_seed(1)
        rnd = torch.rand((n_inf, 5), device="cuda", generator=g)
        rnd[:, :3] += 31.0
        mlp_ms = time_infer(rnd)
        mlp_ms_frame = time_infer(ren.Buffer("infer_input"))
        # FLOP per sample of the configured model: 2 * (E * W + (D - 1) * W^2 + W * 3), E = encoded input dims
        enc = {0: 32, 1: 3, 2: 36, 3: 72}[args.pos_id] + {0: 8, 1: 2, 2: 8}[args.dir_id]
        flop = 2.0 * (enc * args.nn_width + (args.nn_depth - 1) * args.nn_width ** 2 + args.nn_width * 3)
        assert not north_star or flop == MLP_FLOP_PER_SAMPLE
        mlp_tflops = flop * n_inf / (mlp_ms * 1e-3) / 1e12
        if not (0.0 < mlp_tflops < MFMA_F16_PEAK_TFLOPS):
            raise RuntimeError("MLP timing is not physical (%.1f TFLOP/s): the events did not bracket the kernel's stream" % mlp_tflops)
        gen_ms = stats["gen_rays"]
        # algorithmic bytes of the integrator, SURVEY.md 8(d): n_fetch x 1 B (counted on the device for the same seeds) + 16 B
        # framebuffer write + 32 B NRC query I/O per pixel.  What the kernel stores today on top of that (primary colour, info,
        # vertex images for the train rays) is reported separately and is not credited.
        gen_bytes = n_fetch * 1.0 + n_px * (16.0 + 32.0)
        # (the query of a pixel that did not scatter is not written since round 4: the inference walks the frame's live-query list)
        n_live = int((ren.Buffer("info") == 1.0).sum().item())
        gen_store_bytes = n_px * (16 + 4) + n_live * (20 + 4) + ren.VertexImageBytes()
        # `traffic`: HBM-side bytes per launch from the rocprofv3 PMC passes of the SAME command, committed under profiles/
        # (FETCH_SIZE / WRITE_SIZE in passes of their own, gfx950 corrections of MI355X_MICROARCH.md) -- a constant read from that
        # file, not measured in this run; quoted only when this run is the profiled workload, and labelled with its source.
        traffic, traffic_source = {}, None
        def committed(name):      # profiles/rNN_<name>, newest round first
            import glob
            return [os.path.relpath(f, ROOT) for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + name)), reverse=True)]

        for tf in committed("pmc_traffic.json"):
            if os.path.exists(os.path.join(ROOT, tf)) and (W, H, args.volume, world, args.config) == (1920, 1080, 256, 1, "c2"):
                with open(os.path.join(ROOT, tf)) as f:
                    traffic = {k: v["traffic_bytes"] for k, v in json.load(f)["kernels"].items()}
                traffic_source = tf + " (committed rocprofv3 --pmc passes of this command; not measured in this run)"
                break
        # kernel-only duration of the dense inference launch from the committed rocprofv3 kernel trace of tools/bench_mlp.py (the
        # event-timed loop above includes the ~17 us between consecutive launches); a constant read from profiles/, labelled so
        def trace_duration(paths, key):
            import csv
            for path in paths:
                full = os.path.join(ROOT, path)
                if not os.path.exists(full):
                    continue
                for row in csv.DictReader(open(full)):
                    if key in row["Name"]:
                        return dict(source=path + " (committed rocprofv3 --kernel-trace --stats; not measured in this run)", calls=int(row["Calls"]),
                                    avg_us=float(row["AverageNs"]) / 1e3, min_us=float(row["MinNs"]) / 1e3,
                                    frac_of_peak_avg=flop * 2073600 / (float(row["AverageNs"]) * 1e-9) / 1e12 / MFMA_F16_PEAK_TFLOPS)
            return None

        # ... and the steady-state tail of that trace (tools/trace_tail.py: the last 100 launches, behind the GPU's clock ramp): the figure
        # `frac` quotes when a committed trace exists, so that the line's MLP fraction can be reproduced from profiles/ alone
        def trace_tail(paths):
            import re
            for path in paths:
                full = os.path.join(ROOT, path)
                if os.path.exists(full):
                    m = re.search(r"last (\d+): average ([0-9.]+) us, minimum ([0-9.]+) us, maximum ([0-9.]+) us", open(full).read())
                    if m:
                        return dict(source=path + " (committed rocprofv3 kernel trace, tools/trace_tail.py; not measured in this run)",
                                    launches=int(m.group(1)), avg_us=float(m.group(2)), min_us=float(m.group(3)), max_us=float(m.group(4)))
            return None

        mlp_trace, mlp_tail = None, None
        if n_inf == 2073600 and north_star:
            mlp_trace = trace_duration(committed("mlp_kernel_stats.csv"), "k_infer")
            mlp_tail = trace_tail(committed("mlp_kernel_trace_tail.txt"))
        elif n_inf == 2073600 and (args.pos_id, args.dir_id, args.nn_width, args.nn_depth) == (3, 0, 128, 8):
            mlp_trace = trace_duration(committed("mlp128_kernel_stats.csv"), "k_infer_gen")
            mlp_tail = trace_tail(committed("mlp128_kernel_trace_tail.txt"))
        dominant_is_gen = gen_ms >= mlp_ms
        enc_inside = (args.pos_id, args.dir_id) == (3, 0)      # Frequency + OneBlob: encoded inside the MLP kernel
        mlp_kernel = ("k_infer (fused encode + 6x64 MLP)" if north_star else
                      "%sk_infer_gen<%d> (%dx%d MLP%s)" % ("" if enc_inside else "k_encode + ", args.nn_width, args.nn_depth, args.nn_width,
                                                          ", encoding inside" if enc_inside else ""))
        # `achieved` / `frac`: THIS run's event-timed figure (a constant read from a committed trace says nothing about the library
        # being measured).  The committed rocprofv3 trace of the same launch -- kernel-only durations, without the ~17 us between two
        # launches the event-timed loop includes -- is quoted beside it as `kernel_trace` / `kernel_trace_tail`, with the build it is of.
        event_timed = dict(ms_per_launch=mlp_ms, achieved=mlp_tflops, frac=mlp_tflops / MFMA_F16_PEAK_TFLOPS,
                           note="HIP events around 20 back-to-back launches on the launch stream, this run, this box")
        mlp_achieved, mlp_ms_quoted, mlp_frac_source = mlp_tflops, mlp_ms, "event_timed (this run)"
        if mlp_tail is not None:
            mlp_tail["frac"] = flop * n_inf / (mlp_tail["avg_us"] * 1e-6) / 1e12 / MFMA_F16_PEAK_TFLOPS
        roof_mlp = dict(bound="mfma", kernel=mlp_kernel, achieved=mlp_achieved, peak=MFMA_F16_PEAK_TFLOPS, flop_per_sample=flop,
                        unit="TFLOP/s", frac=mlp_achieved / MFMA_F16_PEAK_TFLOPS, frac_source=mlp_frac_source, event_timed=event_timed,
                        traffic=traffic.get("k_infer"), traffic_source=traffic_source,
                        algorithmic_bytes=MLP_BYTES_PER_SAMPLE * n_inf, ms_per_launch=mlp_ms_quoted, samples_per_launch=n_inf,
                        data="uniform random queries", kernel_trace=mlp_trace, kernel_trace_tail=mlp_tail,
                        on_frame_queries=dict(ms_per_launch=mlp_ms_frame,
                                              achieved=flop * n_inf / (mlp_ms_frame * 1e-3) / 1e12,
                                              frac=flop * n_inf / (mlp_ms_frame * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS))
        # `achieved` is the contract's figure: ALGORITHMIC bytes (the look-ups the algorithm makes, counted with every ray
        # walked) over the kernel's duration.  The kernel answers part of them without touching memory (the empty-space early-out;
        # look-ups into cells the LDS occupancy bits prove empty): `executed` is the rate of the look-ups the timed kernel really
        # walks -- an upper bound of what reaches the memory system, not credited as bandwidth.
        gen_s = gen_ms * 1e-3
        roof_gen = dict(bound="hbm", kernel="k_gen_rays (delta/ratio tracking path integrator; ALU/latency-bound, quoted against HBM)",
                        achieved=gen_bytes / gen_s / 1e9 if gen_ms > 0 else 0.0, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=(gen_bytes / gen_s / 1e9) / HBM_PEAK_GBS if gen_ms > 0 else 0.0,
                        achieved_is="algorithmic-equivalent bytes / kernel time (SURVEY 8d); not bytes that reached memory",
                        executed=dict(bytes=n_fetch_executed * 1.0 + n_px * (16.0 + 32.0),
                                      gb_per_s=(n_fetch_executed * 1.0 + n_px * (16.0 + 32.0)) / gen_s / 1e9 if gen_ms > 0 else 0.0,
                                      note="look-ups walked by the timed kernel (early-out on); those answered from the LDS occupancy bits included"),
                        traffic=traffic.get("k_gen_rays"), traffic_source=traffic_source, algorithmic_bytes=gen_bytes,
                        bytes_per_pixel="fetches x 1 B + 16 B framebuffer + 32 B query I/O (SURVEY 8d)", stored_bytes=gen_store_bytes,
                        ms_per_launch=gen_ms, fetches_per_pixel=n_fetch / n_px, fetches_executed_per_pixel=n_fetch_executed / n_px,
                        issue=issue_roofline(gen_ms) if (W, H, args.volume, world, args.config) == (1920, 1080, 256, 1, "c2") and north_star else None)
        model = "NRC %dx%d %s+%s" % (args.nn_depth, args.nn_width, {0: "HashGrid(16x2,2^19)", 1: "Identity", 2: "TriangleWave(12)", 3: "Frequency(12)"}[args.pos_id],
                                     {0: "OneBlob(4)", 1: "Identity", 2: "TriangleWave(4)"}[args.dir_id])
        volume = "%d^3 seeded %s" % (args.volume, "smoke plume" if args.smoke_volume else "fBm cloud")
        train_rays = 1 << log2_train
        fix = int(args.compat_fix)
        quirks = ("faithful to the reference as shipped: quirks Q1 (train-grid y stride) and Q2 (SINGLE-VERTEX training targets; the CLI's "
                  "trainRayLength 32 is ignored)" if fix == 0 else
                  "compat_fix %d: %s" % (fix, " + ".join(n for b, n in ((1, "Q1 fixed"), (2, "Q2 fixed (training targets of up to 32 vertices, as the CLI asks)")) if fix & b)))
        if args.config == "c4" or strong:
            workload = (("configs[3]" if args.config == "c4" else "configs[1]+[2] as strong scaling" if args.config == "c2" else "configs[4] as strong scaling") +
                        ": ONE %dx%d frame sharded into %d tiles of interleaved 8-column strips (%d columns on rank 0), %s, %d spp/step, %s, HDR sky "
                        "env map, scene preset 4, train=%d (global batch 16384 rays = %d per rank + 1 Adam step per sub-frame); %s"
                        % (gw, gh, world, local_w, volume, spp, model, args.train, train_rays, quirks))
        else:
            per_gpu = "%dx%d per GPU" % (W, H) if world == 1 else "1/%d of the same view at %dx%d (%d x %d pixels on rank 0; %dx%d on one GPU)" % (world, gw, gh, local_w, gh, W, H)
            workload = ("%s: %s (global %dx%d, tiles of interleaved 8-column strips), %s, %d spp/step, %s, HDR sky env map, scene preset 4, "
                        "train=%d (%d train rays + 1 Adam step per sub-frame); %s"
                        % ("configs[4]" if args.config == "c5" else "configs[1]+[2]", per_gpu, gw, gh, volume, spp, model, args.train, train_rays, quirks))
        exchange = dict(exchange, dtype=nrc_dtype, allreduce_us_per_step=allreduce_us, frame_assembly=assembly)
        if world > 1 and args.train:
            # an N-GPU line is a line about N ranks exchanging gradients: refuse to print one whose exchange saw another number of ranks
            assert exchange["rccl_ranks"] == world, "the gradient exchange ran over %r ranks, the line says %d GPUs" % (exchange["rccl_ranks"], world)
        out = {
            "metric": "Msamples/s + ms/frame at 1080p, 256^3 cloud (NRC path)", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_frame": ms_per_step / spp, "host_enqueue_ms_per_frame": host_enqueue_ms, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f16 (fp16 MFMA operands, fp32 accumulate; fp32 integrator)", "data": "synthetic",
            "config": {"workload": workload, "preset": args.config,
                       "width": W, "height": H, "spp": spp, "volume": args.volume, "train": args.train, "compat_fix": int(args.compat_fix),
                       "parallelism": "pixel-column tiles x%d%s" % (world, " + RCCL grad all-reduce" if use_dist and args.train else "")},
            "exchange": exchange,
            "stage_ms": {k: stats[k] for k in ("gen_rays", "prep_train", "train", "infer", "render", "total")},
            "loss": loss,
            "schedule": schedule,      # what the renderer's tuner chose on this run's own frames (nrc_schedule; placement only)
            "build_id": api.build_id(),
            "roofline": roof_gen if dominant_is_gen else roof_mlp,
            "roofline_mlp": roof_mlp,
            "roofline_integrator": roof_gen,
            "gpu_mc_baseline": mc_baseline,
            "quality": quality,
        }
    job.close()

    # ---- N > 1, default preset (weak scaling): the two STRONG-scaling figures beside it, same ranks, after the timed region -- the metric's
    # own 1920x1080 frame split over the N ranks, and configs[3] (one 3840x2160 frame, 8 spp)
    if world > 1 and args.config == "c2" and (W, H) == (1920, 1080) and not strong:
        for key, preset, what in (("strong_scaling_1080p", "c2", "the metric's frame: ONE 1920x1080 frame, 4 spp/step"),
                                  ("strong_scaling_c4", "c4", "configs[3]: ONE 3840x2160 frame, 8 spp/step")):
            a4 = parse_args(sys.argv[1:])
            a4.config, a4.strong = preset, True
            s4 = apply_preset(a4)
            j4 = Job(a4, s4, rank, world, use_dist, shared_gpu)
            k4, w4 = max(3, args.steps // 4), max(1, args.warmup // 2)
            r4 = j4.timed(k4, w4)
            ar4 = j4.allreduce_us() if a4.train else None
            if rank == 0:
                out[key] = dict(value=r4["value"], unit="Msamples/s", ms_per_step=r4["ms_per_step"], ms_per_frame=r4["ms_per_step"] / a4.spp,
                                steps=k4, warmup=w4, scaling="strong", exchange=dict(j4.exchange, allreduce_us_per_step=ar4),
                                stage_ms={k: j4.stats[k] for k in ("gen_rays", "prep_train", "train", "infer", "render", "total")},
                                workload="%s, %d tiles of interleaved 8-column strips (%d columns per rank), global train batch 16384 rays = %d per rank"
                                         % (what, world, j4.local_w, 1 << j4.log2_train))
            j4.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # cpu_baseline runs the reference's ground-truth algorithm (mc/render.comp, PATH_LENGTH 32: the reference has no CPU
        # renderer); `value` is the NRC path (2 vertices + cache query).  gpu_mc_baseline is the same algorithm as the CPU
        # baseline on the GPU -- the like-for-like ratio is gpu_mc_vs_cpu, gpu_vs_cpu compares the two different estimators.
        out["cpu_baseline"] = cpu_baseline(scene, W, H)
        out["gpu_vs_cpu"] = value / out["cpu_baseline"]["value"]
        out["gpu_mc_vs_cpu"] = out["gpu_mc_baseline"]["value"] / out["cpu_baseline"]["value"]
    elif rank == 0:
        out["cpu_baseline"] = None
        out["cpu_baseline_reason"] = ("--no-cpu-baseline" if args.no_cpu_baseline else
                                      "reported on rank 0 at N=1 only (the contract's bounded host-core sample); this is an N=%d line" % world)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
