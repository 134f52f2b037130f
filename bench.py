#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NRC-HPM hot path on MI355X.

Metric (BASELINE.json): Msamples/s + ms/frame at 1080p, 256^3 cloud.  One *step* = one 4-spp frame of configs[1]
(1x MI355X NRC path: 256^3 cloud, 1920x1080, 4 spp, 6x64 fp16 MLP, HDR env map) = 4 blended sub-frames; every
sub-frame is what one reference frame is, `NrcHpmRenderer::Render(queue, true)` (src/main.cu:287): path integrator,
NRC inference for every pixel, train-ray generation, 16 384 train rays + one Adam step (configs[2]), compositing.
`--train 0` drops the training step.  One sample = one pixel path.

  python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU; the frame is sharded by interleaved pixel columns
(weak scaling: every rank keeps a 1920x1080-pixel tile of a larger frame) and the MLP gradients are all-reduced
over RCCL each training step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the renderer overlaps two HIP streams; give the runtime enough hardware queues that they never share one (must be set
# before the HIP runtime initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

MFMA_F16_PEAK_TFLOPS = 2500.0      # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0              # HBM3E spec (6.3 TB/s measured copy)
MLP_FLOP_PER_SAMPLE = 51584.0      # 2*(80*64 + 5*64*64 + 64*3), SURVEY.md 8(d)
MLP_BYTES_PER_SAMPLE = 32.0        # 20 B query + 12 B radiance


def global_frame(n_gpus, w, h):
    """weak scaling: every rank renders w*h pixels (interleaved columns) of a larger frame"""
    table = {1: (w, h), 2: (2 * w, h), 4: (2 * w, 2 * h), 8: (4 * w, 2 * h)}
    if n_gpus in table:
        return table[n_gpus]
    return (n_gpus * w, h)


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


def main():
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
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run (also with one rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from nrc_hpm_renderer_amd import api, scene as sc, parallel

    W, H, spp = args.width, args.height, args.spp
    # ---- synthetic inputs: seeded 256^3 fBm cloud, procedural HDR sky, scene preset 4 values
    vol = sc.cached_volume("smoke" if args.smoke_volume else "cloud", args.volume, seed=1337)
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
    gw, gh = global_frame(world, W, H)
    tile = parallel.column_tile(rank, world, gw, gh)          # (x_offset, x_stride, global_w, global_h), local width
    local_w = parallel.local_width(rank, world, gw)
    cam = sc.make_camera(aspect=gw / gh)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=14, log2_infer_batch_size=21, scene_id=4,
                        primary_ray_length=1, primary_ray_prob=0.0, train_spp=1, train_ring_buf_size=1.0, seed=1337,
                        pos_id=args.pos_id, dir_id=args.dir_id, nn_width=args.nn_width, nn_depth=args.nn_depth)
    north_star = (args.pos_id, args.dir_id, args.nn_width, args.nn_depth) == (3, 0, 64, 6)
    nrc = api.NeuralRadianceCache(cfg)
    ren = api.NrcHpmRenderer(local_w, gh, True, cam, cfg, scene, nrc, tile=tile)
    if use_dist and args.train:
        # RCCL all-reduce of the MLP gradients every training step: issued by the library itself on its training stream; if the
        # library cannot bring up its own communicator the same exchange goes through torch.distributed's RCCL communicator
        try:
            parallel.attach_gradient_allreduce(nrc, world)
        except RuntimeError as e:
            print("warning: native RCCL exchange unavailable (%s); using the torch.distributed hook" % e, file=sys.stderr)
            parallel.attach_gradient_allreduce(nrc, world, native=False)
    randoms = sc.frame_randoms((args.steps + args.warmup) * spp + 8, seed=1337)
    ri = [0]

    def step():
        ren.SetBlend(True)            # progressive blend restarts: sub-frame i has blendFactor 1/(i+1)
        for _ in range(spp):
            ren.SetFrameRandom(randoms[ri[0] % len(randoms)])
            ri[0] += 1
            ren.Render(None, bool(args.train))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ren.StageStats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    stats = ren.StageStats(reset=True)
    if use_dist:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = nrc.GetLoss() if args.train else None
    samples = float(local_w) * gh * spp * args.steps * world
    value = samples / dt / 1e6
    ms_per_step = dt / args.steps * 1e3

    # ---- integrator traffic model: density fetches counted on the device for one extra (untimed) sub-frame
    ren.CountFetches(True)
    ren.SetFrameRandom(randoms[0])
    ren.Render(None, False)
    torch.cuda.synchronize()
    n_fetch = ren.CountFetches(False)
    n_px = local_w * gh

    out = None
    if rank == 0:
        # ---- per-launch figures.  k_infer: event-timed loop on the launch stream (same buffers the frame uses); k_gen_rays:
        # the renderer's own HIP events around the launch, averaged over the timed region.  `traffic` = HBM-side bytes per
        # launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE, gfx950
        # corrections of MI355X_MICROARCH.md) -- only quoted when this run is the profiled workload.
        n_inf = n_px
        d_out = ren.Buffer("infer_output")

        def time_infer(d_in):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                nrc.Infer(d_in, d_out, True)
            reps = 20
            e0.record()
            for _ in range(reps):
                nrc.Infer(d_in, d_out, True)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        # (a) dense pass over uniformly random queries (positions in the quirk-Q3 range): the conservative figure -- zero or
        #     repetitive operands let the chip hold a higher clock (cdna_hip_programming.md rule 25);
        # (b) dense pass over this frame's own query buffer (78 % all-zero queries of unscattered pixels)
        g = torch.Generator(device="cuda").manual_seed(1)
        rnd = torch.rand((n_inf, 5), device="cuda", generator=g)
        rnd[:, :3] += 31.0
        mlp_ms = time_infer(rnd)
        mlp_ms_frame = time_infer(ren.Buffer("infer_input"))
        mlp_tflops = MLP_FLOP_PER_SAMPLE * n_inf / (mlp_ms * 1e-3) / 1e12
        if not (0.0 < mlp_tflops < MFMA_F16_PEAK_TFLOPS):
            raise RuntimeError("MLP timing is not physical (%.1f TFLOP/s): the events did not bracket the kernel's stream" % mlp_tflops)
        gen_ms = stats["gen_rays"]
        gen_store_bytes = n_px * (16 + 4 + 16 + 16 + 20)                    # primary, info, origin, dir, query
        gen_bytes = n_fetch * 1.0 + gen_store_bytes
        traffic = {}
        tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tf) and (W, H, args.volume, world) == (1920, 1080, 256, 1):
            with open(tf) as f:
                traffic = {k: v["traffic_bytes"] for k, v in json.load(f)["kernels"].items()}
        # VALU issue utilisation of k_gen_rays from the committed SQ counter pass (profiles/r01_pmc_sq_counters.txt): busy
        # quad-cycles x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) -- the bound that actually limits the integrator
        valu_busy = None
        sq = os.path.join(ROOT, "profiles", "r01_pmc_sq_counters.txt")
        if os.path.exists(sq) and (W, H, args.volume, world) == (1920, 1080, 256, 1):
            vals, on = {}, False
            for line in open(sq):
                if not line.startswith(" "):
                    on = line.startswith("k_gen_rays")
                elif on and len(line.split()) == 2:
                    vals[line.split()[0]] = float(line.split()[1])
            if "SQ_ACTIVE_INST_VALU" in vals and vals.get("GRBM_GUI_ACTIVE"):
                valu_busy = vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        dominant_is_gen = gen_ms >= mlp_ms
        roof_mlp = dict(bound="mfma", kernel="k_infer (fused encode + 6x64 MLP)", achieved=mlp_tflops, peak=MFMA_F16_PEAK_TFLOPS,
                        unit="TFLOP/s", frac=mlp_tflops / MFMA_F16_PEAK_TFLOPS, traffic=traffic.get("k_infer"),
                        algorithmic_bytes=MLP_BYTES_PER_SAMPLE * n_inf, ms_per_launch=mlp_ms, samples_per_launch=n_inf,
                        data="uniform random queries",
                        on_frame_queries=dict(ms_per_launch=mlp_ms_frame,
                                              achieved=MLP_FLOP_PER_SAMPLE * n_inf / (mlp_ms_frame * 1e-3) / 1e12,
                                              frac=MLP_FLOP_PER_SAMPLE * n_inf / (mlp_ms_frame * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS))
        roof_gen = dict(bound="hbm", kernel="k_gen_rays (delta/ratio tracking path integrator; ALU/latency-bound, quoted against HBM)",
                        achieved=gen_bytes / (gen_ms * 1e-3) / 1e9 if gen_ms > 0 else 0.0, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=(gen_bytes / (gen_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if gen_ms > 0 else 0.0,
                        traffic=traffic.get("k_gen_rays"), algorithmic_bytes=gen_bytes,
                        ms_per_launch=gen_ms, fetches_per_pixel=n_fetch / n_px, valu_issue_busy_pmc=valu_busy)
        out = {
            "metric": "Msamples/s + ms/frame at 1080p, 256^3 cloud (NRC path)", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_frame": ms_per_step / spp, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (fp16 MFMA operands, fp32 accumulate; fp32 integrator)", "data": "synthetic",
            "config": {"workload": "configs[1]+[2]: %dx%d per GPU (global %dx%d, interleaved column tiles), %d^3 seeded fBm cloud, "
                                   "%d spp/step, NRC 6x64 Frequency(12)+OneBlob(4), HDR sky env map, scene preset 4, "
                                   "train=%d (16384 train rays + 1 Adam step per sub-frame)" % (W, H, gw, gh, args.volume, spp, args.train),
                       "width": W, "height": H, "spp": spp, "volume": args.volume, "train": args.train,
                       "parallelism": "pixel-column tiles x%d%s" % (world, " + RCCL grad all-reduce" if world > 1 and args.train else "")},
            "stage_ms": {k: stats[k] for k in ("gen_rays", "prep_train", "train", "infer", "render", "total")},
            "loss": loss,
            "roofline": roof_gen if dominant_is_gen else roof_mlp,
            "roofline_mlp": roof_mlp,
            "roofline_integrator": roof_gen,
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene, W, H)
        out["gpu_vs_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    ren.Destroy()
    nrc.Destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
