"""Headless counterpart of the reference's main loop (src/main.cu:152-419) for the hot path.

    python -m nrc_hpm_renderer_amd.cli [17 positional AppConfig args] [--frames N] [--vdb FILE | --volume N] ...

The 17 positional arguments are the reference's (src/AppConfig.cpp:154-182); without any, the reference's own defaults apply
(src/main.cu:428-439: HashGrid position encoding, OneBlob direction encoding, 6x64, 4 train batches of 2^14).  Every frame is `NrcHpmRenderer::Render(queue, true)` (src/main.cu:287); with
--benchmark each frame is also evaluated like `Benchmark()` (src/main.cu:140-150): the NRC image without training from the same
camera against a reference image, one line `frame mse relBias CV` in `output/ <config-name>/log.txt` (the literal space is the
reference's, src/main.cu:240,446).  The reference image is `--reference FILE.exr` or, like Reference::GenRefImages
(src/Reference.cpp:566-606), a blended MC render (PATH_LENGTH 64, --ref-frames frames).

Multi-GPU (new): `--gpus N` starts N ranks (python -m torch.distributed.run; or start the module under that
launcher yourself).  The frame is sharded by interleaved strips of 8 pixel columns, the MLP gradients are all-reduced every training
step, the per-frame metrics are reduced over the ranks (five fp64 sums, nrc_compare_images_sharded) and `--export` writes the WHOLE
frame from rank 0 (nrc_renderer_gather_frame).  Rank 0 owns the log.  NRC_CLI_SHARED_GPU=1 rehearses the ranks on one device (gloo).
"""
import argparse
import math
import os
import sys

import numpy as np

DEFAULT_ARGV = ["RelativeL2Luminance", "Adam", "0.01", "0.99", "0", "0", "64", "6", "21", "14", "4", "4", "1.0", "1", "1", "0.0", "32"]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("config", nargs="*", help="17 positional AppConfig arguments (all or none)")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--vdb", default=None, help="OpenVDB FloatGrid file (Texture3D::FromVDB semantics)")
    ap.add_argument("--volume", type=int, default=256, help="edge of the procedural fBm cloud when no --vdb is given")
    ap.add_argument("--env", choices=["white", "black", "sky"], default="white")
    ap.add_argument("--benchmark", action="store_true")
    ap.add_argument("--reference", default=None, help="reference EXR (e.g. reference/4/0.exr of the reference repo)")
    ap.add_argument("--ref-frames", type=int, default=256)
    ap.add_argument("--output", default="output")
    ap.add_argument("--export", default=None, help="write the final NRC image to this EXR")
    ap.add_argument("--gpus", type=int, default=1, help="ranks the frame is sharded over (one GPU each)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "RANK" not in os.environ:
        # start the ranks before anything touches the GPU in this process (a process that has initialised HIP must not spawn them)
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "nrc_hpm_renderer_amd.cli"] + list(sys.argv[1:] if argv is None else argv)
        return subprocess.call(cmd, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    import torch
    from . import api, io_exr, io_vdb, parallel, scene as sc

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1) and "RANK" in os.environ:
        raise SystemExit("SkyRenderer ERROR: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    shared_gpu = os.environ.get("NRC_CLI_SHARED_GPU") == "1"

    if args.config and len(args.config) != 17:
        raise SystemExit("SkyRenderer ERROR: Argument count does not match requirements for AppConfig")
    cfg = api.AppConfig(["NRC-HPM-Renderer"] + (args.config or DEFAULT_ARGV))
    if args.vdb:
        vol, _ = io_vdb.from_vdb(args.vdb)
        density = sc.quantize_density(vol)
    else:
        density = sc.quantize_density(sc.fbm_cloud_volume(args.volume, seed=1337))
    env = {"white": None, "black": sc.black_env(), "sky": sc.procedural_sky()}[args.env]
    scene = sc.make_scene(density, scene_id=cfg.scene_id, env=env)
    W, H = args.width, args.height
    camera = sc.make_camera(aspect=W / H)            # src/main.cu:180-187
    torch.cuda.set_device(0 if shared_gpu else int(os.environ.get("LOCAL_RANK", "0")))
    tile, lw, cols = None, W, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if shared_gpu else "nccl", rank=rank, world_size=world)
        tile, lw = parallel.column_tile(rank, world, W, H), parallel.local_width(rank, world, W)
        cols = torch.from_numpy(parallel.rank_columns(rank, world, W)).cuda()
        if (lw * H) % 16:
            raise SystemExit("SkyRenderer ERROR: rank %d's %d x %d tile is not a multiple of 16 pixels" % (rank, lw, H))

    nrc = api.NeuralRadianceCache(cfg)
    if world > 1:
        # one exchange step per train batch: the library's own RCCL all-reduce (one device per rank), or the gloo hook of a rehearsal
        parallel.attach_gradient_allreduce(nrc, world, native=not shared_gpu)
        if shared_gpu:
            nrc.SetCollectiveHooks(rank, world)      # (with the native communicator the gather / metric reduction use RCCL too)
    nrc_renderer = api.NrcHpmRenderer(lw, H, False, camera, cfg, scene, nrc, tile=tile)
    out_dir = os.path.join(args.output, " " + cfg.GetName())
    log = None
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        log = open(os.path.join(out_dir, "log.txt"), "w")

    ref = None
    if args.benchmark:
        if args.reference:
            ref = torch.from_numpy(io_exr.read_exr(args.reference)).cuda()
            if ref.shape[0] != H or ref.shape[1] != W:
                raise SystemExit("SkyRenderer ERROR: reference image resolution mismatch")     # src/Reference.cpp:627
            if cols is not None:
                ref = ref[:, cols, :].contiguous()      # this rank's columns
        else:
            mc = api.McHpmRenderer(lw, H, 64, True, camera, scene, tile=tile)      # (every rank blends its own columns)
            for _ in range(args.ref_frames):
                mc.Render()
            ref = mc.GetImage().clone()
            mc.Destroy()
        eval_renderer = api.NrcHpmRenderer(lw, H, False, camera, cfg, scene, nrc, tile=tile)        # Reference::CompareNrc renders with train=false

    # A NaN / Inf loss ends the run (src/main.cu:380-384).  With several ranks the decision is COLLECTIVE: the poll below never blocks, so
    # ranks can see the (all-reduced, identical) bad value at different frames -- a rank that left the loop on its own would enter the
    # collective export while its peers are still inside Render()'s gradient exchange, and the job would hang.  Every rank therefore
    # only NOTES a bad loss, the flag is max-reduced every kStopEvery frames (and at the last one), and all ranks leave at that frame.
    kStopEvery = 8
    bad, failed = False, False
    for frame in range(args.frames):
        nrc_renderer.Render(None, True)
        loss = nrc.GetLoss(wait=False)          # src/main.cu:376: polled every frame, never blocks the frame pipeline
        if math.isnan(loss) or math.isinf(loss):                    # src/main.cu:380-384
            print("SkyRenderer ERROR: NRC Loss is %s" % loss, file=sys.stderr)
            bad = True
        if world == 1:
            failed = bad
        elif frame % kStopEvery == kStopEvery - 1 or frame == args.frames - 1:
            flag = torch.tensor([1.0 if bad else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = bool(flag.item() > 0.0)
        if failed:
            break
        if ref is not None:
            eval_renderer.Render(None, False)
            if world > 1:
                r = api.CompareImagesSharded(nrc, ref, eval_renderer.GetImage().contiguous())      # collective: the whole frame's Result
            else:
                r = api.CompareImages(ref, eval_renderer.GetImage())
            rel_bias = (r["own_mean"] - r["ref_mean"]) / r["ref_mean"] if r["ref_mean"] else 0.0
            cv = math.sqrt(max(r["own_var"], 0.0)) / r["own_mean"] if r["own_mean"] else 0.0
            if log is not None:
                log.write("%d %g %g %g\n" % (frame, r["mse"], rel_bias, cv))
        if rank == 0 and (frame % 16 == 0 or frame == args.frames - 1):
            print("frame %d: loss %.5f, %.3f ms" % (frame, loss, nrc_renderer.GetFrameTimeMS()))
    if log is not None:
        log.close()
    if args.export and not failed:
        nrc_renderer.ExportOutputImageToFile(None, args.export)      # sharded: collective, rank 0 writes the whole frame
    nrc_renderer.Destroy()
    if ref is not None:
        eval_renderer.Destroy()
    nrc.Destroy()
    if world > 1:
        if not failed:
            dist.barrier()
        dist.destroy_process_group()
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
