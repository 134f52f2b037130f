"""The product at N = 2: two fresh ranks (torch.distributed.run, gloo, both on cuda:0 -- the GPU box has one device) each render
their tile of interleaved 8-column strips with training on.  Checked here, in the parent, against single-process runs:
  * the replicas stay bit-identical (weights, EMA weights, Adam moments, step, per-step losses);
  * the all-reduced gradient of a step == the gradient one process computes for the two ranks' train rays as ONE batch;
  * the tiles' primary images and -- rendered with the replicas' common weights -- their composited frames reassemble to the
    frame a single GPU renders, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_train_identical_replicas_and_reassemble_the_frame(api, sc, torch_gpu, tmp_path):
    from nrc_hpm_renderer_amd import parallel
    out = str(tmp_path / "two")
    port = 29900 + (os.getpid() % 300)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "workers", "dist_two_rank_worker.py"), out],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (np.load(out + ".%d.npz" % k) for k in range(2))
    assert int(a["world"]) == 2 and int(b["rank"]) == 1
    # replicas: identical after every step
    for key in ("w", "ema", "m", "v", "w_prev", "ema_prev", "grad"):
        assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), key
    assert int(a["step"]) == int(b["step"]) == 6 and np.array_equal(a["losses"], b["losses"]) and np.isfinite(a["losses"]).all()
    assert not np.array_equal(a["w"], a["w_prev"])                     # training moved the weights
    assert not np.array_equal(a["train_in"], b["train_in"])            # each rank trained on its own tile's rays
    # the summed gradient of the last step == one process, both ranks' rays as one batch (same weights, same global normaliser)
    W, H = 256, 96
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=10, log2_infer_batch_size=14)
    full = api.NeuralRadianceCache(cfg)
    full.load_state_dict(dict(w=a["w_prev"], ema=a["ema_prev"], m=a["m_prev"], v=a["v_prev"], step=int(a["step_prev"])))
    x = torch_gpu.from_numpy(np.concatenate([a["train_in"], b["train_in"]])).cuda()
    t = torch_gpu.from_numpy(np.concatenate([a["train_target"], b["train_target"]])).cuda()
    assert x.shape[0] == 1024
    full.Backward(x, t)
    g_full = full.GetParams(4)
    assert np.linalg.norm(a["grad"] - g_full) <= 2e-5 * np.linalg.norm(g_full)      # fp32 sums in a different order
    assert abs(full.GetLoss() - a["losses"][-1]) <= 1e-5 * abs(a["losses"][-1])
    full.OptimizerStep()
    assert np.linalg.norm(full.GetParams(0) - a["w"]) <= 1e-4 * np.linalg.norm(a["w"] - a["w_prev"]) + 1e-7 * np.linalg.norm(a["w"])
    # the frame: tiles of the sharded run == the single-GPU frame, with the replicas' weights
    vol = sc.quantize_density(sc.fbm_cloud_volume(48, seed=3))
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(32, 16))
    cam = sc.make_camera(aspect=W / H)
    one = api.NeuralRadianceCache(api.AppConfig(train_batch_count=1, log2_train_batch_size=9, log2_infer_batch_size=14))
    one.load_state_dict(dict(w=a["w"], ema=a["ema"], m=a["m"], v=a["v"], step=int(a["step"])))
    ren = api.NrcHpmRenderer(W, H, False, cam, one.cfg, scene, one)
    frs = a["frame_randoms"]
    ren.SetFrameRandom(frs[5])
    ren.Render(None, False)
    prim = ren.Buffer("primary").cpu().numpy().reshape(H, W, 4).copy()
    assert np.array_equal(parallel.gather_columns([a["primary"], b["primary"]], W).view(np.uint32), prim.view(np.uint32))
    ren.SetFrameRandom(frs[6])
    ren.Render(None, False)
    img = ren.GetImage().cpu().numpy()
    got = parallel.gather_columns([a["img"], b["img"]], W)
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))
    assert (prim[..., 3] < 1.0).mean() > 0.02 and img[..., :3].std() > 0.01      # the cache contributes to visible pixels
    # the product's own assembly (nrc_renderer_gather_frame: all-gather + de-interleave kernel): the whole frame on BOTH ranks, the file
    # rank 1 exported, and the sharded metric reduction == nrc_compare_images of the single-GPU frame
    assert np.array_equal(a["gathered"].view(np.uint32), img.view(np.uint32)) and np.array_equal(b["gathered"].view(np.uint32), img.view(np.uint32))
    from nrc_hpm_renderer_amd import io_exr
    sys.path.insert(0, os.path.join(ROOT, "tests", "workers"))
    import dist_two_rank_worker as worker
    exr = io_exr.read_exr(out + ".gathered.exr")
    assert exr.shape == (H, W, 4) and np.array_equal(exr.view(np.uint32), img.view(np.uint32))
    ref = torch_gpu.from_numpy(worker.reference_image(W, H)).cuda()
    want = api.CompareImages(ref, ren.GetImage())
    want5 = np.asarray([want[k] for k in ("mse", "ref_mean", "own_mean", "own_var", "valid")], np.float32)
    assert np.array_equal(a["result"], b["result"]) and want5[4] > 1000 and a["result"][4] == want5[4]
    assert np.allclose(a["result"], want5, rtol=3e-7, atol=0.0)                   # fp64 sums in another order, rounded to fp32
    ren.Destroy()
    one.Destroy()
    full.Destroy()


@pytest.mark.gpu
def test_bench_self_launches_its_ranks(torch_gpu):
    """`python bench.py --gpus 2` as the driver starts it (no launcher environment): refuses on a one-GPU box instead of running one
    rank under a two-GPU label; with NRC_BENCH_SHARED_GPU=1 (rehearsal: both ranks on cuda:0, gloo hook exchange) it starts its two
    ranks itself, relays ONE line with n_gpus 2, the weak-scaling figure and the configs[3] strong-scaling figure"""
    import json
    import torch
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NRC_BENCH_SHARED_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
        assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr
    env["NRC_BENCH_SHARED_GPU"] = "1"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["value"] > 1
    assert d["exchange"]["rccl_ranks"] == 2 and "hook" in d["exchange"]["path"]
    fa = d["exchange"]["frame_assembly"]                                # the product's own gather + metric reduction, timed
    assert fa["gather_frame_ms"] > 0 and fa["compare_sharded_ms"] > 0 and d["cpu_baseline"] is None and d["cpu_baseline_reason"]
    assert "2712x1528" in d["config"]["workload"]      # the one-GPU view at sqrt(2) x the resolution: the same aspect, 2 x the pixels (- 0.08 %)
    c4 = d["strong_scaling_c4"]
    # (the rehearsal's RATES mean nothing -- two ranks share one GPU and every gradient exchange is a host-side gloo all-reduce, tens to
    # hundreds of milliseconds when the host is busy --: the line's arithmetic and labels are what is checked)
    assert c4["scaling"] == "strong" and c4["value"] > 1 and "1920 columns per rank" in c4["workload"] and "8192 per rank" in c4["workload"]
    assert abs(c4["ms_per_step"] - 3840 * 2160 * 8 / c4["value"] / 1e3) < 1e-6 * c4["ms_per_step"] + 1e-9
    s2 = d["strong_scaling_1080p"]                   # the metric's own 1920x1080 frame split over the two ranks
    assert s2["scaling"] == "strong" and s2["value"] > 1 and "960 columns per rank" in s2["workload"] and "8192 per rank" in s2["workload"]
    assert abs(s2["ms_per_step"] - 1920 * 1080 * 4 / s2["value"] / 1e3) < 1e-6 * s2["ms_per_step"] + 1e-9


@pytest.mark.gpu
def test_cli_two_ranks_log_the_whole_frames_metrics_and_export_the_whole_frame(torch_gpu, tmp_path):
    """`python -m nrc_hpm_renderer_amd.cli --gpus 2 --benchmark --export` (rehearsed on one device: NRC_CLI_SHARED_GPU=1, gloo): the
    main loop of src/main.cu:152-419 with the frame sharded over two ranks -- rank 0's log holds one `frame mse relBias CV` line per
    frame (metrics of the WHOLE frame, reduced over the ranks), and the exported EXR is the whole 256x96 frame with both ranks' columns"""
    from nrc_hpm_renderer_amd import io_exr
    env = dict(os.environ, NRC_CLI_SHARED_GPU="1", GPU_MAX_HW_QUEUES="8")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out, exr = str(tmp_path / "out"), str(tmp_path / "whole.exr")
    argv = ["RelativeL2Luminance", "Adam", "0.01", "0.99", "3", "0", "64", "6", "14", "9", "1", "4", "1.0", "1", "1", "0.0", "32"]
    cmd = [sys.executable, "-m", "nrc_hpm_renderer_amd.cli"] + argv + ["--frames", "5", "--width", "256", "--height", "96", "--volume", "32", "--env", "sky",
                                                                      "--benchmark", "--ref-frames", "8", "--output", out, "--export", exr, "--gpus", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    logs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f == "log.txt"]
    assert len(logs) == 1                                             # rank 0 owns the log
    rows = [ln.split() for ln in open(logs[0]).read().splitlines()]
    assert [int(x[0]) for x in rows] == list(range(5)) and all(len(x) == 4 and np.isfinite([float(v) for v in x[1:]]).all() for x in rows)
    assert all(float(x[1]) > 0.0 for x in rows)                        # an MSE of the whole frame
    img = io_exr.read_exr(exr)
    assert img.shape == (96, 256, 4)
    col_energy = np.abs(img[..., :3]).sum(axis=(0, 2))
    assert (col_energy > 0).all(), (np.nonzero(col_energy == 0)[0].tolist(), r.stdout[-1500:], r.stderr[-1500:])      # every strip of both ranks is there (the sky lights every column)
