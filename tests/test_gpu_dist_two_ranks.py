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
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["value"] > 100
    assert d["exchange"]["rccl_ranks"] == 2 and "hook" in d["exchange"]["path"]
    assert "3840x1080" in d["config"]["workload"]
    c4 = d["strong_scaling_c4"]
    assert c4["scaling"] == "strong" and c4["value"] > 100 and "1920 columns per rank" in c4["workload"] and "8192 per rank" in c4["workload"]
    assert abs(c4["ms_per_step"] - 3840 * 2160 * 8 / c4["value"] / 1e3) < 1e-6 * c4["ms_per_step"] + 1e-9
