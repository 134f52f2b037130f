"""Host-side logic of the package (no GPU): config surface, scene inputs, sharding helpers."""
import math

import numpy as np
import pytest


def test_appconfig_defaults_and_argv(api):
    c = api.AppConfig()
    assert (c.loss_fn, c.optimizer) == ("RelativeL2Luminance", "Adam")
    assert abs(c.learning_rate - 0.01) < 1e-9 and abs(c.ema_decay - 0.99) < 1e-7
    assert (c.pos_id, c.dir_id, c.nn_width, c.nn_depth) == (3, 0, 64, 6)
    assert (c.log2_infer_batch_size, c.log2_train_batch_size, c.train_batch_count) == (21, 14, 4)
    assert (c.scene_id, c.train_spp, c.primary_ray_length, c.train_ray_length) == (4, 1, 1, 32)
    # the reference's 18-entry argv (src/main.cu:432-439)
    argv = ["NRC-HPM-Renderer", "RelativeL2Luminance", "Adam", "0.01", "0.99", "3", "0", "64", "6", "21", "14", "4",
            "4", "1.0", "1", "1", "0.0", "32"]
    a = api.AppConfig(argv)
    assert a.GetName() == "RelativeL2Luminance_Adam_0.010000_0.990000_3_0_64_6_21_14_4_4_1.000000_1_1_0.000000_32"
    with pytest.raises(RuntimeError, match="Argument count"):
        api.AppConfig(argv[:-1])


def test_scene_presets_and_volume_size(sc):
    assert sc.SCENE_PRESETS[4] == (8.0, 0.0, 0.1, 0.6) and sc.SCENE_PRESETS[0][0] == 16.0
    s = sc.volume_size((126, 86, 154))
    assert np.allclose(s, [62.486, 42.649, 76.372], atol=2e-3)
    d = sc.dir_light_dir()
    assert abs(d[0]) < 1e-7 and abs(d[1] - 7.963e-4) < 1e-6 and abs(d[2] + 1.0) < 1e-6


def test_camera_matrix_unprojects_centre_ray(sc):
    cam = sc.make_camera(aspect=16 / 9)
    m = cam["inv_proj_view"].reshape(4, 4).T.astype(np.float64)           # column-major -> math
    p = m @ np.array([0.0, 0.0, 0.0, 1.0])
    p = p[:3] / p[3]
    d = p - np.array([64.0, 0, 0])
    d /= np.linalg.norm(d)
    assert np.allclose(d, [-1, 0, 0], atol=1e-6)
    # top of the image (ndc y = +1) is 30 degrees up
    p = m @ np.array([0.0, 1.0, 0.0, 1.0])
    p = p[:3] / p[3]
    d = p - np.array([64.0, 0, 0])
    d /= np.linalg.norm(d)
    assert abs(math.degrees(math.asin(d[1])) - 30.0) < 1e-3


def test_quantize_density_truncates(sc):
    v = np.zeros((2, 3, 4), np.float32)
    v[1, 2, 3] = 1.0
    v[0, 0, 0] = 0.999
    q = sc.quantize_density(v)
    assert q.shape == (4, 3, 2) and q[3, 2, 1] == 255 and q[0, 0, 0] == 254      # uint8(0.999*255) = 254


def test_column_tiles_partition_the_frame():
    """strips of `block` columns dealt round-robin: every global column belongs to exactly one rank, the local widths add up, and
    the tile tuple states the mapping of include/nrc_hpm.h (local column i -> (x_offset + (i // b) * x_stride) * b + i % b)"""
    from nrc_hpm_renderer_amd import parallel
    for block in (1, 2, 8, None):
        for world in (1, 2, 3, 4, 8):
            for gw in (16, 1920, 3841):
                cols = []
                for r in range(world):
                    x0, st, w_, h_, b = parallel.column_tile(r, world, gw, 4, block)
                    lw = parallel.local_width(r, world, gw, block)
                    mine = parallel.rank_columns(r, world, gw, block)
                    assert len(mine) == lw and (w_, h_) == (gw, 4)
                    assert [(x0 + (i // b) * st) * b + i % b for i in range(lw)] == mine.tolist()
                    cols += mine.tolist()
                assert sorted(cols) == list(range(gw))
    imgs = [np.full((2, parallel.local_width(r, 2, 5, 1), 1), r, np.float32) for r in range(2)]
    g = parallel.gather_columns(imgs, 5, 1)
    assert g[0, :, 0].tolist() == [0, 1, 0, 1, 0]
    imgs = [np.full((2, parallel.local_width(r, 2, 11, 2), 1), r, np.float32) for r in range(2)]
    assert parallel.gather_columns(imgs, 11, 2)[0, :, 0].tolist() == [0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1]
    assert parallel.DEFAULT_BLOCK == 8


def test_frame_randoms_deterministic(sc):
    a, b = sc.frame_randoms(5), sc.frame_randoms(5)
    assert np.array_equal(a, b) and a.shape == (5, 4) and (a >= 0).all() and (a < 1).all()


def test_hpm_scene_update_rotates_only_the_dynamic_preset(sc):
    """HpmScene::Update (src/HpmScene.cpp:56-76): preset 3's directional light advances azimuth by dt/2 (wrapped at 2*3.141)
    when the scene is dynamic; every other case leaves the scene alone"""
    import math
    vol = sc.quantize_density(sc.sphere_volume(16))
    s = sc.HpmScene(vol, scene_id=3, dynamic=True)
    d0 = np.array(s.scene["dir_light_dir"], np.float32)
    assert np.allclose(d0, sc.dir_light_dir(-1.57, 0.0))
    assert s.Update(0.5) and abs(s.azimuth - 0.25) < 1e-7
    d1 = np.array(s.scene["dir_light_dir"], np.float32)
    assert np.allclose(d1, sc.dir_light_dir(-1.57, 0.25)) and not np.allclose(d0, d1)
    assert abs(np.linalg.norm(d1) - 1.0) < 1e-5
    for _ in range(60):
        s.Update(0.5)
    assert 0.0 <= s.azimuth < 2.0 * 3.141 and abs(s.azimuth - math.fmod(0.25 * 61, 2.0 * 3.141)) < 1e-4
    assert not sc.HpmScene(vol, scene_id=3, dynamic=False).Update(0.5)
    assert not sc.HpmScene(vol, scene_id=4, dynamic=True).Update(0.5)


def test_bench_refuses_to_run_fewer_ranks_than_asked_for():
    """`python bench.py --gpus 2` without a launcher environment on a box with fewer than two GPUs: a non-zero exit and no result
    line -- never one rank under an N-GPU label.  The launching parent must not touch the HIP runtime (it starts the ranks): it counts
    devices in the KFD topology only and refuses there; without a KFD view (this container) it trusts --gpus and every rank without a
    device fails by itself"""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NRC_BENCH_SHARED_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=root)
    assert r.returncode != 0 and r.stdout.strip() == "" and ("refusing" in r.stderr or "has no device" in r.stderr)
    src = open(os.path.join(root, "bench.py")).read()
    launcher = src[src.index("def visible_gpus"):src.index("# ---------------------------------------------------------------------------------------------------------------- workloads")]
    assert "import torch" not in launcher and "torch.cuda" not in launcher


def test_weak_scaling_frames_keep_the_one_gpu_view():
    """bench.py --gpus N (default preset): the global frame is the one-GPU view at sqrt(N) x the resolution -- same aspect ratio (the medium
    covers the same share of every rank's pixels), whole 8-pixel tiles, N x 1080p pixels within 0.2 % -- and the ranks' interleaved strips
    add up to it; strong scaling (configs[3]) keeps the frame it was given"""
    import bench
    from nrc_hpm_renderer_amd import parallel
    assert bench.global_frame(1, 1920, 1080) == (1920, 1080) and bench.global_frame(4, 1920, 1080) == (3840, 2160)
    assert bench.global_frame(8, 3840, 2160, strong=True) == (3840, 2160)
    for n in (2, 3, 4, 6, 8):
        gw, gh = bench.global_frame(n, 1920, 1080)
        assert gw % 8 == 0 and gh % 8 == 0
        assert abs(gw / gh - 16.0 / 9.0) < 0.01 * 16.0 / 9.0
        assert abs(gw * gh / (n * 1920.0 * 1080.0) - 1.0) < 2e-3
        widths = [parallel.local_width(r, n, gw) for r in range(n)]
        assert sum(widths) == gw and max(widths) - min(widths) <= parallel.DEFAULT_BLOCK

