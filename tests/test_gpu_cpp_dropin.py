"""The C++ drop-in surface (include/nrc_hpm.hpp, namespace en) exercised by a real host program: tests/cpp/dropin_main.cpp
drives AppConfig(argv) -> NeuralRadianceCache -> NrcHpmRenderer -> Render(queue, true) like src/main.cu does, and must produce
the same frames, bit for bit, as the Python mirror over the same C ABI."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["RelativeL2Luminance", "Adam", "0.01", "0.99", "3", "0", "64", "6", "14", "10", "1", "4", "1.0", "1", "1", "0.0", "32"]


@pytest.mark.gpu
def test_cpp_host_program_matches_python_mirror(api, sc, torch_gpu, tmp_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    exe = entry.build_cpp_dropin()
    W, H, frames = 96, 64, 3
    vol = sc.quantize_density(sc.sphere_volume(32))
    scene = sc.make_scene(vol, scene_id=4, env=sc.white_env())
    cam = sc.make_camera(aspect=W / H)
    frs = sc.frame_randoms(frames, seed=5)
    with open(tmp_path / "scene.bin", "wb") as f:
        f.write(struct.pack("5I", W, H, *scene["dims"]))
        f.write(np.asarray(scene["env"], np.float32).reshape(-1)[:4].tobytes())
        f.write(np.asarray(frs, np.float32).tobytes())
        f.write(np.ascontiguousarray(scene["density"], np.uint8).tobytes())
    ref_root = str(tmp_path / "reference") + "/"
    r = subprocess.run([exe, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin"), str(frames)] + ARGS + [ref_root],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    assert "name RelativeL2Luminance_Adam_0.010000_0.990000_3_0_64_6_14_10_1_4_1.000000_1_1_0.000000_32" in r.stdout
    raw = np.fromfile(tmp_path / "out.bin", np.float32)
    n_img = H * W * 4
    loss_cpp, cam_cpp, light_cpp, img_cpp = raw[0], raw[1:20], raw[20:23], raw[23:23 + n_img].reshape(H, W, 4)
    # ---- en::Reference (src/Reference.cpp): ground truth generated on first use and exported, loaded back through the C++ EXR
    # reader, CompareNrc / CompareMc and the Result helpers; checked against the Python reader and CompareImages
    tail = raw[23 + n_img:]
    res = tail[:24].reshape(3, 8)
    ref_cpp, own_cpp = tail[24:24 + n_img].reshape(H, W, 4), tail[24 + n_img:24 + 2 * n_img].reshape(H, W, 4)
    assert "Reference folder for scene 4 was not found. Creating reference images" in r.stdout and r.stdout.count("MSE: ") == 3
    from nrc_hpm_renderer_amd import io_exr
    exr = io_exr.read_exr(os.path.join(ref_root, "4", "0.exr"))
    assert np.array_equal(exr.view(np.uint32), ref_cpp.view(np.uint32))          # C++ reader == Python reader == what was exported
    assert (ref_cpp[..., 3] > 0).mean() > 0.05 and np.isfinite(ref_cpp).all()
    got = api.CompareImages(torch_gpu.from_numpy(ref_cpp.copy()).cuda(), torch_gpu.from_numpy(own_cpp.copy()).cuda())
    mse, ref_mean, own_mean, own_var, valid, rel_bias, cv, rel_var = res[2]
    assert (mse, ref_mean, own_mean, own_var, valid) == tuple(np.float32(got[k]) for k in ("mse", "ref_mean", "own_mean", "own_var", "valid"))
    assert valid == (ref_cpp[..., 3] != 0).sum()
    assert rel_bias == np.float32((own_mean - ref_mean) / ref_mean) and cv == np.float32(np.sqrt(own_var) / own_mean) and rel_var == np.float32(own_var / ref_mean)
    assert 0 < res[0][0] < 10 and 0 < res[1][0] < 10 and abs(res[1][5]) < 0.5      # one noisy frame against 16 blended ones
    # en::Camera (glm's fp32 perspective * lookAt, cofactor inverse) agrees with the Python mirror's float64 construction to
    # fp32 rounding; en::HpmScene places the directional light like scene.dir_light_dir().  The frames are then compared bit for
    # bit for the camera the C++ program actually used.
    assert np.allclose(cam_cpp[:16], cam["inv_proj_view"], rtol=1e-5, atol=1e-6) and np.array_equal(cam_cpp[16:], cam["pos"])
    assert np.allclose(light_cpp, scene["dir_light_dir"], atol=1e-7)
    cam = dict(inv_proj_view=cam_cpp[:16].copy(), pos=cam_cpp[16:19].copy())
    scene = dict(scene, dir_light_dir=light_cpp.copy())

    cfg = api.AppConfig(["NRC-HPM-Renderer"] + ARGS)
    nrc = api.NeuralRadianceCache(cfg)
    ren = api.NrcHpmRenderer(W, H, False, cam, cfg, scene, nrc)
    ren.SetBlend(True)
    for k in range(frames):
        ren.SetFrameRandom(frs[k])
        ren.Render(None, True)
        loss = nrc.GetLoss()
    img = ren.GetImage().cpu().numpy()
    assert np.array_equal(img.view(np.uint32), img_cpp.view(np.uint32))
    assert np.float32(loss) == loss_cpp
    assert img[..., 3].min() == 1.0 and np.isfinite(img).all() and img[..., :3].max() > 0
    ren.Destroy()
    nrc.Destroy()


@pytest.mark.gpu
def test_cpp_sharded_reference_and_gather_over_host_hooks(api, sc, torch_gpu, tmp_path):
    """tests/cpp/sharded_main.cpp: two host threads as two ranks of a 200x64 frame (strips of 8 columns: 104 / 96 columns) over a
    host-staged transport installed with NeuralRadianceCache::SetCollectiveHooks -- en::Reference(..., &tile, &nrc)::CompareNrc gives
    both ranks the WHOLE frame's Result (== nrc_compare_images of the single-GPU frame up to the rounding of the fp64 sums),
    NrcHpmRenderer::GatherFrame gives both ranks the single-GPU frame bit for bit, and the collective ExportOutputImageToFile writes it"""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    from nrc_hpm_renderer_amd import io_exr
    exe = entry.build_cpp_sharded()
    W, H = 200, 64
    vol = sc.quantize_density(sc.sphere_volume(32))
    scene = sc.make_scene(vol, scene_id=4, env=sc.white_env())
    with open(tmp_path / "scene.bin", "wb") as f:
        f.write(struct.pack("5I", W, H, *scene["dims"]))
        f.write(np.asarray(scene["env"], np.float32).reshape(-1)[:4].tobytes())
        f.write(np.ascontiguousarray(scene["density"], np.uint8).tobytes())
    # a ground-truth image of the GLOBAL size where Reference expects it: <root>/<scene id>/0.exr
    rng = np.random.default_rng(17)
    ref = rng.random((H, W, 4), dtype=np.float32)
    ref[..., 3] = (rng.random((H, W)) < 0.6).astype(np.float32)
    ref_root = str(tmp_path / "reference") + "/"
    os.makedirs(os.path.join(ref_root, "4"))
    io_exr.write_exr(os.path.join(ref_root, "4", "0.exr"), ref, compression="zip")
    exr_out = str(tmp_path / "whole.exr")
    r = subprocess.run([exe, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin"), ref_root, exr_out] + ARGS, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "sharded ok" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
    raw = np.fromfile(tmp_path / "out.bin", np.float32)
    cam = dict(inv_proj_view=raw[:16].copy(), pos=raw[16:19].copy())      # en::Camera's fp32 matrices: the counterpart uses exactly these
    scene = dict(scene, dir_light_dir=raw[19:22].copy())
    res = raw[22:32].reshape(2, 5)
    frames = raw[32:].reshape(2, H, W, 4)
    # the single-GPU counterpart through the Python mirror: the frame CompareNrc renders is the renderer's first unpinned frame from
    # the reference camera (same seed, same draw on every tile); the gathered frame was rendered with the pinned random
    cfg = api.AppConfig(["NRC-HPM-Renderer"] + ARGS)
    nrc = api.NeuralRadianceCache(cfg)
    ren = api.NrcHpmRenderer(W, H, False, cam, cfg, scene, nrc)
    ren.Render(None, False)
    want = api.CompareImages(torch_gpu.from_numpy(ref).cuda(), ren.GetImage())
    want5 = np.asarray([want[k] for k in ("mse", "ref_mean", "own_mean", "own_var", "valid")], np.float32)
    assert np.array_equal(res[0], res[1]) and res[0][4] == want5[4] == (ref[..., 3] != 0).sum()
    assert np.allclose(res[0], want5, rtol=3e-7, atol=0.0), (res[0], want5)
    ren.SetCamera(None, cam)
    ren.SetFrameRandom([0.6180339887, 0.4142135623, 0.7320508075, 0.2360679775])
    ren.Render(None, False)
    img = ren.GetImage().cpu().numpy()
    assert np.array_equal(frames[0].view(np.uint32), img.view(np.uint32)) and np.array_equal(frames[1].view(np.uint32), img.view(np.uint32))
    assert np.array_equal(io_exr.read_exr(exr_out).view(np.uint32), img.view(np.uint32))
    assert np.isfinite(img).all() and (np.abs(img[..., :3]).sum(axis=(0, 2)) > 0).all()
    ren.Destroy()
    nrc.Destroy()
