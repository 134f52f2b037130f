"""The BASELINE.json configurations no other GPU test runs at their stated sizes, against the oracle:
   configs[3]  8 x MI355X pixel-tile shard, 3840x2160, 8 spp   -> one rank's column tile of the 4K frame, and all eight tiles
   configs[4]  512^3 heterogeneous smoke, 8x128 MLP + one-blob   -> the 2^27-voxel volume (24-bit index mads, raw-buffer range)
(the reference-default 2^19 HashGrid is in test_gpu_mlp.py::test_hashgrid_model_matches_oracle[reference-default-2^19])"""
import numpy as np
import pytest

from conftest import FRAME_RANDOM
from test_gpu_integrator import _nrc_setup, frac_close, rel, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def smoke512(sc):
    """configs[4]'s volume: the seeded 512^3 smoke plume bench.py --smoke-volume --volume 512 renders"""
    vol = sc.cached_volume("smoke", 512, seed=1337)
    assert vol.shape == (512, 512, 512) and vol.max() == 255
    return vol


def test_c5_smoke512_rows_bitwise_against_oracle(api, orc, sc, smoke512, torch_gpu):
    """1920x1080 on the 512^3 smoke: bands of gen_rays rows (colour, didScatter, packed queries) and one band of the MC renderer
    are bit-identical to the oracle -- voxel indices up to 2^27 - 1 through the 24-bit multiply-adds and the raw-buffer view"""
    W, H = 1920, 1080
    scene = sc.make_scene(smoke512, scene_id=4, env=sc.procedural_sky(64, 32))
    cfg, nrc, cam, ren = _nrc_setup(api, sc, scene, W, H, log2_train_batch_size=14, log2_infer_batch_size=21, nn_width=128, nn_depth=8)
    ren.SetFrameRandom(FRAME_RANDOM)
    ren.Render(None, True)
    prim = ren.Buffer("primary").cpu().numpy().reshape(H, W, 4)
    info = ren.Buffer("info").cpu().numpy().reshape(H, W)
    q = ren.Buffer("infer_input").cpu().numpy().reshape(W, H, 5)
    assert 0.02 < info.mean() < 0.9 and np.isfinite(nrc.GetLoss())
    scattered = 0
    for y0, y1 in ((0, 4), (300, 304), (538, 546), (1076, 1080)):
        o = orc.nrc_gen_rays(scene, cam, W, H, 1, 0.0, FRAME_RANDOM, rows=(y0, y1), threads=8)
        assert np.array_equal(info[y0:y1], o["info"][y0:y1])
        assert np.array_equal(prim[y0:y1].view(np.uint32), o["primary"][y0:y1].view(np.uint32))
        assert same_bits(q[:, y0:y1], o["infer_input"].reshape(W, H, 5)[:, y0:y1])
        scattered += int(o["info"][y0:y1].sum())
    assert scattered > 1000
    img = ren.GetImage().cpu().numpy()
    assert np.isfinite(img).all()
    ren.Destroy()
    nrc.Destroy()
    mc = api.McHpmRenderer(W, H, 32, False, cam, scene)
    mc.SetFrameRandom(FRAME_RANDOM)
    mc.Render()
    got = mc.GetImage().cpu().numpy()
    ref, _, _ = orc.mc_render(scene, cam, W, H, 32, FRAME_RANDOM, rows=(540, 544), threads=8)
    assert np.array_equal(got[540:544].view(np.uint32), ref[540:544].view(np.uint32))
    mc.Destroy()


def test_c5_smoke512_full_nrc_frames_8x128_match_oracle_pipeline(api, orc, sc, smoke512, torch_gpu):
    """configs[4] end to end at a size the oracle renders whole: two trained, blended NRC frames of the 512^3 smoke with the
    8x128 Frequency+OneBlob model -- gen_rays, train rays, ring buffer, inference, loss, Adam step, compositing"""
    W, H = 128, 80
    scene = sc.make_scene(smoke512, scene_id=4)
    cfg, nrc, cam, ren = _nrc_setup(api, sc, scene, W, H, nn_width=128, nn_depth=8)
    ren.SetBlend(True)
    onn = orc.nn_create(width=128, depth=8)
    tg = ren.TrainGrid()
    T = tg["tw"] * tg["th"]
    head_tail = np.zeros(2, np.uint32)
    ring = np.zeros((T, 6), np.float32)
    ring[:, 5] = 1.0
    ref = np.zeros((H, W, 4), np.float32)
    frs = sc.frame_randoms(2, seed=9)
    for f in range(2):
        ren.SetFrameRandom(frs[f])
        ren.Render(None, True)
        o = orc.nrc_gen_rays(scene, cam, W, H, 1, 0.0, frs[f], threads=8)
        tin, tgt = orc.nrc_prep_train(scene, W, H, tg["tw"], tg["th"], tg["x_dist"], tg["y_dist"], 1, 1, tg["ring_size"],
                                      frs[f], o["info"], o["origin"], o["dir"], head_tail, ring, threads=8)
        assert same_bits(ren.Buffer("primary").cpu().numpy().reshape(H, W, 4), o["primary"])
        assert same_bits(ren.Buffer("infer_input").cpu().numpy(), o["infer_input"])
        assert same_bits(ren.Buffer("train_input").cpu().numpy(), tin) and same_bits(ren.Buffer("train_target").cpu().numpy(), tgt)
        y = onn.forward(o["infer_input"], use_ema=True, mode=1)
        loss_ref = onn.backward(tin, tgt)
        onn.optimizer_step()
        ref = orc.nrc_composite(W, H, 1, 1.0 / (f + 1), o["primary"], o["info"], y, ref)
        assert abs(nrc.GetLoss() - loss_ref) < 5e-3 * abs(loss_ref)
    assert o["info"].mean() > 0.02
    img = ren.GetImage().cpu().numpy()
    assert np.isfinite(img).all()
    assert rel(img[..., :3], ref[..., :3]) < 2e-3
    assert frac_close(img.reshape(-1, 4), ref.reshape(-1, 4), atol=5e-3) >= 0.995
    assert rel(nrc.GetParams(1), onn.buffer(1)) < 3e-3
    ren.Destroy()
    nrc.Destroy()


# ------------------------------------------------------------------------------------------------ configs[3]
GW, GH, WORLD = 3840, 2160, 8


def test_c4_one_rank_tile_of_the_4k_frame_bitwise_against_oracle(api, orc, sc, torch_gpu):
    """rank 3 of 8 of the 3840x2160 frame (480 columns in interleaved strips of 8 x 2160 rows, training on, global loss
    normaliser): its gen_rays outputs equal columns 24..31, 88..95, ... of the oracle's full-width rows bit for bit"""
    from nrc_hpm_renderer_amd import parallel
    rank = 3
    vol = sc.cached_volume("cloud", 256, seed=1337)
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(64, 32))
    cam = sc.make_camera(aspect=GW / GH)
    lw = parallel.local_width(rank, WORLD, GW)
    cols = parallel.rank_columns(rank, WORLD, GW)
    assert lw == 480 and cols[:9].tolist() == [24, 25, 26, 27, 28, 29, 30, 31, 88]
    # configs[3] keeps the global train batch at 16 384 rays: 2 048 per rank
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=11, log2_infer_batch_size=21)
    nrc = api.NeuralRadianceCache(cfg)
    nrc.SetLossNormFactor(WORLD)
    ren = api.NrcHpmRenderer(lw, GH, True, cam, cfg, scene, nrc, tile=parallel.column_tile(rank, WORLD, GW, GH))
    ren.SetFrameRandom(FRAME_RANDOM)
    ren.Render(None, True)
    prim = ren.Buffer("primary").cpu().numpy().reshape(GH, lw, 4)
    info = ren.Buffer("info").cpu().numpy().reshape(GH, lw)
    q = ren.Buffer("infer_input").cpu().numpy().reshape(lw, GH, 5)
    assert np.isfinite(nrc.GetLoss()) and np.isfinite(ren.GetImage().cpu().numpy()).all()
    scattered = 0
    for y0, y1 in ((0, 2), (1078, 1084), (2158, 2160)):
        o = orc.nrc_gen_rays(scene, cam, GW, GH, 1, 0.0, FRAME_RANDOM, rows=(y0, y1), threads=8)
        assert np.array_equal(info[y0:y1], o["info"][y0:y1][:, cols])
        assert np.array_equal(prim[y0:y1].view(np.uint32), o["primary"][y0:y1][:, cols].view(np.uint32))
        assert same_bits(q[:, y0:y1], o["infer_input"].reshape(GW, GH, 5)[cols][:, y0:y1])
        scattered += int(o["info"][y0:y1][:, cols].sum())
    assert scattered > 500
    ren.Destroy()
    nrc.Destroy()


def test_c4_eight_tiles_equal_the_whole_4k_frame_at_8spp(api, sc, torch_gpu):
    """configs[3] on one GPU: the eight column tiles of the 3840x2160 frame, rendered one after another (8 blended sub-frames =
    8 spp, NRC inference + compositing on, identical weight replicas), reassemble to the frame a single renderer produces, bit
    for bit -- so the sharded frame IS the single-GPU frame; 8.3 M-query inference buffers on the whole-frame side"""
    from nrc_hpm_renderer_amd import parallel
    vol = sc.cached_volume("cloud", 256, seed=1337)
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(64, 32))
    cam = sc.make_camera(aspect=GW / GH)
    frs = sc.frame_randoms(8, seed=44)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=11, log2_infer_batch_size=21)

    def frames(width, tile):
        nrc = api.NeuralRadianceCache(cfg)
        ren = api.NrcHpmRenderer(width, GH, True, cam, cfg, scene, nrc, tile=tile)
        for f in range(8):
            ren.SetFrameRandom(frs[f])
            ren.Render(None, False)          # weights stay at their (identical) initial values on every "rank"
        img = ren.GetImage().cpu().numpy().copy()
        n_batches = nrc.GetInferBatchCount()
        ren.Destroy()
        nrc.Destroy()
        return img, n_batches

    whole, nb = frames(GW, None)
    assert nb == 4                               # 8 294 400 queries = 3 full batches of 2^21 + a remainder
    parts = []
    for r in range(WORLD):
        img, nb = frames(parallel.local_width(r, WORLD, GW), parallel.column_tile(r, WORLD, GW, GH))
        assert nb == 1
        parts.append(img)
    got = parallel.gather_columns(parts, GW)
    assert np.isfinite(whole).all() and (whole[..., 3] == 1.0).all()
    # bit for bit.  (Round 2 saw 3 pixels differ twice in ~60 runs and loosened this assertion; the cause is found -- a k_gen_rays wave
    # sharing its SIMD with waves of a higher issue priority could leave new_ray_dir with a different direction in lanes 48..63 --
    # and the camera kernels now run at a wave priority no lower than any neighbour's: DESIGN.md section 7, tests/test_gpu_stress.py.)
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
    assert whole[..., :3].std() > 0.01
