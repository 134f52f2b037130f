"""nrc_renderer_gather_frame / nrc_compare_images_sharded with RAGGED shards, in one process: three column tiles of a 200-pixel-wide
frame (strips of 8 columns: 9 / 8 / 8 strips = 72 / 64 / 64 columns) are rendered one after another; each tile's renderer then takes
part in a gather whose all-gather hook plays the other two ranks from their stored images.  The assembled frame on every "rank" is the
single-GPU frame, bit for bit; the sharded metric reduction gives the whole frame's Result."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gather_of_ragged_column_tiles_and_sharded_metrics(api, sc, torch_gpu):
    torch = torch_gpu
    from nrc_hpm_renderer_amd import parallel
    W, H, world = 200, 96, 3
    vol = sc.quantize_density(sc.fbm_cloud_volume(48, seed=3))
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(32, 16))
    cam = sc.make_camera(aspect=W / H)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=8, log2_infer_batch_size=14)
    frs = sc.frame_randoms(3, seed=21)

    def render(ren):
        ren.SetBlend(True)
        for f in frs:
            ren.SetFrameRandom(f)
            ren.Render(None, False)
        return ren.GetImage().clone()

    nrc1 = api.NeuralRadianceCache(cfg)
    one = api.NrcHpmRenderer(W, H, True, cam, cfg, scene, nrc1)
    full = render(one)
    lws = [parallel.local_width(r, world, W) for r in range(world)]
    assert lws == [72, 64, 64]
    caches, rens, imgs = [], [], []
    for r in range(world):
        c = api.NeuralRadianceCache(cfg)
        t = api.NrcHpmRenderer(lws[r], H, True, cam, cfg, scene, c, tile=parallel.column_tile(r, world, W, H))
        caches.append(c); rens.append(t); imgs.append(render(t))
    max_lw = max(lws)
    padded = [torch.zeros((H, max_lw, 4), device="cuda") for _ in range(world)]
    for r in range(world):
        padded[r][:, :lws[r]] = imgs[r]
    rng = np.random.default_rng(5)
    ref = rng.random((H, W, 4), dtype=np.float32)
    ref[..., 3] = (rng.random((H, W)) < 0.7).astype(np.float32)
    ref_full = torch.from_numpy(ref).cuda()
    want = api.CompareImages(ref_full, full)
    # raw sums of every rank's pixels, as the library's pass 1 / pass 2 leave them (fp64): what the other ranks contribute
    keep = []
    for r in range(world):
        def allgather(_user, send, recv, nbytes, stream, r=r):
            assert nbytes == H * max_lw * 16
            torch.cuda.synchronize()
            mine = api._wrap_device(send, nbytes, torch.float32, (H, max_lw, 4))
            assert torch.equal(mine, padded[r])                       # the library padded this rank's image to the widest rank's width
            out = api._wrap_device(recv, nbytes * world, torch.float32, (world, H, max_lw, 4))
            for k in range(world):
                out[k].copy_(padded[k])
            torch.cuda.synchronize()
            return 0

        def allreduce(_user, buf, n, stream):
            return 1                                                   # (not used by the gather)

        hooks = (api.ALLREDUCE_F64_HOOK(allreduce), api.ALLGATHER_HOOK(allgather))
        keep.append(hooks)
        api._check(caches[r].L.nrc_cache_set_collective_hooks(caches[r].h, C.c_int(r), C.c_int(world), hooks[0], hooks[1], None))
        got = rens[r].GatherFrame()
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int32), full.view(torch.int32)), "rank %d" % r
    # sharded metrics: the ranks' calls are played one after another -- rank r's all-reduce adds the sums the OTHER ranks left in theirs
    cols = [torch.from_numpy(parallel.rank_columns(r, world, W)).cuda() for r in range(world)]
    refs = [ref_full[:, cols[r], :].contiguous() for r in range(world)]
    partial = {}                                                       # (rank, call index) -> this rank's local sums

    def run(r, others):
        calls = []

        def allreduce(_user, buf, n, stream):
            torch.cuda.synchronize()
            t = api._wrap_device(buf, int(n) * 8, torch.float64, (int(n),))
            partial[(r, len(calls))] = t.clone()
            if others is not None:
                for k in range(world):
                    if k != r:
                        t += others[(k, len(calls))]
            calls.append(int(n))
            torch.cuda.synchronize()
            return 0

        hooks = (api.ALLREDUCE_F64_HOOK(allreduce), keep[r][1])
        keep.append(hooks)
        api._check(caches[r].L.nrc_cache_set_collective_hooks(caches[r].h, C.c_int(r), C.c_int(world), hooks[0], hooks[1], None))
        res = api.CompareImagesSharded(caches[r], refs[r], imgs[r].contiguous())
        assert calls == [4, 1]
        return res

    for r in range(world):                 # first round: collect every rank's pass-1 sums (pass 2 then uses a local mean: discarded)
        run(r, None)
    first = {k: v for k, v in partial.items() if k[1] == 0}
    # second round, pass 1 complete: the pass-2 sums are now taken around the GLOBAL mean
    for r in range(world):
        run(r, {**first, **{(k, 1): torch.zeros(1, dtype=torch.float64, device="cuda") for k in range(world)}})
    second = dict(partial)
    for r in range(world):
        res = run(r, second)
        for key in ("mse", "ref_mean", "own_mean", "own_var"):
            assert abs(res[key] - want[key]) <= 3e-7 * abs(want[key]), (r, key, res[key], want[key])
        assert res["valid"] == want["valid"] > 1000
    for x in rens + [one]:
        x.Destroy()
    for c in caches + [nrc1]:
        c.Destroy()


def test_unsharded_renderer_and_single_rank_cache_take_the_trivial_paths(api, sc, torch_gpu):
    """one rank: nrc_renderer_gather_frame copies the framebuffer, nrc_compare_images_sharded needs no communicator and equals
    nrc_compare_images up to the rounding of its fp64 folds"""
    torch = torch_gpu
    W, H = 128, 64
    vol = sc.quantize_density(sc.fbm_cloud_volume(32, seed=3))
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(32, 16))
    cam = sc.make_camera(aspect=W / H)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=8, log2_infer_batch_size=13)
    nrc = api.NeuralRadianceCache(cfg)
    ren = api.NrcHpmRenderer(W, H, False, cam, cfg, scene, nrc)
    ren.Render(None, True)
    img = ren.GetImage().clone()
    got = ren.GatherFrame()
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), img.view(torch.int32))
    rng = np.random.default_rng(9)
    ref = rng.random((H, W, 4), dtype=np.float32)
    ref[..., 3] = (rng.random((H, W)) < 0.5).astype(np.float32)
    ref = torch.from_numpy(ref).cuda()
    a, b = api.CompareImages(ref, img), api.CompareImagesSharded(nrc, ref, img)
    assert a["valid"] == b["valid"] > 1000
    for k in ("mse", "ref_mean", "own_mean", "own_var"):
        assert abs(a[k] - b[k]) <= 3e-7 * abs(a[k]), (k, a[k], b[k])
    ren.Destroy()
    nrc.Destroy()
