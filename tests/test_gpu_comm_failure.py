"""Failure detection on the exchange (include/nrc_hpm.h, nrc_cache_comm_status; SURVEY.md section 5 "RCCL error -> status code"), in one process
on one GPU: a tile renderer of a two-rank frame whose transport (collective hooks) fails, or whose collective never completes, must hand the
caller NRC_ERR_COMM -- api.CommError -- from GatherFrame / CompareImagesSharded instead of hanging, and every later exchange must fail at once."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_rank0(api, sc, W=64, H=32):
    from nrc_hpm_renderer_amd import parallel
    vol = sc.quantize_density(sc.fbm_cloud_volume(32, seed=3))
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(32, 16))
    cam = sc.make_camera(aspect=W / H)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=8, log2_infer_batch_size=12)
    nrc = api.NeuralRadianceCache(cfg)
    ren = api.NrcHpmRenderer(parallel.local_width(0, 2, W), H, False, cam, cfg, scene, nrc, tile=parallel.column_tile(0, 2, W, H))
    ren.Render(None, False)
    return nrc, ren


def test_a_failing_transport_is_a_comm_error_and_stays_one(api, sc, torch_gpu):
    nrc, ren = make_rank0(api, sc)
    calls = []

    def allreduce(_user, buf, n, stream):
        calls.append("allreduce")
        return 0

    def allgather(_user, send, recv, nbytes, stream):
        calls.append("allgather")
        return 7                                  # "the transport failed"

    nrc.SetRawCollectiveHooks(0, 2, allreduce, allgather)
    nrc.CommStatus()                              # healthy so far
    with pytest.raises(api.CommError, match="all-gather hook failed.*aborted"):
        ren.GatherFrame()
    assert calls == ["allgather"]
    # the exchange is over: nothing calls the transport again, every collective entry point fails at once with the same class
    with pytest.raises(api.CommError):
        nrc.CommStatus()
    with pytest.raises(api.CommError):
        ren.GatherFrame()
    a = torch_gpu.zeros((32, 32, 4), device="cuda")
    with pytest.raises(api.CommError):
        api.CompareImagesSharded(nrc, a, a)
    with pytest.raises(api.CommError):
        ren.Render(None, True)                    # a training frame would enqueue the gradient exchange
    assert calls == ["allgather"]
    ren.Render(None, False)                       # rendering without the exchange still works
    ren.Destroy()
    nrc.Destroy()


def test_a_collective_that_does_not_complete_runs_into_the_deadline(api, sc, torch_gpu):
    """the transport `succeeds` but leaves ~1.5 s of work on the library's stream (a bounded spin kernel: the stand-in for a collective whose
    peer never arrives); with a 150 ms deadline the metric reduction returns CommError well before that work ends -- no hang"""
    torch = torch_gpu
    nrc, ren = make_rank0(api, sc)

    def allreduce(_user, buf, n, stream):
        with torch.cuda.stream(torch.cuda.ExternalStream(int(stream or 0))):
            torch.cuda._sleep(int(3.0e9))         # ~1.5 s at 2 GHz; ends by itself
        return 0

    def allgather(_user, send, recv, nbytes, stream):
        return 0

    nrc.SetRawCollectiveHooks(0, 2, allreduce, allgather)
    nrc.SetCommTimeoutMs(150)
    a = torch.rand((32, 32, 4), device="cuda")
    side = torch.cuda.Stream()                    # (an explicit stream: handle 0 does not name the same stream to torch and to the library)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with pytest.raises(api.CommError, match="did not complete within 150 ms"):
        api.CompareImagesSharded(nrc, a, a, stream=side)
    dt = time.perf_counter() - t0
    assert 0.1 < dt < 1.0, dt
    with pytest.raises(api.CommError):
        nrc.CommStatus()
    torch.cuda.synchronize()                      # the spin kernel ends; nothing of the library is left waiting on it
    ren.Destroy()
    nrc.Destroy()


def test_without_a_deadline_and_with_a_healthy_transport_nothing_changes(api, sc, torch_gpu):
    torch = torch_gpu
    nrc, ren = make_rank0(api, sc)

    def allreduce(_user, buf, n, stream):
        return 0                                  # (one rank's sums are the frame's: the other rank contributes nothing)

    def allgather(_user, send, recv, nbytes, stream):
        torch.cuda.synchronize()
        api._wrap_device(recv, nbytes * 2, torch.uint8, (2, nbytes))[0].copy_(api._wrap_device(send, nbytes, torch.uint8, (nbytes,)))
        api._wrap_device(recv, nbytes * 2, torch.uint8, (2, nbytes))[1].zero_()
        torch.cuda.synchronize()
        return 0

    nrc.SetRawCollectiveHooks(0, 2, allreduce, allgather)
    nrc.SetCommTimeoutMs(0)
    full = ren.GatherFrame()
    assert full.shape == (32, 64, 4)
    own = ren.GetImage()
    # rank 0 of two renders the strips 0, 2, 4, ... of 8 columns
    assert torch.equal(full[:, 0:8], own[:, 0:8]) and torch.equal(full[:, 16:24], own[:, 8:16])
    nrc.CommStatus()
    ren.Destroy()
    nrc.Destroy()


def test_xcd_aware_mappings_are_placement_only(api, sc, torch_gpu, monkeypatch):
    """on a device that does not report eight XCDs the XCD-aware launch order and the level-per-XCD mappings of the HashGrid kernels are
    off (csrc/nrc_common.hpp device_xcds); they only place work, so the frames and the trained weights are the same bit for bit"""
    torch = torch_gpu
    W, H = 128, 64
    vol = sc.quantize_density(sc.fbm_cloud_volume(32, seed=3))
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(32, 16))
    cam = sc.make_camera(aspect=W / H)
    frs = sc.frame_randoms(6, seed=2)

    def run():
        cfg = api.AppConfig(pos_id=0, hashgrid_log2_size=12, nn_depth=3, train_batch_count=1, log2_train_batch_size=9, log2_infer_batch_size=13)
        nrc = api.NeuralRadianceCache(cfg)
        ren = api.NrcHpmRenderer(W, H, True, cam, cfg, scene, nrc)
        for f in frs:
            ren.SetFrameRandom(f)
            ren.Render(None, True)
        img, w = ren.GetImage().clone(), nrc.GetParams(0)[:nrc.ParamCount() - 0]
        ren.Destroy()
        nrc.Destroy()
        return img, w

    img8, w8 = run()
    monkeypatch.setenv("NRC_DEBUG", "assume_xcds=4")
    img4, w4 = run()
    # (HashGrid training is bitwise repeatable since round 5 -- the table gradient's sums are exact --, so the comparison is strict)
    assert np.array_equal(w8, w4)
    assert torch.equal(img8, img4)
