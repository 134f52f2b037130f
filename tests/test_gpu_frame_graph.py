"""Host-visible contracts of the frame graph that the reference's single queue gave for free: presenting every frame from a
consumer stream without tearing (release_frame), polling the loss every frame without draining the pipeline (asynchronous
GetLoss), the reference's start / finished semaphore pair as HIP events, and one cache driven by two renderers."""
import time

import numpy as np
import pytest

from conftest import FRAME_RANDOM
from test_gpu_integrator import _nrc_setup

pytestmark = pytest.mark.gpu


def test_consumer_stream_reads_every_frame_without_host_sync(api, sc, cloud16, torch_gpu):
    """GetImage(stream) + ReleaseImage(stream): a read-back stream copies every frame while the renderer runs ahead -- no host
    synchronisation anywhere in the loop; each copy equals the frame a synchronously read renderer produces.  (Compositing
    blends ONE framebuffer in place: without the release the next frame's compositing may overwrite it under the copy.)"""
    W, H, frames = 512, 288, 12
    scene = sc.make_scene(cloud16, scene_id=4)
    frs = sc.frame_randoms(frames, seed=33)
    side = torch_gpu.cuda.Stream()
    cfg, nrc, cam, ren = _nrc_setup(api, sc, scene, W, H)
    ren.SetBlend(True)
    pinned = [torch_gpu.empty((H, W, 4), dtype=torch_gpu.float32).pin_memory() for _ in range(frames)]
    for f in range(frames):
        ren.SetFrameRandom(frs[f])
        ren.Render(None, True)
        img = ren.GetImage(side)
        with torch_gpu.cuda.stream(side):
            for _ in range(4):                       # a slow consumer: several passes over the image before the copy
                tmp = img * 1.0
            pinned[f].copy_(tmp, non_blocking=True)
        ren.ReleaseImage(side)
    torch_gpu.cuda.synchronize()
    got = [p.numpy().copy() for p in pinned]
    ren.Destroy()
    nrc.Destroy()
    cfg, nrc, cam, ren = _nrc_setup(api, sc, scene, W, H)
    ren.SetBlend(True)
    for f in range(frames):
        ren.SetFrameRandom(frs[f])
        ren.Render(None, True)
        ref = ren.GetImage().cpu().numpy()
        assert np.array_equal(ref.view(np.uint32), got[f].view(np.uint32)), f
    ren.Destroy()
    nrc.Destroy()


def test_per_frame_loss_poll_does_not_drain_the_pipeline(api, sc, torch_gpu):
    """src/main.cu:303,376 polls GetLoss() every frame.  The non-blocking GetLoss (the C++ surface's default) returns the last
    COMPLETED step's loss and keeps the frame rate of an unpolled loop (within 10 %); it converges to the blocking value once the
    work has drained, and every polled value is one the blocking poll has seen."""
    W, H, frames = 1920, 1080, 240
    vol = sc.cached_volume("cloud", 256, seed=1337)
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
    frs = sc.frame_randoms(frames, seed=5)

    def loop(mode):
        cfg, nrc, cam, ren = _nrc_setup(api, sc, scene, W, H, log2_train_batch_size=14, log2_infer_batch_size=21)
        ren.SetBlend(True)
        seen = []
        for f in range(20):
            ren.SetFrameRandom(frs[f])
            ren.Render(None, True)
            if mode == "block":
                seen.append(nrc.GetLoss(wait=True))
        torch_gpu.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(frames):
            ren.SetFrameRandom(frs[f])
            ren.Render(None, True)
            if mode == "poll":
                seen.append(nrc.GetLoss(wait=False))
            elif mode == "block":
                seen.append(nrc.GetLoss(wait=True))
        t_enqueued = time.perf_counter() - t0                            # the host is done; the GPU still has frames queued
        torch_gpu.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames
        final = (nrc.GetLoss(wait=False), nrc.GetLoss(wait=True))
        ren.Destroy()
        nrc.Destroy()
        return dt, seen, final, t_enqueued

    # (the two loops alternate and each keeps its better time: a 60 ms loop right after an idle millisecond runs at the GPU's ramping
    # clock -- the comparison is about the poll, not about which loop ran second)
    t_none, _, _, _ = loop("none")
    t_poll, polled, final, t_enq = loop("poll")
    t_none = min(t_none, loop("none")[0])
    t_poll = min(t_poll, loop("poll")[0])
    t_block, blocked, _, _ = loop("block")
    assert final[0] == final[1]                                      # drained: both polls agree
    # the poll keeps up with training: while the host was enqueueing, the GPU retired about t_enq / t_poll frames, each with a new
    # loss; the poll must have seen at least half of them (it lags by the frames in flight, it does not freeze)
    retired = min(frames, int(t_enq / t_poll))
    assert np.isfinite(polled).all() and len(set(polled)) >= max(8, retired // 2), (len(set(polled)), retired, t_enq, t_poll)
    assert set(polled) <= set(blocked)                               # the same deterministic loss sequence, only delayed (the host
    #                                                                  enqueues frames faster than the GPU retires them)
    assert t_poll <= 1.10 * t_none, (t_poll, t_none, t_block)
    assert t_none < 0.6e-3                                           # the loop itself runs at the bench frame rate


def test_init_with_start_and_finished_events(api, orc, torch_gpu):
    """NeuralRadianceCache::Init(..., cudaStartSemaphore, cudaFinishedSemaphore) with HIP events: InferAndTrain waits for the
    producer's event and records its own; a consumer stream that waits for that one reads finished outputs"""
    n = 4096
    cfg = api.AppConfig(log2_infer_batch_size=12, log2_train_batch_size=10, train_batch_count=1)
    c = api.NeuralRadianceCache(cfg)
    rng = np.random.default_rng(3)
    x = rng.random((n, 5), dtype=np.float32)
    x[:, :3] += 31.0
    producer, consumer, work = torch_gpu.cuda.Stream(), torch_gpu.cuda.Stream(), torch_gpu.cuda.Stream()
    d_in = torch_gpu.zeros((n, 5), device="cuda")
    d_out = torch_gpu.zeros((n, 3), device="cuda")
    d_tin = torch_gpu.from_numpy(x[:1024].copy()).cuda()
    d_tt = torch_gpu.rand((1024, 3), device="cuda")
    start, finished = torch_gpu.cuda.Event(), torch_gpu.cuda.Event()
    src = torch_gpu.from_numpy(x).cuda()
    torch_gpu.cuda.synchronize()
    c.Init(n, d_in, d_out, d_tin, d_tt, stream=work, cudaStartEvent=start, cudaFinishedEvent=finished)
    with torch_gpu.cuda.stream(producer):
        for _ in range(8):                       # the producer is slow; the inputs are complete only at `start`
            d_in.copy_(src * 1.0)
        start.record(producer)
    c.InferAndTrain(None, False)
    with torch_gpu.cuda.stream(consumer):
        consumer.wait_event(finished)
        got = d_out.clone()
    torch_gpu.cuda.synchronize()
    ref = orc.nn_create().forward(x, True, 1)
    assert np.linalg.norm(got.cpu().numpy() - ref) / np.linalg.norm(ref) < 2e-3
    c.Destroy()


def test_two_renderers_share_one_cache_without_host_sync(api, sc, cloud16, torch_gpu):
    """Reference::CompareNrc renders the evaluation view from the cache the training renderer updates (src/Reference.cpp:71-107).
    Two renderers with streams of their own drive one cache alternately; nothing is synchronised on the host between them.  The
    result equals the same sequence with a device-wide synchronisation after every Render, bit for bit."""
    W, H, frames = 256, 160, 6
    scene = sc.make_scene(cloud16, scene_id=4)
    frs = sc.frame_randoms(2 * frames, seed=71)
    cam_eval = sc.make_camera(pos=(0.0, 10.0, 64.0), view_dir=(0.0, -0.1, -1.0), aspect=W / H)

    def run(sync):
        cfg, nrc, cam, train_ren = _nrc_setup(api, sc, scene, W, H, train_batch_count=2, log2_train_batch_size=9)
        eval_ren = api.NrcHpmRenderer(W, H, False, cam_eval, cfg, scene, nrc)
        for f in range(frames):
            train_ren.SetFrameRandom(frs[2 * f])
            train_ren.Render(None, True)
            if sync:
                torch_gpu.cuda.synchronize()
            eval_ren.SetFrameRandom(frs[2 * f + 1])
            eval_ren.Render(None, False)
            if sync:
                torch_gpu.cuda.synchronize()
        out = (train_ren.GetImage().cpu().numpy().copy(), eval_ren.GetImage().cpu().numpy().copy(), nrc.GetLoss(), nrc.GetParams(1).copy())
        eval_ren.Destroy()
        train_ren.Destroy()
        nrc.Destroy()
        return out

    a, b = run(True), run(False)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert a[2] == b[2] and np.array_equal(a[3], b[3])
    assert np.isfinite(a[1]).all() and a[1][..., :3].max() > 0


def test_training_features_encoded_ahead_survive_changes_of_the_cache_owner(api, sc, cloud16, torch_gpu):
    """A generic model whose encoding has no trainable state gets its training batch encoded on the train-ray stream, ahead of the
    backward pass (Mlp::pre_encode) -- only for a renderer that has held the cache for its last two frames, because a change of owner
    orders the new owner's inference and training streams behind the old one's work, not the stream the features are written on.  Two
    renderers that both train one 4x32 cache, in runs of several frames and in alternation, nothing synchronised on the host: the same
    frames, loss and weights as with a device-wide synchronisation after every Render, bit for bit."""
    W, H = 256, 160
    scene = sc.make_scene(cloud16, scene_id=4)
    order = [0] * 4 + [1] + [0] * 4 + [1] * 4 + [0, 1] * 3 + [0] * 3      # which renderer renders (and trains on) frame f
    frs = sc.frame_randoms(len(order), seed=73)
    cam_b = sc.make_camera(pos=(0.0, 10.0, 64.0), view_dir=(0.0, -0.1, -1.0), aspect=W / H)

    def run(sync):
        cfg, nrc, cam, ren_a = _nrc_setup(api, sc, scene, W, H, pos_id=2, dir_id=2, nn_width=32, nn_depth=4, train_batch_count=2,
                                          log2_train_batch_size=9)
        ren_b = api.NrcHpmRenderer(W, H, False, cam_b, cfg, scene, nrc)
        rens = (ren_a, ren_b)
        for f, who in enumerate(order):
            rens[who].SetFrameRandom(frs[f])
            rens[who].Render(None, True)
            if sync:
                torch_gpu.cuda.synchronize()
        out = (ren_a.GetImage().cpu().numpy().copy(), ren_b.GetImage().cpu().numpy().copy(), nrc.GetLoss(), nrc.GetParams(0).copy(),
               nrc.GetParams(1).copy())
        ren_b.Destroy()
        ren_a.Destroy()
        nrc.Destroy()
        return out

    a = run(True)
    for _ in range(3):      # (races are rare: the free-running sequence several times)
        b = run(False)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        assert a[2] == b[2] and np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)) and np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32))
    assert np.isfinite(a[0]).all() and a[0][..., :3].max() > 0
