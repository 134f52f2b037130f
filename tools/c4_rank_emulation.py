"""One rank's share of configs[3] (3840x2160 at 8 spp over 8 ranks, 2 048 of the 16 384 train rays) rendered alone on one GPU:
what a rank's sub-frame costs before any communication.  Strip width as arguments (default: 1 = single interleaved columns, and
parallel.DEFAULT_BLOCK):   python tools/c4_rank_emulation.py [block ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nrc_hpm_renderer_amd import api, scene as sc, parallel
GW, GH, WORLD = 3840, 2160, 8
torch.cuda.set_device(0)
vol = sc.cached_volume("cloud", 256, seed=1337)
scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(64, 32))
cam = sc.make_camera(aspect=GW / GH)
blocks = [int(a) for a in sys.argv[1:]] or [1, parallel.DEFAULT_BLOCK]
for block, rank in [(b, r) for b in blocks for r in (0, 3)]:
    lw = parallel.local_width(rank, WORLD, GW, block)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=11, log2_infer_batch_size=21)
    nrc = api.NeuralRadianceCache(cfg)
    nrc.SetLossNormFactor(WORLD)
    ren = api.NrcHpmRenderer(lw, GH, True, cam, cfg, scene, nrc, tile=parallel.column_tile(rank, WORLD, GW, GH, block))
    frs = sc.frame_randoms(64, seed=3)
    ren.SetBlend(True)
    frs = np.asarray(frs, np.float32)
    ren.RenderFrames(frs[:24], True)
    torch.cuda.synchronize()
    ren.StageStats(reset=True)
    n = 384
    t0 = time.perf_counter()
    for _ in range(n // 64):
        ren.RenderFrames(frs, True)      # one call per 64 frames; every frame's successor is known (hot-tile list one frame ahead)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = ren.StageStats(reset=True)
    print("strips of %d columns, rank %d tile %dx%d: %.4f ms per sub-frame (%.1f Msamples/s per rank; x8 = %.1f), stages %s" % (block, rank, lw, GH, dt * 1e3, lw * GH / dt / 1e6, 8 * lw * GH / dt / 1e6, {k: round(v, 3) for k, v in st.items()}))
    ren.Destroy(); nrc.Destroy()
