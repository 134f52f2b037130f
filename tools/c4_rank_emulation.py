import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from nrc_hpm_renderer_amd import api, scene as sc, parallel
GW, GH, WORLD = 3840, 2160, 8
torch.cuda.set_device(0)
vol = sc.cached_volume("cloud", 256, seed=1337)
scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(64, 32))
cam = sc.make_camera(aspect=GW / GH)
for rank in (0, 3):
    lw = parallel.local_width(rank, WORLD, GW)
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=11, log2_infer_batch_size=21)
    nrc = api.NeuralRadianceCache(cfg)
    nrc.SetLossNormFactor(WORLD)
    ren = api.NrcHpmRenderer(lw, GH, True, cam, cfg, scene, nrc, tile=parallel.column_tile(rank, WORLD, GW, GH))
    frs = sc.frame_randoms(64, seed=3)
    ren.SetBlend(True)
    for f in range(24):
        ren.SetFrameRandom(frs[f]); ren.Render(None, True)
    torch.cuda.synchronize()
    ren.StageStats(reset=True)
    n = 400
    t0 = time.perf_counter()
    for f in range(n):
        ren.SetFrameRandom(frs[f % 64]); ren.Render(None, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = ren.StageStats(reset=True)
    print("rank %d tile %dx%d: %.4f ms per sub-frame (%.1f Msamples/s per rank; x8 = %.1f), stages %s" % (rank, lw, GH, dt * 1e3, lw * GH / dt / 1e6, 8 * lw * GH / dt / 1e6, {k: round(v, 3) for k, v in st.items()}))
    ren.Destroy(); nrc.Destroy()
