import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrc_hpm_renderer_amd import api, scene as sc
W, H = 1920, 1080
vol = sc.cached_volume("cloud", 256, seed=1337)
scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky(64, 32))
cam = sc.make_camera(aspect=W / H)
cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=14, log2_infer_batch_size=21)
nrc = api.NeuralRadianceCache(cfg)
ren = api.NrcHpmRenderer(W, H, True, cam, cfg, scene, nrc)
ren.Render(None, True)
info = ren.Buffer("info").cpu().numpy().reshape(H, W)      # [y][x]
live = info > 0
print("scattered pixels %d (%.1f %%)" % (live.sum(), 100 * live.mean()))
col = live.T.reshape(-1)          # x*H + y order
n = col.size // 32
t = col[:n * 32].reshape(n, 32)
lt = t.any(axis=1)
print("32-query tiles: %d, live %d (%.1f %%), ideal %d, live/ideal %.2f, mean fill of live tiles %.1f %%" % (n, lt.sum(), 100 * lt.mean(), int(np.ceil(live.sum() / 32)), lt.sum() / np.ceil(live.sum() / 32), 100 * t[lt].mean()))
