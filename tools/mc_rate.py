import sys, time
sys.path.insert(0, "/root/repo")
import torch
from nrc_hpm_renderer_amd import api, scene as sc
torch.cuda.set_device(0)
vol = sc.cached_volume("cloud", 256, seed=1337)
scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
cam = sc.make_camera(aspect=1920 / 1080)
mc = api.McHpmRenderer(1920, 1080, 32, True, cam, scene)
frs = sc.frame_randoms(32, seed=1)
for f in range(20):
    mc.SetFrameRandom(frs[f]); mc.Render()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for f in range(n):
    mc.SetFrameRandom(frs[f % 32]); mc.Render()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("MC PATH_LENGTH 32: %.3f ms/frame  %.1f Msamples/s" % (dt * 1e3, 1920 * 1080 / dt / 1e6))
