"""A/B of nrc_schedule.composite_defer (a frame's compositing on the train-ray stream, behind the next frame's train rays, instead of the
inference stream) at the bench shape, the other knobs pinned:   python3 tools/defer_ab.py [--config c2|c5|hash] [--frames F]"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nrc_hpm_renderer_amd import api, scene as sc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--frames", type=int, default=1000)
a = ap.parse_args()
torch.cuda.set_device(0)
W, H = 1920, 1080
if a.config == "c5":
    vol, kw, pin = sc.cached_volume("smoke", 512, seed=1337), dict(pos_id=3, nn_width=128, nn_depth=8), (1, 3, 16)
elif a.config == "hash":
    vol, kw, pin = sc.cached_volume("cloud", 256, seed=1337), dict(pos_id=0, nn_width=64, nn_depth=6), (0, 3, 16)
else:
    vol, kw, pin = sc.cached_volume("cloud", 256, seed=1337), dict(pos_id=3, nn_width=64, nn_depth=6), (0, 2, 2)
scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
cam = sc.make_camera(aspect=W / H)
frs = sc.frame_randoms(a.frames, seed=3)


def run(defer):
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=14, log2_infer_batch_size=21, scene_id=4, **kw)
    nrc = api.NeuralRadianceCache(cfg)
    ren = api.NrcHpmRenderer(W, H, True, cam, cfg, scene, nrc)
    ren.SetSchedule(pin[0], pin[1], pin[2], defer)
    ren.SetBlend(True)
    for k in range(0, 200, 40):
        ren.RenderFrames(frs[k:k + 40], True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200, a.frames, 40):
        ren.RenderFrames(frs[k:k + 40], True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (a.frames - 200)
    st = ren.StageMs() if hasattr(ren, "StageMs") else None
    ren.Destroy()
    nrc.Destroy()
    return dt * 1e3


for rep in range(3):
    for d in (0, 1):
        print("%s composite_defer %d: %.4f ms/frame" % (a.config, d, run(d)), flush=True)
