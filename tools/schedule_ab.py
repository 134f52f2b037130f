"""Does the renderer's tuner find the schedule an exhaustive A/B finds?   python3 tools/schedule_ab.py [--width W --height H --volume N --nn-width K --nn-depth D --frames F]
Every combination of the three knobs pinned (nrc_renderer_set_schedule) and timed over F frames, then a renderer left to itself."""
import argparse, itertools, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nrc_hpm_renderer_amd import api, scene as sc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1600)
ap.add_argument("--height", type=int, default=900)
ap.add_argument("--volume", type=int, default=384)
ap.add_argument("--nn-width", type=int, default=128)
ap.add_argument("--nn-depth", type=int, default=4)
ap.add_argument("--pos-id", type=int, default=3)
ap.add_argument("--frames", type=int, default=1200)
ap.add_argument("--smoke", action="store_true")
a = ap.parse_args()
torch.cuda.set_device(0)
vol = sc.cached_volume("smoke" if a.smoke else "cloud", a.volume, seed=1337)
scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
cam = sc.make_camera(aspect=a.width / a.height)
frs = sc.frame_randoms(a.frames, seed=3)


def run(pin):
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=14, log2_infer_batch_size=21, scene_id=4, pos_id=a.pos_id, nn_width=a.nn_width, nn_depth=a.nn_depth)
    nrc = api.NeuralRadianceCache(cfg)
    ren = api.NrcHpmRenderer(a.width, a.height, True, cam, cfg, scene, nrc)
    if pin is not None:
        ren.SetSchedule(*pin)
    ren.SetBlend(True)
    n_warm = 400 if pin is None else 120      # (the free renderer finishes tuning inside the warm-up)
    for k in range(0, n_warm, 40):
        ren.RenderFrames(frs[k:k + 40], True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(0, a.frames - 400, 40):
        ren.RenderFrames(frs[400 + k:440 + k], True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (a.frames - 400)
    s = ren.GetSchedule()
    ren.Destroy()
    nrc.Destroy()
    return dt * 1e3, s


res = []
for pri, lag, win in itertools.product((0, 1), (2, 3), (0, 2, 16)):
    ms, _ = run((pri, lag, win))
    res.append((ms, (pri, lag, win)))
    print("pinned priority %d lag %d window %2d: %.4f ms/frame" % (pri, lag, win, ms), flush=True)
best = min(res)
ms, s = run(None)
print("best pinned %s %.4f ms; tuner chose (%d, %d, %d), done %s: %.4f ms/frame = %+.1f %% of the best pinned" %
      (best[1], best[0], s["camera_priority_low"], s["cost_order_lag"], s["xcd_window"], s["tuning_done"], ms, 100.0 * (ms / best[0] - 1.0)))
