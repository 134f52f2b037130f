#!/usr/bin/env python3
"""The reference's own self-check applied to the hot path (VERDICT r05 "next" 1): what `Benchmark()` logs every frame of its main loop
(src/main.cu:140-150) -- `frame mse relBias CV` of the NRC image, rendered WITHOUT training from the reference camera
(Reference::CompareNrc, src/Reference.cpp:72-107; a single un-blended frame, quirk Q15), against reference/<scene>/0.exr through the
metrics of data/shader/ref/cmp1.comp:23-41 / cmp2.comp (here nrc_compare_images) -- for the scenes whose EXR the checked-in estimator
produced (0 and 4), on the reference's cloud (data/volume/wdas_cloud_sixteenth.vdb as tests/golden/cloud_sixteenth_u8.npz;
the EXRs were rendered with the absent quarter-resolution file, the one systematic: DESIGN.md section 2).

Per scene, three renderers against the same two references (the EXR; this build's own MC ground truth on the SAME cloud, generated like
Reference::GenRefImages, src/Reference.cpp:566-606: PATH_LENGTH 64, blended -- the comparison free of the cloud-resolution systematic):
  nrc     NrcHpmRenderer, faithful quirks (Q2: single-vertex training targets), the north-star model (Frequency(12) + OneBlob(4), 6 x 64,
          16 384 train rays + one Adam step per frame); `--hash` adds the reference's default model (HashGrid, 4 x 2^14 train rays)
  nrc_q2  the same with compat_fix = Q2 (train ray length 32: what the CLI asks for, src/NrcHpmRenderer.cu:991-994 vs :1036-1055)
  mc      McHpmRenderer, PATH_LENGTH 32 (src/main.cu:213), progressive blend

Columns of the per-frame table: frames trained so far, cumulative time of the training frames (ms, host clock around drained blocks --
the evaluation frames are not in it, as the reference's frame time excludes them), loss, then {mse relBias CV} of the evaluation frame
against the EXR and {mse relBias} against the own ground truth.  The MC table is the blended image after k frames.
The equal-time table: the image each renderer shows after a wall-clock budget with blending on (NRC: every training frame blended, as
`SetBlend(true)` does while training; MC: every frame blended), frames counted at the renderer's own pipelined rate, and the NRC's
single evaluation frame at that moment.

  python tools/convergence.py --scenes 0,4 --frames 2048 --out profiles/r06_convergence
"""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
GOLDEN = os.path.join(ROOT, "tests", "golden")
W, H = 1920, 1080


def log_frames(n):
    """frames after which the evaluation frame is rendered: every one up to 64, every 8th up to 512, every 32nd beyond"""
    out = [f for f in range(1, n + 1) if f <= 64 or (f <= 512 and f % 8 == 0) or f % 32 == 0]
    if out[-1] != n:
        out.append(n)
    return out


def load_exr_reference(torch, sid):
    z = np.load(os.path.join(GOLDEN, "exr_%d_1920x1080.npz" % sid))
    img = np.stack([z["L"], z["L"], z["L"], z["A"]], axis=-1).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(img)).cuda()


def metrics(api, ref, img):
    r = api.CompareImages(ref, img)
    rel_bias = (r["own_mean"] - r["ref_mean"]) / r["ref_mean"] if r["ref_mean"] else 0.0
    cv = math.sqrt(max(r["own_var"], 0.0)) / r["own_mean"] if r["own_mean"] else 0.0
    return r["mse"], rel_bias, cv


def make_cfg(api, sid, mode):
    kw = dict(train_batch_count=1, log2_train_batch_size=14, log2_infer_batch_size=21, scene_id=sid, primary_ray_length=1,
              primary_ray_prob=0.0, train_spp=1, train_ring_buf_size=1.0, seed=1337, train_ray_length=32, pos_id=3, dir_id=0,
              nn_width=64, nn_depth=6)
    if mode.endswith("_q2"):
        kw["compat_fix"] = api.NRC_FIX_Q2_TRAIN_RAY_LEN
    if mode.startswith("hash"):
        kw.update(pos_id=0, train_batch_count=4)          # src/main.cu:434-438
    return api.AppConfig(**kw)


def own_ground_truth(torch, api, sc, scene, cam, frames, seed, path_length=64):
    """Reference::GenRefImages: PATH_LENGTH 64, `frames` blended frames (8192 in the reference)"""
    mc = api.McHpmRenderer(W, H, path_length, True, cam, scene)
    frs = sc.frame_randoms(frames, seed=seed)
    for f in range(frames):
        mc.SetFrameRandom(frs[f])
        mc.Render()
    ref = mc.GetImage().clone()
    torch.cuda.synchronize()
    mc.Destroy()
    return ref


def nrc_curve(torch, api, sc, scene, cam, sid, mode, frames, refs, out):
    cfg = make_cfg(api, sid, mode)
    nrc = api.NeuralRadianceCache(cfg)
    ren = api.NrcHpmRenderer(W, H, False, cam, cfg, scene, nrc)                 # src/main.cu:203-210: blend off
    ev = api.NrcHpmRenderer(W, H, False, cam, cfg, scene, nrc)                  # CompareNrc's frame: train = false
    frs = sc.frame_randoms(frames, seed=1337)
    efr = sc.frame_randoms(frames, seed=4242)
    marks = log_frames(frames)
    out.write("# %s scene %d: %s\n" % (mode, sid, cfg.GetName()))
    out.write("# frame train_ms loss | vs EXR: mse relBias CV | vs own MC-64 ground truth: mse relBias\n")
    done, t_ms, rows = 0, 0.0, []
    for m in marks:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ren.RenderFrames(frs[done:m], True)
        torch.cuda.synchronize()
        t_ms += (time.perf_counter() - t0) * 1e3
        done = m
        loss = nrc.GetLoss()
        ev.SetFrameRandom(efr[m - 1])
        ev.Render(None, False)
        img = ev.GetImage()
        a = metrics(api, refs[0], img)
        b = metrics(api, refs[1], img)
        rows.append((m, t_ms, loss) + a + b[:2])
        out.write("%d %.3f %.5f  %.6g %.5f %.4f  %.6g %.5f\n" % rows[-1])
    out.write("\n")
    out.flush()
    # the pipelined rate of the training frames (no drain between frames), for the equal-time table
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ren.RenderFrames(sc.frame_randoms(256, seed=99), True)
    torch.cuda.synchronize()
    rate_ms = (time.perf_counter() - t0) * 1e3 / 256
    ren.Destroy()
    ev.Destroy()
    nrc.Destroy()
    return rows, rate_ms


def mc_curve(torch, api, sc, scene, cam, sid, frames, refs, out):
    mc = api.McHpmRenderer(W, H, 32, True, cam, scene)                          # src/main.cu:213 (blended here: the curve is its convergence)
    frs = sc.frame_randoms(frames, seed=1337)
    marks = log_frames(frames)
    out.write("# mc scene %d: McHpmRenderer PATH_LENGTH 32, progressive blend\n" % sid)
    out.write("# frame render_ms - | vs EXR: mse relBias CV | vs own MC-64 ground truth: mse relBias\n")
    done, t_ms, rows = 0, 0.0, []
    for m in marks:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(done, m):
            mc.SetFrameRandom(frs[f])
            mc.Render()
        torch.cuda.synchronize()
        t_ms += (time.perf_counter() - t0) * 1e3
        done = m
        img = mc.GetImage()
        a = metrics(api, refs[0], img)
        b = metrics(api, refs[1], img)
        rows.append((m, t_ms, 0.0) + a + b[:2])
        out.write("%d %.3f -  %.6g %.5f %.4f  %.6g %.5f\n" % ((m, t_ms) + a + b[:2]))
    out.write("\n")
    out.flush()
    rate_ms = t_ms / frames
    mc.Destroy()
    return rows, rate_ms


def equal_time(torch, api, sc, scene, cam, sid, modes, rates, budgets, refs, out):
    out.write("# equal time, scene %d: the image after a wall-clock budget (frames = budget / the renderer's pipelined ms per frame, measured time in the row)\n" % sid)
    out.write("# renderer budget_ms frames measured_ms | blended image vs EXR: mse relBias | vs own: mse relBias | NRC single evaluation frame vs EXR: mse relBias | vs own: mse relBias\n")
    res = {}
    for budget in budgets:
        for mode in modes:
            n = max(1, int(budget / rates[mode]))
            frs = sc.frame_randoms(n, seed=555)
            if mode == "mc":
                r = api.McHpmRenderer(W, H, 32, True, cam, scene)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for f in range(n):
                    r.SetFrameRandom(frs[f])
                    r.Render()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                img = r.GetImage()
                a, b = metrics(api, refs[0], img), metrics(api, refs[1], img)
                e = None
                r.Destroy()
            else:
                cfg = make_cfg(api, sid, mode)
                nrc = api.NeuralRadianceCache(cfg)
                r = api.NrcHpmRenderer(W, H, True, cam, cfg, scene, nrc)
                ev = api.NrcHpmRenderer(W, H, False, cam, cfg, scene, nrc)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r.RenderFrames(frs, True)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                img = r.GetImage()
                a, b = metrics(api, refs[0], img), metrics(api, refs[1], img)
                ev.SetFrameRandom(sc.frame_randoms(1, seed=777)[0])
                ev.Render(None, False)
                eimg = ev.GetImage()
                e = metrics(api, refs[0], eimg)[:2] + metrics(api, refs[1], eimg)[:2]
                r.Destroy()
                ev.Destroy()
                nrc.Destroy()
            res[(mode, budget)] = dict(frames=n, ms=ms, blend_exr=a, blend_own=b, eval=e)
            line = "%s %g %d %.2f  %.6g %.5f  %.6g %.5f" % (mode, budget, n, ms, a[0], a[1], b[0], b[1])
            if e is not None:
                line += "  %.6g %.5f  %.6g %.5f" % e
            out.write(line + "\n")
    out.write("\n")
    out.flush()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="0,4")
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--gt-frames", type=int, default=4096, help="blended PATH_LENGTH-64 frames of the own ground truth (reference: 8192)")
    ap.add_argument("--budgets", default="10,50,200")
    ap.add_argument("--hash", action="store_true", help="also the reference's default model (HashGrid)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "convergence"))
    args = ap.parse_args()
    import torch
    from nrc_hpm_renderer_amd import api, scene as sc
    cloud = np.load(os.path.join(GOLDEN, "cloud_sixteenth_u8.npz"))["density"]
    cam = sc.make_camera(aspect=W / H)                 # = Reference::CreateRefCameras, src/Reference.cpp:443-455
    modes = ["nrc", "nrc_q2"] + (["hash", "hash_q2"] if args.hash else [])
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    for sid in [int(s) for s in args.scenes.split(",")]:
        scene = sc.make_scene(cloud, scene_id=sid)      # env = None: the constant white map the reference ends up with (quirk Q9)
        with open("%s_%d.txt" % (args.out, sid), "w") as out:
            out.write("# tools/convergence.py: scene %d, %dx%d, cloud %s, build %s, %s\n" % (sid, W, H, "x".join(map(str, cloud.shape[::-1])),
                                                                                           api.build_id(), torch.cuda.get_device_name(0)))
            t0 = time.time()
            refs = (load_exr_reference(torch, sid), own_ground_truth(torch, api, sc, scene, cam, args.gt_frames, seed=31337))
            g = metrics(api, refs[0], refs[1])
            out.write("# own MC-64 ground truth (%d frames, %.1f s) vs EXR: mse %.6g relBias %.5f CV %.4f (EXR vs itself: CV %.4f)\n"
                      % (args.gt_frames, time.time() - t0, g[0], g[1], g[2], metrics(api, refs[0], refs[0])[2]))
            mc3 = own_ground_truth(torch, api, sc, scene, cam, 1024, seed=999, path_length=3)
            l3, l3o = metrics(api, refs[0], mc3), metrics(api, refs[1], mc3)
            out.write("# the faithful mode's limit -- quirk Q2: single-vertex training targets, the frame can converge to McHpmRenderer PATH_LENGTH 3 at best "
                      "(1024 frames) -- vs EXR: relBias %.5f; vs own MC-64: relBias %.5f\n\n" % (l3[1], l3o[1]))
            del mc3
            rates = {}
            for mode in modes:
                _, rates[mode] = nrc_curve(torch, api, sc, scene, cam, sid, mode, args.frames, refs, out)
                print("scene %d %s: %.4f ms per pipelined training frame" % (sid, mode, rates[mode]), flush=True)
            _, rates["mc"] = mc_curve(torch, api, sc, scene, cam, sid, min(args.frames, 512), refs, out)
            print("scene %d mc: %.4f ms per frame" % (sid, rates["mc"]), flush=True)
            out.write("# pipelined ms per frame: %s\n\n" % " ".join("%s %.4f" % kv for kv in rates.items()))
            equal_time(torch, api, sc, scene, cam, sid, modes + ["mc"], rates, [float(b) for b in args.budgets.split(",")], refs, out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
