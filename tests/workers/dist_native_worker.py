"""Worker of tests/test_gpu_dist_native.py, launched by `python -m torch.distributed.run` (one rank per GPU): drives the
product's own exchange path -- nrc_cache_comm_init -> ncclAllReduce of gradient vector + loss cell on the training stream inside
the renderer's frame graph -- and writes what it saw to a JSON file.

With one rank the all-reduce is the identity, so losses, weights and frames must equal a run without any communicator bit for
bit; with N ranks each rank renders its column tile and the replicas must stay identical."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(api, sc, parallel, scene, cam, W, H, rank, world, frames, exchange, dtype="f32", **model):
    """exchange: None | "native" | "hook"; dtype: what travels (nrc_cache_set_exchange_dtype)"""
    cfg = api.AppConfig(train_batch_count=1, log2_train_batch_size=10, log2_infer_batch_size=14, **model)
    nrc = api.NeuralRadianceCache(cfg)
    if exchange == "native":
        parallel.attach_gradient_allreduce(nrc, world, native=True, dtype=dtype)
    elif exchange == "hook":
        parallel.attach_gradient_allreduce(nrc, world, native=False, dtype=dtype)
    lw = parallel.local_width(rank, world, W)
    ren = api.NrcHpmRenderer(lw, H, True, cam, cfg, scene, nrc, tile=parallel.column_tile(rank, world, W, H))
    frs = sc.frame_randoms(frames, seed=77)
    losses = []
    for f in range(frames):
        ren.SetFrameRandom(frs[f])
        ren.Render(None, True)
        losses.append(nrc.GetLoss())
    img = ren.GetImage().cpu().numpy().copy()
    out = dict(losses=losses, w=nrc.GetParams(0), ema=nrc.GetParams(1), img=img, comm=nrc.CommInfo(), step=nrc.GetStep(),
               sparse=nrc.CommSparse())
    ren.Destroy()
    nrc.Destroy()
    return out


def main():
    out_path = sys.argv[1]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    from nrc_hpm_renderer_amd import api, parallel, scene as sc
    vol = sc.quantize_density(sc.fbm_cloud_volume(48, seed=3))
    scene = sc.make_scene(vol, scene_id=4)
    W, H, frames = 256, 96, 5
    cam = sc.make_camera(aspect=W / H)
    native = run(api, sc, parallel, scene, cam, W, H, rank, world, frames, "native")
    hook = run(api, sc, parallel, scene, cam, W, H, rank, world, frames, "hook")
    res = dict(rank=rank, world=world, comm=list(native["comm"]), comm_hook=list(hook["comm"]), losses=native["losses"],
               step=native["step"], finite=bool(np.isfinite(native["img"]).all() and np.isfinite(native["losses"]).all()),
               native_equals_hook=bool(np.array_equal(native["w"], hook["w"]) and native["losses"] == hook["losses"]
                                       and np.array_equal(native["img"], hook["img"])))
    # the fp16 exchange (BASELINE.json configs[3]): ncclAllReduce(ncclHalf) inside the frame graph against the hook path's statement of it
    # (values rounded to fp16, summed, rounded) -- the same bits for this world size; and it is NOT the fp32 run (the rounding is real)
    native16 = run(api, sc, parallel, scene, cam, W, H, rank, world, frames, "native", dtype="f16")
    hook16 = run(api, sc, parallel, scene, cam, W, H, rank, world, frames, "hook", dtype="f16")
    res["fp16_native_equals_hook"] = bool(np.array_equal(native16["w"], hook16["w"]) and native16["losses"] == hook16["losses"]
                                          and np.array_equal(native16["img"], hook16["img"]))
    res["fp16_differs_from_fp32"] = bool(not np.array_equal(native16["w"], native["w"]))
    res["fp16_weight_rel_diff"] = float(np.linalg.norm(native16["w"] - native["w"]) / np.linalg.norm(native["w"]))
    res["fp16_finite"] = bool(np.isfinite(native16["losses"]).all() and np.isfinite(native16["w"]).all())
    res["fp16_losses"] = native16["losses"]
    # HashGrid model (the reference's default encoding): the table gradient travels as all-gathered (entry, value) lists; its
    # packed-fp16 atomics sum in a different order every run, so the dense exchange is the reference only up to that noise
    hg = dict(pos_id=0, hashgrid_log2_size=14)
    sparse = run(api, sc, parallel, scene, cam, W, H, rank, world, frames, "native", **hg)
    os.environ["NRC_DEBUG"] = "dense_grid_exchange"
    dense = run(api, sc, parallel, scene, cam, W, H, rank, world, frames, "native", **hg)
    del os.environ["NRC_DEBUG"]
    sparse16 = run(api, sc, parallel, scene, cam, W, H, rank, world, frames, "native", dtype="f16", **hg)
    res["hashgrid_fp16_finite"] = bool(np.isfinite(sparse16["losses"]).all() and np.isfinite(sparse16["w"]).all())
    res["hashgrid_fp16_weight_rel_diff"] = float(np.linalg.norm(sparse16["w"] - sparse["w"]) / np.linalg.norm(sparse["w"]))
    res["hashgrid_sparse_flags"] = [bool(sparse["sparse"]), bool(dense["sparse"]), bool(native["sparse"])]
    res["hashgrid_losses_sparse"], res["hashgrid_losses_dense"] = sparse["losses"], dense["losses"]
    res["hashgrid_finite"] = bool(np.isfinite(sparse["img"]).all() and np.isfinite(sparse["losses"]).all())
    res["hashgrid_weight_rel_diff"] = float(np.linalg.norm(sparse["w"] - dense["w"]) / np.linalg.norm(dense["w"]))
    if world == 1:
        plain = run(api, sc, parallel, scene, cam, W, H, 0, 1, frames, None)
        res["native_equals_no_communicator"] = bool(np.array_equal(native["w"], plain["w"]) and np.array_equal(native["ema"], plain["ema"])
                                                    and native["losses"] == plain["losses"] and np.array_equal(native["img"], plain["img"]))
    else:       # replicas identical on every rank
        t = torch.from_numpy(native["w"]).cuda()
        ref = t.clone()
        dist.broadcast(ref, 0)
        res["replicas_identical"] = bool(torch.equal(t, ref))
    with open(out_path + ".%d" % rank, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
