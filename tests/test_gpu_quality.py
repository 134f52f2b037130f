"""The reference's own self-check on the TRAINED path (src/main.cu:140-150 -> Reference::CompareNrc, src/Reference.cpp:72-107 -> cmp1/cmp2.comp):
the NRC image, rendered without training on the reference's cloud from the reference camera, against reference/{0,4}/0.exr -- the only
reference-held data that reaches the encoding / MLP / loss / optimizer half of the path, whose arithmetic is otherwise parity-unpinned
(tiny-cuda-nn is an empty submodule).  Statistics, bounds and their calibration: tests/quality.py; the curves over 2 048 frames:
profiles/r06_convergence_{0,4}.txt (tools/convergence.py)."""
import numpy as np
import pytest

import quality

pytestmark = pytest.mark.gpu

TRAIN_FRAMES, EVAL_FRAMES = 512, 32


@pytest.fixture(scope="module")
def setup(api, sc, cloud16, torch_gpu):
    cam = sc.make_camera(aspect=quality.W / quality.H)
    refs = {}
    for sid in (0, 4):
        scene = sc.make_scene(cloud16, scene_id=sid)
        refs[sid] = dict(exr=quality.load_exr(torch_gpu, sid), mc3=quality.mc_image(torch_gpu, api, sc, scene, cam, 3, 384))
    return cam, refs


@pytest.mark.parametrize("sid", [0, 4])
def test_trained_nrc_frame_against_the_reference_exr(api, sc, cloud16, torch_gpu, setup, sid):
    """512 frames of Render(queue, true), then Reference::CompareNrc's frame (train = false; 32 of them blended: one frame's MSE is the
    primary path's noise).  With quirk Q2 fixed the frame estimates the EXR's 64-vertex series: relBias within the window around the EXR;
    as shipped (Q2) it converges to the three-vertex series -- the image McHpmRenderer renders with PATH_LENGTH 3 -- and must do THAT."""
    cam, refs = setup
    scene = sc.make_scene(cloud16, scene_id=sid)
    b = quality.bounds(sid)
    q2 = quality.train_and_evaluate(torch_gpu, api, sc, scene, cam, quality.nrc_config(api, sid, True), TRAIN_FRAMES, EVAL_FRAMES, refs[sid])
    assert np.isfinite(q2["loss"])
    assert b["q2_rel_bias"][0] <= q2["exr"]["rel_bias"] <= b["q2_rel_bias"][1], q2["exr"]
    assert q2["exr"]["mse"] <= b["q2_mse32"][1], q2["exr"]
    assert q2["exr"]["valid"] == int((refs[sid]["exr"][..., 3] != 0).sum().item())      # cmp1.comp:34: pixels the reference's alpha marks
    faithful = quality.train_and_evaluate(torch_gpu, api, sc, scene, cam, quality.nrc_config(api, sid, False), TRAIN_FRAMES, EVAL_FRAMES, refs[sid])
    mc3_vs_exr = quality.result(api, refs[sid]["exr"], refs[sid]["mc3"])["rel_bias"]
    d = faithful["exr"]["rel_bias"] - mc3_vs_exr
    assert b["faithful_minus_mc3"][0] <= d <= b["faithful_minus_mc3"][1], (faithful["exr"], mc3_vs_exr)
    # and the truncation itself is what separates the two modes: the shipped behaviour sits 9-16 % below the EXR
    assert faithful["exr"]["rel_bias"] < q2["exr"]["rel_bias"] - 0.06


@pytest.mark.parametrize("fault", [f for f in quality.FAULTS if f not in quality.INVISIBLE])
def test_the_quality_bounds_reject_a_broken_trainer(api, sc, cloud16, torch_gpu, setup, fault):
    """every fault is reached through the public interface (tests/quality.py: no training, the cache's term dropped, training targets
    scaled, another loss, a diverging learning rate) and must leave the window the healthy trainer sits in, on both scenes (the
    closest call: targets x 1.5 on scene 0, +5.1 % against the window's +2.9 %)"""
    cam, refs = setup
    for sid in (0, 4):
        b = quality.bounds(sid)
        r = quality.run_fault(torch_gpu, api, sc, cloud16, cam, sid, fault, TRAIN_FRAMES, EVAL_FRAMES, refs[sid])
        ok = b["q2_rel_bias"][0] <= r["exr"]["rel_bias"] <= b["q2_rel_bias"][1] and r["exr"]["mse"] <= b["q2_mse32"][1]
        assert not ok, (fault, sid, r["exr"])


def test_what_the_image_cannot_see(api, sc, cloud16, torch_gpu, setup):
    """recorded, not hidden: a loss normaliser off by 2 x is divided out by Adam (the step is m / sqrt(v): the frame does not move by more
    than its own noise, while the LOSS VALUE halves -- which is how such a fault shows), and inference without the EMA is unbiased"""
    cam, refs = setup
    sid = 4
    base = quality.run_fault(torch_gpu, api, sc, cloud16, cam, sid, "none", TRAIN_FRAMES, EVAL_FRAMES, refs[sid])
    norm = quality.run_fault(torch_gpu, api, sc, cloud16, cam, sid, "loss_norm_x2", TRAIN_FRAMES, EVAL_FRAMES, refs[sid])
    assert abs(norm["exr"]["rel_bias"] - base["exr"]["rel_bias"]) < 0.01
    assert 0.4 < norm["loss"] / base["loss"] < 0.6
    ema = quality.run_fault(torch_gpu, api, sc, cloud16, cam, sid, "no_ema", TRAIN_FRAMES, EVAL_FRAMES, refs[sid])
    b = quality.bounds(sid)["q2_rel_bias"]
    assert b[0] <= ema["exr"]["rel_bias"] <= b[1]


@pytest.mark.parametrize("sid", [0, 4])
def test_the_references_default_model_against_the_reference_exr(api, sc, cloud16, torch_gpu, setup, sid):
    """the same window for the model the reference starts with (src/main.cu:434-438: HashGrid position encoding, 4 train batches of 2^14
    rays), quirk Q2 fixed: measured -4.3 % / +1.2 % after 512 frames (profiles/r06_convergence_{0,4}.txt) -- the trainable table, its exact
    gradient sums and sparse optimizer reach the same image as the frequency-encoded model"""
    cam, refs = setup
    scene = sc.make_scene(cloud16, scene_id=sid)
    cfg = quality.nrc_config(api, sid, True, pos_id=0, train_batch_count=4)
    r = quality.train_and_evaluate(torch_gpu, api, sc, scene, cam, cfg, TRAIN_FRAMES, EVAL_FRAMES, refs[sid])
    b = quality.bounds(sid)
    assert np.isfinite(r["loss"])
    assert b["q2_rel_bias"][0] <= r["exr"]["rel_bias"] <= b["q2_rel_bias"][1], r["exr"]
    assert r["exr"]["mse"] <= b["q2_mse32"][1] * 1.05, r["exr"]
