"""Cases tests/test_oracle_asan.py runs against the sanitizer build of the oracle (collected only when named: the file name does not
match pytest's test_*.py pattern).  Small versions of the golden / parity cases: every entry point of oracle/nrc_oracle.h, on the
reference's cloud fixture where an image is rendered."""
import numpy as np

from conftest import FRAME_RANDOM


def test_sanitizer_build_is_loaded(orc):
    print("oracle library:", orc.lib._name)
    assert "asan" in orc.lib._name


def test_integrator_entry_points(orc, sc, cloud16):
    W, H = 48, 28
    scene = sc.make_scene(cloud16, scene_id=4, env=sc.procedural_sky(16, 8))
    cam = sc.make_camera(aspect=W / H)
    img, info, n_fetch = orc.mc_render(scene, cam, W, H, 8, FRAME_RANDOM, threads=4)
    assert np.isfinite(img).all() and n_fetch > 0
    o = orc.nrc_gen_rays(scene, cam, W, H, 1, 0.0, FRAME_RANDOM, threads=4)
    assert o["info"].sum() > 10 and np.isfinite(o["primary"]).all()
    # scene 1 (point light) and an empty / a solid volume walk the other branches (cap of 128 collisions, immediate hits)
    for vol, sid in ((np.zeros((6, 5, 7), np.uint8), 5), (np.full((6, 5, 7), 255, np.uint8), 1)):
        s2 = sc.make_scene(vol, scene_id=sid)
        img2, _, _ = orc.mc_render(s2, cam, 16, 12, 4, FRAME_RANDOM, threads=2)
        assert np.isfinite(img2).all()
    tw, th = 8, 4
    ring_size = tw * th
    head_tail = np.zeros(2, np.uint32)
    ring = np.zeros((ring_size, 6), np.float32)
    ring[:, 5] = 1.0
    for _ in range(2):                                     # the second frame pops what the first pushed
        tin, tgt = orc.nrc_prep_train(scene, W, H, tw, th, W // tw, W // tw, 2, 2, ring_size, FRAME_RANDOM, o["info"], o["origin"], o["dir"],
                                      head_tail, ring, threads=2)
        assert np.isfinite(tgt).all() and tin.shape == (tw * th, 5)
    out = np.zeros((H, W, 4), np.float32)
    orc.nrc_composite(W, H, 1, 1.0, o["primary"], o["info"], np.zeros((W * H, 3), np.float32), out)
    assert np.isfinite(out).all()
    res = orc.compare(img, out)
    assert res["valid"] > 0
    a = np.linspace(-3.0, 3.0, 1001, dtype=np.float32)
    for fn in range(8):
        orc.math_eval(fn, a, a[::-1].copy())


def test_nn_entry_points(orc):
    rng = np.random.default_rng(1)
    x = rng.random((256, 5), dtype=np.float32)
    x[:, :3] += 31.0
    x[::7, 4] = np.nan                                   # quirk Q5
    t = rng.random((256, 3), dtype=np.float32)
    for kw in (dict(), dict(pos_id=0, hashgrid_log2_size=10, depth=2), dict(pos_id=1, dir_id=2, width=32, depth=2), dict(pos_id=2, dir_id=0, width=16, depth=1, optimizer="SGD")):
        nn = orc.nn_create(**kw)
        for mode in (0, 1):
            y = nn.forward(x, use_ema=True, mode=mode)
            assert y.shape == (256, 3)
        loss = nn.backward(x, t)
        assert np.isfinite(loss) or kw.get("dir_id", 0) != 0
        nn.optimizer_step()
        assert np.isfinite(np.array(nn.buffer(0))).all() or kw.get("dir_id", 0) != 0
