"""Parity of the fused encode+MLP HIP kernels (inference, backward, optimizer) against the oracle, through the C ABI.
Tolerances: MLP fp16 path vs oracle fp16-storage mode: rel-L2 <= 2e-3 on outputs; gradients: the bulk of a batch <= 3e-5,
overall <= 2e-3 with at most a handful of rays whose ReLU decision is a tie (gradient_agreement below; profiles/r06_grad_drift.txt);
weights after a step rel-L2 <= 1e-3.  (fp32 MFMA accumulation order differs from the oracle's wide accumulate, and a
1-ulp fp16 rounding flip of an activation propagates.)"""
import numpy as np
import pytest

from conftest import nrc_debug

pytestmark = pytest.mark.gpu


def queries(n, seed=0, nan_frac=0.1):
    rng = np.random.default_rng(seed)
    x = rng.random((n, 5), dtype=np.float32)
    x[:, :3] += 31.0
    x[:, 3] = x[:, 3] * 2.0 - 0.5
    x[rng.random(n) < nan_frac, 4] = np.nan
    return x


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture()
def cache(api, torch_gpu):
    c = api.NeuralRadianceCache(api.AppConfig(train_batch_count=1, log2_train_batch_size=10))
    yield c
    c.Destroy()


def randomize(cache, onn, seed=5, scale=1.5):
    """non-trivial asymmetric weights in both implementations (EMA != master)"""
    rng = np.random.default_rng(seed)
    w = np.array(onn.buffer(0)) * scale + (rng.standard_normal(onn.n_params) * 0.01).astype(np.float32)
    e = w + (rng.standard_normal(onn.n_params) * 0.02).astype(np.float32)
    onn.buffer(0)[:] = w
    onn.buffer(1)[:] = e
    cache.SetParams(0, w)
    cache.SetParams(1, e)


def test_initial_weights_match_oracle(cache, orc):
    onn = orc.nn_create()
    assert cache.ParamCount() == onn.n_params == 25792
    assert np.array_equal(cache.GetParams(0), onn.buffer(0))
    assert np.array_equal(cache.GetParams(1), onn.buffer(1))


@pytest.mark.parametrize("model", [dict(), dict(nn_width=128, nn_depth=8), dict(pos_id=0, hashgrid_log2_size=12, nn_depth=3)],
                         ids=["6x64", "8x128", "hashgrid"])
def test_tcnn_parameter_layout_roundtrip(api, orc, torch_gpu, model):
    """nrc_cache_{get,set}_params_tcnn: the vectors in tiny-cuda-nn's own layout (output matrix 16 x width) -- equal to the oracle's
    statement of tiny-cuda-nn's initialisation bit for bit, 26 624 numbers for the 6 x 64 model; a dump set into a cache of another
    seed gives the same network (same outputs, same dump back, dead rows included)"""
    a = api.NeuralRadianceCache(api.AppConfig(**model))
    width, depth = model.get("nn_width", 64), model.get("nn_depth", 6)
    onn = orc.nn_create(pos_id=model.get("pos_id", 3), width=width, depth=depth, hashgrid_log2_size=model.get("hashgrid_log2_size", 0))
    assert a.ParamCountTcnn() == a.ParamCount() + 13 * width
    if not model:
        assert a.ParamCountTcnn() == 26624
    t0 = a.GetParamsTcnn(0)
    assert np.array_equal(t0, onn.tcnn_params(0, width)) and np.array_equal(a.GetParamsTcnn(1), onn.tcnn_params(1, width))
    assert not a.GetParamsTcnn(2).any()
    # rows 3..15 of the output matrix are tiny-cuda-nn's padding rows: present in the dump, absent from the model
    n_mlp = onn.n_mlp
    assert np.array_equal(t0[:n_mlp], a.GetParams(0)[:n_mlp]) and t0[n_mlp:n_mlp + 13 * width].any()
    assert np.array_equal(t0[n_mlp + 13 * width:], a.GetParams(0)[n_mlp:])
    rng = np.random.default_rng(3)
    dump = (t0 * 1.25 + rng.standard_normal(t0.size).astype(np.float32) * 0.003).astype(np.float32)
    b = api.NeuralRadianceCache(api.AppConfig(seed=7, **model))
    assert not np.array_equal(b.GetParamsTcnn(0), t0)
    for which in (0, 1):
        b.SetParamsTcnn(which, dump)
        a.SetParamsTcnn(which, dump)
        assert np.array_equal(b.GetParamsTcnn(which), dump)
    x = torch_gpu.from_numpy(queries(512, seed=9)).cuda()
    oa, ob = torch_gpu.empty((512, 3), device="cuda"), torch_gpu.empty((512, 3), device="cuda")
    a.Infer(x, oa, True)
    b.Infer(x, ob, True)
    assert torch_gpu.equal(oa, ob)
    a.Destroy()
    b.Destroy()


@pytest.mark.parametrize("use_ema", [True, False])
def test_inference_matches_oracle(cache, orc, torch_gpu, use_ema):
    onn = orc.nn_create()
    randomize(cache, onn)
    x = queries(8192)
    d_in = torch_gpu.from_numpy(x).cuda()
    d_out = torch_gpu.empty((x.shape[0], 3), device="cuda")
    cache.Infer(d_in, d_out, useEma=use_ema)
    got = d_out.cpu().numpy()
    ref = onn.forward(x, use_ema=use_ema, mode=1)
    assert np.isfinite(got).all()
    assert rel(got, ref) < 2e-3
    assert np.abs(got - ref).max() < 2e-2 * max(1.0, np.abs(ref).max())
    # and it tracks the fp32 network (precision >= the reference's fp16 accumulate)
    assert rel(got, onn.forward(x, use_ema=use_ema, mode=0)) < 1e-2


@pytest.mark.parametrize("n", [1, 16, 31, 32, 33, 100, 1000, 4097])
def test_inference_ragged_sizes(cache, orc, torch_gpu, n):
    onn = orc.nn_create()
    x = queries(n, seed=n)
    d_in = torch_gpu.from_numpy(x).cuda()
    d_out = torch_gpu.full((n + 8, 3), -7.0, device="cuda")
    cache.Infer(d_in, d_out[:n], True)
    out = d_out.cpu().numpy()
    assert (out[n:] == -7.0).all()                       # nothing written past the tail
    assert rel(out[:n], onn.forward(x, True, 1)) < 2e-3


def test_sample_independence_and_determinism(cache, torch_gpu):
    x = queries(2048, seed=3)
    d_in = torch_gpu.from_numpy(x).cuda()
    a, b = torch_gpu.empty((2048, 3), device="cuda"), torch_gpu.empty((2048, 3), device="cuda")
    cache.Infer(d_in, a, True)
    perm = torch_gpu.randperm(2048, device="cuda")
    cache.Infer(d_in[perm].contiguous(), b, True)
    assert torch_gpu.equal(a[perm], b)                   # a sample's result does not depend on its tile / lane


def test_infer_and_train_batch_slicing_and_filter(api, orc, torch_gpu):
    """Init/InferAndTrain (src/NeuralRadianceCache.cu:42-103): batches of 2^log2InferBatchSize + remainder, host filter"""
    cfg = api.AppConfig(log2_infer_batch_size=10, log2_train_batch_size=8, train_batch_count=2)
    c = api.NeuralRadianceCache(cfg)
    n = 2560 + 512
    x = queries(n, seed=9, nan_frac=0.0)
    d_in = torch_gpu.from_numpy(x).cuda()
    d_out = torch_gpu.zeros((n, 3), device="cuda")
    d_tin = torch_gpu.from_numpy(queries(512, seed=10, nan_frac=0.0)).cuda()
    d_tt = torch_gpu.rand((512, 3), device="cuda")
    c.Init(n, d_in, d_out, d_tin, d_tt)
    assert c.GetInferBatchSize() == 1024 and c.GetTrainBatchSize() == 256
    assert c.GetInferBatchCount() == 3 + 0 and c.GetTrainBatchCount() == 2        # 3072 = 3 x 1024
    c.InferAndTrain(np.array([1, 0, 1], np.uint32), False)
    out = d_out.cpu().numpy()
    ref = orc.nn_create().forward(x, True, 1)
    assert rel(out[:1024], ref[:1024]) < 2e-3 and rel(out[2048:], ref[2048:]) < 2e-3
    assert (out[1024:2048] == 0).all()                   # filtered batch untouched
    with pytest.raises(RuntimeError, match="multiple of 16"):
        c.Init(1001, d_in, d_out, d_tin, d_tt)
    c.InferAndTrain(None, True)                          # two train batches -> two optimizer steps
    assert c.GetStep() == 2 and np.isfinite(c.GetLoss())
    c.Destroy()


def gradient_agreement(c, onn, x, t, torch, max_flip_tiles=4):
    """How the kernels' gradient of batch (x, t) agrees with the oracle's.  Both round at the same points (weights, activations, deltas to
    fp16) and sum wider (fp32 MFMA accumulators / double), so they agree to ~7e-6 -- EXCEPT for a ray with a pre-activation within fp32
    accumulation error of zero: its ReLU decision can fall either way, and the neuron's whole delta row for that ray appears or vanishes
    (one such ray in 1 024 moves the batch's gradient by 1e-4..1e-3; profiles/r06_grad_drift.txt: present for about half of the seeds).
    Returns (overall rel-L2, per-layer rel-L2, rel-L2 of the BULK = the batch without the 32-ray tiles that hold such a ray, their count)."""
    n = x.shape[0]
    d_x, d_t = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    c.Backward(d_x, d_t)
    loss = c.GetLoss()
    g = c.GetParams(4).astype(np.float64) / 128.0                            # the device vector carries loss_scale
    loss_ref = onn.backward(x, t)
    g_ref = np.array(onn.buffer(4), np.float64)
    per_layer, off = [], 0
    for o, i in [(64, 80)] + [(64, 64)] * 5 + [(3, 64)]:
        per_layer.append(rel(g[off:off + o * i], g_ref[off:off + o * i]))
        off += o * i
    tiles = []
    for a in range(0, n, 32):
        c.Backward(d_x[a:a + 32].contiguous(), d_t[a:a + 32].contiguous(), nNorm=n)
        gt = c.GetParams(4).astype(np.float64) / 128.0
        onn.backward(x[a:a + 32], t[a:a + 32], n_norm=n)
        gr = np.array(onn.buffer(4), np.float64)
        tiles.append((float(np.linalg.norm(gt - gr)), gt, gr))
    med = float(np.median([e for e, _, _ in tiles]))
    flips = [k for k, (e, _, _) in enumerate(tiles) if e > 8.0 * med]
    keep = [k for k in range(len(tiles)) if k not in flips]
    bulk, bulk_ref = sum(tiles[k][1] for k in keep), sum(tiles[k][2] for k in keep)
    assert len(flips) <= max_flip_tiles, "%d tiles disagree: not a handful of ReLU ties" % len(flips)
    return dict(overall=rel(g, g_ref), per_layer=per_layer, bulk=rel(bulk, bulk_ref), flip_tiles=len(flips), loss=loss, loss_ref=loss_ref)


@pytest.mark.parametrize("loss_fn,loss_id", [("RelativeL2Luminance", 0), ("L2", 1), ("RelativeL2", 2), ("L1", 3), ("Mape", 4), ("Smape", 5),
                                             ("LogL1", 6)])
def test_backward_matches_oracle(api, orc, torch_gpu, loss_fn, loss_id):
    """SURVEY 8c asks for a stated tolerance; measured (profiles/r06_grad_drift.txt, 13 seeds x 2 batches): 6e-6..9e-6 when no ray's ReLU
    decision is a tie, up to 9.7e-4 overall / 9e-3 in the first layer with one to three such rays.  Bounds: the bulk of the batch <= 3e-5 (the
    arithmetic itself: 4 x the measured floor), at most four tiles with a tie, overall <= 2e-3, any layer <= 1e-2, loss <= 2e-5 relative."""
    c = api.NeuralRadianceCache(api.AppConfig(loss_fn=loss_fn))
    onn = orc.nn_create(loss_id=loss_id)
    randomize(c, onn, seed=7, scale=1.0)
    n = 2048
    x = queries(n, seed=21)
    rng = np.random.default_rng(22)
    t = (rng.random((n, 3), dtype=np.float32) * 2).astype(np.float32)
    a = gradient_agreement(c, onn, x, t, torch_gpu)
    assert abs(a["loss"] - a["loss_ref"]) < 2e-5 * abs(a["loss_ref"]), a
    assert a["bulk"] < 3e-5, a
    assert a["overall"] < 2e-3, a
    # per-layer check: a transposed or permuted weight-gradient tile would show up as O(1) error in one layer
    assert max(a["per_layer"]) < 1e-2, a
    c.Destroy()


def test_backward_of_a_batch_longer_than_one_round_of_workgroups(api, torch_gpu):
    """k_train_fwd_bwd_light alternates its two weight images in ONE LDS buffer per round of tiles (a workgroup per four 32-ray tiles,
    at most one workgroup per CU): a batch of 40 992 rays = 1 281 tiles takes two rounds on 256 CUs, the last one ragged (one tile in a
    workgroup of four waves).  Its gradient is the sum of its parts' gradients -- each part small enough for one round -- against the same
    normaliser, and the run is bitwise reproducible"""
    c = api.NeuralRadianceCache(api.AppConfig())
    n = 40992
    x = torch_gpu.from_numpy(queries(n, seed=51, nan_frac=0.0)).cuda()
    t = torch_gpu.rand((n, 3), device="cuda")
    c.Backward(x, t)
    full, loss_full = c.GetParams(4), c.GetLoss()
    c.Backward(x, t)
    assert np.array_equal(full, c.GetParams(4))
    parts, loss = np.zeros_like(full, np.float64), 0.0
    for a in range(0, n, 8192):
        b = min(a + 8192, n)
        c.Backward(x[a:b].contiguous(), t[a:b].contiguous(), nNorm=n)
        parts += c.GetParams(4)
        loss += c.GetLoss()
    assert rel(parts.astype(np.float32), full) < 1e-5 and abs(loss - loss_full) < 1e-5 * abs(loss_full)
    c.Destroy()


def test_backward_is_bitwise_reproducible(cache, torch_gpu):
    x = torch_gpu.from_numpy(queries(4096, seed=4, nan_frac=0.0)).cuda()
    t = torch_gpu.rand((4096, 3), device="cuda")
    cache.Backward(x, t)
    g1 = cache.GetParams(4)
    cache.Backward(x, t)
    assert np.array_equal(g1, cache.GetParams(4))         # fixed-order slab reduction, no float atomics


def test_training_steps_match_oracle(cache, orc, torch_gpu):
    """weights after ONE step <= 1e-3.  Every step starts from the same state on both sides (the oracle takes over the device's
    four vectors and step count), so the bound is per step and nothing accumulates.  Adam's step is ~lr x sign(gradient) for every weight: one
    whose gradient is fp16 noise around zero may move the other way on the two sides (a single such weight is 7e-4 of the update's norm), so
    those are COUNTED and bounded -- at most 8, each with |gradient| < 1e-4 of the largest -- and the 1e-3 holds for the others, as in
    __graft_entry__.smoke().  First moment <= 2e-3 (it is 0.1 x the gradient, whose bound with a ReLU tie is 2e-3), second <= 4e-3 (0.001 x its square)."""
    onn = orc.nn_create()
    n = 1024
    x = queries(n, seed=31, nan_frac=0.05)
    rng = np.random.default_rng(32)
    d_x = torch_gpu.from_numpy(x).cuda()
    for step in range(3):
        for which in range(4):                   # identical state in front of the step
            onn.buffer(which)[:] = cache.GetParams(which)
        onn.set_step(cache.GetStep())
        w0 = cache.GetParams(0).copy()
        t = rng.random((n, 3), dtype=np.float32)
        cache.Backward(d_x, torch_gpu.from_numpy(t).cuda())
        cache.OptimizerStep()
        loss_ref = onn.backward(x, t)
        g_ref = np.array(onn.buffer(4))
        onn.optimizer_step()
        assert abs(cache.GetLoss() - loss_ref) < 2e-5 * abs(loss_ref)
        w, w_ref = cache.GetParams(0), np.array(onn.buffer(0))
        flipped = np.sign(w - w0) != np.sign(w_ref - w0)
        assert flipped.sum() <= 8 and (np.abs(g_ref[flipped]) < 1e-4 * np.abs(g_ref).max()).all(), (step, int(flipped.sum()))
        assert rel(w[~flipped], w_ref[~flipped]) < 1e-3, step
        assert rel((w - w0)[~flipped], (w_ref - w0)[~flipped]) < 2e-2, step        # the update itself (~lr per weight), not only the weights
        assert rel(cache.GetParams(1), onn.buffer(1)) < 1e-3, step                  # EMA
        assert rel(cache.GetParams(2), onn.buffer(2)) < 2e-3 and rel(cache.GetParams(3), onn.buffer(3)) < 4e-3, step
    assert cache.GetStep() == 3
    # inference now uses the updated EMA weights
    out = torch_gpu.empty((n, 3), device="cuda")
    cache.Infer(d_x, out, True)
    for which in range(4):
        onn.buffer(which)[:] = cache.GetParams(which)
    assert rel(out.cpu().numpy(), onn.forward(x, True, 1)) < 2e-3


def test_sharded_backward_sums_to_full_batch(api, torch_gpu):
    """multi-GPU exchange step on one device: two half batches against the global normaliser sum to the full gradient"""
    c = api.NeuralRadianceCache(api.AppConfig())
    n = 2048
    x = torch_gpu.from_numpy(queries(n, seed=41, nan_frac=0.0)).cuda()
    t = torch_gpu.rand((n, 3), device="cuda")
    c.Backward(x, t)
    full, loss_full = c.GetParams(4), c.GetLoss()
    c.Backward(x[:1024].contiguous(), t[:1024].contiguous(), nNorm=n)
    a, la = c.GetParams(4), c.GetLoss()
    c.Backward(x[1024:].contiguous(), t[1024:].contiguous(), nNorm=n)
    b, lb = c.GetParams(4), c.GetLoss()
    assert rel(a + b, full) < 1e-5 and abs(la + lb - loss_full) < 1e-5 * abs(loss_full)
    c.Destroy()


@pytest.mark.parametrize("model", [dict(), dict(optimizer="SGD"), dict(nn_width=128, nn_depth=8), dict(nn_width=16, nn_depth=2),
                                   dict(pos_id=1, dir_id=1, nn_width=32, nn_depth=3)],
                         ids=["north-star", "sgd", "8x128", "2x16", "identity-3x32"])
def test_one_launch_optimizer_equals_the_three_launch_path_bitwise(api, torch_gpu, model, monkeypatch):
    """k_opt_pack (update + scatter into the three fragment images + loss publication in one launch) against k_adam_ema /
    k_sgd_ema + k_pack (NRC_DEBUG=no_fused_opt, read when the cache is created): gradients, weights, EMA, moments and both
    inference paths stay identical to the last bit over four training steps"""
    nrc_debug(monkeypatch)
    a = api.NeuralRadianceCache(api.AppConfig(**model))
    nrc_debug(monkeypatch, no_fused_opt=True)
    b = api.NeuralRadianceCache(api.AppConfig(**model))
    nrc_debug(monkeypatch)
    n = 4096
    x = torch_gpu.from_numpy(queries(n, seed=71, nan_frac=0.0)).cuda()
    t = torch_gpu.rand((n, 3), device="cuda")
    for step in range(4):
        state = []
        for c in (a, b):
            c.Backward(x, t)
            g = c.GetParams(4)
            c.OptimizerStep()
            o_ema, o_w = torch_gpu.empty((n, 3), device="cuda"), torch_gpu.empty((n, 3), device="cuda")
            c.Infer(x, o_ema, True)
            c.Infer(x, o_w, False)
            state.append([g] + [c.GetParams(k) for k in range(4)] + [o_ema.cpu().numpy(), o_w.cpu().numpy()])
        for name, p, q in zip(("grad", "w", "ema", "m", "v", "infer(ema)", "infer(w)"), *state):
            assert np.array_equal(p.view(np.uint32), q.view(np.uint32)), (step, name)
    assert abs(a.GetLoss() - b.GetLoss()) == 0.0
    a.Destroy()
    b.Destroy()


@pytest.mark.parametrize("optimizer", ["Adam", "SGD"])
def test_hashgrid_one_launch_table_optimizer_equals_the_separate_kernels_bitwise(api, torch_gpu, optimizer, monkeypatch):
    """HashGrid model: k_opt_pack + k_grid_opt (table gradient read from the packed fp16 table, update, fp16 gather copies) against
    k_grid_grad_f32 + k_adam_ema / k_sgd_ema + k_pack + k_pack_grid (NRC_DEBUG=no_fused_opt).  The packed-fp16 atomics of the table
    gradient sum in a different order in every run, so cache B never runs its own backward: it is handed A's gradient vector
    (which also exercises the path that reads the fp32 vector after an exchange) -- weights, EMA, moments and both inference paths
    (matrix images and table copies) then agree to the last bit, step after step"""
    kw = dict(pos_id=0, hashgrid_log2_size=14, nn_depth=3, optimizer=optimizer)
    nrc_debug(monkeypatch)
    a = api.NeuralRadianceCache(api.AppConfig(**kw))
    a2 = api.NeuralRadianceCache(api.AppConfig(**kw))            # fused too, but fed through the fp32 vector
    nrc_debug(monkeypatch, no_fused_opt=True)
    b = api.NeuralRadianceCache(api.AppConfig(**kw))
    nrc_debug(monkeypatch)
    n = 2048
    rng = np.random.default_rng(9)
    t = torch_gpu.rand((n, 3), device="cuda")
    for step in range(5):
        # a new batch every step: entries touched in one step and not in the next take the optimizer's zero-gradient path with moments
        # and a weight that have moved (k_grid_opt2 leaves their fp16 weight copy alone there)
        x = torch_gpu.from_numpy(rng.random((n, 5), dtype=np.float32)).cuda()
        a.Backward(x, t)
        g = a.GetParams(4)
        assert np.count_nonzero(g[-2 * 16384:]) > 0
        a.OptimizerStep()                       # table gradient from the packed fp16 table
        state = []
        for c in (a2, b):
            c.SetParams(4, g)                   # table gradient from the fp32 vector
            c.OptimizerStep()
        for c in (a, a2, b):
            o_ema, o_w = torch_gpu.empty((n, 3), device="cuda"), torch_gpu.empty((n, 3), device="cuda")
            c.Infer(x, o_ema, True)
            c.Infer(x, o_w, False)
            state.append([c.GetParams(k) for k in range(4)] + [o_ema.cpu().numpy(), o_w.cpu().numpy()])
        for other in (1, 2):
            for name, p, q in zip(("w", "ema", "m", "v", "infer(ema)", "infer(w)"), state[0], state[other]):
                assert np.array_equal(p.view(np.uint32), q.view(np.uint32)), (step, other, name)
    for c in (a, a2, b):
        c.Destroy()


def test_hashgrid_optimizer_reads_the_gradient_vector_the_caller_reduced(api, torch_gpu):
    """the documented stand-alone protocol (include/nrc_hpm.h): nrc_cache_backward -> all-reduce of nrc_cache_grad_ptr ->
    nrc_cache_optimizer_step.  For a HashGrid model the optimizer must take the TABLE gradient from that fp32 vector too, not from
    the packed fp16 table of the local backward pass: a caller that zeroes the vector gets an untouched table (Adam skips table
    entries whose gradient is zero), one that leaves it alone gets the step of a cache that never asked for the pointer"""
    kw = dict(pos_id=0, hashgrid_log2_size=12, nn_depth=3)
    a, b, c = (api.NeuralRadianceCache(api.AppConfig(**kw)) for _ in range(3))
    n = 1024
    rng = np.random.default_rng(3)
    x = torch_gpu.from_numpy(rng.random((n, 5), dtype=np.float32)).cuda()
    t = torch_gpu.rand((n, 3), device="cuda")
    w0 = a.GetParams(0).copy()
    g_b, g_c = b.GradTensor(), c.GradTensor()      # the pointer is handed out BEFORE the backward pass, as a driver would do once
    for k in (a, b, c):
        k.Backward(x, t)
    torch_gpu.cuda.synchronize()
    g_ref = a.GetParams(4)
    table = np.flatnonzero(g_ref != 0)
    table = table[table >= a.ParamCount() - 2 * 16 * (1 << 12)]          # touched table entries (16 levels x <= 2^12 x 2 features)
    assert table.size > 1000
    g_c.zero_()
    torch_gpu.cuda.synchronize()
    for k in (a, b, c):
        k.OptimizerStep()
    wa, wb, wc = a.GetParams(0), b.GetParams(0), c.GetParams(0)
    assert (wa[table] != w0[table]).mean() > 0.9                   # the step moves the touched table entries ...
    assert np.array_equal(wc[table].view(np.uint32), w0[table].view(np.uint32))      # ... unless the caller zeroed their gradient
    # reading the (unmodified) fp32 vector instead of the packed table gives the same step; the packed fp16 atomics of two backward
    # passes sum in different orders, so the comparison is by tolerance
    assert np.abs(wb - wa).max() <= 2.1 * 0.01
    assert np.linalg.norm(wb - wa) <= 0.05 * np.linalg.norm(wa - w0)
    for k in (a, b, c):
        k.Destroy()


def test_loss_decreases_when_training_on_device(api, torch_gpu):
    c = api.NeuralRadianceCache(api.AppConfig())
    x = queries(4096, seed=51, nan_frac=0.0)
    t = np.stack([np.sin(x[:, 0] * 7) * 0.5 + 0.5, x[:, 3] * 0.3 + 0.2, np.full(4096, 0.4, np.float32)], axis=1).astype(np.float32)
    d_x, d_t = torch_gpu.from_numpy(x).cuda(), torch_gpu.from_numpy(t).cuda()
    losses = []
    for _ in range(40):
        c.Backward(d_x, d_t)
        c.OptimizerStep()
        losses.append(c.GetLoss())
    assert np.isfinite(losses).all() and losses[-1] < 0.5 * losses[0]
    c.Destroy()


def test_checkpoint_roundtrip(api, torch_gpu):
    a = api.NeuralRadianceCache(api.AppConfig())
    x = torch_gpu.from_numpy(queries(1024, seed=61, nan_frac=0.0)).cuda()
    t = torch_gpu.rand((1024, 3), device="cuda")
    for _ in range(2):
        a.Backward(x, t)
        a.OptimizerStep()
    sd = a.state_dict()
    b = api.NeuralRadianceCache(api.AppConfig(seed=99))
    b.load_state_dict(sd)
    for c in (a, b):
        c.Backward(x, t)
        c.OptimizerStep()
    assert np.array_equal(a.GetParams(0), b.GetParams(0)) and np.array_equal(a.GetParams(1), b.GetParams(1))
    oa, ob = torch_gpu.empty((1024, 3), device="cuda"), torch_gpu.empty((1024, 3), device="cuda")
    a.Infer(x, oa, True)
    b.Infer(x, ob, True)
    assert torch_gpu.equal(oa, ob)
    a.Destroy()
    b.Destroy()


def test_checkpoint_file_roundtrip_and_rejections(api, torch_gpu, tmp_path):
    """nrc_cache_save_checkpoint / _load_checkpoint: a cache of another seed resumes bit for bit; a file of another model, a truncated
    file and a file with trailing bytes are refused and leave the cache as it was"""
    a = api.NeuralRadianceCache(api.AppConfig())
    x = torch_gpu.from_numpy(queries(1024, seed=61, nan_frac=0.0)).cuda()
    t = torch_gpu.rand((1024, 3), device="cuda")
    for _ in range(3):
        a.Backward(x, t)
        a.OptimizerStep()
    path = str(tmp_path / "nrc.ckpt")
    a.SaveCheckpoint(path)
    assert (tmp_path / "nrc.ckpt").stat().st_size == 64 + 4 * 4 * 26624
    b = api.NeuralRadianceCache(api.AppConfig(seed=99))
    b.LoadCheckpoint(path)
    assert b.GetStep() == a.GetStep() == 3
    for c in (a, b):
        c.Backward(x, t)
        c.OptimizerStep()
    for which in range(4):
        assert np.array_equal(a.GetParamsTcnn(which), b.GetParamsTcnn(which))
    before = b.GetParams(0)
    other = api.NeuralRadianceCache(api.AppConfig(nn_width=32))
    with pytest.raises(RuntimeError, match="not of this model"):
        other.LoadCheckpoint(path)
    raw = (tmp_path / "nrc.ckpt").read_bytes()
    (tmp_path / "short.ckpt").write_bytes(raw[:-100])
    (tmp_path / "long.ckpt").write_bytes(raw + b"x")
    with pytest.raises(RuntimeError, match="truncated"):
        b.LoadCheckpoint(str(tmp_path / "short.ckpt"))
    with pytest.raises(RuntimeError, match="trailing"):
        b.LoadCheckpoint(str(tmp_path / "long.ckpt"))
    with pytest.raises(RuntimeError, match="cannot read"):
        b.LoadCheckpoint(str(tmp_path / "absent.ckpt"))
    assert np.array_equal(b.GetParams(0), before)
    for c in (a, b, other):
        c.Destroy()


@pytest.mark.parametrize("model", [dict(), dict(pos_id=0, hashgrid_log2_size=10, nn_depth=3)], ids=["fused", "hashgrid"])
def test_sgd_optimizer_matches_oracle(api, orc, torch_gpu, model):
    """optimizer "SGD" (second positional argument of the reference's command line, passed through to tiny-cuda-nn inside the
    EMA wrapper): w -= lr * (g + 1e-8 w) on every parameter; three steps against the oracle"""
    cfg = api.AppConfig(optimizer="SGD", learning_rate=0.05, log2_train_batch_size=10, **model)
    c = api.NeuralRadianceCache(cfg)
    onn = orc.nn_create(pos_id=cfg.pos_id, dir_id=cfg.dir_id, width=cfg.nn_width, depth=cfg.nn_depth, lr=0.05,
                        hashgrid_log2_size=model.get("hashgrid_log2_size", 0), optimizer="SGD")
    rng = np.random.default_rng(11)
    w = rng.uniform(-0.3, 0.3, c.ParamCount()).astype(np.float32)
    onn.buffer(0)[:] = w
    onn.buffer(1)[:] = w
    c.SetParams(0, w)
    c.SetParams(1, w)
    for step in range(3):
        x = rng.random((1024, 5), dtype=np.float32)
        t = rng.random((1024, 3), dtype=np.float32)
        c.Backward(torch_gpu.from_numpy(x).cuda(), torch_gpu.from_numpy(t).cuda())
        loss_ref = onn.backward(x, t)
        assert abs(c.GetLoss() - loss_ref) < 5e-3 * abs(loss_ref)
        c.OptimizerStep()
        onn.optimizer_step()
    w3, w3_ref = c.GetParams(0), np.array(onn.buffer(0))
    assert rel(w3 - w, w3_ref - w) < 3e-2                     # the update itself, not just the weights
    assert rel(c.GetParams(1), np.array(onn.buffer(1))) < 1e-3
    assert not np.allclose(w3, w)
    c.Destroy()


def test_unsupported_configurations_fail_loudly(api, torch_gpu):
    for kw in (dict(pos_id=4), dict(dir_id=3), dict(pos_id=0, hashgrid_log2_size=30), dict(nn_width=96), dict(nn_width=8), dict(nn_depth=0), dict(optimizer="Shampoo"),
               dict(loss_fn="CrossEntropy"), dict(loss_fn="Variance"), dict(log2_infer_batch_size=0), dict(train_ring_buf_size=-1.0)):
        with pytest.raises(RuntimeError, match="SkyRenderer ERROR"):
            api.NeuralRadianceCache(api.AppConfig(**kw))


def test_full_size_inference_properties(cache, orc, torch_gpu):
    """BASELINE size (1920x1080 queries): finite everywhere, spot-checked against the oracle"""
    n = 1920 * 1080
    g = torch_gpu.Generator(device="cuda").manual_seed(1)
    x = torch_gpu.rand((n, 5), device="cuda", generator=g)
    x[:, :3] += 31.0
    out = torch_gpu.empty((n, 3), device="cuda")
    cache.Infer(x, out, True)
    assert bool(torch_gpu.isfinite(out).all())
    idx = torch_gpu.randint(0, n, (2048,), device="cuda", generator=g)
    ref = orc.nn_create().forward(x[idx].cpu().numpy(), True, 1)
    assert rel(out[idx].cpu().numpy(), ref) < 2e-3


GENERIC = [  # (posID, dirID, width, depth): everything except the fused 3/0/64/6 model runs the generic kernels
    (1, 0, 64, 6), (2, 2, 64, 4), (3, 1, 64, 2), (3, 0, 64, 5), (3, 0, 128, 8), (1, 1, 128, 3), (2, 0, 128, 1),
    (3, 0, 32, 4), (2, 1, 32, 2),
    (3, 0, 16, 3), (1, 1, 16, 2), (0, 0, 16, 2),      # nnWidth 16: tiny-cuda-nn's smallest FullyFusedMLP width, on the 32-row MFMA tiles
]


@pytest.mark.parametrize("pos_id,dir_id,width,depth", GENERIC)
def test_generic_models_match_oracle(api, orc, torch_gpu, pos_id, dir_id, width, depth):
    """other encodings (src/AppConfig.cpp:11-87: Identity / TriangleWave / Frequency x OneBlob / Identity / TriangleWave),
    widths 64/128 (BASELINE configs[4]: 8x128) and depths: inference, gradients and one optimizer step vs the oracle"""
    hg = 11 if pos_id == 0 else 0
    c = api.NeuralRadianceCache(api.AppConfig(pos_id=pos_id, dir_id=dir_id, nn_width=width, nn_depth=depth, hashgrid_log2_size=hg))
    onn = orc.nn_create(pos_id=pos_id, dir_id=dir_id, width=width, depth=depth, hashgrid_log2_size=hg)
    assert c.ParamCount() == onn.n_params
    assert np.array_equal(c.GetParams(0), onn.buffer(0))
    randomize(c, onn, seed=3, scale=1.0)
    n = 1024 + 37
    x = queries(n, seed=pos_id * 7 + dir_id, nan_frac=0.1 if dir_id == 0 else 0.0)   # only OneBlob absorbs the NaN phi of quirk Q5
    if pos_id in (0, 1):
        x[:, :3] -= 31.0                                     # keep Identity / HashGrid positions in [0, 1)
    d_out = torch_gpu.empty((n, 3), device="cuda")
    c.Infer(torch_gpu.from_numpy(x).cuda(), d_out, useEma=True)
    got, ref = d_out.cpu().numpy(), onn.forward(x, True, 1)
    assert np.isfinite(got).all()
    assert rel(got, ref) < 3e-3
    m = 1024
    rng = np.random.default_rng(5)
    t = rng.random((m, 3), dtype=np.float32)
    c.Backward(torch_gpu.from_numpy(x[:m].copy()).cuda(), torch_gpu.from_numpy(t).cuda())
    loss_ref = onn.backward(x[:m], t)
    assert abs(c.GetLoss() - loss_ref) < 3e-3 * abs(loss_ref)
    g, g_ref = c.GetParams(4) / 128.0, np.array(onn.buffer(4))
    assert rel(g, g_ref) < 2e-2
    c.OptimizerStep()
    onn.optimizer_step()
    assert rel(c.GetParams(0), onn.buffer(0)) < 3e-3
    c.Destroy()


@pytest.mark.parametrize("pos_id,dir_id,width,depth", [(3, 0, 128, 8), (1, 1, 128, 3), (3, 0, 64, 5), (2, 2, 64, 1), (3, 0, 32, 4), (0, 0, 64, 2),
                                                       (0, 0, 128, 2), (3, 0, 16, 3)])
def test_training_kernels_agree(api, torch_gpu, monkeypatch, pos_id, dir_id, width, depth):
    """k_train_gen2 (round 4: every layer's rows split over the waves of a workgroup, operands exchanged through LDS, weight fragments
    read straight from L2) against k_train_gen (NRC_DEBUG=train_gen_old=1): each output element is the same sequence of MFMAs, so loss and
    gradient agree bit for bit -- one and two tiles per sample group, full and ragged batches (the last workgroup half empty)"""
    hg = 11 if pos_id == 0 else 0
    rng = np.random.default_rng(13)
    for n in (16384, 2048 + 96, 32):
        xq = queries(n, seed=5, nan_frac=0.0)
        if pos_id in (0, 1):
            xq[:, :3] -= 31.0
        x = torch_gpu.from_numpy(xq).cuda()
        t = torch_gpu.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda()
        got = {}
        for mode in ("old", "1", "2"):
            if mode == "old":
                nrc_debug(monkeypatch, train_gen_old=1)
            else:
                nrc_debug(monkeypatch, train_gen_old=0, train_gen_nt=int(mode))      # (unset: k_train_gen up to 64 neurons, k_train_gen2 for 128)
            c = api.NeuralRadianceCache(api.AppConfig(pos_id=pos_id, dir_id=dir_id, nn_width=width, nn_depth=depth, hashgrid_log2_size=hg))
            c.Backward(x, t)
            got[mode] = (c.GetParams(4).copy(), c.GetLoss())
            c.Destroy()
        g_old, l_old = got["old"]
        assert np.isfinite(g_old).all() and np.abs(g_old).max() > 0.0
        for mode in ("1", "2"):
            g, l = got[mode]
            assert l == l_old, (n, mode)
            if pos_id == 0:      # the table gradient is summed with fp16 atomics in no fixed order: the MLP part is exact
                nm = c_mlp_params(width, depth, 48)
                assert np.array_equal(g[:nm], g_old[:nm]), (n, mode)
                assert rel(g, g_old) < 2e-2
            else:
                assert np.array_equal(g, g_old), (n, mode)


def c_mlp_params(width, depth, enc):
    return enc * width + (depth - 1) * width * width + width * 3


@pytest.mark.parametrize("width,depth,exact", [(64, 6, True), (128, 8, False), (32, 3, False)], ids=["6x64", "8x128", "3x32"])
def test_weight_gradient_kernels_agree_at_full_batch(api, torch_gpu, monkeypatch, width, depth, exact):
    """k_wgrad2 (round 4: one wave per 32-row block of a layer's delta, up to four accumulators, K-chunks sized for the launch)
    against round 3's k_wgrad (NRC_DEBUG=wgrad_old=1, 128-sample chunks) on a full 16 384-ray batch and on a ragged one: the same products
    summed over other chunk boundaries -- fp32 rounding apart; for the 6x64 model the chunks are the same and so is every bit"""
    rng = np.random.default_rng(11)
    grads = {}
    for n in (16384, 4096 + 96):
        x = torch_gpu.from_numpy(queries(n, seed=3, nan_frac=0.0)).cuda()
        t = torch_gpu.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda()
        for old in (False, True):
            nrc_debug(monkeypatch, wgrad_old=1 if old else 0)      # (unset: k_wgrad up to 64 neurons, k_wgrad2 for 128)
            c = api.NeuralRadianceCache(api.AppConfig(nn_width=width, nn_depth=depth))
            c.Backward(x, t)
            grads[old] = c.GetParams(4).copy()
            c.Destroy()
        assert np.isfinite(grads[False]).all() and np.abs(grads[False]).max() > 0.0
        if exact and n == 16384:
            assert np.array_equal(grads[False], grads[True])
        else:
            assert rel(grads[False], grads[True]) < 2e-6


def test_random_model_configurations_match_oracle(api, orc, torch_gpu):
    """a seeded sweep over the configuration space the command line spans -- encodings x width {32, 64, 128} x depth 1..9 x
    loss x optimizer: inference, loss, gradients and the first optimizer step against the oracle"""
    rng = np.random.default_rng(123)
    losses = ["RelativeL2Luminance", "L2", "RelativeL2"]
    for k in range(20):
        pos, d = (0 if k >= 16 else int(rng.choice([1, 2, 3]))), int(rng.integers(0, 3))       # the last four: HashGrid
        w, depth = int(rng.choice([32, 64, 128])), int(rng.integers(1, 10))
        loss, opt = int(rng.integers(0, 3)), str(rng.choice(["Adam", "SGD"]))
        hg = int(rng.integers(8, 14)) if pos == 0 else 0
        tag = (pos, d, w, depth, losses[loss], opt, hg)
        c = api.NeuralRadianceCache(api.AppConfig(pos_id=pos, dir_id=d, nn_width=w, nn_depth=depth, loss_fn=losses[loss], optimizer=opt,
                                                  hashgrid_log2_size=hg))
        onn = orc.nn_create(pos_id=pos, dir_id=d, width=w, depth=depth, loss_id=loss, optimizer=opt, hashgrid_log2_size=hg)
        randomize(c, onn, seed=3, scale=1.0)
        n = 512 + 37
        x = queries(n, seed=pos * 7 + d, nan_frac=0.1 if d == 0 else 0.0)
        if pos == 1:
            x[:, :3] -= 31.0
        if pos == 0:
            x[:, :3] = np.random.default_rng(k).random((n, 3), dtype=np.float32)       # the grid is designed for [0,1)
        out = torch_gpu.empty((n, 3), device="cuda")
        c.Infer(torch_gpu.from_numpy(x).cuda(), out, useEma=True)
        assert rel(out.cpu().numpy(), onn.forward(x, True, 1)) < 3e-3, tag
        t = np.random.default_rng(5).random((512, 3), dtype=np.float32)
        c.Backward(torch_gpu.from_numpy(x[:512].copy()).cuda(), torch_gpu.from_numpy(t).cuda())
        loss_ref = onn.backward(x[:512], t)
        assert abs(c.GetLoss() - loss_ref) < 3e-3 * abs(loss_ref), tag
        assert rel(c.GetParams(4) / 128.0, np.array(onn.buffer(4))) < (5e-2 if pos == 0 else 2e-2), tag     # fp16 table atomics
        w0 = c.GetParams(0).copy()
        c.OptimizerStep()
        onn.optimizer_step()
        assert rel(c.GetParams(0) - w0, np.array(onn.buffer(0)) - w0) < (1e-1 if pos == 0 else 6e-2 if opt == "Adam" else 2e-2), tag
        c.Destroy()


@pytest.mark.parametrize("dir_id,width,depth,log2", [(0, 64, 6, 12), (1, 64, 2, 10), (0, 128, 4, 14), (0, 64, 6, 19)],
                         ids=["2^12", "2^10-identity-dir", "2^14-128wide", "reference-default-2^19"])
def test_hashgrid_model_matches_oracle(api, orc, torch_gpu, dir_id, width, depth, log2):
    """reference default encoding, AppConfig posID 0 (src/AppConfig.cpp:19-27, src/main.cu:435): HashGrid forward,
    dL/d(table) scatter, Adam that skips untouched entries, EMA table for inference.  The last case is the reference's real
    model -- 2^19 entries per hashed level, 14.2 M table parameters, 6x64 network -- against the oracle on 2 048 samples"""
    c = api.NeuralRadianceCache(api.AppConfig(pos_id=0, dir_id=dir_id, nn_width=width, nn_depth=depth, hashgrid_log2_size=log2))
    onn = orc.nn_create(pos_id=0, dir_id=dir_id, width=width, depth=depth, hashgrid_log2_size=log2)
    assert c.ParamCount() == onn.n_params
    assert np.array_equal(c.GetParams(0), onn.buffer(0))
    rng = np.random.default_rng(17)
    # O(0.3) table entries so that the grid features matter
    w = np.array(onn.buffer(0))
    w[onn.n_mlp:] = (rng.standard_normal(onn.n_params - onn.n_mlp) * 0.3).astype(np.float32)
    e = w.copy()
    e[onn.n_mlp:] += (rng.standard_normal(onn.n_params - onn.n_mlp) * 0.05).astype(np.float32)
    onn.buffer(0)[:] = w
    onn.buffer(1)[:] = e
    c.SetParams(0, w)
    c.SetParams(1, e)
    n = 2048
    x = rng.random((n, 5), dtype=np.float32)              # positions in [0,1): what the encoding is designed for
    x[:256, :3] += 31.0                                   # and a block in the quirk-Q3 range (indices wrap)
    for use_ema in (True, False):
        out = torch_gpu.empty((n, 3), device="cuda")
        c.Infer(torch_gpu.from_numpy(x).cuda(), out, useEma=use_ema)
        ref = onn.forward(x, use_ema, 1)
        assert rel(out.cpu().numpy()[256:], ref[256:]) < 3e-3
        assert rel(out.cpu().numpy()[:256], ref[:256]) < 3e-2     # coordinates ~31: fp32 fractional parts at high levels are coarse
    t = rng.random((n, 3), dtype=np.float32)
    xs = x[256:1280].copy()
    c.Backward(torch_gpu.from_numpy(xs).cuda(), torch_gpu.from_numpy(t[:1024]).cuda())
    loss_ref = onn.backward(xs, t[:1024])
    assert abs(c.GetLoss() - loss_ref) < 3e-3 * abs(loss_ref)
    g, g_ref = c.GetParams(4) / 128.0, np.array(onn.buffer(4))
    assert rel(g[:onn.n_mlp], g_ref[:onn.n_mlp]) < 2e-2
    assert rel(g[onn.n_mlp:], g_ref[onn.n_mlp:]) < 3e-2            # fp32 atomics, fp16 dL/d(feature)
    assert np.array_equal(g[onn.n_mlp:] != 0, g_ref[onn.n_mlp:] != 0) or \
        (np.logical_xor(g[onn.n_mlp:] != 0, g_ref[onn.n_mlp:] != 0).mean() < 1e-3)
    c.OptimizerStep()
    onn.optimizer_step()
    w1, w1_ref = c.GetParams(0), np.array(onn.buffer(0))
    untouched = g_ref[onn.n_mlp:] == 0
    assert np.array_equal(w1[onn.n_mlp:][untouched & (g[onn.n_mlp:] == 0)], w[onn.n_mlp:][untouched & (g[onn.n_mlp:] == 0)])
    assert rel(w1, w1_ref) < 2e-2
    out = torch_gpu.empty((n, 3), device="cuda")
    c.Infer(torch_gpu.from_numpy(x).cuda(), out, useEma=True)          # EMA table after the step
    assert rel(out.cpu().numpy()[256:], onn.forward(x, True, 1)[256:]) < 1e-2
    c.Destroy()


@pytest.mark.parametrize("log2", [19, 16, 12], ids=["2^19", "2^16", "2^12"])
def test_table_gradient_is_exact_whatever_path_its_pairs_take(api, orc, torch_gpu, monkeypatch, log2):
    """k_grid_scatter + k_grid_gather: the pairs of the levels with at least 8 bins of 4 096 entries go to per-bin lists and are summed in
    LDS, the coarse levels and what overflows a list are added into the table's fixed-point shadow -- every sum in 64-bit fixed point,
    i.e. EXACT, and rounded to fp16 once.  So the gradient does not depend on which path a pair takes (NRC_DEBUG=grid_no_bins: everything
    through the shadow) nor on the order of arrival: bit-identical between the two builds of the path and from run to run, on a full
    batch (every list far from full) and on a batch whose samples sit in one corner of the volume (lists overflow into the shadow);
    and it is the oracle's gradient within fp16 rounding."""
    rng = np.random.default_rng(17)
    for n, spread in ((16384, 1.0), (8192, 0.02)):
        xq = queries(n, seed=9, nan_frac=0.0)
        xq[:, :3] = (xq[:, :3] - 31.0) * spread
        x = torch_gpu.from_numpy(xq).cuda()
        tq = rng.random((n, 3), dtype=np.float32)
        t = torch_gpu.from_numpy(tq).cuda()
        g = []
        for no_bins in (False, True, False):
            nrc_debug(monkeypatch, grid_no_bins=no_bins)
            c = api.NeuralRadianceCache(api.AppConfig(pos_id=0, dir_id=0, nn_width=64, nn_depth=2, hashgrid_log2_size=log2))
            c.Backward(x, t)
            first = c.GetParams(4).copy()
            c.Backward(x, t)                                          # (a second step on the same cache: the shadow was left all zero)
            assert np.array_equal(first, c.GetParams(4))
            g.append(first)
            c.Destroy()
        assert np.array_equal(g[0], g[1]) and np.array_equal(g[0], g[2])
        nm = c_mlp_params(64, 2, 48)
        tab = g[0][nm:]
        assert np.isfinite(tab).all() and np.abs(tab).max() > 0.0
        if n == 8192:
            continue
        onn = orc.nn_create(pos_id=0, dir_id=0, width=64, depth=2, hashgrid_log2_size=log2)
        onn.backward(xq[:2048], tq[:2048], n_norm=2048)
        c = api.NeuralRadianceCache(api.AppConfig(pos_id=0, dir_id=0, nn_width=64, nn_depth=2, hashgrid_log2_size=log2))
        c.Backward(x[:2048], t[:2048])
        got = c.GetParams(4)[nm:] / 128.0
        c.Destroy()
        assert rel(got, np.array(onn.buffer(4))[nm:]) < 2e-2


def test_hashgrid_training_is_bitwise_reproducible(api, torch_gpu):
    """the reference's DEFAULT model (HashGrid 2^19) trained for several steps twice from the same seed -- weights, EMA weights and
    Adam moments identical to the last bit (the table gradient's sums are exact: k_grid_gather)"""
    xq = queries(16384, seed=31, nan_frac=0.05)
    xq[:, :3] -= 31.0
    x = torch_gpu.from_numpy(xq).cuda()
    t = torch_gpu.rand((16384, 3), device="cuda", generator=torch_gpu.Generator(device="cuda").manual_seed(4))
    runs = []
    for _ in range(2):
        c = api.NeuralRadianceCache(api.AppConfig(pos_id=0, dir_id=0, nn_width=64, nn_depth=2))
        for _ in range(4):
            c.Backward(x, t)
            c.OptimizerStep()
        runs.append([c.GetParams(k).copy() for k in range(4)] + [np.float32(c.GetLoss())])
        c.Destroy()
    for a, b in zip(*runs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("log2", [12, 19], ids=["2^12", "reference-default-2^19"])
def test_hashgrid_sparse_exchange_lists_add_up_in_rank_order(api, torch_gpu, log2):
    """the multi-GPU exchange of the table gradient on one device: three ranks' backward passes (their shard of the batch against
    the global normaliser) packed as (entry, fp16x2) lists; a list reproduces the rank's dense table gradient exactly, holds every
    entry once, and the lists applied in rank order give ((0 + g0) + g1) + g2 to the last bit -- what every replica computes"""
    from nrc_hpm_renderer_amd import parallel          # the host statement of the protocol (tests/test_dist_gloo.py runs it on two ranks)
    c = api.NeuralRadianceCache(api.AppConfig(pos_id=0, hashgrid_log2_size=log2, log2_train_batch_size=11))
    n, ranks = 2048, 3
    rng = np.random.default_rng(23)
    n_mlp = 64 * 48 + 5 * 64 * 64 + 3 * 64
    n_entries = (c.ParamCount() - n_mlp) // 2
    dense, lists = [], []
    for r in range(ranks):
        x = rng.random((n, 5), dtype=np.float32)
        t = rng.random((n, 3), dtype=np.float32)
        c.Backward(torch_gpu.from_numpy(x).cuda(), torch_gpu.from_numpy(t).cuda(), nNorm=ranks * n)
        g = c.GetParams(4)
        lst = c.GridGradPack()
        count, cap = int(lst[0]), (lst.size - 2) // 2
        assert cap == min(n_entries, n * 16 * 8) and 0 < count <= cap
        ent, val = lst[2::2], lst[3::2]
        assert (ent[count:] == 0xFFFFFFFF).all() and (ent[:count] < n_entries).all()
        assert np.unique(ent[:count]).size == count
        rebuilt = np.zeros((n_entries, 2), np.float32)
        rebuilt[ent[:count]] = val[:count].copy().view(np.float16).reshape(-1, 2).astype(np.float32)
        assert np.array_equal(rebuilt.reshape(-1), g[n_mlp:])
        host = parallel.pack_grid_list(g[n_mlp:].astype(np.float16).view(np.uint32), cap)      # same set, the device's order is free
        assert int(host[0]) == count
        order = np.argsort(ent[:count])
        assert np.array_equal(ent[:count][order], host[2:2 + 2 * count:2]) and np.array_equal(val[:count][order], host[3:3 + 2 * count:2])
        assert count < cap          # the coarse levels' samples share corners: fewer entries than (sample, level, corner) triples
        dense.append(g)
        lists.append(lst)
    c.GridGradApply(lists)
    total = c.GetParams(4)
    expect = np.zeros(2 * n_entries, np.float32)
    for g in dense:
        expect = expect + g[n_mlp:]
    assert np.array_equal(total[n_mlp:], expect)
    assert np.array_equal(parallel.apply_grid_lists(lists, n_entries), expect)
    assert np.array_equal(total[:n_mlp], dense[-1][:n_mlp])       # the matrix part is not the lists' business
    assert not np.array_equal(expect, dense[0][n_mlp:])
    c.Destroy()


def test_hashgrid_default_size_trains(api, torch_gpu):
    """the reference's actual default: 2^19 entries per hashed level = 14.2 M table parameters (57 MB fp32)"""
    c = api.NeuralRadianceCache(api.AppConfig(pos_id=0))
    assert c.ParamCount() == 64 * 48 + 5 * 64 * 64 + 3 * 64 + 2 * 7114752
    rng = np.random.default_rng(3)
    x = rng.random((4096, 5), dtype=np.float32)
    t = np.stack([np.sin(x[:, 0] * 9) * 0.5 + 0.5, x[:, 1], np.full(4096, 0.3, np.float32)], axis=1).astype(np.float32)
    d_x, d_t = torch_gpu.from_numpy(x).cuda(), torch_gpu.from_numpy(t).cuda()
    losses = []
    for _ in range(60):
        c.Backward(d_x, d_t)
        c.OptimizerStep()
        losses.append(c.GetLoss())
    assert np.isfinite(losses).all() and losses[-1] < 0.3 * losses[0]
    c.Destroy()


def test_fp16_exchange_rounds_once_before_and_once_after_the_sum(api, torch_gpu):
    """nrc_cache_set_exchange_dtype(NRC_EXCHANGE_F16) on one device, a gradient hook playing the second rank (it doubles the vector -- a second
    rank with the same shard): the hook is handed round_f16(local fp32 sum x loss_scale), what reaches the optimizer is round_f16 of the hook's
    sum, and the step from that vector is the step an fp32 cache takes when it is handed the same vector -- bit for bit"""
    n = 1024
    x = torch_gpu.from_numpy(queries(n, seed=61, nan_frac=0.0)).cuda()
    t = torch_gpu.rand((n, 3), device="cuda", generator=torch_gpu.Generator(device="cuda").manual_seed(5))
    seen = {}

    def make(dtype, hook):
        c = api.NeuralRadianceCache(api.AppConfig(train_batch_count=1, log2_train_batch_size=10, log2_infer_batch_size=10))
        c.SetExchangeDtype(dtype)
        assert c.GetExchangeDtype() == dtype
        out = torch_gpu.zeros((n, 3), device="cuda")
        c.Init(n, x, out, x, t)
        c.SetGradHook(hook)
        c.InferAndTrain(None, True)
        torch_gpu.cuda.synchronize()
        w = c.GetParams(0).copy()
        g = c.GetParams(4).copy()
        c.Destroy()
        return w, g

    def plain(g, _loss):
        seen["g32"] = g.detach().cpu().numpy().copy()

    def second_rank(g, _loss):
        seen["g16_in"] = g.detach().cpu().numpy().copy()
        g.mul_(2.0)

    make("f32", plain)
    w16, g16 = make("f16", second_rank)
    n_p = seen["g32"].size - 2 if seen["g32"].size > 25792 else seen["g32"].size
    g32 = seen["g32"][:25792]
    with np.errstate(over="ignore"):
        want_in = g32.astype(np.float16).astype(np.float32)
        want_out = (want_in * 2.0).astype(np.float16).astype(np.float32)
    assert np.array_equal(seen["g16_in"][:25792], want_in) and not np.array_equal(want_in, g32)
    bad = np.flatnonzero(g16[:25792] != want_out)
    assert bad.size == 0, "%d of 25792 differ; %d hold the undoubled value, %d the fp32 (unrounded) double; first %d: got %r want %r hook saw %r fp32 %r" % (
        bad.size, int((g16[bad] == want_in[bad]).sum()), int((g16[bad] == 2.0 * g32[bad]).sum()), bad[0], g16[bad[0]], want_out[bad[0]],
        want_in[bad[0]], g32[bad[0]])

    def handed(g, _loss):
        g[:25792].copy_(torch_gpu.from_numpy(want_out).cuda())

    w32, _ = make("f32", handed)
    assert np.array_equal(w16, w32)
