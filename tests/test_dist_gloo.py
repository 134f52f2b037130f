"""N > 1 path on CPU (gloo, world_size 2): the training exchange step.  Each rank back-propagates its shard of a global
batch against the GLOBAL loss normaliser; the all-reduced (summed) gradient must equal the single-process gradient of the
whole batch, and identical optimizer steps keep the replicas identical.  The oracle plays the role of the per-rank
arithmetic (tests may use it as the checker); the product's hook does the same all-reduce on device tensors."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nrc_hpm_renderer_amd import parallel
    from oracle import Oracle
    orc = Oracle()
    nn = orc.nn_create()
    rng = np.random.default_rng(11)
    n = 256
    x = rng.random((n, 5), dtype=np.float32)
    x[:, :3] += 31.0
    t = rng.random((n, 3), dtype=np.float32)
    sl = parallel.shard_train_batch(n, rank, world)
    loss_local = nn.backward(x[sl], t[sl], n_norm=n)
    both = torch.from_numpy(np.concatenate([np.array(nn.buffer(4)), [loss_local, 0.0]]).astype(np.float32))
    dist.all_reduce(both, op=dist.ReduceOp.SUM)          # what parallel.attach_gradient_allreduce's hook does
    nn.buffer(4)[:] = both[:-2].numpy()
    nn.optimizer_step()
    w = np.array(nn.buffer(0))
    gathered = [torch.zeros(w.size) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(w))
    if rank == 0:
        full = orc.nn_create()
        loss_full = full.backward(x, t)
        g_full = np.array(full.buffer(4))
        full.optimizer_step()
        q.put(dict(g_err=float(np.linalg.norm(both[:-2].numpy() - g_full) / np.linalg.norm(g_full)),
                   loss=float(both[-2]), loss_full=loss_full,
                   replicas_equal=bool(all(torch.equal(gathered[0], g) for g in gathered)),
                   w_err=float(np.linalg.norm(w - np.array(full.buffer(0))) / np.linalg.norm(w))))
    dist.destroy_process_group()


def test_sharded_gradient_allreduce_equals_full_batch():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["g_err"] < 1e-5
    assert abs(res["loss"] - res["loss_full"]) < 1e-5 * max(1.0, abs(res["loss_full"]))
    assert res["replicas_equal"]
    assert res["w_err"] < 1e-5


def _list_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nrc_hpm_renderer_amd import parallel
    n_entries, n_batch = 1 << 16, 64
    cap = parallel.grid_list_capacity(n_batch, n_entries)
    rng = np.random.default_rng(100 + rank)
    # a rank's packed fp16x2 table gradient: a few thousand touched entries, half of them shared with the other rank
    shared = np.random.default_rng(7).choice(n_entries, 3000, replace=False)
    own = rng.choice(n_entries, 3000, replace=False)
    g16 = np.zeros((n_entries, 2), np.float16)
    for idx in (shared, own):
        g16[idx] = (rng.standard_normal((idx.size, 2)) * 3.0).astype(np.float16)
    words = g16.view(np.uint32).reshape(-1)
    mine = parallel.pack_grid_list(words, cap)
    gathered = [torch.zeros(mine.size, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(mine.view(np.int32)))          # the padded all-gather of the product
    lists = [t.numpy().view(np.uint32) for t in gathered]
    total = parallel.apply_grid_lists(lists, n_entries)
    dense = torch.from_numpy(g16.astype(np.float32).reshape(-1).copy())
    dist.all_reduce(dense, op=dist.ReduceOp.SUM)                               # what the dense exchange computes
    everyone = [torch.zeros(total.size) for _ in range(world)]
    dist.all_gather(everyone, torch.from_numpy(total))
    if rank == 0:
        q.put(dict(count=int(mine[0]), cap=cap, padding_ok=bool((mine[2 + 2 * int(mine[0])::2] == parallel.GRID_LIST_PADDING).all()),
                   equals_dense=bool(np.array_equal(total, dense.numpy())),
                   replicas_equal=bool(all(torch.equal(everyone[0], e) for e in everyone)),
                   touched=int(np.count_nonzero(total.reshape(-1, 2).any(axis=1)))))
    dist.destroy_process_group()


def test_grid_gradient_list_exchange_equals_dense_allreduce():
    """HashGrid table gradient as all-gathered (entry, value) lists added in rank order: for two ranks the result equals the
    dense sum to the last bit (a + b in either order), and every replica holds the same vector"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_list_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert 5000 < res["count"] <= 6000 and res["cap"] == 64 * 128 and res["padding_ok"]
    assert res["equals_dense"] and res["replicas_equal"]
    assert 8000 < res["touched"] <= 9000


def _half_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nrc_hpm_renderer_amd import parallel
    from oracle import Oracle
    orc = Oracle()
    nn = orc.nn_create()
    rng = np.random.default_rng(11)
    n = 256
    x = rng.random((n, 5), dtype=np.float32)
    x[:, :3] += 31.0
    t = rng.random((n, 3), dtype=np.float32)
    sl = parallel.shard_train_batch(n, rank, world)
    loss_local = nn.backward(x[sl], t[sl], n_norm=n)
    g_local = np.array(nn.buffer(4)) * parallel.LOSS_SCALE                    # the device vector carries loss_scale
    sent = parallel.half_exchange_send(g_local)
    # the transport: every rank's fp16 vector to every rank (what a ring all-reduce amounts to for two ranks), summed in rank order
    gathered = [torch.zeros(sent.size * 2, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(sent.view(np.uint8)))
    total = parallel.half_exchange_sum([g.numpy().view(np.float16) for g in gathered])
    loss = torch.tensor([loss_local, 0.0])
    dist.all_reduce(loss, op=dist.ReduceOp.SUM)                                # the loss cell stays fp32
    nn.buffer(4)[:] = total / parallel.LOSS_SCALE
    nn.optimizer_step()
    w = np.array(nn.buffer(0))
    everyone = [torch.zeros(w.size) for _ in range(world)]
    dist.all_gather(everyone, torch.from_numpy(w))
    if rank == 0:
        full = orc.nn_create()
        loss_full = full.backward(x, t)
        g_full = np.array(full.buffer(4)) * parallel.LOSS_SCALE
        full.optimizer_step()
        halves = [g.numpy().view(np.float16).astype(np.float64) for g in gathered]
        q.put(dict(g_err=float(np.linalg.norm(total - g_full) / np.linalg.norm(g_full)),
                   representable=bool(np.array_equal(total, total.astype(np.float16).astype(np.float32))),
                   one_rounding=bool(np.array_equal(total, (halves[0] + halves[1]).astype(np.float16).astype(np.float32))),
                   finite=bool(np.isfinite(total).all()), max_abs=float(np.abs(total).max()),
                   loss=float(loss[0]), loss_full=loss_full,
                   replicas_equal=bool(all(torch.equal(everyone[0], e) for e in everyone)),
                   w_err=float(np.linalg.norm(w - np.array(full.buffer(0))) / np.linalg.norm(w))))
    dist.destroy_process_group()


def test_fp16_gradient_exchange_keeps_replicas_identical():
    """nrc_cache_set_exchange_dtype(NRC_EXCHANGE_F16), the host statement of the protocol (parallel.half_exchange_*): gradients pre-scaled by
    loss_scale 128, rounded to fp16 once per rank, summed in fp16 -- within fp16's resolution of the full-batch gradient (2^-11 per
    addend), every replica bit-identical after the optimizer step, nothing near fp16's range limits"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_half_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["finite"] and res["max_abs"] < 6.0e4 and res["representable"] and res["one_rounding"]
    assert res["g_err"] < 1e-3                       # 2^-11 relative per number, three roundings
    assert abs(res["loss"] - res["loss_full"]) < 1e-5 * max(1.0, abs(res["loss_full"]))
    assert res["replicas_equal"]
    assert res["w_err"] < 2e-3                       # Adam turns a rounded gradient into a step of the same size: a few weights move the other way
