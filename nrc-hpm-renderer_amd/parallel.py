"""Multi-GPU sharding of the NRC-HPM path (new; the reference is single-GPU -- SURVEY.md section 8e).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Pixels are independent (the RNG seed depends
only on the global pixel coordinate and the per-frame random, data/shader/include/random.glsl:61-64), so a frame shards
by pixel columns with NO data-path collective: strips of 8 adjacent columns are dealt to the ranks round-robin (rank r renders
strips r, r+N, r+2N, ...: every rank sees the same mix of cloud and empty sky, and a wavefront's 8x8-pixel tile stays contiguous).  Training has one exchange step: every rank back-propagates its
own train rays against the GLOBAL loss normaliser, the gradient vectors (25 792 fp32 + the loss cell = 103 KB:
latency-bound, one small all-reduce) are summed, and every rank applies the identical optimizer step, so the weight
replicas stay bit-identical.  The train ring buffer is per rank.
"""


# width of the column strips the ranks interleave (nrc_tile::x_block): 8 keeps the 8x8-pixel tile a wavefront renders contiguous on
# the screen -- single interleaved columns spread it over 8 * world global columns (less coherent walks, more distinct cache lines
# per gather); measured with tools/c4_rank_emulation.py, DESIGN.md section 5.  NRC_TILE_BLOCK overrides.
import os as _os

DEFAULT_BLOCK = int(_os.environ.get("NRC_TILE_BLOCK", "8"))


def _columns(rank, world, global_w, block):
    """global columns of `rank`, in local order: strips of `block` columns, strip s belongs to rank s % world"""
    cols = []
    s = rank
    while s * block < global_w:
        cols.extend(range(s * block, min((s + 1) * block, global_w)))
        s += world
    return cols


def local_width(rank, world, global_w, block=None):
    """number of columns of the global frame that rank renders (strips of `block` columns dealt round-robin)"""
    block = DEFAULT_BLOCK if block is None else block
    if world == 1:
        return global_w
    full, rest = divmod(global_w, block * world)
    return full * block + min(max(rest - rank * block, 0), block)


def column_tile(rank, world, global_w, global_h, block=None):
    """nrc_tile of include/nrc_hpm.h: (x_offset, x_stride, global_w, global_h, x_block)"""
    block = DEFAULT_BLOCK if block is None else block
    if world == 1:
        return (0, 1, global_w, global_h, 1)
    return (rank, world, global_w, global_h, block)


def rank_columns(rank, world, global_w, block=None):
    """the global column of every local column of `rank` (numpy index array)"""
    import numpy as np
    block = DEFAULT_BLOCK if block is None else block
    return np.asarray(_columns(rank, world, global_w, block) if world > 1 else list(range(global_w)), np.int64)


def gather_columns(local_images, global_w, block=None):
    """inverse of the sharding, for tests: list of [h, local_w, c] arrays (rank order) -> [h, global_w, c]"""
    import numpy as np
    world = len(local_images)
    h, _, c = local_images[0].shape
    out = np.zeros((h, global_w, c), local_images[0].dtype)
    for r, img in enumerate(local_images):
        cols = rank_columns(r, world, global_w, block)
        assert img.shape[1] == len(cols)
        out[:, cols, :] = img
    return out


def shard_train_batch(n_global, rank, world):
    """contiguous slice of a global train batch owned by `rank` (tests of the gradient all-reduce)"""
    per = n_global // world
    return slice(rank * per, (rank + 1) * per)


# ---- the list exchange of a HashGrid model's table gradient (nrc_mlp.hip: k_grid_pack / k_grid_apply), host restatement.
# The product runs it on the device inside the library (ncclAllGather); these helpers state the protocol for the gloo tests
# and check the device kernels (tests/test_gpu_mlp.py).
GRID_LIST_PADDING = 0xFFFFFFFF


def grid_list_capacity(n_batch, n_entries, levels=16):
    """list slots a rank needs: one per (sample, level, corner) at most, never more than the table"""
    return min(n_batch * levels * 8, n_entries)


def pack_grid_list(grad16_words, cap):
    """uint32 words {count, 0, (entry, half2 bits) x cap}: the entries whose packed fp16 gradient word is non-zero"""
    import numpy as np
    w = np.asarray(grad16_words, np.uint32)
    ent = np.flatnonzero(w).astype(np.uint32)
    if ent.size > cap:
        raise ValueError("the batch touched %d entries, capacity %d" % (ent.size, cap))
    out = np.full(2 + 2 * cap, GRID_LIST_PADDING, np.uint32)
    out[0], out[1] = ent.size, 0
    out[2:2 + 2 * ent.size:2] = ent
    out[3:3 + 2 * ent.size:2] = w[ent]
    return out


def apply_grid_lists(lists, n_entries):
    """fp32 table gradient [n_entries * 2] := the lists' values added in list (= rank) order, starting from zero"""
    import numpy as np
    g = np.zeros((n_entries, 2), np.float32)
    for lst in lists:
        lst = np.asarray(lst, np.uint32)
        count = int(lst[0])
        ent, val = lst[2:2 + 2 * count:2], lst[3:3 + 2 * count:2]
        g[ent] = g[ent] + val.copy().view(np.float16).reshape(-1, 2).astype(np.float32)      # an entry occurs once per list
    return g.reshape(-1)


# ---- the fp16 gradient exchange (nrc_cache_set_exchange_dtype, BASELINE.json configs[3]), host restatement for the gloo tests:
# the fp32 gradient vector (a local sum, already times loss_scale 128) is rounded to fp16 once, the ranks' vectors are summed in fp16
# (every partial sum rounded), and the sum is widened back to fp32.  For two ranks the result does not depend on the order.
LOSS_SCALE = 128.0


def half_exchange_send(grad_f32):
    """what a rank puts on the wire: the gradient vector as fp16 numbers"""
    import numpy as np
    with np.errstate(over="ignore"):
        return np.asarray(grad_f32, np.float32).astype(np.float16)


def half_exchange_sum(sent):
    """ncclAllReduce(ncclHalf, sum) of the ranks' vectors in rank order, widened to the fp32 vector the optimizer reads"""
    import numpy as np
    acc = np.asarray(sent[0], np.float16).copy()
    with np.errstate(over="ignore"):
        for v in sent[1:]:
            acc = (acc + np.asarray(v, np.float16)).astype(np.float16)
    return acc.astype(np.float32)


def attach_gradient_allreduce(nrc, world, group=None, native=True, dtype="f32"):
    """Installs the exchange step of the training path on a NeuralRadianceCache: the loss normaliser becomes the global
    batch (3 * trainBatchSize * world) and the fp32 gradient vector + loss cell are all-reduced (sum) between backward and
    the optimizer of every train batch, on the stream the training kernels run on.

    native=True (default): the library calls ncclAllReduce itself (RCCL communicator created from a unique id that rank 0
    broadcasts through torch.distributed) -- no Python in the per-frame path.
    native=False: a Python hook calls torch.distributed.all_reduce (used by the CPU/gloo-style tests of the logic).
    dtype "f16": the gradients travel as fp16 numbers (nrc_cache_set_exchange_dtype; the default fp32 vector is 103 KB, this one 52 KB)."""
    import torch
    import torch.distributed as dist
    from . import api
    if dtype != "f32":
        nrc.SetExchangeDtype(dtype)      # (the hook below is then handed fp16-rounded values and its sum is rounded again)
    if native:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8))
        if dist.is_initialized() and world > 1:
            dist.broadcast(idt, 0, group=group)
        nrc.CommInit(bytes(idt.cpu().numpy().tobytes()), rank, world)
        return None
    n = nrc.ParamCount()
    both = api._wrap_device(nrc.L.nrc_cache_grad_ptr(nrc.h), (n + 2) * 4, torch.float32, (n + 2,))
    nrc.SetLossNormFactor(world)

    def hook(_grad, _loss):
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)

    nrc.SetGradHook(hook)
    return both
