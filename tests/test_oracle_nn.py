"""The oracle's NN arithmetic (tiny-cuda-nn v1.6 semantics; PARITY UNPINNED against the reference --
the submodule is absent) checked for internal consistency against a plain PyTorch fp32 statement of the same maths."""
import numpy as np
import pytest
import torch


def queries(n, seed=0, nan_frac=0.1):
    rng = np.random.default_rng(seed)
    x = rng.random((n, 5), dtype=np.float32)
    x[:, :3] += 31.0                                   # quirk Q3: positions land around size/2
    x[:, 3] = x[:, 3] * 2.0 - 0.5                      # quirk Q4: theta in [-0.5, 1.5]
    x[rng.random(n) < nan_frac, 4] = np.nan            # quirk Q5: NaN phi
    return x


def test_pcg32_known_answers(orc):
    """the generator against the vector its authors publish (pcg-c-basic, pcg32-demo: seed 42, stream 54) and tiny-cuda-nn's
    next_float() = bits((u >> 9) | 0x3f800000) - 1"""
    u, f = orc.pcg32(42, 54, 6)
    assert [int(x) for x in u] == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    want = ((u >> 9) | np.uint32(0x3f800000)).view(np.float32) - np.float32(1.0)
    assert np.array_equal(f, want) and (f >= 0).all() and (f < 1).all()


def tcnn_init(width=64, depth=6, enc=80, seed=1337, n_grid=0, orc=None):
    """tiny-cuda-nn v1.6's initialisation written out once more, from the generator's raw output (recalled
    from upstream): pcg32{seed}, stream 1; matrices in order, output matrix 16 x width; x = u * 2 * scale - scale in fp32; then
    the table in generate_random_uniform's GPU order"""
    shapes = [(width, enc)] + [(width, width)] * (depth - 1) + [(16, width)]
    n = sum(r * c for r, c in shapes)
    n_threads = ((n_grid + 3) // 4 + 127) // 128 * 128 if n_grid else 0
    _, f = orc.pcg32(seed, 1, n + 4 * n_threads)
    out, k = [], 0
    for r, c in shapes:
        scale = np.float32(1.0) * np.sqrt(np.float32(6.0) / np.float32(r + c), dtype=np.float32)
        u = f[k:k + r * c]
        k += r * c
        out.append(u * np.float32(2.0) * scale - scale)
    grid = None
    if n_grid:
        u = f[k:k + 4 * n_threads].astype(np.float64)
        # fma(u, 2e-4f, -1e-4f): the product of two floats is exact in double, one rounding to float
        hi = np.float64(np.float32(1e-4) - np.float32(-1e-4))
        v = (u * hi + np.float64(np.float32(-1e-4))).astype(np.float32)
        kk = np.arange(4 * n_threads)
        idx = kk // 4 + n_threads * (kk % 4)
        grid = np.zeros(n_grid, np.float32)
        keep = idx < n_grid
        grid[idx[keep]] = v[keep]
    return out, grid


def test_param_count_and_init(orc):
    nn = orc.nn_create()
    assert nn.enc_dims == 80
    assert nn.n_params == 64 * 80 + 5 * 64 * 64 + 3 * 64
    w = nn.buffer(0)
    s0 = np.sqrt(6.0 / (80 + 64))
    assert np.abs(w[:64 * 80]).max() <= s0 + 1e-6 and np.abs(w[:64 * 80]).max() > 0.9 * s0
    assert np.array_equal(w, nn.buffer(1))             # EMA starts at the weights
    assert np.array_equal(orc.nn_create().buffer(0), w)    # seed 1337 is deterministic


@pytest.mark.parametrize("width,depth", [(64, 6), (128, 8), (16, 2)])
def test_init_follows_tiny_cuda_nn(orc, width, depth):
    """every matrix draws rows x columns numbers of ITS STORED SHAPE from pcg32{1337} (stream 1) under the Xavier bound of that
    shape; the output matrix is stored 16 x width: 16 * width draws, bound sqrt(6 / (16 + width)), rows 3..15 dead"""
    nn = orc.nn_create(width=width, depth=depth)
    mats, _ = tcnn_init(width, depth, 80, orc=orc)
    flat = np.concatenate(mats)
    assert flat.size == nn.n_params + 13 * width == (26624 if (width, depth) == (64, 6) else flat.size)
    t = nn.tcnn_params(0, width)
    assert np.array_equal(t, flat)                      # bit for bit, dead rows included
    own = nn.buffer(0)
    assert np.array_equal(own[:nn.n_mlp - 3 * width], flat[:nn.n_mlp - 3 * width])
    assert np.array_equal(own[nn.n_mlp - 3 * width:], mats[-1][:3 * width])
    bound = np.sqrt(6.0 / (16 + width))
    assert np.abs(mats[-1]).max() <= bound and np.abs(mats[-1]).max() > 0.95 * bound


def test_hashgrid_init_follows_tiny_cuda_nn(orc):
    nn = orc.nn_create(pos_id=0, hashgrid_log2_size=12)
    n_grid = nn.n_params - nn.n_mlp
    enc = nn.enc_dims
    mats, grid = tcnn_init(64, 6, enc, n_grid=n_grid, orc=orc)
    own = nn.buffer(0)
    assert np.array_equal(own[nn.n_mlp:], grid)
    assert np.abs(grid).max() <= 1e-4 and np.abs(grid).max() > 0.99e-4 and abs(grid.mean()) < 5e-6
    assert np.array_equal(own[:64 * enc], mats[0])


def test_frequency_encoding(orc):
    nn = orc.nn_create()
    x = np.zeros((3, 5), np.float32)
    x[0, :3] = [0.25, 0.5, 31.3]
    x[1, :3] = [0.25 + 2.0, 0.5 + 4.0, 31.3]           # period 2 at f = 0
    e = nn.encode(x)
    # out[d*24 + 2f + s] = sin(2^f*pi*x_d + s*pi/2)
    assert abs(e[0, 0] - np.sin(np.pi * 0.25)) < 1e-3 and abs(e[0, 1] - np.cos(np.pi * 0.25)) < 1e-3
    assert abs(e[0, 24 + 2] - np.sin(2 * np.pi * 0.5)) < 1e-3
    assert np.allclose(e[0, :72], e[1, :72], atol=1e-3)
    t = np.float32(31.3) * np.float32(2.0 ** 11)
    r = float(t - 2.0 * np.floor(t / 2.0))
    assert abs(e[0, 48 + 22] - np.sin(np.pi * r)) < 1e-3          # exact argument reduction defines the value (Q3)


def test_oneblob_encoding(orc):
    nn = orc.nn_create()
    x = np.zeros((5, 5), np.float32)
    x[:, 3] = [0.0, 0.125, 0.5, 0.99, np.nan]
    x[:, 4] = [0.3, 0.3, 0.3, 0.3, 0.3]
    e = nn.encode(x)[:, 72:]
    assert np.allclose(e[:4, :4].sum(axis=1), 1.0, atol=2e-3)      # the periodic quartic kernel integrates to 1
    assert np.allclose(e[:, 4:].sum(axis=1), 1.0, atol=2e-3)
    assert e[1, 0] == e[1, :4].max()                               # x = 0.125 is the centre of bin 0
    assert np.array_equal(e[4, :4], np.array([0, 0, 0, 1], np.float32))     # NaN -> (0,0,0,1) (fminf/fmaxf swallow NaN)
    # periodic: x = 0 spreads equally into bins 0 and 3
    assert abs(e[0, 0] - e[0, 3]) < 1e-3 and e[0, 1] < 1e-3


def torch_mlp(nn, enc, use_ema, quant, width=64, depth=6):
    w = torch.from_numpy(np.array(nn.buffer(1 if use_ema else 0)))
    if quant:
        w = w.half().float()
    dims = [(width, 80)] + [(width, width)] * (depth - 1) + [(3, width)]
    mats, off = [], 0
    for o, i in dims:
        mats.append(w[off:off + o * i].view(o, i).clone().requires_grad_(True))
        off += o * i
    h = torch.from_numpy(enc)
    for m in mats[:-1]:
        h = torch.relu(h @ m.t())
        if quant:
            h = (h.half().float() - h).detach() + h        # straight-through fp16 rounding
    return h @ mats[-1].t(), mats


def test_forward_matches_torch(orc):
    nn = orc.nn_create()
    x = queries(512)
    enc = nn.encode(x)
    y32 = nn.forward(x, use_ema=True, mode=0)
    ref32, _ = torch_mlp(nn, enc, True, False)
    assert np.allclose(y32, ref32.detach().numpy(), atol=2e-5, rtol=1e-4)
    y16 = nn.forward(x, use_ema=True, mode=1)
    ref16, _ = torch_mlp(nn, enc, True, True)
    assert np.linalg.norm(y16 - ref16.detach().numpy()) / np.linalg.norm(y16) < 2e-3
    assert np.linalg.norm(y16 - y32) / np.linalg.norm(y32) < 1e-2      # fp16 storage stays close to fp32


@pytest.mark.parametrize("loss_id,width", [(0, 64), (1, 64), (2, 64), (3, 64), (4, 64), (5, 64), (6, 64), (0, 16)],
                         ids=["RelativeL2Luminance", "L2", "RelativeL2", "L1", "Mape", "Smape", "LogL1", "width16"])
def test_backward_matches_autograd(orc, loss_id, width):
    nn = orc.nn_create(loss_id=loss_id, width=width)
    n = 256
    x = queries(n, seed=1)
    rng = np.random.default_rng(2)
    t = rng.random((n, 3), dtype=np.float32) * 2.0
    loss = nn.backward(x, t)
    y, mats = torch_mlp(nn, nn.encode(x), False, True, width=width)
    tt = torch.from_numpy(t)
    N = 3 * n
    if loss_id == 0:
        lum = 0.299 * y[:, 0] + 0.587 * y[:, 1] + 0.114 * y[:, 2]
        den = (lum * lum + 0.01).detach()[:, None]       # tiny-cuda-nn treats the normaliser as a constant
    elif loss_id == 1:
        den = torch.ones_like(y)
    elif loss_id == 2:
        den = (y * y + 0.01).detach()
    if loss_id <= 2:
        lt = ((y - tt) ** 2 / den / N).sum()
    elif loss_id == 3:                                    # L1
        lt = ((y - tt).abs() / N).sum()
    elif loss_id == 4:                                    # Mape
        lt = ((y - tt).abs() / (tt.abs() + 0.01) / N).sum()
    elif loss_id == 5:                                    # Smape: the scale is a constant in tiny-cuda-nn's gradient
        lt = ((y - tt).abs() / (0.5 * (y.abs() + tt.abs()) + 0.01).detach() / N).sum()
    else:                                                 # LogL1
        lt = (torch.log(1.0 + (y - tt).abs()) / N).sum()
    lt.backward()
    g_ref = torch.cat([m.grad.reshape(-1) for m in mats]).numpy()
    g = np.array(nn.buffer(4))
    assert abs(loss - float(lt.detach())) < 1e-4 * max(1.0, abs(float(lt.detach())))
    assert np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref) < 2e-2      # fp16 deltas (loss_scale 128) vs fp32 autograd


def test_optimizer_matches_torch_adam(orc):
    """EMA{Adam}: against torch.optim.Adam where eps is negligible (tiny-cuda-nn folds the bias correction into the
    learning rate without rescaling eps, so the two differ by O(eps/sqrt(v)) per update), and against the SURVEY App. B
    formulas evaluated in float64 everywhere."""
    nn = orc.nn_create(lr=0.01, ema_decay=0.99)
    w0 = np.array(nn.buffer(0))
    p = torch.from_numpy(w0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8)
    rng = np.random.default_rng(3)
    big = np.ones(nn.n_params, bool)
    w64, m64, v64, ema64 = w0.astype(np.float64), 0.0, 0.0, w0.astype(np.float64)
    for step in range(1, 4):
        g = (rng.standard_normal(nn.n_params) * 1e-2).astype(np.float32)
        big &= np.abs(g) > 5e-3
        nn.buffer(4)[:] = g
        nn.optimizer_step()
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        gg = g.astype(np.float64) + 1e-8 * w64
        m64 = 0.9 * m64 + 0.1 * gg
        v64 = 0.999 * v64 + 0.001 * gg * gg
        lr_t = 0.01 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        w64 = w64 - lr_t * m64 / (np.sqrt(v64) + 1e-8)
        d = 0.99
        ema64 = (ema64 * d * (1 - d ** (step - 1)) + w64 * (1 - d)) / (1 - d ** step)
        assert np.allclose(nn.buffer(0), w64, atol=1e-6, rtol=0)
        assert np.allclose(nn.buffer(1), ema64, atol=1e-6, rtol=0)
    assert big.sum() > 1000
    assert np.allclose(nn.buffer(0)[big], p.detach().numpy()[big], atol=3e-6, rtol=0)
    assert not np.array_equal(nn.buffer(0), w0)


def test_optimizer_sgd_matches_torch(orc):
    """EMA{SGD} (tiny-cuda-nn sgd.h: l2_reg 1e-8, no momentum) against torch.optim.SGD and the float64 EMA recursion"""
    nn = orc.nn_create(lr=0.05, ema_decay=0.99, optimizer="SGD")
    w0 = np.array(nn.buffer(0))
    p = torch.from_numpy(w0.copy()).requires_grad_(True)
    opt = torch.optim.SGD([p], lr=0.05, weight_decay=1e-8)
    rng = np.random.default_rng(4)
    ema64 = w0.astype(np.float64)
    for step in range(1, 4):
        g = (rng.standard_normal(nn.n_params) * 1e-2).astype(np.float32)
        nn.buffer(4)[:] = g
        nn.optimizer_step()
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        d = 0.99
        ema64 = (ema64 * d * (1 - d ** (step - 1)) + np.array(nn.buffer(0), np.float64) * (1 - d)) / (1 - d ** step)
        assert np.allclose(nn.buffer(0), p.detach().numpy(), atol=1e-7, rtol=0)
        assert np.allclose(nn.buffer(1), ema64, atol=1e-6, rtol=0)
    assert not np.array_equal(nn.buffer(0), w0)


def test_training_reduces_loss(orc):
    nn = orc.nn_create(lr=0.01)
    x = queries(256, seed=5, nan_frac=0.0)
    t = np.stack([np.sin(x[:, 0] * 7) * 0.5 + 0.5, x[:, 3] * 0.3 + 0.2, np.full(256, 0.4, np.float32)], axis=1).astype(np.float32)
    first = last = None
    for _ in range(30):
        last = nn.backward(x, t)
        first = last if first is None else first
        nn.optimizer_step()
    assert last < 0.5 * first


def numpy_hashgrid(x3, table, offsets):
    """independent statement of tiny-cuda-nn's multiresolution hash encoding (16 levels x 2 features, base 16, scale 2)"""
    n = x3.shape[0]
    out = np.zeros((n, 32), np.float64)
    for l in range(16):
        res = 16 << l
        scale = np.float32(res - 1)
        hsize = offsets[l + 1] - offsets[l]
        pos = np.float32(scale) * x3.astype(np.float32) + np.float32(0.5)      # fma differs from mul+add by <= 1 ulp: fine here
        pg = np.floor(pos)
        fr = (pos - pg).astype(np.float64)
        pg = pg.astype(np.int64)
        for c in range(8):
            w = np.ones(n)
            pl = []
            for d in range(3):
                if c & (1 << d):
                    w *= fr[:, d]
                    pl.append(pg[:, d] + 1)
                else:
                    w *= 1 - fr[:, d]
                    pl.append(pg[:, d])
            pl = [p.astype(np.uint64) & 0xFFFFFFFF for p in pl]
            if res ** 3 <= hsize:
                idx = (pl[0] + pl[1] * res + pl[2] * res * res) & 0xFFFFFFFF
            else:
                idx = ((pl[0] * 1) ^ ((pl[1] * 2654435761) & 0xFFFFFFFF) ^ ((pl[2] * 805459861) & 0xFFFFFFFF)) & 0xFFFFFFFF
            idx = (idx % hsize).astype(np.int64) + offsets[l]
            for f in range(2):
                out[:, 2 * l + f] += w * table[idx * 2 + f]
    return out


def test_hashgrid_encoding_and_layout(orc):
    """AppConfig posID 0 (src/AppConfig.cpp:19-27): table sizes of SURVEY section 5 and the encoding itself"""
    full = orc.nn_create(pos_id=0)
    assert (full.n_params - full.n_mlp) == 2 * 7114752                     # 16^3 + 32^3 + 64^3 + 13 * 2^19 entries
    assert full.enc_dims == 48 and full.n_mlp == 64 * 48 + 5 * 64 * 64 + 3 * 64
    nn = orc.nn_create(pos_id=0, hashgrid_log2_size=12)
    offsets = [0]
    for l in range(16):
        offsets.append(offsets[-1] + min(((16 << l) ** 3 + 7) // 8 * 8, 1 << 12))
    assert (nn.n_params - nn.n_mlp) == 2 * offsets[-1]
    table = np.array(nn.buffer(1)[nn.n_mlp:])
    assert np.abs(table).max() <= 1e-4 and np.abs(table).max() > 0.9e-4      # uniform [-1e-4, 1e-4)
    rng = np.random.default_rng(0)
    nn.buffer(1)[nn.n_mlp:] = rng.standard_normal(table.size).astype(np.float32)    # O(1) features for a meaningful check
    table = np.array(nn.buffer(1)[nn.n_mlp:]).astype(np.float16).astype(np.float64)  # the oracle gathers fp16 copies
    x = rng.random((64, 5), dtype=np.float32)
    x[32:, :3] += 31.0                                                       # quirk Q3 range: indices wrap by modulo / hash
    e = nn.encode(x)
    ref = numpy_hashgrid(x[:, :3], table, offsets)
    # low levels only for the large coordinates: at level l >= 8 an fma-vs-mul+add ulp moves the fractional part visibly
    assert np.allclose(e[:32, :32], ref[:32], atol=2e-3, rtol=2e-3)
    assert np.allclose(e[32:, :8], ref[32:, :8], atol=5e-2, rtol=5e-2)
    assert np.allclose(e[:, 40:], 1.0)                                       # 32 + 8 -> padded to 48 with ones


def test_hashgrid_backward_and_sparse_adam(orc):
    nn = orc.nn_create(pos_id=0, hashgrid_log2_size=10)
    rng = np.random.default_rng(1)
    x = rng.random((64, 5), dtype=np.float32)
    t = rng.random((64, 3), dtype=np.float32)
    w0 = np.array(nn.buffer(0))
    nn.backward(x, t)
    g = np.array(nn.buffer(4))
    gg = g[nn.n_mlp:]
    assert 0 < np.count_nonzero(gg) < gg.size            # sparse: only touched corners
    # finite-difference check of one touched grid entry (loss is smooth in the table)
    k = int(np.argmax(np.abs(gg)))
    nn.optimizer_step()
    w1 = np.array(nn.buffer(0))
    touched = gg != 0
    assert np.array_equal(w1[nn.n_mlp:][~touched], w0[nn.n_mlp:][~touched])      # zero-gradient entries untouched
    assert (w1[nn.n_mlp:][touched] != w0[nn.n_mlp:][touched]).mean() > 0.99
    assert abs(abs(w1[nn.n_mlp + k] - w0[nn.n_mlp + k]) - 0.01) < 1e-3              # first Adam step = lr * sign(g)
    assert np.array_equal(np.array(nn.buffer(2))[nn.n_mlp:][~touched], np.zeros((~touched).sum(), np.float32))


def oneblob4_two_edges(x):
    """The OneBlob(4) bins the HIP kernels compute (nrc_mlp.hip, oneblob4_bins): a quartic kernel of radius 1/4 around x overlaps only
    the two bin edges next to it, so two CDF values A = cdf(left edge), B = cdf(right edge) give all four bins -- the bin left of x's
    gets A, x's own B - A, the next 1 - B, the fourth 0 (indices mod 4: the encoding is periodic)."""
    x = np.asarray(x, np.float32)
    t = (x * np.float32(4.0)).astype(np.float32)
    j = np.floor(t).astype(np.float32)
    fr = (t - j).astype(np.float32)

    def q(u):
        u = u.astype(np.float32)
        u2 = (u * u).astype(np.float32)
        u4 = (u2 * u2).astype(np.float32)
        p = (np.float32(15.0 / 16.0) * u * ((np.float32(1.0) - np.float32(2.0 / 3.0) * u2) + np.float32(1.0 / 5.0) * u4) + np.float32(0.5)).astype(np.float32)
        return np.clip(p, 0.0, 1.0).astype(np.float32)

    a, b = q(-fr), q(np.float32(1.0) - fr)
    ji = j.astype(np.int64) & 3
    out = np.zeros(x.shape + (4,), np.float32)
    idx = np.arange(x.size)
    flat = out.reshape(-1, 4)
    flat[idx, (ji.reshape(-1) + 3) & 3] = a.reshape(-1)
    flat[idx, ji.reshape(-1)] = (b - a).reshape(-1)
    flat[idx, (ji.reshape(-1) + 1) & 3] = (np.float32(1.0) - b).reshape(-1)
    return out


def test_oneblob_two_edge_formula_equals_the_three_image_sum(orc):
    """tiny-cuda-nn's OneBlob sums the kernel's CDF over three periodic images per bin edge (the oracle's statement); over the whole
    range the renderer's direction coordinates take, theta / pi + 0.5 in (-0.5, 1.5], that equals the two-edge form to fp32 rounding"""
    nn = orc.nn_create()
    xs = np.concatenate([np.linspace(-0.5, 1.5, 20001, dtype=np.float32)[1:], np.float32([0.0, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, -0.25, 0.999999, 1e-7])])
    q = np.zeros((xs.size, 5), np.float32)
    q[:, 3] = xs
    q[:, 4] = xs[::-1]
    e = nn.encode(q)[:, 72:]                       # fp16 features
    for got, ref in ((oneblob4_two_edges(xs), e[:, :4]), (oneblob4_two_edges(xs[::-1]), e[:, 4:])):
        got16 = got.astype(np.float16).astype(np.float32)
        off = got16 != ref
        assert off.mean() < 2e-2                                       # fp32 rounding differences (~1e-6) tip one fp16 rounding in a hundred
        assert np.abs(got16 - ref).max() <= 2.0 ** -11                 # ... by one fp16 ulp of a value below 1
        assert np.abs(got - ref).max() < 2.0 ** -11                    # and the fp32 values sit inside that rounding interval
