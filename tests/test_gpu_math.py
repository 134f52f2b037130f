"""Bit parity of the integrator's math spec and RNG between the HIP device code (csrc/nrc_math.h) and the oracle's own
statement (oracle/orc_math.h): this is what makes per-pixel control flow reproducible (DESIGN.md "math spec")."""
import numpy as np
import pytest

from conftest import FRAME_RANDOM

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _inputs(kind, n=200000, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "log":          # 1 - u, u in [0,1)
        u = rng.random(n, dtype=np.float32)
        return np.concatenate([np.float32(1.0) - u, np.array([1.0, 2.0 ** -23, 0.5, 0.70710677, 0.7071068], np.float32)])
    if kind == "angle":        # [0, 2pi] and a bit beyond
        return np.concatenate([rng.random(n, dtype=np.float32) * np.float32(6.2831855),
                               np.array([0.0, 3.1415927, 6.2831855, 1.5707964, 0.7853982, 100.5, -2.5], np.float32)])
    if kind == "unit":         # [-1.2, 1.2]: includes out-of-domain (NaN) arguments of quirk Q5
        return np.concatenate([rng.random(n, dtype=np.float32) * np.float32(2.4) - np.float32(1.2),
                               np.array([-1.0, 1.0, 0.0, 0.5, -0.5, 1.0000001, np.nan], np.float32)])
    if kind == "any":
        return (rng.standard_normal(n) * 10).astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("fn,kind", [(0, "log"), (1, "angle"), (2, "unit"), (3, "unit"), (5, "unit"), (7, "log"), (8, "any")])
def test_unary_math_bit_exact(orc, api, torch_gpu, fn, kind):
    a = _inputs(kind)
    ref, ref2 = orc.math_eval(fn, a)
    out, out2 = api.test_math(fn, torch_gpu.from_numpy(a).cuda())
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(ref))
    if fn == 1:
        assert np.array_equal(_bits(out2.cpu().numpy()), _bits(ref2))


@pytest.mark.parametrize("fn", [4, 6])
def test_binary_math_bit_exact(orc, api, torch_gpu, fn):
    a, b = _inputs("any", seed=1), _inputs("any", seed=2)
    a[:4] = [0.0, 1.0, -1.0, 0.0]
    b[:4] = [0.0, 0.0, 0.0, -2.0]
    if fn == 6:
        b[b == 0] = 1.0
    ref, _ = orc.math_eval(fn, a, b)
    out, _ = api.test_math(fn, torch_gpu.from_numpy(a).cuda(), torch_gpu.from_numpy(b).cuda())
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(ref))


def test_packed_log_matches_scalar_spec(orc, api, torch_gpu):
    """the tracking loops evaluate two free-flight logs per trip with packed fp32 math: same bits as the scalar spec"""
    a, b = _inputs("log", seed=3), _inputs("log", seed=4)[::-1].copy()
    out, out2 = api.test_math(9, torch_gpu.from_numpy(a).cuda(), torch_gpu.from_numpy(b).cuda())
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(orc.math_eval(0, a)[0]))
    assert np.array_equal(_bits(out2.cpu().numpy()), _bits(orc.math_eval(0, b)[0]))


def test_log_is_bit_exact_on_every_argument_the_integrator_can_pass(orc, api, torch_gpu):
    """the free-flight log is only ever taken of 1 - k * 2^-23: all 2^23 arguments, scalar form and both halves of the packed form, against
    the oracle's statement (the table in LDS, the frexp instructions and the packed FMAs against integer field extraction and fmaf)"""
    x = (np.arange(1, 2 ** 23 + 1, dtype=np.float64) * 2.0 ** -23).astype(np.float32)
    ref = _bits(orc.math_eval(0, x)[0])
    xs = torch_gpu.from_numpy(x).cuda()
    out, _ = api.test_math(0, xs)
    assert np.array_equal(_bits(out.cpu().numpy()), ref)
    rev = torch_gpu.flip(xs, dims=[0]).contiguous()
    out, out2 = api.test_math(9, xs, rev)
    assert np.array_equal(_bits(out.cpu().numpy()), ref)
    assert np.array_equal(_bits(out2.cpu().numpy()), ref[::-1])


def test_box_sdf_sqrt_is_correctly_rounded(orc, api, torch_gpu):
    """sky_sdf's lean sqrt (v_sqrt_f32 + one-ulp fix-up, no denormal path) == the oracle's sqrtf on zero and normal inputs"""
    rng = np.random.default_rng(5)
    a = np.concatenate([np.exp(rng.uniform(-60, 60, 400000)).astype(np.float32), rng.random(200000, dtype=np.float32) * 1e4,
                        np.array([0.0, 1.0, 4.0, 2.0, 1e-30, 3e38, 0.015625, 1.0000001, 0.99999994], np.float32)])
    out, _ = api.test_math(10, torch_gpu.from_numpy(a).cuda())
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(orc.math_eval(7, a)[0]))


def test_math_spec_accuracy(orc):
    """the spec itself is a faithful log/sin/cos/acos/atan2 (GLSL precision requirements are far looser)"""
    a = _inputs("log")
    assert np.abs(orc.math_eval(0, a)[0] - np.log(a.astype(np.float64))).max() < 2e-6
    ang = _inputs("angle")[:-2]
    s, c = orc.math_eval(1, ang)
    assert np.abs(s - np.sin(ang.astype(np.float64))).max() < 5e-7 and np.abs(c - np.cos(ang.astype(np.float64))).max() < 5e-7
    u = np.linspace(-1, 1, 10001, dtype=np.float32)
    assert np.abs(orc.math_eval(2, u)[0] - np.arccos(u.astype(np.float64))).max() < 1e-6
    y, x = _inputs("any", seed=3), _inputs("any", seed=4)
    assert np.abs(orc.math_eval(4, y, x)[0] - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 1e-6


def test_rng_stream_on_device(orc, api, torch_gpu):
    for (u, v) in [(0.0, 0.0), (0.5, 0.5), (0.99947923, 0.99907404), (0.123, 0.877)]:
        ref = orc.rng_kat(u, v, FRAME_RANDOM, 64)
        out = api.test_rng(u, v, FRAME_RANDOM, 64).cpu().numpy()
        assert np.array_equal(_bits(out), _bits(ref))
    out = api.test_rng(0.0, 0.0, FRAME_RANDOM, 4).cpu().numpy()
    assert [int(b) for b in _bits(out[1:])] == [0x3E0920B8, 0x3F6DA0F8, 0x3EEA5770, 0x3E845B2C]
