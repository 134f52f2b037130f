"""Oracle vs the RNG known-answer vectors of SURVEY.md App. D (integer-exact restatement of
data/shader/include/random.glsl:24-70)."""
import numpy as np

from conftest import FRAME_RANDOM


def bits(x):
    return int(np.float32(x).view(np.uint32))


def test_hash_kat(orc):
    assert orc.hash(0) == 0x00000000
    assert orc.hash(1) == 0x124EA49D
    assert orc.hash(0x3F800000) == 0xF2496DC0


def test_random1_kat(orc):
    assert bits(orc.random1(0.5)) == 0x3E5FDC00


def test_init_random_streams(orc):
    W, H = 1920, 1080
    cases = {
        (0, 0): (0.0437352657, [0x3E0920B8, 0x3F6DA0F8, 0x3EEA5770, 0x3E845B2C]),
        (960, 540): (0.472251296, [0x3F65851E, 0x3F5C0748, 0x3F743E42, 0x3F2F7944]),
        (1919, 1079): (0.145569921, [0x3F351F08, 0x3E8491BC, 0x3F7A289E, 0x3F6D6D20]),
    }
    for (x, y), (state0, draws) in cases.items():
        u = np.float32(x) * (np.float32(1.0) / np.float32(W))      # ONE_OVER_RENDER_WIDTH first (nrc-constants.glsl:28)
        v = np.float32(y) * (np.float32(1.0) / np.float32(H))
        out = orc.rng_kat(u, v, FRAME_RANDOM, 4)
        assert abs(float(out[0]) - state0) < 1e-9
        assert [bits(d) for d in out[1:]] == draws


def test_state_zero_is_fixed_point(orc):
    # hash(0) = 0 => randomState 0.0 stays 0.0
    assert orc.random1(0.0) == 0.0


def test_draws_in_unit_interval(orc):
    out = orc.rng_kat(0.3, 0.7, [0.1, 0.2, 0.3, 0.4], 4096)
    assert (out >= 0).all() and (out < 1).all()
    assert 0.45 < out[1:].mean() < 0.55


def test_state0_frame_random_puts_the_pixel_into_the_rng_fixed_point(orc):
    """tests/rng_search.py (numpy restatement of random.glsl) agrees with the oracle's hash, and the vector the empty-space test
    uses really starts pixel (2, 3) of a 256x144 frame in state 0, where every draw is 0 (hash(0) = 0)"""
    import rng_search as rs
    xs = np.array([0, 1, 0x3f800000, 0xdeadbeef, 0x7fffffff, 12345], np.uint32)
    assert [int(h) for h in rs.hash1(xs)] == [orc.hash(int(x)) for x in xs]
    fr = [0.7795426845550537, 0.04615384712815285, 0.75, 0.125]
    assert float(rs.init_random(2, 3, 256, 144, fr)) == 0.0
    assert float(rs.init_random(3, 3, 256, 144, fr)) != 0.0
    u, v = np.float32(2) * (np.float32(1.0) / np.float32(256)), np.float32(3) * (np.float32(1.0) / np.float32(144))
    out = orc.rng_kat(u, v, fr, 3)                     # the oracle's own InitRandom + three draws
    assert [float(x) for x in out] == [0.0, 0.0, 0.0, 0.0]
