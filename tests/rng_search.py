"""Finds per-frame random vectors that put a chosen pixel's hash RNG into the chain's fixed point 0 (hash(0) = 0, random.glsl:24-33):
every draw of that pixel is 0, every free flight has length 0 and DeltaTrack runs into its cap of 128 collisions
(path_trace.glsl:161-173) -- the case the empty-space early-out has to get right.  Test infrastructure (numpy restatement of
random.glsl:24-64); the vectors the tests use were found with this script and are checked against the oracle's RNG on the CPU."""
import numpy as np

M23 = np.uint32(0x7FFFFF)


def hash1(x):
    x = np.asarray(x, np.uint32)
    with np.errstate(over="ignore"):
        x = x + (x << np.uint32(10))
        x = x ^ (x >> np.uint32(6))
        x = x + (x << np.uint32(3))
        x = x ^ (x >> np.uint32(11))
        x = x + (x << np.uint32(15))
    return x


def float_construct(h):
    return ((np.asarray(h, np.uint32) & M23) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def f2u(x):
    return np.asarray(x, np.float32).view(np.uint32)


def random2(x, y):
    return float_construct(hash1(f2u(x) ^ hash1(f2u(y))))


def random4(v):
    v = np.asarray(v, np.float32)
    return float_construct(hash1(f2u(v[0]) ^ hash1(f2u(v[1])) ^ hash1(f2u(v[2])) ^ hash1(f2u(v[3]))))


def init_random(gx, y, gw, gh, frame_random):
    """InitRandom(uv) of the pixel (random.glsl:61-64; uv = pixel * (1 / size), nrc-constants.glsl:28-29)"""
    u = np.float32(gx) * (np.float32(1.0) / np.float32(gw))
    v = np.float32(y) * (np.float32(1.0) / np.float32(gh))
    return random2(random2(u, v), random4(frame_random))


def frame_random_for_state0(gx, y, gw, gh, tries=64):
    """a frame random vector (4 floats in [0, 1)) for which pixel (gx, y) of a gw x gh frame starts in RNG state 0"""
    u = np.float32(gx) * (np.float32(1.0) / np.float32(gw))
    v = np.float32(y) * (np.float32(1.0) / np.float32(gh))
    p = f2u(random2(u, v))
    mq = np.arange(1 << 23, dtype=np.uint32)
    q = float_construct(mq)                                   # every value random4 can return
    hit = np.nonzero((hash1(p ^ hash1(f2u(q))) & M23) == 0)[0]
    if hit.size == 0:
        return None
    want = set(int(h) for h in hit)                           # mantissas of suitable Q
    v0 = float_construct(mq)
    for k in range(tries):
        rest = np.array([(k + 1) / (tries + 1.0), 0.75, 0.125], np.float32)
        c = hash1(f2u(rest[0])) ^ hash1(f2u(rest[1])) ^ hash1(f2u(rest[2]))
        got = hash1(f2u(v0) ^ c) & M23
        idx = np.nonzero(np.isin(got, hit))[0]
        if idx.size:
            fr = [float(v0[idx[0]]), float(rest[0]), float(rest[1]), float(rest[2])]
            assert float(init_random(gx, y, gw, gh, fr)) == 0.0
            return fr
    return None


if __name__ == "__main__":
    import sys
    gw, gh = int(sys.argv[1]), int(sys.argv[2])
    for gx, y in [(int(a), int(b)) for a, b in zip(sys.argv[3::2], sys.argv[4::2])]:
        print((gx, y), frame_random_for_state0(gx, y, gw, gh))
