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


def unhash1(x):
    """the inverse of hash1 (each of its five steps is a bijection of uint32)"""
    x = np.asarray(x, np.uint32).copy()
    with np.errstate(over="ignore"):
        x = x * np.uint32(pow(1 + (1 << 15), -1, 1 << 32))
        x = x ^ (x >> np.uint32(11)) ^ (x >> np.uint32(22))
        x = x * np.uint32(pow(9, -1, 1 << 32))
        t = x
        for _ in range(5):
            t = x ^ (t >> np.uint32(6))
        x = t
        x = x * np.uint32(pow(1025, -1, 1 << 32))
    return x


def frame_random_for_value(q_mantissa, tries=64):
    """a frame random vector (4 floats in [0, 1)) with random4(vector) == float_construct(q_mantissa)"""
    v0 = float_construct(np.arange(1 << 23, dtype=np.uint32))
    for k in range(tries):
        rest = np.array([(k + 1) / (tries + 1.0), 0.75, 0.125], np.float32)
        c = hash1(f2u(rest[0])) ^ hash1(f2u(rest[1])) ^ hash1(f2u(rest[2]))
        idx = np.nonzero((hash1(f2u(v0) ^ c) & M23) == np.uint32(q_mantissa))[0]
        if idx.size:
            return [float(v0[idx[0]]), float(rest[0]), float(rest[1]), float(rest[2])]
    return None


def two_state0_pixels_in_one_tile(gw, gh, want=1):
    """frame random vectors for which TWO pixels of one 8x8 tile of a gw x gh frame start in RNG state 0 (k_hot_tiles then lists the
    tile twice: ADVICE r03).  A pixel p starts in state 0 iff hash(bits(seed_uv(p)) ^ hash(bits(q))) has 23 zero low bits, q =
    random4(frame random) -- one of 2^23 floats.  Inverted: hash(bits(q)) must be bits(seed_uv) ^ unhash(hi << 23) for one of the 512
    values of hi; about one q per pixel exists, and two pixels of a tile share theirs for ~8 tiles of a 1920x1080 frame."""
    mq = np.arange(1 << 23, dtype=np.uint32)
    hq = hash1(f2u(float_construct(mq)))
    order = np.argsort(hq)
    hq_sorted = hq[order]
    cand = unhash1(np.arange(512, dtype=np.uint32) << np.uint32(23))
    found = []
    tiles_x = (gw + 7) // 8
    for ty in range((gh + 7) // 8):
        ys, xs = np.meshgrid(np.arange(ty * 8, min(ty * 8 + 8, gh)), np.arange(gw), indexing="ij")
        u = xs.astype(np.float32) * (np.float32(1.0) / np.float32(gw))
        v = ys.astype(np.float32) * (np.float32(1.0) / np.float32(gh))
        p = f2u(random2(u, v)).ravel()
        need = (p[:, None] ^ cand[None, :]).ravel()
        pos = np.minimum(np.searchsorted(hq_sorted, need), hq_sorted.size - 1)
        ok = np.nonzero(hq_sorted[pos] == need)[0]
        if ok.size == 0:
            continue
        pix = ok // 512
        q = order[pos[ok]]                                   # mantissa of the q that puts pixel `pix` into state 0
        tile = xs.ravel()[pix] // 8
        key = tile.astype(np.int64) * (1 << 23) + q
        uniq, counts = np.unique(key, return_counts=True)
        for kk in uniq[counts >= 2]:
            sel = np.nonzero(key == kk)[0]
            px = [(int(xs.ravel()[pix[i]]), int(ys.ravel()[pix[i]])) for i in sel]
            if len(set(px)) < 2:
                continue
            fr = frame_random_for_value(int(kk % (1 << 23)))
            if fr is None:
                continue
            for (x, y) in px:
                assert float(init_random(x, y, gw, gh, fr)) == 0.0
            found.append((fr, px))
            if len(found) >= want:
                return found
    return found


if __name__ == "__main__":
    import sys
    if sys.argv[1] == "pair":
        for fr, px in two_state0_pixels_in_one_tile(int(sys.argv[2]), int(sys.argv[3]), want=int(sys.argv[4]) if len(sys.argv) > 4 else 1):
            print(fr, px)
        sys.exit(0)
    gw, gh = int(sys.argv[1]), int(sys.argv[2])
    for gx, y in [(int(a), int(b)) for a, b in zip(sys.argv[3::2], sys.argv[4::2])]:
        print((gx, y), frame_random_for_state0(gx, y, gw, gh))
