#!/usr/bin/env python3
# NOTE (round 6): -DNRC_DIAG_LASTDIR / -DNRC_DIAG_BISECT left the product source; this tool builds them from the tree of commit aa01da1 (round 5): git worktree add /tmp/r05 aa01da1
"""Offline analysis of the k_gen_rays glitch samples of tests/cpp/stress_main.cpp (-DNRC_DIAG_LASTDIR build): for every affected
pixel the log holds the RNG state and the incoming direction in front of the path's last new_ray_dir, the direction the glitching
wave produced and the direction the other rendering produced.  This script restates new_ray_dir (dir_gen.glsl:22-64 as
nrc_integrator.hip states it, fp32 with single-rounding FMAs) with every intermediate replaceable, checks that the restatement
reproduces the GOOD direction bit for bit, and asks which single wrong intermediate explains the BAD one.

  python tools/lastdir_analyze.py gpurun_out/stress11/lastdir.log
"""
import re
import struct
import sys

import numpy as np

F = np.float32
PI, TWO_PI, HALF_PI = F(3.14159274101257324), F(6.28318548202514648), F(1.57079637050628662)


def u2f(u):
    return np.frombuffer(struct.pack("<I", u & 0xFFFFFFFF), np.float32)[0]


def f2u(f):
    return struct.unpack("<I", np.float32(f).tobytes())[0]


def fma(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def hash1(x):
    x = (x + (x << 10)) & 0xFFFFFFFF
    x ^= x >> 6
    x = (x + (x << 3)) & 0xFFFFFFFF
    x ^= x >> 11
    x = (x + (x << 15)) & 0xFFFFFFFF
    return x


def random1(x):
    return F(u2f((hash1(f2u(x)) & 0x007FFFFF) | 0x3F800000) - F(1.0))


def dot(a, b):
    return fma(a[2], b[2], fma(a[1], b[1], F(a[0] * b[0])))


def asinf(x):
    a = F(abs(x))
    if not a <= 1.0:
        return F(np.nan)
    big = a > 0.5
    if big:
        z = F(F(0.5) * F(F(1.0) - a)); w = F(np.sqrt(z))
    else:
        w = a; z = F(a * a)
    p = F(4.2163199048E-2)
    for c in (2.4181311049E-2, 4.5470025998E-2, 7.4953002686E-2, 1.6666752422E-1):
        p = fma(p, z, F(c))
    p = F(p * z)
    p = fma(p, w, w)
    if big:
        p = F(p + p); p = F(HALF_PI - p)
    return F(-p) if x < 0 else p


def acosf_clamped(x):
    x = F(min(max(x, F(-1.0)), F(1.0)))
    if x < -0.5:
        return F(PI - F(F(2.0) * asinf(F(np.sqrt(F(F(0.5) * F(F(1.0) + x)))))))
    if x > 0.5:
        return F(F(2.0) * asinf(F(np.sqrt(F(F(0.5) * F(F(1.0) - x))))))
    return F(HALF_PI - asinf(x))


def sincosf(x):
    ax = F(abs(x))
    j = int(F(ax * F(1.27323949337005615)))
    j = (j + 1) & ~1
    y = F(j)
    r = fma(-y, F(0.78515625), ax)
    r = fma(-y, F(2.4187564849853515625e-4), r)
    r = fma(-y, F(3.77489497744594108e-8), r)
    z = F(r * r)
    ps = F(-1.9515295891E-4)
    ps = fma(ps, z, F(8.3321608736E-3)); ps = fma(ps, z, F(-1.6666654611E-1)); ps = F(ps * z); ps = fma(ps, r, r)
    pc = F(2.443315711809948E-005)
    pc = fma(pc, z, F(-1.388731625493765E-003)); pc = fma(pc, z, F(4.166664568298827E-002)); pc = F(pc * z); pc = F(pc * z)
    pc = fma(F(-0.5), z, pc); pc = F(pc + F(1.0))
    q = (j >> 1) & 3
    s, c = ((ps, pc), (pc, F(-ps)), (F(-ps), F(-pc)), (F(-pc), ps))[q]
    if x < 0:
        s = F(-s)
    return s, c


class Ctx:
    """overrides: name -> function(value, ctx) -> replacement, applied to the named intermediate"""

    def __init__(self, ov=None):
        self.ov = ov or {}
        self.v = {}

    def tap(self, name, value):
        self.v.setdefault(name, value)
        if name in self.ov:
            value = self.ov[name](value, self)
        return value


def normalize(a, cx, name):
    inv = cx.tap(name + ".inv", F(F(1.0) / F(np.sqrt(dot(a, a)))))
    return [F(a[0] * inv), F(a[1] * inv), F(a[2] * inv)]


def rotate(axis, angle, v, cx, name):
    axis = normalize(axis, cx, name + ".axis")
    s, co = sincosf(angle)
    s, co = cx.tap(name + ".sin", s), cx.tap(name + ".cos", co)
    oc = F(F(1.0) - co)
    ox, oy, oz = F(oc * axis[0]), F(oc * axis[1]), F(oc * axis[2])
    c0 = [fma(ox, axis[0], co), fma(ox, axis[1], F(-F(axis[2] * s))), fma(oz, axis[0], F(axis[1] * s))]
    c1 = [fma(ox, axis[1], F(axis[2] * s)), fma(oy, axis[1], co), fma(oy, axis[2], F(-F(axis[0] * s)))]
    c2 = [fma(oz, axis[0], F(-F(axis[1] * s))), fma(oy, axis[2], F(axis[0] * s)), fma(oz, axis[2], co)]
    return [fma(c2[k], v[2], fma(c1[k], v[1], F(c0[k] * v[0]))) for k in range(3)]


def new_ray_dir(rng, old, g=F(0.8), ov=None):
    cx = Ctx(ov)
    old = normalize([F(x) for x in old], cx, "old")
    ortho = [old[1], F(-old[0]), F(0.0)] if old[2] < old[0] else [F(0.0), F(-old[2]), old[1]]
    if ortho == [0, 0, 0]:
        ortho = [F(0.0), F(1.0), F(0.0)]
    ortho = normalize(ortho, cx, "ortho")
    u1 = cx.tap("u1", random1(rng))
    sqr = F(F(F(1.0) - F(g * g)) / fma(F(F(2.0) * g), u1, F(F(1.0) - g)))
    cos_t = cx.tap("cos_theta", F(fma(F(-sqr), sqr, F(F(1.0) + F(g * g))) / F(F(2.0) * g)))
    angle = cx.tap("angle1", acosf_clamped(cos_t))
    nd = rotate(ortho, angle, old, cx, "rot1")
    u2 = cx.tap("u2", random1(u1))
    angle2 = cx.tap("angle2", F(u2 * TWO_PI))
    nd = rotate(old, angle2, nd, cx, "rot2")
    out = normalize(nd, cx, "out")
    return out, cx


def bits(v):
    return [f2u(x) for x in v]


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stress11/lastdir.log"
    pat = re.compile(r"px \((\d+), (\d+)\) set (\d).*LASTDIR rng ([0-9a-f]{8}) in ([0-9a-f]{8}) ([0-9a-f]{8}) ([0-9a-f]{8}) out_tiles ([0-9a-f]{8}) ([0-9a-f]{8}) ([0-9a-f]{8}) out_whole ([0-9a-f]{8}) ([0-9a-f]{8}) ([0-9a-f]{8})")
    samples = []
    for line in open(path):
        m = pat.search(line)
        if m:
            g = m.groups()
            samples.append(dict(px=(int(g[0]), int(g[1])), rng=u2f(int(g[3], 16)), old=[u2f(int(x, 16)) for x in g[4:7]], a=[u2f(int(x, 16)) for x in g[7:10]],
                                b=[u2f(int(x, 16)) for x in g[10:13]]))
    print("%d samples" % len(samples))
    # single-intermediate hypotheses: stale copies of a sibling value, or a dropped step
    names = ["old.inv", "ortho.inv", "rot1.axis.inv", "rot2.axis.inv", "out.inv", "u1", "u2", "cos_theta", "angle1", "angle2", "rot1.sin", "rot1.cos", "rot2.sin", "rot2.cos"]
    stats = {}
    for s in samples:
        ref, cx = new_ray_dir(s["rng"], s["old"])
        which = "tiles" if bits(ref) == bits(s["b"]) else ("whole" if bits(ref) == bits(s["a"]) else None)
        good, bad = (s["b"], s["a"]) if which == "tiles" else (s["a"], s["b"])
        if which is None:
            print("px %s: the restatement matches NEITHER side (max |d| good-guess %.3g)" % (s["px"], max(abs(np.array(ref) - np.array(s["b"])))))
            continue
        best = []
        for tgt in names:
            for src in names + ["one", "zero"]:
                if src == tgt:
                    continue
                val = F(1.0) if src == "one" else (F(0.0) if src == "zero" else cx.v.get(src))
                if val is None:
                    continue
                try:
                    out, _ = new_ray_dir(s["rng"], s["old"], ov={tgt: (lambda v, c, val=val: val)})
                except Exception:
                    continue
                if not np.isfinite(out).all():
                    continue
                err = float(max(abs(np.array(out, np.float64) - np.array(bad, np.float64))))
                best.append((err, tgt, src))
        best.sort()
        dev = float(np.degrees(np.arccos(min(1.0, float(np.dot(np.array(good, np.float64), np.array(bad, np.float64)))))))
        cos_go = float(np.dot(np.array(good, np.float64), np.array(s["old"], np.float64)) / np.linalg.norm(s["old"]))
        cos_bo = float(np.dot(np.array(bad, np.float64), np.array(s["old"], np.float64)) / np.linalg.norm(s["old"]))
        print("px %s: glitch on the %s side; bad vs good %.3f deg; cos(polar) good %.6f bad %.6f | best single-value explanations: %s"
              % (s["px"], "tiles" if which == "whole" else "whole", dev, cos_go, cos_bo, ", ".join("%s<-%s (%.1e)" % (t, sr, e) for e, t, sr in best[:3])))
        if best:
            stats[(best[0][1], best[0][2])] = stats.get((best[0][1], best[0][2]), 0) + (1 if best[0][0] < 1e-5 else 0)
    print("explanations within 1e-5:", stats)


if __name__ == "__main__":
    main()
