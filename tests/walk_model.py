"""Lane-utilisation model of k_gen_rays' tracking loops from the ORACLE's per-walk free-flight counts (analysis tool, not a test:
`python tests/walk_model.py [W H N]`; it lives under tests/ because it calls the oracle).  For the bench view it prints, per loop of
the kernel (delta / dir-light / environment walk of vertex 1 and 2), the lane utilisation of today's lock-step loops, and the
instruction-slot cost of a few wave organisations."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrc_hpm_renderer_amd import scene as sc  # noqa: E402
from oracle import Oracle  # noqa: E402

# instruction slots (ISA counts of the shipped kernel, tools/loop_isa.py): one two-collision trip of the ratio / delta loop, one pair
# iteration (four collisions), one SDF march iteration, new_ray_dir, the rest of a transition
C_RATIO, C_DELTA, C_PAIR, C_FEE, C_NEWDIR, C_MISC = 181, 215, 237, 41, 300, 60      # (165, 201, 214 since the table-driven log; the comparisons in DESIGN.md used these)
FEE_ITERS = 7


def trips_of(f, delta):
    return (f + 1) // 2


def lockstep(t, pair=True):
    """t: [tiles][64] trips per lane of ONE loop -> (wave iterations in 64-lane units, useful trips); with the pair tail the
    iterations in which at most 31 lanes walk advance two trips each at C_PAIR / C_RATIO the cost"""
    srt = np.sort(t, axis=1)[:, ::-1]            # descending
    full = srt[:, 31] if pair else srt[:, 0]     # iterations with >= 32 lanes alive
    rest = srt[:, 0] - full
    it = full + (np.ceil(rest / 2.0) * (C_PAIR / C_RATIO) if pair else rest)
    return it, t.sum(axis=1)


def main():
    W, H, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 256)
    vol = sc.cached_volume("cloud", N, seed=1337)
    scene = sc.make_scene(vol, scene_id=4, env=sc.procedural_sky())
    cam = sc.make_camera(aspect=W / H)
    fr = sc.frame_randoms(1)[0]
    cache = "/tmp/walk_lengths_%dx%d_%d.npy" % (W, H, N)
    if os.path.exists(cache):
        wl = np.load(cache)
    else:
        t0 = time.time()
        wl, _ = Oracle().nrc_walk_lengths(scene, cam, W, H, 1, 0.0, fr, walks_per_pixel=8, threads=8)
        print("oracle frame: %.1f s" % (time.time() - t0))
        np.save(cache, wl)
    wl = wl[:H // 8 * 8, :W // 8 * 8].astype(np.int64)
    th, tw = wl.shape[0] // 8, wl.shape[1] // 8
    tiles = wl.reshape(th, 8, tw, 8, 8).transpose(0, 2, 1, 3, 4).reshape(th * tw, 64, 8)[:, :, :6]      # [tile][lane][walk]
    # (the kernel's exact empty-space mask skips the tiles whose rays cannot meet a non-empty voxel; approximated here by: no pixel scatters)
    work = tiles[(tiles[:, :, 1] > 0).any(axis=1)]
    print("tiles %d, with a walk %d; flights per pixel %.2f" % (tiles.shape[0], work.shape[0], wl.sum() / (W * H)))
    names = ["delta 1", "dir 1", "env 1", "delta 2", "dir 2", "env 2"]
    trips = (work + 1) // 2                       # [tile][lane][walk]
    tot_today = 0.0
    print("today's lock-step loops (pair tail in the ratio loops):")
    for k in range(6):
        delta = k % 3 == 0
        it, useful = lockstep(trips[:, :, k], pair=not delta)
        cost = it.sum() * (C_DELTA if delta else C_RATIO)
        tot_today += cost
        print("  %-8s lanes with a walk %.2f  mean trips %.1f  max %.1f  utilisation %.3f" %
              (names[k], (trips[:, :, k] > 0).mean(), trips[:, :, k][trips[:, :, k] > 0].mean(), trips[:, :, k].max(axis=1).mean(),
               useful.sum() / (64.0 * lockstep(trips[:, :, k], pair=False)[0].sum())))
    n_vert = (work[:, :, [0, 3]] > 0).any(axis=1).sum()     # wave-level vertex stages
    trans_today = n_vert * (3 * FEE_ITERS * C_FEE + 2 * C_NEWDIR + 3 * C_MISC)
    print("  slots per working tile: loops %.0f + transitions %.0f" % (tot_today / work.shape[0], trans_today / work.shape[0]))
    ideal = (trips[:, :, [0, 3]].sum() * C_DELTA + trips[:, :, [1, 2, 4, 5]].sum() * C_RATIO) / 64.0
    print("  at 100 %% lane use the loops would cost %.0f per working tile" % (ideal / work.shape[0]))

    # flat loop inside a tile: every lane walks its own sequence; a lane whose walk ends parks until `thresh` lanes are parked or nobody
    # walks, then the parked lanes' transitions run (each kind present costs its instructions once per batch)
    c_trip = 1.06 * (C_DELTA + 2 * C_RATIO) / 3.0      # merged body: the longer hash chain for every lane + selects
    t_cost = [FEE_ITERS * C_FEE + C_MISC, C_NEWDIR + FEE_ITERS * C_FEE + C_MISC, C_NEWDIR + FEE_ITERS * C_FEE + C_MISC]
    rng = np.random.default_rng(1)
    sample = work[rng.choice(work.shape[0], size=min(1500, work.shape[0]), replace=False)]
    for thresh in (1, 8, 16, 32, 64):
        slots = 0.0
        for tile in sample:
            seq = (tile + 1) // 2                   # [lane][walk] trips
            nwalk = (tile > 0).sum(axis=1)
            cur = np.zeros(64, np.int64)            # walk index
            left = np.where(nwalk > 0, seq[:, 0], 0)
            parked = np.zeros(64, bool)
            done = nwalk == 0
            while not done.all():
                walking = ~done & ~parked
                if walking.any() and parked.sum() < thresh:
                    slots += c_trip
                    left[walking] -= 1
                    fin = walking & (left <= 0)
                    parked |= fin
                else:
                    kinds = set()
                    for l in np.nonzero(parked)[0]:
                        cur[l] += 1
                        if cur[l] >= nwalk[l]:
                            done[l] = True
                        else:
                            kinds.add(int(cur[l]) % 3)
                            left[l] = seq[l, cur[l]]
                    for kd in kinds:
                        slots += t_cost[(kd + 2) % 3]      # entering walk kind kd: 1 <- T1 (after delta), 2 <- T2, 0 <- T3
                    parked[:] = False
        print("flat loop in a tile, transitions when %2d lanes wait: %.0f slots per working tile" % (thresh, slots / sample.shape[0]))

    # walk pool of G tiles (a wave that owns G tiles, or G waves of a workgroup sharing one pool through LDS): finished walks go to a
    # transition queue, a batch of up to 64 runs when 64 wait or no walk is ready; ready walks refill idle lanes at once.  `lanes` lanes
    # execute trips (one wave: 64).  Slots are per wave-instruction, so a pool served by one wave is directly comparable.
    c_io = 60                                        # walk state to / from LDS per transition batch and per refill round
    order = rng.permutation(work.shape[0])
    for G in (1, 2, 4, 8):
        groups = order[:min(work.shape[0], 1200 * 1) // G * G].reshape(-1, G)[:600 // G + 1]
        slots = 0.0
        n_tiles = 0
        st = dict(trip_it=0, act=0, batches=0, bsz=0, refills=0)
        for g in groups:
            tile = work[g].reshape(-1, 6)                       # [pixel][walk] flights
            seq = (tile + 1) // 2
            nwalk = (tile > 0).sum(axis=1)
            P = tile.shape[0]
            cur = np.zeros(P, np.int64)
            ready = [p for p in range(P) if nwalk[p] > 0]       # first walks located by the (full-width) prologue
            waiting = []
            lane_p = -np.ones(64, np.int64)
            lane_left = np.zeros(64, np.int64)
            remaining = int((nwalk > 0).sum())
            while remaining > 0:
                idle = np.nonzero(lane_p < 0)[0]
                if len(idle) and ready:
                    k = min(len(idle), len(ready))
                    for l in idle[:k]:
                        p = ready.pop()
                        lane_p[l] = p
                        lane_left[l] = seq[p, cur[p]]
                    slots += c_io
                    st['refills'] += 1
                active = lane_p >= 0
                if len(waiting) >= 64 or (not active.any() and waiting) or (active.sum() < 16 and len(waiting) >= 16):
                    batch, waiting = waiting[:64], waiting[64:]
                    st['batches'] += 1; st['bsz'] += len(batch)
                    kinds = set()
                    for p in batch:
                        cur[p] += 1
                        if cur[p] >= nwalk[p]:
                            remaining -= 1
                        else:
                            kinds.add(int(cur[p]) % 3)
                            ready.append(p)
                    for kd in kinds:
                        slots += t_cost[(kd + 2) % 3]
                    slots += c_io + 80                           # state I/O + the located first trip of the new walks
                    continue
                if active.any():
                    slots += c_trip + 12                         # + queue bookkeeping per iteration
                    st['trip_it'] += 1; st['act'] += int(active.sum())
                    lane_left[active] -= 1
                    fin = active & (lane_left <= 0)
                    for l in np.nonzero(fin)[0]:
                        waiting.append(int(lane_p[l]))
                        lane_p[l] = -1
            n_tiles += G
        print("walk pool of %d tiles per wave, batched transitions: %.0f slots per working tile" % (G, slots / n_tiles),
              "| per tile: trip iterations %.1f (lanes active %.1f), batches %.1f (size %.1f), refills %.1f" %
              (st['trip_it'] / n_tiles, st['act'] / max(st['trip_it'], 1), st['batches'] / n_tiles, st['bsz'] / max(st['batches'], 1), st['refills'] / n_tiles))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "groups"):
    main()


def group_mode_model():
    """lock-step loops whose thin trips hand every surviving walk to a GROUP of L lanes (L = the largest power of two <= 64 / walks alive,
    2 L collisions per iteration): instruction slots per working tile against today's loops (pair tail in the ratio loops only)"""
    W, H, N = 1920, 1080, 256
    wl = np.load("/tmp/walk_lengths_%dx%d_%d.npy" % (W, H, N))[:H // 8 * 8, :W // 8 * 8].astype(np.int64)
    th, tw = wl.shape[0] // 8, wl.shape[1] // 8
    tiles = wl.reshape(th, 8, tw, 8, 8).transpose(0, 2, 1, 3, 4).reshape(th * tw, 64, 8)[:, :, :6]
    work = tiles[(tiles[:, :, 1] > 0).any(axis=1)]
    rng = np.random.default_rng(2)
    sample = work[rng.choice(work.shape[0], size=1500, replace=False)]

    def cost_group(L, delta):
        return (52 * L + 140) if delta else (38 * L + 130)

    def run(flights, delta, policy):
        """flights: [64] collisions (free flights) per lane of one loop -> slots"""
        left = flights.copy()
        slots = 0.0
        while True:
            a = int((left > 0).sum())
            if a == 0:
                return slots
            L = policy(a, int(left.max()), delta)
            if L == 1:
                slots += C_DELTA if delta else C_RATIO
                left = np.maximum(left - 2, 0)
            elif L == -2:      # today's pair tail
                slots += C_PAIR
                left = np.maximum(left - 4, 0)
            else:
                slots += cost_group(L, delta)
                left = np.maximum(left - 2 * L, 0)

    def today(a, mx, delta):
        return -2 if (a <= 31 and not delta) else 1

    def groups(cap):
        def pol(a, mx, delta):
            if a > 32:
                return 1
            L = 1 << int(np.floor(np.log2(64 // a)))
            while L > 2 and 2 * L > cap * mx:      # no wider than the longest surviving walk needs
                L //= 2
            return max(L, 2)
        return pol

    for name, pol in (("today", today), ("groups, L <= what the longest walk needs", groups(1.0)), ("groups, L <= 2 x that", groups(2.0)),
                      ("groups, uncapped", groups(1e9))):
        tot = np.zeros(6)
        for tile in sample:
            for k in range(6):
                tot[k] += run(tile[:, k], k % 3 == 0, pol)
        tot /= sample.shape[0]
        print("%-44s per working tile: %6.0f  (delta1 %5.0f dir1 %5.0f env1 %5.0f delta2 %5.0f dir2 %5.0f env2 %5.0f)" % ((name, tot.sum()) + tuple(tot)))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "groups":
    group_mode_model()
