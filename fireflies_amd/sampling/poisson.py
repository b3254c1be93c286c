"""Bridson Poisson-disk sampling with a spatially varying radius (host, numpy).

Counterpart of fireflies/sampling/poisson.py (`bridson(sampling_map)` -> (count, samples)), used
only to initialise a blue-noise pattern (projection/laser.py:95-145).  Sequential by nature and
off the hot path; the sample set depends on the RNG, so there is no golden vector for it — tests
check the defining property (pairwise distance >= local radius) instead.
"""
import numpy as np


def bridson(sampling_map, k: int = 30, rng=None):
    """sampling_map[x, y] = minimum distance wanted around (x, y).  Returns (n, [[x, y], ...])."""
    rng = np.random.default_rng() if rng is None else rng
    W, H = sampling_map.shape
    rmin = float(sampling_map.min())
    cell = rmin / np.sqrt(2.0)
    gw, gh = int(np.ceil(W / cell)), int(np.ceil(H / cell))
    grid = -np.ones((gw, gh), np.int64)
    pts = []

    def radius(p):
        return float(sampling_map[min(int(p[0]), W - 1), min(int(p[1]), H - 1)])

    def fits(p):
        r = radius(p)
        reach = int(np.ceil(r / cell))
        gx, gy = int(p[0] / cell), int(p[1] / cell)
        for ix in range(max(gx - reach, 0), min(gx + reach + 1, gw)):
            for iy in range(max(gy - reach, 0), min(gy + reach + 1, gh)):
                j = grid[ix, iy]
                if j >= 0 and np.hypot(*(pts[j] - p)) < r:
                    return False
        return True

    first = np.array([rng.uniform(0, W), rng.uniform(0, H)])
    pts.append(first)
    grid[int(first[0] / cell), int(first[1] / cell)] = 0
    active = [0]
    while active:
        i = active[rng.integers(len(active))]
        base, r = pts[i], radius(pts[i])
        for _ in range(k):
            ang, rad = rng.uniform(0, 2 * np.pi), rng.uniform(r, 2 * r)
            q = base + rad * np.array([np.cos(ang), np.sin(ang)])
            if not (0 <= q[0] < W and 0 <= q[1] < H) or not fits(q):
                continue
            pts.append(q)
            grid[int(q[0] / cell), int(q[1] / cell)] = len(pts) - 1
            active.append(len(pts) - 1)
            break
        else:
            active.remove(i)
    return len(pts), [p.tolist() for p in pts]
