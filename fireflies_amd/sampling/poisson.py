"""Bridson Poisson-disk sampling with a spatially varying radius (host, numpy).

Counterpart of fireflies/sampling/poisson.py:16-116 (`bridson(radius_map)` -> (count, samples)), used only to
initialise a blue-noise pattern (projection/laser.py:95-145).  Sequential by nature and off the hot path.

The sample set is defined by the random draws, so this follows the reference's draw ORDER and its acceptance
rule exactly, and by default draws from the same GLOBAL numpy generator (`np.random.seed(s)` reproduces the
reference's samples: tests/golden/g11_bridson.npz):
  * first point: two uniforms, (u0 * rows, u1 * cols); the map is indexed [row = coords[0], col = coords[1]];
  * per round: `randint(len(active))` picks an active point; then k candidates, each consuming a radius draw
    (uniform in [r, 2r) for "default", normal(1.5 r, 0.2 r) for "normDist") and an angle draw, placed at
    (+ r sin a, + r cos a); EVERY accepted candidate of the k is kept (the reference does not stop at the first),
    and the active point is retired only when none of its k candidates was accepted;
  * acceptance is on the INTEGER occupancy grid (one cell per map entry): a candidate is rejected if any cell in
    the square of half-width ceil(radius there) around its cell holds a sample — not a Euclidean test.
"""
import numpy as np


class _GlobalNumpy:
    """the module-level numpy generator, with the three draws the algorithm needs"""

    @staticmethod
    def random():
        return np.random.random()

    @staticmethod
    def randint(n):
        return np.random.randint(n)

    @staticmethod
    def normal(mu, sd):
        return np.random.normal(mu, sd)


class _FromGenerator:
    def __init__(self, g):
        self.g = g

    def random(self):
        return self.g.random()

    def randint(self, n):
        return int(self.g.integers(n))

    def normal(self, mu, sd):
        return self.g.normal(mu, sd)


def bridson(radius, k: int = 30, radiusType: str = "default", rng=None):
    """radius[row, col] = minimum spacing wanted around that cell.  Returns (n, array [n, 2] of (row, col) positions).
    `rng`: None = numpy's global generator (like the reference), or a numpy Generator."""
    if radiusType not in ("default", "normDist"):
        raise ValueError(f"unknown radiusType {radiusType!r}")
    draw = _GlobalNumpy if rng is None else _FromGenerator(rng)
    radius = np.asarray(radius, dtype=np.float64)
    rows, cols = radius.shape
    occupied = np.zeros((rows, cols), dtype=bool)

    def cell(p):
        return int(np.floor(p[0])), int(np.floor(p[1]))

    first = (draw.random() * rows, draw.random() * cols)
    occupied[cell(first)] = True
    samples = [np.asarray(first, np.float64)]
    active = [samples[0]]
    while active:
        pick = draw.randint(len(active))
        base = active[pick]
        r_here = radius[cell(base)]
        accepted_any = False
        for _ in range(k):
            dist = r_here * (draw.random() + 1.0) if radiusType == "default" else r_here * draw.normal(1.5, 0.2)
            ang = 2.0 * np.pi * draw.random()
            cand = np.array([base[0] + dist * np.sin(ang), base[1] + dist * np.cos(ang)])
            if not (0.0 <= cand[0] <= rows and 0.0 <= cand[1] <= cols):
                continue
            cr, cc = cell(cand)
            if cr >= rows or cc >= cols:  # exactly on the far border (the reference raises an IndexError there)
                continue
            reach = int(np.ceil(radius[cr, cc]))
            if occupied[max(cr - reach, 0) : min(cr + reach + 1, rows), max(cc - reach, 0) : min(cc + reach + 1, cols)].any():
                continue
            occupied[cr, cc] = True
            samples.append(cand)
            active.append(cand)
            accepted_any = True
        if not accepted_any:
            del active[pick]
    return len(samples), np.asarray(samples)
