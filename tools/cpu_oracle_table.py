#!/usr/bin/env python3
"""CPU-oracle timings per kernel beside the HIP numbers: all host threads and one thread,
on bounded samples of the vocal-fold workload (scaled to 512x512 x 64 spp where noted).
Run on the GPU box's host:  python tools/cpu_oracle_table.py > profiles/<tag>_cpu_oracle_table.json
The oracle is test infrastructure; this script only times it."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    from fireflies_amd import scene_desc, scenes
    from oracle import oracle as orc

    rng = np.random.default_rng(0)
    sc = scenes.vocalfold()
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    geo = orc.Geometry(pool, tris, shape, off)
    cam = scene_desc.camera_from_sensor(sc.camera)
    rows = scenes.material_rows(sc)  # the scene's principled material (bench.py's default workload); None: diffuse
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=0 if rows is None else scenes.MAT_STRIDE)
    alb = alb if rows is None else rows
    one = int(os.environ.get("OMP_NUM_THREADS", "0")) == 1
    spp = 1 if one else 16  # bounded sample; scaled to 64 spp below
    out = {}

    def timed(name, fn, scale=1.0, note=""):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        out[name] = {"ms": round(1e3 * dt * scale, 3), "measured_ms": round(1e3 * dt, 3), "scaled_by": scale, "note": note}

    pts = (rng.random((256, 2)) * 0.9 + 0.05).astype(np.float32)
    timed("K2 splat_fwd sum N=256 500x500", lambda: orc.splat_fwd(pts, 10.0, 0, -1, 500, 500))
    tex = orc.splat_fwd(pts, 10.0, 0, -1, 500, 500)
    g = rng.random((500, 500)).astype(np.float32)
    timed("K2 splat_bwd sum N=256 500x500", lambda: orc.splat_bwd(pts, 10.0, 0, -1, 500, 500, tex, g))
    timed("K3 blur_fwd 500x500", lambda: orc.blur_fwd(tex))
    xf = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    timed("K5+K6 scene_update 53248 tris", lambda: geo.update(xf))
    timed("K7 trace_primary 512x512 x 64 spp", lambda: geo.trace_primary(cam, spp, 1, 3), 64.0 / spp, f"measured at {spp} spp")
    t3 = orc.blur_fwd(tex)[..., None]
    timed("K8 render_fwd 512x512 x 64 spp shadows, principled material", lambda: geo.render_fwd(sd, alb, t3, spp, seed=1), 64.0 / spp, f"measured at {spp} spp")
    gimg = rng.random((512, 512, 3)).astype(np.float32)
    timed("K9 render_bwd (re-trace) 512x512 x 64 spp, principled material", lambda: geo.render_bwd(sd, alb, spp, 1, gimg), 64.0 / spp, f"measured at {spp} spp")
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
        return
    res = {}
    for label, threads in (("all_threads", str(os.cpu_count())), ("one_thread", "1")):
        env = dict(os.environ, OMP_NUM_THREADS=threads)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True, cwd=ROOT)
        if p.returncode != 0:
            raise SystemExit(p.stderr[-2000:])
        res[label] = {"threads": int(threads), "kernels": json.loads(p.stdout.strip().splitlines()[-1])}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
