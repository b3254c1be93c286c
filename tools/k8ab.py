#!/usr/bin/env python3
"""A/B of K8 builds on IDENTICAL work: the same eight randomised poses of the default workload (vocal fold, 512x512, 64 spp,
principled material), each rendered `reps` times back to back with nothing else on the GPU; per build the mean over the poses
of the per-pose median of ffx_render_fwd's duration (apex pre-pass + K8, HIP events on the launch stream).  Resolves ~0.3 %,
which the bench loop (random poses, refits sharing the GPU) does not.

    python tools/k8ab.py [name=path/to/libffx_hip_variant.so ...]        (the in-tree library is always measured as "tree")
    FFX_K8AB_ARGS="--workload colon --res 1024 --spp 256" ...             (workload of the child processes)
Child mode (one library per process, FFX_LIB decides which):  python tools/k8ab.py --child
"""
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import argparse

    import torch

    from fireflies_amd import mi, workloads

    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--workload", default="vocalfold")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--grid", type=int, default=16)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--material", default="principled")
    ap.add_argument("--poses", type=int, default=8)
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    make = workloads.vocalfold if a.workload == "vocalfold" else workloads.colon
    wl = make(device="cuda", width=a.res, height=a.res, grid=a.grid, entity_device="cpu", principled=a.material == "principled")
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    geom = wl.mi_scene.geom
    per_pose = []
    for p in range(a.poses):
        torch.manual_seed(100 + p)
        random.seed(100 + p)
        wl.ff_scene.randomize()
        for _ in range(3):
            mi.render(wl.mi_scene, spp=a.spp, seed=p, fp16=a.fp16)
        torch.cuda.synchronize()
        ev = []
        geom.timing = ev
        for _ in range(a.reps):
            mi.render(wl.mi_scene, spp=a.spp, seed=p, fp16=a.fp16)
        geom.timing = None
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for n, s, e in ev if n == "render_fwd")
        per_pose.append(ms[len(ms) // 2])
    print(json.dumps({"ms": sum(per_pose) / len(per_pose), "per_pose": per_pose}))


def main():
    if "--child" in sys.argv:
        return child()
    builds = [("tree", os.path.join(ROOT, "fireflies_amd", "csrc", "libffx_hip.so"))]
    for spec in sys.argv[1:]:
        name, path = spec.split("=", 1)
        builds.append((name, os.path.abspath(path)))
    extra = os.environ.get("FFX_K8AB_ARGS", "").split()
    base = None
    for rnd in range(2):  # two interleaved rounds: drift of the clocks shows up as a difference between them
        for name, path in builds:
            env = dict(os.environ, FFX_LIB=path, FFX_RENDER_STREAMS="1")  # (one stream: back-to-back launches of ONE kernel at a time — on two they overlap and each reads twice as long)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + extra, env=env, capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                print(f"{name:24s} FAILED\n{r.stderr[-1500:]}")
                continue
            if base is None:
                base = d["ms"]
            print(f"round {rnd} {name:24s} {d['ms']:.4f} ms  ({100.0 * (d['ms'] / base - 1.0):+.2f} % vs the first)   poses: " + " ".join(f"{v:.3f}" for v in d["per_pose"]), flush=True)


if __name__ == "__main__":
    main()
