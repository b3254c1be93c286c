"""The dataset path of the reference's main.py:138-175 as a loop on the device: per sample
`ff_scene.randomize()` -> `mi.render(scene, spp)` -> grey image -> post-processing chain (blur p=0.5, silhouette,
white noise p=0.5) -> segmentation + depth map of the same pose.  Nothing leaves the GPU inside the loop (the
reference copies the render to the host and runs kornia / cv2 / numpy there).  Prints samples/s.
    python tools/datasetbench.py [n_samples] [spp | mix]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fireflies_amd as ff  # noqa: E402
import fireflies_amd.postprocessing as pp  # noqa: E402
from fireflies_amd import mi, workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
spp = sys.argv[2] if len(sys.argv) > 2 else "64"
MIX = spp == "mix"  # main.py:145 draws every sample's spp: fireflies.sampling.AnimationSampler(1, 100, 1, 100) — here uniform in 1..100 from python's `random`
spp = 64 if MIX else int(spp)
wl = workloads.vocalfold(device="cuda")
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
chain = pp.PostProcessor([pp.GaussianBlur((3, 3), (5, 5), 0.5), pp.ApplySilhouette(), pp.WhiteNoise(0.0, 0.05, 0.5, rng="device")])
torch.manual_seed(0)
random.seed(0)


def sample(i):
    wl.ff_scene.randomize()
    img = mi.render(wl.mi_scene, spp=random.randint(1, 100) if MIX else spp, seed=i).torch()
    grey = pp.rgb_to_gray(img)  # cv2.COLOR_RGB2GRAY's weights, one launch (as a torch expression: five)
    out = chain.post_process(grey)
    seg = ff.graphics.depth.get_segmentation_from_camera(wl.mi_scene)
    depth = ff.graphics.depth.from_camera_non_wrapped(wl.mi_scene, spp=1)
    return out, seg, depth


for i in range(10):
    sample(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    keep = sample(10 + i)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"host issued the {n} samples in {1e3 * t_issue / n:.3f} ms each; the device finished {1e3 * (dt - t_issue):.2f} ms after the last was issued")
print(f"dataset path: {n / dt:.1f} samples/s ({1e3 * dt / n:.3f} ms per sample: render {'1..100 (drawn per sample)' if MIX else spp} spp + post-processing + segmentation + depth, 512x512), "
      f"outputs {tuple(keep[0].shape)} {tuple(keep[1].shape)} {tuple(keep[2].shape)}")

if os.environ.get("FFX_DS_PROFILE") == "1":  # where the host's time per sample goes (cProfile, 200 samples at 1 spp so that the device is not the limit)
    import cProfile
    import pstats

    spp = 1
    pr = cProfile.Profile()
    pr.enable()
    for i in range(200):
        sample(i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
