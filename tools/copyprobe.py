"""which host calls touch DEVICE tensors in one render step (ff_scene.randomize + mi.render): Tensor methods that copy or
synchronise are wrapped and logged with the first fireflies_amd frame of their stack.  GPU box only."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fireflies_amd import mi, workloads

LOG = None


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        if LOG is not None:
            devs = {str(self.device)} | {str(x.device) for x in a if isinstance(x, torch.Tensor)} | {str(v) for kk, v in k.items() if kk == "device"} | {
                str(x) for x in a if isinstance(x, (torch.device, str)) and "cuda" in str(x)}
            if any("cuda" in d for d in devs):
                fr = [fs for fs in traceback.extract_stack()[:-1] if "fireflies_amd" in fs.filename]
                where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"
                LOG[(name, tuple(self.shape), "/".join(sorted(devs)), where)] += 1
        return orig(self, *a, **k)

    setattr(torch.Tensor, name, f)


for n in ("copy_", "to", "cpu", "cuda", "item", "tolist", "numpy", "clone", "fill_", "zero_"):
    wrap(n)

which = sys.argv[1] if len(sys.argv) > 1 else "render"
wl = workloads.vocalfold(grid=8 if which == "grad" else 16)
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
if which == "grad":
    from fireflies_amd.optim import PatternOptimizer

    opt = PatternOptimizer(wl.mi_scene, wl.ff_scene, wl.laser, sigma=wl.sigma, tex_size=wl.tex_size, spp=64, samples_per_step=1, base_seed=7)
    step = lambda i: opt.step()  # noqa: E731
else:
    def step(i):
        wl.ff_scene.randomize()
        mi.render(wl.mi_scene, spp=64, seed=i)
for i in range(5):
    step(i)
torch.cuda.synchronize()
N = 4
LOG = collections.Counter()
for i in range(N):
    step(10 + i)
torch.cuda.synchronize()
log, LOG = LOG, None
for (name, shp, devs, where), c in sorted(log.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c / N:5.1f} per step  {name:8s} {str(shp):14s} {devs:12s} {where}")
