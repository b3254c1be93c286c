"""Round 6: bit-identity soak of the emitters' envelopes — N random poses (random materials, light intensities, animation frames) of a workload,
each rendered with FFX_ENVELOPE=3 and =0 (the pose prepared again in between) at two sample seeds: every image must be the same bit for bit, forward
and forward + adjoint.      python tools/envsoak.py [vocalfold|colon] [poses] [spp]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    poses = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else (64 if which == "vocalfold" else 256)
    wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", entity_device="cpu")
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    os.environ["FFX_RENDER_STREAMS"] = "1"
    bad = 0
    for p in range(poses):
        imgs = {}
        for env in ("3", "0"):
            os.environ["FFX_ENVELOPE"] = env
            wl.mi_scene._sd_cache = None
            torch.manual_seed(7000 + p)
            random.seed(7000 + p)
            wl.ff_scene.randomize()
            imgs[env] = [mi.render(wl.mi_scene, spp=spp, seed=11 * p + k, fp16=which == "colon").torch().clone() for k in range(2)]
        same = all(torch.equal(a, b) for a, b in zip(imgs["3"], imgs["0"]))
        if not same:
            bad += 1
            print(f"pose {p}: images differ, worst {max(float((a.float() - b.float()).abs().max()) for a, b in zip(imgs['3'], imgs['0'])):.3e}")
    print(f"{which}: {poses} poses x 2 seeds x {spp} spp with and without the envelopes: {poses - bad} identical bit for bit, {bad} differing")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
