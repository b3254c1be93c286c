#!/usr/bin/env python3
"""Golden vectors for the widened rows f2 / f3 from the REFERENCE's own code (same rules as gen_golden.py: runs
only in the build container; the output under tests/golden/ is data, no reference source text is stored):

  g12_dataset_helpers.npz
     f2  fireflies/postprocessing: WhiteNoise.post_process under np.random.seed (white_noise.py:17-20: the GLOBAL
         numpy generator), the probability gate of BasePostProcessingFunction.apply (base.py:10-14: one
         random.uniform per function) through a PostProcessor chain (postprocessor.py:13-18);
     f3  fireflies/utils/intersections.py rayPlane / sphereSphere.
(ApplySilhouette / GaussianBlur need cv2 / kornia, which are not installed, and fireflies/utils/laser_estimation.py
does not import without Dr.Jit (`dr.wrap_ad` at module level of graphics/depth.py): not capturable.)
"""
import os
import random

import numpy as np
import torch

from gen_golden import OUT, import_reference


def main():
    import_reference()
    import fireflies.postprocessing as PP
    import fireflies.utils.intersections as I

    g = {}
    # ---- f2: white noise values and the gating order of a chain
    rng = np.random.default_rng(3)
    img = rng.random((24, 40)).astype(np.float64)  # the reference adds float64 noise in place
    g["wn_image"] = img.copy()
    np.random.seed(7)
    g["wn_out"] = PP.WhiteNoise(0.02, 0.1, 1.0).post_process(img.copy())
    chain = PP.PostProcessor([PP.WhiteNoise(0.0, 0.05, 0.5), PP.WhiteNoise(0.1, 0.02, 0.5), PP.WhiteNoise(-0.05, 0.2, 0.5)])
    outs, gates = [], []
    random.seed(11)
    np.random.seed(12)
    for k in range(6):
        outs.append(chain.post_process(img))
    g["chain_out"] = np.stack(outs)
    random.seed(11)
    g["chain_gate_draws"] = np.array([random.uniform(0, 1) for _ in range(18)])
    # ---- f3: intersections
    torch.manual_seed(5)
    o, d = torch.randn(16, 3), torch.nn.functional.normalize(torch.randn(16, 3), dim=1)
    d[3] = torch.tensor([1.0, 0.0, 0.0])  # parallel to the plane below: the reference's denom / denom branch
    po, pn = torch.tensor([[0.2, -0.1, 2.0]]), torch.tensor([[0.0, 0.0, -1.0]])
    g["rp_o"], g["rp_d"], g["rp_po"], g["rp_pn"] = o.numpy(), d.numpy(), po.numpy(), pn.numpy()
    g["rp_t"] = I.rayPlane(o, d, po, pn).numpy()
    a, ar, b, br = torch.rand(32, 3), torch.rand(32, 1) * 0.4, torch.rand(32, 3), torch.rand(32, 1) * 0.4
    g["ss_a"], g["ss_ar"], g["ss_b"], g["ss_br"] = a.numpy(), ar.numpy(), b.numpy(), br.numpy()
    g["ss_hit"] = I.sphereSphere(a, ar, b, br).numpy()
    np.savez_compressed(os.path.join(OUT, "g12_dataset_helpers.npz"), **g)
    print("wrote g12_dataset_helpers.npz:", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
