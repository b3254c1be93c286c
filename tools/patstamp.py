#!/usr/bin/env python3
"""patstamp.py — where the one pattern launch of a step (k_pattern_step) spends its time: needs the stamped build
(tools/build_variant_lib.sh patstamp -DFFX_PATSTAMP; FFX_LIB=fireflies_amd/csrc/_stats/libffx_hip_patstamp.so python tools/patstamp.py).
s_memrealtime (100 MHz) at: point 0's workgroup starts [1], the last gradient workgroup has arrived [2], its acquire fence is through [3], the update is
done [4], `go` is published [5], the last helper starts [6], sees the flag [7], has finished its tile [8]."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fireflies_amd import workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402


def main():
    wg = workloads.vocalfold(device="cuda", grid=8)
    opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=64, samples_per_step=1, base_seed=7)
    rows = []
    for k in range(60):
        opt.step()
        torch.cuda.synchronize()
        st = opt._pat_sync[136:136 + 8 * 16].view(torch.int64).cpu().tolist()  # (pad2 starts behind hdr: byte 136)
        if k >= 10 and all(st[i] for i in range(1, 16) if i != 9):
            rows.append([(st[i] - st[1]) / 100.0 for i in range(1, 16)])
    names = ["point 0 starts", "last gradient arrived", "acquire fence done", "update done", "go published", "last helper starts", "last helper sees go", "last helper's tile done",
             "(unused)", "point 1: list + window staged", "point 1: texel loop done", "point 1: gradient stored", "point 1: (k = 0 extras)", "point 1: fence done", "point 1: arrived"]
    print(f"{len(rows)} launches; us after point 0's workgroup started (median / min / max)")
    for i, nm in enumerate(names):
        if nm == "(unused)":
            continue
        col = [r[i] for r in rows]
        print(f"  [{i + 1}] {nm:26s} {statistics.median(col):7.2f} {min(col):7.2f} {max(col):7.2f}")


if __name__ == "__main__":
    main()
