#!/usr/bin/env python3
"""patbench.py — the pattern side of a one-sample gradient step ALONE on the GPU: ffx_pattern_bwd_blur (K3^T + K2-bwd + K1-bwd + Adam + clamp) and
ffx_pattern_fwd_blur (K1 + K2 + K3) on the bench's 64-point pattern, each variant timed over back-to-back launches with HIP events.

    python tools/patbench.py [launches]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fireflies_amd import ops, workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    wg = workloads.vocalfold(device="cuda", grid=8)
    opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=64, samples_per_step=1, base_seed=7)
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    rd = wg.laser._rays.detach()
    pts, tsum, tsor, ws, tex = opt._pat_buf
    s0, s1 = opt.tex_size
    gtex = opt._acc[: s0 * s1].view(tex.shape).clone()
    st, g = opt._adam_state(wg.laser._rays)
    grad = torch.empty_like(rd)
    bk, bs = opt.blur
    KF = opt.laser._KF
    print(f"points {rd.shape[0]}, texture {s0}x{s1}, sigma {opt.sigma}, blur {opt.blur}, reg {opt.reg_weight}", flush=True)

    def bwd(reg=True, adam=True, dot=True, data=True, noupd=False):
        d = (opt._img_stack, opt._lin_g, opt._dot_part) if dot else None
        aa = None
        if adam:
            if noupd:
                aa = ops.adam_args(rd, None, None, None, opt._adam_counter, 0.0, 0.0, 0.0, 0.0, opt.laser._KF_inv, 0.0, 1.0, dot=d)
            else:
                aa = ops.adam_args(rd, st["exp_avg"], st["exp_avg_sq"], st["step"], opt._adam_counter, 0.0, g["betas"][0], g["betas"][1], g["eps"], opt.laser._KF_inv,
                                   1 - 0.95, 0.95, 2, grad_div=1.0, grad_out=grad, dot=d)
        return lambda: ops.pattern_bwd_blur(rd, KF, opt.sigma, s0, s1, tsum, tsor, gtex if data else None, opt.reg_weight if reg else 0.0, ws, bk, bs,
                                            loss_in=None if (dot and adam) else opt._acc[s0 * s1: s0 * s1 + 1], loss_div=1.0, adam=aa, scratch=opt._scratch)

    sync = torch.zeros(35840, dtype=torch.uint8, device="cuda")
    kept = torch.zeros((2,) + tuple(rd.shape), device="cuda")
    acc2 = torch.zeros_like(opt._acc)
    g2 = acc2[: s0 * s1].view(tex.shape)

    def merged(dot=True):
        d = (opt._img_stack, opt._lin_g, opt._dot_part) if dot else None
        aa = ops.adam_args(rd, st["exp_avg"], st["exp_avg_sq"], st["step"], opt._adam_counter, 0.0, g["betas"][0], g["betas"][1], g["eps"], opt.laser._KF_inv,
                           1 - 0.95, 0.95, 2, grad_div=1.0, grad_out=grad, dot=d)

        def fn():
            g2.copy_(gtex)  # (the launch clears its accumulator: put the gradient back — its own launch, not part of the figure under rocprof)
            assert ops.pattern_step(rd, KF, opt.sigma, s0, s1, opt._pat_buf, g2, opt.reg_weight, bk, bs, aa, acc2, sync, rays_kept=kept, check_kept=False,
                                    loss_in=None if dot else acc2[s0 * s1: s0 * s1 + 1], loss_div=1.0) is not None
        return fn

    rows = [("step (one launch)", merged()), ("step (one launch, slots)", merged(dot=False)), ("bwd full (lr 0)", bwd()), ("bwd no dot", bwd(dot=False)), ("bwd no reg", bwd(reg=False)), ("bwd no adam (no dot)", bwd(adam=False, dot=False)),
            ("bwd dot, no update", bwd(noupd=True)), ("bwd no data", bwd(data=False)), ("bwd reg only, no adam", bwd(adam=False, dot=False, data=False)),
            ("fwd_blur", lambda: ops.pattern_fwd_blur(rd, KF, opt.sigma, s0, s1, bk, bs, want_softor=True, out=opt._pat_buf, zero=opt._acc)),
            ("fwd_blur no zero", lambda: ops.pattern_fwd_blur(rd, KF, opt.sigma, s0, s1, bk, bs, want_softor=True, out=opt._pat_buf)),
            ("empty torch op", lambda: grad.add_(0.0))]
    for name, fn in rows:
        print(f"{name:28s} {timed(fn, n):8.2f} us / launch (back to back)", flush=True)


if __name__ == "__main__":
    main()
