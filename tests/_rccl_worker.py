"""worker of tests/test_api_gpu.py::test_two_rank_rccl_step_matches_one_rank (and of the one-rank nccl / poisoned-update tests) — launched by
torch.distributed.run (or directly, with RANK / WORLD_SIZE in the environment), one process per GPU; writes its result to <outdir>/rank<r>.json.

    _rccl_worker.py <outdir> [nccl|gloo] [samples per step] [coverage|l1] [steps]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fireflies_amd import dist, mi, workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer, image_l1_loss  # noqa: E402


def main():
    out = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "nccl"  # "gloo": rehearsal of the same step with both ranks on ONE device
    if backend != "nccl":
        torch.cuda.set_device(0)
    rank, world, local = dist.init(backend)
    dev = torch.cuda.current_device()
    wl = workloads.vocalfold(device="cuda", width=64, height=56, tex=96, grid=6, frames=5, n_fold=20, tube=(20, 24))
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 4  # scene samples per step over all ranks
    loss_kind = sys.argv[4] if len(sys.argv) > 4 else "coverage"
    kw = {}
    if loss_kind == "l1":  # a loss that is not linear in the image: cache-writing forward + K9 (the path whose cache can drop samples)
        with torch.no_grad():
            kw["loss_fn"] = image_l1_loss(mi.render(wl.mi_scene, spp=4, seed=99).torch().clone())
    opt = PatternOptimizer(wl.mi_scene, wl.ff_scene, wl.laser, sigma=10.0, tex_size=(96, 96), spp=4, lr=5e-3, reg_weight=0.1, base_seed=21, samples_per_step=S, **kw)
    rays_before = wl.laser._rays.detach().cpu().tolist()
    force = os.environ.get("FFX_TEST_FORCE_DROPPED_RANK")
    if force is not None and int(force) == rank:
        # what an overflowing arena leaves behind: `dropped` != 0 in this rank's cache header at the end of the step.  The step's samples keep the
        # count (FFX_RENDER_CACHE_KEEP_DROPPED); the pattern launch that opens the step clears it — so it is planted behind that launch
        from fireflies_amd import ops

        real = ops.pattern_fwd_blur

        def planted(*a, **k):
            r = real(*a, **k)
            opt._cache[8:12].view(torch.int32).fill_(5)
            return r

        ops.pattern_fwd_blur = planted
    for _ in range(int(sys.argv[5]) if len(sys.argv) > 5 else 1):
        res = opt.step()
    import torch.distributed as td

    st = opt.opt.state[wl.laser._rays]
    flat = getattr(opt, "_last_flat", None)
    with open(os.path.join(out, f"rank{rank}.json"), "w") as f:
        json.dump({"world": dist.world_size(), "backend": td.get_backend() if td.is_initialized() else "none", "device": dev, "grad": wl.laser._rays.grad.detach().cpu().tolist(),
                   "loss": float(res["loss"]), "rays": wl.laser._rays.detach().cpu().tolist(), "rays_before": rays_before,
                   "exchanged": flat is not None, "flat_len": int(flat.numel()) if flat is not None else 0,
                   "exchanged_dropped": float(flat[-1]) if flat is not None else None,
                   "adam_step": float(st["step"]) if len(st) else 0.0, "exp_avg_max": float(st["exp_avg"].abs().max()) if len(st) else 0.0}, f)
    dist.barrier()
    if td.is_initialized():  # (one rank without FFX_DIST_FORCE: no group was formed)
        td.destroy_process_group()


if __name__ == "__main__":
    main()
