"""worker of tests/test_api_gpu.py::test_two_rank_rccl_step_matches_one_rank — launched by torch.distributed.run,
one process per GPU; writes its result to <outdir>/rank<r>.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fireflies_amd import dist, workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402


def main():
    out = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "nccl"  # "gloo": rehearsal of the same step with both ranks on ONE device
    if backend != "nccl":
        torch.cuda.set_device(0)
    rank, world, local = dist.init(backend)
    dev = torch.cuda.current_device()
    wl = workloads.vocalfold(device="cuda", width=64, height=56, tex=96, grid=6, frames=5, n_fold=20, tube=(20, 24))
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 4  # scene samples per step over all ranks
    opt = PatternOptimizer(wl.mi_scene, wl.ff_scene, wl.laser, sigma=10.0, tex_size=(96, 96), spp=4, lr=5e-3, reg_weight=0.1, base_seed=21, samples_per_step=S)
    res = opt.step()
    import torch.distributed as td

    with open(os.path.join(out, f"rank{rank}.json"), "w") as f:
        json.dump({"world": td.get_world_size(), "backend": td.get_backend(), "device": dev, "grad": wl.laser._rays.grad.detach().cpu().tolist(),
                   "loss": float(res["loss"]), "rays": wl.laser._rays.detach().cpu().tolist()}, f)
    dist.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
