"""Data parallelism over independent scene-randomisation samples.

The reference is a serial loop (examples/vocalfold_scene.py:100-102); this layer is new.
One process per GPU.  A step of S samples is split as {k : k mod world == rank}; every sample's
RNG seed depends only on (base_seed, step, k) — never on the world size — so 1/2/4/8-GPU runs
sum the same set of per-sample gradients.  The only exchange is ONE all-reduce(sum) of a flat
fp32 buffer [3N + 1] (pattern gradient + loss; 0.8-12 KB, latency-bound), RCCL over xGMI on GPUs
(`backend="nccl"` is RCCL on ROCm), gloo in the CPU tests.
"""
import os

import torch
import torch.distributed as td


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def force_exchange():
    """FFX_DIST_FORCE=1: form the process group and run the step's exchange even with ONE rank — the way to take RCCL's load, the
    communicator's device binding, the nccl barrier and the [3N+2] all-reduce through a real run on a box with a single GPU (round-4
    review: no RCCL rank had ever run; tests/test_api_gpu.py::test_one_rank_nccl_group_runs_the_multi_rank_step)"""
    return os.environ.get("FFX_DIST_FORCE", "0") == "1"


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process unless FFX_DIST_FORCE=1)."""
    rank, world, local = env_rank_world()
    if (world > 1 or force_exchange()) and not td.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL on this driver
        kw = {}
        if backend == "nccl":
            # one process per GPU: bind the communicator to this rank's device up front
            n = max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local % n)
            kw["device_id"] = torch.device("cuda", local % n)
        try:
            td.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:  # torch without the device_id argument
            td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return td.get_world_size() if td.is_initialized() else 1


def rank():
    return td.get_rank() if td.is_initialized() else 0


def barrier():
    if td.is_initialized():
        if td.get_backend() == "nccl":
            td.barrier(device_ids=[torch.cuda.current_device()])
        else:
            td.barrier()


def sample_ids(n_samples: int, rank_: int, world: int):
    """sample indices of this rank for one step: k = rank, rank + world, ..."""
    return list(range(rank_, n_samples, world))


def sample_seed(base_seed: int, step: int, n_samples: int, k: int) -> int:
    """seed of sample k of `step`; independent of the world size (cfg4: base + step*32 + k)."""
    return (int(base_seed) + int(step) * int(n_samples) + int(k)) & 0x7FFFFFFF


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """in-place sum over ranks of a flat buffer (one collective per optimisation step)."""
    if td.is_initialized() and (td.get_world_size() > 1 or force_exchange()):
        td.all_reduce(flat, op=td.ReduceOp.SUM)
    return flat


def allreduce_max_(flat: torch.Tensor) -> torch.Tensor:
    """in-place maximum over ranks (integers: the bits of non-negative floats order as they do)"""
    if td.is_initialized() and (td.get_world_size() > 1 or force_exchange()):
        td.all_reduce(flat, op=td.ReduceOp.MAX)
    return flat


def exchanging():
    """whether an optimisation step exchanges its gradient (several ranks, or one rank rehearsing the exchange)"""
    return td.is_initialized() and (td.get_world_size() > 1 or force_exchange())


def accumulate_step(sample_fn, n_values: int, step: int, n_samples: int, base_seed: int = 0, device="cpu"):
    """Runs this rank's samples of one step and returns the all-reduced flat buffer
    [n_values + 1] = (sum of per-sample gradients, sum of per-sample losses), both divided by
    n_samples.  `sample_fn(seed) -> (grad [n_values], loss scalar)` is the per-sample work
    (randomise + render + adjoint); it never communicates."""
    r, w = rank(), world_size()
    flat = torch.zeros(n_values + 1, dtype=torch.float32, device=device)
    for k in sample_ids(n_samples, r, w):
        g, loss = sample_fn(sample_seed(base_seed, step, n_samples, k))
        flat[:n_values] += g.reshape(-1).to(flat.dtype)
        flat[n_values] += float(loss) if not isinstance(loss, torch.Tensor) else loss.detach().to(flat.dtype).reshape(())
    allreduce_sum_(flat)
    flat /= float(n_samples)
    return flat
