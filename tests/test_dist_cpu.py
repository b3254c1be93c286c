"""Multi-process (gloo, world_size 2) tests of the data-parallel layer: sample sharding, seeds that
do not depend on the world size, and the single all-reduce of the flat [3N+1] buffer.  The
per-sample work is injected: a closed-form stand-in and the CPU oracle (test infrastructure)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from fireflies_amd import dist


def test_sharding_and_seeds_are_world_size_invariant():
    S = 32
    for world in (1, 2, 4, 8):
        ids = [dist.sample_ids(S, r, world) for r in range(world)]
        assert sorted(sum(ids, [])) == list(range(S))
        assert all(len(i) == S // world for i in ids)
    assert dist.sample_seed(100, 3, 32, 5) == 100 + 3 * 32 + 5  # cfg4: base + step*32 + k
    assert len({dist.sample_seed(0, s, 32, k) for s in range(4) for k in range(32)}) == 128


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _analytic_sample(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(12, generator=g), torch.rand((), generator=g)


def _oracle_sample_factory():
    from fireflies_amd import scene_desc, scenes
    from oracle import oracle as orc

    sc = scenes.vocalfold(width=24, height=24, tex=32, frames=3, n_fold=8, tube=(12, 12))
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    geo = orc.Geometry(pool, tris, shape, off)
    sd = scene_desc.scene_desc(sc, shadows=False)
    pts = np.stack(np.meshgrid(np.linspace(0.2, 0.8, 3), np.linspace(0.2, 0.8, 3)), -1).reshape(-1, 2).astype(np.float32)

    def sample(seed):
        rng = np.random.default_rng(seed)
        xf = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
        xf[1, 0, 0] = rng.uniform(0.5, 2.0)
        geo.update(xf, off + np.array([0, int(rng.integers(0, 3)) * stride[1]], np.int32))
        tsum = orc.splat_fwd(pts, 10.0, 0, -1, 32, 32)
        img = geo.render_fwd(sd, alb, orc.blur_fwd(tsum), 2, seed=seed)
        gimg = np.zeros_like(img)
        gimg[..., 1] = -1.0 / (24 * 24)
        gtex = geo.render_bwd(sd, alb, 2, seed, gimg)[..., 0]
        gp = orc.splat_bwd(pts, 10.0, 0, -1, 32, 32, tsum, orc.blur_bwd(gtex))
        return torch.from_numpy(gp.reshape(-1)), float(-img[..., 1].mean())

    return sample, pts.size


def _worker(rank, world, port, kind, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init(backend="gloo")
    if kind == "analytic":
        fn, n = _analytic_sample, 12
    else:
        fn, n = _oracle_sample_factory()
    res = [dist.accumulate_step(fn, n, step, 4, base_seed=50) for step in range(2)]
    torch.save(torch.stack(res), os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("kind", ["analytic", "oracle"])
def test_two_ranks_match_one_process(tmp_path, kind):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    torch.testing.assert_close(r0, r1, rtol=0, atol=0)  # every rank holds the reduced buffer
    # single process reference (no process group)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    fn, n = (_analytic_sample, 12) if kind == "analytic" else _oracle_sample_factory()
    ref = torch.stack([dist.accumulate_step(fn, n, step, 4, base_seed=50) for step in range(2)])
    torch.testing.assert_close(r0, ref, rtol=1e-5, atol=1e-7)  # same samples, different summation order
    assert float(ref.abs().sum()) > 0
