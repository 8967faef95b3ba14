"""View data-parallel layer on CPU: world_size=2 `gloo` processes, one view per rank, the product's
GradBucket / statistics reductions, with the torch oracle standing in for the GPU rasterizer.
Checks  mean_rank(grads) == grads of the single-process C=2 batch with a mean-over-views loss."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    from scenes import make_scene
    return make_scene(80, 36, 28, sh_degree=1, n_views=2, seed=21, scale_range=(0.05, 0.4), dist=4.0)


def _params(sc):
    dt = torch.float64
    return [torch.nn.Parameter(torch.tensor(sc[k], dtype=dt)) for k in ("means", "quats", "scales", "opacities", "shs")]


def _view_loss(params, sc, views, target):
    from oracle import torch_oracle as TO
    dt = torch.float64
    V = torch.tensor(sc["viewmats"][views], dtype=dt); K = torch.tensor(sc["Ks"][views], dtype=dt)
    bg = torch.tensor(sc["backgrounds"][views], dtype=dt)
    img, _, meta = TO.rasterization(*params, V, K, sc["width"], sc["height"], sh_degree=1, packed=False,
                                    backgrounds=bg, absgrad=True)
    return ((img - target[views]) ** 2).mean(dim=(1, 2, 3)).sum(), meta


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from easy_gaussian_splatting_amd.distributed import GradBucket, all_reduce_param_grads, all_reduce_statistics, shard_views
    sc = _scene()
    params = _params(sc)
    bucket = GradBucket(params)
    target = torch.rand((2, sc["height"], sc["width"], 3), generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    views = shard_views(2)
    assert views == [rank]
    loss, meta = _view_loss(params, sc, views, target)
    loss.backward()
    assert params[0].grad.data_ptr() == bucket.flat.data_ptr()  # autograd accumulated into the bucket
    loose = [torch.nn.Parameter(p.detach().clone()) for p in params]   # bucket-free variant
    for q, p in zip(loose, params):
        q.grad = p.grad.detach().clone()
    bucket.all_reduce_mean()
    all_reduce_param_grads(loose)
    for q, p in zip(loose, params):
        assert torch.equal(q.grad, p.grad)
    radii = meta["radii"][0].double() / max(sc["width"], sc["height"])
    vis = radii > 0
    g = torch.where(vis, meta["means2d"].absgrad[0].norm(dim=-1), torch.zeros_like(radii))
    cnt = vis.double(); rad = torch.where(vis, radii, torch.zeros_like(radii))
    all_reduce_statistics(g, cnt, rad)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), flat=bucket.flat.numpy(), g=g.numpy(), cnt=cnt.numpy(), rad=rad.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_view_dp_equals_two_camera_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, "dp.npz"))
    sc = _scene()
    params = _params(sc)
    target = torch.rand((2, sc["height"], sc["width"], 3), generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    loss, meta = _view_loss(params, sc, [0, 1], target)
    (loss / 2).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in params]).numpy()
    np.testing.assert_allclose(got["flat"], ref, atol=1e-12 * max(1.0, np.abs(ref).max()))
    radii = meta["radii"].double() / max(sc["width"], sc["height"])
    vis = radii > 0
    g = torch.where(vis, meta["means2d"].absgrad.norm(dim=-1), torch.zeros_like(radii))
    # each rank back-propagated its own un-divided view loss; the batch above used loss/2
    np.testing.assert_allclose(got["g"], 2.0 * g.sum(0).numpy(), atol=1e-12)
    np.testing.assert_allclose(got["cnt"], vis.double().sum(0).numpy())
    np.testing.assert_allclose(got["rad"], torch.where(vis, radii, torch.zeros_like(radii)).max(0).values.numpy())


def test_single_process_helpers_are_noops():
    sys.path.insert(0, ROOT)
    from easy_gaussian_splatting_amd.distributed import GradBucket, is_distributed, shard_views
    assert not is_distributed()
    p = [torch.nn.Parameter(torch.ones(3, 2)), torch.nn.Parameter(torch.ones(5))]
    b = GradBucket(p)
    (p[0].sum() * 2 + p[1].sum() * 3).backward()
    assert b.flat.tolist() == [2.0] * 6 + [3.0] * 5
    assert b.all_reduce_mean() is None
    b.zero_()
    assert p[1].grad.abs().sum() == 0
    assert shard_views(8, rank=1, world=4) == [1, 5]
