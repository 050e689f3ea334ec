"""Rehearsal of the data-parallel path on the one-GPU box: 2 processes share cuda:0 and reduce over gloo (RCCL refuses
two ranks on one device; the collectives are the same torch.distributed calls).  Result must equal the 1-process step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(device):
    from mentflow_amd.harness import build_problem
    return build_problem(ndim=6, num=25, bins=64, xmax=4.0, seed=2, transforms=5, prior_scale=1.0, device=device,
                         meas_samples=50_000, penalty_parameter=500.0)


def _worker(rank, world, port, z, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from mentflow_amd import _lib, dist as mfdist
    _lib.use_library(_lib.DEFAULT_PATH)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    prob = _problem(dev)
    n = z.shape[0]
    n_local = mfdist.local_batch(n)
    start = sum((n // world + (1 if r < n % world else 0)) for r in range(rank))
    prob.model.generator.inject_z = z[start:start + n_local].to(dev)
    L, H, D = prob.model.loss(n)
    L.backward()
    g = torch.cat([p.grad.reshape(-1) for p in prob.model.parameters()]).cpu()
    out[rank] = (float(L.detach()), float(H.detach()), torch.stack(D).detach().cpu(), g)
    dist.destroy_process_group()


def test_two_processes_on_one_gpu_equal_one():
    from mentflow_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    n = 20_001
    z = torch.randn(n, 6)
    prob = _problem(dev)
    prob.model.generator.inject_z = z.to(dev)
    L, H, D = prob.model.loss(n)
    L.backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in prob.model.parameters()]).cpu()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), z, out), nprocs=2, join=True)
    for r in (0, 1):
        Lr, Hr, Dr, gr = out[r]
        assert abs(Lr - float(L.detach())) < 1e-4 + 500 * 2e-6 and abs(Hr - float(H.detach())) < 2e-5
        torch.testing.assert_close(Dr, torch.stack(D).detach().cpu(), rtol=2e-4, atol=1e-7)
        torch.testing.assert_close(gr, g1, rtol=1e-3, atol=2e-5 * float(g1.abs().max()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][3], out[1][3])
