"""The N > 1 path on CPU: 2 ranks over gloo, each running the real Python data-parallel code (mentflow_amd.dist +
MENTFlow.loss) on the host-emulated kernels, must reproduce the single-process step on the concatenated batch:
same (L, H, D) on every rank and the same parameter gradients after the backward all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import EMU_LIB


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(device, gen_name="nsf"):
    from mentflow_amd.harness import build_problem
    kws = dict(gen_name="nn", discrepancy="mae", hidden_layers=2, hidden_units=16) if gen_name == "nn" else {}
    return build_problem(ndim=6, num=5, bins=16, xmax=4.0, seed=2, transforms=2, prior_scale=1.0, device=device,
                         meas_samples=4000, penalty_parameter=50.0, **kws)


def _worker(rank, world, port, z, out, gen_name="nsf"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from mentflow_amd import _lib, dist as mfdist
    _lib.use_library(EMU_LIB)
    dev = mfdist.init_from_env(backend="gloo")
    assert mfdist.world_size() == world and mfdist.rank() == rank
    prob = _problem(dev, gen_name)
    n = z.shape[0]
    n_local = mfdist.local_batch(n)
    start = sum((n // world + (1 if r < n % world else 0)) for r in range(rank))
    prob.model.generator.inject_z = z[start:start + n_local].clone()
    L, H, D = prob.model.loss(n)
    L.backward()
    g = torch.cat([p.grad.reshape(-1) for p in prob.model.parameters()])
    out[rank] = (float(L.detach()), float(H.detach()), torch.stack(D).detach(), g)
    dist.destroy_process_group()


def test_two_ranks_equal_one(emu_library):
    from mentflow_amd import _lib
    _lib.use_library(emu_library)
    torch.manual_seed(7)
    n = 101                                       # odd on purpose: ranks get 51 and 50 particles
    z = torch.randn(n, 6)
    prob = _problem(torch.device("cpu"))
    prob.model.generator.inject_z = z
    L, H, D = prob.model.loss(n)
    L.backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in prob.model.parameters()])

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), z, out), nprocs=2, join=True)
    assert set(out.keys()) == {0, 1}
    for r in (0, 1):
        Lr, Hr, Dr, gr = out[r]
        assert abs(Lr - float(L)) < 1e-5 + 50 * 1e-6 and abs(Hr - float(H)) < 1e-5
        torch.testing.assert_close(Dr, torch.stack(D).detach(), rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(gr, g1, rtol=1e-4, atol=1e-6 * float(g1.abs().max()))
    # both ranks hold bitwise the same reduced values
    assert out[0][0] == out[1][0] and torch.equal(out[0][3], out[1][3])


def test_two_ranks_generic_generator_gradients_are_reduced(emu_library):
    """a generator without the flow's flat-gradient hook (the NN baseline) must still end up with the SUMMED
    parameter gradients on every rank — the forward all-reduce's identity adjoint relies on it."""
    from mentflow_amd import _lib
    _lib.use_library(emu_library)
    torch.manual_seed(9)
    n = 64
    z = torch.randn(n, 6)
    prob = _problem(torch.device("cpu"), "nn")
    prob.model.generator.inject_z = z
    L, H, D = prob.model.loss(n)
    L.backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in prob.model.parameters()])
    assert float(g1.abs().max()) > 0

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), z, out, "nn"), nprocs=2, join=True)
    for r in (0, 1):
        Lr, Hr, Dr, gr = out[r]
        assert abs(Lr - float(L.detach())) < 1e-5 + 50 * 1e-6
        torch.testing.assert_close(gr, g1, rtol=1e-4, atol=1e-6 * float(g1.abs().max()))
    assert torch.equal(out[0][3], out[1][3])


def test_local_batch_split():
    from mentflow_amd import dist as mfdist
    assert mfdist.world_size() == 1 and mfdist.rank() == 0 and mfdist.local_batch(17) == 17
