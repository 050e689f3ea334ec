"""Pick-up tests for the zuko-generated flow fixtures (oracle/gen_flow_golden.py -> tests/golden/ref_flow_*.npz).

While the fixtures are absent (zuko==1.3.1 is not in /root/reference nor in this image) every test here SKIPS with the
reason "flow parity UNPINNED" — the status DESIGN.md §2 records.  The moment the generator has been run where zuko 1.3.1
exists, the same tests pin: (i) the state_dict key / shape list of the generator against the reference's real module
tree (checkpoint compatibility, mentflow/core.py:122-143); (ii) oracle/flow.py against zuko's outputs; (iii) -m gpu:
the HIP kernels against zuko's outputs."""
import json
import os

import numpy as np
import pytest
import torch

import mentflow_amd as mf
from conftest import GOLDEN
from oracle import flow as of
from oracle.harness import flow_spec_from_generator

CASES = ["nsf6", "nsf2", "maf2"]
UNPINNED = ("flow parity UNPINNED: tests/golden/ref_flow_{}.npz is absent — run `python -m oracle.gen_flow_golden` in an "
            "environment that has zuko==1.3.1 (reference pyproject.toml:11; not in /root/reference, not in this image)")


def _load(case):
    path = os.path.join(GOLDEN, f"ref_flow_{case}.npz")
    if not os.path.exists(path):
        pytest.skip(UNPINNED.format(case))
    with np.load(path) as f:
        data = {k: f[k] for k in f.files}
    meta = json.loads(str(data["meta_json"]))
    return data, meta


def _generator(meta, data, tag, device):
    kws = dict(meta["build_kwargs"])
    name = kws.pop("name")
    gen = mf.generate.build_generator(name, device=torch.device("cpu"), **kws)
    # strict load: every key of the reference's state_dict must exist here with the same shape, and vice versa
    sd = {k: torch.from_numpy(np.asarray(data[f"{tag}_sd_{i}"])) for i, k in enumerate(meta["sd_keys"])}
    gen.load_state_dict(sd, strict=True)
    return gen.to(device)


def test_generator_script_refuses_to_fabricate_without_zuko():
    """Always runs: without zuko the generator must write nothing and say so (exit status 3)."""
    import importlib.util
    if importlib.util.find_spec("zuko") is not None:
        pytest.skip("zuko is importable here: run the generator instead")
    from oracle import gen_flow_golden
    before = set(os.listdir(GOLDEN))
    assert gen_flow_golden.main() == 3
    assert set(os.listdir(GOLDEN)) == before


@pytest.mark.parametrize("case", CASES)
def test_state_dict_keys_match_zuko(case):
    data, meta = _load(case)
    kws = dict(meta["build_kwargs"])
    gen = mf.generate.build_generator(kws.pop("name"), device=torch.device("cpu"), **kws)
    ours = gen.state_dict()
    assert list(ours.keys()) == meta["sd_keys"]
    assert [list(v.shape) for v in ours.values()] == meta["sd_shapes"]
    assert [n for n, _ in gen.named_parameters()] == meta["param_names"]


@pytest.mark.parametrize("tag", ["default", "steep"])
@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_zuko(case, tag):
    data, meta = _load(case)
    gen = _generator(meta, data, tag, torch.device("cpu"))
    spec = flow_spec_from_generator(gen, torch.float32)
    # masks and orders: the closed forms must equal zuko's buffers (they came in through the strict load above)
    z = torch.from_numpy(data[f"{tag}_z"])
    x, lp = of.sample_and_log_prob(z, spec)
    tol_lp = 1e-4 if tag == "default" else 5e-4
    assert (x - torch.from_numpy(data[f"{tag}_x"])).abs().max() < 1e-5
    assert (lp - torch.from_numpy(data[f"{tag}_log_prob"])).abs().max() < tol_lp
    steps = of.flow_forward_steps(z, spec)
    assert (torch.stack(steps) - torch.from_numpy(data[f"{tag}_forward_steps"])).abs().max() < 1e-5
    xin = torch.from_numpy(data[f"{tag}_x"])
    assert (of.flow_inverse(xin, spec) - torch.from_numpy(data[f"{tag}_inverse"])).abs().max() < 5e-5
    assert (of.log_prob(xin, spec) - torch.from_numpy(data[f"{tag}_log_prob_of_x"])).abs().max() < 5e-4
    # parameter gradients of the fixed functional
    ps = spec.parameters()
    for p in ps:
        p.requires_grad_(True)
    xo, lo = of.sample_and_log_prob(z, spec)
    ((xo * torch.from_numpy(data[f"{tag}_cx"])).sum() + (lo * torch.from_numpy(data[f"{tag}_cl"])).sum()).backward()
    go = torch.cat([p.grad.reshape(-1) for p in ps])
    gr = torch.cat([torch.from_numpy(data[f"{tag}_grad_{i}"]).reshape(-1) for i in range(len(meta["param_names"]))])
    assert (go - gr).abs().max() < 5e-4 * gr.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["default", "steep"])
@pytest.mark.parametrize("case", CASES)
def test_hip_kernels_match_zuko(case, tag):
    data, meta = _load(case)
    from mentflow_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda", 0)
    gen = _generator(meta, data, tag, dev)
    z = torch.from_numpy(data[f"{tag}_z"]).to(dev)
    x, lp = gen.sample_and_log_prob(z.shape[0], z=z)
    ((x * torch.from_numpy(data[f"{tag}_cx"]).to(dev)).sum() + (lp * torch.from_numpy(data[f"{tag}_cl"]).to(dev)).sum()).backward()
    tol_lp = 1e-4 if tag == "default" else 5e-4
    assert (x.detach().cpu() - torch.from_numpy(data[f"{tag}_x"])).abs().max() < (1e-5 if tag == "default" else 5e-5)
    assert (lp.detach().cpu() - torch.from_numpy(data[f"{tag}_log_prob"])).abs().max() < tol_lp
    g = torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).cpu()
    gr = torch.cat([torch.from_numpy(data[f"{tag}_grad_{i}"]).reshape(-1) for i in range(len(meta["param_names"]))])
    assert (g - gr).abs().max() < (5e-4 if tag == "default" else 2e-3) * gr.abs().max()
    xin = torch.from_numpy(data[f"{tag}_x"]).to(dev)
    assert (gen.inverse(xin).cpu() - torch.from_numpy(data[f"{tag}_inverse"])).abs().max() < 5e-5
    assert (gen.log_prob(xin).cpu() - torch.from_numpy(data[f"{tag}_log_prob_of_x"])).abs().max() < 5e-4
    steps = gen.forward_steps(z)
    assert (torch.stack([s.cpu() for s in steps]) - torch.from_numpy(data[f"{tag}_forward_steps"])).abs().max() < 5e-5
