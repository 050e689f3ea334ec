"""ORACLE (test infrastructure) — ONE-COMMAND PIN of the flow arithmetic against the real third-party library.

    python -m oracle.gen_flow_golden            # needs zuko==1.3.1 importable AND /root/reference present

The flow arithmetic of MENT-Flow lives in ``zuko==1.3.1`` (reference ``pyproject.toml:11``), which is neither under
/root/reference nor installed in this image, so ``oracle/flow.py`` is "parity unpinned" (see its header).  This script
closes that the moment an environment WITH zuko 1.3.1 exists (a build container — never the GPU box): it builds the
flows exactly as the reference does — ``mentflow/generate/build.py:13-46`` (``build_flow``: ``zuko.flows.NSF/MAF(features=,
hidden_features=, transforms=, bins=)`` then ``zuko.flows.Flow(flow.transform.inv, flow.base)``) wrapped in
``mentflow/generate/flows/zuko.py:10-53`` (``WrappedZukoFlow``) — with fixed seeds, and writes DATA ONLY to
``tests/golden/ref_flow_{nsf6,nsf2,maf2}.npz``:

    state_dict keys / shapes / values (the checkpoint-compatibility contract of mentflow/core.py:122-143),
    z (base draw), x = forward(z), log_prob (rsample_and_log_prob arithmetic with z injected), forward_steps,
    inverse(x), inverse_steps, log_prob(x) through the inverse path, and the parameter gradients of a fixed
    linear functional of (x, log_prob), for the default initialisation and for a "steep" re-scaled copy.

``tests/test_flow_golden_fixture.py`` picks the files up (CPU: oracle/flow.py and the state_dict key list against
them; ``-m gpu``: the HIP kernels against them) and SKIPS loudly ("flow parity UNPINNED") while they are absent.
Without zuko this script exits with status 3 and writes nothing — it never fabricates a fixture.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
ZUKO_VERSION = "1.3.1"

# name -> build_flow kwargs (experiments/setup.py:115-124 + config/gen/flow.yaml: 3 x 64 hidden, 5 transforms, bins 20)
CASES = {
    "nsf6": dict(name="nsf", input_features=6, output_features=6, hidden_layers=3, hidden_units=64, transforms=5, bins=20),
    "nsf2": dict(name="nsf", input_features=2, output_features=2, hidden_layers=3, hidden_units=64, transforms=5, bins=20),
    "maf2": dict(name="maf", input_features=2, output_features=2, hidden_layers=3, hidden_units=64, transforms=5),
}
N = 1024


def _boot_reference_generate():
    """mentflow.generate imported by path behind an empty package shell (mentflow/__init__.py pulls POT, scikit-image,
    psdist, ultraplot: absent).  zuko is the REAL library here."""
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("mentflow")
    pkg.__path__ = [os.path.join(REF, "mentflow")]
    sys.modules["mentflow"] = pkg
    import mentflow.generate as gen
    return gen


def _flat_grads(params):
    return [np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.detach().cpu().numpy() for p in params]


def _record(wrapped, tag: str, out: dict, seed: int) -> None:
    g = torch.Generator().manual_seed(seed)
    dist = wrapped._flow()
    d = dist.base.sample(()).shape[-1]
    z = torch.randn(N, d, generator=g) * 1.5
    z[0, 0], z[1, d - 1], z[2, 0] = 6.0, -5.5, 5.0                         # outside / on the spline domain
    cx = torch.randn(N, d, generator=g)
    cl = torch.randn(N, generator=g)
    params = list(wrapped.parameters())
    for p in params:
        p.grad = None
    # NormalizingFlow.rsample_and_log_prob with the base draw injected: x = transform.inv(z), logp = base.log_prob(z) - ladj
    x, ladj = dist.transform.inv.call_and_ladj(z)
    logp = dist.base.log_prob(z) - ladj
    ((x * cx).sum() + (logp * cl).sum()).backward()
    with torch.no_grad():
        steps = wrapped.forward_steps(z)
        xin = x.detach()
        zinv = wrapped.inverse(xin)
        isteps = wrapped.inverse_steps(xin)
        logp_x = wrapped.log_prob(xin)
        x_fw = wrapped.forward(z)
    out.update({f"{tag}_z": z.numpy(), f"{tag}_cx": cx.numpy(), f"{tag}_cl": cl.numpy(), f"{tag}_x": x.detach().numpy(),
                f"{tag}_log_prob": logp.detach().numpy(), f"{tag}_forward": x_fw.numpy(),
                f"{tag}_forward_steps": np.stack([s.numpy() for s in steps]), f"{tag}_inverse": zinv.numpy(),
                f"{tag}_inverse_steps": np.stack([s.numpy() for s in isteps]), f"{tag}_log_prob_of_x": logp_x.numpy()})
    for i, gnp in enumerate(_flat_grads(params)):
        out[f"{tag}_grad_{i}"] = gnp
    sd = wrapped.state_dict()
    for i, k in enumerate(sd.keys()):
        out[f"{tag}_sd_{i}"] = sd[k].detach().cpu().numpy()


def main() -> int:
    try:
        import zuko
    except ModuleNotFoundError:
        print("zuko is not importable here: nothing written; flow parity stays UNPINNED "
              f"(needs zuko=={ZUKO_VERSION}, reference pyproject.toml:11)", file=sys.stderr)
        return 3
    version = getattr(zuko, "__version__", "unknown")
    if version != ZUKO_VERSION:
        print(f"zuko {version} found, the reference pins {ZUKO_VERSION}: refusing to write fixtures", file=sys.stderr)
        return 3
    if not os.path.isdir(REF):
        print(f"{REF} is absent: the fixtures must come from the reference's own build_flow", file=sys.stderr)
        return 3
    gen = _boot_reference_generate()
    os.makedirs(OUT, exist_ok=True)
    for case, kws in CASES.items():
        torch.manual_seed(1000 + len(case) + kws["output_features"])
        wrapped = gen.build_generator(device=torch.device("cpu"), **kws)      # build.py:80-123 -> build_flow
        out = {}
        sd = wrapped.state_dict()
        params = [n for n, _ in wrapped.named_parameters()]
        meta = {"zuko": version, "case": case, "build_kwargs": kws, "n": N, "sd_keys": list(sd.keys()),
                "sd_shapes": [list(v.shape) for v in sd.values()], "sd_dtypes": [str(v.dtype) for v in sd.values()],
                "param_names": params, "generated_by": "oracle/gen_flow_golden.py"}
        _record(wrapped, "default", out, seed=11)
        # "steep" copy: last linear of every hyper-network scaled so that the conditioner matters (as tests/test_flow_kernels.py)
        with torch.no_grad():
            gs = torch.Generator().manual_seed(5)
            for name, p in wrapped.named_parameters():
                idx = name.split(".hyper.")[-1].split(".")[0] if ".hyper." in name else None
                last = str(2 * kws["hidden_layers"])
                if idx == last and name.endswith("weight"):
                    p.mul_(4.0)
                if idx == last and name.endswith("bias"):
                    p.add_(torch.randn(p.shape, generator=gs))
        _record(wrapped, "steep", out, seed=12)
        out["meta_json"] = np.array(json.dumps(meta))
        path = os.path.join(OUT, f"ref_flow_{case}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB, {len(sd)} state_dict entries)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
