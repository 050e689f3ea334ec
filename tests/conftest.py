import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: torch.from_numpy(np.asarray(f[k])) if f[k].ndim > 0 else torch.tensor(f[k]) for k in f.files}


@pytest.fixture
def golden():
    return load_golden


# ---------------------------------------------------------------------------------------------------------------
# Backends for the kernel tests: "hip" = the product library on a real MI355X (marked gpu); "emu" = the SAME kernel
# sources compiled for the host against tests/emu/hip_emu.h (test infrastructure) so that kernel logic is also
# exercised in the GPU-less container.  The emulated build is never used by the product package.
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libmentflow_emu.so")


def _emu_up_to_date():
    if not os.path.exists(EMU_LIB):
        return False
    t = os.path.getmtime(EMU_LIB)
    srcs = [os.path.join(ROOT, "mentflow_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "mentflow_amd", "csrc"))
            if f.endswith((".hip", ".h", ".inc"))]
    srcs += [os.path.join(EMU_DIR, f) for f in ("hip_emu.h", "hip_emu.cpp")]
    srcs.append(os.path.join(ROOT, "include", "mentflow_hip.h"))
    return all(os.path.getmtime(s) <= t for s in srcs)


@pytest.fixture(scope="session")
def emu_library():
    import subprocess
    if not _emu_up_to_date():
        subprocess.run(["bash", os.path.join(EMU_DIR, "build_emu.sh")], check=True, capture_output=True)
    return EMU_LIB


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """Yields the torch device on which the loaded library computes."""
    from mentflow_amd import _lib
    if request.param == "emu":
        _lib.use_library(request.getfixturevalue("emu_library"))
        yield torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
        _lib.use_library(_lib.DEFAULT_PATH)
        yield torch.device("cuda", 0)


def set_bwd_variant(monkeypatch, value):
    """Select the flow backward variant for the rest of the test: "1" / True = fused kernel, "0" / False = two-kernel path,
    None = default.  (The library reads MENTFLOW_BWD_FUSED once; tests switch through the C ABI's mf_flow_set_bwd_variant.)"""
    from mentflow_amd import _lib
    v = None if value is None else (str(value) not in ("0", "False"))
    _lib.set_flow_bwd_variant(v)


@pytest.fixture(autouse=True)
def _reset_bwd_variant():
    yield
    from mentflow_amd import _lib
    if _lib._lib is not None:
        try:
            _lib.set_flow_bwd_variant(None)
        except Exception:
            pass
