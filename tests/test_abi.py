"""C-ABI checks that need no GPU: the built gfx950 library loads and exports every symbol include/mentflow_hip.h
declares, the ctypes prototypes cover exactly those symbols, and the product refuses to compute on the CPU."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mentflow_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def hip_library():
    import __graft_entry__ as g
    if not os.path.exists(g.LIB):
        g.build()
    return g.LIB


def test_header_symbols_are_exported_by_the_gfx950_library(hip_library):
    lib = ctypes.CDLL(hip_library)
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/mentflow_hip.h but not exported"
    lib.mf_abi_version.restype = ctypes.c_int
    lib.mf_is_emulation.restype = ctypes.c_int
    assert lib.mf_abi_version() == 5
    assert lib.mf_is_emulation() == 0                 # the product library is the real thing
    # the code object really is gfx950
    blob = open(hip_library, "rb").read()
    assert b"gfx950" in blob


def test_ctypes_prototypes_match_header(hip_library):
    from mentflow_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()


def test_no_cpu_fallback(hip_library):
    """With the product library loaded, CPU tensors are refused loudly (no silent eager path)."""
    from mentflow_amd import _lib
    import mentflow_amd as mf
    _lib.use_library(hip_library)
    diag = mf.diagnostics.Histogram1D(edges=torch.linspace(-1, 1, 9), bandwidth=0.5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        diag(torch.randn(16, 2))
    gen = mf.generate.build_generator("nsf", input_features=2, output_features=2, hidden_layers=3, hidden_units=64,
                                      transforms=1, bins=20)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gen.sample(8)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mentflow_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "DEFAULT_PATH", str(tmp_path / "libmentflow_hip.so"))
    with pytest.raises(_lib.LibraryError, match="no CPU fallback"):
        _lib.get_lib()


def test_unsupported_pieces_raise():
    import mentflow_amd as mf
    with pytest.raises(ValueError):
        mf.generate.build_generator("not-a-flow", input_features=2, output_features=2, hidden_layers=3, hidden_units=64,
                                    transforms=1)
    nn_gen = mf.generate.build_generator("nn", input_features=2, output_features=2, hidden_layers=3, hidden_units=64)
    assert nn_gen.log_prob(None) is None and nn_gen.sample_and_log_prob(5)[0].shape == (5, 2)
    with pytest.raises(NotImplementedError):
        mf.generate.build_generator("nsf", input_features=2, output_features=2, hidden_layers=3, hidden_units=196,
                                    transforms=1)
