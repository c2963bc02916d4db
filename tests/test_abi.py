"""The C-ABI library loads on a CPU-only box and exports exactly what include/neuman_hip.h declares."""
import os
import re

from neuman_hip import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "neuman_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(nm_[a-z0-9_]+)\s*\(", src))


def test_header_symbols_are_exported_and_bound():
    lib = _lib.lib()
    decl = declared_symbols()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in neuman_hip.h but not exported"
    assert decl == set(_lib.SIGNATURES), (decl ^ set(_lib.SIGNATURES))


def test_version_and_error_channel():
    lib = _lib.lib()
    assert lib.nm_version() == 1
    # argument validation happens before any device work, so it is testable without a GPU
    rc = lib.nm_composite(None, None, None, 4, 8, 1, None, None, None, None, None, None, None)
    assert rc == -1 and b"nm_composite" in lib.nm_last_error()
    rc = lib.nm_importance_z(None, None, 4, 8, None, 4, 1, None, None)
    assert rc == -1


def test_no_cpu_fallback_in_host_mirror():
    """CPU tensors must be refused loudly (there is no torch/CPU evaluation of the hot path in the package)."""
    import pytest
    import torch
    from neuman_hip import render_utils, synthetic
    if torch.cuda.is_available():
        pytest.skip("GPU box: the refusal path is exercised on the CPU-only runner")
    with pytest.raises(_lib.NeumanHipError):
        render_utils.raw2outputs(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))
    net = synthetic.make_joiner(0)
    with pytest.raises(_lib.NeumanHipError):
        net(torch.zeros(5, 3), torch.zeros(5, 3))
