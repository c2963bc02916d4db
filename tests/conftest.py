import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "ml-neuman_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run with -m gpu on the MI355X box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without a HIP device (or without a built libneuman_hip.so) skips the `gpu` tests
    instead of failing them; on a GPU box nothing is skipped (a missing library there is an error the tests report)."""
    import torch
    lib = os.path.join(ROOT, "ml-neuman_amd", "lib", "libneuman_hip.so")
    if torch.cuda.is_available():
        return
    why = "no HIP device" if os.path.exists(lib) else "no HIP device and libneuman_hip.so is not built"
    skip = pytest.mark.skip(reason=why)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return {name: dict(np.load(os.path.join(GOLDEN, name + ".npz"))) for name in ("ray_ops", "mlp", "render")}


def weight_checksum(sd):
    return np.array([float(sum(np.abs(v).sum(dtype=np.float64) for v in sd.values())),
                     float(sd['nerf.pts_linears.0.weight'][0, 0]), float(sd['nerf.rgb_linear.weight'][2, 5])])


@pytest.fixture(scope="session")
def nets(golden):
    """Seeded synthetic nets (torch modules on CPU + numpy state for the oracle); checks the weights are the ones
    the golden vectors were generated with."""
    from neuman_hip import synthetic
    from oracle.nerf_mlp import JoinerSpec
    out = {}
    for seed, mapping, key in [(0, 'posenc', 'checksum_seed0'), (1, 'posenc', 'checksum_seed1'), (2, 'rotate', 'checksum_seed2')]:
        j = synthetic.make_joiner(seed, mapping)
        sd = synthetic.state_numpy(j)
        np.testing.assert_allclose(weight_checksum(sd), golden['render'][key], rtol=1e-6,
                                   err_msg="nn.Linear default init changed: regenerate tests/golden (make_golden.py)")
        out[seed] = (j, sd, JoinerSpec(mapping=mapping))
    return out
