"""The activation-stationary i8x3 kernel (csrc/mlp_i8s.hip: the default for whole-network NM_PREC_I8X3 launches) against the
wave-specialised one (csrc/mlp.hip nerf_mlp_i8w_kernel, NEUMAN_I8_KERNEL=w): the same 16-bit fixed-point arithmetic with the operand roles
swapped -- every integer sum exact, every float operation the same and in the same order -- so the outputs are bit-identical, and every
parity statement made for one holds for the other.  The library reads the switch once per process: the other side runs as a script."""
import os
import subprocess
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import i8_outputs  # noqa: E402

pytestmark = pytest.mark.gpu


def test_as_kernel_is_bit_identical_to_the_wave_specialised_kernel(tmp_path):
    assert os.environ.get("NEUMAN_I8_KERNEL", "as") == "as", "this test wants the default kernel in-process"
    mine = i8_outputs.outputs()
    other = tmp_path / "w.pt"
    env = dict(os.environ, NEUMAN_I8_KERNEL="w")
    subprocess.run([sys.executable, i8_outputs.__file__, str(other)], check=True, env=env, timeout=600)
    theirs = torch.load(other)
    assert set(mine) == set(theirs) and len(mine) >= 40
    for k in sorted(mine):
        assert torch.isfinite(mine[k]).all(), k
        assert torch.equal(mine[k], theirs[k]), f"{k}: max |diff| {(mine[k] - theirs[k]).abs().max().item():.3e}"


def test_density_of_the_full_launch_equals_the_density_only_launch():
    """In-process cross-check of the two kernels: the density-only form of a launch stays on the wave-specialised kernel."""
    from neuman_hip import synthetic
    net = synthetic.make_joiner(1).to('cuda')
    g = torch.Generator(device='cuda').manual_seed(3)
    for R, S in [(3, 5), (257, 64), (1000, 129)]:
        o = torch.randn((R, 3), device='cuda', generator=g) * 0.3
        d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g), dim=-1)
        z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0, dim=1).values.contiguous()
        with torch.no_grad():
            full = net.forward_rays(o, d, z, precision='i8x3')
            dens = net.forward_rays(o, d, z, precision='i8x3', sigma_only=True)
        assert torch.equal(full[..., 3], dens[..., 3]), (R, S)
