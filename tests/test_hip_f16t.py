"""nerf_sigma_f16t_kernel (csrc/mlp_f16t.hip + the generated stream csrc/mlp_f16t_body.h): the density-only NM_PREC_FP16X3 network in its
activation-stationary form is the default of nm_mlp_sigma_rays / nm_mlp_sigma_ray_chunk.  Bit-identical to nerf_mlp_kernel
(NEUMAN_SIGMA_KERNEL=w) on every entry point and after every stage; follows nm_mlp_refresh_f16."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
pytestmark = pytest.mark.gpu


@pytest.fixture()
def nets():
    from neuman_hip import synthetic
    return {m: synthetic.make_joiner(1 if m == 'posenc' else 2, m).to('cuda') for m in ('posenc', 'rotate')}


def both(fn, monkeypatch):
    out = {}
    for k in ('t', 'w'):
        monkeypatch.setenv("NEUMAN_SIGMA_KERNEL", k)
        with torch.no_grad():
            out[k] = fn()
        torch.cuda.synchronize()
    monkeypatch.delenv("NEUMAN_SIGMA_KERNEL")
    return out['t'], out['w']


@pytest.mark.parametrize("mapping", ["posenc", "rotate"])
@pytest.mark.parametrize("R,S", [(1, 1), (3, 43), (129, 128), (700, 37), (4099, 127), (40000, 3)])      # (the last two: several tiles per workgroup)
def test_rays_are_bit_identical(nets, monkeypatch, mapping, R, S):
    net = nets[mapping]
    g = torch.Generator(device='cuda').manual_seed(R * 1000 + S)
    o = torch.randn((R, 3), device='cuda', generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g), dim=-1)
    z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0, dim=1).values.contiguous()
    mine, ref = both(lambda: net.forward_rays(o, d, z, precision='fp16x3', sigma_scale=1.7, sigma_only=True), monkeypatch)
    assert torch.isfinite(mine).all()
    assert torch.equal(mine, ref), float((mine - ref).abs().max())
    assert (mine[..., :3] == 0).all()
    with torch.no_grad():
        full = net.forward_rays(o, d, z, precision='fp16x3', sigma_scale=1.7)
    assert torch.equal(mine[..., 3], full[..., 3])


def test_ray_chunks_are_bit_identical(nets, monkeypatch):
    """nm_mlp_sigma_ray_chunk: a compacted list of live rays whose length stays on the device, samples s0 .. s0 + chunk - 1"""
    net = nets['posenc']
    g = torch.Generator(device='cuda').manual_seed(11)
    R, S = 3000, 96
    o = torch.randn((R, 3), device='cuda', generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g), dim=-1)
    z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0, dim=1).values.contiguous()
    live = torch.randperm(R, device='cuda', generator=g)[:1777].to(torch.int32).contiguous()
    idx = torch.cat([live, torch.zeros(R - live.numel(), dtype=torch.int32, device='cuda')]).contiguous()
    cnt = torch.tensor([live.numel()], dtype=torch.int32, device='cuda')

    def run():
        out = torch.full((R, S, 4), -7.0, device='cuda')
        for s0, c in ((0, 32), (32, 48), (80, 16)):
            net.forward_ray_chunk(o, d, z, idx, cnt, s0, c, out, precision='fp16x3', sigma_only=True)
        return out
    mine, ref = both(run, monkeypatch)
    assert torch.equal(mine, ref)
    untouched = torch.ones(R, dtype=torch.bool, device='cuda')
    untouched[live.long()] = False
    assert (mine[untouched] == -7.0).all() and (mine[live.long()][..., 3] != -7.0).all()


@pytest.mark.parametrize("mapping", ["posenc", "rotate"])
def test_every_stage_matches(nets, mapping):
    """a lane's resident inputs of stage st + 1 (nm_mlp_sigma_f16t_debug), decoded, are nerf_mlp_kernel's activations after stage st"""
    from neuman_hip import _lib
    import f16t_debug
    net = nets[mapping]
    n = 1024
    g = torch.Generator(device='cuda').manual_seed(5)
    pts = (torch.rand((n, 3), device='cuda', generator=g) * 2 - 1).contiguous()
    dirs = torch.nn.functional.normalize(torch.randn((n, 3), device='cuda', generator=g), dim=-1).contiguous()
    for st in range(8):
        state = torch.zeros((256, 128), device='cuda', dtype=torch.int32)
        o = torch.zeros((n, 4), device='cuda')
        _lib.check(_lib.lib().nm_mlp_sigma_f16t_debug(net.handle(), _lib.dev_ptr(pts), _lib.dev_ptr(dirs), n, st, ctypes.c_void_p(state.data_ptr()), _lib.dev_ptr(o),
                                                      _lib.stream_ptr()), "nm_mlp_sigma_f16t_debug")
        torch.cuda.synchronize()
        got = f16t_debug.decode(state.cpu().numpy())
        ref = net.forward_debug(pts[:128], dirs[:128], st, precision="fp16x3").cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (mapping, st)


def test_follows_a_device_side_refresh(monkeypatch):
    """nm_mlp_refresh_f16 repacks the fp16 image from live parameters; the density-only stream is re-cut from it in the same call"""
    from neuman_hip import _lib, synthetic
    net = synthetic.make_joiner(1).cuda().train()
    g = torch.Generator(device='cuda').manual_seed(2)
    R, S = 500, 64
    o = torch.randn((R, 3), device='cuda', generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g), dim=-1)
    z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0, dim=1).values.contiguous()
    handle = net.train_handle()

    def sigma():
        out = torch.zeros((R, S, 4), device='cuda')
        _lib.check(_lib.lib().nm_mlp_sigma_rays(handle, _lib.dev_ptr(o), _lib.dev_ptr(d), _lib.dev_ptr(z), R, S, 4, 1.0, _lib.dev_ptr(out), _lib.stream_ptr()), "nm_mlp_sigma_rays")
        return out
    first = None
    for step in range(2):
        ptrs = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in net.nerf.ordered_params()])
        _lib.check(_lib.lib().nm_mlp_refresh_f16(handle, ptrs, _lib.stream_ptr()), "nm_mlp_refresh_f16")
        mine, ref = both(sigma, monkeypatch)
        assert torch.equal(mine, ref)
        if first is None:
            first = mine
        else:
            assert not torch.equal(mine, first)
        with torch.no_grad():
            for p in net.nerf.parameters():
                p.add_(0.01 * torch.randn(p.shape, device='cuda', generator=g))
