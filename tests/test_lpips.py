"""neuman_hip.lpips: the structure of the AlexNet LPIPS metric (render_test_views.py:35-38) on synthetic weights -- the real weights
ship with packages that are absent here, so parity with `lpips.LPIPS(net='alex')` itself is UNPINNED (stated in the module)."""
import numpy as np
import pytest
import torch


def synthetic_state(seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    cin = 3
    for i, (fidx, ch, k) in enumerate([(0, 64, 11), (3, 192, 5), (6, 384, 3), (8, 256, 3), (10, 256, 3)]):
        sd[f'net.slice{i + 1}.{fidx}.weight'] = torch.randn((ch, cin, k, k), generator=g) / (cin * k * k) ** 0.5
        sd[f'net.slice{i + 1}.{fidx}.bias'] = torch.randn(ch, generator=g) * 0.1
        sd[f'lin{i}.model.1.weight'] = torch.rand((1, ch, 1, 1), generator=g) / ch
        cin = ch
    sd['scaling_layer.shift'] = torch.tensor([-.030, -.088, -.188])[None, :, None, None]
    sd['scaling_layer.scale'] = torch.tensor([.458, .448, .450])[None, :, None, None]
    return sd


def test_structure_and_metric_properties():
    from neuman_hip.lpips import LPIPS, lpips_uint8
    m = LPIPS(synthetic_state())
    rng = np.random.default_rng(0)
    a = torch.tensor(rng.uniform(-1, 1, size=(2, 3, 96, 128)).astype(np.float32))
    b = (a + torch.tensor(rng.normal(size=a.shape).astype(np.float32)) * 0.1).clamp(-1, 1)
    c = (a + torch.tensor(rng.normal(size=a.shape).astype(np.float32)) * 0.4).clamp(-1, 1)
    f = m.features(a)
    assert [t.shape[1] for t in f] == [64, 192, 384, 256, 256]
    assert [tuple(t.shape[2:]) for t in f] == [(23, 31), (11, 15), (5, 7), (5, 7), (5, 7)]      # AlexNet's spatial pyramid for 96x128
    daa, dab, dba, dac = m(a, a), m(a, b), m(b, a), m(a, c)
    assert daa.shape == (2, 1, 1, 1) and float(daa.abs().max()) == 0.0
    assert torch.allclose(dab, dba) and (dab > 0).all() and (dac > dab).all()                   # symmetric, positive, grows with the distortion
    # an independent evaluation of the definition with torch modules
    ref = 0
    h0, h1 = (a - m.shift) / m.scale, (b - m.shift) / m.scale
    sd = synthetic_state()
    convs = [torch.nn.Conv2d(3, 64, 11, 4, 2), torch.nn.Conv2d(64, 192, 5, 1, 2), torch.nn.Conv2d(192, 384, 3, 1, 1),
             torch.nn.Conv2d(384, 256, 3, 1, 1), torch.nn.Conv2d(256, 256, 3, 1, 1)]
    for i, (fidx, conv) in enumerate(zip((0, 3, 6, 8, 10), convs)):
        conv.weight.data, conv.bias.data = sd[f'net.slice{i + 1}.{fidx}.weight'], sd[f'net.slice{i + 1}.{fidx}.bias']
        if i in (1, 2):
            h0, h1 = torch.nn.functional.max_pool2d(h0, 3, 2), torch.nn.functional.max_pool2d(h1, 3, 2)
        h0, h1 = torch.relu(conv(h0)), torch.relu(conv(h1))
        u0 = h0 / (h0.norm(dim=1, keepdim=True) + 1e-10)
        u1 = h1 / (h1.norm(dim=1, keepdim=True) + 1e-10)
        ref = ref + torch.nn.functional.conv2d((u0 - u1) ** 2, sd[f'lin{i}.model.1.weight']).mean(dim=(2, 3), keepdim=True)
    assert torch.allclose(dab, ref.detach(), rtol=1e-5, atol=1e-8)
    # the uint8 entry of render_test_views.py, and a bare torchvision-style feature dict
    p = rng.integers(0, 256, size=(64, 80, 3)).astype(np.uint8)
    q = np.clip(p.astype(np.int32) + rng.integers(-20, 20, size=p.shape), 0, 255).astype(np.uint8)
    assert lpips_uint8(m, p, p) == 0.0 and lpips_uint8(m, p, q) > 0
    tv = {k.replace(f'net.slice{i + 1}.', 'features.'): v for i in range(5) for k, v in sd.items() if k.startswith(f'net.slice{i + 1}.')}
    tv.update({k: v for k, v in sd.items() if k.startswith('lin')})
    assert torch.allclose(LPIPS(tv)(a, b), dab)
    with pytest.raises(KeyError):
        LPIPS({k: v for k, v in sd.items() if not k.startswith('lin3')})
    # differentiable: the trainer's patch loss (human_nerf_trainer.py:431-435)
    x = a[:1].clone().requires_grad_(True)
    m(x, b[:1]).sum().backward()
    assert x.grad is not None and float(x.grad.abs().sum()) > 0
