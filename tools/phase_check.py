import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "ml-neuman_amd"))
import torch
from neuman_hip import ray_utils, synthetic
dev = torch.device('cuda')
net = synthetic.make_joiner(0).to(dev)
g = torch.Generator(device='cuda').manual_seed(1)
for R, S in [(1,1),(64,128),(700,37),(2100,64),(4099,127),(100000,128)]:
    o = torch.randn((R, 3), device='cuda', generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g), dim=-1)
    z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0, dim=1).values.contiguous()
    with torch.no_grad():
        full = net.forward_rays(o, d, z, precision='fp16x3', sigma_scale=1.7)
        dens = net.forward_rays(o, d, z, precision='fp16x3', sigma_scale=1.7, sigma_only=True)
    torch.cuda.synchronize()
    eq = torch.equal(dens[..., 3], full[..., 3])
    print(R, S, 'bit-identical sigma:', eq, 'max diff', (dens[...,3]-full[...,3]).abs().max().item(), 'nan', torch.isnan(dens).any().item())
