"""Outputs of the i8x3 network kernel on a few sizes -> gpurun_out/i8s_<tag>.pt (tag = $NEUMAN_I8_KERNEL or 'i8w'); with two files present
compares them.  python tools/i8s_check.py"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "ml-neuman_amd"))
import torch
from neuman_hip import synthetic
tag = os.environ.get("NEUMAN_I8_KERNEL", "i8w")
dev = torch.device('cuda')
outs = {}
for seed, mapping in ((0, 'posenc'), (2, 'rotate')):
    net = synthetic.make_joiner(seed, mapping).to(dev)
    g = torch.Generator(device='cuda').manual_seed(7)
    for R, S in [(1, 1), (5, 7), (64, 128), (700, 37), (3000, 100)]:
        o = torch.randn((R, 3), device='cuda', generator=g) * 0.3
        d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g), dim=-1)
        z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0, dim=1).values.contiguous()
        with torch.no_grad():
            outs[f"{mapping}_rays_{R}x{S}"] = net.forward_rays(o, d, z, precision='i8x3', sigma_scale=1.3).cpu()
            pts = (o[:, None, :] + d[:, None, :] * z[..., None]).contiguous()
            outs[f"{mapping}_pts_{R}x{S}"] = net(pts, d[:, None, :].expand(pts.shape).contiguous(), precision='i8x3').cpu()
            ref = net.forward_rays(o, d, z, precision='fp16x3', sigma_scale=1.3).cpu()
        print(tag, mapping, R, S, 'vs fp16x3: rgb', (outs[f"{mapping}_rays_{R}x{S}"][..., :3] - ref[..., :3]).abs().max().item(), 'sigma', (outs[f"{mapping}_rays_{R}x{S}"][..., 3] - ref[..., 3]).abs().max().item(), 'nan', torch.isnan(outs[f"{mapping}_rays_{R}x{S}"]).any().item())
os.makedirs("gpurun_out", exist_ok=True)
torch.save(outs, f"gpurun_out/i8s_{tag}.pt")
other = f"gpurun_out/i8s_{'i8w' if tag != 'i8w' else 'as'}.pt"
if os.path.exists(other):
    b = torch.load(other)
    for k in outs:
        print(k, 'bit-identical' if torch.equal(outs[k], b[k]) else f"DIFFERS max {(outs[k] - b[k]).abs().max().item():.3e}")
