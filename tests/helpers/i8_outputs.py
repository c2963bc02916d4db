"""Outputs of the NM_PREC_I8X3 network kernel on a fixed set of seeded cases (points, rays and ray-chunk entry points, both encodings,
ragged sizes) -- the kernel the library picks: nerf_mlp_i8s_kernel by default, nerf_mlp_i8w_kernel under NEUMAN_I8_KERNEL=w.
Imported by tests/test_hip_i8_as.py, and run by it as a script (python i8_outputs.py OUT.pt) under the other setting."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "ml-neuman_amd"))
import torch  # noqa: E402

from neuman_hip import synthetic  # noqa: E402

CASES = [(1, 1), (5, 7), (2, 127), (3, 129), (64, 128), (700, 37), (1500, 100)]


def outputs():
    dev = torch.device('cuda')
    outs = {}
    for seed, mapping in ((0, 'posenc'), (2, 'rotate')):
        net = synthetic.make_joiner(seed, mapping).to(dev)
        g = torch.Generator(device='cuda').manual_seed(7)
        for R, S in CASES:
            o = torch.randn((R, 3), device='cuda', generator=g) * 0.3
            d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g), dim=-1)
            z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0, dim=1).values.contiguous()
            with torch.no_grad():
                outs[f"{mapping}_rays_{R}x{S}"] = net.forward_rays(o, d, z, precision='i8x3', sigma_scale=1.3).cpu()
                pts = (o[:, None, :] + d[:, None, :] * z[..., None]).contiguous()
                outs[f"{mapping}_pts_{R}x{S}"] = net(pts, d[:, None, :].expand(pts.shape).contiguous(), precision='i8x3').cpu()
                if S >= 7:                                                   # a chunk of a march: every other ray, samples 3 .. 3 + S // 2 - 1
                    idx = torch.arange(0, R, 2, device='cuda', dtype=torch.int32)
                    live = torch.tensor([idx.numel()], device='cuda', dtype=torch.int32)
                    buf = torch.full((R, S, 4), -7.0, device='cuda')
                    net.forward_ray_chunk(o, d, z, idx, live, 3, S // 2, buf, precision='i8x3', sigma_scale=0.7)
                    outs[f"{mapping}_chunk_{R}x{S}"] = buf.cpu()
    return outs


if __name__ == "__main__":
    torch.save(outputs(), sys.argv[1])
