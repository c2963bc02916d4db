"""Steady-state time of the fine (shading) launch of the C2 frame in i8x3: 10 back-to-back launches after 4 warm-up launches (the board's
clock settles at its power limit after the first launches), mean and spread.  NEUMAN_HIP_LIB selects an experimental build."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-neuman_amd"))
import torch  # noqa: E402

from neuman_hip import ray_utils, synthetic  # noqa: E402

dev = torch.device('cuda')
net = synthetic.make_joiner(1).to(dev)
cap = synthetic.SimpleCapture(800, 800)
o, d = ray_utils.shot_all_rays_dev(cap, dev)
z = torch.sort(torch.rand((o.shape[0], 256), device=dev) * 3.14, dim=1).values.contiguous()
with torch.no_grad():
    for _ in range(4):
        out = net.forward_rays(o, d, z, precision="i8x3")
    torch.cuda.synchronize()
    ms = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = net.forward_rays(o, d, z, precision="i8x3")
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
print(f"{os.environ.get('NEUMAN_HIP_LIB', 'tree').split('/')[-1]:36s} mean {sum(ms) / len(ms):7.1f} ms  min {min(ms):7.1f}  max {max(ms):7.1f}  checksum {float(out.double().abs().mean()):.6f}")
