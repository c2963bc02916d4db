"""Time the coarse (sampling) launch of the C2 frame -- 81.92 M evaluations, density head only, split-fp16 x3 -- a few times and print
ms / TFLOP/s.  NEUMAN_HIP_LIB selects an experimental build of the library (tools/build_variant.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "ml-neuman_amd"))
import torch
from neuman_hip import ray_utils, synthetic
dev = torch.device('cuda')
net = synthetic.make_joiner(0).to(dev)
cap = synthetic.SimpleCapture(800, 800)
o, d = ray_utils.shot_all_rays_dev(cap, dev)
R = o.shape[0]
z = torch.sort(torch.rand((R, 128), device=dev) * 3.14, dim=1).values.contiguous()
with torch.no_grad():
    net.forward_rays(o[:8192], d[:8192], z[:8192], precision="fp16x3", sigma_only=True)
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = net.forward_rays(o, d, z, precision="fp16x3", sigma_only=True)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
best = min(ms)
print(f"{os.environ.get('NEUMAN_HIP_LIB', 'tree').split('/')[-1]:32s} coarse fp16x3 sigma-only: {best:7.1f} ms  {R * 128 * 1186816 / best / 1e9:6.0f} TFLOP/s  (all: {' '.join(f'{m:.1f}' for m in ms)})")
