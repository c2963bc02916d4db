"""Time the fine (shading) launch of the C2 frame -- 163.84 M evaluations through nm_mlp_forward_rays in i8x3 -- a few times and print
ms / TFLOP/s.  NEUMAN_I8_KERNEL=w|r selects the schedule, NEUMAN_HIP_LIB an experimental build of the library."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-neuman_amd"))
import torch  # noqa: E402

from neuman_hip import ray_utils, synthetic  # noqa: E402

dev = torch.device('cuda')
net = synthetic.make_joiner(1).to(dev)
cap = synthetic.SimpleCapture(800, 800)
o, d = ray_utils.shot_all_rays_dev(cap, dev)
R = o.shape[0]
z = torch.sort(torch.rand((R, 256), device=dev) * 3.14, dim=1).values.contiguous()
with torch.no_grad():
    net.forward_rays(o[:8192], d[:8192], z[:8192], precision="i8x3")
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = net.forward_rays(o, d, z, precision="i8x3")
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
best = min(ms)
print(f"{os.environ.get('NEUMAN_HIP_LIB', 'tree').split('/')[-1]:32s} kernel {os.environ.get('NEUMAN_I8_KERNEL', 'r')}: {best:7.1f} ms  "
      f"{R * 256 * 1186816 / best / 1e9:6.0f} TFLOP/s  (all: {' '.join(f'{m:.1f}' for m in ms)})  checksum {float(out.double().abs().mean()):.6f}")
