"""What does keeping the activations cost the training step's forward pass?  The same points through (a) the rendering launch of the same arithmetic
(nerf_mlp_kernel fp16x3, nothing kept), (b) the training forward (nerf_mlp_kernel<4, false, 2>: fp16 copies of nine layers, bits, encodings) and (c) its
backward-data pass, at the trainers' batch sizes.  One JSON line.
    python tools/train_forward_probe.py"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch  # noqa: E402
from neuman_hip import synthetic  # noqa: E402

dev = torch.device('cuda')
net = synthetic.make_joiner(1).to(dev)


def ms(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


line = {}
for n in (131072, 262144, 1048576):
    pts = torch.randn((n, 3), device=dev) * 0.7
    dirs = torch.nn.functional.normalize(torch.randn((n, 3), device=dev), dim=-1)
    net.eval()
    with torch.no_grad():
        t_inf = ms(lambda: net(pts, dirs, precision="fp16x3"))
    net.train()
    keep = {}

    def fwd():
        keep['out'] = net(pts, dirs)

    t_fwd = ms(fwd)
    g = torch.randn_like(keep['out'])

    def both():
        out = net(pts, dirs)
        out.backward(g)
        for p in net.parameters():
            p.grad = None

    t_both = ms(both)
    line[str(n)] = {"rendering_launch_ms": t_inf, "training_forward_ms": t_fwd, "forward_plus_backward_ms": t_both,
                    "ns_per_evaluation": {"rendering": t_inf * 1e6 / n, "training_forward": t_fwd * 1e6 / n, "forward_plus_backward": t_both * 1e6 / n}}
print(json.dumps(line))
