"""Is the k-loop issue-bound or power-bound?  Same kernel, same instruction stream, on (a) the synthetic-dense weights
and random points, (b) all-zero weights: identical work, minimal operand toggling (guide: DVFS give-back)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch
from neuman_hip import synthetic

n = 128 * 256 * 200
g = torch.Generator(device='cuda').manual_seed(0)
pts = (torch.rand((n, 3), device='cuda', generator=g) * 3 - 1.5).contiguous()
dirs = torch.nn.functional.normalize(torch.randn((n, 3), device='cuda', generator=g), dim=-1).contiguous()
for tag in ("random", "zero-weights"):
    j = synthetic.make_joiner(0).cuda()
    if tag == "zero-weights":
        with torch.no_grad():
            for p in j.parameters():
                p.zero_()
    for prec in ("bf16x3", "i8x3", "bf16"):
        j(pts, dirs, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            j(pts, dirs, precision=prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{tag:13s} {prec:7s}: {ms:8.2f} ms -> {n * 1186816 / ms / 1e9:7.1f} TFLOP/s algorithmic")
