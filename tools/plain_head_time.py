"""The plain-head net (use_viewdirs=False, `--specular_can no`) on its two shading kernels: nerf_mlp_i8s_kernel<true> (i8x3, what "mixed" now picks for a
shading pass) against nerf_mlp_kernel fp16x3 (what it ran before), and the view-dependent net's i8 launch beside them.  One JSON line.
    python tools/plain_head_time.py [millions of evaluations, default 16]"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch  # noqa: E402
from neuman_hip import synthetic  # noqa: E402

N = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 16_000_000
dev = torch.device('cuda')
plain = synthetic.make_variant_joiner(5, use_viewdirs=False).to(dev)
full = synthetic.make_joiner(1).to(dev)
pts = torch.randn((N, 3), device=dev) * 0.7
dirs = torch.nn.functional.normalize(torch.randn((N, 3), device=dev), dim=-1)


def ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


with torch.no_grad():
    line = {"evaluations": N,
            "plain_head_i8x3_ms": ms(lambda: plain(pts, None, precision="i8x3")),
            "plain_head_fp16x3_ms": ms(lambda: plain(pts, None, precision="fp16x3")),
            "view_dependent_i8x3_ms": ms(lambda: full(pts, dirs, precision="i8x3"))}
    e = (plain(pts[:1 << 20], None, precision="i8x3") - plain(pts[:1 << 20], None, precision="fp32")).abs()
line["ns_per_evaluation"] = {k[:-3]: v * 1e6 / N for k, v in list(line.items())[1:]}
line["plain_i8x3_vs_f32_kernel_linf"] = {"rgb_pre_sigmoid": float(e[:, :3].max()), "sigma": float(e[:, 3].max())}
print(json.dumps(line))
