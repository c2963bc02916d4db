"""How much of the two MLP launches of the C2 frame is DATA-DEPENDENT power (VERDICT r4, item 2's one-run experiment; MI355X_MICROARCH.md saw
+19 % on zero inputs): the shading launch (163.84 M evaluations, nerf_mlp_i8s_kernel) and the sampling launch (81.92 M, nerf_sigma_f16t_kernel)
timed with three weight sets of the SAME shapes --

    default    synthetic-dense random weights (what bench.py times)
    zero       every weight 0 (biases 0.01): the weight operand of every MFMA is zero, every product is zero
    alt        the default magnitudes with checkerboard signs: products cancel pairwise, operand bits toggle as with `default`

-- alternated `--rounds` times (box drift cancels), HIP events around each launch.  One JSON line per (launch, weights) with the best and the
median time, and a summary line with the ratios."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-neuman_amd"))
import torch  # noqa: E402

from neuman_hip import ray_utils, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
dev = torch.device('cuda')
cap = synthetic.SimpleCapture(800, 800)
o, d = ray_utils.shot_all_rays_dev(cap, dev)
R = o.shape[0]
g = torch.Generator(device=dev).manual_seed(0)
z_fine = torch.sort(torch.rand((R, 256), device=dev, generator=g) * 3.14, dim=1).values.contiguous()
z_coarse = torch.sort(torch.rand((R, 128), device=dev, generator=g) * 3.14, dim=1).values.contiguous()


def make(kind):
    net = synthetic.make_joiner(1)
    with torch.no_grad():
        for p in net.parameters():
            if kind == "zero":
                p.zero_() if p.dim() == 2 else p.fill_(0.01)      # (biases 0.01: every activation row is a non-zero constant, so the i8 path's per-row scale stays finite)
            elif kind == "alt" and p.dim() == 2:
                i = torch.arange(p.shape[0])[:, None] + torch.arange(p.shape[1])[None, :]
                p.copy_(p.abs() * (1.0 - 2.0 * (i % 2).to(p.dtype)))
    return net.to(dev).eval()


nets = {k: make(k) for k in ("default", "zero", "alt")}
times = {(l, k): [] for l in ("fine_i8s", "coarse_f16t") for k in nets}
with torch.no_grad():
    for k, net in nets.items():                                       # handles, images, first-launch costs
        net.forward_rays(o[:8192], d[:8192], z_fine[:8192], precision="i8x3")
        net.forward_rays(o[:8192], d[:8192], z_coarse[:8192], precision="fp16x3", sigma_only=True)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for k, net in nets.items():
            for launch, zz, prec, so in (("fine_i8s", z_fine, "i8x3", False), ("coarse_f16t", z_coarse, "fp16x3", True)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                net.forward_rays(o, d, zz, precision=prec, sigma_only=so)
                e1.record()
                torch.cuda.synchronize()
                times[(launch, k)].append(e0.elapsed_time(e1))
summary = {}
for (launch, k), ms in times.items():
    ms = sorted(ms)
    print(json.dumps({"launch": launch, "weights": k, "best_ms": ms[0], "median_ms": ms[len(ms) // 2], "all_ms": ms}), flush=True)
    summary[f"{launch}/{k}"] = ms[len(ms) // 2]
for launch in ("fine_i8s", "coarse_f16t"):
    base = summary[f"{launch}/default"]
    print(json.dumps({"launch": launch, "median_ms_default": base, "zero_over_default": summary[f"{launch}/zero"] / base, "alt_over_default": summary[f"{launch}/alt"] / base,
                      "reading": "zero < default: the difference is what operand toggling costs at the board's power limit; alt ~ default: it is the toggling, not the magnitudes"}), flush=True)
