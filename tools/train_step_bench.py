"""Time one training iteration of the background NeRF (trainers/vanilla_nerf_trainer.py:45-96 + backward + Adam) on the HIP
training slice (neuman_hip/train.py): rays_per_batch rays, S coarse samples through the coarse net, S + NI through the fine net.
Prints one JSON line.     python tools/train_step_bench.py [rays] [S] [NI]"""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from neuman_hip import ray_utils, render_utils, synthetic  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
NI = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda")
coarse, fine = synthetic.make_joiner(0).to(dev).train(), synthetic.make_joiner(1).to(dev).train()
opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
cap = synthetic.SimpleCapture(800, 800)
o_all, d_all = ray_utils.shot_all_rays_dev(cap, dev)
g = torch.Generator(device='cpu').manual_seed(0)
color = torch.rand((R, 3), generator=g).to(dev)


def step(i):
    idx = torch.randint(0, o_all.shape[0], (R,), generator=g).to(dev)
    o, d = o_all[idx].contiguous(), d_all[idx].contiguous()
    near, far = torch.full((R,), float(cap.near['bkg']), device=dev), torch.full((R,), float(cap.far['bkg']), device=dev)
    opt.zero_grad()
    pts, _, z = ray_utils.sample_z(o, d, near, far, S, want_points=True)
    dirs = d[:, None, :].expand(pts.shape)
    out = coarse(pts, dirs)
    rgb_map, _, _, weights, _ = render_utils.raw2outputs(out, z, d, white_bkg=True)
    loss = F.mse_loss(rgb_map, color)
    with torch.no_grad():
        zf = ray_utils.importance_z(z, weights.detach(), NI)
    ptsf = o[:, None, :] + d[:, None, :] * zf[..., None]
    outf = fine(ptsf, d[:, None, :].expand(ptsf.shape))
    loss = loss + F.mse_loss(render_utils.raw2outputs(outf, zf, d, white_bkg=True)[0], color)
    loss.backward()
    opt.step()
    return loss


for i in range(2):
    step(i)
torch.cuda.synchronize()
ts = []
for i in range(5):
    t0 = time.perf_counter()
    l = step(i)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ms = sorted(ts)[len(ts) // 2] * 1e3
evals = R * (S + S + NI)
flops = evals * 1_186_816 * 3                      # forward + backward-data + backward-weights
print(json.dumps({"rays": R, "samples": [S, S + NI], "evaluations": evals, "ms_per_iteration": ms, "iterations_per_s": 1e3 / ms,
                  "mlp_tflops": flops / ms / 1e9, "f32_mfma_peak_tflops": 157.3, "loss": float(l.detach())}))
