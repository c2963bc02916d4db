"""Time one iteration of the human trainer (trainers/human_nerf_trainer.py:180-446 + backward + Adam) on the device pieces at the
reference's batch: `rays` rays of one frame of an SMPL-sized body (6890 vertices, 13776 faces), frozen background 128 + 128 samples,
human 128 samples through offset net / differentiable skinning / warp / human net, the seven loss terms.  Prints one JSON line.
    python tools/human_step_bench.py [rays] [iterations]

A training benchmark that TRAINS (round 5): the target colours are the rendering of the same rays by a second, frozen human net (a field the
trained one can approach, not torch.rand), the trained net starts from the plain nn.Linear initialisation (mixed-sign density: the
synthetic-dense preset's 40x alpha weights make Adam overshoot into an all-negative density within ten steps, which the reference's
restart of human_nerf_trainer.py:437-442 then turns into iterations that train nothing) and the learning rate is 2e-4.  Every timed iteration
must be alive, and the mean total loss of the last five iterations must be below that of the first five -- asserted, and both are printed."""
import json
import os
import sys
import time
import types

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neuman_hip import human_trainer, ray_utils, smpl, synthetic, vanilla  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda')


class HumanNeRFLike(torch.nn.Module):
    """the attributes of models/human_nerf.py HumanNeRF the trainer reads, on the synthetic SMPL-sized body"""

    def __init__(self, human_seed=2, dense=False):
        super().__init__()
        self.coarse_bkg_net, self.fine_bkg_net = synthetic.make_joiner(0).to(dev), synthetic.make_joiner(1).to(dev)
        self.coarse_human_net = synthetic.make_joiner(human_seed, 'rotate', dense=dense).to(dev)
        opt = synthetic.default_opt(offset_scale=0.05, offset_scale_type='linear')
        torch.manual_seed(3)
        self.offset_nets = torch.nn.ModuleList([vanilla.build_offset_net(opt).to(dev)])
        self.body = smpl.SMPLDiff(synthetic.smpl_like_model(0), dev)
        pose, betas, align = synthetic.smpl_like_frames(3, 0)
        al = np.stack([np.concatenate([align[f'{i:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1) for i in range(3)]).astype(np.float32)
        al[:, :3, :3] = np.eye(3)[None]
        al[:, 3, :3] = 0.0
        self.poses = torch.nn.Parameter(torch.tensor(pose * 0.3, device=dev))
        self.betas = torch.nn.Parameter(torch.tensor(betas * 0.3, device=dev))
        self.alignments = torch.nn.Parameter(torch.tensor(al, device=dev))
        self.scale = 1.0

    def vertex_forward(self, idx):
        return self.body.vertex_forward(self.poses[idx][None], self.betas[idx][None], self.alignments[idx], self.scale)


net = HumanNeRFLike()
for n in (net.coarse_bkg_net, net.fine_bkg_net):
    n.eval()
net.coarse_human_net.train()
net.offset_nets.train()
model = synthetic.smpl_like_model(0)
faces = model['f'].astype(np.int32)
with torch.no_grad():
    world, _ = net.vertex_forward(1)
    T_da, v_shaped = net.body.transformations(net.body.da_smpl, net.betas[1][None])
    can_verts = torch.einsum('vab,vb->va', T_da, torch.cat([v_shaped, torch.ones_like(v_shaped[:, :1])], 1))[:, :3].cpu().numpy()
cap = synthetic.SimpleCapture(256, 256, fx=560., c2w=synthetic.spherical_c2w(15., -5., 3.0), near=0.5, far=5.0)
coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
coords = coords[np.random.default_rng(1).choice(len(coords), R, replace=False)]
o, d = ray_utils.shot_rays(cap, coords)
o, d = torch.tensor(o, dtype=torch.float32, device=dev), torch.tensor(d, dtype=torch.float32, device=dev)
near, far = ray_utils.geometry_guided_near_far(o, d, world[0], 0.2)
hit = near < far
near, far = torch.where(hit, near, torch.full_like(near, 2.0)), torch.where(hit, far, torch.full_like(far, 3.0))
batch = {'origin': o, 'direction': d, 'bkg_near': torch.full((R, 1), cap.near['bkg'], device=dev), 'bkg_far': torch.full((R, 1), cap.far['bkg'], device=dev),
         'human_near': near[:, None].contiguous(), 'human_far': far[:, None].contiguous(), 'is_hit': hit, 'is_bkg': (~hit).long(),
         'color': torch.rand((R, 3), device=dev), 'cur_view_f': 0.35, 'cap_id': 1, 'patch_counter': 0}
opt = types.SimpleNamespace(samples_per_ray=128, importance_samples_per_ray=128, perturb=1.0, white_bkg=True, penalize_smpl_alpha=1.0,
                            penalize_symmetric_alpha=0.1, penalize_dummy=1.0, penalize_hard_surface=0.1, penalize_color_range=0.1, penalize_mask=0.01,
                            penalize_lpips=0.0, penalize_sharp_edge=0.1, penalize_outside_factor=2.0, dist_exponent=2.0)
can_caps = [synthetic.SimpleCapture(64, 64, fx=80., c2w=synthetic.spherical_c2w(a, 0., 3.0)) for a in (0., 90., 200.)]
loss = human_trainer.HumanNeRFLoss(opt, net, faces, (can_verts, faces), can_caps, interval_comp=0.8, seed=4)
# the target: the same batch rendered through a second human net (dense preset: an opaque-ish body in front of the same background)
target = HumanNeRFLike(human_seed=9, dense=True)
for n in (target.coarse_bkg_net, target.fine_bkg_net, target.coarse_human_net):
    n.eval()
target.offset_nets.eval()
with torch.no_grad():
    tl = human_trainer.HumanNeRFLoss(opt, target, faces, (can_verts, faces), can_caps, interval_comp=0.8, seed=4)
    _, rgb_t = tl.loss_func(batch, return_rgb=True)
batch['color'] = rgb_t.detach().clamp(0.0, 1.0).contiguous()
del target, tl
params = list(net.coarse_human_net.parameters()) + list(net.offset_nets.parameters()) + [net.poses, net.betas, net.alignments]
optim = torch.optim.Adam(params, lr=2e-4)
for _ in range(3):
    loss.train_step(batch, optim)
torch.cuda.synchronize()
ts, totals, alive = [], [], []
for _ in range(ITERS):
    t0 = time.perf_counter()
    terms, total = loss.train_step(batch, optim)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
    totals.append(total)
    alive.append(bool(loss.last['alive']))
ms = sorted(ts)[len(ts) // 2] * 1e3
first, last = sum(totals[:5]) / 5, sum(totals[-5:]) / 5
line = {"rays": R, "hit_rays": int(hit.sum()), "body": [6890, int(faces.shape[0])], "samples": {"bkg": [128, 256], "human": 128},
        "ms_per_iteration": ms, "iterations_per_s": 1e3 / ms, "iterations_timed": ITERS, "alive_on_every_timed_iteration": all(alive),
        "total_loss_first5_mean": first, "total_loss_last5_mean": last, "total_loss_first": totals[0], "total_loss_last": totals[-1], "terms_last": terms,
        "learning_rate": 2e-4, "target": "the batch rendered through a second frozen human net (seed 9, dense preset)"}
print(json.dumps(line), flush=True)
if os.environ.get('NEUMAN_HOST_PROFILE') == '1':                        # where the HOST's time goes (cProfile over 20 iterations, by own time), to stderr
    import cProfile
    import pstats
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(20):
        loss.train_step(batch, optim)
    pr.disable()
    torch.cuda.synchronize()
    print(f"[host profile] 20 iterations in {(time.perf_counter() - t0) * 1e3:.1f} ms under cProfile", file=sys.stderr)
    pstats.Stats(pr, stream=sys.stderr).sort_stats('tottime').print_stats(45)
if os.environ.get('NEUMAN_LAUNCH_SOURCES') == '1':                      # where the iteration's launches come from (tools/launch_sources.py), to stderr
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import launch_sources
    launch_sources.count(lambda: loss.train_step(batch, optim), top=70)
assert all(alive), "the human net died during the timed iterations"
assert all(v == v for v in totals) and (last < first or ITERS < 10), (first, last)
