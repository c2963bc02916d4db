"""Diagnostic (GPU box): the human-loss golden's gradients with the device's closest-point query (float32 search) replaced by the oracle's
float64 one on the same points -- does the deviation of the pose / shape / alignment gradients come from WHICH FACE a far-away sample's
foot lands on?   python tools/human_loss_diag.py"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
GOLDEN = os.path.join(ROOT, "tests", "golden", "human_loss.npz")


def make():
    from neuman_hip import human_nerf, human_trainer, synthetic, vanilla
    g = dict(np.load(GOLDEN))
    dev = torch.device('cuda')
    opt = synthetic.default_opt(use_cuda=True, num_offset_nets=1, offset_scale=0.05, offset_scale_type='linear', posenc='posenc')
    pose, betas, align = synthetic.smpl_like_frames(3, 0)
    al = np.stack([np.concatenate([align[f'{i:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1) for i in range(3)]).astype(np.float32)
    al[:, :3, :3] = np.eye(3)[None]
    al[:, 3, :3] = 0.0
    model = synthetic.smpl_like_model(0)
    net = human_nerf.HumanNeRF(opt, pose * 0.3, betas * 0.3, al, scale=1.0, smpl_dir=model)
    for name, seed, mapping in (("coarse_bkg_net", 0, "posenc"), ("fine_bkg_net", 1, "posenc"), ("coarse_human_net", 2, "rotate")):
        getattr(net, name).load_state_dict(synthetic.make_joiner(seed, mapping).state_dict(), strict=True)
    torch.manual_seed(3)
    net.offset_nets[0].load_state_dict(vanilla.build_offset_net(synthetic.default_opt(offset_scale=0.05, offset_scale_type='linear')).state_dict(), strict=True)
    net = net.to(dev)
    net.coarse_bkg_net.eval()
    net.fine_bkg_net.eval()
    net.coarse_human_net.train()
    net.offset_nets.train()
    faces = model['f'].astype(np.int32)
    batch = {k[6:]: torch.as_tensor(v).to(dev) for k, v in g.items() if k.startswith('batch_')}
    batch['cap_id'], batch['cur_view_f'], batch['patch_counter'] = int(g['batch_cap_id']), float(g['batch_cur_view_f']), int(g['batch_patch_counter'])
    opt_l = types.SimpleNamespace(samples_per_ray=24, importance_samples_per_ray=24, perturb=0.0, white_bkg=True, penalize_smpl_alpha=1.0,
                                  penalize_symmetric_alpha=0.1, penalize_dummy=1.0, penalize_hard_surface=0.1, penalize_color_range=0.1, penalize_mask=0.01,
                                  penalize_lpips=0.0, penalize_sharp_edge=0.1, penalize_outside_factor=2.0, dist_exponent=2.0)
    can_caps = [synthetic.SimpleCapture(32, 32, fx=40., c2w=c) for c in g['can_c2w']]
    loss = human_trainer.HumanNeRFLoss(opt_l, net, faces, (g['can_verts'], faces), can_caps, interval_comp=0.8)
    loss.replay = {'offset_net': int(g['offset_net_choice']), 'dummy_dirs_randn': g['dummy_dirs_randn'], 'dummy_pts_rand': g['dummy_pts_rand'],
                   'can_cap': int(g['can_cap_choice']), 'can_pixel_choice': g['can_pixel_choice']}
    return types.SimpleNamespace(g=g, net=net, loss=loss, batch=batch)




def grads(S):
    for p in S.net.parameters():
        p.grad = None
    ld = S.loss.loss_func(S.batch)
    sum(ld.values()).backward()
    P = {"human.pts_linears.0.weight": S.net.coarse_human_net.nerf.pts_linears[0].weight, "poses": S.net.poses, "betas": S.net.betas, "alignments": S.net.alignments}
    return {k: float(np.abs(p.grad.cpu().numpy() - S.g['grad_' + k]).max() / np.abs(S.g['grad_' + k]).max()) for k, p in P.items()}


def main():
    from neuman_hip import ray_utils
    from oracle import warp
    S = make()
    print("device search        ", grads(S))
    real = ray_utils.signed_distance_dev

    def f64_query(pts, mesh):
        s, f, c = real(pts, mesh)
        verts = mesh.verts.cpu().numpy() if hasattr(mesh, 'verts') else None
        os_, of, oc = warp.signed_distance(pts.reshape(-1, 3).cpu().numpy(), f64_query.verts, f64_query.faces)
        print(f"   query of {of.size} points: face ids differ on {(of != f.cpu().numpy()).mean() * 100:.1f} %, closest Linf {np.abs(oc - c.cpu().numpy()).max():.1e}")
        return torch.as_tensor(os_.astype(np.float32)).to(pts.device), torch.as_tensor(of.astype(np.int32)).to(pts.device), torch.as_tensor(oc.astype(np.float32)).to(pts.device)
    # only the warp's query (posed mesh) is replaced: the canonical-mesh query of the shape regulariser keeps the device's
    orig_diff = ray_utils.warp_samples_to_canonical_diff

    def diff(pts, verts, faces, T):
        f64_query.verts, f64_query.faces = verts.detach().cpu().numpy(), np.asarray(faces)[:, :3]
        ray_utils.signed_distance_dev = f64_query
        try:
            return orig_diff(pts, verts, faces, T)
        finally:
            ray_utils.signed_distance_dev = real
    ray_utils.warp_samples_to_canonical_diff = diff
    print("oracle (f64) query   ", grads(S))


if __name__ == "__main__":
    main()
