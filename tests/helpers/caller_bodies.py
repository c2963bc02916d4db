"""The loop bodies of the reference's render scripts, written ONCE against a namespace of modules, so that the very same driver code runs
  * in the build container on the REFERENCE's own modules (tests/golden/make_golden_callers.py -> tests/golden/callers.npz), and
  * on the GPU box on stand-in `utils.render_utils` / `utils.ray_utils` / `models.vanilla` modules that neuman_hip.install() has filled
    (tests/test_hip_install_callers.py) -- the reference tree does not exist there (a Python reference cannot travel), so what install()
    rebinds there are empty modules registered under the reference's names; every call below resolves through them.

What the scripts do around these bodies -- read a scene from disk, load a checkpoint, write PNGs -- needs assets that do not exist offline
and is outside the hot path (f3 / f4).  M: namespace with `render_utils` (the module the scripts import as `from utils import
render_utils`); `make_cap(i)`: the capture of frame i (the reference's ResizedPinholeCapture there, the duck-typed capture here)."""
import numpy as np


def canonical_360(M, net, make_cap, n_frames, static_vert, faces, opt, can_bone_mean):
    """render_360.py:52-76 (main_canonical_360's loop): every pose of the 360 path through render_smpl_nerf(render_can=True) -> [n, H, W, 3]"""
    frames = []
    for i in range(n_frames):
        can_cap = make_cap(i)
        out = M.render_utils.render_smpl_nerf(
            net, can_cap, static_vert, faces, Ts=None, rays_per_batch=opt.rays_per_batch, samples_per_ray=opt.samples_per_ray,
            render_can=True, return_mask=False, return_depth=False, interval_comp=opt.geo_threshold / can_bone_mean)
        frames.append(np.asarray(out))
    return np.stack(frames)


def test_views(M, net, make_cap, frame_ids, verts, faces, Ts, opt):
    """render_test_views.py:69-82 (main's loop): every test view through render_hybrid_nerf -> [n, H, W, 3]"""
    frames = []
    for i in frame_ids:
        cap = make_cap(i)
        out = M.render_utils.render_hybrid_nerf(
            net, cap, verts[i], faces, Ts[i], rays_per_batch=opt.rays_per_batch, samples_per_ray=opt.samples_per_ray,
            geo_threshold=opt.geo_threshold, return_depth=False)
        frames.append(np.asarray(out))
    return np.stack(frames)




def posed_360(M, net, make_cap, n_frames, verts, faces, Ts, opt):
    """render_360.py:108-126 (main_posed_360's loop): the posed body of frame 0 from every pose of the 360 path through render_smpl_nerf(render_can=False)
    -> [n, H, W, 3]"""
    frames = []
    for i in range(n_frames):
        out = M.render_utils.render_smpl_nerf(
            net, make_cap(i), verts, faces, Ts, rays_per_batch=opt.rays_per_batch, samples_per_ray=opt.samples_per_ray, white_bkg=opt.white_bkg,
            render_can=False, geo_threshold=opt.geo_threshold)
        frames.append(np.asarray(out))
    return np.stack(frames)

def gathering(M, bkg_net, nets_list, make_cap, n_frames, verts_list, faces, Ts_list, opt):
    """render_gathering.py:189-202 (main's loop): every novel camera through render_hybrid_nerf_multi_persons, the actors' vertices and transforms
    sliced out of the stacked arrays as the script slices them (verts_list [A, frames, V, 3], Ts_list [A, frames, V, 4, 4]) -> [n, H, W, 3]"""
    frames = []
    for i in range(n_frames):
        out = M.render_utils.render_hybrid_nerf_multi_persons(
            bkg_net, make_cap(i), nets_list, verts_list[:, i, ...], [faces] * len(nets_list), Ts_list[:, i, ...],
            rays_per_batch=opt.rays_per_batch, samples_per_ray=opt.samples_per_ray, geo_threshold=opt.geo_threshold, return_depth=False)
        frames.append(np.asarray(out))
    return np.stack(frames)

# the synthetic stand-in for what the scripts read from a scene directory (both sides build it from these definitions)
W360, H360, N360, S360 = 48, 40, 3, 64
WTV, HTV, STV = 40, 32, 64
TV_FRAMES = (0, 1)


WP, HP, SP, NP = 40, 48, 64, 2
WG, HG, SG, NG = 40, 32, 48, 2
G_SHIFTS = ((0.0, 0.0, 0.0), (0.35, 0.0, 0.2), (-0.3, 0.05, -0.15))                 # three actors side by side (tests/golden/make_golden_posed.py's)


def gathering_inputs():
    """what read_actors (render_gathering.py:155-165) stacks: per actor and frame the posed vertices and the Da-pose -> scene transforms"""
    inp = scene_inputs()
    verts, Ts = [], []
    for s in G_SHIFTS:
        s = np.asarray(s, np.float64)
        v_f, t_f = [], []
        for k in range(NG):
            t = np.array(inp['Ts'][k], dtype=np.float64, copy=True)
            t[:, :3, 3] += s
            v_f.append((np.asarray(inp['verts'][k], np.float64) + s).astype(np.float32))
            t_f.append(t)
        verts.append(v_f)
        Ts.append(t_f)
    return {'faces': inp['faces'], 'verts_list': np.array(verts), 'Ts_list': np.array(Ts)}


def scene_inputs():
    from neuman_hip import synthetic
    verts_c, faces = synthetic.capsule_mesh()
    posed, T = synthetic.twist_transforms(verts_c)
    posed2, T2 = synthetic.twist_transforms(verts_c, twist=0.5, shift=(-0.04, 0.03, 0.02))
    return {'static_vert': verts_c, 'faces': faces, 'verts': [posed, posed2], 'Ts': [T, T2]}


# ---- train.py's background trainer: the iteration body -------------------------------------------------------------------------------
R_TR, S_TR, NI_TR, ITERS_TR = 64, 24, 16, 5


def trainer_opt():
    import types
    return types.SimpleNamespace(ablate_nerft=False, samples_per_ray=S_TR, importance_samples_per_ray=NI_TR, perturb=0.0, raw_noise_std=0.0, white_bkg=True,
                                 margin=0.9, delay_iters=2, lrate_decay=250, learning_rate=5e-4, penalize_empty_space=0.1)


def trainer_batch():
    rng = np.random.default_rng(2024)
    ro = (rng.normal(size=(R_TR, 3)) * 0.3).astype(np.float32)
    rd = rng.normal(size=(R_TR, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    near = rng.uniform(0.1, 0.6, size=(R_TR, 1)).astype(np.float32)
    far = (near + rng.uniform(1.0, 2.5, size=(R_TR, 1))).astype(np.float32)
    return dict(origin=ro, direction=rd, near=near, far=far, color=rng.uniform(size=(R_TR, 3)).astype(np.float32),
                depth=rng.uniform(0.8, 2.0, size=(R_TR,)).astype(np.float32))


def background_trainer_iterations(M, coarse_net, fine_net, optim, opt, batch, device, n_iter):
    """n_iter iterations of train.py's background trainer on one batch, the calls of trainers/vanilla_nerf_trainer.py:45-96 (loss_func: ray_to_samples,
    the coarse net, raw2outputs, MSE, the empty-space term, ray_to_importance_samples, the fine net, raw2outputs, MSE) and :206-248 (train_batch:
    the delay, backward, Adam, the learning-rate and penalty schedules) resolved through the namespace M (`ray_utils`, `render_utils`) -> [n_iter, 4] terms.
    The batch has no DataLoader axis (utils.remove_first_axis is the script's own plumbing)."""
    import torch
    import torch.nn.functional as F
    terms_log = []
    penalty = opt.penalize_empty_space
    color = batch['color'].to(device)
    for it in range(n_iter):
        optim.zero_grad()
        pts, dirs, z_vals = M.ray_utils.ray_to_samples(batch, opt.samples_per_ray, perturb=opt.perturb, append_t=None)
        pts, dirs, z_vals = pts.to(device), dirs.to(device), z_vals.to(device)
        n = pts.shape[1]
        out = coarse_net(pts, dirs)
        rgb_map, _, _, weights, _ = M.render_utils.raw2outputs(out, z_vals, dirs[:, 0, :], raw_noise_std=opt.raw_noise_std, white_bkg=opt.white_bkg)
        terms = [F.mse_loss(rgb_map, color), torch.zeros((), device=device)]
        if penalty > 0:
            closer = z_vals < (batch['depth'][:, None].repeat(1, n).to(device) * opt.margin)
            sel = out[closer][:, 3]
            terms[1] = terms[1] + F.mse_loss(torch.tanh(torch.relu(sel)), torch.zeros_like(sel)) * penalty
        F_pts, F_dirs, F_z = M.ray_utils.ray_to_importance_samples(batch, z_vals, weights, opt.importance_samples_per_ray, device=device, append_t=None)
        F_out = fine_net(F_pts, F_dirs)
        F_rgb, _, _, _, _ = M.render_utils.raw2outputs(F_out, F_z, F_dirs[:, 0, :], raw_noise_std=opt.raw_noise_std, white_bkg=opt.white_bkg)
        terms += [F.mse_loss(F_rgb, color), torch.zeros((), device=device)]
        if penalty > 0:
            F_closer = F_z < (batch['depth'][:, None].repeat(1, F_z.shape[1]).to(device) * opt.margin)
            sel = F_out[F_closer][:, 3]
            terms[3] = terms[3] + F.mse_loss(torch.tanh(torch.relu(sel)), torch.zeros_like(sel)) * penalty
        assert float(out.detach()[..., 3].max()) > 0.0 and float(F_out.detach()[..., 3].max()) > 0.0, "a dead network (:88-94: the restart) is not a case of this fixture"
        rgb_loss, empty = terms[0] + terms[2], terms[1] + terms[3]
        total = rgb_loss + empty if it >= opt.delay_iters else rgb_loss
        total.backward()
        optim.step()
        terms_log.append([float(t.detach()) for t in terms])
        lr = opt.learning_rate * (0.1 ** (it / (opt.lrate_decay * 1000)))
        for group in optim.param_groups:
            group['lr'] = lr
        penalty = opt.penalize_empty_space * max(0, 1 - (it / 60000))
    return np.asarray(terms_log, np.float64)
