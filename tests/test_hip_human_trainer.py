"""-m gpu: the human trainer's seven-term loss (neuman_hip.human_trainer, reference trainers/human_nerf_trainer.py:180-446) on the
device pieces: every term re-derived in float64 on the CPU from the iteration's own intermediates (network outputs, signed
distances, compositing inputs), gradients reaching every trained tensor -- human net, offset net AND the SMPL pose / shape /
alignment through the differentiable skinning -- and a few optimiser steps that lower the loss."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import compositing

pytestmark = pytest.mark.gpu


class TinyHumanNeRF(torch.nn.Module):
    """the attributes of models/human_nerf.py HumanNeRF that the trainer reads (:21-50, 92-122), on the synthetic body"""

    def __init__(self, device):
        super().__init__()
        from neuman_hip import smpl, synthetic, vanilla
        self.coarse_bkg_net, self.fine_bkg_net = synthetic.make_joiner(0).to(device), synthetic.make_joiner(1).to(device)
        self.coarse_human_net = synthetic.make_joiner(2, 'rotate').to(device)
        opt = synthetic.default_opt(offset_scale=0.05, offset_scale_type='linear')
        torch.manual_seed(3)
        self.offset_nets = torch.nn.ModuleList([vanilla.build_offset_net(opt).to(device)])
        self.body = smpl.SMPLDiff(synthetic.smpl_like_model(0), device)
        pose, betas, align = synthetic.smpl_like_frames(3, 0)
        al = np.stack([np.concatenate([align[f'{i:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1) for i in range(3)]).astype(np.float32)
        al[:, :3, :3] = np.eye(3)[None] * 1.0                     # upright body at the origin: rays below are aimed at it
        al[:, 3, :3] = 0.0
        self.poses = torch.nn.Parameter(torch.tensor(pose * 0.3, device=device))
        self.betas = torch.nn.Parameter(torch.tensor(betas * 0.3, device=device))
        self.alignments = torch.nn.Parameter(torch.tensor(al, device=device))
        self.scale = 1.0

    def vertex_forward(self, idx):
        return self.body.vertex_forward(self.poses[idx][None], self.betas[idx][None], self.alignments[idx], self.scale)


@pytest.fixture(scope="module")
def setup():
    from neuman_hip import human_trainer, ray_utils, synthetic
    dev = torch.device('cuda')
    net = TinyHumanNeRF(dev)
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
    cap = synthetic.SimpleCapture(48, 48, fx=110., c2w=synthetic.spherical_c2w(15., -5., 3.0), near=0.5, far=5.0)
    coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
    rng = np.random.default_rng(1)
    coords = coords[rng.choice(len(coords), 256, replace=False)]
    o, d = ray_utils.shot_rays(cap, coords)
    o, d = torch.tensor(o, dtype=torch.float32, device=dev), torch.tensor(d, dtype=torch.float32, device=dev)
    near, far = ray_utils.geometry_guided_near_far(o, d, world[0], 0.2)
    hit = near < far
    assert 30 < int(hit.sum()) < 240, int(hit.sum())
    near = torch.where(hit, near, torch.full_like(near, 2.0))          # the dataset gives miss rays a dummy human interval
    far = torch.where(hit, far, torch.full_like(far, 3.0))
    batch = {'origin': o, 'direction': d, 'bkg_near': torch.full((256, 1), cap.near['bkg'], device=dev), 'bkg_far': torch.full((256, 1), cap.far['bkg'], device=dev),
             'human_near': near[:, None].contiguous(), 'human_far': far[:, None].contiguous(), 'is_hit': hit, 'is_bkg': (~hit).long(),
             'color': torch.rand((256, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(5)), 'cur_view_f': 0.35, 'cap_id': 1, 'patch_counter': 0}
    opt = types.SimpleNamespace(samples_per_ray=24, importance_samples_per_ray=24, perturb=0.0, white_bkg=True, penalize_smpl_alpha=1.0,
                                penalize_symmetric_alpha=0.1, penalize_dummy=1.0, penalize_hard_surface=0.1, penalize_color_range=0.1, penalize_mask=0.01,
                                penalize_lpips=0.0, penalize_sharp_edge=0.1, penalize_outside_factor=2.0, dist_exponent=2.0)
    can_caps = [synthetic.SimpleCapture(32, 32, fx=40., c2w=synthetic.spherical_c2w(a, 0., 3.0)) for a in (0., 90., 200.)]
    loss = human_trainer.HumanNeRFLoss(opt, net, faces, (can_verts, faces), can_caps, interval_comp=0.8, seed=4)
    return types.SimpleNamespace(net=net, batch=batch, loss=loss, opt=opt, ht=human_trainer)


@pytest.mark.parametrize("fused", [True, False])
def test_seven_terms_and_their_gradients(setup, fused, monkeypatch):
    """(fused: the five regularisers as csrc/loss.hip's value-and-gradient kernels; False: NEUMAN_FUSED_LOSS=0's torch spelling of the same terms)"""
    from neuman_hip import loss_ops
    monkeypatch.setattr(loss_ops, 'FUSED', fused)
    S = setup
    for p in list(S.net.parameters()):
        p.grad = None
    torch.manual_seed(11)
    ld = S.loss.loss_func(S.batch)
    assert list(ld.keys()) == S.ht.LOSS_NAMES
    vals = {k: float(v.detach()) for k, v in ld.items()}
    print("[trainer]", " ".join(f"{k} {v:.4e}" for k, v in vals.items()))
    assert vals['lpips_loss'] == 0.0 and all(np.isfinite(v) for v in vals.values())
    assert all(vals[k] > 0 for k in ('fine_rgb_loss', 'color_range_reg', 'smpl_sym_reg', 'smpl_shape_reg', 'mask_loss', 'sparsity_reg'))
    L = S.loss.last
    # ---- rgb term: oracle merge + compositing of the iteration's own network outputs
    z = np.concatenate([L['fine_bkg_z_vals'].cpu().numpy(), L['human_z_vals'].cpu().numpy()], 1)
    raw = np.concatenate([L['fine_bkg_out'].cpu().numpy(), L['human_out'].detach().cpu().numpy()], 1)
    order = np.argsort(z, 1, kind='stable')
    o_rgb = compositing.raw2outputs(np.take_along_axis(raw, order[..., None], 1), np.take_along_axis(z, order, 1), S.batch['direction'].cpu().numpy())[0]
    hit = S.batch['is_hit'].cpu().numpy()
    ref = np.mean((o_rgb[hit].astype(np.float64) - S.batch['color'].cpu().numpy()[hit]) ** 2)
    assert abs(vals['fine_rgb_loss'] - ref) < 2e-5 * max(1.0, ref)
    # ---- mask term
    m = compositing.raw2outputs(L['human_out'].detach().cpu().numpy(), L['human_z_vals'].cpu().numpy(), S.batch['direction'].cpu().numpy())[2]
    ref = np.mean((np.clip(m, 0, 1).astype(np.float64) - hit.astype(np.float64)) ** 2) * S.opt.penalize_mask
    assert abs(vals['mask_loss'] - ref) < 1e-6
    # ---- shape term from the device's signed distances and network outputs (:305-343)
    sig = L['human_out'].detach().cpu().double().reshape(-1, 4)[:, 3]
    dh = L['dist_human'].cpu()
    ref = F.mse_loss(1 - torch.exp(-torch.relu(sig[dh < 0])), torch.ones(int((dh < 0).sum()), dtype=torch.float64)) * S.opt.penalize_smpl_alpha
    dd, do = L['dist_dummy'].cpu(), L['dummy_out'].detach().cpu().double().reshape(-1, 4)[:, 3]
    ref = ref + F.mse_loss(1 - torch.exp(-torch.relu(do[dd < 0])), torch.ones(int((dd < 0).sum()), dtype=torch.float64)) * S.opt.penalize_dummy
    w = torch.pow(dd[dd > 0].double().abs() * S.opt.penalize_outside_factor, S.opt.dist_exponent)
    ref = ref + ((1 - torch.exp(-torch.relu(do[dd > 0]))) * w).abs().mean() * S.opt.penalize_dummy
    assert (dh < 0).any() and (dd < 0).any() and (dd > 0).any()
    assert abs(vals['smpl_shape_reg'] - float(ref)) < 2e-5 * max(1.0, float(ref))
    # ---- sparsity term from the canonical render's mask and weights (:368-379)
    cm, cw = L['can_mask'].detach().cpu().double().clamp(0, 1), L['can_weights'].detach().cpu().double().clamp(0, 1)    # (:366-367)
    f = lambda x: torch.mean(-torch.log(torch.exp(-x.abs()) + torch.exp(-(1 - x).abs())) + S.ht.HARD_SURFACE_OFFSET)   # noqa: E731
    ref = f(cm) * S.opt.penalize_sharp_edge + f(cw) * S.opt.penalize_hard_surface
    assert abs(vals['sparsity_reg'] - float(ref)) < 1e-5
    # ---- gradients reach everything the reference trains
    sum(ld.values()).backward()
    for name, p in [("human net", S.net.coarse_human_net.nerf.pts_linears[0].weight), ("offset net", S.net.offset_nets[0].nerf.pts_linears[0].weight),
                    ("poses", S.net.poses), ("betas", S.net.betas), ("alignments", S.net.alignments)]:
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0, name
    assert float(S.net.poses.grad[0].abs().sum()) == 0 and float(S.net.poses.grad[1].abs().sum()) > 0     # only the batch's frame
    assert all(p.grad is None for p in S.net.coarse_bkg_net.parameters())                                   # the background is frozen (:196, 238)


def test_training_steps_lower_the_loss(setup):
    S = setup
    params = list(S.net.coarse_human_net.parameters()) + list(S.net.offset_nets.parameters()) + [S.net.poses, S.net.betas, S.net.alignments]
    optim = torch.optim.Adam(params, lr=5e-4)
    torch.manual_seed(12)
    totals = []
    for _ in range(8):
        _, total = S.loss.train_step(S.batch, optim)
        totals.append(total)
    print("[trainer] totals", " ".join(f"{t:.4f}" for t in totals))
    assert np.isfinite(totals).all() and min(totals[4:]) < totals[0]


def test_trainer_loop_schedules_and_checkpoint(setup, tmp_path):
    """HumanNeRFTrainer: train_batch's grouping / delay / schedules (human_nerf_trainer.py:540-601) and the checkpoint round trip"""
    S = setup
    opt = types.SimpleNamespace(**{**vars(S.opt), 'delay_iters': 2, 'lrate_decay': 1, 'learning_rate': 5e-4, 'smpl_lr': 3e-4, 'prior_knowledge_decay': True,
                                   'offset_lim': 1.0, 'offset_scale': 0.05, 'offset_delay': 3, 'max_iter': 5, 'valid_iter': 0, 'out': str(tmp_path / 'human'),
                                   'resume': False, 'load_weights': False})
    optim = torch.optim.Adam([{"params": [S.net.poses], "lr": opt.smpl_lr}, {"params": S.net.coarse_human_net.parameters(), "lr": opt.learning_rate},
                              {"params": S.net.offset_nets.parameters(), "lr": opt.learning_rate}])
    masked = []

    def mask(cap_id):                                             # stands in for the DensePose visibility mask: freeze every joint but the root
        masked.append(cap_id)
        m = np.zeros((24, 3), np.float32)
        m[0] = 1
        return m
    tr = S.ht.HumanNeRFTrainer(opt, S.net, optim, S.loss.faces, S.loss.can_mesh, S.loss.can_caps, batches=lambda: S.batch, pose_grad_mask=mask,
                               interval_comp=0.8, seed=4)
    pose_before = S.net.poses.detach().clone()
    log = []
    torch.manual_seed(13)
    tr.train(on_step=lambda it, rep: log.append((it, rep, S.net.offset_nets[0].nerf.scale)))
    assert [it for it, _, _ in log] == [0, 1, 2, 3, 4, 5] and masked == [1] * 6
    for it, rep, _ in log:
        can = rep['smpl_sym_reg'] + rep['smpl_shape_reg']
        rgb = rep['fine_rgb_loss'] + rep['color_range_reg'] + rep['lpips_loss']
        want = can + rep['mask_loss'] + rep['sparsity_reg'] + (rgb if it >= 2 else 0.0)          # photometric terms join at delay_iters
        assert rep['total_loss'] == pytest.approx(want, rel=1e-5) and rep['rgb_loss'] == pytest.approx(rgb, rel=1e-6)
    k = 0.1 ** (5 / 1000)
    assert optim.param_groups[0]['lr'] == pytest.approx(3e-4 * k, rel=1e-12) and optim.param_groups[2]['lr'] == pytest.approx(5e-4 * k, rel=1e-12)
    keep = 1 - 5 / 60000
    assert tr.penalize_mask == pytest.approx(S.opt.penalize_mask * keep) and tr.penalize_dummy == pytest.approx(S.opt.penalize_dummy * keep)
    scales = [s for _, _, s in log]                                # set AFTER each step: 0 until offset_delay, then growing from offset_scale
    assert scales[:3] == [0, 0, 0] and scales[3] == pytest.approx(0.05) and scales[5] == pytest.approx(0.05 + 0.95 * 2 / 60000)
    moved = (S.net.poses.detach() - pose_before)[1].reshape(24, 3).abs().sum(1)
    assert float(moved[0]) > 0 and float(moved[1:].sum()) == 0     # the mask froze the other joints; other frames got no gradient at all
    assert float((S.net.poses.detach() - pose_before)[0].abs().sum()) == 0
    rep = tr.validate(n_batches=1)
    assert S.net.coarse_human_net.training and np.isfinite(rep['total_loss'])
    ckpt = torch.load(tmp_path / 'human' / 'checkpoint.pth.tar', map_location='cpu', weights_only=False)
    assert set(ckpt) == {'epoch', 'iteration', 'optim_state_dict', 'hybrid_model_state_dict'} and ckpt['iteration'] == 5
    assert any(k.startswith('coarse_human_net.') for k in ckpt['hybrid_model_state_dict']) and 'poses' in ckpt['hybrid_model_state_dict']


def test_shape_regulariser_uses_the_frames_own_canonical_mesh():
    """human_nerf_trainer.py:308 takes captures[batch['cap_id']].can_mesh: with one canonical body per frame (dict / list / callable)
    the inside / outside test must change with cap_id; with one (verts, faces) tuple it must not"""
    import types
    from neuman_hip import human_trainer, synthetic
    verts, faces = synthetic.capsule_mesh(12, 16)
    pts = torch.tensor(np.random.default_rng(0).uniform(-0.6, 0.6, (4000, 3)).astype(np.float32), device='cuda')
    opt = types.SimpleNamespace(penalize_smpl_alpha=1.0, penalize_symmetric_alpha=0.0, penalize_dummy=0.0, penalize_hard_surface=0.0, penalize_color_range=0.0,
                                penalize_mask=0.0, penalize_lpips=0.0, penalize_sharp_edge=0.0)
    fat = (verts * np.array([1.6, 1.0, 1.6], np.float32), faces)
    for per_frame in ({0: (verts, faces), 1: fat}, [(verts, faces), fat], lambda i: (verts, faces) if i == 0 else fat):
        L = human_trainer.HumanNeRFLoss(opt, None, faces, per_frame, [])
        d0, d1 = L._signed_distance(pts, 0), L._signed_distance(pts, 1)
        assert int((d1 < 0).sum()) > int((d0 < 0).sum()) > 0 and len(L._can_tree) == 2
        assert torch.equal(d0, L._signed_distance(pts, 0))
    L = human_trainer.HumanNeRFLoss(opt, None, faces, (verts, faces), [])
    assert torch.equal(L._signed_distance(pts, 0), L._signed_distance(pts, 1)) and len(L._can_tree) == 1


def _human_state(S):
    return {k: v.detach().clone() for k, v in list(S.net.coarse_human_net.state_dict().items()) + [('off.' + k, v) for k, v in S.net.offset_nets.state_dict().items()]}


def test_nan_density_is_not_a_dead_network(setup):
    """the reference's restart test is `max <= 0.0` (human_nerf_trainer.py:437) -- False for NaN, so a NaN batch only trips the NaN
    guard (:476-478) and the trained networks survive.  Both forms of the flag (direct, and deferred to the read-back)."""
    S = setup
    saved = {k: v.clone() for k, v in S.net.coarse_human_net.state_dict().items()}
    try:
        with torch.no_grad():
            S.net.coarse_human_net.nerf.alpha_linear.bias.fill_(float('nan'))
        before = _human_state(S)
        resets = []
        orig = S.loss._reset_dead_networks
        S.loss._reset_dead_networks = lambda: resets.append(1) or orig()
        try:
            ld = S.loss.loss_func(S.batch)                                  # direct form
            assert not resets and not np.isfinite(float(sum(ld.values()).detach()))
            S.loss.defer_dead_check = True
            try:
                S.loss.loss_func(S.batch)
            finally:
                S.loss.defer_dead_check = False
            assert bool(S.loss.last['alive']) and not resets
        finally:
            S.loss._reset_dead_networks = orig
        after = _human_state(S)
        assert all(torch.equal(before[k], after[k]) or (torch.isnan(before[k]).any() and torch.isnan(after[k]).any()) for k in before)
    finally:
        S.net.coarse_human_net.load_state_dict(saved)


def test_dead_network_restart_leaves_the_fresh_weights_alone(setup):
    """on a dead network the reference's losses are fresh zero tensors -- no parameter gets a gradient and Adam's step() skips them
    all (:437-442).  The deferred form multiplies the built loss by 0 and backward() fills zero gradients: they must be dropped, or the OLD
    moments would move the re-initialised weights."""
    S = setup
    saved_h = {k: v.clone() for k, v in S.net.coarse_human_net.state_dict().items()}
    saved_o = {k: v.clone() for k, v in S.net.offset_nets.state_dict().items()}
    try:
        params = list(S.net.coarse_human_net.parameters()) + list(S.net.offset_nets.parameters())
        optim = torch.optim.Adam(params, lr=1e-2)
        torch.manual_seed(14)
        for _ in range(2):                                                  # Adam has moments now
            S.loss.train_step(S.batch, optim)
        with torch.no_grad():
            S.net.coarse_human_net.nerf.alpha_linear.weight.zero_()
            S.net.coarse_human_net.nerf.alpha_linear.bias.fill_(-5.0)       # sigma <= 0 everywhere: dead
        snap = {}
        orig = S.loss._reset_dead_networks

        def spy():
            orig()
            snap.update(_human_state(S))
        S.loss._reset_dead_networks = spy
        try:
            terms, total = S.loss.train_step(S.batch, optim)
        finally:
            S.loss._reset_dead_networks = orig
        assert snap and total == 0.0 and all(v == 0.0 for v in terms.values())
        after = _human_state(S)
        assert all(torch.equal(snap[k], after[k]) for k in snap)              # step() moved nothing
        assert all(p.grad is None for p in params)
        assert float(S.net.coarse_human_net.nerf.alpha_linear.bias.detach().abs().max()) < 1.0      # really re-initialised
    finally:
        S.net.coarse_human_net.load_state_dict(saved_h)
        S.net.offset_nets.load_state_dict(saved_o)
