"""The human trainer's loss on the device pieces: reference trainers/human_nerf_trainer.py:180-446.

One training iteration of the NeuMan human model is, per batch of rays of one frame:

    background samples   coarse pass -> importance samples -> fine pass, both nets frozen (:180-239)   the RENDERING kernels
                                                                                                        (fp16x3 / i8x3, no autograd)
    human samples        ray_to_samples, offset net, differentiable skinning (pose refinement),         nm_ray_to_samples, MFMA GEMMs      
                         warp to canonical space with T^-1 of the closest surface point, human net      (neuman_hip/train.py), tree search
                                                                                          (:241-278)    (nm_signed_distance), SMPLDiff
    seven loss terms     rgb, lpips, colour range, symmetry, smpl shape, mask, sparsity (:382-446)       differentiable compositing
                                                                                                        (nm_composite / nm_composite_backward)

`HumanNeRFLoss.loss_func(batch)` returns the reference's `loss_dict` (same seven names, LOSS_NAMES); `train_step` adds the backward
pass and the optimiser step of `train_batch` (:470-494 of the reference).  The elementwise algebra between kernels (mse, tanh, exp,
clip: a few thousand values per iteration) is torch on the device, as in the reference.  Random draws (offset net choice, dummy
directions / points, the canonical camera and its pixels) come from `self.rng` / the device generator so that a test can replay them.
"""
import math
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import loss_ops, ray_utils, render_utils
from .train import upload as train_upload

LOSS_NAMES = ['fine_rgb_loss', 'lpips_loss', 'color_range_reg', 'smpl_sym_reg', 'smpl_shape_reg', 'mask_loss', 'sparsity_reg']   # :31-39
HARD_SURFACE_OFFSET = 0.31326165795326233          # utils/constant.py:7
PATCH_SIZE = 32                                    # :8
CANONICAL_CAMERA_DIST = 3.0                        # :13


BATCH_NET_CALLS = os.environ.get('NEUMAN_BATCH_NET_CALLS', '1') != '0'     # the iteration's five human-network evaluations as one call


def _occupancy(raw):
    """1 - exp(-relu(sigma)) of a raw network output [..., 4]: the opacity of a unit interval"""
    return 1 - torch.exp(-torch.relu(raw.reshape(-1, 4)[:, 3]))


def _bimodal_prior(x):
    """mean(-log(e^-|x| + e^-|1-x|)) + HARD_SURFACE_OFFSET: zero-mean pull of x towards 0 or 1 (the trainer's sharp-edge and
    hard-surface terms, human_nerf_trainer.py:368-379)"""
    return torch.mean(-torch.log(torch.exp(-x.abs()) + torch.exp(-(1 - x).abs())) + HARD_SURFACE_OFFSET)


def _fused(*tensors):
    """the regularisers as single value-and-gradient kernels (loss_ops) when the term's inputs live on the GPU"""
    return loss_ops.FUSED and all(t is None or t.is_cuda for t in tensors)


def _unit(v):
    return v / torch.norm(v, dim=-1, keepdim=True)


class HumanNeRFLoss:
    """opt carries the reference's option names (samples_per_ray, importance_samples_per_ray, perturb, white_bkg, penalize_*,
    penalize_outside_factor, dist_exponent).  `net` is a HumanNeRF-like holder: coarse_bkg_net, fine_bkg_net, coarse_human_net
    (Joiners), offset_nets (list), and vertex_forward(cap_id) -> (world_verts [1,V,3], T [1,V,4,4]) with autograd (SMPLDiff).
    `faces` [F,3] are the body's triangles, `can_mesh` = (verts [V,3], faces) the canonical (da-pose) body for the shape
    regulariser -- or, as the reference keeps one per frame (captures[cap_id].can_mesh, built from that frame's betas:
    human_nerf_trainer.py:308), a dict / list / callable cap_id -> (verts, faces) -- `can_caps` the canonical cameras of the sparsity
    regulariser (:157-172)."""

    def __init__(self, opt, net, faces, can_mesh, can_caps, interval_comp=1.0, lpips_loss_fn=None, seed=0):
        self.opt, self.net, self.faces, self.can_mesh, self.can_caps = opt, net, np.ascontiguousarray(np.asarray(faces)[:, :3], np.int32), can_mesh, can_caps
        self.interval_comp, self.lpips_loss_fn = interval_comp, lpips_loss_fn
        for k in ('penalize_smpl_alpha', 'penalize_symmetric_alpha', 'penalize_dummy', 'penalize_hard_surface', 'penalize_color_range',
                  'penalize_mask', 'penalize_lpips', 'penalize_sharp_edge'):
            setattr(self, k, getattr(opt, k))                                            # :143-152
        self.rng = random.Random(seed)
        self.np_rng = np.random.default_rng(seed)
        # tests: the random draws of one loss_func call given instead of drawn -- {'offset_net': i, 'dummy_dirs_randn': [R,S,3] (before
        # normalisation), 'dummy_pts_rand': [R,S,3] in [0, 1), 'can_cap': i, 'can_pixel_choice': [128] indices into np.argwhere(np.ones(shape))}
        # -- so that a recording of the reference's own run (tests/golden/make_golden_human_loss.py) can be replayed term by term
        self.replay = None
        self.last = {}                                                                   # intermediates of the last call (tests, logging)
        self._can_tree = {}                                                              # cap_id (None: the one shared mesh) -> search tree

    # ---- :180-239: frozen background, rendering kernels, nothing kept for autograd
    def _eval_bkg_samples(self, batch, device):
        o, d = batch['origin'].to(device, torch.float32).contiguous(), batch['direction'].to(device, torch.float32).contiguous()
        near, far = batch['bkg_near'].to(device, torch.float32).reshape(-1).contiguous(), batch['bkg_far'].to(device, torch.float32).reshape(-1).contiguous()
        with torch.no_grad():
            raw, z = render_utils.bkg_pass_rays_fused(self.net.coarse_bkg_net, self.net.fine_bkg_net, o, d, near, far, self.opt.samples_per_ray,
                                                      self.opt.importance_samples_per_ray, self.opt.white_bkg)
        return d, z, raw

    # ---- :241-278
    def _eval_human_samples(self, batch, device):
        human_batch = {'origin': batch['origin'].to(device), 'direction': batch['direction'].to(device),          # (ray_to_samples writes none of them)
                       'near': batch['human_near'].to(device), 'far': batch['human_far'].to(device)}
        human_pts, human_dirs, human_z_vals = ray_utils.ray_to_samples(human_batch, self.opt.samples_per_ray, device=device, perturb=self.opt.perturb)
        b, n, _ = human_pts.shape
        cur_time = torch.full_like(human_pts[..., 0:1], float(batch['cur_view_f']))
        nets = list(self.net.offset_nets)
        offset_net = nets[int(self.replay['offset_net'])] if self.replay else self.rng.choice(nets)
        offset = offset_net(torch.cat([human_pts, cur_time], dim=-1), const_time=float(batch['cur_view_f']))
        mesh, raw_Ts = self.net.vertex_forward(int(batch['cap_id']))                      # autograd: pose / shape / alignment refinement
        flat = human_pts.reshape(-1, 3)
        # :262-266: T_interp_inv [p; 1] with T_interp the barycentric blend of the closest triangle's vertex transforms -- one fused
        # differentiable call (blend, inverse and product; ray_utils.warp_samples_to_canonical_diff is the reference-shaped form)
        can_flat, _, _ = ray_utils.warp_points_to_canonical_diff(flat.detach(), mesh[0], self.faces, raw_Ts[0])
        can_pts = can_flat.reshape(b, n, 3)
        can_pts = can_pts + offset
        step = can_pts[:, 1:] - can_pts[:, :-1]                                          # view direction of a warped sample: towards the next one
        can_dirs = _unit(torch.cat([step, step[:, -1:]], dim=1))
        return human_pts, human_dirs, human_z_vals, can_pts, can_dirs

    def _human_net(self, queries):
        """coarse_human_net on every (points, directions) set of an iteration in ONE call.  The reference calls the network five times
        (:276 the rays' samples, :286 / :299 / :331 / :364 the regularisers' points); a sample's output does not depend on what else is in
        the batch, and `cat` / `split` are differentiable, so one forward, one backward-data chain and one set of backward-weights products
        serve all of them (a fifth of the launches; the parameter gradients are the same sums in another order).  BATCH_NET_CALLS = False:
        one call per set."""
        net = self.net.coarse_human_net
        outs = [None] * len(queries)
        # a query set that asks about the FIRST set's points again (the colour-range term: the same tensor, other directions) does not need the trunk a
        # second time: its colours come from the first call's feature vectors through the views head alone (Joiner.forward_two_views; the gradients are
        # the sums the separate calls would give)
        twin = next((i for i in range(1, len(queries)) if queries[i][0] is queries[0][0]), None)
        if twin is not None and hasattr(net, 'forward_two_views'):
            pair = net.forward_two_views(queries[0][0], queries[0][1], queries[twin][1])
            if pair is not None:
                outs[0], outs[twin] = pair
        rest = [i for i in range(len(queries)) if outs[i] is None]
        if not BATCH_NET_CALLS or len(rest) == 1:
            for i in rest:
                outs[i] = net(*queries[i])
        elif rest:
            sizes = [queries[i][0].shape[0] for i in rest]
            out = net(torch.cat([queries[i][0] for i in rest], 0), torch.cat([queries[i][1] for i in rest], 0))
            for i, o in zip(rest, torch.split(out, sizes, 0)):
                outs[i] = o
        return outs

    # ---- :280-290
    def _color_range_query(self, pts, dirs):
        draw = torch.as_tensor(self.replay['dummy_dirs_randn']).to(dirs) if self.replay else torch.randn_like(dirs)
        return pts, _unit(draw)                                                           # the same points seen from random directions

    def _color_range_regularization(self, other_view, tgts):
        if _fused(other_view, tgts):
            return loss_ops.color_range(other_view, tgts, self.penalize_color_range)
        rgb = lambda raw: torch.sigmoid(raw.reshape(-1, 4)[:, :3])                       # noqa: E731
        return self.penalize_color_range * F.mse_loss(rgb(other_view), rgb(tgts))

    # ---- :292-304
    def _smpl_symmetry_query(self, pts, dirs):
        if self._mirror is None or self._mirror.device != pts.device:                   # the canonical body is left-right symmetric in x
            self._mirror = torch.tensor([-1.0, 1.0, 1.0], device=pts.device)
        return pts.detach() * self._mirror, dirs.detach()                                      # (dummy directions: only the occupancy is compared)

    def _smpl_symmetry_regularization(self, mirrored, tgts):
        if _fused(mirrored, tgts):
            return loss_ops.symmetry(mirrored, tgts, self.penalize_symmetric_alpha)
        squash = lambda raw: torch.tanh(torch.relu(raw[..., 3]))                        # noqa: E731
        return self.penalize_symmetric_alpha * F.mse_loss(squash(tgts), squash(mirrored))

    def _signed_distance(self, pts, cap_id=None):
        """igl.signed_distance of the reference (:310, :326) on the device: negative inside the canonical body of frame `cap_id`
        (:308; one search tree per canonical mesh, built on first use).

        `can_mesh` is either ONE mesh -- a (verts [V,3], faces [F,>=3]) pair -- or PER-FRAME meshes: a dict {cap_id: (verts, faces)}
        or a callable cap_id -> (verts, faces).  (A list / tuple of per-frame pairs is taken as a dict over its indices; anything whose
        first element is not a [V,3] array is rejected rather than guessed at.)  At most `_CAN_TREE_MAX` trees stay on the device."""
        cm = self.can_mesh
        if callable(cm) or isinstance(cm, dict):
            per_frame = True
        elif isinstance(cm, (tuple, list)) and len(cm) == 2 and hasattr(cm[0], 'shape') and len(cm[0].shape) == 2 and cm[0].shape[-1] == 3:
            per_frame = False
        elif isinstance(cm, (tuple, list)) and len(cm) > 0 and isinstance(cm[0], (tuple, list)) and len(cm[0]) == 2:
            per_frame = True
        else:
            raise ValueError("can_mesh must be (verts [V,3], faces) or per-frame meshes as a dict / list of such pairs / callable cap_id -> pair")
        if per_frame and cap_id is None:
            raise ValueError("per-frame canonical meshes need the batch's cap_id")
        key = int(cap_id) if per_frame else None
        if key not in self._can_tree:
            verts, faces = (cm(key) if callable(cm) else cm[key]) if per_frame else cm
            v = torch.as_tensor(np.ascontiguousarray(verts, dtype=np.float32))
            if v.ndim != 2 or v.shape[1] != 3:
                raise ValueError(f"canonical mesh vertices must be [V,3], got {tuple(v.shape)}")
            while len(self._can_tree) >= self._CAN_TREE_MAX:                             # (oldest first: dicts keep insertion order)
                self._can_tree.pop(next(iter(self._can_tree)))
            self._can_tree[key] = ray_utils.Mesh(v, np.ascontiguousarray(np.asarray(faces)[:, :3], np.int32),
                                                 torch.zeros((v.shape[0], 16), dtype=torch.float64), pts.device)
        return ray_utils.signed_distance_dev(pts.reshape(-1, 3).detach(), self._can_tree[key])[0]

    _CAN_TREE_MAX = 8
    _mirror = None

    # ---- :305-343
    def _dummy_points(self, pts):
        if not self.penalize_dummy > 0:
            return None                                                                   # random points of a 3-unit box around the canonical body
        return ((torch.as_tensor(self.replay['dummy_pts_rand']).to(pts) if self.replay else torch.rand_like(pts)) - 0.5) * 3

    def _smpl_shape_regularization(self, batch, pts, pred, dummy_pts, dummy_out):
        # both signed-distance queries of the iteration (:310, :326) go against the same canonical body: ONE search launch
        both = self._signed_distance(pts if dummy_pts is None else torch.cat([pts.reshape(-1, 3), dummy_pts.reshape(-1, 3)], 0), batch.get('cap_id'))
        n_h = pts.reshape(-1, 3).shape[0]
        dist_human, dist_dummy = both[:n_h], (None if dummy_pts is None else both[n_h:])
        self.last.update(dist_human=dist_human)
        if dummy_pts is not None:
            self.last.update(dummy_pts=dummy_pts, dist_dummy=dist_dummy, dummy_out=dummy_out)
        if _fused(pred, dummy_out):
            return loss_ops.shape_prior(pred, dist_human, dummy_out, dist_dummy, self.penalize_smpl_alpha, self.penalize_dummy,
                                        self.opt.penalize_outside_factor, self.opt.dist_exponent)
        smpl_reg = torch.zeros((), device=pts.device)

        def masked_mean(x, mask):                                                        # x[mask].mean(), 0 for an empty mask -- without asking the
            m = mask.to(x.dtype)                                                         # host whether it is empty (a synchronisation per question)
            return (x * m).sum() / m.sum().clamp_min(1.0)

        def filled(raw, mask, weight):                                                   # occupancy 1 where the body is
            return weight * masked_mean((1 - _occupancy(raw)) ** 2, mask)

        smpl_reg = smpl_reg + filled(pred, dist_human < 0, self.penalize_smpl_alpha)
        if dummy_pts is not None:
            smpl_reg = smpl_reg + filled(dummy_out, dist_dummy < 0, self.penalize_dummy)
            outside = dist_dummy > 0                                                     # occupancy 0 outside, weighted by the distance from the surface
            falloff = (dist_dummy.abs() * self.opt.penalize_outside_factor) ** self.opt.dist_exponent
            smpl_reg = smpl_reg + self.penalize_dummy * masked_mean((_occupancy(dummy_out) * falloff).abs(), outside)
        return smpl_reg

    # ---- :345-380
    def _sparsity_query(self, device):
        num_can_rays = 128
        can_cap = self.can_caps[int(self.replay['can_cap'])] if self.replay else self.rng.choice(self.can_caps)
        coords = np.argwhere(np.ones(can_cap.shape))
        coords = coords[np.asarray(self.replay['can_pixel_choice']) if self.replay else self.np_rng.integers(0, len(coords), num_can_rays)][:, ::-1]
        can_orig, can_dir = ray_utils.shot_rays(can_cap, coords)
        can_pts, can_dirs, can_z_vals = ray_utils.ray_to_samples(
            # (the rays change every iteration: through pinned memory, so that the copy does not wait for the kernels queued before it)
            {'origin': train_upload(can_orig, device), 'direction': train_upload(can_dir, device),
             'near': torch.zeros((num_can_rays, 1), device=device), 'far': torch.full((num_can_rays, 1), CANONICAL_CAMERA_DIST * 1.667, device=device)},
            samples_per_ray=self.opt.samples_per_ray, device=device, perturb=self.opt.perturb)
        return can_pts, can_dirs, can_z_vals

    def _sparsity_regularization(self, can_out, can_dirs, can_z_vals):
        sparsity_reg = torch.zeros((), device=can_out.device)
        can_out = torch.cat([can_out[..., :3], can_out[..., 3:] * self.interval_comp], -1)            # `can_out[..., -1] *= interval_comp`, out of place
        _, _, can_mask, can_weights, _ = render_utils.raw2outputs(can_out, can_z_vals, can_dirs[:, 0, :].contiguous(), white_bkg=True)
        self.last.update(can_mask=can_mask, can_weights=can_weights)                     # (before the clamp to [0, 1] of :366-367)
        if _fused(can_mask, can_weights):                                                # clamp, prior, mean and the gradient of all three: one kernel a term
            if self.penalize_sharp_edge > 0:
                sparsity_reg = self.penalize_sharp_edge * loss_ops.bimodal_prior(can_mask, HARD_SURFACE_OFFSET)
            if self.penalize_hard_surface > 0:
                hard = self.penalize_hard_surface * loss_ops.bimodal_prior(can_weights, HARD_SURFACE_OFFSET)
                sparsity_reg = sparsity_reg + hard if self.penalize_sharp_edge > 0 else hard
            return sparsity_reg
        can_weights, can_mask = can_weights.clamp(0.0, 1.0), can_mask.clamp(0.0, 1.0)
        if self.penalize_sharp_edge > 0:                                                 # silhouettes: a ray is inside or outside
            sparsity_reg = sparsity_reg + self.penalize_sharp_edge * _bimodal_prior(can_mask)
        if self.penalize_hard_surface > 0:                                               # surfaces: a sample carries all of the weight or none
            sparsity_reg = sparsity_reg + self.penalize_hard_surface * _bimodal_prior(can_weights)
        return sparsity_reg

    # ---- :382-446
    def loss_func(self, batch, return_rgb=False):
        device = next(self.net.coarse_human_net.parameters()).device
        zero = torch.zeros((), device=device)
        loss_dict = {name: zero for name in LOSS_NAMES}                                  # (a term that is switched off stays this zero)
        self.last = {}
        is_hit = batch['is_hit'].to(device).bool()
        fine_bkg_dir, fine_bkg_z_vals, fine_bkg_out = self._eval_bkg_samples(batch, device)
        _, human_dirs, human_z_vals, can_pts, can_dirs = self._eval_human_samples(batch, device)
        # every set of points the human network is asked about in this iteration, in the order the reference draws its random numbers
        # (:286 directions, :317 dummy points, :347-362 canonical rays), then ONE evaluation
        queries, slot = [(can_pts, can_dirs)], {}
        if self.penalize_symmetric_alpha > 0:
            slot['sym'] = len(queries)
            queries.append(self._smpl_symmetry_query(can_pts, can_dirs))
        if self.penalize_color_range > 0:
            slot['color'] = len(queries)
            queries.append(self._color_range_query(can_pts, can_dirs))
        dummy_pts = self._dummy_points(can_pts) if self.penalize_smpl_alpha > 0 else None
        if dummy_pts is not None:
            slot['dummy'] = len(queries)
            queries.append((dummy_pts, can_dirs))
        sparse = None
        if self.penalize_sharp_edge > 0 or self.penalize_hard_surface > 0:
            sparse = self._sparsity_query(device)
            slot['sparse'] = len(queries)
            queries.append((sparse[0], sparse[1]))
        outs = self._human_net(queries)
        human_out = outs[0]
        if 'sym' in slot:
            loss_dict['smpl_sym_reg'] = self._smpl_symmetry_regularization(outs[slot['sym']], human_out)
        if 'color' in slot:
            loss_dict['color_range_reg'] = self._color_range_regularization(outs[slot['color']], human_out)
        if self.penalize_mask > 0:
            _, _, human_mask, _, _ = render_utils.raw2outputs(human_out, human_z_vals, human_dirs[:, 0, :].contiguous(), white_bkg=self.opt.white_bkg)
            loss_dict['mask_loss'] = F.mse_loss(torch.clamp(human_mask, min=0.0, max=1.0),
                                                                         (1 - batch['is_bkg'].to(device)).float()) * self.penalize_mask
        if self.penalize_smpl_alpha > 0:
            loss_dict['smpl_shape_reg'] = self._smpl_shape_regularization(
                batch, can_pts, human_out, dummy_pts, outs[slot['dummy']] if 'dummy' in slot else None)
        if sparse is not None:
            loss_dict['sparsity_reg'] = self._sparsity_regularization(outs[slot['sparse']], sparse[1], sparse[2])
        # RGB loss: the two sample lists merged by depth (:415-422), composited once (:423-428)
        fine_total_zvals, fine_order = torch.sort(torch.cat([fine_bkg_z_vals, human_z_vals], -1), -1)
        fine_total_out = torch.gather(torch.cat([fine_bkg_out, human_out], 1), 1, fine_order[..., None].expand(-1, -1, 4))
        fine_rgb_map, _, _, _, _ = render_utils.raw2outputs(fine_total_out, fine_total_zvals, fine_bkg_dir, white_bkg=self.opt.white_bkg)
        color = batch['color'].to(device)
        # mse over the hit rays (:429): a masked mean, so that the host need not wait for the index list of the hits
        hit_w = is_hit.to(fine_rgb_map.dtype)[:, None]
        loss_dict['fine_rgb_loss'] = (((fine_rgb_map - color) ** 2) * hit_w).sum() / (hit_w.sum() * fine_rgb_map.shape[1]).clamp_min(1.0)
        if self.penalize_lpips > 0 and int(batch.get('patch_counter', 0)) == 1 and self.lpips_loss_fn is not None:   # :431-435
            n = PATCH_SIZE * PATCH_SIZE
            a = fine_rgb_map[:n].reshape(PATCH_SIZE, PATCH_SIZE, -1).permute(2, 0, 1) * 2 - 1
            b = color[:n].reshape(PATCH_SIZE, PATCH_SIZE, -1).permute(2, 0, 1) * 2 - 1
            loss_dict['lpips_loss'] = loss_dict['lpips_loss'] + (self.lpips_loss_fn(a, b) * self.penalize_lpips).flatten()[0]
        self.last.update(human_out=human_out, can_pts=can_pts, can_dirs=can_dirs, human_z_vals=human_z_vals, fine_bkg_out=fine_bkg_out,
                         fine_bkg_z_vals=fine_bkg_z_vals, fine_rgb_map=fine_rgb_map, is_hit=is_hit)
        # :437-442: a dead network (no positive density anywhere) is re-initialised and the iteration's losses are zero.  `defer_dead_check`
        # (train_step): the flag stays on the device -- the losses are multiplied by it, the host reads it together with the loss values after
        # the backward pass and re-initialises then (before the optimiser step, as the reference does) -- so that the loss is built without
        # a host synchronisation
        # the reference's test is `max <= 0.0` (:437): a NaN density (torch.max propagates it) is NOT a dead network -- the iteration is then
        # the NaN guard's to drop (:476-478), the trained weights stay
        alive = ~(human_out[..., 3].detach().max() <= 0.0)
        if self.defer_dead_check:
            self.last['alive'] = alive
            terms = torch.stack([loss_dict[name] for name in LOSS_NAMES])
            terms = torch.where(alive, terms, torch.zeros_like(terms))                   # (where, not x 0: an inf loss of a dead network stays 0)
            self.last['terms'] = terms                                                   # (the seven as one tensor: one sum, one read-back)
            loss_dict = dict(zip(LOSS_NAMES, terms.unbind(0)))
        elif not bool(alive):
            self._reset_dead_networks()
            loss_dict = {name: torch.zeros((), device=device, requires_grad=True) for name in LOSS_NAMES}
        return (loss_dict, fine_rgb_map) if return_rgb else loss_dict

    defer_dead_check = False

    def _reset_dead_networks(self):
        from .vanilla import weight_reset
        for m in list(self.net.offset_nets) + [self.net.coarse_human_net]:
            m.apply(weight_reset)

    def train_step(self, batch, optimizer):
        """train_batch (:470-494): zero_grad, loss, backward, step -> {name: float}, total"""
        optimizer.zero_grad()
        self.defer_dead_check = True
        try:
            loss_dict = self.loss_func(batch)
        finally:
            self.defer_dead_check = False
        total = self.last['terms'].sum()
        total.backward()
        vals = torch.cat([self.last['terms'].detach(), torch.stack([total.detach(), self.last['alive'].to(total.dtype)])]).tolist()   # ONE read-back
        if vals[-1] == 0.0:
            # :437-442: the reference's zero losses are detached from the graph, so no parameter has a gradient and step() skips every one of
            # them -- here backward() has filled zeros: drop them, or Adam's moments of the OLD weights would move the re-initialised ones
            self._reset_dead_networks()
            optimizer.zero_grad(set_to_none=True)
        optimizer.step()
        return dict(zip(loss_dict.keys(), vals[:-2])), vals[-2]


# DensePose body-part labels (1..24) that show a limb, and the SMPL joints whose pose gradient is zeroed when none of them is in
# the frame's DensePose map (human_nerf_trainer.py:40-105: an occluded limb's rotation is not refined from that frame)
_LIMB_LABELS_TO_JOINTS = (
    ((8, 10), (1,)), ((7, 9), (2,)),            # upper leg left / right -> hips
    ((12, 14), (4,)), ((11, 13), (5,)),         # lower leg left / right -> knees
    ((5,), (7, 10)), ((6,), (8, 11)),           # foot left / right -> ankle, toes
    ((15, 17), (16,)), ((16, 18), (17,)),       # upper arm left / right -> shoulders
    ((19, 21), (18,)), ((20, 22), (19,)),       # lower arm left / right -> elbows
    ((4,), (20, 22)), ((3,), (21, 23)),         # hand left / right -> wrist, fingers
    ((23, 24), (12, 15)),                       # head -> neck, head
)


def densepose_pose_mask(dp_mask):
    """turn_smpl_gradient_off (human_nerf_trainer.py:70-105): [72] multipliers for one frame's pose gradient, 0 for the three
    axis-angle components of every joint whose limb has no pixel in the DensePose label map `dp_mask`."""
    assert dp_mask is not None
    seen = set(int(v) for v in np.unique(dp_mask))
    mask = np.ones((24, 3))
    for labels, joints in _LIMB_LABELS_TO_JOINTS:
        if not any(l in seen for l in labels):
            mask[list(joints)] = 0
    return mask.reshape(-1)


class HumanNeRFTrainer(HumanNeRFLoss):
    """The loop around the loss (human_nerf_trainer.py:447-494, 540-601, 636-680 + base_trainer.py:66-108): loss grouping and the
    photometric delay, NaN guard, optimiser step, the schedules of the learning rates (parameter group 0 = the SMPL parameters at
    smpl_lr, groups 1-2 = the networks at learning_rate), of the prior penalties and of the offset nets' scale, and checkpoints with
    the reference's keys.  `batches` / `val_batches` are callables returning a batch (ray_batches.HumanRayBatcher).
    `pose_grad_mask(cap_id)` -> [72] multipliers or None; with opt.block_grad and `captures` carrying `.densepose` label maps it is
    densepose_pose_mask of the batch's frame (:560-573)."""

    def __init__(self, opt, net, optimizer, faces, can_mesh, can_caps, batches=None, val_batches=None, pose_grad_mask=None, captures=None, **kw):
        super().__init__(opt, net, faces, can_mesh, can_caps, **kw)
        kw_captures = captures
        self.optim, self.batches, self.val_batches, self.pose_grad_mask = optimizer, batches, val_batches, pose_grad_mask
        captures = kw_captures
        if pose_grad_mask is None and getattr(opt, 'block_grad', False) and captures is not None:      # :560-573
            def from_densepose(cap_id):
                dp = getattr(captures[cap_id], 'densepose', None)
                return None if dp is None else densepose_pose_mask(dp)
            self.pose_grad_mask = from_densepose
        self.epoch, self.iteration = 0, 0
        self._geometry_only = {}
        self.out = getattr(opt, 'out', None)
        if self.out:
            os.makedirs(self.out, exist_ok=True)
        if getattr(opt, 'resume', False):
            self.resume()
        if getattr(opt, 'load_weights', False):
            self.load_pretrained_weights()

    @staticmethod
    def _grouped(loss_dict, photometric=True):
        g = dict(loss_dict)
        g['rgb_loss'] = g['fine_rgb_loss'] + g['color_range_reg'] + g['lpips_loss']
        g['can_loss'] = g['smpl_sym_reg'] + g['smpl_shape_reg']
        g['total_loss'] = g['can_loss'] + g['mask_loss'] + g['sparsity_reg'] + (g['rgb_loss'] if photometric else 0.0)
        return g

    def train_batch(self, batch):
        opt, it = self.opt, self.iteration
        self.optim.zero_grad()
        # the loss is built and differentiated without a question to the host; its values, the NaN test of :476-478 and the dead-network flag
        # of :437-442 are ONE read-back after the backward pass (a NaN loss: the gradients are dropped, as if backward() had not run)
        self.defer_dead_check = True
        try:
            self.loss_func(batch)
        finally:
            self.defer_dead_check = False
        # the grouping of :540-551 is sums of the seven terms: the total that is differentiated is one (masked) sum on the device, the groups the
        # report shows are added up on the host from the terms' values
        terms, photometric = self.last['terms'], it >= opt.delay_iters
        if photometric:
            total = terms.sum()
        else:
            # before delay_iters the reference leaves the rgb group out of the sum (:540-551): the geometry terms are SELECTED (an index, not a
            # multiplication by zero: a NaN photometric term must not reach the total, and the rgb graph stays out of the backward pass)
            dev = terms.device
            if dev not in self._geometry_only:
                self._geometry_only[dev] = torch.tensor([i for i, n in enumerate(LOSS_NAMES) if n not in ('fine_rgb_loss', 'color_range_reg', 'lpips_loss')],
                                                        device=dev, dtype=torch.long)
            total = terms.index_select(0, self._geometry_only[dev]).sum()
        total.backward()
        vals = torch.cat([terms.detach().float(), torch.stack([total.detach().float(), self.last['alive'].float()])]).tolist()
        f32 = lambda v: float(np.float32(v))                                              # noqa: E731  (the device's additions are float32)
        report = dict(zip(LOSS_NAMES, vals[:len(LOSS_NAMES)]))
        report['rgb_loss'] = f32(f32(report['fine_rgb_loss'] + report['color_range_reg']) + report['lpips_loss'])
        report['can_loss'] = f32(report['smpl_sym_reg'] + report['smpl_shape_reg'])
        report['total_loss'] = vals[-2]
        report['lr'] = self.optim.param_groups[0]['lr']
        if vals[-1] == 0.0:                                                              # (as train_step: no gradient reaches the fresh weights)
            self._reset_dead_networks()
            self.optim.zero_grad(set_to_none=True)
        if math.isnan(report['total_loss']):
            print('loss is nan during training', report)
            self.optim.zero_grad()
        elif vals[-1] != 0.0:
            mask = self.pose_grad_mask(int(batch['cap_id'])) if self.pose_grad_mask is not None else None
            if mask is not None and getattr(self.net, 'poses', None) is not None and self.net.poses.grad is not None:
                cap = int(batch['cap_id'])
                self.net.poses.grad[cap] *= torch.as_tensor(mask, dtype=torch.float32, device=self.net.poses.device).reshape(self.net.poses.grad[cap].shape)
        self.optim.step()
        if opt.lrate_decay is not None:
            k = 0.1 ** (it / (opt.lrate_decay * 1000))
            for group in self.optim.param_groups[:1]:
                group['lr'] = opt.smpl_lr * k
            for group in self.optim.param_groups[1:3]:
                group['lr'] = opt.learning_rate * k
            keep = max(0, 1 - it / 60000)                                                # the priors fade out over 60k iterations
            self.penalize_mask = opt.penalize_mask * keep
            if opt.prior_knowledge_decay:
                self.penalize_symmetric_alpha = opt.penalize_symmetric_alpha * keep
                self.penalize_dummy = opt.penalize_dummy * keep
                self.penalize_smpl_alpha = opt.penalize_smpl_alpha * keep
            assert opt.offset_lim >= opt.offset_scale >= 0
            grown = (opt.offset_lim - opt.offset_scale) * max(0, (it - opt.offset_delay) / 60000) + opt.offset_scale
            for off in self.net.offset_nets:                                             # the offsets switch on at offset_delay and grow to offset_lim
                off.nerf.scale = min(grown, opt.offset_lim) if it >= opt.offset_delay else 0
        return report

    def validate_batch(self, batch):
        self.optim.zero_grad()
        with torch.no_grad():
            g = self._grouped(self.loss_func(batch))
        return {k: float(v) for k, v in g.items()}

    def _modules(self):
        return [m for m in [self.net.coarse_human_net, *list(self.net.offset_nets)] if hasattr(m, 'train')]

    def validate(self, n_batches=10, save=True):
        was_training = self.net.coarse_human_net.training
        for m in self._modules():
            m.eval()
        source = self.val_batches or self.batches
        reports = [self.validate_batch(source()) for _ in range(n_batches)]
        if save and self.out:
            self.save_model()
        if was_training:
            for m in self._modules():
                m.train()
        return {k: sum(r[k] for r in reports) / len(reports) for k in reports[0]}

    def train(self, max_iter=None, on_step=None):
        max_iter = self.opt.max_iter if max_iter is None else max_iter
        for m in self._modules():
            m.train()
        while True:
            if getattr(self.opt, 'valid_iter', 0) and self.iteration % self.opt.valid_iter == 0:
                self.validate()
            report = self.train_batch(self.batches())
            if on_step is not None:
                on_step(self.iteration, report)
            if self.iteration >= max_iter:
                break
            self.iteration += 1
        return report

    def save_model(self, path=None):
        torch.save({'epoch': self.epoch, 'iteration': self.iteration, 'optim_state_dict': self.optim.state_dict(),
                    'hybrid_model_state_dict': self.net.state_dict()}, path or os.path.join(self.out, 'checkpoint.pth.tar'))

    def resume(self):
        path = os.path.join(self.opt.out, 'checkpoint.pth.tar')
        if not os.path.isfile(path):
            raise FileNotFoundError(f'model check point cannot found: {path}')
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        self.epoch, self.iteration = ckpt['epoch'], ckpt['iteration']
        self.load_pretrained_weights()
        self.optim.load_state_dict(ckpt['optim_state_dict'])

    def load_pretrained_weights(self):
        from .data_io import safe_load_weights
        path = self.opt.load_weights_path
        assert os.path.isfile(path), path
        safe_load_weights(self.net, torch.load(path, map_location='cpu', weights_only=False)['hybrid_model_state_dict'])
