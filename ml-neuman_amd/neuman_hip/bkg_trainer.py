"""The background NeRF's training loop on the HIP training slice: what trainers/vanilla_nerf_trainer.py
NeRFTrainer + trainers/base_trainer.py do around the hot path, with the batches of neuman_hip/ray_batches.py already on the device.

    loss_func      :45-96    coarse + fine photometric MSE, empty-space penalty in front of the MVS depth, dead-network restart
    train_batch    :205-248  delay of the penalty, NaN guard, Adam step, learning-rate and penalty schedules
    validate_batch :98-117, save_model :181-191, resume / load_pretrained_weights :281-319 (same checkpoint keys, so files move
    between this trainer and the reference's in both directions)

Forward and backward of the two networks are the f32-MFMA GEMMs of csrc/train.hip through `Joiner.forward` in training mode; the
sampling, importance resampling and compositing (with its hand-written adjoint) are the kernels the renderers use.  No
tensorboard, no DataLoader: `train()` pulls batches from a callable.
"""
import math
import os

import torch
import torch.nn.functional as F

from . import ray_utils, render_utils
from .data_io import safe_load_weights
from .vanilla import weight_reset

LOSS_TERMS = ('coarse_rgb_loss', 'coarse_empty_space_loss', 'fine_rgb_loss', 'fine_empty_space_loss')


def decayed_rate(base, iteration, decay_k):
    """learning rate after `iteration` steps: a factor 0.1 every decay_k thousand iterations (vanilla_nerf_trainer.py:237-240)"""
    return base * 0.1 ** (iteration / (decay_k * 1000))


def fade(iteration, span=60000):
    """linear fade of a prior-knowledge penalty to zero over `span` iterations (:244)"""
    return max(0, 1 - iteration / span)


class BackgroundNeRFTrainer:
    """`data_parallel`: train over the initialised process group as the reference's nn.DataParallel nets do over the visible GPUs
    (train.py:26-28; neuman_hip/dp.py): this rank's rays of every batch, ONE all_reduce of the flat gradient buffer per iteration.
    None = on when a process group of more than one rank exists.  `batches` must then yield this rank's slice
    (BackgroundRayBatcher(rank=, world=)) unless `shard_batches` (full batches, identical on every rank, sliced here)."""

    def __init__(self, opt, coarse_net, optimizer, fine_net=None, batches=None, val_batches=None, penalize_empty_space=None,
                 data_parallel=None, shard_batches=False, group=None):
        self.opt = opt
        self.coarse_net, self.fine_net, self.optim = coarse_net, fine_net, optimizer
        self.batches, self.val_batches = batches, val_batches
        self.penalize_empty_space = opt.penalize_empty_space if penalize_empty_space is None else penalize_empty_space
        self.empty_space_loss_fn = {'l1': F.l1_loss, 'mse': F.mse_loss}[getattr(opt, 'empty_space_loss_fn', 'mse')]
        self.epoch, self.iteration = 0, 0
        self.out = getattr(opt, 'out', None)
        if self.out:
            os.makedirs(self.out, exist_ok=True)
        if getattr(opt, 'resume', False):
            self.resume()
        if getattr(opt, 'load_weights', False):
            self.load_pretrained_weights()
        from . import dp
        self.rank, self.world = dp.rank_world(group)
        if data_parallel is None:
            data_parallel = self.world > 1
        self.sync, self.shard_batches, self.group = None, shard_batches, group
        if data_parallel:
            dp.broadcast_parameters(self._nets(), group=group)    # (after a resume too: every rank starts from rank 0's weights ...
            if getattr(opt, 'resume', False):
                dp.broadcast_optimizer_state(self.optim, group=group)      # ... and rank 0's Adam moments and step counts)
            dp.decorrelate_device_rng(self.rank)                  # same batches on every rank, independent per-sample jitter / noise on the shards
            self.sync = dp.GradSync([p for n in self._nets() for p in n.parameters()], n_extra=5, group=group)

    # ---------------------------------------------------------------------------------------------
    def _empty_space(self, raw, z_vals, depth):
        """density in front of `margin` x the MVS depth is pushed to zero through tanh(relu(sigma)) (:66-72)"""
        if not self.penalize_empty_space > 0:
            return torch.zeros((), device=raw.device)
        sigma = raw[..., 3][z_vals < depth[:, None] * self.opt.margin]
        return self.empty_space_loss_fn(torch.tanh(torch.relu(sigma)), torch.zeros_like(sigma)) * self.penalize_empty_space

    def _pass(self, net, batch, z_vals, time):
        o, d = batch['origin'], batch['direction']
        pts, dirs = ray_utils.z_to_points(o, d, z_vals)
        if time is not None:
            pts = torch.cat([pts, time.expand(*z_vals.shape)[..., None]], dim=-1)
        raw = net(pts, dirs)
        rgb, _, _, weights, _ = render_utils.raw2outputs(raw, z_vals, d, raw_noise_std=self.opt.raw_noise_std, white_bkg=self.opt.white_bkg)
        return raw, rgb, weights

    def loss_func(self, batch, device=None):
        """-> (coarse_rgb, coarse_empty_space, fine_rgb, fine_empty_space); batch as ray_batches.BackgroundRayBatcher makes it"""
        opt = self.opt
        o, d = batch['origin'], batch['direction']
        time = batch['viewf_list'] if getattr(opt, 'ablate_nerft', False) else None
        _, _, z = ray_utils.sample_z(o, d, batch['near'].reshape(-1), batch['far'].reshape(-1), opt.samples_per_ray, perturb=opt.perturb)
        raw, rgb, weights = self._pass(self.coarse_net, batch, z, time)
        terms = [F.mse_loss(rgb, batch['color']), self._empty_space(raw, z, batch.get('depth'))]
        dead = raw[..., 3].max() <= 0.0                           # (:88: `max() <= 0.0` -- a NaN density is the NaN guard's business, not a restart)
        if self.fine_net is not None:
            with torch.no_grad():
                z_fine = ray_utils.importance_z(z, weights.detach(), opt.importance_samples_per_ray)
            raw_f, rgb_f, _ = self._pass(self.fine_net, batch, z_fine, time)
            terms += [F.mse_loss(rgb_f, batch['color']), self._empty_space(raw_f, z_fine, batch.get('depth'))]
            dead = dead | (raw_f[..., 3].max() <= 0.0)
        else:
            terms += [torch.zeros_like(terms[0]), torch.zeros_like(terms[0])]
        if bool(dead):                                            # no sample with positive density anywhere: redraw the weights (:88-94)
            print('bad weights, reinitializing')
            for net in (self.coarse_net, self.fine_net):
                if net is not None:
                    net.apply(weight_reset)
            terms = [torch.zeros((), device=o.device, requires_grad=True) for _ in range(4)]
        return tuple(terms)

    def loss_func_dp(self, batch):
        """loss_func on THIS RANK'S rays with every mean taken over the whole batch: the four terms are this rank's SHARE of the full-batch
        terms (their sum over the ranks is what loss_func returns on the concatenated batch, and so are the gradients).  The global
        element counts and the dead-network test need one small all_gather before the backward pass.  -> (terms, dead)"""
        from . import dp
        opt = self.opt
        o, d = batch['origin'], batch['direction']
        time = batch['viewf_list'] if getattr(opt, 'ablate_nerft', False) else None
        _, _, z = ray_utils.sample_z(o, d, batch['near'].reshape(-1), batch['far'].reshape(-1), opt.samples_per_ray, perturb=opt.perturb)
        depth = batch.get('depth')

        def pieces(raw, rgb, zz):
            sq = ((rgb - batch['color']) ** 2).sum()
            if self.penalize_empty_space > 0:                                # (:66-72 as masked sums: no index list, no host question)
                closer = zz < depth[:, None] * opt.margin
                x = torch.tanh(torch.relu(raw[..., 3]))
                fx = x * x if self.empty_space_loss_fn is F.mse_loss else x.abs()
                es = (fx * closer.to(fx.dtype)).sum() * self.penalize_empty_space
                cnt = closer.sum()
            else:
                es, cnt = torch.zeros((), device=raw.device), torch.zeros((), device=raw.device)
            return sq, es, cnt, raw[..., 3].detach().max()
        raw, rgb, weights = self._pass(self.coarse_net, batch, z, time)
        sq_c, es_c, cnt_c, max_c = pieces(raw, rgb, z)
        sq_f = es_f = None
        cnt_f, max_f = torch.zeros((), device=o.device), torch.ones((), device=o.device)
        if self.fine_net is not None:
            with torch.no_grad():
                z_fine = ray_utils.importance_z(z, weights.detach(), opt.importance_samples_per_ray)
            raw_f, rgb_f, _ = self._pass(self.fine_net, batch, z_fine, time)
            sq_f, es_f, cnt_f, max_f = pieces(raw_f, rgb_f, z_fine)
        # the global counts and the dead-network test stay ON THE DEVICE (no host question between the two passes): NaN-aware largest density --
        # `max <= 0` must be False for NaN on any rank (the reference's test on the whole batch)
        stats = dp.all_gather_floats([float(rgb.numel()), cnt_c, cnt_f, torch.nan_to_num(max_c, nan=1.0), torch.nan_to_num(max_f, nan=1.0)], self.group, on_device=True)
        n_rgb, n_c, n_f = stats[:, 0].sum().float(), stats[:, 1].sum().float().clamp_min(1.0), stats[:, 2].sum().float().clamp_min(1.0)
        dead = (stats[:, 3].max() <= 0.0) | (stats[:, 4].max() <= 0.0)
        zero = torch.zeros((), device=o.device)
        terms = [sq_c / n_rgb, es_c / n_c]                                 # (an empty selection: es = 0 over a count clamped to 1)
        terms += [sq_f / n_rgb, es_f / n_f] if sq_f is not None else [zero, zero]
        return [torch.where(dead, torch.zeros_like(t), t) for t in terms], dead

    def _train_batch_dp(self, batch):
        """train_batch over the process group: local share of the loss, backward, ONE all_reduce of gradients + loss values, the NaN guard and
        the dead-network restart decided on the GLOBAL values (identically on every rank), Adam"""
        from . import dp
        if self.shard_batches:
            batch = dp.shard_batch(batch, self.rank, self.world)
        self.sync.zero()
        terms, dead = self.loss_func_dp(batch)
        rgb_loss, empty_loss = terms[0] + terms[2], terms[1] + terms[3]
        total = rgb_loss + empty_loss if self.iteration >= self.opt.delay_iters else rgb_loss
        if total.requires_grad:
            total.backward()
        # ONE collective for the gradients, the four loss values and the dead flag (identical on every rank already), ONE read-back
        vals = self.sync.reduce(extra=[t.detach() for t in terms] + [dead.to(torch.float32) / self.world])
        dead = vals[4] > 0.5
        vals = vals[:4]
        if dead:                                                  # (:88-94) redraw on rank 0, everybody takes those weights; no step on them
            print('bad weights, reinitializing')
            for net in self._nets():
                net.apply(weight_reset)
            dp.broadcast_parameters(self._nets(), group=self.group)
            vals = [0.0, 0.0, 0.0, 0.0]
        report = dict(zip(LOSS_TERMS, vals))
        report.update(rgb_loss=vals[0] + vals[2], empty_space_loss=vals[1] + vals[3], lr=self.optim.param_groups[0]['lr'])
        report['total_loss'] = report['rgb_loss'] + (report['empty_space_loss'] if self.iteration >= self.opt.delay_iters else 0.0)
        if dead or math.isnan(report['total_loss']):
            if not dead:
                print('loss is nan during training')
            self.sync.zero()
            for p in self.sync.params:
                p.grad = None
        self.optim.step()
        return report

    # ---------------------------------------------------------------------------------------------
    def train_batch(self, batch):
        if self.sync is not None:
            report = self._train_batch_dp(batch)
            self._schedules()
            return report
        self.optim.zero_grad()
        terms = self.loss_func(batch)
        rgb_loss, empty_loss = terms[0] + terms[2], terms[1] + terms[3]
        total = rgb_loss + empty_loss if self.iteration >= self.opt.delay_iters else rgb_loss
        report = dict(zip(LOSS_TERMS, (float(t.detach()) for t in terms)))
        report.update(rgb_loss=float(rgb_loss.detach()), empty_space_loss=float(empty_loss.detach()), total_loss=float(total.detach()),
                      lr=self.optim.param_groups[0]['lr'])
        if math.isnan(report['total_loss']):
            print('loss is nan during training')
            self.optim.zero_grad()
        else:
            total.backward()
        self.optim.step()
        self._schedules()
        return report

    def _schedules(self):
        if self.opt.lrate_decay is not None:
            for group in self.optim.param_groups:
                group['lr'] = decayed_rate(self.opt.learning_rate, self.iteration, self.opt.lrate_decay)
        if self.opt.penalize_empty_space > 0:
            self.penalize_empty_space = self.opt.penalize_empty_space * fade(self.iteration)

    def validate_batch(self, batch):
        self.optim.zero_grad()
        assert not self.coarse_net.training and (self.fine_net is None or not self.fine_net.training)
        with torch.no_grad():
            terms = self.loss_func(batch)
        report = dict(zip(LOSS_TERMS, (float(t) for t in terms)))
        report['rgb_loss'] = report['coarse_rgb_loss'] + report['fine_rgb_loss']
        report['empty_space_loss'] = report['coarse_empty_space_loss'] + report['fine_empty_space_loss']
        report['total_loss'] = report['rgb_loss'] + report['empty_space_loss']
        return report

    def _nets(self):
        return [n for n in (self.coarse_net, self.fine_net) if n is not None]

    def validate(self, n_batches=10, save=True):
        """mean losses over `n_batches` validation batches (VALIDATION_SET_LENGTH = 10), then the checkpoint (:119-179 minus the
        tensorboard renders)"""
        was_training = self.coarse_net.training
        for n in self._nets():
            n.eval()
        source = self.val_batches or self.batches
        reports = [self.validate_batch(source()) for _ in range(n_batches)]
        if save and self.out:
            self.save_model()
        if was_training:
            for n in self._nets():
                n.train()
        return {k: sum(r[k] for r in reports) / len(reports) for k in reports[0]}

    def train(self, max_iter=None, on_step=None):
        """BaseTrainer.train / train_epoch (base_trainer.py:66-108): validate every valid_iter iterations, train until max_iter"""
        max_iter = self.opt.max_iter if max_iter is None else max_iter
        for n in self._nets():
            n.train()
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

    # ---------------------------------------------------------------------------------------------
    def save_model(self, path=None):
        """checkpoint with the reference's keys.  Under data parallelism rank 0 writes, and the networks' keys carry the 'module.' prefix of
        the nn.DataParallel wrappers the reference's train.py saves through (train.py:26-28; both forms load either way, utils/utils.py:225-254)"""
        if self.sync is not None and self.rank != 0:
            return
        pre = 'module.' if (self.sync is not None and getattr(self.opt, 'module_prefix', True)) else ''
        state = {'epoch': self.epoch, 'iteration': self.iteration, 'optim_state_dict': self.optim.state_dict(),
                 'coarse_model_state_dict': {pre + k: v for k, v in self.coarse_net.state_dict().items()}}
        if self.fine_net is not None:
            state['fine_model_state_dict'] = {pre + k: v for k, v in self.fine_net.state_dict().items()}
        torch.save(state, path or os.path.join(self.out, 'checkpoint.pth.tar'))

    def resume(self):
        path = os.path.join(self.opt.out, 'checkpoint.pth.tar')
        if not os.path.isfile(path):
            raise FileNotFoundError(f'model check point cannnot found: {path}')
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        self.epoch, self.iteration = ckpt['epoch'], ckpt['iteration']
        self.load_pretrained_weights()
        self.optim.load_state_dict(ckpt['optim_state_dict'])

    def load_pretrained_weights(self):
        path = self.opt.load_weights_path
        assert os.path.isfile(path), path
        saved = torch.load(path, map_location='cpu', weights_only=False)
        safe_load_weights(self.coarse_net, saved['coarse_model_state_dict'])
        if 'fine_model_state_dict' in saved and self.fine_net is not None:
            safe_load_weights(self.fine_net, saved['fine_model_state_dict'])
