"""neuman_hip/bkg_trainer.py: the background trainer's loss against the reference's own NeRFTrainer.loss_func (tests/golden/train.npz,
made by make_golden_train.py), and the loop around it -- device batches in, Adam, schedules, checkpoints with the reference's keys."""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "helpers"))
import batch_scene  # noqa: E402

pytestmark = pytest.mark.gpu
S, NI = 24, 16                                                   # make_golden_train.py


def trainer_opt(**over):
    o = dict(samples_per_ray=S, importance_samples_per_ray=NI, perturb=0.0, raw_noise_std=0.0, white_bkg=True, margin=0.9,
             penalize_empty_space=0.0, empty_space_loss_fn='mse', delay_iters=0, lrate_decay=250, learning_rate=5e-4, ablate_nerft=False,
             rays_per_batch=512, max_iter=30, valid_iter=0, out=None, resume=False, load_weights=False)
    o.update(over)
    return types.SimpleNamespace(**o)


@pytest.fixture(scope="module")
def G():
    from neuman_hip import bkg_trainer, data_io, ray_batches, synthetic, train
    g = np.load(os.path.join(HERE, "golden", "train.npz"))
    return types.SimpleNamespace(bt=bkg_trainer, io=data_io, rb=ray_batches, syn=synthetic, train=train, g=g)


@pytest.mark.parametrize("tag,white,penalty", [("white", True, 0.0), ("black_penalty", False, 0.1)])
def test_loss_func_matches_the_reference_trainer(G, tag, white, penalty, monkeypatch):
    monkeypatch.setattr(G.train, "GEMM_PRECISION", "f32")
    g = G.g
    batch = {k: torch.from_numpy(g[k]).cuda() for k in ('origin', 'direction', 'near', 'far', 'color', 'depth')}
    coarse, fine = G.syn.make_joiner(0).cuda().train(), G.syn.make_joiner(1).cuda().train()
    opt = trainer_opt(white_bkg=white, penalize_empty_space=penalty)
    tr = G.bt.BackgroundNeRFTrainer(opt, coarse, torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4), fine_net=fine)
    terms = tr.loss_func(batch)
    got = np.array([float(t.detach()) for t in terms])
    want = g[f'{tag}/losses']
    print(f"[bkg trainer] {tag}: loss terms {got} vs the reference's {want}")
    np.testing.assert_allclose(got, want, rtol=3e-5, atol=1e-7)
    sum(terms).backward()
    # parameter gradients of the summed loss against the reference's autograd (the per-layer summaries of the golden file)
    from test_oracle_train import check_grads
    for name, net in (("coarse", coarse), ("fine", fine)):
        worst = check_grads({n: p.grad.cpu().numpy() for n, p in net.named_parameters()}, g, f'{tag}/{name}')
        print(f"[bkg trainer] {tag}/{name}: worst relative parameter-gradient error vs the reference {worst:.2e}")


def _scene_store(G, dilation=4):
    spec = batch_scene.make(seed=11)
    caps = []
    yy, xx = np.mgrid[0:spec['h'], 0:spec['w']]
    for i, c in enumerate(spec['captures']):
        cam = G.io.PinholeCamera(spec['w'], spec['h'], *c['intrinsics'])
        cap = G.io.Capture(os.path.join('/nowhere/images', c['name']), cam, G.io.CameraPose(c['t'].astype(np.float32), c['q'].astype(np.float32)),
                           frame_id={'frame_id': i, 'total_frames': 3})
        img = np.stack([xx * 4 + 10 * i, yy * 5, 120 + 0 * xx], -1).clip(0, 255).astype(np.uint8)      # a smooth picture a field can fit
        cap.image, cap.mask, cap.depth_map = img, c['mask'], c['depth']
        cap.near, cap.far = dict(c['near']), dict(c['far'])
        caps.append(cap)
    return G.rb.FrameStore(caps, 'cuda', dilation=dilation)


def test_training_loop_reduces_the_loss_and_checkpoints(G, tmp_path):
    store = _scene_store(G)
    opt = trainer_opt(out=str(tmp_path / 'run'), penalize_empty_space=0.1, max_iter=29, lrate_decay=1, perturb=1.0)
    coarse, fine = G.syn.make_joiner(0).cuda(), G.syn.make_joiner(1).cuda()
    optim = torch.optim.Adam([{"params": coarse.parameters(), "lr": opt.learning_rate}, {"params": fine.parameters(), "lr": opt.learning_rate}])
    batches = G.rb.BackgroundRayBatcher(opt, store, draws='device', seed=1)
    tr = G.bt.BackgroundNeRFTrainer(opt, coarse, optim, fine_net=fine, batches=batches)
    log = []
    tr.train(on_step=lambda it, rep: log.append(rep))
    assert len(log) == 30 and tr.iteration == 29
    first, last = np.mean([r['rgb_loss'] for r in log[:3]]), np.mean([r['rgb_loss'] for r in log[-3:]])
    print(f"[bkg trainer] 30 iterations of 512 rays: rgb loss {first:.4f} -> {last:.4f}; empty-space term {log[0]['empty_space_loss']:.4f} -> {log[-1]['empty_space_loss']:.4f}")
    assert last < 0.7 * first and all(math.isfinite(r['total_loss']) for r in log)
    # schedules (vanilla_nerf_trainer.py:236-244): lr = base * 0.1^(it / (decay * 1000)), penalty fades over 60k iterations
    assert optim.param_groups[0]['lr'] == pytest.approx(5e-4 * 0.1 ** (29 / 1000), rel=1e-12) and optim.param_groups[1]['lr'] == optim.param_groups[0]['lr']
    assert tr.penalize_empty_space == pytest.approx(0.1 * (1 - 29 / 60000), rel=1e-12)
    # validation uses the rendering kernels (eval mode), leaves the nets in training mode and writes the checkpoint
    rep = tr.validate(n_batches=2)
    assert coarse.training and math.isfinite(rep['total_loss']) and abs(rep['rgb_loss'] - last) < 0.5 * first
    ckpt_path = tmp_path / 'run' / 'checkpoint.pth.tar'
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    assert set(ckpt) == {'epoch', 'iteration', 'optim_state_dict', 'coarse_model_state_dict', 'fine_model_state_dict'} and ckpt['iteration'] == 29
    # resume into fresh networks: iteration, weights and Adam moments come back; the readers of data_io take the same file
    c2, f2 = G.syn.make_joiner(5).cuda(), G.syn.make_joiner(6).cuda()
    o2 = torch.optim.Adam([{"params": c2.parameters(), "lr": 1.0}, {"params": f2.parameters(), "lr": 1.0}])
    opt2 = trainer_opt(out=str(tmp_path / 'run'), resume=True, load_weights_path=str(ckpt_path))
    tr2 = G.bt.BackgroundNeRFTrainer(opt2, c2, o2, fine_net=f2, batches=batches)
    assert tr2.iteration == 29
    for a, b in zip(list(coarse.parameters()) + list(fine.parameters()), list(c2.parameters()) + list(f2.parameters())):
        assert torch.equal(a, b)
    assert o2.param_groups[0]['lr'] == optim.param_groups[0]['lr'] and len(o2.state) == len(optim.state)
    c3, f3 = G.syn.make_joiner(7).cuda(), G.syn.make_joiner(8).cuda()
    G.io.load_background_checkpoint(str(ckpt_path), c3, f3)
    assert torch.equal(next(c3.parameters()), next(coarse.parameters()))


def test_dead_network_is_reinitialised(G, capsys):
    """vanilla_nerf_trainer.py:88-94: no positive density anywhere -> both nets are redrawn and the step contributes nothing"""
    store = _scene_store(G)
    opt = trainer_opt()
    coarse, fine = G.syn.make_joiner(0).cuda().train(), G.syn.make_joiner(1).cuda().train()
    with torch.no_grad():
        coarse.nerf.alpha_linear.weight.zero_()
        coarse.nerf.alpha_linear.bias.fill_(-1.0)
    before = coarse.nerf.alpha_linear.bias.clone()
    optim = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    tr = G.bt.BackgroundNeRFTrainer(opt, coarse, optim, fine_net=fine, batches=G.rb.BackgroundRayBatcher(opt, store, seed=2))
    rep = tr.train_batch(tr.batches())
    assert 'bad weights' in capsys.readouterr().out
    assert rep['total_loss'] == 0.0 and not torch.equal(coarse.nerf.alpha_linear.bias, before)
