"""Golden vectors for SURVEY 8f-1 at a batch large enough for the 16-bit storage path of the training step (neuman_hip/train.py STORE16:
fp16 copies of the activations / dZ from 32768 samples on), from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_train_big.py

The reference's own NeRFTrainer.loss_func (trainers/vanilla_nerf_trainer.py:45-96), unmodified, + torch autograd's backward, as
make_golden_train.py -- 2048 rays x 64 coarse / 64 + 64 fine samples (131072 / 262144 evaluations: the order of the trainers' batches).  Kept: the batch, both z arrays, the
losses, rgb maps, and per parameter tensor the summary of make_golden_train.grad_summary (three full rows, sum, sum of magnitudes, a fixed
random projection)."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_train as M  # noqa: E402  (sets up the reference imports)

R, S, NI = 2048, 64, 64


def main():
    rng = np.random.default_rng(78)
    out = {}
    ro = (rng.normal(size=(R, 3)) * 0.3).astype(np.float32)
    rd = rng.normal(size=(R, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    near = rng.uniform(0.1, 0.6, size=(R, 1)).astype(np.float32)
    far = (near + rng.uniform(1.0, 2.5, size=(R, 1))).astype(np.float32)
    color = rng.uniform(size=(R, 3)).astype(np.float32)
    depth = rng.uniform(0.8, 2.0, size=(R,)).astype(np.float32)
    out.update(origin=ro, direction=rd, near=near, far=far, color=color, depth=depth)
    tag, white, penal = 'black_penalty', False, 0.1
    coarse, fine = M.ref_net(0), M.ref_net(1)
    opt = types.SimpleNamespace(ablate_nerft=False, samples_per_ray=S, importance_samples_per_ray=NI, perturb=0.0, raw_noise_std=0.0, white_bkg=white, margin=0.9)
    fake = types.SimpleNamespace(opt=opt, coarse_net=coarse, fine_net=fine, penalize_empty_space=penal, empty_space_loss_fn=F.mse_loss)
    captured = {}
    real_r2o = M.R_render.raw2outputs

    def spy(raw, z, d, **kw):
        raw.retain_grad()
        res = real_r2o(raw, z, d, **kw)
        captured.setdefault('calls', []).append((raw, z, d, res))
        return res
    M.R_trainer.render_utils.raw2outputs = spy
    batch = {k: torch.from_numpy(v)[None] for k, v in dict(origin=ro, direction=rd, near=near, far=far, color=color, depth=depth).items()}
    losses = M.R_trainer.NeRFTrainer.loss_func(fake, batch, 'cpu')
    M.R_trainer.render_utils.raw2outputs = real_r2o
    sum(losses).backward()
    out[f'{tag}/losses'] = np.array([float(x.detach()) for x in losses])
    for name, net, (raw, z, d, res) in zip(('coarse', 'fine'), (coarse, fine), captured['calls']):
        p = f'{tag}/{name}'
        out[f'{p}/z'] = z.detach().numpy()
        out[f'{p}/rgb_map'] = res[0].detach().numpy()
        g = raw.grad.numpy()
        out[f'{p}/d_raw_stats'] = np.array([np.abs(g).max(), np.abs(g).sum()])
        for n, prm in net.named_parameters():
            M.grad_summary(n, prm.grad, out, p)
    np.savez_compressed(os.path.join(HERE, 'train_big.npz'), **out)
    print({k: out[k] for k in out if k.endswith('losses')})
    print(len(out), "arrays,", os.path.getsize(os.path.join(HERE, 'train_big.npz')), "bytes")


if __name__ == '__main__':
    main()
