"""Five iterations of the REFERENCE's own background trainer (build container only):

    python tests/golden/make_golden_callers_train.py   ->  tests/golden/callers_train.npz

What runs is trainers/vanilla_nerf_trainer.py NeRFTrainer.train_batch (:206-248) -- and through it loss_func (:45-96) -- UNMODIFIED, on a stand-in `self`
holding the attributes the two methods read (the reference's own nets from models.vanilla.build_nerf with the synthetic weights loaded, torch's Adam built as
train.py:57-61 builds it), on one fixed batch (tests/helpers/caller_bodies.py trainer_batch / trainer_opt: the definitions both sides share).  The GPU test
runs the same iterations through neuman_hip.install()'s names (caller_bodies.background_trainer_iterations) and compares the loss terms of every iteration and the
parameters afterwards with what is stored here."""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))

from models import vanilla as R_vanilla  # noqa: E402  (reference)
from trainers import vanilla_nerf_trainer as R_trainer  # noqa: E402
from neuman_hip import synthetic  # noqa: E402  (ours: workload definitions only)
import caller_bodies as CB  # noqa: E402


def ref_net(seed):
    net, _ = R_vanilla.build_nerf(synthetic.default_opt())
    net.load_state_dict(synthetic.make_joiner(seed).state_dict(), strict=True)
    return net.train()


def main():
    torch.manual_seed(0)
    coarse, fine = ref_net(0), ref_net(1)
    opt = CB.trainer_opt()
    optim = torch.optim.Adam([{"params": coarse.parameters(), "lr": opt.learning_rate}, {"params": fine.parameters(), "lr": opt.learning_rate}], betas=(0.9, 0.999))
    pushed = []
    fake = types.SimpleNamespace(opt=opt, coarse_net=coarse, fine_net=fine, optim=optim, iteration=0, penalize_empty_space=opt.penalize_empty_space,
                                 empty_space_loss_fn=F.mse_loss, push_training_data=lambda batch, losses, lr: pushed.append((dict(losses), lr)))
    fake.loss_func = types.MethodType(R_trainer.NeRFTrainer.loss_func, fake)
    b = CB.trainer_batch()
    for it in range(CB.ITERS_TR):
        fake.iteration = it
        batch = {k: torch.from_numpy(v)[None] for k, v in b.items()}                  # the DataLoader's leading axis (utils.remove_first_axis takes it off)
        R_trainer.NeRFTrainer.train_batch(fake, batch)
    names = ('coarse_rgb_loss', 'coarse_empty_space_loss', 'fine_rgb_loss', 'fine_empty_space_loss')
    out = {'terms': np.array([[l[n] for n in names] for l, _ in pushed]), 'lr_seen': np.array([lr for _, lr in pushed]),
           'total': np.array([l['total_loss'] for l, _ in pushed])}
    for tag, net in (('coarse', coarse), ('fine', fine)):
        sd = net.state_dict()
        for k in ('nerf.pts_linears.0.weight', 'nerf.pts_linears.5.weight', 'nerf.pts_linears.7.bias', 'nerf.views_linears.0.weight', 'nerf.rgb_linear.weight', 'nerf.alpha_linear.bias'):
            out[f'{tag}/{k}'] = sd[k].numpy().copy()
        out[f'{tag}/abs_sum'] = np.array([float(sum(v.abs().sum(dtype=torch.float64) for v in sd.values()))])
    np.savez_compressed(os.path.join(HERE, "callers_train.npz"), **out)
    print("wrote callers_train.npz; terms per iteration:\n", out['terms'], "\ntotals", out['total'], "lr", out['lr_seen'])


if __name__ == "__main__":
    main()
