"""HumanNeRF: the scene's networks and per-frame body parameters in one module, as reference models/human_nerf.py:20-122 holds them --
same constructor, same attribute and state_dict names (coarse_bkg_net, fine_bkg_net, offset_nets, coarse_human_net, poses, betas,
alignments, da_smpl, body_model.*), so that `hybrid_model_state_dict` checkpoints (trainers/human_nerf_trainer.py:496-505,
render_*.py) load strictly in both directions.  The networks are this package's Joiners (HIP kernels behind them); `vertex_forward`
is the differentiable skinning of neuman_hip.smpl.SMPLDiff (pose / shape / alignment refinement), float32 like the reference's.

The SMPL asset (licensed, absent offline) is looked up like the reference does -- `data/smplx/smpl/SMPL_NEUTRAL.pkl` -- under
`smpl_dir` (argument), $NEUMAN_SMPL_DIR, or the working directory; tests pass the synthetic model of neuman_hip.synthetic.
"""
import copy
import os

import numpy as np
import torch
import torch.nn as nn

from . import data_io, vanilla
from .smpl import SMPLDiff, da_pose


class HumanNeRF(nn.Module):
    def __init__(self, opt, poses=None, betas=None, alignments=None, scale=None, smpl_dir=None):
        super().__init__()
        self.coarse_bkg_net, self.fine_bkg_net = vanilla.build_nerf(opt)                                            # :23
        self.offset_nets = nn.ModuleList([vanilla.build_offset_net(opt) for _ in range(opt.num_offset_nets)])       # :24
        temp_opt = copy.deepcopy(opt)                                  # canonical space: minimum frequency 0, its own head and encoding (:26-30)
        temp_opt.pos_min_freq = 0
        temp_opt.use_viewdirs = temp_opt.specular_can
        temp_opt.posenc = temp_opt.can_posenc
        self.coarse_human_net, _ = vanilla.build_nerf(temp_opt)
        device = torch.device('cuda') if getattr(opt, 'use_cuda', False) else torch.device('cpu')
        if poses is not None:                                          # :31-50
            assert betas is not None and alignments is not None and scale is not None
            self.poses = nn.Parameter(torch.from_numpy(np.asarray(poses)).float().to(device), requires_grad=True)
            self.betas = nn.Parameter(torch.from_numpy(np.asarray(betas)).float().to(device), requires_grad=True)
            self.alignments = nn.Parameter(torch.from_numpy(np.asarray(alignments)).float().to(device), requires_grad=True)
            self.scale = scale
            model = smpl_dir if isinstance(smpl_dir, dict) else os.path.join(
                smpl_dir or os.environ.get('NEUMAN_SMPL_DIR') or os.path.join(os.getcwd(), 'data/smplx/smpl'), 'SMPL_NEUTRAL.pkl')
            self.body_model = SMPLDiff(model, device)
            self.da_smpl = nn.Parameter(torch.from_numpy(da_pose(len(self.body_model.parents_list)))[None].to(device),
                                        requires_grad=getattr(opt, 'use_cuda', False) is False)   # (the reference's CUDA branch freezes it, :90)
            self.poses_orig, self.betas_orig = np.array(poses, copy=True), np.array(betas, copy=True)
        for what, key in (('load_background', None), ('load_can', 'coarse_human_net.')):               # :52-74: optional pre-trained parts
            name = getattr(opt, what, None)
            if not name or not hasattr(opt, 'out_dir'):
                continue
            path = os.path.join(opt.out_dir, name, 'checkpoint.pth.tar')
            try:
                ckpt = torch.load(path, map_location='cpu', weights_only=False)
                if key is None:
                    data_io.safe_load_weights(self.coarse_bkg_net, ckpt['coarse_model_state_dict'])
                    data_io.safe_load_weights(self.fine_bkg_net, ckpt['fine_model_state_dict'])
                    print(f'pretrained background model loaded from {path}')
                else:
                    sd = {k.split(key, 1)[1]: v for k, v in ckpt['hybrid_model_state_dict'].items() if key in k}
                    data_io.safe_load_weights(self.coarse_human_net, sd)
                    print(f'pretrained canonical human model loaded from {path}')
            except Exception as e:                                    # as the reference: report and train from scratch
                print(e)
                print('train from scratch')
        if getattr(opt, 'use_cuda', False):                           # :76-90
            self.coarse_bkg_net, self.fine_bkg_net = self.coarse_bkg_net.cuda(), self.fine_bkg_net.cuda()
            self.offset_nets, self.coarse_human_net = self.offset_nets.cuda(), self.coarse_human_net.cuda()

    def vertex_forward(self, idx, pose=None, beta=None):
        """reference :92-122 -> (world_verts [1,V,3], T_da2scene [1,V,4,4]), differentiable in poses / betas / alignments"""
        if pose is None:
            pose = self.poses[idx][None]
        if beta is None:
            beta = self.betas[idx][None]
        return self.body_model.vertex_forward(pose, beta, self.alignments[idx], self.scale, da_pose=self.da_smpl)
