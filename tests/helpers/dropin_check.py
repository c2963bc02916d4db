"""Executed by tests/test_install_dropin.py in its own interpreter (the stubbed wheels must not leak into the test session):
import the REFERENCE unmodified from /root/reference, neuman_hip.install() over it, then check the drop-in boundary:

  * every rebound hot-path function keeps the reference's signature (equal, or the reference's plus trailing keyword arguments
    with defaults);
  * the reference's own HumanNeRF(opt) (models/human_nerf.py:21-31) builds on the rebound models.vanilla;
  * load_state_dict(strict=True) works in both directions between the reference's modules and ours (checkpoint compatibility).

Prints one JSON object.
"""
import argparse
import importlib
import inspect
import json
import os
import sys
from unittest import mock

REF = "/root/reference"
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))

import torch  # noqa: E402
from utils import ray_utils, render_utils  # noqa: E402  (reference)
from models import vanilla  # noqa: E402
from options import options  # noqa: E402

# the reference's own objects, before anything is rebound
REF_FNS = {}
NAMES = {ray_utils: ["shot_ray", "shot_rays", "shot_all_rays", "to_homogeneous", "ray_to_samples", "ray_to_importance_samples", "sample_pdf",
                     "geometry_guided_near_far", "geometry_guided_near_far_torch", "geometry_guided_near_far_np",
                     "warp_samples_to_canonical", "warp_samples_to_canonical_diff"],
         render_utils: ["raw2outputs", "render_vanilla", "render_smpl_nerf", "render_hybrid_nerf", "render_hybrid_nerf_multi_persons"]}
for mod, names in NAMES.items():
    for n in names:
        REF_FNS[(mod.__name__, n)] = getattr(mod, n)


def parse_opt():
    parser = argparse.ArgumentParser()
    for f in (options.set_general_option, options.set_nerf_option, options.set_pe_option, options.set_render_option):
        f(parser)
    parser.add_argument('--offset_scale', type=float, default=1.0)
    parser.add_argument('--num_offset_nets', type=int, default=1)
    parser.add_argument('--offset_scale_type', type=str, default='linear')
    parser.add_argument('--out_dir', type=str, default='./out')
    parser.add_argument('--load_background', type=str, default='none')
    parser.add_argument('--load_can', type=str, default='none')
    parser.add_argument('--posenc', type=str, default='posenc')
    return parser.parse_args(['--use_cuda', 'no'])


opt = parse_opt()
torch.manual_seed(0)
r_coarse, r_fine = vanilla.build_nerf(opt)                # the reference's own classes (models.vanilla is rebound below)

import neuman_hip  # noqa: E402

neuman_hip.install(ray_utils, render_utils, vanilla)

out = {"signatures": {}, "rebound": 0}
for (modname, n), ref_fn in REF_FNS.items():
    mod = importlib.import_module(modname)
    ours = getattr(mod, n)
    out["rebound"] += int(ours is not ref_fn)
    rs, os_ = inspect.signature(ref_fn), inspect.signature(ours)
    rp, op = list(rs.parameters.values()), list(os_.parameters.values())
    same_prefix = len(op) >= len(rp) and all(a.name == b.name and a.default == b.default and a.kind == b.kind for a, b in zip(rp, op))
    extras_ok = all(p.default is not inspect.Parameter.empty for p in op[len(rp):])
    out["signatures"][f"{modname}.{n}"] = {"ok": bool(same_prefix and extras_ok), "reference": str(rs), "ours": str(os_)}

# the reference's HumanNeRF on the rebound models.vanilla
from models import human_nerf  # noqa: E402  (imports models.vanilla -> ours now)
import contextlib  # noqa: E402
import io  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    net = human_nerf.HumanNeRF(opt)
sd = net.state_dict()
out["human_nerf"] = {"state_dict_tensors": len(sd), "parameters": int(sum(p.numel() for p in net.parameters())),
                     "bkg_is_ours": type(net.coarse_bkg_net).__module__.startswith("neuman_hip"),
                     "human_mapping": net.coarse_human_net.pos_pe.mapping}

# checkpoint compatibility, both directions, with the reference's ORIGINAL classes
o_coarse, o_fine = neuman_hip.vanilla.build_nerf(opt)
o_coarse.load_state_dict(r_coarse.state_dict(), strict=True)
r_fine.load_state_dict(o_fine.state_dict(), strict=True)
out["state_dict"] = {"keys_equal": list(r_coarse.state_dict().keys()) == list(o_coarse.state_dict().keys()),
                     "shapes_equal": all(a.shape == b.shape for a, b in zip(r_coarse.state_dict().values(), o_coarse.state_dict().values())),
                     "reference_class": f"{type(r_coarse).__module__}.{type(r_coarse).__name__}",
                     "values_round_trip": all(torch.equal(a, b) for a, b in zip(r_coarse.state_dict().values(), o_coarse.state_dict().values()))}

# HumanNeRF with per-frame body parameters (train.py:103): the reference's module (its hard-coded SMPL asset directory redirected to a
# synthetic model in SMPL's file layout; the class itself unmodified, built on the reference's ORIGINAL models.vanilla) against
# neuman_hip.human_nerf.HumanNeRF -- same state_dict keys and shapes, strict loads in both directions
import pickle  # noqa: E402
import tempfile  # noqa: E402
import numpy as np  # noqa: E402
from neuman_hip import human_nerf as our_hn, synthetic  # noqa: E402
for name in ("models.vanilla", "models.human_nerf"):
    sys.modules.pop(name, None)
ref_vanilla = importlib.import_module("models.vanilla")            # a fresh, un-rebound copy
ref_hn = importlib.import_module("models.human_nerf")
from models import smpl as ref_smpl  # noqa: E402
pose, betas, align = synthetic.smpl_like_frames(3, 0)
al = np.stack([np.concatenate([align[f'{i:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1) for i in range(3)]).astype(np.float32)
with tempfile.TemporaryDirectory() as tmp:
    with open(os.path.join(tmp, 'SMPL_NEUTRAL.pkl'), 'wb') as f:
        pickle.dump(synthetic.smpl_like_model(0), f, protocol=2)
    real_smpl = ref_smpl.SMPL
    ref_hn.SMPL = lambda path, gender='neutral', device=None: real_smpl(tmp, gender=gender, device=device)
    with contextlib.redirect_stdout(io.StringIO()):
        r_net = ref_hn.HumanNeRF(opt, pose.copy(), betas.copy(), al.copy(), scale=1.3)
        o_net = our_hn.HumanNeRF(opt, pose.copy(), betas.copy(), al.copy(), scale=1.3, smpl_dir=tmp)
rs, os_ = r_net.state_dict(), o_net.state_dict()
o_net.load_state_dict(rs, strict=True)
r_net.load_state_dict(os_, strict=True)
with torch.no_grad():
    rv, rT = r_net.vertex_forward(1)
    ov, oT = o_net.vertex_forward(1)
out["human_nerf_with_body"] = {"keys_equal": sorted(rs.keys()) == sorted(os_.keys()), "n_keys": len(rs),
                               "shapes_equal": all(rs[k].shape == os_[k].shape and rs[k].dtype == os_[k].dtype for k in rs),
                               "only_reference": sorted(set(rs) - set(os_)), "only_ours": sorted(set(os_) - set(rs)),
                               "reference_class_module": type(r_net.coarse_bkg_net).__module__,
                               "vertex_forward_linf": [float((rv - ov).abs().max()), float((rT - oT).abs().max())]}
print(json.dumps(out))
