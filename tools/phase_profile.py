"""Per-wave cycle buckets of the phase-shifted density kernel (probe build: python tools/build_variant.py prof --src mlp_phase.hip
-DNM_PHASE_PROF; run with NEUMAN_HIP_LIB=ml-neuman_amd/lib/exp/libneuman_hip_prof.so python tools/phase_profile.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "ml-neuman_amd"))
import torch
from neuman_hip import ray_utils, synthetic
dev = torch.device('cuda')
net = synthetic.make_joiner(0).to(dev)
cap = synthetic.SimpleCapture(800, 800)
o, d = ray_utils.shot_all_rays_dev(cap, dev)
R = 256 * 40                                             # 40 tiles per workgroup
z = torch.sort(torch.rand((R, 128), device=dev) * 3.14, dim=1).values.contiguous()
with torch.no_grad():
    out = net.forward_rays(o[:R].contiguous(), d[:R].contiguous(), z, precision="fp16x3", sigma_only=True)
torch.cuda.synchronize()
c = out.reshape(-1)[:256 * 8 * 8].double().reshape(256, 8, 8).cpu()
names = ["kloop", "bar_after_kloop", "convert(+store B)", "bar_after_convert", "store A", "bar_after_store/idle", "idle-slot work", "alpha"]
for grp, sl in (("A waves 0-3", slice(0, 4)), ("B waves 4-7", slice(4, 8))):
    m = c[:, sl].mean((0, 1))
    print(grp, f"total {m.sum():.3e} cycles;", "  ".join(f"{k} {v / m.sum() * 100:5.1f}%" for k, v in zip(names, m)))
