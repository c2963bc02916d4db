"""Per-wave cycle buckets of the wave-specialised i8x3 MLP kernel (nm_mlp_forward_profile, NM_PREC_I8X3).
Usage (GPU box): python tools/mlp_profile_i8.py [n_samples]"""
import ctypes
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch  # noqa: E402
from neuman_hip import _lib, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 256 * 40
j = synthetic.make_joiner(0).cuda()
g = torch.Generator(device='cuda').manual_seed(0)
pts = (torch.rand((n, 3), device='cuda', generator=g) * 3 - 1.5).contiguous()
dirs = torch.nn.functional.normalize(torch.randn((n, 3), device='cuda', generator=g), dim=-1).contiguous()
out = torch.empty((n, 4), device='cuda')
j(pts, dirs, precision="i8x3")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    j(pts, dirs, precision="i8x3")
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"i8x3: {ms:.2f} ms for {n} samples -> {n * 1186816 / ms / 1e9:.1f} TFLOP/s algorithmic")
cus = torch.cuda.get_device_properties(0).multi_processor_count
grid = min(cus, (n + 127) // 128)
cyc = torch.zeros((grid * 8, 8), device='cuda', dtype=torch.int64)
_lib.check(_lib.lib().nm_mlp_forward_profile(j.handle(), _lib.dev_ptr(pts), _lib.dev_ptr(dirs), n, _lib.NM_PREC_I8X3, _lib.dev_ptr(out),
                                             ctypes.c_void_p(cyc.data_ptr()), _lib.stream_ptr()), "profile")
torch.cuda.synchronize()
c = cyc.cpu().double().reshape(grid, 8, 8)
names = ["fill", "kloop", "M-end-bar", "epi1", "E-mid-bar", "epi2", "E-end-bar", "rest"]
tiles = (n + 127) // 128 / grid
print(f"profile build: mean counter ticks per wave {c.sum(-1).mean():.3e} over {grid} workgroups ({tiles:.1f} tiles each)")
for grp, sl in (("group A (waves 0-3)", slice(0, 4)), ("group B (waves 4-7)", slice(4, 8))):
    m = c[:, sl].mean((0, 1))
    print(grp, "  ".join(f"{k} {v / m.sum() * 100:5.1f}%" for k, v in zip(names, m)))
    print(" " * len(grp), "ticks per tile:", "  ".join(f"{k} {v / tiles:7.0f}" for k, v in zip(names, m)))
