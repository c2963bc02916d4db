"""Per-wave cycle buckets of the bf16x3 MLP kernel (nm_mlp_forward_profile) + plain timing of the three precisions.
Usage (GPU box): python tools/mlp_profile.py [n_samples]"""
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
for prec in ("bf16x3", "bf16"):
    j(pts, dirs, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        j(pts, dirs, precision=prec)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{prec}: {ms:.2f} ms for {n} samples -> {n * 1186816 / ms / 1e9:.1f} TFLOP/s algorithmic")
cus = torch.cuda.get_device_properties(0).multi_processor_count
grid = min(cus, (n + 127) // 128)
cyc = torch.zeros((grid * 8, 8), device='cuda', dtype=torch.int64)
_lib.check(_lib.lib().nm_mlp_forward_profile(j.handle(), _lib.dev_ptr(pts), _lib.dev_ptr(dirs), n, _lib.NM_PREC_BF16X3, _lib.dev_ptr(out),
                                             ctypes.c_void_p(cyc.data_ptr()), _lib.stream_ptr()), "profile")
torch.cuda.synchronize()
c = cyc.cpu().double().reshape(grid, 8, 8)[:, :, :6]
names = ["pe", "kloop", "wait_pre", "epilogue", "wait_post", "tail"]
tot = c.sum(-1).mean()
print(f"profile build: mean cycles per wave {tot:.3e} over {grid} workgroups ({(n + 127) // 128 / grid:.1f} tiles each)")
for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    m = c[:, sl].mean((0, 1))
    print(grp, "  ".join(f"{k} {v / m.sum() * 100:5.1f}%" for k, v in zip(names, m)))
