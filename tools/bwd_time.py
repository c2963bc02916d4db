"""Time nm_mlp_backward_chain alone on the fine net's batch of a background-trainer iteration (2048 rays x 256 samples)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "ml-neuman_amd"))
import torch
from neuman_hip import _lib, synthetic
dev = torch.device('cuda')
net = synthetic.make_joiner(1).to(dev).train()
n = 2048 * 256
g = torch.Generator(device='cuda').manual_seed(1)
dz = torch.randn((n, 256), device=dev, generator=g) * 1e-3
acts = torch.randn((9, n, 256), device=dev, generator=g)
bits = torch.randint(-2**31, 2**31 - 1, (8, n, 8), device=dev, dtype=torch.int32, generator=g)
out = torch.empty((7, n, 256), device=dev)
gb = torch.empty((7, 256), device=dev)
ws = torch.empty(int(_lib.lib().nm_mlp_backward_chain_workspace_floats(n)), device=dev)
ptrs = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in net.nerf.ordered_params()])
h = net.train_handle()
ms = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().nm_mlp_backward_chain(h, ptrs, _lib.dev_ptr(dz), None, None, _lib.dev_ptr(acts), ctypes.c_void_p(bits.data_ptr() if os.environ.get('BWD_BITS', '1') == '1' else 0), n, _lib.dev_ptr(out), _lib.dev_ptr(gb), _lib.dev_ptr(ws), ws.numel(), _lib.stream_ptr()), "chain")
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print(f"{os.environ.get('NEUMAN_HIP_LIB', 'tree').split('/')[-1]:34s} backward chain, {n} samples: {min(ms):.3f} ms  ({7 * 2 * 256 * 256 * n / min(ms) / 1e9:.0f} TFLOP/s)  all: {' '.join(f'{m:.2f}' for m in ms)}")
