"""LPIPS (AlexNet variant) -- the third number of the reference's evaluation triple (render_test_views.py:35-38:
`lpips.LPIPS(net='alex')(pred / 127.5 - 1, gt / 127.5 - 1)`) and the trainer's patch loss (human_nerf_trainer.py:431-435).

The metric (Zhang et al. 2018, the `lpips` package pinned by the reference's environment.yml): both images go through the five
convolution stages of an ImageNet AlexNet; at each stage the activations are normalised to unit length across channels, the squared
difference is weighted per channel by a learned non-negative vector (a 1x1 convolution), averaged over the image, and the five numbers
are added.  The WEIGHTS (torchvision's alexnet + the package's five linear heads, ~10 MB) ship with those packages, which are not
installed here and cannot be downloaded: this module is the computation, taking the weights as a state_dict in the package's own
key layout (`LPIPS.state_dict()`): "scaling_layer.shift/scale", "net.slice{1..5}.{i}.weight/bias", "lin{0..4}.model.1.weight".
**Parity unpinned** (no weights, no package to compare with); the test checks the published structure and the metric's properties on
synthetic weights.  Dense convolutions go through torch's device ops, like the reference's: this is frame egress, not the ray path.
"""
import torch
import torch.nn.functional as F

# torchvision alexnet.features: (index in `features`, out channels, kernel, stride, padding); max-pools precede stages 2 and 3
_STAGES = [(0, 64, 11, 4, 2), (3, 192, 5, 1, 2), (6, 384, 3, 1, 1), (8, 256, 3, 1, 1), (10, 256, 3, 1, 1)]
_POOL_BEFORE = (False, True, True, False, False)
SHIFT = (-.030, -.088, -.188)
SCALE = (.458, .448, .450)


class LPIPS(torch.nn.Module):
    """`LPIPS(state_dict)(in0, in1)` with images [N,3,H,W] in [-1, 1] -> [N,1,1,1], like `lpips.LPIPS(net='alex')`."""

    def __init__(self, state_dict, device=None):
        super().__init__()
        sd = {k: v.detach().float() for k, v in state_dict.items()}
        self.register_buffer('shift', sd.get('scaling_layer.shift', torch.tensor(SHIFT)[None, :, None, None]).reshape(1, 3, 1, 1))
        self.register_buffer('scale', sd.get('scaling_layer.scale', torch.tensor(SCALE)[None, :, None, None]).reshape(1, 3, 1, 1))
        for i, (fidx, ch, k, _, _) in enumerate(_STAGES):
            w = self._find(sd, i, fidx, 'weight')
            b = self._find(sd, i, fidx, 'bias')
            if w.shape[0] != ch or w.shape[2] != k:
                raise ValueError(f"LPIPS: stage {i + 1} convolution has shape {tuple(w.shape)}, AlexNet's is [{ch}, *, {k}, {k}]")
            self.register_buffer(f'w{i}', w)
            self.register_buffer(f'b{i}', b)
            lin = sd.get(f'lin{i}.model.1.weight', sd.get(f'lins.{i}.model.1.weight'))
            if lin is None:
                raise KeyError(f"LPIPS: lin{i}.model.1.weight is missing from the state_dict")
            self.register_buffer(f'lin{i}', lin.reshape(1, ch, 1, 1))
        if device is not None:
            self.to(device)

    @staticmethod
    def _find(sd, stage, fidx, what):
        for key in (f'net.slice{stage + 1}.{fidx}.{what}', f'features.{fidx}.{what}'):        # the package's layout / a bare torchvision alexnet
            if key in sd:
                return sd[key]
        raise KeyError(f"LPIPS: no {what} for AlexNet features.{fidx} (stage {stage + 1}) in the state_dict")

    def features(self, x):
        outs = []
        h = (x - self.shift) / self.scale
        for i, (_, _, _, stride, pad) in enumerate(_STAGES):
            if _POOL_BEFORE[i]:
                h = F.max_pool2d(h, kernel_size=3, stride=2)
            h = F.relu(F.conv2d(h, getattr(self, f'w{i}'), getattr(self, f'b{i}'), stride=stride, padding=pad))
            outs.append(h)
        return outs

    def forward(self, in0, in1):
        total = 0
        for i, (f0, f1) in enumerate(zip(self.features(in0), self.features(in1))):
            n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True)) + 1e-10)
            n1 = f1 / (torch.sqrt(torch.sum(f1 ** 2, dim=1, keepdim=True)) + 1e-10)
            total = total + (getattr(self, f'lin{i}') * (n0 - n1) ** 2).sum(dim=1, keepdim=True).mean(dim=(2, 3), keepdim=True)
        return total


def lpips_uint8(model, pred, gt):
    """render_test_views.py:35-38 for two uint8 [H,W,3] frames (numpy or tensors) -> float"""
    dev = model.shift.device
    a = torch.as_tensor(pred).to(dev).permute(2, 0, 1)[None].float() / 127.5 - 1
    b = torch.as_tensor(gt).to(dev).permute(2, 0, 1)[None].float() / 127.5 - 1
    with torch.no_grad():
        return float(model(a, b)[0, 0, 0, 0])
